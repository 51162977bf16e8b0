"""Socket power / shader clock telemetry (hwmon, ~20 Hz) across ~8 s loops of one kernel each: the split-operand stack kernel of the headline
loop, the fused bf16 layer groups (64- and 128-frame tiles), and a bare-MFMA ceiling kernel (tools/hw/mfma_ceiling).  -> one JSON line per
kernel: mean / max power, mean sclk, time per launch in the sustained loop.  usage: python tools/power_probe.py [x3|bf64|bf128|idle ...]"""
import ctypes as C, glob, json, os, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import set_amd  # noqa
from set_amd import _lib

dev = torch.device("cuda:0")


def hwmon_files():
    """hwmon files of the DRM card that IS HIP device 0 (matched by PCI address: the box shows every GPU of the node in sysfs)."""
    pr = torch.cuda.get_device_properties(0)
    pci = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
    out = {"pci": pci}
    for card in sorted(glob.glob("/sys/class/drm/card*")):
        if not os.path.basename(card).replace("card", "").isdigit():
            continue
        real = os.path.realpath(os.path.join(card, "device"))
        if pci not in real:
            continue
        out["card"] = card
        for h in glob.glob(os.path.join(card, "device", "hwmon", "hwmon*")):
            for name in ("power1_average", "power1_input", "freq1_input", "temp2_input", "power1_cap"):
                p = os.path.join(h, name)
                if os.path.exists(p) and name not in out:
                    out[name] = p
    return out


class Sampler(threading.Thread):
    def __init__(self, files, hz=20.0):
        super().__init__(daemon=True)
        self.files, self.dt, self.rows, self.stop_flag = files, 1.0 / hz, [], False

    def run(self):
        while not self.stop_flag:
            row = {"t": time.time()}
            for k, p in self.files.items():
                if k in ("pci", "card"):
                    continue
                try:
                    row[k] = float(open(p).read().strip())
                except Exception:
                    pass
            self.rows.append(row)
            time.sleep(self.dt)


def smi_snapshot():
    import subprocess
    try:
        return subprocess.run(["/opt/rocm/bin/rocm-smi", "-P", "-c", "--showmaxpower"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=20).stdout[-1500:]
    except Exception as e:
        return "rocm-smi failed: %r" % (e,)


def make_bf16_layers(tile, nl):
    L = _lib.lib()
    B, T = 32, 800
    n = L.set_diffnet_layer_bf16_image_size()
    imgs = torch.empty(nl, n, dtype=torch.bfloat16, device=dev)
    for l in range(nl):
        wd, wc, wo = torch.randn(512, 256, 3, device=dev) * 0.03, torch.randn(512, 192, device=dev) * 0.05, torch.randn(512, 256, device=dev) * 0.05
        _lib.check(L.set_pack_diffnet_layer_bf16(wd.data_ptr(), wc.data_ptr(), wo.data_ptr(), imgs[l].data_ptr(), None), "pack")
    x, xo, sk = (torch.randn(B, 256, T, device=dev) for _ in range(3))
    cond = torch.randn(B, 192, T, device=dev)
    dst = torch.randn(nl, B, 256, device=dev)
    bias = [torch.zeros(nl, 512, device=dev) for _ in range(3)]
    n_ws = L.set_diffnet_layers_bf16_scratch_floats(B, T, 0, nl, 1)
    ws = torch.empty(max(64, n_ws), device=dev)
    a = _lib.SetDiffnetLayersBf16Args()
    a.x_in, a.x_out, a.skip, a.cond, a.dstep, a.img = x.data_ptr(), xo.data_ptr(), sk.data_ptr(), cond.data_ptr(), dst.data_ptr(), imgs.data_ptr()
    a.b_dil, a.b_cond, a.b_out = (b.data_ptr() for b in bias)
    a.scratch, a.scratch_floats = ws.data_ptr(), ws.numel()
    a.d_bs, a.d_cs, a.d_ls, a.B, a.T, a.l0, a.nl, a.dilation_cycle_length, a.first = 256, 1, B * 256, B, T, 0, nl, 1, 1
    os.environ["SET_AMD_BF16_FUSE_TILE"] = str(tile)
    keep = (imgs, x, xo, sk, cond, dst, bias, ws)

    def launch():
        _lib.check(L.set_diffnet_layers_fwd_bf16(C.byref(a), None), "layers")
    return launch, keep, nl


def make_x3():
    """The split-operand stack kernel at B = 32, T = 800 (setup as tools/x3_pair_probe.py)."""
    from set_amd import ops
    L = 20
    g = torch.Generator().manual_seed(1)
    w1 = torch.empty(L, 512 * 768, device=dev); w2 = torch.empty(L, 512 * 256, device=dev)
    wx3 = ops.SplitOperandImages(L, ops.split_operand_mode(), dev)
    for l in range(L):
        wd, wo = (torch.randn(512, 256, 3, generator=g) / 27.7).to(dev), (torch.randn(512, 256, 1, generator=g) / 16).to(dev)
        ops.pack_diffnet_layer(wd, wo, w1[l], w2[l]); wx3.pack(l, wd, wo)
    bd = torch.zeros(L, 512, device=dev); bo = torch.zeros(L, 512, device=dev)
    packs = (w1, w2, bd, bo, None, None, None, None, wx3)
    os.environ["SET_AMD_X3"] = "2"
    B, T = 32, 800
    x0 = torch.randn(B, 256, T, device=dev); cp = torch.randn(B, L * 512, T, device=dev) * 0.5
    dtab = torch.randn(L * 256, 100, device=dev)
    xa, xb, skip = x0.clone(), torch.empty_like(x0), torch.empty_like(x0)

    def launch():
        xa.copy_(x0)
        ops.diffnet_stack(xa, xb, skip, cp, dtab.data_ptr(), 0, 100, 256 * 100, packs, 1)
    return launch, (w1, w2, wx3, bd, bo, x0, cp, dtab, xa, xb, skip), L


def run(name, launch, layers, seconds=8.0):
    files = hwmon_files()
    for _ in range(5):
        launch()
    torch.cuda.synchronize()
    s = Sampler(files)
    s.start()
    time.sleep(0.5)
    idle_rows = len(s.rows)
    t0 = time.time()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < seconds:
        for _ in range(50):
            launch()
        n += 50
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    busy_rows = len(s.rows)
    time.sleep(0.3)
    s.stop_flag = True
    s.join()
    rows = s.rows[idle_rows + 5:busy_rows]  # skip the ramp
    pk = "power1_average" if "power1_average" in files else ("power1_input" if "power1_input" in files else None)
    out = {"kernel": name, "launches": n, "ms_per_launch_sustained": e0.elapsed_time(e1) / max(1, n), "layers_per_launch": layers, "samples": len(rows)}
    if pk and rows:
        pw = [r[pk] / 1e6 for r in rows if pk in r]
        out.update(power_W_mean=sum(pw) / len(pw), power_W_max=max(pw), power_W_idle=(s.rows[0].get(pk, 0) / 1e6))
    if "freq1_input" in files and rows:
        fq = [r["freq1_input"] / 1e9 for r in rows if "freq1_input" in r]
        out.update(sclk_GHz_mean=sum(fq) / len(fq), sclk_GHz_min=min(fq), sclk_GHz_max=max(fq))
    if "power1_cap" in files:
        out["power_cap_W"] = float(open(files["power1_cap"]).read()) / 1e6
    if "temp2_input" in files and rows:
        out["temp_C_max"] = max(r.get("temp2_input", 0) for r in rows) / 1e3
    print(json.dumps(out), flush=True)
    return out


if __name__ == "__main__":
    which = sys.argv[1:] or ["bf64", "bf128"]
    print(json.dumps({"hwmon": hwmon_files()}), flush=True)
    print(smi_snapshot(), flush=True)
    for wname in which:
        if wname == "bf64":
            l, keep, nl = make_bf16_layers(64, 5); run("bf16 layer groups, 64-frame tiles, 5 layers per launch", l, nl)
        elif wname == "bf128":
            l, keep, nl = make_bf16_layers(128, 10); run("bf16 layer groups, 128-frame tiles, 10 layers per launch", l, nl)
        elif wname == "x3":
            l, keep, nl = make_x3(); run("split-operand stack kernel (headline loop), 20 layers per launch", l, nl)
        elif wname == "ceiling":  # tools/hw/mfma_ceiling (bare MFMA / + LDS fragments / + L2 fragments) under the same telemetry
            import subprocess
            exe = os.path.join(ROOT, "tools", "hw", "mfma_ceiling")
            for mode in (0, 1, 2):
                files = hwmon_files()
                smp = Sampler(files)
                smp.start()
                t0 = time.time()
                r = subprocess.run([exe, "6", str(mode)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                t1 = time.time()
                smp.stop_flag = True
                smp.join()
                rows = [x for x in smp.rows if x["t"] > t0 + 0.5 * (t1 - t0)]  # the second half: clocks settled, kernel running
                pw = [x["power1_input"] / 1e6 for x in rows if "power1_input" in x]
                fq = [x["freq1_input"] / 1e9 for x in rows if "freq1_input" in x]
                print(json.dumps({"kernel": "mfma_ceiling mode %d" % mode, "stdout": r.stdout.strip(), "samples": len(rows),
                                  "power_W_mean": sum(pw) / max(1, len(pw)), "power_W_max": max(pw) if pw else None,
                                  "sclk_GHz_mean": sum(fq) / max(1, len(fq))}), flush=True)
        elif wname == "idle":
            run("idle (no launches)", lambda: time.sleep(0.001), 0, seconds=3.0)
