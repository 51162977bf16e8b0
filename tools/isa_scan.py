"""Static checks on the gfx950 ISA of the kernels (no GPU needed): the compiler pathologies of DESIGN 3.5b / 3.5c.

    python tools/isa_scan.py                 # every csrc/*.hip: kernels whose MFMAs sit behind a drain, kernels whose stores wait for each other
    python tools/isa_scan.py bf16.hip conv1d_wgrad3_bf16_kernel     # one-character-per-instruction trace of the kernels matching the name

(1) `v_mfma` with `s_waitcnt vmcnt(0|1)` among the three instructions in front of it: a weight-fragment ring that is drained every k-step
    instead of being pipelined (`a = A[p]; A[p] = load(); mfma(a)` idiom, flat accesses near the loop, prefetch values consumed at issue);
(2) basic blocks holding one or two stores behind `s_waitcnt vmcnt(0)`: `if (valid) store` per element, or load -> store -> load chains
    (vmcnt retires loads and stores in issue order on gfx9).
Trace legend: L load, S store, M MFMA, r / w LDS read / write, (vN) s_waitcnt vmcnt(N), |B| barrier, ^ branch; one line per basic block."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "speech-editing-toolkit_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only"]


def asm_of(src):
    out = os.path.join(tempfile.gettempdir(), "isa_scan_" + os.path.basename(src) + ".s")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.run(["hipcc"] + FLAGS + ["-o", out, src], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def kernels(lines):
    starts = [(i, l) for i, l in enumerate(lines) if re.match(r"^_Z[A-Za-z0-9_]*:|^[a-z][a-z_0-9]*kernel[A-Za-z0-9_]*:", l)]
    for n, (i, l) in enumerate(starts):
        yield l.split(":")[0], lines[i:(starts[n + 1][0] if n + 1 < len(starts) else len(lines))]


# (3) the store-data hazard of round 5: a buffer store of more than 64 bits whose soffset is an SGPR, followed IMMEDIATELY by a
#     VALU instruction that writes one of its data registers.  ROCm 7.2 pads that pair with a wait state only when soffset is not a register;
#     on gfx950 the unpadded pair sometimes stores the overwritten dword (run-to-run differences in conv1d_mfma_v2_kernel's outputs).
WIDE_STORE = re.compile(r"(buffer_store_dwordx[34]|buffer_store_format_xyzw?|buffer_store_format_d16_xyzw) v\[(\d+):(\d+)\], "
                        r"(?:v\d+|off|v\[\d+:\d+\]), s\[\d+:\d+\], (\S+)")


def store_data_hazards(body):
    """[(store, next instruction)] pairs of one kernel body that match (3)."""
    out = []
    ins = [b.strip() for b in body if b.strip() and not b.strip().startswith((".", ";"))]
    for k, t in enumerate(ins[:-1]):
        m = WIDE_STORE.match(t)
        if not m or not re.match(r"s\d+", m.group(4)):
            continue
        lo, hi, n = int(m.group(2)), int(m.group(3)), ins[k + 1]
        if not n.startswith("v_"):
            continue
        w, wr = re.match(r"v_\w+ v(\d+)", n), re.match(r"v_\w+ v\[(\d+):(\d+)\]", n)
        if (w and lo <= int(w.group(1)) <= hi) or (wr and not (int(wr.group(2)) < lo or int(wr.group(1)) > hi)):
            out.append((t, n))
    return out


def scan(src):
    for name, body in kernels(asm_of(src)):
        nm = sum("v_mfma" in b for b in body)
        bad = sum(1 for k, b in enumerate(body) if "v_mfma" in b and re.search(r"vmcnt\((0|1)\)", " ".join(body[max(0, k - 3):k])))
        blocks, cur = [], []
        for b in body:
            t = b.strip()
            if t.startswith(".LBB") or t.startswith("; %bb."):
                blocks.append(cur); cur = []
            cur.append(t)
        blocks.append(cur)
        ser = sum(1 for blk in blocks if 1 <= sum(x.startswith(("buffer_store", "global_store", "flat_store")) for x in blk) <= 2
                  and any(re.search(r"vmcnt\(0\)", x) for x in blk))
        hz = store_data_hazards(body)
        if hz:
            print("%-18s %-90s %d wide stores with an SGPR soffset whose data registers the next VALU instruction rewrites, e.g. %s ; %s" % (
                os.path.basename(src), name[:90], len(hz), hz[0][0][:60], hz[0][1][:40]))
        if (nm and bad * 5 >= nm) or ser >= 4:
            print("%-18s %-90s MFMAs %4d, behind vmcnt(0|1): %4d | store blocks behind vmcnt(0): %d" % (os.path.basename(src), name[:90], nm, bad, ser))


def trace(src, pat):
    for name, body in kernels(asm_of(src)):
        if pat not in name:
            continue
        out = []
        for b in body:
            t = b.strip()
            if t.startswith(".LBB"): out.append("\n" + t.split(":")[0] + ":")
            elif t.startswith("v_mfma"): out.append("M")
            elif t.startswith(("buffer_load", "global_load", "flat_load")): out.append("L")
            elif t.startswith(("buffer_store", "global_store", "flat_store")): out.append("S")
            elif t.startswith("ds_read"): out.append("r")
            elif t.startswith("ds_write"): out.append("w")
            elif t.startswith("s_barrier"): out.append("|B|")
            elif t.startswith("s_waitcnt"):
                m = re.search(r"vmcnt\((\d+)\)", t)
                if m: out.append("(v%s)" % m.group(1))
            elif t.startswith(("s_cbranch", "s_branch")): out.append("^")
        s = re.sub(r"\n\.LBB\d+_\d+:\^*(?=\n|$)", "", "".join(out))
        print(name); print(s)


if __name__ == "__main__":
    if len(sys.argv) >= 3:
        trace(os.path.join(CS, sys.argv[1]), sys.argv[2])
    else:
        for f in sorted(os.listdir(CS)):
            if f.endswith(".hip"):
                scan(os.path.join(CS, f))
