"""Round 6: tools/hw/mfma_ceiling_w (operand-stream models of the Winograd GEMM 1 at 64- / 96- / 128-frame tiles) mode by mode, socket power and
shader clock sampled over each run.  usage: python tools/ceiling_w_probe.py [seconds per mode]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from power_probe import Sampler, hwmon_files  # noqa: E402

secs = sys.argv[1] if len(sys.argv) > 1 else "6"
exe = os.path.join(ROOT, "build", "exp", "mfma_ceiling_w")
files = hwmon_files()
for mode in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("0", "1", "2", "3", "4", "2", "3")):
    smp = Sampler(files)
    smp.start()
    r = subprocess.run([exe, secs, mode], capture_output=True, text=True, timeout=120)
    smp.stop_flag = True
    smp.join()
    rows = smp.rows[len(smp.rows) // 3:]
    pw = [x.get("power1_average", x.get("power1_input", 0.0)) / 1e6 for x in rows]
    fq = [x.get("freq1_input", 0.0) / 1e6 for x in rows]
    print(r.stdout.strip(), "| power %.0f W mean, sclk %.0f MHz mean" % (sum(pw) / max(1, len(pw)), sum(fq) / max(1, len(fq))), flush=True)
    if r.returncode:
        print(r.stderr[-400:])
