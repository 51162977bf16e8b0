"""Fused attention kernels at the CampNet shapes (B=16, 2 heads x 96, T=800, T_txt=100): us per launch, forward and backward,
fp32 and bf16 operands, against the three-launch composition (bmm -> softmax -> bmm)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import set_amd  # noqa
from set_amd import ops
dev = torch.device("cuda:0")
MV = ops.MatView
B, heads, d = 16, 2, 96
H = heads * d


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


for name, Tq, Tk in (("self 800x800", 800, 800), ("cross 800x100", 800, 100), ("enc 100x100", 100, 100)):
    q = torch.randn(B, H, Tq, device=dev); kv = torch.randn(B, 2 * H, Tk, device=dev); do = torch.randn(B, H, Tq, device=dev)
    views = (MV.heads(q, heads), MV.heads(kv, heads, 0, H), MV.heads(kv, heads, H, H))
    dq, dkv = torch.empty_like(q), torch.empty_like(kv)
    dviews = (MV.heads(dq, heads), MV.heads(dkv, heads, 0, H), MV.heads(dkv, heads, H, H))
    for dt in ("bf16", "f32"):
        ops.set_compute_dtype(dt)
        o, lse, _ = ops.attention_fused(*views, heads, None, float("-inf"), d ** -0.5)
        f = timed(lambda: ops.attention_fused(*views, heads, None, float("-inf"), d ** -0.5))
        bw = timed(lambda: ops.attention_fused_bwd(*views, o, lse, do, *dviews, heads, None, float("-inf"), d ** -0.5))
        ops.set_compute_dtype("f32")
        print("%-14s %-4s fused fwd %7.1f us  bwd %7.1f us" % (name, dt, f, bw), flush=True)
    fc = timed(lambda: ops.attention_views(*views, heads, None, float("-inf"), d ** -0.5))
    print("%-14s f32  composition fwd %7.1f us" % (name, fc), flush=True)
