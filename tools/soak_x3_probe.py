"""Soak of the round-2 persistent kernels (split-operand throughput kernel with 64- and 32-frame tiles, both splittings, and
the row-split small-batch kernels): N launches per shape, every result compared bit for bit with the first, no time-outs."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import set_amd  # noqa
from set_amd import ops
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
N = int(os.environ.get("SOAK_N", 300))
# B, T, L, dcl, x3 mode, env
SHAPES = ((32, 800, 20, 1, 2, dict(SET_AMD_X3="2", SET_AMD_SPLIT="0")), (24, 797, 20, 1, 2, dict(SET_AMD_X3="2", SET_AMD_SPLIT="0")),
          # round 6: the Winograd form on 64-frame tiles (no 96-frame chain per CU) and on 96-frame tiles whose last block is partial
          (24, 800, 20, 1, 2, dict(SET_AMD_X3="2", SET_AMD_SPLIT="0")), (33, 802, 20, 1, 2, dict(SET_AMD_X3="2", SET_AMD_SPLIT="0")),
          (32, 800, 8, 4, 3, dict(SET_AMD_X3="2", SET_AMD_SPLIT="0")), (8, 800, 20, 1, 2, dict(SET_AMD_X3="2", SET_AMD_SPLIT="0")),
          (7, 1548, 20, 1, 2, dict(SET_AMD_X3="2", SET_AMD_SPLIT="0", SET_AMD_X3_TILE="32")),
          (1, 800, 20, 1, 2, dict(SET_AMD_SPLIT="2")), (2, 800, 20, 1, 2, dict(SET_AMD_SPLIT="2")),
          (3, 203, 20, 3, 2, dict(SET_AMD_SPLIT="2")), (2, 800, 20, 1, 2, dict(SET_AMD_SPLIT="2", SET_AMD_SPLIT_F32="1")))
for (B, T, L, dcl, mode, env) in SHAPES:
    g = torch.Generator().manual_seed(B * 7 + T)
    x0 = torch.randn(B, 256, T, generator=g).to(dev)
    cp = (torch.randn(B, L * 512, T, generator=g) * 0.5).to(dev)
    dtab = torch.randn(L * 256, 1, generator=g).to(dev)
    w1 = torch.empty(L, 512 * 768, device=dev); w2 = torch.empty(L, 512 * 256, device=dev)
    bd = (torch.randn(L, 512, generator=g) * 0.1).to(dev); bo = (torch.randn(L, 512, generator=g) * 0.1).to(dev)
    wx3 = ops.SplitOperandImages(L, mode, dev)
    for l in range(L):
        wd = (torch.randn(512, 256, 3, generator=g) / 27.7).to(dev); wo = (torch.randn(512, 256, 1, generator=g) / 16.0).to(dev)
        ops.pack_diffnet_layer(wd, wo, w1[l], w2[l]); wx3.pack(l, wd, wo)
    packs = (w1, w2, bd, bo, None, None) + ops.split_images(w1, w2) + (wx3,)
    for k in ("SET_AMD_X3", "SET_AMD_SPLIT", "SET_AMD_X3_TILE", "SET_AMD_SPLIT_F32"):
        os.environ.pop(k, None)
    os.environ.update(env)
    variant = ops.stack_variant(B, T, dcl, x3_mode=mode)
    ref, bad, aborts = None, 0, 0
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    for it in range(N):
        xa, xb, skip = x0.clone(), torch.empty_like(x0), torch.empty_like(x0)
        ws = ops.diffnet_stack(xa, xb, skip, cp, dtab.data_ptr(), 0, 1, 256, packs, dcl, err_flag=err)
        out = xa if L % 2 == 0 else xb
        if ref is None:
            ref = (out.clone(), skip.clone())
        else:
            bad += int(not (torch.equal(out, ref[0]) and torch.equal(skip, ref[1])))
        if it % 50 == 0:
            aborts += int(ws[1])
    print("B=%d T=%d L=%d dcl=%d mode %d %s -> %s: %d launches, %d mismatches, aborts %d, err word %d" % (
        B, T, L, dcl, mode, env, ops.STACK_VARIANT_NAMES[variant], N, bad, aborts, int(err)))
