"""CPU restatement of the operand splitting of csrc/diffnet_x3.hip / csrc/conv_x2.hip (numpy, no GPU): an fp32 value is the sum
of two fp16 (three bf16) pieces; a product needs three (six) piece products; accumulation is fp32.  The claims the kernels'
documentation makes -- fp32-equivalent accuracy, harmless subnormal residuals, the power-of-two weight scale, the range
limit -- checked on the arithmetic itself, independent of the hardware tests in tests/test_gpu_parity.py."""
import numpy as np


def split_f16x2(a):
    a0 = a.astype(np.float16)
    a1 = (a - a0.astype(np.float32)).astype(np.float16)
    return a0.astype(np.float32), a1.astype(np.float32)


def to_bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000  # round to nearest even
    return u.astype(np.uint32).view(np.float32)


def split_bf16x3(a):
    a0 = to_bf16(a)
    r1 = (a - a0).astype(np.float32)
    a1 = to_bf16(r1)
    a2 = to_bf16((r1 - a1).astype(np.float32))
    return a0, a1, a2


def gemm_f32_chain(a, b):
    """fp32 FMA-free chain: one rounding per product and per add, k in order (the fp32 MFMA kernels' association)."""
    acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
    for k in range(a.shape[1]):
        acc = (acc + (a[:, k:k + 1] * b[k:k + 1, :]).astype(np.float32)).astype(np.float32)
    return acc


def gemm_split(pa, pb, pairs, kstep=16):
    """Piece products exact (fp64 here = exact for 11 x 11 / 8 x 8 bit mantissas), 16 of them summed per 'MFMA' and added to
    the fp32 accumulator, smallest terms first -- the kernels' order."""
    M, K = pa[0].shape
    acc = np.zeros((M, pb[0].shape[1]), np.float32)
    for k0 in range(0, K, kstep):
        for (qa, qb) in pairs:
            part = pa[qa][:, k0:k0 + kstep].astype(np.float64) @ pb[qb][k0:k0 + kstep, :].astype(np.float64)
            acc = (acc.astype(np.float64) + part).astype(np.float32)
    return acc


F16X2 = ((1, 0), (0, 1), (0, 0))
BF16X3 = ((2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0))


def test_pieces_reconstruct_the_fp32_value():
    rng = np.random.default_rng(0)
    a = (rng.standard_normal(100000) * np.exp(rng.uniform(-8, 8, 100000))).astype(np.float32)
    a0, a1 = split_f16x2(a[np.abs(a) < 6.0e4])
    x = a[np.abs(a) < 6.0e4]
    err = np.abs(x.astype(np.float64) - a0.astype(np.float64) - a1.astype(np.float64))
    assert np.all(err <= np.maximum(2.0 ** -22 * np.abs(x), 2.0 ** -25))  # 22 bits, or the fp16 subnormal floor
    b0, b1, b2 = split_bf16x3(a)
    errb = np.abs(a.astype(np.float64) - b0.astype(np.float64) - b1.astype(np.float64) - b2.astype(np.float64))
    assert np.all(errb <= 2.0 ** -24 * np.abs(a))                            # 24 bits at any magnitude


def test_split_gemm_is_as_accurate_as_the_fp32_chain():
    rng = np.random.default_rng(1)
    M, K, N = 64, 768, 48
    a = (rng.standard_normal((M, K)) * 0.05).astype(np.float32)   # weights of the DiffNet layers' magnitude
    b = rng.standard_normal((K, N)).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    e32 = np.abs(gemm_f32_chain(a, b) - ref).max()
    scale = np.float32(2.0 ** (4 - np.frexp(np.abs(a).max())[1]))           # the pack-time power of two: max|w| 2^k in [8, 16)
    assert 8.0 <= np.abs(a).max() * scale < 16.0
    e2 = np.abs(gemm_split(split_f16x2(a * scale), split_f16x2(b), F16X2) / scale - ref).max()
    e3 = np.abs(gemm_split(split_bf16x3(a), split_bf16x3(b), BF16X3) - ref).max()
    assert e2 < 1.5 * e32 and e3 < e32, (e32, e2, e3)


def test_small_weights_need_the_power_of_two_scale():
    """|w| ~ 1e-4: the fp16 residual piece would be subnormal (absolute precision 6e-8 = 2^-11 of such a weight) -- the scale
    moves it back into the normal range and the scaled splitting is again 22 bits wide."""
    rng = np.random.default_rng(2)
    w = (rng.standard_normal(20000) * 1e-4).astype(np.float32)
    w0, w1 = split_f16x2(w)
    rel_unscaled = (np.abs(w.astype(np.float64) - w0 - w1) / np.abs(w)).max()
    k = 4 - np.frexp(np.abs(w).max())[1]
    s = np.float32(2.0 ** k)
    s0, s1 = split_f16x2(w * s)
    big = np.abs(w * s) >= 2.0 ** -3                                        # residual piece normal from here on
    rel_scaled = (np.abs((w * s).astype(np.float64) - s0 - s1) / np.abs(w * s))[big].max()
    assert rel_unscaled > 2.0 ** -17 and rel_scaled <= 2.0 ** -22


def test_range_limit_of_the_fp16_splitting():
    with np.errstate(over="ignore"):
        hi, _ = split_f16x2(np.array([7.0e4], np.float32))
    assert np.isinf(hi[0])                                                  # what the kernels' guard (|x| >= 32768) prevents
    ok0, ok1 = split_f16x2(np.array([32767.0], np.float32))
    assert np.isfinite(ok0[0]) and abs(32767.0 - ok0[0] - ok1[0]) <= 2.0 ** -22 * 32767.0
    b0, b1, b2 = split_bf16x3(np.array([3.0e38], np.float32))
    assert np.isfinite(b0[0]) and abs(3.0e38 - float(b0[0]) - float(b1[0]) - float(b2[0])) <= 2.0 ** -24 * 3.0e38
