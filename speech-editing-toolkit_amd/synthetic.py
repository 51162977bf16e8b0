"""Synthetic [B, 80, T] batches of the benchmark shape (numpy default_rng(seed)):
txt_tokens U{1..79}, sorted mel2ph in {1..T_txt}, clipped-normal log-mels, one contiguous 30-80 % edit mask,
log2-Hz f0, Bernoulli(0.3) uv, N(0,1)/16 speaker embeddings.  Used by bench.py and smoke()."""
import numpy as np
import torch


def synthetic_inputs(B, T, T_txt, seed=1234, pad_tail=False, n_tokens=80):
    rng = np.random.default_rng(seed)
    txt = rng.integers(1, n_tokens, size=(B, T_txt), dtype=np.int64)
    mel2ph = np.sort(rng.integers(1, T_txt + 1, size=(B, T), dtype=np.int64), axis=1)
    ref = np.clip(rng.normal(-3.0, 1.5, size=(B, T, 80)), -6.0, 1.5).astype(np.float32)
    mask = np.zeros((B, T, 1), dtype=np.float32)
    for b in range(B):
        frac = rng.uniform(0.3, 0.8)
        n = max(1, int(round(frac * T)))
        s = int(rng.integers(0, T - n + 1))
        mask[b, s:s + n] = 1.0
    f0 = rng.uniform(6.5, 9.2, size=(B, T)).astype(np.float32)
    uv = (rng.uniform(size=(B, T)) < 0.3).astype(np.float32)
    spk = (rng.standard_normal(size=(B, 256)) / 16.0).astype(np.float32)
    if pad_tail:
        npad = max(1, T // 10)
        for b in range(B):
            k = npad if b % 2 == 0 else npad // 2
            if k > 0:
                mel2ph[b, T - k:] = 0
                ref[b, T - k:] = 0.0
                f0[b, T - k:] = 0.0
                uv[b, T - k:] = 0.0
                mask[b, T - k:] = 0.0
    return {
        "txt_tokens": torch.from_numpy(txt), "mel2ph": torch.from_numpy(mel2ph), "ref_mels": torch.from_numpy(ref),
        "time_mel_masks": torch.from_numpy(mask), "f0": torch.from_numpy(f0), "uv": torch.from_numpy(uv),
        "spk_embed": torch.from_numpy(spk),
    }
