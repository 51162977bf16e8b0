"""Deterministic synthetic weights + inputs shared by the golden generator, the
parity tests and bench.py's cpu_baseline leg.

TEST INFRASTRUCTURE (see oracle/oracle.py header).  Weights are NOT committed
(the DiffNet alone is 57.7 MB): both sides regenerate them from
numpy.random.default_rng(seed) walking a manifest of (state_dict key, shape)
in order, with the per-tensor scales documented below.  The manifest for the
reference model is committed as tests/golden/manifest_*.json (generated from
the reference's own state_dict by oracle/make_golden.py).
"""
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

_BUFFER_KEYS = {
    "timesteps", "timescale", "betas", "alphas_cumprod", "alphas_cumprod_prev",
    "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
    "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
    "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2",
    "spec_min", "spec_max",
}


def load_manifest(name):
    with open(os.path.join(GOLDEN_DIR, "manifest_%s.json" % name)) as f:
        return [(k, tuple(s)) for k, s in json.load(f)]


def is_buffer(key):
    return key in _BUFFER_KEYS


def _is_norm_weight(key, shape, manifest_shapes):
    # LayerNorm affine weight: 1-D "weight" whose sibling is not a conv/linear
    # (conv/linear weights are >= 2-D, so any 1-D *.weight is a norm scale).
    return key.endswith(".weight") and len(shape) == 1


def seeded_weights(manifest, seed, dtype=torch.float32):
    """{key: tensor} for every non-buffer key of `manifest`, in manifest order.

    scales:  >=2-D weight  -> N(0,1)/sqrt(fan_in)    (fan_in = prod(shape[1:]))
             weight_g      -> |N(0,1)|*0.5 + 0.5     (weight-norm gain, [C,1,1])
             weight_v      -> N(0,1)/sqrt(fan_in)
             embeddings    -> N(0,1)/sqrt(dim)       (2-D, handled by the rule above w/ fan_in=dim)
             1-D weight    -> 1 + 0.1 N(0,1)         (LayerNorm scale)
             bias          -> 0.1 N(0,1)
    """
    rng = np.random.default_rng(seed)
    out = {}
    for key, shape in manifest:
        if is_buffer(key):
            continue
        z = rng.standard_normal(size=shape, dtype=np.float32) if len(shape) else np.float32(rng.standard_normal())
        if key.endswith("weight_g"):
            w = np.abs(z) * 0.5 + 0.5
        elif key.endswith(".bias"):
            w = 0.1 * z
        elif len(shape) == 1:
            w = 1.0 + 0.1 * z
        else:
            fan_in = int(np.prod(shape[1:]))
            w = z / np.sqrt(max(fan_in, 1))
        out[key] = torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32)).to(dtype)
    return out


def synthetic_inputs(B, T, T_txt, seed=1234, pad_tail=False, n_tokens=80):
    """BASELINE.md section 3 synthetic batch (numpy default_rng(seed))."""
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
        "txt_tokens": torch.from_numpy(txt),
        "mel2ph": torch.from_numpy(mel2ph),
        "ref_mels": torch.from_numpy(ref),
        "time_mel_masks": torch.from_numpy(mask),
        "f0": torch.from_numpy(f0),
        "uv": torch.from_numpy(uv),
        "spk_embed": torch.from_numpy(spk),
    }


def synthetic_edit_sample(seed, n_words=6, region=(3, 4), n_edited_words=3, max_ph=3, max_dur=6, n_tokens=80):
    """One utterance + an edit request in the batch format of the reference's
    SpecDenoiserInfer.input_to_batch (inference/tts/spec_denoiser.py:198-248), synthetic:
    words 1..n_words with 1..max_ph phonemes each and 1..max_dur frames per phoneme, a trailing
    zero-duration end token that belongs to the last word (so the reference's `+2` re-indexing of the
    tail, :110, stays inside the edited token sequence); words region[0]..region[1] (1-based,
    inclusive) are replaced by `n_edited_words` new words.  numpy default_rng(seed)."""
    rng = np.random.default_rng(seed)
    w0, w1 = region
    assert 1 <= w0 <= w1 <= n_words

    def phones_of(nw):
        return [int(rng.integers(1, max_ph + 1)) for _ in range(nw)]

    cnt = phones_of(n_words)
    ph2word = [w + 1 for w, c in enumerate(cnt) for _ in range(c)] + [n_words]
    dur = [int(rng.integers(1, max_dur + 1)) for _ in range(sum(cnt))] + [0]
    mel2ph = [i + 1 for i, d in enumerate(dur) for _ in range(d)]
    mel2word = [ph2word[p - 1] for p in mel2ph]
    T = len(mel2ph)
    ecnt = cnt[:w0 - 1] + phones_of(n_edited_words) + cnt[w1:]
    n_ew = len(ecnt)
    edited_ph2word = [w + 1 for w, c in enumerate(ecnt) for _ in range(c)] + [n_ew]
    c0, c1 = w0, w0 + n_edited_words - 1
    Te = len(edited_ph2word)
    mel = np.clip(rng.normal(-3.0, 1.5, size=(1, T, 80)), -6.0, 1.5).astype(np.float32)
    f0 = rng.uniform(6.5, 9.2, size=(1, T)).astype(np.float32)
    uv = (rng.uniform(size=(1, T)) < 0.3).astype(np.float32)
    f0[uv > 0] = 0.0
    return {
        "edited_txt_tokens": torch.from_numpy(rng.integers(1, n_tokens, size=(1, Te), dtype=np.int64)),
        "mel": torch.from_numpy(mel),
        "mel2ph": torch.tensor([mel2ph], dtype=torch.int64),
        "mel2word": torch.tensor([mel2word], dtype=torch.int64),
        "dur": torch.tensor([dur], dtype=torch.int64),
        "ph2word": torch.tensor([ph2word], dtype=torch.int64),
        "edited_ph2word": torch.tensor([edited_ph2word], dtype=torch.int64),
        "f0": torch.from_numpy(f0), "uv": torch.from_numpy(uv),
        "words_region": [[w0, w1]], "edited_words_region": [[c0, c1]],
        "spk_embed": torch.from_numpy((rng.standard_normal(size=(1, 256)) / 16.0).astype(np.float32)),
        "text": ["synthetic"], "item_name": ["edit_%d" % seed],
    }


def synthetic_noises(B, T, steps, seed=4321, M=80):
    """x_T followed by one eps per executed step (i = steps-1 .. 0)."""
    rng = np.random.default_rng(seed)
    return [torch.from_numpy(rng.standard_normal(size=(B, 1, M, T), dtype=np.float32)) for _ in range(steps + 1)]


HIFIGAN_V1 = {
    "resblock": "1",
    "upsample_rates": [8, 8, 2, 2],
    "upsample_kernel_sizes": [16, 16, 4, 4],
    "upsample_initial_channel": 512,
    "resblock_kernel_sizes": [3, 7, 11],
    "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
}

# the other two generator shapes of the HiFi-GAN paper (the reference's real config.yaml is an external download,
# tasks/tts/vocoder_infer/hifigan.py:14-18: generic-h is the only defence, so it is pinned on all three published shapes)
HIFIGAN_V2 = {
    "resblock": "1",
    "upsample_rates": [8, 8, 2, 2],
    "upsample_kernel_sizes": [16, 16, 4, 4],
    "upsample_initial_channel": 128,
    "resblock_kernel_sizes": [3, 7, 11],
    "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
}

HIFIGAN_V3 = {
    "resblock": "2",
    "upsample_rates": [8, 8, 4],
    "upsample_kernel_sizes": [16, 16, 8],
    "upsample_initial_channel": 256,
    "resblock_kernel_sizes": [3, 5, 7],
    "resblock_dilation_sizes": [[1, 2], [2, 6], [3, 12]],
}

HIFIGAN_TINY = {
    "resblock": "1",
    "upsample_rates": [4, 2],
    "upsample_kernel_sizes": [8, 4],
    "upsample_initial_channel": 64,
    "resblock_kernel_sizes": [3, 5],
    "resblock_dilation_sizes": [[1, 3, 5], [1, 2, 3]],
}

HIFIGAN_TINY_RB2 = {
    "resblock": "2",
    "upsample_rates": [4, 4],
    "upsample_kernel_sizes": [8, 8],
    "upsample_initial_channel": 64,
    "resblock_kernel_sizes": [3, 7],
    "resblock_dilation_sizes": [[1, 3], [1, 3]],
}
