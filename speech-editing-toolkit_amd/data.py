"""Data feed of the hot path: the reference's on-disk format and batch layout, host side.

* `IndexedDataset` / `IndexedDatasetBuilder`: `<prefix>.idx` = np.save({'offsets': [...]}), `<prefix>.data` =
  concatenated pickle.dumps(item)  (utils/commons/indexed_datasets.py:7-54); items carry
  `item_name, txt, ph_token, mel[T,80] (log10), mel2ph[T], f0[T] (Hz, 0 = unvoiced), pitch[T], spk_embed[256], wav_fn`.
* `StutterSpeechDataset`: per-item tensors + the editing mask, and `collater` padding to a batch with the keys
  `run_model` reads (tasks/speech_editing/dataset_utils.py:13-170, utils/commons/dataset_utils.py:13-62).
* `norm_interp_f0` (utils/audio/pitch/utils.py:41-68) and the three mask generators
  (utils/spec_aug/time_mask.py:6-93) are integer / small-array host logic and stay in numpy, like the reference.
"""
import os
import pickle
import random
import sys

import numpy as np
import torch


class IndexedDataset:
    def __init__(self, path):
        self.path = path
        self.data_offsets = np.load("%s.idx" % path, allow_pickle=True).item()["offsets"]
        self.data_file = open("%s.data" % path, "rb", buffering=-1)

    def __len__(self):
        return len(self.data_offsets) - 1

    def __getitem__(self, i):
        if i < 0 or i >= len(self):
            raise IndexError("index out of range")
        self.data_file.seek(self.data_offsets[i])
        return pickle.loads(self.data_file.read(self.data_offsets[i + 1] - self.data_offsets[i]))

    def __del__(self):
        f = getattr(self, "data_file", None)
        if f:
            f.close()


class IndexedDatasetBuilder:
    def __init__(self, path):
        self.path = path
        self.out_file = open("%s.data" % path, "wb")
        self.byte_offsets = [0]

    def add_item(self, item):
        self.byte_offsets.append(self.byte_offsets[-1] + self.out_file.write(pickle.dumps(item)))

    def finalize(self):
        self.out_file.close()
        with open("%s.idx" % self.path, "wb") as f:
            np.save(f, {"offsets": self.byte_offsets})


def norm_interp_f0(f0):
    """Hz -> log2 Hz with unvoiced frames linearly interpolated; returns (f0 fp32, uv fp32)."""
    f0 = np.asarray(f0, dtype=np.float64).copy()
    uv = f0 == 0
    f0 = np.log2(f0 + 1e-8)
    f0[uv] = 0
    if uv.sum() == len(f0):
        f0[uv] = 0
    elif uv.sum() > 0:
        f0[uv] = np.interp(np.where(uv)[0], np.where(~uv)[0], f0[~uv])
    return torch.FloatTensor(f0), torch.FloatTensor(uv.astype(np.float32))


def _ph_to_mel_mask(ph_mask, mel2ph):
    ph_mask = np.concatenate([[0.0], ph_mask]).astype(np.float32)  # F.pad(ph_mask, [1, 0]); mel2ph is 1-based
    return torch.from_numpy(ph_mask)[mel2ph]


def generate_time_mask(n_frames, ratio):
    """time_mask.py:6-48: one random contiguous span of int(D*ratio) frames (torch RNG, like the reference)."""
    length = int(n_frames * ratio)
    pos = int(torch.randint(0, max(1, n_frames - length), (1, 1)).item())
    ar = torch.arange(n_frames)
    return ((pos <= ar) & (ar < pos + length)).float()


def generate_alignment_aware_time_mask(mel2ph, ratio):
    """time_mask.py:50-70: mask a random subset of phonemes (numpy RNG), expand to frames through mel2ph."""
    n = int(mel2ph.max()) + 1
    ph_mask = np.zeros(n)
    idx = np.random.choice(np.arange(0, int(mel2ph.max()), dtype=float), size=int(n * ratio), replace=False).astype(np.uint8)
    ph_mask[idx] = 1.0
    return _ph_to_mel_mask(ph_mask, mel2ph)


def generate_inference_mask(mel2ph, ratio):
    """time_mask.py:72-93: one contiguous span of phonemes (python RNG)."""
    mx = int(mel2ph.max())
    ph_mask = np.zeros(mx + 1)
    start = random.randint(0, int(mx - mx * ratio))
    ph_mask[start:int(start + mx * ratio)] = 1.0
    return _ph_to_mel_mask(ph_mask, mel2ph)


def collate_1d_or_2d(values, pad_idx=0):
    size = max(v.shape[0] for v in values)
    shape = (len(values), size) + tuple(values[0].shape[1:])
    res = values[0].new_full(shape, pad_idx)
    for i, v in enumerate(values):
        res[i, :v.shape[0]] = v
    return res


def batch_by_size(indices, num_tokens_fn, max_tokens=None, max_sentences=None, required_batch_size_multiple=1):
    """Token-budget bucketing of an ordered index list (utils/commons/dataset_utils.py:65-119, fairseq's rule):
    a batch is closed when adding the next item would make (items + 1) x (longest item so far) exceed `max_tokens` or
    when it already holds `max_sentences` items; a closed batch is cut to a multiple of `required_batch_size_multiple`
    (the remainder opens the next batch).  Returns a list of index lists."""
    max_tokens = sys.maxsize if max_tokens is None else max_tokens
    max_sentences = sys.maxsize if max_sentences is None else max_sentences
    mult = required_batch_size_multiple
    batches, batch, lens, longest = [], [], [], 0
    for idx in indices:
        n = num_tokens_fn(idx)
        lens.append(n)
        longest = max(longest, n)
        if longest > max_tokens:
            raise AssertionError("sentence at index %d of size %d exceeds max_tokens limit of %d!" % (idx, longest, max_tokens))
        full = len(batch) > 0 and (len(batch) == max_sentences or (len(batch) + 1) * longest > max_tokens)
        if full:
            keep = max(mult * (len(batch) // mult), len(batch) % mult)
            batches.append(batch[:keep])
            batch, lens = batch[keep:], lens[keep:]
            longest = max(lens) if lens else 0
        batch.append(idx)
    if batch:
        batches.append(batch)
    return batches


def build_batches(dataset, shuffle, max_tokens=None, max_sentences=None, required_batch_size_multiple=-1, endless=False,
                  use_batch_by_size=True, world=1, rank=0, devices_cnt=None):
    """The batch list `SpeechBaseTask.build_dataloader` hands to its DataLoader (tasks/tts/speech_base.py:91-137):
    budgets are per device and scaled by the device count, batches are built over `dataset.ordered_indices()`,
    shuffled as whole batches (numpy global RNG, re-shuffled for each of the 1000 repetitions of an endless loader),
    and under data parallelism every rank keeps `batch[rank::world]` of the batches whose size divides evenly."""
    devices_cnt = max(1, world if devices_cnt is None else devices_cnt)
    if required_batch_size_multiple == -1:
        required_batch_size_multiple = devices_cnt
    if max_tokens is not None:
        max_tokens *= devices_cnt
    if max_sentences is not None:
        max_sentences *= devices_cnt
    indices = dataset.ordered_indices()
    if use_batch_by_size:
        sampler = batch_by_size(indices, dataset.num_tokens, max_tokens, max_sentences, required_batch_size_multiple)
    else:
        sampler = [list(indices[i:i + max_sentences]) for i in range(0, len(indices), max_sentences)]
    # One epoch's batches as python ints, ONCE; the (up to 1000) repetitions of an endless loader only repeat references
    # to these lists, like the reference does.  (Copying every repetition cost GBs of host memory and tens of seconds
    # per rank on a VCTK-sized set: ~2.7 k batches x 1000 repetitions.)
    sampler = [[int(i) for i in b] for b in sampler]
    if world > 1:  # speech_base.py:128-131: the rank's share of the batches whose size divides evenly
        kept = [b[rank::world] if len(b) % world == 0 else None for b in sampler]
    else:
        kept = sampler
    n = len(sampler)

    def epoch_order():
        # np.random.shuffle on a python list of n items draws the same numbers and applies the same swaps whatever the
        # items are, so shuffling the batch NUMBERS reproduces the reference's shuffle of the batch list itself
        order = list(range(n))
        if shuffle:
            np.random.shuffle(order)
        return order

    if shuffle:
        order = epoch_order()  # (the reference shuffles once, then again for each repetition of an endless loader)
        if endless:
            order = [i for _ in range(1000) for i in epoch_order()]
    else:
        order = [i for _ in range(1000) for i in range(n)] if endless else list(range(n))
    return [kept[i] for i in order if kept[i] is not None]


class StutterSpeechDataset:
    def __init__(self, prefix, hparams, items=None, data_dir=None, shuffle=False):
        self.hparams = hparams
        self.prefix = prefix
        self.shuffle = shuffle
        self.sort_by_len = bool(hparams.get("sort_by_len", True))
        data_dir = data_dir or hparams["binary_data_dir"]
        if items is not None:
            self.ds, self.avail = items, list(range(len(items)))
            self.sizes = [1] * len(items)
        else:
            self.ds = IndexedDataset("%s/%s" % (data_dir, prefix))
            ids = hparams.get("test_ids") or []
            self.avail = list(ids) if (prefix == "test" and len(ids) > 0) else list(range(len(self.ds)))
            # `<prefix>_lengths.npy` = mel frames per item (tasks/tts/dataset_utils.py:27-34); train drops short items
            lp = "%s/%s_lengths.npy" % (data_dir, prefix)
            sizes = np.load(lp) if os.path.exists(lp) else np.array([len(self.ds[i]["mel"]) for i in range(len(self.ds))])
            if prefix == "train" and hparams.get("min_frames", 0) > 0:
                self.avail = [i for i in self.avail if sizes[i] >= hparams["min_frames"]]
            self.sizes = [int(sizes[i]) for i in self.avail]

    def __len__(self):
        return len(self.avail)

    def size(self, index):
        return min(self.sizes[index], self.hparams["max_frames"])

    num_tokens = size

    def ordered_indices(self):
        """utils/commons/dataset_utils.py:202-211: a numpy-RNG permutation, stably sorted by length when `sort_by_len`."""
        if self.shuffle:
            indices = np.random.permutation(len(self))
            if self.sort_by_len:
                indices = indices[np.argsort(np.array(self.sizes)[indices], kind="mergesort")]
        else:
            indices = np.arange(len(self))
        return indices

    def __getitem__(self, index):
        hp = self.hparams
        item = self.ds[self.avail[index]]
        spec = torch.Tensor(np.asarray(item["mel"]))[:hp["max_frames"]]
        T = spec.shape[0] // hp["frames_multiple"] * hp["frames_multiple"]
        spec = spec[:T]
        sample = {"id": index, "item_name": item["item_name"], "text": item.get("txt"), "wav_fn": item.get("wav_fn"),
                  "txt_token": torch.LongTensor(np.asarray(item["ph_token"])[:hp["max_input_tokens"]]), "mel": spec}
        if hp["use_spk_embed"]:
            sample["spk_embed"] = torch.Tensor(np.asarray(item["spk_embed"]))
        sample["mel2ph"] = mel2ph = torch.LongTensor(np.asarray(item["mel2ph"]))[:T]
        if hp["use_pitch_embed"]:  # dataset_utils.py:109-127; without it the three keys are None
            f0, uv = norm_interp_f0(np.asarray(item["f0"])[:T])
            sample["f0"], sample["uv"] = f0, uv
            sample["pitch"] = torch.LongTensor(np.asarray(item.get(hp.get("pitch_key", "pitch"), np.zeros(T))))[:T]
        else:
            sample["f0"], sample["uv"], sample["pitch"] = None, None, None
        if not hp["infer"]:
            if hp.get("mask_type") == "random":
                m = generate_time_mask(T, hp["training_mask_ratio"])
            else:
                m = generate_alignment_aware_time_mask(mel2ph, hp["training_mask_ratio"])
        else:
            m = generate_inference_mask(mel2ph, 0.5)  # dataset_utils.py:143-145
        sample["time_mel_mask"] = m
        return sample

    def collater(self, samples):
        if not samples:
            return {}
        batch = {
            "id": torch.LongTensor([s["id"] for s in samples]),
            "item_name": [s["item_name"] for s in samples], "text": [s["text"] for s in samples],
            "wav_fn": [s["wav_fn"] for s in samples], "nsamples": len(samples),
            "txt_tokens": collate_1d_or_2d([s["txt_token"] for s in samples], 0),
            "mels": collate_1d_or_2d([s["mel"] for s in samples], 0.0),
            "txt_lengths": torch.LongTensor([s["txt_token"].numel() for s in samples]),
            "mel_lengths": torch.LongTensor([s["mel"].shape[0] for s in samples]),
            "mel2ph": collate_1d_or_2d([s["mel2ph"] for s in samples], 0),
            "f0": None, "uv": None, "pitch": None,
            "time_mel_masks": collate_1d_or_2d([s["time_mel_mask"] for s in samples], 0.0),
        }
        if self.hparams["use_pitch_embed"]:  # dataset_utils.py:154-159
            batch["f0"] = collate_1d_or_2d([s["f0"] for s in samples], 0.0)
            batch["uv"] = collate_1d_or_2d([s["uv"] for s in samples], 0.0)
            batch["pitch"] = collate_1d_or_2d([s["pitch"] for s in samples], 0)
        if self.hparams["use_spk_embed"]:
            batch["spk_embed"] = torch.stack([s["spk_embed"] for s in samples])
        return batch
