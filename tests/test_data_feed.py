"""Host-side data feed (speech-editing-toolkit_amd/data.py) against fixtures produced by the reference's own
functions / IndexedDatasetBuilder (oracle/make_golden.py::data_feed_case).  CPU only."""
import os
import random

import numpy as np
import pytest
import torch

from conftest import GOLDEN, base_hparams, load_golden
import set_amd  # noqa: F401,E402
from set_amd import data as D  # noqa: E402


def test_norm_interp_f0_and_masks_match_reference():
    import set_amd  # noqa: F401
    from set_amd import data as D
    g = load_golden("data_feed")
    f0, uv = D.norm_interp_f0(g["f0"])
    assert np.array_equal(f0.numpy(), g["f0_norm"]) and np.array_equal(uv.numpy(), g["uv"])
    mel2ph = torch.from_numpy(g["mel2ph"])
    random.seed(7)
    assert np.array_equal(D.generate_inference_mask(mel2ph, 0.5).numpy(), g["infer_mask_seed7"])
    np.random.seed(11)
    assert np.array_equal(D.generate_alignment_aware_time_mask(mel2ph, 0.8).numpy(), g["align_mask_seed11"])
    torch.manual_seed(13)
    assert np.array_equal(D.generate_time_mask(200, 0.3).numpy(), g["time_mask_seed13"])
    z = np.zeros(10)
    f0z, uvz = D.norm_interp_f0(z)  # fully unvoiced utterance
    assert float(f0z.abs().sum()) == 0.0 and float(uvz.sum()) == 10.0


def test_indexed_dataset_reads_reference_files_and_round_trips(tmp_path):
    import set_amd  # noqa: F401
    from set_amd import data as D
    ds = D.IndexedDataset(os.path.join(GOLDEN, "binary_tiny", "test"))  # written by the reference's builder
    assert len(ds) == 3 and [ds[i]["item_name"] for i in range(3)] == ["utt0", "utt1", "utt2"]
    assert ds[1]["mel"].shape == (56, 80) and ds[2]["mel2ph"].dtype == np.int64
    try:
        ds[3]
        assert False
    except IndexError:
        pass
    b = D.IndexedDatasetBuilder(str(tmp_path / "x"))
    for i in range(3):
        b.add_item(ds[i])
    b.finalize()
    assert open(tmp_path / "x.data", "rb").read() == open(os.path.join(GOLDEN, "binary_tiny", "test.data"), "rb").read()
    rt = D.IndexedDataset(str(tmp_path / "x"))
    assert all(np.array_equal(rt[i]["mel"], ds[i]["mel"]) for i in range(3))


def test_dataset_items_and_collater_layout():
    import set_amd  # noqa: F401
    from set_amd import data as D
    hp = base_hparams(binary_data_dir=os.path.join(GOLDEN, "binary_tiny"), infer=True, test_ids=[])
    ds = D.StutterSpeechDataset("test", hp)
    random.seed(1)
    samples = [ds[i] for i in range(3)]
    batch = ds.collater(samples)
    assert batch["mels"].shape == (3, 56, 80) and batch["txt_tokens"].shape == (3, 10)
    assert batch["mel2ph"].dtype == torch.int64 and batch["time_mel_masks"].shape == (3, 56)
    assert batch["mel_lengths"].tolist() == [40, 56, 33] and batch["nsamples"] == 3
    assert float(batch["mels"][0, 40:].abs().sum()) == 0.0 and int(batch["mel2ph"][2, 33:].sum()) == 0
    for s in samples:  # inference mask = one contiguous phoneme span, expanded through mel2ph
        m = s["time_mel_mask"]
        assert set(m.tolist()) <= {0.0, 1.0} and m.sum() > 0
        ph = torch.unique(s["mel2ph"][m == 1])
        assert int(ph.max() - ph.min()) + 1 >= len(ph)
    hp2 = dict(hp, infer=False, mask_type="alignment_aware")
    np.random.seed(2)
    s = D.StutterSpeechDataset("test", hp2)[1]
    assert s["time_mel_mask"].shape == (56,) and s["f0"].shape == (56,) and s["uv"].max() <= 1
    # use_pitch_embed false (egs/spec_denoiser_libritts.yaml): the pitch keys are None (dataset_utils.py:125-127,158-159)
    ds3 = D.StutterSpeechDataset("test", dict(hp, use_pitch_embed=False))
    random.seed(1)
    b3 = ds3.collater([ds3[i] for i in range(3)])
    assert b3["f0"] is None and b3["uv"] is None and b3["pitch"] is None
    assert torch.equal(b3["mels"], batch["mels"]) and torch.equal(b3["mel2ph"], batch["mel2ph"])


def test_batch_by_size_matches_reference_fixture():
    """utils/commons/dataset_utils.py:65-119: fixture produced by the reference's own batch_by_size
    (oracle/make_golden.py::batch_by_size_case) on six (sizes, order, budget, multiple) cases."""
    import json
    cases = json.load(open(os.path.join(GOLDEN, "batch_by_size.json")))
    assert len(cases) == 6
    for c in cases:
        got = D.batch_by_size(c["order"], lambda i: c["sizes"][i], c["max_tokens"], c["max_sentences"], c["mult"])
        assert got == c["batches"]
        flat = [i for b in got for i in b]
        assert flat == c["order"]                                   # a partition of the ordered indices, in order
        if c["max_sentences"]:
            assert max(len(b) for b in got) <= c["max_sentences"]
        if c["max_tokens"]:
            assert all(len(b) * max(c["sizes"][i] for i in b) <= c["max_tokens"] for b in got)
    with pytest.raises(AssertionError):
        D.batch_by_size([0], lambda i: 500, max_tokens=400)


def test_build_batches_striding_and_loader_is_position_seeded():
    """tasks/tts/speech_base.py:91-137: budgets scale with the device count, each rank keeps batch[rank::world] of the
    evenly divisible batches; trainer.BatchLoader re-seeds the mask generators per batch position, so batch k is the same
    whether or not batches 0..k-1 were fetched (what makes a resumed run continue exactly)."""
    import numpy as np
    from set_amd.trainer import BatchLoader
    hp = base_hparams(binary_data_dir=os.path.join(GOLDEN, "binary_tiny"), infer=False, test_ids=[], max_frames=1548,
                      max_input_tokens=1550, frames_multiple=1, sort_by_len=True, min_frames=0)
    ds = D.StutterSpeechDataset("test", hp, shuffle=True)
    assert [ds.num_tokens(i) for i in range(len(ds))] == [40, 56, 33]
    np.random.seed(3)
    order = ds.ordered_indices().tolist()
    assert order == [2, 0, 1]                                       # sorted by length (stable) after the permutation
    np.random.seed(3)
    one = D.build_batches(ds, True, max_tokens=1000, max_sentences=2, endless=False, world=1, rank=0)
    assert sorted(sorted(b) for b in one) == [[0, 2], [1]]
    np.random.seed(3)
    r0 = D.build_batches(ds, True, max_tokens=1000, max_sentences=1, endless=False, world=2, rank=0)
    np.random.seed(3)
    r1 = D.build_batches(ds, True, max_tokens=1000, max_sentences=1, endless=False, world=2, rank=1)
    # max_sentences 1 x 2 devices = global batches of 2, multiple of 2: [2,0] is split, the odd remainder [1] is dropped
    assert len(r0) == len(r1) == 1 and sorted(r0[0] + r1[0]) == [0, 2]
    np.random.seed(3)
    endless = D.build_batches(ds, True, max_tokens=1000, max_sentences=2, endless=True)
    assert len(endless) == 2000
    ld = BatchLoader(ds, one, seed=1234)
    a = ld.fetch(1)
    ld.fetch(0)
    b = ld.fetch(1)
    assert torch.equal(a["time_mel_masks"], b["time_mel_masks"]) and torch.equal(a["mels"], b["mels"])
    assert set(a) >= {"txt_tokens", "mels", "mel2ph", "f0", "uv", "time_mel_masks", "spk_embed", "nsamples"}


def test_endless_batch_list_follows_the_reference_shuffle_and_shares_its_batches():
    """The endless training list (tasks/tts/speech_base.py:113-121: the epoch's batch list shuffled once, then once more
    for each of 1000 repetitions, numpy global generator) restated literally here on a synthetic index set; the
    product builds it from shuffled batch NUMBERS and must give the same sequence, on every rank -- and the repetitions
    must share one epoch's lists instead of copying them (GBs of host memory on a VCTK-sized set)."""
    import numpy as np

    class Stub:
        sizes = [17, 60, 33, 41, 25, 58, 12, 49, 30, 44, 21]

        def ordered_indices(self):
            return np.argsort(np.array(self.sizes), kind="mergesort")

        def num_tokens(self, i):
            return self.sizes[i]

    ds = Stub()
    for world in (1, 2):
        sampler = D.batch_by_size(ds.ordered_indices(), ds.num_tokens, 130 * world, 3 * world, world)
        np.random.seed(11)

        def shuffled(bs):
            np.random.shuffle(bs)
            return bs
        want = shuffled(list(sampler))
        want = [b for _ in range(1000) for b in shuffled(list(sampler))]
        for rank in range(world):
            np.random.seed(11)
            got = D.build_batches(ds, True, 130, 3, endless=True, world=world, rank=rank)
            exp = [[int(i) for i in b[rank::world]] for b in want if len(b) % world == 0]
            assert got == exp
            assert all(type(i) is int for b in got[:5] for i in b)
            assert len({id(b) for b in got}) <= len(sampler)  # one epoch's lists, referenced 1000 times
    np.random.seed(5)
    fixed = D.build_batches(ds, False, 130, 3, endless=True)
    assert fixed[:len(fixed) // 1000] * 1000 == fixed


def test_fetch_is_a_pure_function_of_k_under_a_concurrent_prefetch_worker():
    """the prefetch worker (ds_workers > 0) assembles batch k + 1 while the main thread may run a validation
    pass or rebuild a loader -- all of them reseed / draw from the process-global numpy, python and torch generators.
    BatchLoader.fetch now holds trainer.RNG_LOCK from its reseed to its last draw and restores the three generator states, and
    the loader construction takes the same lock: every batch is the one a serial fetch produces, whatever the interleaving, and
    fetching leaves no trace in the global streams."""
    import random
    import threading
    import time
    import numpy as np
    from set_amd.trainer import BatchLoader, RNG_LOCK

    class SlowDataset:
        """draws from all three global generators with a sleep in between (widens the race window)"""

        def __getitem__(self, i):
            a = float(np.random.rand())
            time.sleep(0.002)
            b = random.random()
            time.sleep(0.002)
            c = float(torch.rand(()))
            return {"v": torch.tensor([i, a, b, c], dtype=torch.float64)}

        def collater(self, samples):
            return {"v": torch.stack([s["v"] for s in samples])}

    train = BatchLoader(SlowDataset(), [[0, 1], [2, 3], [4, 5], [6, 7]], seed=7)
    val = BatchLoader(SlowDataset(), [[10, 11], [12, 13]], seed=7)
    serial_t = [train.fetch(k)["v"] for k in range(8)]
    serial_v = [val.fetch(k)["v"] for k in range(4)]
    np.random.seed(123); random.seed(123); torch.manual_seed(123)
    before = (np.random.get_state()[1].copy(), random.getstate(), torch.get_rng_state().clone())
    got_t, got_v, stop = {}, {}, threading.Event()

    def worker():
        for k in range(8):
            got_t[k] = train.fetch(k)["v"]

    def reseeder():  # what tasks._loader does while it builds a batch list
        while not stop.is_set():
            with RNG_LOCK:
                st = np.random.get_state()
                np.random.seed(99)
                np.random.permutation(50)
                np.random.set_state(st)
            time.sleep(0.001)

    th, rs = threading.Thread(target=worker), threading.Thread(target=reseeder)
    th.start(); rs.start()
    for k in range(4):
        got_v[k] = val.fetch(k)["v"]
    th.join(); stop.set(); rs.join()
    assert all(torch.equal(got_t[k], serial_t[k]) for k in range(8))
    assert all(torch.equal(got_v[k], serial_v[k]) for k in range(4))
    after = (np.random.get_state()[1], random.getstate(), torch.get_rng_state())
    assert np.array_equal(before[0], after[0]) and before[1] == after[1] and torch.equal(before[2], after[2])
