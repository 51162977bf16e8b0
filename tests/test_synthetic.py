import torch


def test_product_synthetic_equals_oracle_synthetic():
    import set_amd  # noqa: F401
    from set_amd.synthetic import synthetic_inputs
    from oracle import weights as Wt
    for pad in (False, True):
        a = synthetic_inputs(3, 40, 10, seed=5, pad_tail=pad)
        b = Wt.synthetic_inputs(3, 40, 10, seed=5, pad_tail=pad)
        assert a.keys() == b.keys()
        for k in a:
            assert torch.equal(a[k], b[k]), k
    s = synthetic_inputs(2, 800, 100)
    assert s["mel2ph"].min() >= 1 and (s["mel2ph"][:, 1:] >= s["mel2ph"][:, :-1]).all()
    assert s["ref_mels"].min() >= -6 and s["ref_mels"].max() <= 1.5
