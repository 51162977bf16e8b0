import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {k: d[k] for k in d.files}
    if "meta" in out:
        out["meta"] = json.loads(str(out["meta"]))
    return out


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def built_lib():
    """Build (if stale) and load libset_amd.so -- works without a GPU."""
    import set_amd  # noqa: F401
    from set_amd import _lib
    _lib.build()
    return _lib.lib()


def base_hparams(**over):
    """The spec_denoiser.yaml values the hot path reads (egs/spec_denoiser.yaml in the reference)."""
    import yaml
    with open(os.path.join(ROOT, "speech-editing-toolkit_amd", "egs", "spec_denoiser.yaml")) as f:
        hp = yaml.safe_load(f)
    hp.update(over)
    return hp
