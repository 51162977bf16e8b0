"""Import the upstream reference (/root/reference) in THIS container only.

TEST INFRASTRUCTURE -- never imported by the product path.  Used by
oracle/make_golden.py to (1) validate the oracle restatement and (2) emit the
golden vectors committed under tests/golden/.  The reference never travels to
the GPU box; nothing under tests/ -m gpu, smoke() or bench.py may import this.

Recipe: stub the third-party modules that are only
pulled in by import chains (librosa, textgrid, ...), populate the reference's
global `hparams` dict from egs/spec_denoiser.yaml BEFORE importing model
modules (diffusion_utils.py:71,100 read hparams at import time), then import.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("SET_REFERENCE_ROOT", "/root/reference")

_STUBS = [
    "librosa", "librosa.feature", "librosa.core", "librosa.filters", "librosa.util",
    "pyloudnorm", "textgrid", "webrtcvad", "skimage", "skimage.transform",
    "parselmouth", "resemblyzer", "g2p_en", "g2p_en.expand", "fastdtw", "pycwt",
    "nltk", "nltk.tokenize", "torch.utils.tensorboard", "pesq", "pystoi",
    "python_speech_features",
]


class _Dummy:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return None

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Dummy()


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        # capitalised names may be subclassed by the reference (g2p_en.G2p)
        if name[:1].isupper():
            return _Dummy
        return _Dummy()


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "modules", "speech_editing"))


def install(timesteps=None, overrides=None):
    """Make `modules.*`, `utils.*`, `tasks.*` of the reference importable.

    Returns the reference's global hparams dict (already populated)."""
    if not available():
        raise RuntimeError("reference not present at %s" % REF_ROOT)
    import numpy as np
    import yaml
    if not hasattr(np, "Inf"):
        np.Inf = np.inf  # trainer.py:93 predates NumPy 2
    for name in _STUBS:
        if name in sys.modules:
            continue
        try:
            __import__(name)
        except Exception:
            m = _StubModule(name)
            m.__path__ = []
            sys.modules[name] = m
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    from utils.commons.hparams import hparams  # reference's global dict
    with open(os.path.join(REF_ROOT, "egs", "spec_denoiser.yaml")) as f:
        hparams.update(yaml.safe_load(f))
    if timesteps is not None:
        hparams["timesteps"] = int(timesteps)
    if overrides:
        hparams.update(overrides)
    return hparams


def import_spec_denoiser_infer():
    """The reference's inference caller, inference/tts/spec_denoiser.py.  It imports its own base class and helpers
    through a package name that is absent from the repository (`inference_acl.tts.*`, :13-14); alias that name to
    the modules that ARE in the repository (`inference.tts.base_tts_infer`, `inference.tts.infer_utils`)."""
    import importlib
    base = importlib.import_module("inference.tts.base_tts_infer")
    utils_ = importlib.import_module("inference.tts.infer_utils")
    pkg = types.ModuleType("inference_acl")
    pkg.__path__ = []
    sub = types.ModuleType("inference_acl.tts")
    sub.__path__ = []
    sys.modules.setdefault("inference_acl", pkg)
    sys.modules.setdefault("inference_acl.tts", sub)
    sys.modules.setdefault("inference_acl.tts.base_tts_infer", base)
    sys.modules.setdefault("inference_acl.tts.infer_utils", utils_)
    return importlib.import_module("inference.tts.spec_denoiser")
