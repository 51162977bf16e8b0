"""MI355X-native (gfx950) FluentSpeech spec_denoiser hot path + HiFi-GAN generator.

Host side mirrors the reference's operator surface (registries, module names,
state_dict keys, forward signatures); all arithmetic runs in hand-written HIP
kernels reached through the C ABI of libset_amd.so (include/set_amd.h).
There is NO CPU / eager fallback: every op raises if the library is missing.
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401
