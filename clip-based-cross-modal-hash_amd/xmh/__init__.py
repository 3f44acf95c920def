"""xmh -- MI355X-native encode-and-retrieve path behind the plugin surface of
kalenforn/clip-based-cross-modal-hash.

Host side only: Python/PyTorch-ROCm for device memory, streams and
torch.distributed; every bit of arithmetic on the hot path runs in hand-written
HIP (gfx950) inside ``libxmh.so`` reached through the C ABI in ``include/xmh.h``.
There is no CPU fallback: importing :mod:`xmh._lib` without the built library,
or calling an op without a GPU, raises.
"""
__all__ = ["__version__"]
__version__ = "0.1.0"
