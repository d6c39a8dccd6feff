"""gnnepcsaft_amd — MI355X-native (gfx950) forward/backward of gnnepcsaft's PNA / GINE message-passing models.

Host side mirrors the reference's ``gnnepcsaft.train.models`` API; all arithmetic runs in hand-written HIP kernels
behind the C ABI of ``include/gnx.h`` (``libgnnepcsaft_hip.so``).  Importing the package needs no GPU; any compute call
without the library or without a HIP device raises (no CPU fallback).
"""
__version__ = "0.1.0"

from . import data  # noqa: F401  (pure host code)
