"""MI355X-native render path for the per-pixel trace loop of WereDiggle/rust-raytracer.

Layout:
  csrc/     HIP kernels + the C ABI (include/rt_amd.h) -> librt_amd.so
  _abi.py / render.py / dist.py   thin ctypes + torch.distributed plumbing over the C ABI
(the C++ stand-in for the Rust caller and the transcribed test scenes live in ../harness)
"""
from . import _abi  # noqa: F401
