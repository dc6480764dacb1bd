"""MI355X-native render path for the per-pixel trace loop of WereDiggle/rust-raytracer.

Layout:
  csrc/     HIP kernels + the C ABI (include/rt_amd.h) → librt_amd.so
  host/     C++ mirror of the reference's scene API + its test scenes → libraytracer_host.so
  scenes.py / render.py / dist.py   thin ctypes + torch.distributed plumbing for tests and bench
"""
from . import _abi  # noqa: F401
