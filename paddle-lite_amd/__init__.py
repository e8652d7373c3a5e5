"""paddle-lite_amd — MI355X (gfx950) INT8 conv / depthwise / fc backend for Paddle-Lite's KernelLite plugin API.

Layout:
  csrc/        hand-written HIP kernels + the C ABI (include/plhip.h)  -> libplhip.so
  lite/        C++ host side mirroring the reference's plugin interface (lite/core, lite/operators,
               lite/backends/hip, lite/kernels/hip) + a mini program runner -> libpaddle_lite_hip.so
  capi.py      ctypes binding of libplhip.so (tests / bench only)
  liteapi.py   ctypes binding of the C++ kernel-class harness (tests / bench only)

The directory name contains a hyphen, so import it through `__graft_entry__.import_package()`
(it registers the package as `paddle_lite_amd`).
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
