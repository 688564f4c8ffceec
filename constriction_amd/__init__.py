"""constriction_amd -- MI355X-native batched entropy-coding backend for constriction's stream-coder hot path.

Layout:
  csrc/        hand-written HIP (gfx950) kernels + the C ABI of include/constriction_amd.h
  _native.py   ctypes binding of that ABI (no CPU fallback)
  batched.py   device-resident batched API (thousands of independent coders per call)
  stream/      drop-in mirror of `constriction.stream.{stack,queue,model}` for single coders
"""
__version__ = "0.1.0"

from . import _native  # noqa: F401
