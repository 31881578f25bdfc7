"""cpuraytracer_amd — MI355X-native render-loop hot path of SakibSaikia/CPURayTracer.

Holds only what the path needs: csrc/ (HIP kernels + C ABI + C++ host mirror) and a thin ctypes
layer used by tests, bench.py and the torch.distributed launcher.
"""
from ._capi import LIB_PATH, RtError, cyclic_rows, whole_image  # noqa: F401
from .renderer import HipRenderer  # noqa: F401
