"""ctypes binding of cpuraytracer_amd/lib/librt_hip.so (the C ABI of include/rt_api.h).

The library is hand-written HIP for gfx950; there is no CPU fallback.  Loading succeeds without a
GPU (so the symbol table can be checked on a CPU-only host) but rt_create fails loudly there.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "librt_hip.so")


class RtSphere(C.Structure):
    _fields_ = [("cx", C.c_float), ("cy", C.c_float), ("cz", C.c_float), ("r", C.c_float)]


class RtMaterial(C.Structure):
    _fields_ = [("type", C.c_uint32), ("tex_type", C.c_uint32), ("smoothness", C.c_float), ("ior", C.c_float),
                ("tiling", C.c_float), ("rgb0", C.c_float * 3), ("rgb1", C.c_float * 3), ("luminance", C.c_float)]


class RtCamera(C.Structure):
    _fields_ = [("origin", C.c_float * 4), ("x", C.c_float * 4), ("y", C.c_float * 4),
                ("origin_image_plane", C.c_float * 4), ("aperture", C.c_float), ("focal_length", C.c_float)]


class RtLight(C.Structure):
    _fields_ = [("direction", C.c_float * 3), ("color", C.c_float * 3), ("luminance", C.c_float)]


class RtRowset(C.Structure):
    _fields_ = [("first_row", C.c_uint32), ("num_rows", C.c_uint32), ("block_rows", C.c_uint32),
                ("shard", C.c_uint32), ("nshards", C.c_uint32)]


class RtStats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("traversals", C.c_uint64), ("segments", C.c_uint64),
                ("ms_render", C.c_double), ("ms_accumulate", C.c_double), ("ms_resolve", C.c_double),
                ("local_rows", C.c_uint32), ("passes", C.c_uint32)]


SPHERE_DTYPE = np.dtype([("cx", "<f4"), ("cy", "<f4"), ("cz", "<f4"), ("r", "<f4")])
MATERIAL_DTYPE = np.dtype([("type", "<u4"), ("tex_type", "<u4"), ("smoothness", "<f4"), ("ior", "<f4"),
                           ("tiling", "<f4"), ("rgb0", "<f4", (3,)), ("rgb1", "<f4", (3,)), ("luminance", "<f4")])

RT_OK = 0
RT_ERR_NO_DEVICE = 1
RT_SAMPLER_COSINE_HEMISPHERE, RT_SAMPLER_SQRT_DISK = 1, 2

# every symbol include/rt_api.h declares
EXPORTS = [
    "rt_last_error", "rt_api_version", "rt_create", "rt_destroy", "rt_set_stream", "rt_set_workspace_limit", "rt_set_sampler", "rt_set_frame_pipelining", "rt_set_frame_batch", "rt_set_frame_lookahead", "rt_committed_samples",
    "rt_scene_upload", "rt_render", "rt_clear", "rt_resolve", "rt_last_resolve_ms", "rt_download", "rt_copy_to_device",
    "rt_synchronize", "rt_rowset_local_rows", "rt_rowset_global_row", "rt_unit_halton", "rt_unit_math",
    "rt_unit_primary_rays", "rt_unit_closest_hit", "rt_unit_trace", "rt_unit_camera_rays", "rt_unit_scatter", "rt_unit_tonemap", "rt_unit_layout", "rt_unit_layout_info", "rt_unit_grid_rows",
]

_lib = None


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("librt_hip: [%d] %s" % (code, msg))
        self.code = code


def load():
    """dlopen librt_hip.so; raises if the library has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("librt_hip.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'`; "
                           "there is no CPU fallback for the render path" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.rt_last_error.restype = C.c_char_p
    L.rt_api_version.restype = C.c_int
    L.rt_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.rt_destroy.argtypes = [C.c_void_p]
    L.rt_destroy.restype = None
    L.rt_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.rt_set_workspace_limit.argtypes = [C.c_void_p, C.c_uint64]
    L.rt_set_sampler.argtypes = [C.c_void_p, C.c_uint32]
    L.rt_set_frame_pipelining.argtypes = [C.c_void_p, C.c_uint32]
    L.rt_set_frame_batch.argtypes = [C.c_void_p, C.c_uint32]
    L.rt_committed_samples.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    L.rt_set_frame_lookahead.argtypes = [C.c_void_p, C.c_uint32]
    L.rt_scene_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(RtCamera), C.POINTER(RtLight), C.c_uint32,
                                  C.POINTER(RtMaterial), C.c_float]
    L.rt_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, RtRowset, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                            C.POINTER(RtStats)]
    L.rt_clear.argtypes = [C.c_void_p]
    L.rt_resolve.argtypes = [C.c_void_p, C.c_uint32]
    L.rt_last_resolve_ms.argtypes = [C.c_void_p]
    L.rt_last_resolve_ms.restype = C.c_double
    L.rt_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.rt_copy_to_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.rt_synchronize.argtypes = [C.c_void_p]
    L.rt_rowset_local_rows.argtypes = [RtRowset]
    L.rt_rowset_local_rows.restype = C.c_uint32
    L.rt_rowset_global_row.argtypes = [RtRowset, C.c_uint32]
    L.rt_rowset_global_row.restype = C.c_uint32
    L.rt_unit_halton.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    L.rt_unit_math.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.rt_unit_primary_rays.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
    L.rt_unit_closest_hit.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.rt_unit_trace.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64,
                                C.c_void_p, C.c_void_p]
    L.rt_unit_camera_rays.argtypes = [C.c_void_p, C.POINTER(RtCamera), C.c_void_p, C.c_uint32, C.c_void_p]
    L.rt_unit_scatter.argtypes = [C.c_void_p, C.POINTER(RtMaterial), C.POINTER(RtLight), C.POINTER(C.c_float), C.c_void_p, C.c_uint32,
                                  C.c_void_p]
    L.rt_unit_tonemap.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    L.rt_unit_layout.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p]
    L.rt_unit_layout_info.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    _lib = L
    return L


def check(rc):
    if rc != RT_OK:
        raise RtError(rc, load().rt_last_error().decode())


def whole_image(H):
    return RtRowset(0, H, H, 0, 1)


BLOCK_ROWS = 1  # rows per block of the cyclic partition (distributed.py says why single rows)


def cyclic_rows(H, rank, world, block_rows=BLOCK_ROWS):
    """Row blocks b = rank (mod world) of block_rows rows."""
    return RtRowset(0, H, block_rows, rank, world)
