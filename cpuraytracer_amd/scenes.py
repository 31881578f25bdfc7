"""Scene tables from the product's own host mirror (csrc/host/spheres-app.cpp: InitScene/InitCamera).

Returns the rt_api.h records as numpy structured arrays + ctypes structs, ready for HipRenderer.upload.
"""
import ctypes as C
import os

import numpy as np

from . import _capi

_HOST_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "librt_host.so")
_host = None


def _load_host():
    global _host
    if _host is None:
        if not os.path.exists(_HOST_LIB):
            raise RuntimeError("librt_host.so is missing (%s): run __graft_entry__.build()" % _HOST_LIB)
        H = C.CDLL(_HOST_LIB)
        H.rth_build_scene.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_int, C.c_float, C.c_float, C.c_uint32, C.c_void_p,
                                      C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(_capi.RtCamera), C.POINTER(_capi.RtLight),
                                      C.POINTER(_capi.RtMaterial), C.POINTER(C.c_float)]
        _host = H
    return _host


class Scene:
    def __init__(self, spheres, materials, camera, sun, sky, exposure_scale, name, seed):
        self.spheres, self.materials = spheres, materials
        self.camera, self.sun, self.sky = camera, sun, sky
        self.exposure_scale = float(exposure_scale)
        self.name, self.seed = name, seed

    @property
    def n(self):
        return int(self.spheres.shape[0])


def build_scene(name="cover", seed=1, width=1200, height=800, vfov=-1.0, aperture=-1.0):
    """name: 'cover' (InitScene, 488 spheres), 'three' (C1), 'grid10k' (C5).  vfov/aperture < 0: scene default."""
    H = _load_host()
    cap = 10100
    sph = np.zeros(cap, dtype=_capi.SPHERE_DTYPE)
    mat = np.zeros(cap, dtype=_capi.MATERIAL_DTYPE)
    n = C.c_uint32(0)
    cam, sun, sky, exp = _capi.RtCamera(), _capi.RtLight(), _capi.RtMaterial(), C.c_float(0)
    rc = H.rth_build_scene(name.encode(), seed, width, height, vfov, aperture, cap, sph.ctypes.data, mat.ctypes.data, C.byref(n),
                           C.byref(cam), C.byref(sun), C.byref(sky), C.byref(exp))
    if rc != 0:
        raise RuntimeError("rth_build_scene(%r) failed with %d" % (name, rc))
    return Scene(sph[:n.value].copy(), mat[:n.value].copy(), cam, sun, sky, exp.value, name, seed)
