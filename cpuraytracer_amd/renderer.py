"""Python-side handle over the C ABI: scene upload, render, resolve, download.

Plumbing for tests, bench.py and the torch.distributed launcher — the host mirror of the
reference's class API is C++ (cpuraytracer_amd/csrc/host/).
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import RtRowset, RtStats, check, whole_image


class HipRenderer:
    """One context per device ordinal (rt_create).  Raises when no GPU is present."""

    def __init__(self, device=0):
        self._L = _capi.load()
        self._h = C.c_void_p()
        check(self._L.rt_create(int(device), C.byref(self._h)))
        self.W = self.rows = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.rt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream_ptr):
        check(self._L.rt_set_stream(self._h, C.c_void_p(hip_stream_ptr)))

    def set_workspace_limit(self, nbytes):
        check(self._L.rt_set_workspace_limit(self._h, int(nbytes)))

    def set_sampler(self, flags):
        """RT_SAMPLER_* flags (0 = the reference's uniform hemisphere and linear-r disk)."""
        check(self._L.rt_set_sampler(self._h, int(flags)))

    def set_frame_pipelining(self, depth):
        """rt_set_frame_pipelining: up to `depth` stats-less render calls may stay in flight (0 = off)."""
        check(self._L.rt_set_frame_pipelining(self._h, int(depth)))

    def set_frame_batch(self, frames):
        """rt_set_frame_batch: stats-less render calls that continue each other are rendered `frames` sample planes per launch (1 = off)."""
        check(self._L.rt_set_frame_batch(self._h, int(frames)))

    def set_frame_lookahead(self, frames):
        """rt_set_frame_lookahead: a stats-less render call traces the next `frames` sample planes with its one launch and adds only its own (1 = off)."""
        check(self._L.rt_set_frame_lookahead(self._h, int(frames)))

    def committed_samples(self):
        n = C.c_uint32(0)
        check(self._L.rt_committed_samples(self._h, C.byref(n)))
        return n.value

    def upload(self, scene):
        """scene: object with .spheres/.materials (numpy structured arrays in the rt_api.h layouts), .camera, .sun,
        .sky (ctypes structs of identical layout), .exposure_scale."""
        sph = np.ascontiguousarray(scene.spheres)
        mat = np.ascontiguousarray(scene.materials)
        assert sph.dtype.itemsize == 16 and mat.dtype.itemsize == 48
        cam = _capi.RtCamera.from_buffer_copy(bytes(scene.camera))
        # m_lights (spheres-app.h:38): scene.lights when the scene carries a list (it may be empty), else the generators' single sun
        ls = getattr(scene, "lights", None)
        ls = [scene.sun] if ls is None else list(ls)
        lights = (_capi.RtLight * max(1, len(ls)))()
        for k, l in enumerate(ls):
            lights[k] = _capi.RtLight.from_buffer_copy(bytes(l))
        sky = _capi.RtMaterial.from_buffer_copy(bytes(scene.sky))
        check(self._L.rt_scene_upload(self._h, sph.ctypes.data, mat.ctypes.data, sph.shape[0], C.byref(cam), lights, len(ls),
                                      C.byref(sky), float(scene.exposure_scale)))

    def render(self, W, H, s0, s1, max_depth, seed, rowset=None, stats=True):
        """rt_render; stats=False passes out_stats = NULL: the call only enqueues work on the context's stream (no host
        wait; progressive 1-spp frames are launch bound) and returns None."""
        rs = rowset if rowset is not None else whole_image(H)
        rs = RtRowset.from_buffer_copy(bytes(rs))
        if not stats:
            check(self._L.rt_render(self._h, W, H, rs, s0, s1, max_depth, seed, None))
            self.W, self.rows = W, self._L.rt_rowset_local_rows(rs)
            return None
        st = RtStats()
        check(self._L.rt_render(self._h, W, H, rs, s0, s1, max_depth, seed, C.byref(st)))
        self.W, self.rows = W, st.local_rows
        return st

    def clear(self):
        check(self._L.rt_clear(self._h))

    def resolve(self, n=0):
        check(self._L.rt_resolve(self._h, n))
        return self._L.rt_last_resolve_ms(self._h)

    def download(self, hdr=True, ldr=True):
        h = np.zeros((self.rows, self.W, 3), dtype=np.float32) if hdr else None
        l = np.zeros((self.rows, self.W, 3), dtype=np.uint8) if ldr else None
        check(self._L.rt_download(self._h, h.ctypes.data if hdr else None, l.ctypes.data if ldr else None))
        return h, l

    def copy_to_device(self, dev_hdr_ptr=None, dev_ldr_ptr=None):
        check(self._L.rt_copy_to_device(self._h, C.c_void_p(dev_hdr_ptr or 0), C.c_void_p(dev_ldr_ptr or 0)))

    def synchronize(self):
        check(self._L.rt_synchronize(self._h))

    # ---- unit entries
    def unit_halton(self, index, base):
        index = np.ascontiguousarray(index, dtype=np.uint32)
        out = np.zeros(index.shape[0], dtype=np.float32)
        check(self._L.rt_unit_halton(self._h, index.ctypes.data, base, index.shape[0], out.ctypes.data))
        return out

    def unit_math(self, op, x, y=None):
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.ascontiguousarray(y if y is not None else np.zeros_like(x), dtype=np.float32)
        out = np.zeros_like(x)
        check(self._L.rt_unit_math(self._h, op, x.ctypes.data, y.ctypes.data, x.shape[0], out.ctypes.data))
        return out

    def unit_primary_rays(self, W, H, ijs):
        ijs = np.ascontiguousarray(ijs, dtype=np.uint32).reshape(-1, 3)
        out = np.zeros((ijs.shape[0], 6), dtype=np.float32)
        check(self._L.rt_unit_primary_rays(self._h, W, H, ijs.ctypes.data, ijs.shape[0], out.ctypes.data))
        return out

    def unit_closest_hit(self, rays):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        out = np.zeros((rays.shape[0], 10), dtype=np.float32)
        check(self._L.rt_unit_closest_hit(self._h, rays.ctypes.data, rays.shape[0], out.ctypes.data))
        return out

    def unit_trace(self, W, H, ijs, max_depth, seed):
        ijs = np.ascontiguousarray(ijs, dtype=np.uint32).reshape(-1, 3)
        rgb = np.zeros((ijs.shape[0], 3), dtype=np.float32)
        trav = np.zeros(ijs.shape[0], dtype=np.uint32)
        check(self._L.rt_unit_trace(self._h, W, H, ijs.ctypes.data, ijs.shape[0], max_depth, seed, rgb.ctypes.data,
                                    trav.ctypes.data))
        return rgb, trav
