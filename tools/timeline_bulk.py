"""Diagnostic (librt_hip_tl.so, -DRT_TIMELINE): wave lifetimes inside one C2 trace kernel (spp 128)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cpuraytracer_amd import _capi
_capi.LIB_PATH = os.path.join(ROOT, "cpuraytracer_amd", "lib", "exp", "librt_hip_tl.so")
from cpuraytracer_amd import HipRenderer, scenes
W, H, spp = 1200, 800, int(sys.argv[1]) if len(sys.argv) > 1 else 128
DEPTH = int(sys.argv[2]) if len(sys.argv) > 2 else 50
r = HipRenderer(0); r.upload(scenes.build_scene("cover", 1, W, H))
L = _capi.load(); L.rt_debug_timeline.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
out = (C.c_ulonglong * 16)()
L.rt_debug_timeline_hist.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
hist = (C.c_uint * 1024)()
r.render(W, H, 1, 1 + spp, DEPTH, 1)
L.rt_debug_timeline(r._h, out)
L.rt_debug_timeline_hist(r._h, hist)
for rep in range(3):
    st = r.render(W, H, 1, 1 + spp, DEPTH, 1)
    L.rt_debug_timeline(r._h, out)
    L.rt_debug_timeline_hist(r._h, hist)
    hh = [(k * 25, c) for k, c in enumerate(hist) if c]
    if rep == 2: print("   waves leaving per 25-us bucket of lifetime:", " ".join("%d:%d" % kc for kc in hh))
    v = list(out); t0 = v[0]
    print("kernel %.3f ms: first exit %+.1f us, last exit %+.1f us, mean wave life %.1f us (%.1f %% of the kernel), waves %d"
          % (st.ms_render, (v[7] - t0) / 100.0, (v[2] - t0) / 100.0, v[3] / max(1, v[4]) / 100.0, 100.0 * v[3] / max(1, v[4]) / max(1, v[2] - t0), v[4]))
