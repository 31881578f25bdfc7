"""Diagnostic (librt_hip_tl.so, -DRT_TIMELINE): wall-clock landmarks inside the carrying trace kernel of one pipelined frame."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cpuraytracer_amd import _capi
_capi.LIB_PATH = os.path.join(ROOT, "cpuraytracer_amd", "lib", "exp", "librt_hip_tl.so")
from cpuraytracer_amd import HipRenderer, scenes
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1200, 800)
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 8
r = HipRenderer(0); r.upload(scenes.build_scene("cover", 1, W, H)); r.set_frame_pipelining(depth)
L = _capi.load(); L.rt_debug_timeline.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
out = (C.c_ulonglong * 16)()
for s in range(1, 21):
    r.render(W, H, s, s + 1, 50, 1, stats=False)
L.rt_debug_timeline(r._h, out)
for s in range(21, 26):
    r.render(W, H, s, s + 1, 50, 1, stats=False)
    L.rt_debug_timeline(r._h, out)
    v = list(out); t0 = v[0]
    print("frame %d: first staged %+.1f us, last staged %+.1f us, first exit %+.1f, last exit %+.1f us, mean wave life %.1f us, waves %d, mean iterations %.1f"
          % (s, (v[6] - t0) / 100.0, (v[1] - t0) / 100.0, (v[7] - t0) / 100.0, (v[2] - t0) / 100.0, v[3] / max(1, v[4]) / 100.0, v[4], v[5] / max(1, v[4])))
    print("   max iterations %d, waves beyond 14 iterations %d; their blocked iterations: queue not empty %d, cache not empty %d, not carriable %d, other %d"
          % (v[12], v[13], v[8], v[9], v[10], v[11]))
