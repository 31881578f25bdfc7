"""Diagnostic (librt_hip_tl.so, -DRT_TIMELINE [-DRT_TIMELINE_RING]): per-wave records of one trace kernel of the cover scene --
when each wave claimed its last block, when it had nothing left to start, when it left -- to see what the end of a launch is
made of.  The records are plain per-wave stores; shared counters would serialise the leaving waves and fake a tail.
usage: timeline_bulk.py [spp [depth]]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cpuraytracer_amd import _capi
_capi.LIB_PATH = os.path.join(ROOT, "cpuraytracer_amd", "lib", "exp", "librt_hip_tl.so")
from cpuraytracer_amd import HipRenderer, scenes
W, H, spp = 1200, 800, int(sys.argv[1]) if len(sys.argv) > 1 else 128
DEPTH = int(sys.argv[2]) if len(sys.argv) > 2 else 50
r = HipRenderer(0); r.upload(scenes.build_scene("cover", 1, W, H))
L = _capi.load()
L.rt_debug_timeline_waves.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
buf = (C.c_ulonglong * (4096 * 8))()
r.render(W, H, 1, 1 + spp, DEPTH, 1)
for rep in range(3):
    st = r.render(W, H, 1, 1 + spp, DEPTH, 1)
    L.rt_debug_timeline_waves(r._h, buf)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
    start, drain, exit_, nblk, lastblk, lastclaim, iters, diters = [a[:, k] for k in range(8)]
    t0 = start.min(); us = lambda t: (t - t0) / 100.0
    life = (exit_ - start) / 100.0
    print("kernel %.3f ms: first exit %+.1f us, last exit %+.1f us, mean wave life %.1f us (%.1f %% of the longest), iterations per wave %.0f"
          % (st.ms_render, us(exit_.min()), us(exit_.max()), life.mean(), 100.0 * life.mean() / us(exit_.max()), iters.mean()))
    print("   nothing left to start: first wave %+.1f us, last %+.1f us; then per wave: mean %.1f us, %.1f iterations (max %d)"
          % (us(drain.min()), us(drain.max()), ((exit_ - drain) / 100.0).mean(), diters.mean(), diters.max()))
h, edges = np.histogram(us(exit_), bins=np.arange(us(exit_.min()) // 25 * 25, us(exit_.max()) + 25, 25))
print("   waves leaving per 25-us bucket:", " ".join("%d:%d" % (e, c) for e, c in zip(edges, h) if c))
order = np.argsort(drain)
nb_total = (W * H * spp + 127) // 128
for lo, hi in ((0, 1024), (1024, 2048), (2048, 3072), (3072, 3584), (3584, 4096)):
    w = order[lo:hi]
    print("   waves %4d..%4d by time of nothing-left-to-start %.0f..%.0f us: last claim at %.0f..%.0f us (mean %.0f us before), last block "
          "%.4f..%.4f of the launch, blocks per wave %d..%d, iterations after %.1f, exit mean %.0f us"
          % (lo, hi, us(drain[w].min()), us(drain[w].max()), us(lastclaim[w].min()), us(lastclaim[w].max()), (drain[w] - lastclaim[w]).mean() / 100.0,
             lastblk[w].min() / nb_total, lastblk[w].max() / nb_total, nblk[w].min(), nblk[w].max(), diters[w].mean(), us(exit_[w]).mean()))
L.rt_debug_timeline_last.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
lb = (C.c_uint * (4096 * 4))()
L.rt_debug_timeline_last(r._h, lb)
last = np.frombuffer(lb, dtype=np.uint32).reshape(4096, 4).astype(np.int64)
print("the ten waves that left last -- their last path: started at, depth, pixel (column, row), sample index, traversals")
for w in np.argsort(exit_)[-10:]:
    born, depth, q, trav = last[w]
    tile, rem = divmod(q, 64 * spp)
    pl = tile * 64 + (rem & 63)
    print("   wave %4d exit %.0f us, nothing to start since %.0f us: path started %.0f us, depth %d, pixel (%d, %d), sample %d, %d traversals"
          % (w, us(exit_[w]), us(drain[w]), born / 100.0, depth, pl % W, pl // W, rem >> 6, trav))
if hasattr(L, "rt_debug_timeline_ring") and os.environ.get("RT_TIMELINE_RING"):
    L.rt_debug_timeline_ring.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    rb = (C.c_ulonglong * (4096 * 16))()
    L.rt_debug_timeline_ring(r._h, rb)
    ring = np.frombuffer(rb, dtype=np.uint64).reshape(4096, 16)
    print("last claims of the four latest and two earliest waves (time us : live lanes at the claim : block/launch):")
    for w in list(order[-4:]) + list(order[:2]):
        rec = sorted((int(x) >> 24, (int(x) >> 17) & 127, (int(x) & 0x1ffff) * 8) for x in ring[w] if x)
        print("  wave %4d:" % w, " ".join("%.0f:%d:%.4f" % (t / 100.0, l, b / nb_total) for t, l, b in rec))
