"""Diagnostic (librt_hip_tl.so, -DRT_TIMELINE): per-wave records of the carrying trace kernel of pipelined 1-spp frames
(plain per-wave stores; shared counters would serialise the leaving waves).  usage: timeline.py [W H [depth]]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cpuraytracer_amd import _capi
_capi.LIB_PATH = os.path.join(ROOT, "cpuraytracer_amd", "lib", "exp", "librt_hip_tl.so")
from cpuraytracer_amd import HipRenderer, scenes
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1200, 800)
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 8
r = HipRenderer(0); r.upload(scenes.build_scene("cover", 1, W, H)); r.set_frame_pipelining(depth)
L = _capi.load(); L.rt_debug_timeline_waves.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
buf = (C.c_ulonglong * (4096 * 8))()
for s in range(1, 21):
    r.render(W, H, s, s + 1, 50, 1, stats=False)
for s in range(21, 25):
    r.render(W, H, s, s + 1, 50, 1, stats=False)
    L.rt_debug_timeline_waves(r._h, buf)   # waits for the stream: the records are those of this frame's kernel
    a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
    start, staged, exit_, nblk, carried, lastclaim, iters, _ = [a[:, k] for k in range(8)]
    t0 = start.min(); us = lambda t: (t - t0) / 100.0
    life = us(exit_) - us(start)
    print("frame %d: waves start %.1f..%.1f us, staged %.1f..%.1f, exits %.1f..%.1f us (5/50/95 %%: %.0f / %.0f / %.0f); mean wave life %.1f us = %.0f %% of the kernel"
          % (s, us(start).min(), us(start).max(), us(staged).min(), us(staged).max(), us(exit_).min(), us(exit_).max(),
             *np.percentile(us(exit_), [5, 50, 95]), life.mean(), 100 * life.mean() / us(exit_).max()))
    print("   iterations per wave: mean %.1f, 5/50/95 %%: %d / %d / %d, max %d; blocks opened per wave %.2f; paths carried out per wave: mean %.1f, max %d, total %d"
          % (iters.mean(), *np.percentile(iters, [5, 50, 95]), iters.max(), nblk.mean(), carried.mean(), carried.max(), carried.sum()))
    cin = lastclaim & 255; lc = (lastclaim >> 8) / 100.0
    for lo, hi in ((0, 9), (9, 11), (11, 13), (13, 15), (15, 99)):
        m = (iters >= lo) & (iters < hi)
        if m.sum():
            print("   waves with %2d..%2d iterations: %4d; carried in %.1f, blocks opened %.2f (max %d), last block opened at %.0f us, iterations with nothing left to start %.1f, carried out %.1f"
                  % (lo, hi - 1, m.sum(), cin[m].mean(), nblk[m].mean(), nblk[m].max(), lc[m].mean(), a[:, 7][m].mean(), carried[m].mean()))
    it_us = (us(exit_) - us(staged)) / np.maximum(iters, 1)
    print("   time per iteration: mean %.1f us (5/95 %%: %.1f / %.1f)" % (it_us.mean(), *np.percentile(it_us, [5, 95])))
