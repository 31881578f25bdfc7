"""Time per pipelined frame as a function of samples per frame and image size: fixed cost vs per-sample cost."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpuraytracer_amd import HipRenderer, scenes
for (W, H) in ((1200, 800), (600, 400)):
    sc = scenes.build_scene("cover", 1, W, H)
    for depth in (0, 8):
        for spf in (1, 2, 4, 8):
            r = HipRenderer(0); r.upload(sc); r.set_frame_pipelining(depth)
            N = 96
            r.render(W, H, 1, 1 + spf, 50, 1, stats=False); r.synchronize(); r.clear()
            best = 1e9
            for rep in range(3):
                r.clear()
                t0 = time.perf_counter()
                for f in range(N):
                    r.render(W, H, 1 + f * spf, 1 + (f + 1) * spf, 50, 1, stats=False)
                r.synchronize()
                best = min(best, time.perf_counter() - t0)
            print("%dx%d depth %d spf %d: %.3f ms/frame  %.0f Msamples/s" % (W, H, depth, spf, best / N * 1e3, W * H * spf * N / best / 1e6), flush=True)
            r.close()
