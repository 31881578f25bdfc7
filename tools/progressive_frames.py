"""Progressive use like the reference's OnRender (one sample per pixel per frame, spheres-app.cpp:163-222): rt_render
without a stats request only enqueues the frame.  Prints the time per 1-spp frame and checks that 128 such frames equal
one 128-spp call bit for bit.  (A captured hipGraph per frame was measured here and gave the same 0.56 ms per frame: a
1-spp frame is bound by the start-up and tail of the persistent kernel, not by its nine API calls.)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cpuraytracer_amd import HipRenderer, scenes
W, H, N = 1200, 800, 128
sc = scenes.build_scene("cover", 1, W, H)
r = HipRenderer(0); r.upload(sc)
r.render(W, H, 1, 2, 50, 1); r.synchronize()
t0 = time.perf_counter()
for s in range(1, N + 1):
    r.render(W, H, s, s + 1, 50, 1, stats=False)
t_enqueue = time.perf_counter() - t0
r.synchronize()
dt = time.perf_counter() - t0
r.resolve(); hdr, _ = r.download()
print("%d frames of 1 spp: %.2f ms per frame (%.2f ms of it host enqueue), %.0f Msamples/s" % (N, dt / N * 1e3, t_enqueue / N * 1e3, W * H * N / dt / 1e6))
one = HipRenderer(0); one.upload(sc); one.render(W, H, 1, N + 1, 50, 1); one.resolve(); h1, _ = one.download()
print("progressive == one shot:", np.array_equal(hdr.view(np.uint32), h1.view(np.uint32)))
