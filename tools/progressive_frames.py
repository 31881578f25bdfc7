"""Progressive use like the reference's OnRender (one sample per pixel per frame, spheres-app.cpp:163-222): rt_render
without a stats request only enqueues the frame.  Prints the time per 1-spp frame without and with frame pipelining
(rt_set_frame_pipelining: the kernel of frame j carries its unfinished paths into the kernel of frame j+1 instead of running
their 51-iteration tail) and checks that the frames equal one N-spp call bit for bit."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cpuraytracer_amd import _capi
if os.environ.get("RT_HIP_LIB"):  # A/B runs against another build of the library (tool only)
    _capi.LIB_PATH = os.path.abspath(os.environ["RT_HIP_LIB"])
from cpuraytracer_amd import HipRenderer, scenes
W, H, N = 1200, 800, 256
sc = scenes.build_scene("cover", 1, W, H)
one = HipRenderer(0); one.upload(sc); one.render(W, H, 1, N + 1, 50, 1); one.resolve(); h1, _ = one.download()
MODES = [("pipelining depth", int(x)) for x in os.environ.get("RT_DEPTHS", "0,2,4,8").split(",") if x] + \
        [("frame batch", int(x)) for x in os.environ.get("RT_BATCHES", "4,8,16,32").split(",") if x] + \
        [("render-ahead", int(x)) for x in os.environ.get("RT_AHEADS", "4,8,16,32").split(",") if x]
for mode, depth in MODES:
    r = HipRenderer(0); r.upload(sc)
    if mode == "frame batch":
        r.set_frame_batch(depth)
    elif mode == "render-ahead":
        r.set_frame_lookahead(depth)
    else:
        r.set_frame_pipelining(depth)
    r.render(W, H, 1, 2, 50, 1, stats=False); r.synchronize()   # warm-up (buffers, LDS attribute)
    best = None
    for rep in range(3):
        r.clear()
        t0 = time.perf_counter()
        for s in range(1, N + 1):
            r.render(W, H, s, s + 1, 50, 1, stats=False)
        t_enqueue = time.perf_counter() - t0
        r.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    r.resolve(); hdr, _ = r.download()
    # display lag: frames made minus samples in the strip, seen by a reader right after frame 100 of a fresh accumulation
    r.clear()
    for s in range(1, 101):
        r.render(W, H, s, s + 1, 50, 1, stats=False)
    lag = 100 - r.committed_samples()
    r.synchronize()
    print("%s %d: %d frames of 1 spp: %.3f ms per frame (%.3f ms host enqueue), %.0f Msamples/s, display lag after frame 100: %d frames, == one shot: %s"
          % (mode, depth, N, best / N * 1e3, t_enqueue / N * 1e3, W * H * N / best / 1e6, lag, np.array_equal(hdr.view(np.uint32), h1.view(np.uint32))), flush=True)
    r.close()
