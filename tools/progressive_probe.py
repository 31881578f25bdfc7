"""N pipelined 1-spp frames (for rocprofv3 --kernel-trace): argv = depth [W H]."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpuraytracer_amd import HipRenderer, scenes
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 8
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
H = int(sys.argv[3]) if len(sys.argv) > 3 else 800
N = 64
r = HipRenderer(0); r.upload(scenes.build_scene("cover", 1, W, H))
r.set_frame_pipelining(depth)
for s in range(1, N + 1):
    r.render(W, H, s, s + 1, 50, 1, stats=False)
r.synchronize()
print("committed", r.committed_samples())
