"""What ONE rank does in bench.py --gpus N (weak scaling: spp = 128 N over 800 / N rows): trace and accumulate times of rank 0's
strip on one GPU.  The per-rank work is the same 122.88 M samples at every N; this checks that its shape (few rows, many
samples) costs no more."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpuraytracer_amd import HipRenderer, scenes
W, H = 1200, 800
sc = scenes.build_scene("cover", 1, W, H)
r = HipRenderer(0); r.upload(sc)
from cpuraytracer_amd import distributed as D
for N, spp in ((1, 128), (8, 1024), (4, 512), (2, 256)):
    rs = D.shard_rowset(H, 0, N)
    for rep in range(2):
        st = r.render(W, H, 1, 1 + spp, 50, 1, rowset=rs)
    print("N=%d spp=%d rows=%d: trace %.3f ms, accumulate %.3f ms, samples %d" % (N, spp, st.local_rows, st.ms_render, st.ms_accumulate, st.samples))
