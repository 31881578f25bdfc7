"""How even are the ranks' strips?  bench.py --gpus 8 takes the slowest rank's time: trace time and traversals per sample of every
rank's strip (spp 1024 over 800 / 8 rows of the cover image), for several row-block sizes of the cyclic partition, on one GPU.
usage: shard_balance_probe.py [block_rows ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpuraytracer_amd import HipRenderer, scenes, distributed as D
W, H, N = 1200, 800, 8
sc = scenes.build_scene("cover", 1, W, H)
r = HipRenderer(0); r.upload(sc)
for br in [int(x) for x in sys.argv[1:]] or [4, 2, 1]:
    ms, tr = [], []
    for rank in range(N):
        rs = D.shard_rowset(H, rank, N, br)
        for rep in range(2):
            st = r.render(W, H, 1, 1 + 128 * N, 50, 1, rowset=rs)
        ms.append(st.ms_render); tr.append(st.traversals / st.samples)
    print("block_rows %d: trace ms per rank %s -> max / mean = %.4f; traversals per sample max / mean = %.4f"
          % (br, " ".join("%.2f" % m for m in ms), max(ms) / (sum(ms) / N), max(tr) / (sum(tr) / N)))
