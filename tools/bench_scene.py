"""Timing of arbitrary named scenes/configs through the C ABI (no oracle): scene W H spp depth [aperture]."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cpuraytracer_amd import HipRenderer, scenes
name, W, H, spp, depth = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
ap = float(sys.argv[6]) if len(sys.argv) > 6 else -1.0
r = HipRenderer(0)
sc = scenes.build_scene(name, 1, W, H, aperture=ap)
r.upload(sc)
r.render(W, H, 1, 2, depth, 1)
best = None
for _ in range(2):
    st = r.render(W, H, 1, 1 + spp, depth, 1)
    ms = st.ms_render + st.ms_accumulate
    best = ms if best is None or ms < best else best
print(json.dumps({"scene": name, "n": sc.n, "W": W, "H": H, "spp": spp, "depth": depth, "aperture": ap, "env": {k: v for k, v in os.environ.items() if k.startswith("RT_")},
                  "ms": best, "Msamples_per_s": st.samples / best / 1e3, "trav_per_sample": st.traversals / st.samples, "passes": st.passes}))
