"""Timing of arbitrary named scenes/configs through the C ABI (no oracle): scene W H spp depth [aperture]."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cpuraytracer_amd import _capi
if os.environ.get("RT_HIP_LIB"):  # A/B runs against another build of the library (tool only)
    _capi.LIB_PATH = os.path.abspath(os.environ["RT_HIP_LIB"])
from cpuraytracer_amd import HipRenderer, scenes
name, W, H, spp, depth = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
ap = float(sys.argv[6]) if len(sys.argv) > 6 else -1.0
r = HipRenderer(0)
sc = scenes.build_scene(name, 1, W, H, aperture=ap)
r.upload(sc)
r.render(W, H, 1, 2, depth, 1)
best = None
for _ in range(int(os.environ.get("RT_BENCH_REPS", "3"))):
    st = r.render(W, H, 1, 1 + spp, depth, 1)
    ms = st.ms_render + st.ms_accumulate
    best = ms if best is None or ms < best else best
import hashlib
_h, _ = r.download(ldr=False)
sig = hashlib.sha1(_h.tobytes()).hexdigest()[:12]  # A/B variants must give the same bits
print("%s n=%d %dx%d spp=%d: %.2f ms  %.1f Msamples/s  env=%s" % (name, sc.n, W, H, spp, best, st.samples / best / 1e3,
      {k: v for k, v in os.environ.items() if k.startswith("RT_")}), file=sys.stderr)
print(json.dumps({"scene": name, "n": sc.n, "W": W, "H": H, "spp": spp, "depth": depth, "aperture": ap, "env": {k: v for k, v in os.environ.items() if k.startswith("RT_")},
                  "ms": best, "Msamples_per_s": st.samples / best / 1e3, "trav_per_sample": st.traversals / st.samples, "passes": st.passes, "hdr_sha1": sig}))
