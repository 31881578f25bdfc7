"""Differential fuzz campaign (GPU box): many random scenes -- flat-filter, cell-grid and hierarchy class -- rendered as whole
images through the C ABI against the oracle (list semantics).  usage: fuzz_campaign.py first_seed n_scenes [RT_* env as usual]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle_py as oracle
import test_gpu_parity as T
from cpuraytracer_amd import HipRenderer
first, count = int(sys.argv[1]), int(sys.argv[2])
r = HipRenderer(0)
orc = oracle.Oracle()
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.choice([40, 60, 200, 400, 520, 700, 1500, 3000, 6000]))
    scale = float(rng.choice([1.0, 1.0, 1.0, 0.01, 30.0]))
    off = rng.choice([0.0, 0.0, 0.0, 50.0, 2000.0]) * rng.uniform(-1, 1, 3)
    sc, _ = T._fuzz_scene(oracle, seed, n, scale, tuple(off))
    nl = int(os.environ.get("FUZZ_LIGHTS", "0"))  # > 0: the scene gets a LIST of 1 + (0..nl) lights (one may point below the horizon)
    if nl:
        extra = int(rng.integers(0, nl + 1))
        sc.lights = [sc.sun] + [oracle.make_light(rng.normal(size=3) + [0, 0.4, 0], rng.uniform(0, 1, 3), float(rng.uniform(2000, 40000))) for _ in range(extra)]
    W, H, spp, depth = int(rng.choice([257, 320, 403])), int(rng.choice([160, 203])), int(rng.choice([1, 2, 3])), int(rng.choice([4, 20, 50]))
    r.upload(sc); orc.upload(sc)
    sg = r.render(W, H, 1, 1 + spp, depth, 9 + seed); hg, _ = r.download(ldr=False)
    so = orc.render(W, H, 1, 1 + spp, depth, 9 + seed, accel=oracle.ACCEL_PADDED_LIST, threads=16); ho, _ = orc.download()
    npx = int((hg.view(np.uint32) != ho.view(np.uint32)).any(axis=-1).sum())
    ok = npx == 0 and sg.traversals == so.traversals and sg.segments == so.segments
    if (seed - first) % 100 == 99:
        print("  ... %d scenes, %d mismatching, %.0f s" % (seed - first + 1, bad, time.time() - t0), flush=True)
    if not ok:
        bad += 1
        print("MISMATCH seed %d n %d scale %g off %s %dx%d spp %d depth %d: %d pixels, trav %d vs %d, seg %d vs %d" % (
            seed, sc.n, scale, off.tolist(), W, H, spp, depth, npx, sg.traversals, so.traversals, sg.segments, so.segments), flush=True)
print("fuzz campaign: %d scenes from seed %d, %d mismatching, %.0f s, env %s" % (count, first, bad, time.time() - t0,
      {k: v for k, v in os.environ.items() if k.startswith("RT_")}), flush=True)
sys.exit(1 if bad else 0)
