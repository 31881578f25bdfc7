import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, ctypes as C
from oracle import oracle_py as oracle
import test_gpu_parity as T
from cpuraytracer_amd import HipRenderer
seed, n = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(1000 + seed)
extent = rng.choice([3.0, 12.0, 60.0])
centers = rng.uniform(-extent, extent, size=(n, 3)); centers[:, 1] = np.abs(centers[:, 1]) * 0.25
radii = np.exp(rng.uniform(np.log(0.02), np.log(1.5), n)) * extent / 12.0
if seed % 2 == 0:
    centers = np.concatenate([centers, [[0.0, -400.0 - radii.max(), 0.0]]]); radii = np.concatenate([radii, [400.0]])
n = len(radii)
types = rng.choice([0, 0, 0, 1, 2, 3], n).astype(np.uint32)
sc = T._custom_scene(oracle, centers.astype(np.float32), radii.astype(np.float32), types, rng.uniform(-1, 1, 3) * extent * 1.2 + [0, extent * 0.4, 0],
                     rng.uniform(-0.3, 0.3, 3) * extent, float(rng.uniform(20, 70)), 1.5, aperture=float(rng.choice([0.0, 0.3, 2.0])))
k255 = np.float32(1) / np.float32(255)
sc.materials["tex_type"] = rng.integers(0, 2, n); sc.materials["tiling"] = rng.choice([4.0, 50.0, 2500.0], n)
sc.materials["rgb0"] = rng.integers(0, 256, (n, 3)).astype(np.float32) * k255; sc.materials["rgb1"] = rng.integers(0, 256, (n, 3)).astype(np.float32) * k255
sc.materials["smoothness"] = np.where(types == 1, 0.0, rng.uniform(1.0, 64.0, n)).astype(np.float32)
sc.materials["ior"] = rng.uniform(1.1, 2.4, n).astype(np.float32)
sc.materials["luminance"] = np.where(types == 3, rng.uniform(100.0, 20000.0, n), 0.0).astype(np.float32)
print("extent", extent, "n", n, "types", np.bincount(types))
orc = oracle.Oracle(); orc.upload(sc)
W, H, m = 300, 200, 3000
ijs = np.stack([rng.integers(0, W, m), rng.integers(0, H, m), rng.integers(1, 600, m)], 1).astype(np.uint32)
ro, to = orc.trace(W, H, ijs, 30, 77 + seed, accel=oracle.ACCEL_BVH)
for mode in ("mfma", "valu"):
    os.environ["RT_SCAN"] = mode
    r = HipRenderer(0); r.upload(sc)
    rg, tg = r.unit_trace(W, H, ijs, 30, 77 + seed)
    bad = np.nonzero((rg.view(np.uint32) != ro.view(np.uint32)).any(axis=1))[0]
    print(mode, "bad samples", len(bad), "trav mismatches", int((tg != to).sum()))
    for b in bad[:6]:
        print("   ", ijs[b], "gpu", rg[b], "orc", ro[b], "trav", tg[b], to[b])
    # find the depth where they diverge
    for b in bad[:3]:
        for d in range(0, 31):
            a1, t1 = r.unit_trace(W, H, ijs[b:b+1], d, 77 + seed); a2, t2 = orc.trace(W, H, ijs[b:b+1], d, 77 + seed, accel=oracle.ACCEL_BVH)
            if not np.array_equal(a1.view(np.uint32), a2.view(np.uint32)) or t1[0] != t2[0]:
                print("      first divergence at max_depth", d, a1, a2, t1, t2); break
    r.close()
