"""Image-level differential fuzz (tool): random scenes rendered as whole images on the GPU (work order from pilot rays, queue,
ordered accumulation, resolve) against the oracle's render of the same scene.  usage: debug_fuzz_image.py seed n_spheres"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle_py as oracle
import test_gpu_parity as T
from cpuraytracer_amd import HipRenderer
seed, n = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(5000 + seed)
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
W, H, spp, depth = int(rng.choice([257, 320, 403])), int(rng.choice([160, 203])), int(rng.choice([2, 3, 5])), 20
orc = oracle.Oracle(); orc.upload(sc)
so = orc.render(W, H, 1, 1 + spp, depth, 9 + seed, accel=oracle.ACCEL_BVH, threads=16); orc.resolve(); ho, lo = orc.download()
r = HipRenderer(0); r.upload(sc)
sg = r.render(W, H, 1, 1 + spp, depth, 9 + seed); r.resolve(); hg, lg = r.download()
bad = int((hg.view(np.uint32) != ho.view(np.uint32)).any(axis=-1).sum())
print("seed %d n %d extent %g %dx%d spp %d: HDR pixels differing %d, LDR bytes differing %d, traversals %s segments %s"
      % (seed, n, extent, W, H, spp, bad, int((lg != lo).sum()), sg.traversals == so.traversals, sg.segments == so.segments))
sys.exit(1 if bad else 0)
