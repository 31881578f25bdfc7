"""Diagnostic (GPU box): the fuzz image of tests/test_gpu_parity.py::test_random_scene_images_equal_the_oracle_render with the hit
stash on and off against the oracle, per pixel and per sample.  usage: debug_stash.py [seed n W H spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle_py as oracle
import test_gpu_parity as T
from cpuraytracer_amd import HipRenderer
a = [int(v) for v in sys.argv[1:6]] if len(sys.argv) >= 6 else [31, 60, 403, 203, 3]
seed, n, W, H, spp = a
depth = 20
sc, _ = T._fuzz_scene(oracle, seed, n, 1.0, (0.0, 0.0, 0.0))
orc = oracle.Oracle(); orc.upload(sc)
so = orc.render(W, H, 1, 1 + spp, depth, 9 + seed, accel=oracle.ACCEL_BVH, threads=16); ho, _ = orc.download()
for mode in ("1", "0"):
    os.environ["RT_STASH"] = mode
    r = HipRenderer(0); r.upload(sc)
    sg = r.render(W, H, 1, 1 + spp, depth, 9 + seed); hg, _ = r.download(ldr=False)
    bad = np.argwhere((hg.view(np.uint32) != ho.view(np.uint32)).any(axis=-1))
    print("RT_STASH=%s: %d pixels differ; traversals %d vs %d, segments %d vs %d" % (mode, len(bad), sg.traversals, so.traversals, sg.segments, so.segments))
    for j, i in bad[:8]:
        ijs = np.array([[i, j, s] for s in range(1, 1 + spp)], dtype=np.uint32)
        rg, tg = r.unit_trace(W, H, ijs, depth, 9 + seed)
        ro, to = orc.trace(W, H, ijs, depth, 9 + seed)
        print("  pixel (%d, %d): image gpu %s oracle %s" % (i, j, hg[j, i], ho[j, i]))
        for s in range(spp):
            print("    s=%d unit_trace gpu %s trav %d | oracle %s trav %d %s" % (s + 1, rg[s], tg[s], ro[s], to[s],
                  "" if np.array_equal(rg[s].view(np.uint32), ro[s].view(np.uint32)) else "<-- differs"))
    r.close()
