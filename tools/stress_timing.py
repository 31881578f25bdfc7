"""Timing of the cell-grid stress scenes of tests/test_gpu_parity.py (GPU box): Msamples/s at 768x512, spp 8, per scene, under the
environment it is started with (compare RT_GRID=0).  usage: stress_timing.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle_py as oracle
import test_gpu_parity as T
from cpuraytracer_amd import HipRenderer
r = HipRenderer(0)
cases = {"grazing": (1, dict(n=4000, side=60.0, cam_o=(70.0, 0.25, 0.3), cam_l=(-60.0, 0.2, 0.0), vfov=20.0)),
         "far_camera": (2, dict(n=3000, side=40.0, cam_o=(2500.0, 900.0, -1800.0), cam_l=(0.0, 0.0, 0.0), vfov=2.5)),
         "wall": (3, dict(n=2500, side=30.0, cam_o=(5.0, 35.0, -60.0), cam_l=(0.0, 31.0, 0.0), plane="xy")),
         "clumped": (4, dict(n=2000, side=100.0, cam_o=(6.0, 3.0, -6.0), cam_l=(0.0, 0.2, 0.0), clump=1500)),
         "inside_layer": (6, dict(n=5000, side=50.0, cam_o=(0.37, 0.21, 0.41), cam_l=(20.0, 0.2, 3.0), vfov=80.0)),
         "tiny_spheres": (7, dict(n=6000, side=3.0, cam_o=(4.0, 0.6, -3.0), cam_l=(0.0, 0.02, 0.0), r_small=0.02))}
out = []
for name, (seed, kw) in cases.items():
    rng = np.random.default_rng(seed)
    sc = T._layer_scene(oracle, rng, kw.pop("n"), kw.pop("side"), kw.pop("cam_o"), kw.pop("cam_l"), **kw)
    r.upload(sc)
    r.render(768, 512, 1, 2, 20, 11)
    st = r.render(768, 512, 1, 9, 20, 11)
    out.append("%s %.0f" % (name, st.samples / (st.ms_render + st.ms_accumulate) / 1e3))
print("Msamples/s %s: %s" % ({k: v for k, v in os.environ.items() if k.startswith("RT_")}, ", ".join(out)))
