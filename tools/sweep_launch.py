"""Launch-geometry sweep on the headline workload (cover 1200x800 depth 50): one process per
configuration because the knobs are read at rt_create."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
from cpuraytracer_amd import HipRenderer, scenes
r = HipRenderer(0); r.upload(scenes.build_scene("cover", 1, 1200, 800))
spp = int(os.environ.get("SWEEP_SPP", "64"))
r.render(1200, 800, 1, 1 + spp, 50, 1)
best = None
for _ in range(3):
    st = r.render(1200, 800, 1, 1 + spp, 50, 1)
    ms = st.ms_render + st.ms_accumulate
    best = ms if best is None or ms < best else best
print(json.dumps({"scan": os.environ.get("RT_SCAN", "mfma"), "threads": os.environ.get("RT_BLOCK_THREADS"), "blocks_per_cu": os.environ.get("RT_BLOCKS_PER_CU"), "spp": spp,
                  "ms": best, "ms_trace": st.ms_render, "ms_acc": st.ms_accumulate, "Msamples_per_s": st.samples / best / 1e3}))
''' % ROOT
CONFIGS = os.environ.get("SWEEP_CONFIGS")
cfgs = [tuple(int(v) for v in c.split("x")) for c in CONFIGS.split(",")] if CONFIGS else None
for threads, bpcs in ([(t, (b,)) for t, b in cfgs] if cfgs else ((256, (2, 3, 4)), (512, (1, 2, 3)), (1024, (1, 2)))):
    for bpc in bpcs:
        env = dict(os.environ, RT_BLOCK_THREADS=str(threads), RT_BLOCKS_PER_CU=str(bpc))
        p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
        print(p.stdout.strip() or ("FAILED: " + p.stderr[-300:]), flush=True)
