"""Ad-hoc GPU bring-up script: HIP library vs oracle on small cases + a first timing."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle_py as O
from cpuraytracer_amd import HipRenderer

def eq(name, a, b):
    a = np.asarray(a); b = np.asarray(b)
    same = np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)
    nbad = int(np.count_nonzero(a != b))
    print(f"[{'OK ' if same else 'BAD'}] {name}: mismatches={nbad}/{a.size}" + ("" if same else f" maxabs={np.nanmax(np.abs(a.astype(np.float64)-b.astype(np.float64)))}"), flush=True)
    return same

r = HipRenderer(0)
rng = np.random.default_rng(0)
# units
idx = np.concatenate([np.arange(0, 3000), rng.integers(0, 2**31, 5000)]).astype(np.uint32)
for base in (2, 3, 4, 5, 7):
    eq(f"halton base {base}", r.unit_halton(idx, base), O.halton_array(idx, base))
x = np.concatenate([rng.uniform(0, 2*np.pi, 100000), [0, np.pi/2, np.pi, 1.5*np.pi, 2*np.pi]]).astype(np.float32)
eq("sin", r.unit_math(0, x), O.math_array(0, x)); eq("cos", r.unit_math(1, x), O.math_array(1, x))
xb = np.concatenate([rng.uniform(0, 1, 100000), [0, 1, 1e-30, 1e-45, 0.5]]).astype(np.float32)
for yv in (5.0, 16.0, 37.3, 1/2.2, 0.0, 32.0, 39.99):
    y = np.full_like(xb, np.float32(yv)); eq(f"pow y={yv}", r.unit_math(2, xb, y), O.math_array(2, xb, y))
yy = rng.uniform(0, 40, xb.shape[0]).astype(np.float32); eq("pow random y", r.unit_math(2, xb, yy), O.math_array(2, xb, yy))

orc = O.Oracle()
for name, aspect, W, H, depth in (("three", 2.0, 200, 100, 8), ("cover", 1.5, 1200, 800, 50)):
    sc = O.build_scene(name, 1, aspect)
    orc.upload(sc); r.upload(sc)
    n = 4000
    ijs = np.stack([rng.integers(0, W, n), rng.integers(0, H, n), rng.integers(1, 1025, n)], 1).astype(np.uint32)
    ro = orc.primary_rays(W, H, ijs); rg = r.unit_primary_rays(W, H, ijs)
    eq(f"{name}: primary rays", rg, ro)
    ho = orc.closest_hit(ro); hg = r.unit_closest_hit(ro)
    eq(f"{name}: closest hit", hg, ho)
    to, tro = orc.trace(W, H, ijs, depth, 1); tg, trg = r.unit_trace(W, H, ijs, depth, 1)
    ok = eq(f"{name}: per-sample radiance", tg, to); eq(f"{name}: per-sample traversals", trg, tro)
    if not ok:
        bad = np.nonzero((tg != to).any(axis=1))[0][:5]
        for b in bad: print("   sample", ijs[b], "gpu", tg[b], "orc", to[b], "trav", trg[b], tro[b])

# C1 full image
sc = O.build_scene("three", 1, 2.0); orc.upload(sc); r.upload(sc)
so = orc.render(200, 100, 1, 2, 8, 1); orc.resolve(); ho, lo = orc.download()
sg = r.render(200, 100, 1, 2, 8, 1); r.resolve(); hg, lg = r.download()
eq("C1 hdr", hg, ho); eq("C1 ldr", lg, lo); print("   C1 traversals gpu/orc", sg.traversals, so.traversals, "segments", sg.segments, so.segments)
# cover crop-ish: small full image, 4 spp
sc = O.build_scene("cover", 1, 1.5); orc.upload(sc); r.upload(sc)
so = orc.render(192, 128, 1, 5, 50, 1, threads=8); orc.resolve(); ho, lo = orc.download()
sg = r.render(192, 128, 1, 5, 50, 1); r.resolve(); hg, lg = r.download()
eq("cover 192x128x4 hdr", hg, ho); eq("cover 192x128x4 ldr", lg, lo); print("   traversals gpu/orc", sg.traversals, so.traversals)
# progressive continuation + sharded rows
sg2 = r.render(192, 128, 1, 3, 50, 1); sg3 = r.render(192, 128, 3, 5, 50, 1); r.resolve(); hg2, lg2 = r.download()
eq("cover progressive (1..2 then 3..4) == one shot", hg2, hg)
from cpuraytracer_amd import cyclic_rows
parts = []
for rank in range(4):
    rs = cyclic_rows(128, rank, 4)
    r.render(192, 128, 1, 5, 50, 1, rowset=rs); h, _ = r.download(ldr=False); parts.append(h)
full = np.zeros_like(hg)
L = r._L
for rank in range(4):
    rs = cyclic_rows(128, rank, 4)
    for lr in range(parts[rank].shape[0]):
        full[L.rt_rowset_global_row(rs, lr)] = parts[rank][lr]
eq("cover 4-way cyclic row shards reassembled == one shot", full, hg)

# first timing
for spp in (4, 16):
    st = r.render(1200, 800, 1, 1 + spp, 50, 1)
    ms = st.ms_render + st.ms_accumulate
    print(json.dumps({"cfg": "cover 1200x800", "spp": spp, "ms_trace": st.ms_render, "ms_acc": st.ms_accumulate,
                      "Msamples_per_s": st.samples / ms / 1e3, "trav_per_sample": st.traversals / st.samples,
                      "sphere_tests_per_s_T": st.traversals * 488 / (st.ms_render * 1e-3) / 1e12}), flush=True)
