"""Diagnostic (GPU box): locate the paths on which the HIP render of a full-size config differs from the oracle's.
1. GPU render of the whole job; HDR rows against the CRCs in tests/golden/full_size_oracle_digests.json.
2. Rows whose pixels agree but whose traversal totals do not are found by bisection over row bands (oracle and GPU render the
   same row set; both count traversals and segments).
3. For every suspicious row: per pixel, then per sample (rt_unit_trace vs orc_unit_trace), radiance bits and traversal counts.
usage: find_mismatch.py c4 [RT_STASH value]"""
import json, os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle_py as O
name = sys.argv[1]
if len(sys.argv) > 2:
    os.environ["RT_STASH"] = sys.argv[2]
from cpuraytracer_amd import HipRenderer, scenes
rec = json.load(open(os.path.join(ROOT, "tests", "golden", "full_size_oracle_digests.json")))[name]
W, H, spp, depth, seed = rec["W"], rec["H"], rec["spp"], rec["depth"], rec["render_seed"]
sc = scenes.build_scene(rec["scene"], rec["scene_seed"], W, H, aperture=rec["aperture"])
r = HipRenderer(0); r.upload(sc)
orc = O.Oracle(); orc.upload(sc)
T = min(16, os.cpu_count() or 1)
st = r.render(W, H, 1, 1 + spp, depth, seed)
hdr, _ = r.download(ldr=False)
print("GPU totals", st.traversals, st.segments, "oracle", rec["traversals"], rec["segments"], flush=True)
rows = [j for j in range(H) if zlib.crc32(np.ascontiguousarray(hdr[j], dtype="<f4").tobytes()) != rec["hdr_row_crc32"][j]]
print("rows whose HDR differs from the oracle digest:", rows[:20], flush=True)

def band(j0, j1):
    rs = O.RtRowset(j0, j1 - j0, j1 - j0, 0, 1)
    so = orc.render(W, H, 1, 1 + spp, depth, seed, rowset=rs, accel=O.ACCEL_PADDED_LIST, threads=T)
    sg = r.render(W, H, 1, 1 + spp, depth, seed, rowset=rs)
    return (sg.traversals, sg.segments), (so.traversals, so.segments)

if (st.traversals, st.segments) != (rec["traversals"], rec["segments"]):
    lo, hi = 0, H
    while hi - lo > 1:
        mid = (lo + hi) // 2
        g, o = band(lo, mid)
        print("  rows [%d, %d): gpu %s oracle %s" % (lo, mid, g, o), flush=True)
        if g != o:
            hi = mid
        else:
            lo = mid
    if lo not in rows:
        rows.append(lo)
    print("row with a differing traversal count:", lo, flush=True)
for j in rows[:4]:
    ijs = np.array([[i, j, s] for i in range(W) for s in range(1, spp + 1)], dtype=np.uint32)
    rg, tg = r.unit_trace(W, H, ijs, depth, seed)
    from concurrent.futures import ThreadPoolExecutor
    chunks = np.array_split(np.arange(len(ijs)), T * 4)
    with ThreadPoolExecutor(T) as ex:
        parts = list(ex.map(lambda idx: orc.trace(W, H, ijs[idx], depth, seed, accel=O.ACCEL_PADDED_LIST), chunks))
    ro = np.concatenate([p[0] for p in parts]); to = np.concatenate([p[1] for p in parts])
    bad = np.flatnonzero((rg.view(np.uint32) != ro.view(np.uint32)).any(axis=1) | (tg != to))
    print("row %d: %d of %d samples differ" % (j, bad.size, len(ijs)))
    for k in bad[:10]:
        print("   (i, j, s) = %s: gpu rgb %s trav %d | oracle rgb %s trav %d" % (ijs[k].tolist(), rg[k], tg[k], ro[k], to[k]))
    np.save(os.path.join(ROOT, "gpurun_out", "mismatch_%s_row%d.npy" % (name, j)), ijs[bad])
