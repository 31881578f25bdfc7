"""Summarise tools/c5_placement_sweep.sh's output directory into profiles/r04_c5_placement.json (BASELINE configs[4])."""
import csv, glob, json, os, sys
src = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r04_c5_placement.json")
LABEL = {
    "default": "cell-grid scan: float4 one-sphere bounds + exact tables through L1/L2, cells + 63-record hit stash in LDS, 256x256 shadow index (shipped)",
    "quant_lds": "cell-grid scan: QUANTISED one-sphere bounds in LDS (RT_GRID_QUANT=1; 40 KB, stash 38 records), exact tables through L1/L2",
    "shadow64": "as shipped but the round-3 shadow index (64x64 cells, RT_SHADOW_CELLS=64)",
    "hierarchy_lds": "bounds hierarchy (RT_GRID=0): four levels of bounds in LDS, exact tables through L1/L2",
    "hierarchy_l2": "bounds hierarchy with its bounds through L1/L2 as well (RT_GRID=0 RT_TREE_LDS=0)",
    "valu_streamed": "exhaustive VALU scan, the whole sphere list streamed per ray from L2/HBM (RT_SCAN=valu; 1024x1024 spp 2: the literal north-star kernel)",
}
rows = []
for name, label in LABEL.items():
    p = os.path.join(src, name + ".json")
    if not os.path.exists(p) or not open(p).read().strip():
        continue
    b = json.loads(open(p).read().strip().split("\n")[-1])
    rec = {"placement": name, "what": label, "image": "%dx%d spp %d" % (b["W"], b["H"], b["spp"]), "ms_trace_plus_accumulate": b["ms"],
           "Gsamples_per_s": b["Msamples_per_s"] / 1e3, "hdr_sha1": b["hdr_sha1"]}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        tot, n = 0.0, 0
        for f in glob.glob(os.path.join(src, "%s_%s" % (name, c), "*", "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if "rt_trace_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    tot += float(r["Counter_Value"]); n += 1
        # the PMC run renders twice (1-spp warm-up + the job): the job's launch dominates; counters are in KB
        rec[c + "_KB_all_trace_launches"] = tot
    fe, wr = rec.get("FETCH_SIZE_KB_all_trace_launches", 0.0), rec.get("WRITE_SIZE_KB_all_trace_launches", 0.0)
    sec = b["ms"] * 1e-3
    rec["hbm_read_GBps (FETCH_SIZE x 2: the guide's gfx950 correction)"] = 2.0 * fe * 1024.0 / sec / 1e9
    rec["hbm_write_GBps"] = wr * 1024.0 / sec / 1e9
    rows.append(rec)
json.dump({"config": "c5 (BASELINE configs[4]): grid10k, 10,004 spheres, one MI355X", "rows": rows,
           "note": "same bits for every placement at the same image size (hdr_sha1); rates are whole-launch averages; HBM peak 8,000 GB/s"}, open(out, "w"), indent=1)
for r in rows:
    print("%-14s %7.3f Gs/s  read %7.1f GB/s  write %7.1f GB/s  %s" % (r["placement"], r["Gsamples_per_s"], r["hbm_read_GBps (FETCH_SIZE x 2: the guide's gfx950 correction)"], r["hbm_write_GBps"], r["image"]))
