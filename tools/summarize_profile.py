"""Turn a tools/profile_round.sh output directory (gpurun_out/<dir>) into the committed summaries under profiles/:
  <prefix>_bench_line.json, <prefix>_bench_under_rocprof.json, <prefix>_bench_kernel_stats.csv,
  <prefix>_pmc_summary.json, <prefix>_hbm_traffic.json.
usage: python tools/summarize_profile.py gpurun_out/profR2 profiles/r03 [c3|c4|c5]
(a config suffix gives profiles/r03_*_c4.json etc.; the default, headline config has none)"""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src, prefix = sys.argv[1], sys.argv[2]
suffix = ("_" + sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] != "c2" else ""

def last_json_line(path):
    for ln in reversed(open(path).read().strip().split("\n")):
        if ln.startswith("{"):
            return json.loads(ln)
    return None

bench = last_json_line(os.path.join(src, "bench.json"))
json.dump(bench, open(prefix + "_bench_line" + suffix + ".json", "w"), indent=1)
json.dump(last_json_line(os.path.join(src, "bench_kt.json")), open(prefix + "_bench_under_rocprof" + suffix + ".json", "w"), indent=1)
ks = glob.glob(os.path.join(src, "kt", "*", "*kernel_stats.csv"))[0]
shutil.copy(ks, prefix + "_bench_kernel_stats" + suffix + ".csv")
raw = {"trace": {}, "accumulate": {}}
rows_per_counter = {}
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = "trace" if "rt_trace_kernel" in r["Kernel_Name"] else ("accumulate" if "rt_accumulate" in r["Kernel_Name"] else None)
        if k:
            raw[k][r["Counter_Name"]] = raw[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            if k == "trace":
                rows_per_counter.setdefault(r["Counter_Name"], set()).add(r.get("Dispatch_Id", r.get("Correlation_Id", "0")))
t = raw["trace"]
launches = max(len(v) for v in rows_per_counter.values())  # dispatches of the trace kernel in one PMC pass (bench --steps 1 --warmup 0: one)
assert all(len(v) == launches for v in rows_per_counter.values()), "PMC passes saw different numbers of trace launches"
for k in raw:
    raw[k] = {c: v / launches for c, v in raw[k].items()}  # per launch
t = raw["trace"]
stats = {r["Name"]: r for r in csv.DictReader(open(ks))}
trace_name = [n for n in stats if "rt_trace_kernel" in n][0]
launch_ms = float(stats[trace_name]["AverageNs"]) * 1e-6
# FETCH_SIZE / WRITE_SIZE are in KiB per MI355X_MICROARCH.md; FETCH_SIZE doubled per its gfx950 correction (an upper bound here)
fetch_b = t.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0
write_b = t.get("WRITE_SIZE", 0.0) * 1024.0
xcds, simds = 8, 1024
cyc = t["GRBM_GUI_ACTIVE"] / xcds  # GRBM_GUI_ACTIVE is summed over the 8 XCDs
CYC_PER_VALU = 2.164  # SIMD cycles per wave-wide VOP2 instruction at the best measured rate (profiles/r01_valu_rate*_microbench.jsonl)
import bench as _bench
derived = {
    "launches_in_pmc_pass": 1, "raw_counters_are": "per launch (sums over the pass / %d trace launches)" % launches,
    "launch_ms_rocprof_avg": launch_ms,
    "hbm_fetch_bytes": fetch_b, "hbm_write_bytes": write_b, "hbm_bytes_per_launch": fetch_b + write_b,
    "note_traffic": "WRITE_SIZE is the per-sample buffer: paths x 12 B algorithmic (C2: 122.88e6 x 12 = 1.4746e9 B). FETCH_SIZE is in KiB and doubled per the guide's gfx950 correction (an upper bound for the narrow table reads here).",
    "valu_insts_per_launch": t["SQ_INSTS_VALU"], "salu_insts_per_launch": t["SQ_INSTS_SALU"], "lds_insts_per_launch": t["SQ_INSTS_LDS"],
    "mfma_insts_per_launch": t["SQ_INSTS_MFMA"],
    "valu_lane_utilization": t["SQ_THREAD_CYCLES_VALU"] / (t["SQ_INSTS_VALU"] * 64.0),
    "effective_clock_GHz": cyc / (launch_ms * 1e-3) / 1e9,
    "valu_issue_busy_estimate": t["SQ_ACTIVE_INST_VALU"] * CYC_PER_VALU / (cyc * simds),
    "mfma_busy": t["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * simds),
}
wc = t["SQ_WAVE_CYCLES"]
derived["wave_time_split"] = {"issuing": t["SQ_ACTIVE_INST_ANY"] / wc, "wait_inst_issue": t["SQ_WAIT_INST_ANY"] / wc, "wait_cnt_or_barrier": t["SQ_WAIT_ANY"] / wc}
if "SQ_LDS_BANK_CONFLICT" in t and t.get("SQ_LDS_IDX_ACTIVE"):
    derived["lds_bank_conflict_frac"] = t["SQ_LDS_BANK_CONFLICT"] / t["SQ_LDS_IDX_ACTIVE"]
# the hash of the kernel sources comes from the RUNS that were profiled (every bench line prints roofline.kernel_sources_sha256),
# never from the working tree at summary time: an edit between profile_round.sh and this script must not restamp old counters
run_hashes = {}
for f in ["bench.json", "bench_kt.json"] + sorted(os.path.basename(x) for x in glob.glob(os.path.join(src, "pmc_*.json"))):
    ln = last_json_line(os.path.join(src, f))
    assert ln is not None, "no bench line in " + f
    run_hashes[f] = ln["roofline"]["kernel_sources_sha256"]
    assert ln["n_gpus"] == 1 and ln["config"]["name"] == bench["config"]["name"], f + " profiled another workload"
assert len(set(run_hashes.values())) == 1, "the passes of this profile ran different kernel builds: %s" % run_hashes
profiled_hash = run_hashes["bench.json"]
if profiled_hash != _bench.kernel_sources_hash():
    print("summarize_profile.py: NOTE: the working tree's kernel sources differ from the profiled build (%s...): the summary keeps "
          "the profiled hash and bench.py will withhold the roofline until the profile is redone" % profiled_hash[:12], file=sys.stderr)
_cfg = _bench.CONFIGS[bench["config"]["name"]]
out = {"kernel_sources_sha256": profiled_hash,
       "workload_key": {"n_gpus": bench["n_gpus"], "W": _cfg["W"], "H": _cfg["H"], "spp": bench["config"]["spp"], "config": bench["config"]["name"]},
       "command": "tools/profile_round.sh: rocprofv3 --kernel-trace --stats (durations) and four separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ group | SQ+MFMA+GRBM group) over python3 bench.py; summarised by tools/summarize_profile.py",
       "workload": bench["config"]["workload"], "kernel": trace_name, "raw_counters": raw, "derived_trace_kernel": derived}
prev = prefix + "_pmc_summary" + suffix + ".json"
if os.path.exists(prev):
    try:
        old = json.load(open(prev))
        hist = old.get("history", {})
        if "derived_trace_kernel" in old and old.get("kernel") != trace_name:
            d = old["derived_trace_kernel"]
            hist["before_" + trace_name[:40]] = {k: d.get(k) for k in ("valu_insts_per_launch", "mfma_insts_per_launch", "valu_lane_utilization", "valu_issue_busy_estimate", "mfma_f32_busy", "mfma_busy")}
        out["history"] = hist
    except Exception:
        pass
json.dump(out, open(prev, "w"), indent=1)
# the bench line of this profile run was printed before this summary existed: fill in its roofline object now, from the
# live launch duration that line measured and the counters of the PMC passes of the same run
for tag in ("_bench_line", "_bench_under_rocprof"):
    path = prefix + tag + suffix + ".json"
    b = json.load(open(path))
    if b and b.get("roofline", {}).get("frac") is None:
        cfg = _bench.CONFIGS[b["config"].get("name", "c2")]
        n_sph = b["config"].get("n_spheres") or int(b["config"]["workload"].split("(")[1].split(" ")[0])
        trav = b["config"]["traversals_per_sample"] * b["config"]["samples_per_step"]
        rl, cull = _bench.roofline_object(cfg, b["roofline"]["kernel"], b["roofline"]["launch_ms"], trav, n_sph, out["kernel_sources_sha256"])
        rl["filled_in_by"] = "tools/summarize_profile.py (the PMC passes ran after this line was printed)"
        b["roofline"], b["cull_factor"] = rl, cull
        json.dump(b, open(path, "w"), indent=1)
json.dump({"hbm_bytes_per_launch": fetch_b + write_b, "source": os.path.basename(prev) + " (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes; FETCH doubled per the gfx950 correction)"},
          open(prefix + "_hbm_traffic" + suffix + ".json", "w"))
print(json.dumps(derived, indent=1))
