"""bench.py plumbing that needs no GPU: the roofline object is a fraction of a hardware limit, recomputable from the committed
PMC summary, and that summary belongs to the kernel sources in the tree."""
import json
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


@pytest.mark.parametrize("name", ["c2", "c3", "c4", "c5"])
def test_committed_pmc_summary_belongs_to_these_kernel_sources(name):
    """tools/profile_round.sh + summarize_profile.py must be re-run after the last kernel edit of a round: bench.py withholds the
    roofline fraction when the hash recorded with the counters is not the hash of the sources the library is built from."""
    cfg = bench.CONFIGS[name]
    path = os.path.join(ROOT, "profiles", cfg["pmc"])
    assert os.path.exists(path), "run tools/profile_round.sh <dir> %s and tools/summarize_profile.py" % name
    d = json.load(open(path))
    assert d["kernel_sources_sha256"] == bench.kernel_sources_hash(), "profiles/%s is stale: re-profile" % cfg["pmc"]


@pytest.mark.parametrize("name", ["c2", "c3", "c4", "c5"])
def test_roofline_object_is_a_fraction_recomputable_from_the_summary(name):
    cfg = bench.CONFIGS[name]
    d = json.load(open(os.path.join(ROOT, "profiles", cfg["pmc"])))
    t, k = d["raw_counters"]["trace"], d["derived_trace_kernel"]
    launch_ms = k["launch_ms_rocprof_avg"]
    rl, cull = bench.roofline_object(cfg, "kernel", launch_ms, 4.0e8, 488, d["kernel_sources_sha256"])
    assert rl["bound"] == "valu-issue" and rl["unit"] == "Tlane-op/s" and rl["peak"] == 78.6
    want = t["SQ_THREAD_CYCLES_VALU"] / (launch_ms * 1e-3) / 1e12
    assert abs(rl["achieved"] - want) < 1e-9 * want and 0.0 < rl["frac"] <= 1.0 and abs(rl["frac"] - want / 78.6) < 1e-12
    assert 0.0 < rl["issue_slot_frac"] <= 1.0 and 0.0 < rl["lane_utilisation"] <= 1.0
    assert rl["traffic"] == k["hbm_bytes_per_launch"] and cull["value"] > 0.0  # (the scan count passed here is made up)
    # another build's counters are refused
    rl2, cull2 = bench.roofline_object(cfg, "kernel", launch_ms, 4.0e8, 488, "0" * 64)
    assert rl2["frac"] is None and "error" in rl2 and cull2 is None


def test_bench_line_committed_with_the_profile_has_the_contract_fields():
    b = json.load(open(os.path.join(ROOT, "profiles", "r02_bench_line.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in b, key
    assert b["vs_baseline"] is None and b["dtype"] == "f32" and "workload" in b["config"] and "model" not in b["config"]
    assert 0.0 < b["roofline"]["frac"] <= 1.0 and b["cpu_baseline"]["kind"] == "port" and b["cpu_baseline"]["cores"] >= 1
