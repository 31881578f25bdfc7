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
    b = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_line.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in b, key
    assert b["vs_baseline"] is None and b["dtype"] == "f32" and "workload" in b["config"] and "model" not in b["config"]
    assert 0.0 < b["roofline"]["frac"] <= 1.0 and b["cpu_baseline"]["kind"] == "port" and b["cpu_baseline"]["cores"] >= 1


def test_roofline_is_withheld_for_multi_gpu_lines_and_for_another_workload(tmp_path, monkeypatch):
    """the committed counters are those of the 1-GPU launch; an N-rank line must not price rank 0's strip with them."""
    cfg = bench.CONFIGS["c2"]
    d = json.load(open(os.path.join(ROOT, "profiles", cfg["pmc"])))
    for n in (2, 4, 8):
        rl, cull = bench.roofline_object(cfg, "kernel", 13.5, 4.0e8, 488, d["kernel_sources_sha256"], n_gpus=n)
        assert rl["frac"] is None and rl["achieved"] is None and rl["traffic"] is None and cull is None
        assert "withheld at n_gpus=%d" % n in rl["error"] and rl["launch_ms"] == 13.5
    # a summary that records another workload than the config's is refused like a stale hash
    other = dict(d, workload_key={"n_gpus": 1, "W": 640, "H": 480, "spp": 128})
    (tmp_path / "profiles").mkdir()
    json.dump(other, open(tmp_path / "profiles" / cfg["pmc"], "w"))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    rl, cull = bench.roofline_object(cfg, "kernel", 13.5, 4.0e8, 488, d["kernel_sources_sha256"])
    assert rl["frac"] is None and "workload" in rl["error"] and cull is None


def test_multi_gpu_line_extras_have_the_documented_shape():
    """What a --gpus N line adds: where the time of a step went per rank, and north_star's HBM-read equivalent
    as a LABELLED extra beside the measured HBM rate."""
    rows = [(13.1, 0.28, 0.02, 0.40, 14.0), (13.9, 0.28, 0.02, 0.05, 14.4), (12.7, 0.27, 0.02, 0.31, 13.5)]
    per, slowest = bench.per_rank_report(rows)
    assert slowest == 1 and [p["rank"] for p in per] == [0, 1, 2]
    assert set(per[0]) == {"rank", "render_ms", "accumulate_ms", "resolve_ms", "gather_ms", "step_wall_ms"} and per[2]["gather_ms"] == 0.31
    h = bench.hbm_read_equivalent(13.5, 475966656.0, 488, 1.562e9)
    assert h["bytes_per_launch"] == 16.0 * 488 * 475966656.0
    assert abs(h["GBps"] - h["bytes_per_launch"] / 13.5e-3 / 1e9) < 1e-6 and abs(h["frac_of_hbm_peak"] - h["GBps"] / 8000.0) < 1e-12
    assert h["frac_of_hbm_peak"] > 1.0 > h["measured_frac_of_hbm_peak"] > 0.0 and "NOT bytes moved" in h["label"]
    h2 = bench.hbm_read_equivalent(13.5, 4.0e8, 488, None)
    assert h2["measured_hbm_GBps"] is None and h2["measured_frac_of_hbm_peak"] is None


def test_strip_exchange_hands_rccl_the_same_buffers_as_the_gloo_rehearsal(monkeypatch):
    """The `nccl` branch of StripExchange.gather has only run at N = 1 on hardware; its tensors (one uint8 buffer per rank:
    HDR floats then LDR bytes, padded to the largest shard) must have the shapes, dtypes and contents the host-memory gloo
    branch exchanges, on every rank, for even and ragged heights (torch.distributed.gather is replaced by a recorder)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from cpuraytracer_amd import distributed as D
    for H, W, world in ((800, 12, 8), (50, 7, 3), (9, 5, 4)):
        for through_host in (False, True):
            calls = []

            def fake_gather(src, out, dst=0, group=None):
                calls.append((src, out, dst))
                if out is not None:
                    for r, o in enumerate(out):
                        o.copy_(sent[r])
            monkeypatch.setattr(dist, "gather", fake_gather)
            xs = [D.StripExchange(H, W, r, world, "cpu") for r in range(world)]
            rng = np.random.default_rng(H)
            for x in xs:
                x.hdr[:x.rows] = torch.from_numpy(rng.random((x.rows, W, 3), dtype=np.float32))
                x.ldr[:x.rows] = torch.from_numpy(rng.integers(0, 256, (x.rows, W, 3), dtype=np.uint8))
            sent = [x.buf.clone() for x in xs]
            for r in (world - 1, 0):
                ph, pl = xs[r].gather(through_host=through_host)
                src, out, dst = calls[-1]
                assert dst == 0 and src.dtype == torch.uint8 and src.is_contiguous()
                assert src.numel() == xs[0].rows_max * W * 3 * 5 == xs[r].buf.numel()  # 4 HDR bytes + 1 LDR byte per value
                assert torch.equal(src, sent[r])
                if r != 0:
                    assert out is None and ph is None and pl is None
                    continue
                assert len(out) == world and all(o.dtype == torch.uint8 and o.shape == src.shape for o in out)
                assert all(p.dtype == torch.float32 and tuple(p.shape) == (xs[0].rows_max, W, 3) for p in ph)
                assert all(p.dtype == torch.uint8 and tuple(p.shape) == (xs[0].rows_max, W, 3) for p in pl)
                full_h = D.assemble([p.numpy() for p in ph], H, world)
                full_l = D.assemble([p.numpy() for p in pl], H, world)
                for q in range(world):
                    for lr in range(xs[q].rows):
                        j = D.global_row(lr, q, world)
                        assert np.array_equal(full_h[j], xs[q].hdr[lr].numpy()) and np.array_equal(full_l[j], xs[q].ldr[lr].numpy())
