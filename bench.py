#!/usr/bin/env python3
"""bench.py — headline benchmark of the render-loop hot path on MI355X.

Metric (BASELINE.json): Msamples/s = W*H*spp / render-seconds on the RTIOW-style cover scene
(488 spheres, all materials), 1200x800, depth 50.  A "step" is one pass of the hot path over one
batch: ray generation + trace + ordered accumulate + tonemap of the whole image at spp samples per
pixel (the reference's DrawBitmap bracket, spheres-app.cpp:20-26, minus its Direct2D blit).
Inputs (scene tables) are resident in HBM before the timed region starts.

--config c2 (default): BASELINE.json configs[1], 1200x800 spp 128 per GPU.
        N > 1: spp = 128 * N, image rows sharded cyclically across ranks, one RCCL gather of the strips
        to rank 0 per step (configs[2] is N = 8: spp 1024).  Per-GPU work is constant: weak scaling.
--config c4: configs[3], cover scene, aperture 2.0, 1920x1080, spp 512 (divergent lens sampling).
--config c5: configs[4], grid10k (10,004 spheres), 4096x4096, spp 64 (the cell-grid scan).
The line has the same shape for every config; the driver's headline run is the default.

Launch: python bench.py --gpus 1 ...   or
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
               --master-port P bench.py --gpus N --steps K --warmup W
Rank 0 prints ONE JSON line.

roofline: the trace kernel is bound by VALU ISSUE (the matrix cores share the SIMD's
issue port).  `achieved` = executed VALU lane-operations per launch (SQ_THREAD_CYCLES_VALU = SQ_INSTS_VALU x 64 x
lane utilisation, from the committed rocprofv3 --pmc passes of this same command, profiles/r04_pmc_summary*.json)
divided by the launch duration measured LIVE in this run with HIP events on the kernel's stream; `peak` = 78.6 T
lane-op/s (256 CU x 4 SIMD x 64 lanes / 2 cycles x 2.4 GHz).  The instruction counts are a property of (kernel
binary, workload): the PMC summary records the hash of the kernel sources it was taken from, and the roofline object
is withheld (frac = null, error set, warning on stderr) when that hash is not the one of the library benchmarked.
SURVEY.md §8(d)'s algorithmic unit (18 flop x N spheres per list scan) is reported separately as `cull_factor` =
algorithmic flops / executed lane-operations: how much of the exhaustive scan the filters avoid.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DEPTH = 50
SCENE_SEED, RENDER_SEED = 1, 1
CONFIGS = {
    # name: scene, W, H, spp per GPU, aperture (-1 = the scene's own), metric text, PMC summary
    "c2": dict(scene="cover", W=1200, H=800, spp=128, aperture=-1.0,
               metric="Msamples/sec (WxHxspp/s), 1200x800 cover scene spp=128 d=50", pmc="r04_pmc_summary.json"),
    "c3": dict(scene="cover", W=1200, H=800, spp=1024, aperture=-1.0,  # BASELINE config 3's whole job on ONE GPU: one 11.8 GB pass
               metric="Msamples/sec (WxHxspp/s), 1200x800 cover scene spp=1024 d=50", pmc="r04_pmc_summary_c3.json"),
    "c4": dict(scene="cover", W=1920, H=1080, spp=512, aperture=2.0,
               metric="Msamples/sec (WxHxspp/s), 1920x1080 cover scene aperture=2.0 spp=512 d=50", pmc="r04_pmc_summary_c4.json"),
    "c5": dict(scene="grid10k", W=4096, H=4096, spp=64, aperture=-1.0,
               metric="Msamples/sec (WxHxspp/s), 4096x4096 grid10k scene (10,004 spheres) spp=64 d=50", pmc="r04_pmc_summary_c5.json"),
}
PEAK_VALU_TLANEOPS = 78.6   # 256 CU x 4 SIMD x 64 lanes / 2 cycles per wave-instruction x 2.4 GHz (MI355X_MICROARCH.md: SIMD-32, 2 cycles)
PEAK_HBM_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E ~8 TB/s
FLOPS_PER_SPHERE_TEST = 18  # Sphere::Intersect up to the discriminant: 9 mul + 9 add/sub


def kernel_sources_hash():
    """Identity of the device code: sha256 over the files librt_hip.so is compiled from (the .so itself is not tracked)."""
    import __graft_entry__ as g
    csrc = os.path.join(ROOT, "cpuraytracer_amd", "csrc")
    files = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith((".h", ".hip", ".inc")) or f == "Makefile"]
    files.append(os.path.join(ROOT, "include", "rt_api.h"))
    return g._sources_hash(files)


def cpu_baseline(cfg, target_seconds=15.0):
    """The oracle (CPU restatement; kind 'port') timed on this box's host cores on a bounded sample of
    the same workload: the full frame at a reduced spp (cost is exactly linear in spp),
    BvhNode traversal as in the reference, std::thread workers."""
    from oracle import oracle_py as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # a one-GPU box's CPU share is 16 cores; RT_CPU_BASELINE_THREADS overrides
    cores = int(os.environ.get("RT_CPU_BASELINE_THREADS", min(avail, 16)))
    W, H = cfg["W"], cfg["H"]
    reduced = ""
    if cfg["scene"] == "grid10k" and W * H > 4000000:
        # C5 only: one full 4096x4096 frame is ~1 minute of oracle time, so the sample is the SAME camera and aspect (1:1) at
        # 1024x1024 -- a 4x coarser pixel grid over the same picture, not a crop; every other config renders its own frame size
        # and is bounded by spp alone
        W, H = 1024, 1024
        reduced = " (same camera at 1/4 linear resolution of %dx%d)" % (cfg["W"], cfg["H"])
    sc = O.build_scene(cfg["scene"], SCENE_SEED, cfg["W"] / float(cfg["H"]), cfg["aperture"])
    orc = O.Oracle()
    orc.upload(sc)
    t0 = time.perf_counter()
    st = orc.render(W, H, 1, 2, DEPTH, RENDER_SEED, accel=O.ACCEL_BVH, threads=cores)
    orc.resolve()
    t1 = time.perf_counter() - t0
    spp = int(max(1, min(cfg["spp"], round(target_seconds / max(t1, 1e-3)))))
    if spp > 1:
        t0 = time.perf_counter()
        st = orc.render(W, H, 1, 1 + spp, DEPTH, RENDER_SEED, accel=O.ACCEL_BVH, threads=cores)
        orc.resolve()
        t1 = time.perf_counter() - t0
    return {"value": st.samples / t1 / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "%s scene %dx%d%s depth %d at spp=%d (of %d), BvhNode traversal, %d std::thread workers, %.1f s"
                      % (cfg["scene"], W, H, reduced, DEPTH, spp, cfg["spp"], cores, t1)}


def roofline_object(cfg, kernel_name, avg_ms, avg_trav, n_spheres, src_hash, n_gpus=1):
    """Executed-work roofline of the trace kernel (module docstring).  The committed counters belong to ONE workload — this
    config on one GPU: at --gpus N > 1 rank 0 renders every N-th row at N x the samples per pixel, another instruction
    count, so the fraction is withheld there (frac = null + reason) instead of pricing a different launch with them."""
    rl = {"bound": "valu-issue", "kernel": kernel_name, "achieved": None, "peak": PEAK_VALU_TLANEOPS, "unit": "Tlane-op/s",
          "frac": None, "traffic": None, "launch_ms": avg_ms, "pmc_source": "profiles/" + cfg["pmc"],
          "kernel_sources_sha256": src_hash}
    if n_gpus != 1:
        rl["pmc_source"] = None
        rl["error"] = ("withheld at n_gpus=%d: the committed PMC counters (profiles/%s) are those of the 1-GPU launch; rank 0's strip "
                       "(every %d-th row at %dx the spp) executes a different instruction count" % (n_gpus, cfg["pmc"], n_gpus, n_gpus))
        return rl, None
    path = os.path.join(ROOT, "profiles", cfg["pmc"])
    try:
        d = json.load(open(path))
    except Exception as e:  # no profile for this config yet
        rl["error"] = "no PMC summary (%s): run tools/profile_round.sh" % e
        return rl, None
    if d.get("kernel_sources_sha256") != src_hash:
        rl["error"] = ("PMC summary %s was taken from kernel sources %s..., this library is %s...: re-run tools/profile_round.sh"
                       % (cfg["pmc"], str(d.get("kernel_sources_sha256"))[:12], src_hash[:12]))
        print("bench.py: WARNING: " + rl["error"], file=sys.stderr, flush=True)
        return rl, None
    wl = d.get("workload_key")  # summaries written since round 3 record what was profiled (n_gpus, W, H, spp)
    if wl is not None and (wl.get("n_gpus", 1), wl.get("W"), wl.get("H"), wl.get("spp")) != (1, cfg["W"], cfg["H"], cfg["spp"]):
        rl["error"] = "PMC summary %s was taken on workload %s, not on %dx%d spp %d on 1 GPU" % (cfg["pmc"], wl, cfg["W"], cfg["H"], cfg["spp"])
        print("bench.py: WARNING: " + rl["error"], file=sys.stderr, flush=True)
        return rl, None
    t = d["raw_counters"]["trace"]
    k = d["derived_trace_kernel"]
    launches = max(1, int(k.get("launches_in_pmc_pass", 1)))
    lane_ops = t["SQ_THREAD_CYCLES_VALU"] / launches   # active lanes summed over the VALU wave-instructions of one launch
    insts = t["SQ_INSTS_VALU"] / launches
    achieved = lane_ops / (avg_ms * 1e-3) / 1e12
    rl.update({
        "achieved": achieved, "frac": achieved / PEAK_VALU_TLANEOPS,
        "traffic": k.get("hbm_bytes_per_launch"),
        "valu_wave_insts_per_launch": insts, "lane_utilisation": lane_ops / (insts * 64.0),
        # issue-slot view: 2 cycles per wave-instruction on a SIMD-32, 1024 SIMDs, at the clock the PMC pass measured
        "issue_slot_frac": insts * 2.0 / (1024.0 * avg_ms * 1e-3 * k["effective_clock_GHz"] * 1e9),
        "clock_GHz_in_pmc_pass": k["effective_clock_GHz"], "pmc_launch_ms": k["launch_ms_rocprof_avg"],
        "note": "achieved = SQ_THREAD_CYCLES_VALU per launch (rocprofv3 --pmc, committed) / launch_ms (HIP events, this run); "
                "frac x (1/lane_utilisation) = fraction of VALU issue slots used at 2.4 GHz",
    })
    cull = FLOPS_PER_SPHERE_TEST * avg_trav * n_spheres / lane_ops
    # SURVEY.md 8(d)'s own unit beside the executed-work fraction: 18 flop x N x T per launch / time / peak.  The kernel does not
    # run exhaustive scans (it culls, as the reference's BVH does), so this exceeds 1 by about the cull factor x frac.
    rl["frac_algorithmic"] = FLOPS_PER_SPHERE_TEST * avg_trav * n_spheres / (avg_ms * 1e-3) / 1e12 / PEAK_VALU_TLANEOPS
    rl["frac_algorithmic_note"] = "18*N*T / t / peak (SURVEY.md 8d); > 1 => culled: the filters skip that share of the exhaustive scan"
    return rl, {"value": cull, "algorithmic": "18 flop x %d spheres x %.0f list scans per launch (SURVEY.md 8d) / executed VALU lane-operations"
                                              % (n_spheres, avg_trav)}


def hbm_read_equivalent(avg_ms, avg_trav, n_spheres, measured_traffic_bytes):
    """BASELINE.json north_star: "fraction of the HBM-read roofline".  SURVEY.md §8(d): an exhaustive list scan reads 16 bytes per
    sphere record, so the ALGORITHMIC read stream of a launch is 16 x N x T bytes; divided by the launch time it is the rate a
    kernel streaming the list from HBM would have to sustain.  The list lives in LDS (and the filters skip most of it), so this
    legitimately exceeds 1.0 x HBM peak; the MEASURED HBM rate of the same launch (PMC FETCH_SIZE + WRITE_SIZE) stands beside it."""
    alg = 16.0 * n_spheres * avg_trav
    sec = avg_ms * 1e-3
    return {"label": "algorithmic sphere-record bytes an exhaustive scan would read (16 B x N x T), NOT bytes moved",
            "bytes_per_launch": alg, "GBps": alg / sec / 1e9, "hbm_peak_GBps": PEAK_HBM_GBPS, "frac_of_hbm_peak": alg / sec / 1e9 / PEAK_HBM_GBPS,
            "measured_hbm_GBps": (measured_traffic_bytes / sec / 1e9) if measured_traffic_bytes else None,
            "measured_frac_of_hbm_peak": (measured_traffic_bytes / sec / 1e9 / PEAK_HBM_GBPS) if measured_traffic_bytes else None}


def per_rank_report(rows):
    """rows[r] = (render_ms, accumulate_ms, resolve_ms, gather_ms, step_wall_ms) averaged over the timed steps of rank r ->
    the `per_rank` list and the id of the rank whose device work per step was longest (where an N-GPU step's time went)."""
    per = [{"rank": r, "render_ms": v[0], "accumulate_ms": v[1], "resolve_ms": v[2], "gather_ms": v[3], "step_wall_ms": v[4]}
           for r, v in enumerate(rows)]
    slowest = max(range(len(rows)), key=lambda r: rows[r][0] + rows[r][1] + rows[r][2])
    return per, slowest


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    W_IMG, H_IMG, SPP_PER_GPU = cfg["W"], cfg["H"], cfg["spp"]

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world
    N = args.gpus
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    # RT_BENCH_REHEARSAL=1: run the N-rank code path on fewer devices than ranks (ranks share GPUs, gloo instead
    # of RCCL, strips gathered through host memory).  For checking the distributed plumbing on a 1-GPU box only;
    # its numbers mean nothing and the JSON line says so.
    rehearsal = os.environ.get("RT_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if N > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=N)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=N, device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as g
    if rank == 0:
        g.build()
    if dist is not None:
        dist.barrier()
    from cpuraytracer_amd import HipRenderer, scenes, distributed as D

    spp = SPP_PER_GPU * N
    r = HipRenderer(local_rank)
    stream = torch.cuda.Stream()  # a real (non-null) HIP stream: kernels, copies, events and the gather are ordered on it
    torch.cuda.set_stream(stream)
    r.set_stream(stream.cuda_stream)
    sc = scenes.build_scene(cfg["scene"], SCENE_SEED, W_IMG, H_IMG, aperture=cfg["aperture"])
    r.upload(sc)  # scene resident in HBM before the timed region
    rs = D.shard_rowset(H_IMG, rank, N)
    rows = D.local_rows(H_IMG, rank, N)
    xch = D.StripExchange(H_IMG, W_IMG, rank, N, "cuda")  # HDR + LDR strip of this rank in one buffer: one gather per step

    kernel_ms, trav, last = [], [], None
    phase = {"render": 0.0, "accumulate": 0.0, "resolve": 0.0, "gather": 0.0, "wall": 0.0}
    gather_events = []

    def step(timed=False):
        nonlocal last
        w0 = time.perf_counter()
        st = r.render(W_IMG, H_IMG, 1, 1 + spp, DEPTH, RENDER_SEED, rowset=rs)
        res_ms = r.resolve()
        r.copy_to_device(xch.hdr.data_ptr(), xch.ldr.data_ptr())
        if rehearsal:
            r.synchronize()
        g0 = time.perf_counter()
        if timed and N > 1 and not rehearsal:  # events on the stream the gather is enqueued on (torch's current stream here)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
        parts_h, parts_l = xch.gather(through_host=rehearsal)  # RCCL gather of the strips (no-op at N = 1)
        if timed and N > 1 and not rehearsal:
            e1.record(stream)
            gather_events.append((e0, e1))
        elif timed and N > 1:
            phase["gather"] += (time.perf_counter() - g0) * 1e3
        last = (st, parts_h, parts_l)
        if timed:
            phase["render"] += st.ms_render
            phase["accumulate"] += st.ms_accumulate
            phase["resolve"] += float(res_ms or 0.0)
            phase["wall"] += (time.perf_counter() - w0) * 1e3
        return st

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st = step(timed=True)
        kernel_ms.append(st.ms_render)
        trav.append(st.traversals)
    sync()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    for e0, e1 in gather_events:
        phase["gather"] += e0.elapsed_time(e1)
    k = float(max(1, args.steps))
    mine = [phase["render"] / k, phase["accumulate"] / k, phase["resolve"] / k, phase["gather"] / k, phase["wall"] / k]
    rank_rows = [mine]
    if dist is not None:
        rank_rows = [None] * N
        dist.all_gather_object(rank_rows, mine)

    # per-rank kernel facts for the roofline object (rank 0's own kernel)
    n_spheres = sc.n
    avg_ms = sum(kernel_ms) / max(1, len(kernel_ms))
    avg_trav = sum(trav) / max(1, len(trav))

    if rank == 0:
        total_samples = W_IMG * H_IMG * spp
        value = total_samples * args.steps / dt / 1e6
        st_last, parts_h, parts_l = last
        # assembling the gathered strips is outside the timed region (host side de-interleave for the PPM)
        full = D.assemble([p.cpu().numpy() for p in parts_l], H_IMG, N)
        assert full.shape == (H_IMG, W_IMG, 3)
        if os.environ.get("RT_BENCH_DUMP_LDR"):  # tests: the gathered image of the last step
            import numpy as np
            np.save(os.environ["RT_BENCH_DUMP_LDR"], full)
        kernel_name = ("rt_trace_kernel<cell-grid scan, 1024 threads, pooled slab items, hit stash>" if args.config == "c5"
                       else "rt_trace_kernel<LDS tables, 1024 threads, matrix-core filter, hit stash>")
        roofline, cull = roofline_object(cfg, kernel_name, avg_ms, avg_trav, n_spheres, kernel_sources_hash(), n_gpus=N)
        per_rank, slowest = per_rank_report(rank_rows)
        out = {
            "metric": cfg["metric"],
            "value": value, "unit": "Msamples/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic" + (" (REHEARSAL: ranks share GPUs over gloo, not a measurement)" if rehearsal else ""),
            "config": {"workload": "%s scene (%d spheres, scene seed %d) %dx%d spp=%d depth=%d%s, render seed %d; "
                                   "rows sharded cyclically (row j to rank j mod N) over %d GPU(s), RCCL gather of strips to rank 0"
                                   % (cfg["scene"], n_spheres, SCENE_SEED, W_IMG, H_IMG, spp, DEPTH,
                                      (" aperture=%.1f" % cfg["aperture"]) if cfg["aperture"] >= 0 else "", RENDER_SEED, N),
                       "name": args.config, "n_spheres": n_spheres, "spp": spp, "spp_per_gpu": SPP_PER_GPU, "samples_per_step": total_samples,
                       "traversals_per_sample": avg_trav / (W_IMG * rows * spp), "passes": st_last.passes},
            "roofline": roofline,
            "cull_factor": cull,
            # labelled extra (north_star): rank 0's launch; `traffic` is only known for the profiled 1-GPU workload
            "hbm_read_equivalent": hbm_read_equivalent(avg_ms, avg_trav, n_spheres, roofline.get("traffic")),
            "per_rank": per_rank, "slowest_rank": slowest,
        }
        if N == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    r.close()


if __name__ == "__main__":
    main()
