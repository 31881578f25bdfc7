#!/usr/bin/env python3
"""Replay REAL accelerator queries of a BASELINE config through the plain list.

The full-size digests of C5 are made with the oracle's PaddedListTree (ACCEL_PADDED_LIST); its claim is "the list scan's record for
every ray".  The unit test checks that on synthetic random and grazing rays; this tool checks it on the rays the job actually casts:
it traces random (i, j, s) samples of the config with every accelerator query recorded (orc_unit_trace_path: closest-hit rays of
every segment and the sun-occlusion rays of every hit), then evaluates each recorded ray with ACCEL_LIST (Sphere::Intersect for every
sphere, smallest t, lower index on ties) and with ACCEL_PADDED_LIST and compares the two hit records bit for bit, and the recorded
outcome (t, or occluded / visible) with the list's.

usage: python tools/harvest_accel_queries.py [c4|c5] [n_queries] [threads]      (CPU only; ~2 min per 1e7 queries of c5 on 6 threads)
"""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # noqa: E402

CONFIGS = {"c4": ("cover", 1920, 1080, 512, 2.0), "c5": ("grid10k", 4096, 4096, 64, -1.0), "c2": ("cover", 1200, 800, 128, -1.0)}


def harvest(orc, W, H, spp, n_queries, threads, seed=1, rng_seed=2026):
    """Recorded queries [n, 8] of random samples of the job (render seed `seed`), at least n_queries of them."""
    rng = np.random.default_rng(rng_seed)
    out, have = [], 0

    def one(ijs):
        return [orc.trace_path(W, H, int(i), int(j), int(s), 50, seed, accel=O.ACCEL_PADDED_LIST, cap=256)[0] for i, j, s in ijs]
    with ThreadPoolExecutor(threads) as ex:
        while have < n_queries:
            m = max(1024, min(400000, (n_queries - have) // 3))
            ijs = np.stack([rng.integers(0, W, m), rng.integers(0, H, m), rng.integers(1, spp + 1, m)], 1)
            for part in ex.map(one, np.array_split(ijs, threads * 8)):
                for q in part:
                    out.append(q)
                    have += len(q)
    return np.concatenate(out)[:max(n_queries, 1)]


def replay(orc, queries, threads):
    rays = np.ascontiguousarray(queries[:, :6])
    chunks = np.array_split(np.arange(len(rays)), max(1, threads * 16))
    with ThreadPoolExecutor(threads) as ex:
        hl = np.concatenate(list(ex.map(lambda ix: orc.closest_hit(rays[ix], O.ACCEL_LIST), chunks)))
        hp = np.concatenate(list(ex.map(lambda ix: orc.closest_hit(rays[ix], O.ACCEL_PADDED_LIST), chunks)))
    same = np.all(hl.view(np.uint32) == hp.view(np.uint32), axis=1)
    hit = hl[:, 1].view(np.int32) >= 0
    kind, res = queries[:, 6], queries[:, 7]
    closest = kind == 0.0
    # what the path itself saw (through the padded tree) against the list: t bit for bit (-1: miss), occlusion flag
    t_list = np.where(hit, hl[:, 0], np.float32(-1.0)).astype(np.float32)
    ok_closest = t_list[closest].view(np.uint32) == res[closest].astype(np.float32).view(np.uint32)
    ok_occl = (res[~closest] == 1.0) == hit[~closest]
    return {"queries": int(len(rays)), "closest_hit_queries": int(closest.sum()), "occlusion_queries": int((~closest).sum()),
            "list_hits": int(hit.sum()), "records_differing_list_vs_padded_list": int((~same).sum()),
            "recorded_closest_t_differing_from_list": int((~ok_closest).sum()), "recorded_occlusion_differing_from_list": int((~ok_occl).sum())}


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c5"
    n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10000000
    threads = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    scene, W, H, spp, ap = CONFIGS[cfg]
    sc = O.build_scene(scene, 1, W / H, ap)
    orc = O.Oracle()
    orc.upload(sc)
    t0 = time.time()
    q = harvest(orc, W, H, spp, n, threads)
    t1 = time.time()
    rep = replay(orc, q, threads)
    rep.update({"config": cfg, "scene": scene, "n_spheres": sc.n, "W": W, "H": H, "spp": spp, "harvest_seconds": round(t1 - t0, 1),
                "replay_seconds": round(time.time() - t1, 1), "threads": threads,
                "method": "random (i, j, s) samples of the job traced with orc_unit_trace_path (ACCEL_PADDED_LIST), every recorded ray "
                          "re-evaluated with ACCEL_LIST and ACCEL_PADDED_LIST"})
    print(json.dumps(rep))
    bad = rep["records_differing_list_vs_padded_list"] + rep["recorded_closest_t_differing_from_list"] + rep["recorded_occlusion_differing_from_list"]
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
