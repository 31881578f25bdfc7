#!/bin/bash
# rocprofv3 kernel stats of the C5 configuration (grid10k, 10,004 spheres, 4096x4096, spp 64, depth 50): the hierarchy scan.
export TMPDIR=/tmp
d=gpurun_out/${1:-profC5}
mkdir -p $d
RT_BENCH_REPS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $d/kt -- python3 tools/bench_scene.py grid10k 4096 4096 64 50 > $d/c5.json 2> $d/c5.err
f=$(find $d/kt -name "*kernel_stats.csv" | head -1); head -5 "$f"; tail -2 $d/c5.err
# HBM traffic of the hierarchy scan (separate PMC passes; FETCH_SIZE / WRITE_SIZE in KiB, FETCH doubled for gfx950 as the guide prescribes)
for c in FETCH_SIZE WRITE_SIZE; do
  RT_BENCH_REPS=1 rocprofv3 --pmc $c --output-format csv -d $d/pmc_$c -- python3 tools/bench_scene.py grid10k 4096 4096 64 50 > /dev/null 2> $d/pmc_$c.err
  grep -h rt_trace_kernel $d/pmc_$c/*/*counter_collection.csv | awk -F, '{print $(NF-3), $(NF-2)}' | tr -d '"'
done
