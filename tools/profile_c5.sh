#!/bin/bash
# rocprofv3 kernel stats of the C5 configuration (grid10k, 10,004 spheres, 4096x4096, spp 64, depth 50): the hierarchy scan.
export TMPDIR=/tmp
d=gpurun_out/${1:-profC5}
mkdir -p $d
RT_BENCH_REPS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $d/kt -- python3 tools/bench_scene.py grid10k 4096 4096 64 50 > $d/c5.json 2> $d/c5.err
f=$(find $d/kt -name "*kernel_stats.csv" | head -1); head -5 "$f"; tail -2 $d/c5.err
