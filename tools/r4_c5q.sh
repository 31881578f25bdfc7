#!/bin/bash
# C5: quantised LDS bounds on/off (runtime knob), with result hashes
export RT_BENCH_REPS=3
for round in 1 2; do
  for q in 0 1; do
    r=$(RT_GRID_QUANT=$q python tools/bench_scene.py grid10k 4096 4096 64 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Ms/s %.3f ms  hdr %s' % (d['Msamples_per_s'], d['ms'], d['hdr_sha1']))")
    echo "round $round quant=$q: $r"
  done
done
RT_VERBOSE=1 RT_BENCH_REPS=1 python tools/bench_scene.py grid10k 512 512 4 50 2>&1 | grep "rt_trace launch" | tail -1
