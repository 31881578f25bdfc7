#!/bin/bash
# C5: run-time layout knobs of the grid scan (cells per sphere, shadow-index resolution), interleaved
export RT_BENCH_REPS=3
run() { r=$(env "$@" python tools/bench_scene.py grid10k 4096 4096 64 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Ms/s %.3f ms  hdr %s' % (d['Msamples_per_s'], d['ms'], d['hdr_sha1']))"); echo "$* : $r"; }
for round in 1 2; do
  for d in 0.5 0.75 1 1.5 2; do run RT_GRID_DENSITY=$d; done
  for c in 192 256 384 512; do run RT_SHADOW_CELLS=$c; done
done
