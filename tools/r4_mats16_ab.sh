#!/bin/bash
export RT_BENCH_REPS=4
for round in 1 2 3; do
  for m in 0 1 2; do
    r=$(RT_MATS16=$m python tools/bench_scene.py cover 1200 800 128 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Ms/s %.3f ms  hdr %s' % (d['Msamples_per_s'], d['ms'], d['hdr_sha1']))")
    echo "round $round c2 mats16=$m: $r"
  done
done
export RT_BENCH_REPS=2
for round in 1 2; do
  for m in 0 1; do
    r=$(RT_MATS16=$m python tools/bench_scene.py grid10k 4096 4096 64 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Ms/s %.3f ms  hdr %s' % (d['Msamples_per_s'], d['ms'], d['hdr_sha1']))")
    echo "round $round c5 mats16=$m: $r"
  done
done
RT_VERBOSE=1 RT_BENCH_REPS=1 python tools/bench_scene.py cover 300 200 4 50 2>&1 | grep "rt_trace launch" | tail -1
