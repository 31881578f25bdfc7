#!/bin/bash
# C5 A/B: shadow-index resolution (runtime) x bounds per step (build)
export RT_BENCH_REPS=3
for round in 1 2; do
for lib in cpuraytracer_amd/lib/librt_hip.so cpuraytracer_amd/lib/exp/librt_hip_step8.so; do
  for cells in 64 128 256; do
    r=$(RT_HIP_LIB=$lib RT_SHADOW_CELLS=$cells python tools/bench_scene.py grid10k 4096 4096 64 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Ms/s %.3f ms  hdr %s' % (d['Msamples_per_s'], d['ms'], d['hdr_sha1']))")
    echo "round $round $(basename $lib) cells=$cells: $r"
  done
done
done
