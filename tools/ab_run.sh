#!/bin/bash
# On the GPU box: interleaved A/B timing of library variants built by tools/ab_build.sh.
# usage: tools/ab_run.sh [-s "scene W H spp depth [aperture]"] name1 name2 ...   ("base" = the shipped librt_hip.so)
scene="cover 1200 800 128 50"
if [ "$1" = "-s" ]; then scene="$2"; shift 2; fi
export RT_BENCH_REPS=${RT_BENCH_REPS:-5}
for round in 1 2 3; do
  for n in "$@"; do
    if [ "$n" = "base" ]; then lib=cpuraytracer_amd/lib/librt_hip.so; else lib=cpuraytracer_amd/lib/exp/librt_hip_$n.so; fi
    r=$(RT_HIP_LIB=$lib python tools/bench_scene.py $scene 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Ms/s %.3f ms  hdr %s' % (d['Msamples_per_s'], d['ms'], d['hdr_sha1']))")
    echo "round $round $n: $r"
  done
done
