#!/bin/bash
# Build A/B variants of librt_hip.so HERE (hipcc cross-compiles; the .so files travel with the gpurun snapshot):
#   tools/ab_build.sh name1:"-DFLAG1 -DFLAG2" name2:"-DFLAG3" ...     -> cpuraytracer_amd/lib/exp/librt_hip_<name>.so
# then on the GPU box: tools/ab_run.sh name1 name2 ...  (interleaved timing of the same scene through tools/bench_scene.py)
set -e
cd "$(dirname "$0")/../cpuraytracer_amd/csrc"
mkdir -p ../lib/exp
FLAGS=$(make -pn | sed -n 's/^HIPFLAGS = //p' | head -1 | sed 's/\$(ARCH)/gfx950/')
for spec in "$@"; do
  name=${spec%%:*}; extra=${spec#*:}; [ "$extra" = "$spec" ] && extra=""
  echo "== $name: $extra"
  /opt/rocm/bin/hipcc $FLAGS $extra -shared -o ../lib/exp/librt_hip_$name.so rt_capi.hip &
done
wait
ls -la ../lib/exp/
