#!/bin/bash
# BASELINE configs[4]: "10k random spheres 4096x4096 spp=64 -- LDS-tiled sphere list vs. HBM-streamed, rocprof HBM GB/s sweep".
# For every placement of the scan's data: Gsamples/s (trace + accumulate kernels, HIP events) and the trace kernel's FETCH_SIZE /
# WRITE_SIZE (rocprofv3 --pmc, separate passes).  Run on the GPU box; then python tools/c5_placement_sweep.py gpurun_out/<dir>
export TMPDIR=/tmp
d=gpurun_out/${1:-c5place}
mkdir -p $d
run() {  # name, W, H, spp, env...
  name=$1; W=$2; H=$3; spp=$4; shift 4
  env "$@" RT_BENCH_REPS=2 python3 tools/bench_scene.py grid10k $W $H $spp 50 > $d/$name.json 2> $d/$name.err || echo "$name failed"
  for c in FETCH_SIZE WRITE_SIZE; do
    # (the program after -- must be python itself: no env wrapper under rocprofv3; the knobs are exported around the call)
    ( export "$@" RT_BENCH_REPS=1; rocprofv3 --pmc $c --output-format csv -d $d/${name}_$c -- python3 tools/bench_scene.py grid10k $W $H $spp 50 > /dev/null 2> $d/${name}_$c.err ) || echo "$name $c failed"
  done
  echo "$name done"
}
run default       4096 4096 64 RT_PLACEMENT=default
run quant_lds     4096 4096 64 RT_GRID_QUANT=1
run shadow64      4096 4096 64 RT_SHADOW_CELLS=64
run hierarchy_lds 4096 4096 64 RT_GRID=0
run hierarchy_l2  4096 4096 64 RT_GRID=0 RT_TREE_LDS=0
run valu_streamed 1024 1024 2  RT_SCAN=valu
