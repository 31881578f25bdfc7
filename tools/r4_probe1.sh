#!/bin/bash
# round 4, GPU call 1: new regression test, baseline numbers on this box, PC-sampling feasibility probe
export TMPDIR=/tmp
d=gpurun_out/r4a
mkdir -p $d
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "readers_render or batched_progressive or launch_and_layout_knobs" > $d/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $d/pytest.log
timeout -k 10 200 python bench.py --config c2 --steps 10 --warmup 2 --no-cpu-baseline > $d/bench_c2.json 2> $d/bench_c2.err; echo "bench c2 rc=$?"; cut -c1-250 $d/bench_c2.json
timeout -k 10 200 python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline > $d/bench_c5.json 2> $d/bench_c5.err; echo "bench c5 rc=$?"; cut -c1-250 $d/bench_c5.json
rocprofv3-avail list --pc-sampling > $d/pcs_avail.txt 2>&1; echo "avail rc=$?"; head -30 $d/pcs_avail.txt
rocprofv3-avail info --pc-sampling > $d/pcs_info.txt 2>&1; head -40 $d/pcs_info.txt
