#!/bin/bash
# round 4, GPU call 2: regression test, trip counts of the marked regions (C2, C5), PMC profile of C2
export TMPDIR=/tmp
d=gpurun_out/r4b
mkdir -p $d
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "readers_render" > $d/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $d/pytest.log
timeout -k 10 300 python tools/phase_budget.py dynamic c2 > $d/dyn_c2.log 2>&1; echo "dyn c2 rc=$?"; tail -2 $d/dyn_c2.log
timeout -k 10 400 python tools/phase_budget.py dynamic c5 > $d/dyn_c5.log 2>&1; echo "dyn c5 rc=$?"; tail -2 $d/dyn_c5.log
timeout -k 10 600 bash tools/profile_round.sh r4b/prof_c2 c2 > $d/prof_c2.log 2>&1; echo "profile c2 rc=$?"; tail -3 $d/prof_c2.log
