#!/bin/bash
# round 4, GPU call 3: trip counts of the marked regions (C2, C5) + PMC profiles of C2 and C5 at the same sources
export TMPDIR=/tmp
d=gpurun_out/r4c
mkdir -p $d
timeout -k 10 300 python tools/phase_budget.py dynamic c2 > $d/dyn_c2.log 2>&1; echo "dyn c2 rc=$?"; tail -2 $d/dyn_c2.log
timeout -k 10 400 python tools/phase_budget.py dynamic c5 > $d/dyn_c5.log 2>&1; echo "dyn c5 rc=$?"; tail -2 $d/dyn_c5.log
timeout -k 10 600 bash tools/profile_round.sh r4c/prof_c2 c2 > $d/prof_c2.log 2>&1; echo "profile c2 rc=$?"; tail -3 $d/prof_c2.log | cut -c1-200
timeout -k 10 900 bash tools/profile_round.sh r4c/prof_c5 c5 > $d/prof_c5.log 2>&1; echo "profile c5 rc=$?"; tail -3 $d/prof_c5.log | cut -c1-200
