#!/bin/bash
# round 4: everything profiles/r04_* is made from, in one GPU call (progress lines keep the call alive)
export TMPDIR=/tmp
d=r4final
mkdir -p gpurun_out/$d
for c in c2 c5 c4 c3; do
  timeout -k 10 400 bash tools/profile_round.sh $d/prof_$c $c > gpurun_out/$d/prof_$c.log 2>&1; echo "profile $c rc=$?"
done
timeout -k 10 300 python tools/phase_budget.py dynamic c2 > gpurun_out/$d/dyn_c2.log 2>&1; echo "dyn c2 rc=$?"
timeout -k 10 300 python tools/phase_budget.py dynamic c5 > gpurun_out/$d/dyn_c5.log 2>&1; echo "dyn c5 rc=$?"
timeout -k 10 500 bash tools/c5_placement_sweep.sh $d/c5place; echo "placement rc=$?"
