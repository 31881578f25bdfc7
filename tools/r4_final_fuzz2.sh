#!/bin/bash
# round 4, final kernels: second fuzz campaign (other seeds, other variant families)
d=gpurun_out/r4fuzz2
mkdir -p $d
{
python tools/fuzz_campaign.py 50000 900
FUZZ_LIGHTS=3 python tools/fuzz_campaign.py 51000 300
RT_STASH=0 python tools/fuzz_campaign.py 52000 200
RT_MATS_L2=0 RT_SHADOW_GRID=0 python tools/fuzz_campaign.py 53000 150
} 2>&1 | tee $d/fuzz.txt | grep -v "^  \.\.\."
