#!/bin/bash
# round 4, last kernels (grid scan without the per-item re-check): variants' full-size digests + a third fuzz campaign
d=gpurun_out/r4fuzz3
mkdir -p $d
bash tools/r4_variants.sh
{
python tools/fuzz_campaign.py 60000 600
FUZZ_LIGHTS=2 python tools/fuzz_campaign.py 61000 200
RT_GRID=2 python tools/fuzz_campaign.py 62000 200
} 2>&1 | tee $d/fuzz.txt | grep -v "^  \.\.\."
