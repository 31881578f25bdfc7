#!/bin/bash
# round 4, final kernels: fuzz campaign (random scenes vs the live oracle) + the full-size digests under every variant family
d=gpurun_out/r4fuzz
mkdir -p $d
{
python tools/fuzz_campaign.py 40000 700
FUZZ_LIGHTS=3 python tools/fuzz_campaign.py 41000 250
RT_GRID=0 python tools/fuzz_campaign.py 42000 200
RT_GRID_QUANT=1 python tools/fuzz_campaign.py 43000 150
} 2>&1 | tee $d/fuzz.txt | grep -v "^  \.\.\."
bash tools/r4_variants.sh
