#!/bin/bash
# round 4, the very last kernels (grid bound test without the behind half, 10 / 704 feed): a fourth, short fuzz campaign
d=gpurun_out/r4fuzz4
mkdir -p $d
{
python tools/fuzz_campaign.py 70000 400
RT_GRID=2 python tools/fuzz_campaign.py 71000 300
FUZZ_LIGHTS=3 RT_GRID=2 python tools/fuzz_campaign.py 72000 100
} 2>&1 | tee $d/fuzz.txt | grep -v "^  \.\.\."
