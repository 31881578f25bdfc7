#!/bin/bash
# round 4: fuzz campaign on the final kernels + the full-size digests under the variant families (GPU box; prints progress)
d=gpurun_out/r4par
mkdir -p $d
{
python tools/fuzz_campaign.py 0 1500
FUZZ_LIGHTS=3 python tools/fuzz_campaign.py 20000 500
RT_GRID_QUANT=1 python tools/fuzz_campaign.py 30000 300
RT_GRID=0 python tools/fuzz_campaign.py 31000 300
RT_SG_SPH=1 RT_SHADOW_CELLS=128 python tools/fuzz_campaign.py 32000 300
} 2>&1 | tee $d/fuzz.txt | grep -v "^  \.\.\." 
for v in "RT_GRID_QUANT=1" "RT_SG_SPH=1" "RT_SHADOW_CELLS=64" "RT_GRID=0" "RT_STASH=0" "RT_MATS_L2=0" "RT_STASH_CAP=30" "RT_SHADOW_GRID=0"; do
  r=$(env $v python -m pytest tests/test_gpu_dense_differential.py -q -k "full_size_config" 2>&1 | tail -1)
  echo "== $v: $r" | tee -a $d/variants.txt
done
