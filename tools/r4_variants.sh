#!/bin/bash
# round 4: the committed full-size digests under every variant family of the final kernels (GPU box; one line per family)
d=gpurun_out/r4par
mkdir -p $d
: > $d/variants.txt
for v in "RT_MATS16=0" "RT_MATS16=1" "RT_MATS16=2" "RT_GRID_QUANT=1" "RT_SG_SPH=1" "RT_SHADOW_CELLS=64" "RT_GRID=0" "RT_STASH=0" "RT_MATS_L2=0" "RT_STASH_CAP=30" "RT_SHADOW_GRID=0"; do
  r=$(env $v timeout -k 10 200 python -m pytest tests/test_gpu_dense_differential.py -q -k "full_size_config" 2>&1 | tail -1)
  echo "== $v: $r" | tee -a $d/variants.txt
done
