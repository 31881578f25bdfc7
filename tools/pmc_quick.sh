#!/bin/bash
# usage: tools/pmc_quick.sh <tag>  — SQ instruction mix of one bench step (run on the GPU box)
export TMPDIR=/tmp
tag=$1; mkdir -p gpurun_out/prof
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM"; do
n=$(echo $grp | cut -d" " -f1)
rocprofv3 --pmc $grp --output-format csv -d gpurun_out/prof/pmcq_${tag}_$n -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof/pmcq_${tag}_$n.json 2> gpurun_out/prof/pmcq_${tag}_$n.err
done
python3 - <<PY
import csv, glob
for d in sorted(glob.glob('gpurun_out/prof/pmcq_${tag}_*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(d)):
        if 'rt_trace_kernel' in r['Kernel_Name']:
            print(r['Counter_Name'], r['Counter_Value'], 'vgpr', r['VGPR_Count'], 'agpr', r['Accum_VGPR_Count'], 'lds', r['LDS_Block_Size'], 'grid', r['Grid_Size'], 'wg', r['Workgroup_Size'])
PY
grep -h -o '"value": [0-9.]*' gpurun_out/prof/pmcq_${tag}_*.json
