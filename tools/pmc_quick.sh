#!/bin/bash
# usage: tools/pmc_quick.sh <tag>  — SQ instruction mix of one bench step (run on the GPU box)
export TMPDIR=/tmp
tag=$1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d gpurun_out/prof/pmcq_$tag -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof/pmcq_$tag.json 2> gpurun_out/prof/pmcq_$tag.err
python3 - <<PY
import csv, glob
for d in glob.glob('gpurun_out/prof/pmcq_$tag/*/*counter_collection.csv'):
    for r in csv.DictReader(open(d)):
        if 'rt_trace_kernel' in r['Kernel_Name']:
            print(r['Counter_Name'], r['Counter_Value'], 'vgpr', r['VGPR_Count'], 'lds', r['LDS_Block_Size'], 'grid', r['Grid_Size'], 'wg', r['Workgroup_Size'])
PY
