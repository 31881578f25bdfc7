#!/bin/bash
# Diagnostic PMC passes over one bench.py launch set (GPU box): instruction cache, outstanding-instruction levels, per-type issue
# cycles, rate-limited VALU classes, FIFO stalls.  usage: tools/pmc_diag.sh <outdir under gpurun_out> [c2|c5]
export TMPDIR=/tmp
d=gpurun_out/${1:-pmcdiag}
cfg=${2:-c2}
mkdir -p $d
i=0
for p in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
         "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_LDS_DATA_FIFO_FULL" \
         "SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $p --output-format csv -d $d/p$i -- python3 bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline > $d/p$i.json 2> $d/p$i.err || { echo "pass $i failed"; tail -3 $d/p$i.err; }
done
python3 - "$d" <<'PY'
import csv, glob, sys, collections
tot = collections.Counter()
for f in glob.glob(sys.argv[1] + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "rt_trace_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot):
    print("%-34s %.6g" % (k, tot[k]))
PY
