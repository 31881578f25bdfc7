#!/bin/bash
# Vector-memory path counters of one bench.py launch (GPU box): is the kernel bound by the L1's request rate?
export TMPDIR=/tmp
d=gpurun_out/${1:-pmcmem}
cfg=${2:-c5}
mkdir -p $d
i=0
for p in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_READ_sum TCP_TOTAL_ACCESSES_sum" \
         "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" \
         "TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SMEM"; do
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
    print("%-40s %.6g" % (k, tot[k]))
PY
