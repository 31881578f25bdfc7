#!/bin/bash
# Full profile of bench.py for one config (run on the GPU box): bench line, rocprofv3 kernel stats, four PMC passes.
# usage: tools/profile_round.sh <outdir under gpurun_out> [c2|c3|c4|c5]   then: python tools/summarize_profile.py gpurun_out/<outdir> profiles/r03 [c3|c4|c5]
export TMPDIR=/tmp
d=gpurun_out/${1:-profR}
cfg=${2:-c2}
mkdir -p $d
python bench.py --config $cfg --steps 5 --warmup 1 > $d/bench.json 2> $d/bench.err || { echo "bench failed"; tail -5 $d/bench.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $d/kt -- python3 bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline > $d/bench_kt.json 2> $d/bench_kt.err || { echo "kernel trace failed"; tail -5 $d/bench_kt.err; exit 1; }
f=$(find $d/kt -name "*kernel_stats.csv" | head -1); head -4 "$f"
for p in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $p | cut -d" " -f1)
  rocprofv3 --pmc $p --output-format csv -d $d/pmc_$n -- python3 bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline > $d/pmc_$n.json 2> $d/pmc_$n.err || { echo "pmc pass $n failed"; tail -5 $d/pmc_$n.err; exit 1; }
  echo "$n ok"
done
cut -c1-300 $d/bench.json
