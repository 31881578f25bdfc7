#!/bin/bash
# Full profile of bench.py for the round: bench line, rocprofv3 kernel stats, four PMC passes (run on the GPU box).
export TMPDIR=/tmp
d=gpurun_out/${1:-profR}
mkdir -p $d
python bench.py --steps 5 --warmup 1 > $d/bench.json 2> $d/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $d/kt -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $d/bench_kt.json 2> $d/bench_kt.err
f=$(find $d/kt -name "*kernel_stats.csv" | head -1); head -4 "$f"
for p in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $p | cut -d" " -f1)
  rocprofv3 --pmc $p --output-format csv -d $d/pmc_$n -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $d/pmc_$n.json 2> $d/pmc_$n.err; echo "$n rc=$?"
done
cut -c1-400 $d/bench.json
