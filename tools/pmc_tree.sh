export TMPDIR=/tmp
mkdir -p gpurun_out/prof
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT TCC_HIT_sum"; do
n=$(echo $grp | cut -d" " -f1)
RT_BENCH_REPS=1 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/prof/tree_$n -- python3 tools/bench_scene.py grid10k 2048 2048 32 50 > /dev/null 2> gpurun_out/prof/tree_$n.err
done
python3 - <<PY
import csv, glob
acc={}
for d in sorted(glob.glob('gpurun_out/prof/tree_*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(d)):
        if 'rt_trace_kernel' in r['Kernel_Name']:
            acc.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
for k,v in acc.items(): print(k, max(v), len(v))
PY
