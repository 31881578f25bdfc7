export TMPDIR=/tmp
mkdir -p gpurun_out/prof
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof/pmc_lds -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof/pmc_lds.json 2> gpurun_out/prof/pmc_lds.err
python3 - <<PY
import csv, glob
for d in sorted(glob.glob('gpurun_out/prof/pmc_lds/*/*counter_collection.csv')):
    for r in csv.DictReader(open(d)):
        if 'rt_trace_kernel' in r['Kernel_Name']:
            print(r['Counter_Name'], r['Counter_Value'])
PY
tail -3 gpurun_out/prof/pmc_lds.err
