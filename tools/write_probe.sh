#!/bin/bash
# WRITE_SIZE (KiB) and duration of the trace kernel for library variants: tools/write_probe.sh base nt ...  (GPU box)
export TMPDIR=/tmp
for n in "$@"; do
  if [ "$n" = "base" ]; then lib=cpuraytracer_amd/lib/librt_hip.so; else lib=cpuraytracer_amd/lib/exp/librt_hip_$n.so; fi
  d=gpurun_out/wprobe_$n; rm -rf $d; mkdir -p $d
  RT_HIP_LIB=$lib rocprofv3 --pmc WRITE_SIZE --output-format csv -d $d -- python3 tools/bench_scene.py cover 1200 800 128 50 > $d/out.json 2> $d/err.txt
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$n" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rt_trace_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE"]
vals = sorted(float(r["Counter_Value"]) for r in rows)
print(sys.argv[2], "trace launches", len(vals), "WRITE_SIZE KiB of the largest launches", vals[-3:], "=> GB", [round(v * 1024 / 1e9, 3) for v in vals[-3:]])
PY
  cut -c1-160 $d/out.json
done
