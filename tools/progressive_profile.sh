#!/bin/bash
# On the GPU box: rocprofv3 kernel statistics of tools/progressive_frames.py (1-spp frames with frame pipelining)
export TMPDIR=/tmp
d=gpurun_out/${1:-prog}
mkdir -p $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d/kt -- python3 tools/progressive_frames.py > $d/out.txt 2> $d/err.txt || { tail -5 $d/err.txt; exit 1; }
f=$(find $d/kt -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-160
cat $d/out.txt
