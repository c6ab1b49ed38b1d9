#!/bin/bash
# GPU box: kernel trace of the per-frame driver (tools/frame_trace.py), per-kernel totals + one frame's timeline
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03r
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/prof -o frame --output-format csv -- python3 $R/tools/frame_trace.py 8 12 $1 > $out/trace.log 2>&1
grep "ms per" $out/trace.log
f=$(find $out/prof -name '*kernel_stats.csv' | head -1)
cp $f $out/frame_kernel_stats.csv
find $out/prof -name '*kernel_trace.csv' -exec cp {} $out/frame_kernel_trace.csv \;
rm -rf $out/prof
