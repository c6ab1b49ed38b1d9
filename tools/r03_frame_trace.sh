#!/bin/bash
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03r
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/frame_trace.py 8 10 2>&1 | grep -v amdgpu.ids
rocprofv3 --kernel-trace --stats -d $out/prof -o frame --output-format csv -- python3 $R/tools/frame_trace.py 8 10 > $out/trace.log 2>&1
f=$(find $out/prof -name '*kernel_stats.csv' | head -1)
cp $f $out/frame_kernel_stats.csv
find $out/prof -name '*kernel_trace.csv' -exec cp {} $out/frame_kernel_trace.csv \;
rm -rf $out/prof
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$out/frame_kernel_stats.csv")))
for r in rows[:45]:
    print(r['Name'][:90], r['Calls'], round(int(r['TotalDurationNs'])/1e6,2), round(float(r['AverageNs'])/1e3,1))
PY
