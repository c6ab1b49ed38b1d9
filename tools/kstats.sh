#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats of the serialised crop pass (f16x3 leg only) -> gpurun_out/$1_kernel_stats.csv
set -e
tag=${1:-kstats}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
FUSG_STREAMS=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/stats --output-format csv -- python3 $R/bench.py --steps 10 --warmup 5 --precision f16x3 --no-cpu-baseline --no-clip > $out/stats.log 2>&1
grep '^{"metric"' $out/stats.log > $out/${tag}_bench_serial_under_rocprof.json
cp $(ls $out/stats/*/*_kernel_stats.csv | head -1) $out/${tag}_kernel_stats.csv
rm -rf $out/stats
