#!/bin/bash
# GPU box, round 3 first call: full GPU suite, default bench, per-dispatch kernel trace of the serialised pass
set -e
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03a
mkdir -p $out
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
cp gpurun_out/parity_observed.json $out/ 2>/dev/null || true
( time timeout -k 10 500 python bench.py ) > $out/bench.log 2>&1
grep '^{"metric"' $out/bench.log > $out/bench_cfg1.json
tail -4 $out/bench.log | cut -c1-600
cd /tmp && export TMPDIR=/tmp
FUSG_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace -d $out/trace --output-format csv -- python3 $R/bench.py --steps 2 --warmup 2 --settle-s 0 --precision f16x3 --no-cpu-baseline --no-clip --no-prof > $out/trace.log 2>&1
cp $(ls $out/trace/*/*_kernel_trace.csv | head -1) $out/kernel_trace.csv
rm -rf $out/trace
ls -la $out
