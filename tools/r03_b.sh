#!/bin/bash
# GPU box: new frame tests first (fast feedback), then the full suite, the default bench and a per-dispatch kernel trace
set -e
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03k
mkdir -p $out
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_frame.py -x -q > $out/pytest_frame.log 2>&1 || { tail -60 $out/pytest_frame.log; echo FRAME_TESTS_FAILED; }
tail -3 $out/pytest_frame.log
timeout -k 10 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_frame.py > $out/pytest.log 2>&1 || { tail -60 $out/pytest.log; echo SUITE_FAILED; }
tail -3 $out/pytest.log
cp gpurun_out/parity_observed.json $out/ 2>/dev/null || true
( time timeout -k 10 500 python bench.py ) > $out/bench.log 2>&1 || { tail -30 $out/bench.log; exit 1; }
grep '^{"metric"' $out/bench.log > $out/bench_cfg1.json
tail -4 $out/bench.log | cut -c1-900
cd /tmp && export TMPDIR=/tmp
FUSG_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace -d $out/trace --output-format csv -- python3 $R/bench.py --steps 2 --warmup 2 --settle-s 0 --precision f16x3 --no-cpu-baseline --no-clip --no-prof > $out/trace.log 2>&1
cp $(ls $out/trace/*/*_kernel_trace.csv | head -1) $out/kernel_trace.csv
rm -rf $out/trace
ls -la $out
