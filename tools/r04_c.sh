#!/bin/bash
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r04c
mkdir -p $out
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "small_image" > $out/ops_tests.log 2>&1; echo "ops tests rc=$?"; tail -5 $out/ops_tests.log
timeout -k 10 300 python tools/small_exp.py 32 > $out/small_exp_b32.txt 2>&1; echo "small_exp rc=$?"; cat $out/small_exp_b32.txt
timeout -k 10 200 python tools/small_exp.py 1 > $out/small_exp_b1.txt 2>&1; echo "small_exp b1 rc=$?"; tail -3 $out/small_exp_b1.txt
for arm in old new old new; do
  if [ $arm = old ]; then export FUSG_NO_SMALL=1; else unset FUSG_NO_SMALL; fi
  timeout -k 10 200 python bench.py --precision f16x3 --no-cpu-baseline --no-clip --steps 20 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$arm', d['value'], r['frac'], r['conv_ms_per_step'], r['launches_per_step'])"
done
unset FUSG_NO_SMALL
for arm in old new; do
  if [ $arm = old ]; then export FUSG_NO_SMALL=1; else unset FUSG_NO_SMALL; fi
  timeout -k 10 200 python tools/small_batch.py 2>/dev/null | tail -8 | sed "s/^/$arm /"
done
unset FUSG_NO_SMALL
timeout -k 10 300 python tools/graph_capture_probe.py measure 1 > $out/graph_measure_b1.log 2>&1; echo "measure rc=$?"; tail -c 2500 $out/graph_measure_b1.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --hip-trace --memory-copy-trace --kernel-trace --stats -d $out/prof -o onepass -- python3 $R/tools/one_pass.py 32 10 > $out/onepass.log 2>&1; echo "rocprof rc=$?"
ls $out/prof/*/ 2>/dev/null | head; for f in $out/prof/*/*hip_api_stats.csv $out/prof/*/*memory_copy_stats.csv; do echo "== $f"; head -25 $f | cut -c1-160; done
