#!/bin/bash
# GPU box, round 4: the small-image kernel - parity, per-layer A/B, pass-level A/B; the hipGraph measurement again
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r04b
mkdir -p $out
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "small_image or conv_vs_torch or two_sources" > $out/ops_tests.log 2>&1; echo "ops tests rc=$?"; tail -15 $out/ops_tests.log
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
for mh in 8 16 32; do
  FUSG_BNECK_MINHW=$mh timeout -k 10 200 python bench.py --precision f16x3 --no-cpu-baseline --no-clip --steps 20 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('bneck_minhw $mh', d['value'], r['frac'], r['conv_ms_per_step'], r['launches_per_step'])"
done
timeout -k 10 300 python tools/graph_capture_probe.py measure 1 > $out/graph_measure_b1.log 2>&1; echo "measure rc=$?"; tail -c 3000 $out/graph_measure_b1.log
timeout -k 10 200 python tools/copy_sites.py 32 > $out/copy_sites.txt 2>&1; echo "copy_sites rc=$?"; head -40 $out/copy_sites.txt
timeout -k 10 600 python -m pytest tests/test_bench_gpu.py -x -q -m gpu > $out/bench_gpu_tests.log 2>&1; echo "bench_gpu tests rc=$?"; tail -5 $out/bench_gpu_tests.log; grep -h "^OBS\|diff\|MISMATCH" $out/bench_gpu_tests.log | head
cp gpurun_out/parity_observed.json $out/parity_observed_bench_gpu.json 2>/dev/null
