#!/bin/bash
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r04e
mkdir -p $out
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > $out/ops_tests.log 2>&1; echo "ops tests rc=$?"; tail -6 $out/ops_tests.log
for big in 0 14 22 0 14 22; do
  FUSG_BF16_BIG=$big timeout -k 10 300 python bench.py --precision bf16 --no-cpu-baseline --no-clip --steps 20 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('bf16 256 big=$big', d['value'], r['frac'], r['conv_ms_per_step'], r['launches_per_step'])"
done
for big in 0 14 22; do
  FUSG_BF16_BIG=$big timeout -k 10 300 python bench.py --precision bf16 --res 512 --batch 16 --no-cpu-baseline --no-clip --steps 10 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('bf16 512 big=$big', d['value'], r['frac'], r['conv_ms_per_step'], r['launches_per_step'])"
done
FUSG_BF16_BIG=14 timeout -k 10 600 python -m pytest tests/test_gpu_nets.py tests/test_gpu_ops.py -x -q -m gpu -k "bf16" > $out/bf16_tests_big14.log 2>&1; echo "bf16 tests (big14) rc=$?"; tail -4 $out/bf16_tests_big14.log
FUSG_BF16_BIG=22 timeout -k 10 600 python -m pytest tests/test_gpu_nets.py tests/test_gpu_ops.py -x -q -m gpu -k "bf16" > $out/bf16_tests_big22.log 2>&1; echo "bf16 tests (big22) rc=$?"; tail -4 $out/bf16_tests_big22.log
timeout -k 10 300 python bench.py --precision f32 --no-cpu-baseline --no-clip --steps 10 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('f32', d['value'], r['frac'], r['conv_ms_per_step'], r['launches_per_step'])"
