#!/bin/bash
# GPU box: exact-fp32 fused Bottleneck - parity, then the f32 leg with and without it
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r04h
mkdir -p $out
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "bottleneck or bneck" > $out/bneck_tests.log 2>&1; echo "bneck tests rc=$?"; tail -6 $out/bneck_tests.log
timeout -k 10 600 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -k "hourglass or hg or fp32 or f32" > $out/hg_tests.log 2>&1; echo "hg tests rc=$?"; tail -4 $out/hg_tests.log
for arm in "FUSG_NO_BNECK_F32=1" "base" "FUSG_NO_BNECK_F32=1" "base"; do
  if [ "$arm" = base ]; then e=""; else e="$arm"; fi
  env $e timeout -k 10 300 python bench.py --precision f32 --no-cpu-baseline --no-clip --steps 10 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('f32 $arm', d['value'], r['frac'], r['conv_ms_per_step'], r['launches_per_step'])"
done
