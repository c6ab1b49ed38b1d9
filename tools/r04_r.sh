#!/bin/bash
# GPU box: fused Bottleneck with the L2 touch on a loader wave of its own (small grids) vs FUSG_NO_LOADER_WAVE=1 (round 3's form)
R=$GRAFT_REPO_ROOT
P=$R/future_urban_scene_generation_amd
cd $R
timeout -k 10 90 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "hg_bottleneck_fused and f16x3 and 4-4" 2>&1 | tail -3 || exit 1
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "bottleneck or bneck" 2>&1 | tail -3 || exit 1
for sw in "FUSG_NO_LOADER_WAVE=1" "X=1"; do
  echo "== $sw"
  env $sw FUSG_LIB=$P/libfusg_stamps.so timeout -k 10 300 python tools/bneck_stamps.py 32 2>&1 | grep -v amdgpu | head -12
done
for sw in "FUSG_NO_LOADER_WAVE=1" "X=1" "FUSG_NO_LOADER_WAVE=1" "X=1"; do
  env $sw timeout -k 10 300 python bench.py --precision f16x3 --no-cpu-baseline --no-clip --steps 20 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$sw', d['value'], 'crops/s  frac', r['frac'], 'conv', r['conv_ms_per_step'], 'ms  launches', r['launches_per_step'])"
done
for sw in "FUSG_NO_LOADER_WAVE=1" "X=1"; do
  echo "== small batches $sw"
  env $sw timeout -k 10 300 python tools/small_batch.py 16 2>&1 | grep -v amdgpu | cut -c1-200
done
