#!/bin/bash
# GPU box: fused Bottleneck with its residual tile fetched before conv3 (libfusg_rp.so) vs after
R=$GRAFT_REPO_ROOT
P=$R/future_urban_scene_generation_amd
cd $R
FUSG_LIB=$P/libfusg_rp.so timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "bottleneck or bneck" 2>&1 | tail -2 || exit 1
FUSG_LIB=$P/libfusg_rps.so timeout -k 10 300 python tools/bneck_stamps.py 32 2>&1 | grep -v amdgpu | grep -A1 "64x64 warm\|32x32 warm\|4x4  warm"
for lib in libfusg.so libfusg_rp.so; do
  echo "== $lib"; FUSG_LIB=$P/$lib timeout -k 10 300 python tools/bneck_exp.py 32 2>&1 | grep -v amdgpu | cut -c1-60
done
for lib in libfusg.so libfusg_rp.so libfusg.so libfusg_rp.so; do
  FUSG_LIB=$P/$lib timeout -k 10 300 python bench.py --precision f16x3 --no-cpu-baseline --no-clip --steps 20 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$lib', d['value'], 'crops/s  frac', r['frac'], 'conv', r['conv_ms_per_step'], 'ms')"
done
