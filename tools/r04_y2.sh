#!/bin/bash
R=$GRAFT_REPO_ROOT
P=$R/future_urban_scene_generation_amd
cd $R
for lib in libfusg.so libfusg_h64.so libfusg.so libfusg_h64.so libfusg.so libfusg_h64.so; do
  FUSG_LIB=$P/$lib timeout -k 10 300 python bench.py --precision f16x3 --no-cpu-baseline --no-clip --steps 30 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$lib', d['value'], 'crops/s  frac', r['frac'], 'conv', r['conv_ms_per_step'], 'ms')"
done
for lib in libfusg.so libfusg_h64.so; do
  FUSG_LIB=$P/$lib timeout -k 10 300 python tools/small_batch.py 1 8 2>&1 | grep -v amdgpu | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$lib', d['batch'], 'replay ms', d['replay']['ms_per_pass'], 'crops/s', d['replay']['crops_per_s'])"
done
