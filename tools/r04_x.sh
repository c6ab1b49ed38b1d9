#!/bin/bash
# GPU box: two halo images in LDS (FUSG_HALO_DBUF=1: chunk boundary = commit + ONE barrier) vs one (default), and vs the build before the change
R=$GRAFT_REPO_ROOT
P=$R/future_urban_scene_generation_amd
cd $R
FUSG_HALO_DBUF=1 timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -2 || exit 1
FUSG_HALO_DBUF=1 timeout -k 10 900 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -k "not full_size" 2>&1 | tail -2 || exit 1
one() {  # $1 = env, $2 = lib, $3 = precision
  env $1 FUSG_LIB=$P/$2 timeout -k 10 300 python bench.py --precision $3 --no-cpu-baseline --no-clip --steps 20 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1 $2 $3', d['value'], 'crops/s  frac', r['frac'], 'conv', r['conv_ms_per_step'], 'ms')"
}
for rep in 1 2; do
  one X=1 libfusg_base.so f16x3
  one X=1 libfusg.so f16x3
  one FUSG_HALO_DBUF=1 libfusg.so f16x3
done
one X=1 libfusg_base.so bf16
one X=1 libfusg.so bf16
one FUSG_HALO_DBUF=1 libfusg.so bf16
one X=1 libfusg_base.so f32
one X=1 libfusg.so f32
one FUSG_HALO_DBUF=1 libfusg.so f32
