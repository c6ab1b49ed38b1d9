#!/bin/bash
# GPU box: A/B of one libfusg variant against the saved baseline build, one bench process per arm, same card
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03ab
mkdir -p $out
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -k "halo or conv" > $out/ops.log 2>&1; tail -2 $out/ops.log
for arm in base new base new; do
  if [ $arm = base ]; then export FUSG_LIB=$R/future_urban_scene_generation_amd/libfusg_base.so; else unset FUSG_LIB; fi
  python bench.py --precision f16x3 --no-cpu-baseline --no-clip --steps 30 --warmup 15 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$arm', d['value'], r['frac'], r['conv_ms_per_step'])"
done
