#!/bin/bash
# GPU box: A/B of the tree's libfusg.so against a saved baseline build (future_urban_scene_generation_amd/libfusg_base.so), one bench
# process per arm on the same card, alternating.  usage: tools/r03_ab_lib.sh "<pytest -k expression>" [bench args of a second pair of arms]
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03ab
mkdir -p $out
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_nets.py -x -q -k "$1" > $out/ops.log 2>&1; tail -2 $out/ops.log
arm() {
  if [ $1 = base ]; then export FUSG_LIB=$R/future_urban_scene_generation_amd/libfusg_base.so; else unset FUSG_LIB; fi
  shift
  python bench.py --precision f16x3 --no-cpu-baseline --no-clip "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['conv_ms_per_step'])"
}
for a in base new base new; do echo -n "$a B=32: "; arm $a --steps 30 --warmup 15; done
if [ -n "$2" ]; then for a in base new base new; do echo -n "$a $2: "; arm $a $2; done; fi
