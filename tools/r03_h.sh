#!/bin/bash
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03h
mkdir -p $out
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_nets.py -q -x -k "bf16" > $out/pytest_bf16.log 2>&1; tail -2 $out/pytest_bf16.log
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --precision bf16 --no-cpu-baseline --no-clip --steps 20 --warmup 8 > $out/$name.log 2>&1
  grep '^{"metric"' $out/$name.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$name', d['value'], 'crops/s', d['ms_per_step'], 'ms/step  conv', r['conv_ms_per_step'], 'frac', r['frac'], r['frac_executed'], 'launches', r['launches_per_step'])"
}
run prev_a FUSG_LIB=$R/future_urban_scene_generation_amd/libfusg_prev.so
run new_a FUSG_X=1
run prev_b FUSG_LIB=$R/future_urban_scene_generation_amd/libfusg_prev.so
run new_b FUSG_X=1
