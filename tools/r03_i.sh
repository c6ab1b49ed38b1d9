#!/bin/bash
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03j
mkdir -p $out
cd $R
run() {  # name, precision, env...
  name=$1; prec=$2; shift; shift
  env "$@" timeout -k 10 300 python bench.py --precision $prec --no-cpu-baseline --no-clip --steps 20 --warmup 8 > $out/$name.log 2>&1
  grep '^{"metric"' $out/$name.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$name', d['value'], 'crops/s', d['ms_per_step'], 'ms/step  conv', r['conv_ms_per_step'], 'frac', r['frac'], r['frac_executed'], 'launches', r['launches_per_step'])"
}
L=$R/future_urban_scene_generation_amd
run prev_a f16x3 FUSG_LIB=$L/libfusg_prev.so

run w14_a f16x3 FUSG_LIB=$L/libfusg_w14.so
run prev_b f16x3 FUSG_LIB=$L/libfusg_prev.so

run w14_b f16x3 FUSG_LIB=$L/libfusg_w14.so

run bf_w14 bf16 FUSG_LIB=$L/libfusg_w14.so
run bf_prev bf16 FUSG_LIB=$L/libfusg_prev.so

