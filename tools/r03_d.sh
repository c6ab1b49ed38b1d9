#!/bin/bash
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03d
mkdir -p $out
cd $R
timeout -k 10 200 python tools/layer_variants_exp.py icn 2>&1 | grep -v amdgpu.ids | tee $out/variants_icn.txt
timeout -k 10 200 python tools/layer_variants_exp.py vu 2>&1 | grep -v amdgpu.ids | tee $out/variants_vu.txt
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --precision f16x3 --no-cpu-baseline --no-clip --steps 20 --warmup 8 > $out/$name.log 2>&1
  grep '^{"metric"' $out/$name.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$name', d['value'], 'crops/s', d['ms_per_step'], 'ms/step  conv', r['conv_ms_per_step'], 'frac', r['frac'], r['frac_executed'], 'launches', r['launches_per_step'])"
}
run notouch_a FUSG_NO_TOUCH=1
run pd2_a FUSG_NO_TOUCH=1 FUSG_LIB=$R/future_urban_scene_generation_amd/libfusg_pd2.so
run notouch_b FUSG_NO_TOUCH=1
run pd2_b FUSG_NO_TOUCH=1 FUSG_LIB=$R/future_urban_scene_generation_amd/libfusg_pd2.so
