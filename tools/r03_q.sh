#!/bin/bash
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03q
mkdir -p $out
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_nets.py -q -x -k "up2 or icn or ICN or pipeline or golden or high_res or non_square" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
run() {  # name, precision, env...
  name=$1; prec=$2; shift; shift
  env "$@" timeout -k 10 300 python bench.py --precision $prec --no-cpu-baseline --no-clip --steps 20 --warmup 8 > $out/$name.log 2>&1
  grep '^{"metric"' $out/$name.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$name', d['value'], 'crops/s', d['ms_per_step'], 'ms/step  conv', r['conv_ms_per_step'], 'frac', r['frac'], r['frac_executed'], 'launches', r['launches_per_step'])"
}
run ring25_a f16x3 FUSG_UP2_RING25=1
run ring9_a f16x3 FUSG_X=1
run ring25_b f16x3 FUSG_UP2_RING25=1
run ring9_b f16x3 FUSG_X=1
