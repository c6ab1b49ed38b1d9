#!/bin/bash
# GPU box: A/B of the L2 weight touch and of the narrow-tile occupancy variant (one bench process per arm, f16x3 leg only)
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03c
mkdir -p $out
cd $R
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --precision f16x3 --no-cpu-baseline --no-clip --steps 20 --warmup 8 > $out/$name.log 2>&1
  grep '^{"metric"' $out/$name.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$name', d['value'], 'crops/s', d['ms_per_step'], 'ms/step  conv', r['conv_ms_per_step'], 'frac', r['frac'], r['frac_executed'], 'launches', r['launches_per_step'])"
}
run notouch_a FUSG_NO_TOUCH=1
run touch_a FUSG_X=1
run occ_a FUSG_LIB=$R/future_urban_scene_generation_amd/libfusg_occ.so
run notouch_b FUSG_NO_TOUCH=1
run touch_b FUSG_X=1
run occ_b FUSG_LIB=$R/future_urban_scene_generation_amd/libfusg_occ.so
timeout -k 10 200 python -m pytest tests/test_gpu_ops.py -q -x -k "conv or bottleneck or halo" > $out/pytest_ops.log 2>&1; tail -2 $out/pytest_ops.log
cd /tmp && export TMPDIR=/tmp
FUSG_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace -d $out/trace --output-format csv -- python3 $R/bench.py --steps 2 --warmup 2 --settle-s 0 --precision f16x3 --no-cpu-baseline --no-clip --no-prof > $out/trace.log 2>&1
cp $(ls $out/trace/*/*_kernel_trace.csv | head -1) $out/kernel_trace_touch.csv
rm -rf $out/trace
