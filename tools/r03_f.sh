#!/bin/bash
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03f
mkdir -p $out
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -q -x > $out/pytest_ops.log 2>&1; tail -2 $out/pytest_ops.log
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --precision f16x3 --no-cpu-baseline --no-clip --steps 20 --warmup 8 > $out/$name.log 2>&1
  grep '^{"metric"' $out/$name.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$name', d['value'], 'crops/s', d['ms_per_step'], 'ms/step  conv', r['conv_ms_per_step'], 'frac', r['frac'], r['frac_executed'], 'launches', r['launches_per_step'])"
}
run prev_a FUSG_LIB=$R/future_urban_scene_generation_amd/libfusg_prev.so
run new_a FUSG_X=1
run prev_b FUSG_LIB=$R/future_urban_scene_generation_amd/libfusg_prev.so
run new_b FUSG_X=1
cd /tmp && export TMPDIR=/tmp
FUSG_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace -d $out/trace --output-format csv -- python3 $R/bench.py --steps 2 --warmup 2 --settle-s 0 --precision f16x3 --no-cpu-baseline --no-clip --no-prof > $out/trace.log 2>&1
cp $(ls $out/trace/*/*_kernel_trace.csv | head -1) $out/kernel_trace_new.csv
rm -rf $out/trace
FUSG_LIB=$R/future_urban_scene_generation_amd/libfusg_prev.so FUSG_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace -d $out/trace --output-format csv -- python3 $R/bench.py --steps 2 --warmup 2 --settle-s 0 --precision f16x3 --no-cpu-baseline --no-clip --no-prof > $out/trace2.log 2>&1
cp $(ls $out/trace/*/*_kernel_trace.csv | head -1) $out/kernel_trace_prev.csv
rm -rf $out/trace
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_nets.py -q -x -k "not full_size and not config3 and not 512" > $out/pytest_nets.log 2>&1; tail -2 $out/pytest_nets.log
