#!/bin/bash
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03m
mkdir -p $out
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -q -x > $out/pytest_ops.log 2>&1; tail -3 $out/pytest_ops.log
run() {  # name, precision, env...
  name=$1; prec=$2; shift; shift
  env "$@" timeout -k 10 300 python bench.py --precision $prec --no-cpu-baseline --no-clip --steps 20 --warmup 8 > $out/$name.log 2>&1
  grep '^{"metric"' $out/$name.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$name', d['value'], 'crops/s', d['ms_per_step'], 'ms/step  conv', r['conv_ms_per_step'], 'frac', r['frac'], r['frac_executed'], 'launches', r['launches_per_step'])"
}
run msplit_a f16x3 FUSG_NO_KSPLIT=1
run ksplit_a f16x3 FUSG_X=1
run msplit_b f16x3 FUSG_NO_KSPLIT=1
run ksplit_b f16x3 FUSG_X=1
run bf_msplit bf16 FUSG_NO_KSPLIT=1
run bf_ksplit bf16 FUSG_X=1
cd /tmp && export TMPDIR=/tmp
FUSG_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace -d $out/trace --output-format csv -- python3 $R/bench.py --steps 2 --warmup 2 --settle-s 0 --precision f16x3 --no-cpu-baseline --no-clip --no-prof > $out/trace.log 2>&1
cp $(ls $out/trace/*/*_kernel_trace.csv | head -1) $out/kernel_trace_ksplit.csv
rm -rf $out/trace
FUSG_NO_KSPLIT=1 FUSG_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace -d $out/trace --output-format csv -- python3 $R/bench.py --steps 2 --warmup 2 --settle-s 0 --precision f16x3 --no-cpu-baseline --no-clip --no-prof > $out/trace2.log 2>&1
cp $(ls $out/trace/*/*_kernel_trace.csv | head -1) $out/kernel_trace_msplit.csv
rm -rf $out/trace
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_nets.py -q -x -k "not full_size and not config3 and not 512" > $out/pytest_nets.log 2>&1; tail -3 $out/pytest_nets.log
