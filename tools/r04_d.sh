#!/bin/bash
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r04d
mkdir -p $out
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > $out/ops_tests.log 2>&1; echo "ops tests rc=$?"; tail -8 $out/ops_tests.log
timeout -k 10 900 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -k "not full_size and not config3" > $out/nets_tests.log 2>&1; echo "nets tests rc=$?"; tail -8 $out/nets_tests.log
for arm in old new old new; do
  if [ $arm = old ]; then export FUSG_NO_F32_HALO=1; else unset FUSG_NO_F32_HALO; fi
  timeout -k 10 300 python bench.py --precision f32 --no-cpu-baseline --no-clip --steps 10 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('f32 $arm', d['value'], r['frac'], r['conv_ms_per_step'], r['launches_per_step'])"
done
unset FUSG_NO_F32_HALO
timeout -k 10 300 python tools/layer_profile.py > $out/layer_profile_f16x3.txt 2>&1; head -12 $out/layer_profile_f16x3.txt
FUSG_PRECISION=f32 timeout -k 10 400 python tools/layer_profile.py > $out/layer_profile_f32.txt 2>&1; head -40 $out/layer_profile_f32.txt
