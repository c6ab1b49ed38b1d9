#!/bin/bash
# GPU box: "fusion by cache blocking" of the VUnet's high-resolution 32-channel blocks (FUSG_VU_SUBBATCH)
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r04g
mkdir -p $out
cd $R
FUSG_VU_SUBBATCH=1 timeout -k 10 600 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -k "vunet and not full_size" > $out/vunet_tests_sub1.log 2>&1; echo "vunet tests (subbatch 1) rc=$?"; tail -3 $out/vunet_tests_sub1.log
for n in 0 8 4 16 0 8; do
  FUSG_VU_SUBBATCH=$n timeout -k 10 300 python bench.py --precision f16x3 --no-cpu-baseline --no-clip --steps 20 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('subbatch $n', d['value'], r['frac'], r['conv_ms_per_step'], r['launches_per_step'])"
done
