#!/bin/bash
# GPU box: the few-channel stems in single-pass bf16 (tap-unit kernel MODE 1) under FUSG_PREC_BF16 vs FUSG_NO_BF16_TAPUNIT=1 (split-fp16 stems)
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_nets.py -x -q -m gpu -k "bf16 or tapunit" 2>&1 | tail -3 || exit 1
one() {
  env $1 timeout -k 10 300 python bench.py --precision bf16 --no-cpu-baseline --no-clip --steps 20 --warmup 8 $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1 $2', d['value'], 'crops/s  frac', r['frac'], 'conv', r['conv_ms_per_step'], 'ms')"
}
for rep in 1 2; do
  one FUSG_NO_BF16_TAPUNIT=1 ""
  one X=1 ""
done
one FUSG_NO_BF16_TAPUNIT=1 "--res 512 --batch 16"
one X=1 "--res 512 --batch 16"
one FUSG_NO_BF16_TAPUNIT=1 "--inpaint"
one X=1 "--inpaint"
