#!/bin/bash
# GPU box: bf16 leg with the 64-column tile everywhere (FUSG_HALO_BN=64: 64 pixels x 32 columns per wave, 114-168 VGPRs = 3-4 waves per SIMD)
R=$GRAFT_REPO_ROOT
cd $R
one() {
  env $1 timeout -k 10 300 python bench.py --precision bf16 --no-cpu-baseline --no-clip --steps 20 --warmup 8 $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1 $2', d['value'], 'crops/s  frac', r['frac'], 'conv', r['conv_ms_per_step'], 'ms')"
}
one X=1 ""
one FUSG_HALO_BN=64 ""
one X=1 ""
one FUSG_HALO_BN=64 ""
one X=1 "--res 512 --batch 16"
one FUSG_HALO_BN=64 "--res 512 --batch 16"
