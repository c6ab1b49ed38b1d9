#!/bin/bash
# GPU box: bf16 mode of the halo kernel at three (libfusg.so) / four (libfusg_occ4.so) waves per SIMD vs two (libfusg_base.so)
R=$GRAFT_REPO_ROOT
P=$R/future_urban_scene_generation_amd
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_nets.py -x -q -m gpu -k "bf16 or halo" 2>&1 | tail -3 || exit 1
one() {  # $1 = lib, $2 = precision, $3.. = bench args
  lib=$1; prec=$2; shift; shift
  FUSG_LIB=$P/$lib timeout -k 10 300 python bench.py --precision $prec --no-cpu-baseline --no-clip --steps 20 --warmup 8 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$lib $prec $*', d['value'], 'crops/s  frac', r['frac'], 'conv', r['conv_ms_per_step'], 'ms')"
}
for rep in 1 2; do
  one libfusg_base.so bf16
  one libfusg.so bf16
  one libfusg_occ4.so bf16
done
one libfusg_base.so bf16 --res 512 --batch 16
one libfusg.so bf16 --res 512 --batch 16
one libfusg_occ4.so bf16 --res 512 --batch 16
one libfusg_base.so bf16 --inpaint
one libfusg.so bf16 --inpaint
one libfusg_base.so f16x3
one libfusg.so f16x3
