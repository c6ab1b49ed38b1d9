#!/bin/bash
# GPU box: halo kernel with the A fragments read one tap ahead (bf16 mode + the 32-column split-fp16 tile) vs the build before it
R=$GRAFT_REPO_ROOT
P=$R/future_urban_scene_generation_amd
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -3 || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -k "not full_size" 2>&1 | tail -3 || exit 1
FUSG_LIB=$P/libfusg_hstamps.so timeout -k 10 300 python tools/halo_stamps.py 2>&1 | grep -v "amdgpu\|no stamps"
one() {  # $1 = lib, $2 = precision
  FUSG_LIB=$P/$1 timeout -k 10 300 python bench.py --precision $2 --no-cpu-baseline --no-clip --steps 20 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1 $2', d['value'], 'crops/s  frac', r['frac'], 'conv', r['conv_ms_per_step'], 'ms  launches', r['launches_per_step'])"
}
for rep in 1 2; do
  one libfusg_base.so bf16
  one libfusg.so bf16
done
for rep in 1 2; do
  one libfusg_base.so f16x3
  one libfusg_bfonly.so f16x3
  one libfusg.so f16x3
done
FUSG_LIB=$P/libfusg_base.so timeout -k 10 300 python bench.py --res 512 --batch 16 --precision bf16 --no-cpu-baseline --no-clip --steps 10 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base 512 bf16', d['value'], d['roofline']['frac'])"
timeout -k 10 300 python bench.py --res 512 --batch 16 --precision bf16 --no-cpu-baseline --no-clip --steps 10 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new  512 bf16', d['value'], d['roofline']['frac'])"
