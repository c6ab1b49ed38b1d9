#!/bin/bash
# GPU box: the wide split-fp16 tile at three waves per SIMD with its MFMAs in patch-row-major order (libfusg_fd.so: NI <= 6, no affine pre-op) vs two
R=$GRAFT_REPO_ROOT
P=$R/future_urban_scene_generation_amd
cd $R
FUSG_LIB=$P/libfusg_fd.so timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "halo or conv_vs or scale_sweep" 2>&1 | tail -2 || exit 1
for f in "vu 128->128 3x3 @256" "icn 256->256"; do
  for lib in libfusg.so libfusg_fd.so; do
    FUSG_LIB=$P/$lib timeout -k 10 120 python tools/halo_exp.py "$f" 2>&1 | grep -v "amdgpu\|^kernel"
  done
done
for lib in libfusg.so libfusg_fd.so libfusg.so libfusg_fd.so; do
  FUSG_LIB=$P/$lib timeout -k 10 300 python bench.py --precision f16x3 --no-cpu-baseline --no-clip --steps 20 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$lib', d['value'], 'crops/s  frac', r['frac'], 'conv', r['conv_ms_per_step'], 'ms')"
done
