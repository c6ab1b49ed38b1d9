#!/bin/bash
# GPU box: transposed accumulators + direct epilogue of the halo kernel (libfusg.so) against the build before it (libfusg_base.so)
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r04l
mkdir -p $out
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > $out/ops_tests.log 2>&1; rc=$?; echo "ops tests rc=$rc"; tail -4 $out/ops_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -k "not full_size" > $out/net_tests.log 2>&1; rc=$?; echo "net tests rc=$rc"; tail -4 $out/net_tests.log
[ $rc -eq 0 ] || exit 1
one() {  # $1 = label, $2 = lib, $3 = precision
  FUSG_LIB=$R/future_urban_scene_generation_amd/$2 timeout -k 10 300 python bench.py --precision $3 --no-cpu-baseline --no-clip --steps 20 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1 $3', d['value'], 'crops/s  frac', r['frac'], 'conv', r['conv_ms_per_step'], 'ms  launches', r['launches_per_step'])"
}
for rep in 1 2; do
  one base libfusg_base.so f16x3
  one new  libfusg.so f16x3
done
one base libfusg_base.so bf16
one new  libfusg.so bf16
one base libfusg_base.so f32
one new  libfusg.so f32
for lib in libfusg_base.so libfusg.so; do
  FUSG_LIB=$R/future_urban_scene_generation_amd/$lib timeout -k 10 300 python tools/icn_layer_exp.py 2>&1 | grep -v amdgpu
done
for f in "vu 128->128 3x3 @256" "hg 256->128 1x1" "vu 32->32 1x1" "vu 64->32" "hg 128->256 1x1"; do
  for lib in libfusg_base.so libfusg.so; do
    FUSG_LIB=$R/future_urban_scene_generation_amd/$lib timeout -k 10 120 python tools/halo_exp.py "$f" 2>&1 | grep -v "amdgpu\|^kernel"
  done
done
