#!/bin/bash
# GPU box: fused Bottleneck, scheduling variant V2 (all conv1 fragments read up front, conv2 fragments one step ahead) vs base
R=$GRAFT_REPO_ROOT
P=$R/future_urban_scene_generation_amd
cd $R
FUSG_LIB=$P/libfusg_v2.so timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "bottleneck or bneck" 2>&1 | tail -3
for lib in libfusg_stamps.so libfusg_v2s.so; do
  echo "== $lib"
  FUSG_LIB=$P/$lib timeout -k 10 300 python tools/bneck_stamps.py 32 2>&1 | grep -v amdgpu
done
for lib in libfusg.so libfusg_v2.so libfusg.so libfusg_v2.so; do
  echo "== $lib"
  FUSG_LIB=$P/$lib timeout -k 10 300 python tools/bneck_exp.py 32 2>&1 | grep -v amdgpu | cut -c1-60
done
