#!/bin/bash
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03l
mkdir -p $out
cd $R
timeout -k 10 200 python -m pytest tests/test_gpu_ops.py -q -x -k "pointwise" 2>&1 | tail -3
for f in "vu 6->128" "vu 3->32"; do
  timeout -k 10 100 python tools/halo_exp.py "$f" 2>&1 | grep -v amdgpu.ids
  FUSG_NO_POINTWISE=1 timeout -k 10 100 python tools/halo_exp.py "$f" 2>&1 | grep -v amdgpu.ids
done
