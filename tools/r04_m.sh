#!/bin/bash
# GPU box: phase stamps of the fused Bottleneck on the small levels (diagnostic build)
R=$GRAFT_REPO_ROOT
cd $R
FUSG_LIB=$R/future_urban_scene_generation_amd/libfusg_stamps.so timeout -k 10 300 python tools/bneck_stamps.py 32 2>&1 | grep -v amdgpu
FUSG_LIB=$R/future_urban_scene_generation_amd/libfusg_stamps.so timeout -k 10 300 python tools/bneck_stamps.py 8 2>&1 | grep -v amdgpu
timeout -k 10 300 python tools/later_frame_time.py 2>&1 | grep -v amdgpu | cut -c1-170
