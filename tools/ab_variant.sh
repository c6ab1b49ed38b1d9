#!/bin/bash
# GPU box: sustained A/B of libfusg.so against libfusg_$1.so on the halo_exp.py layers matching $2 (two interleaved rounds)
v=$1; filt=$2
mkdir -p gpurun_out/ab_$v
for rep in 1 2; do
  unset FUSG_LIB
  timeout -k 10 400 python tools/halo_exp.py $filt 2>&1 | grep -v amdgpu > gpurun_out/ab_$v/base_$rep.txt
  export FUSG_LIB=$PWD/future_urban_scene_generation_amd/libfusg_$v.so
  timeout -k 10 400 python tools/halo_exp.py $filt 2>&1 | grep -v amdgpu > gpurun_out/ab_$v/${v}_$rep.txt
done
paste -d'\n' gpurun_out/ab_$v/base_1.txt gpurun_out/ab_$v/${v}_1.txt gpurun_out/ab_$v/base_2.txt gpurun_out/ab_$v/${v}_2.txt
