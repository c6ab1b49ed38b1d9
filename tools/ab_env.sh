#!/bin/bash
# GPU box: sustained A/B of one environment switch on the halo_exp.py layers matching $3:  tools/ab_env.sh NAME=VALUE_A NAME=VALUE_B filter
a=$1; b=$2; filt=$3
for rep in 1 2; do
  env $a python tools/halo_exp.py $filt 2>&1 | grep -v amdgpu | sed "s/^/[$a] /"
  env $b python tools/halo_exp.py $filt 2>&1 | grep -v amdgpu | sed "s/^/[$b] /"
done
