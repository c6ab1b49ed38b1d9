#!/bin/bash
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r04i
mkdir -p $out
cd $R
timeout -k 10 300 python tools/small_exp.py ring 32 > $out/ring_exp.txt 2>&1; echo "ring exp rc=$?"; cat $out/ring_exp.txt
