#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
export FUSG_NO_TOUCH=1
bash tools/pmc_halo.sh "vu 64->32" r03e_vu64 > /dev/null 2>&1; cat gpurun_out/r03e_vu64/summary.txt
bash tools/pmc_halo.sh "vu 32->32 3x3" r03e_vu32 > /dev/null 2>&1; cat gpurun_out/r03e_vu32/summary.txt
bash tools/pmc_halo.sh "hg 256->128 1x1" r03e_hg1x1 > /dev/null 2>&1; cat gpurun_out/r03e_hg1x1/summary.txt
bash tools/pmc_halo.sh "icn 256->256" r03e_icn > /dev/null 2>&1; cat gpurun_out/r03e_icn/summary.txt
rm -rf gpurun_out/r03e_*/p[0-9]
