#!/bin/bash
# GPU box: instruction counts of the 32-column halo kernel with its parts cut off (needs libfusg_abl1/2.so: builds of conv_halo_32
# with an early return after the set-up / after the loop - the hooks were temporary and are not in the tree; see profiles/r03_narrow_ablation.txt): where do the
# VALU / SALU instructions of a narrow launch go?  abl1 = set-up only, abl2 = set-up + main loop, full.
R=$GRAFT_REPO_ROOT
export PMC_SETS="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAVE_CYCLES;SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CU_CYCLES"
for v in abl1 abl2 full; do
  if [ $v = full ]; then unset FUSG_LIB; else export FUSG_LIB=$R/future_urban_scene_generation_amd/libfusg_$v.so; fi
  for layer in "vu 64->32 3x3" "vu 32->32 3x3"; do
    tag=$(echo "$v $layer" | tr ' >' '__' | tr -d '-')
    bash $R/tools/pmc_halo.sh "$layer" r03abl/$tag || exit 1
    echo "== $v $layer"; grep -E "SQ_INSTS_|SQ_WAVE_CYCLES|SQ_BUSY_CU" $R/gpurun_out/r03abl/$tag/summary.txt | cut -c1-120
  done
done
