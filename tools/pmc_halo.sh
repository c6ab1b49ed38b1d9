#!/bin/bash
# GPU box: PMC passes over one halo layer (analysis tool).
# usage: tools/pmc_halo.sh "icn 256" outdir ["1 4"]     (optional list of pass numbers to run)
# env PMC_SETS="A B C;D E" overrides the counter sets (one rocprofv3 pass per ';'-separated set)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$2
mkdir -p $out
DEF="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS;SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL;SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL;SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY;SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_IFETCH SQ_LDS_CMD_FIFO_FULL"
IFS=';' read -ra SETS <<< "${PMC_SETS:-$DEF}"
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  if [ -n "$3" ] && [[ " $3 " != *" $i "* ]]; then continue; fi
  timeout -k 10 120 rocprofv3 --pmc $set --kernel-trace -d $out/p$i --output-format csv -- python3 $R/tools/halo_exp.py "$1" --burst > $out/p$i.log 2>&1 || { tail -5 $out/p$i.log; exit 1; }
done
python3 $R/tools/pmc_sum.py $out > $out/summary.txt
