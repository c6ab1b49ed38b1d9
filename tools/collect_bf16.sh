#!/bin/bash
# GPU box: the evidence for the bf16 leg (BASELINE configs[4]'s "bf16 MFMA conv path") at 256x256 B=32 and 512x512 B=16:
# rocprofv3 kernel stats of the serialised pass, MFMA-pipe busy, HBM bytes (FETCH_SIZE / WRITE_SIZE, separate passes) and the
# wave-cycle stall breakdown.  usage: tools/collect_bf16.sh r03   (then copy gpurun_out/TAG_bf16/TAG_* to profiles/)
set -e
tag=${1:-r03}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${tag}_bf16
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for cfg in "256 32" "512 16"; do
  set -- $cfg; res=$1; b=$2
  args="--res $res --batch $b --precision bf16 --no-cpu-baseline --no-clip"
  FUSG_STREAMS=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/stats --output-format csv -- python3 $R/bench.py --steps 6 --warmup 3 $args > $out/stats_$res.log 2>&1
  grep '^{"metric"' $out/stats_$res.log > $out/${tag}_bench_bf16_${res}_serial_under_rocprof.json
  cp $(ls $out/stats/*/*_kernel_stats.csv | head -1) $out/${tag}_bench_bf16_${res}_kernel_stats.csv
  rm -rf $out/stats
  FUSG_STREAMS=0 timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace -d $out/pmc_m --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --settle-s 0 $args --no-prof > $out/pmc_m_$res.log 2>&1
  python3 $R/tools/mfma_util.py $out/pmc_m > $out/${tag}_mfma_util_bf16_$res.txt
  rm -rf $out/pmc_m
  FUSG_STREAMS=0 timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_f --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --settle-s 0 $args --no-prof > $out/pmc_f_$res.log 2>&1
  FUSG_STREAMS=0 timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_w --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --settle-s 0 $args --no-prof > $out/pmc_w_$res.log 2>&1
  python3 $R/tools/hbm_traffic.py $out/pmc_f $out/pmc_w bf16_$res > $out/${tag}_hbm_traffic_bf16_$res.json
  rm -rf $out/pmc_f $out/pmc_w
  FUSG_STREAMS=0 timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --kernel-trace -d $out/pmc_s --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --settle-s 0 $args --no-prof > $out/pmc_s_$res.log 2>&1
  python3 $R/tools/pmc_sum.py $out/pmc_s --match conv_halo > $out/${tag}_stalls_bf16_halo_$res.txt
  rm -rf $out/pmc_s
  echo "bf16 $res done"
done
cp $R/profiles/hbm_traffic_latest.json $out/hbm_traffic_latest.json
ls $out
