#!/bin/bash
# GPU box: regenerate the artefacts kept under profiles/ (run through gpurun, then copy gpurun_out/TAG/TAG_* to profiles/).
# usage: tools/collect_profiles.sh r02 [quick]
set -e
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout -k 10 600 python bench.py > $out/bench_cfg1.log 2>&1 && grep '^{"metric"' $out/bench_cfg1.log > $out/${tag}_bench_cfg1.json
echo "bench cfg1 (both precision legs) done"
timeout -k 10 500 python bench.py --inpaint --no-cpu-baseline > $out/bench_cfg2.log 2>&1 && grep '^{"metric"' $out/bench_cfg2.log > $out/${tag}_bench_cfg2_inpaint.json
echo "bench cfg2 done"
timeout -k 10 500 python bench.py --res 512 --batch 16 --precision f16x3 --no-cpu-baseline --no-clip > $out/bench_512.log 2>&1 && grep '^{"metric"' $out/bench_512.log > $out/${tag}_bench_512_b16_f16x3.json
timeout -k 10 300 python bench.py --res 512 --batch 16 --precision bf16 --no-cpu-baseline --no-clip > $out/bench_512_bf16.log 2>&1 && grep '^{"metric"' $out/bench_512_bf16.log > $out/${tag}_bench_512_b16_bf16.json
echo "bench 512 done"
FUSG_DIST_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --batch 16 --precision f16x3 --no-cpu-baseline --no-clip --no-prof > $out/bench_2rank.log 2>&1 && grep '^{"metric"' $out/bench_2rank.log > $out/${tag}_bench_2ranks_one_card_gloo.json
echo "2-rank rehearsal done"
timeout -k 10 300 python tools/small_batch.py 2>&1 | grep -v amdgpu.ids > $out/${tag}_small_batch.jsonl
echo "small batch done"
cd /tmp && export TMPDIR=/tmp
# kernel-trace stats of the serialised pass (what the roofline leg measures)
FUSG_STREAMS=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/stats --output-format csv -- python3 $R/bench.py --steps 10 --warmup 5 --precision f16x3 --no-cpu-baseline --no-clip > $out/stats.log 2>&1
grep '^{"metric"' $out/stats.log > $out/${tag}_bench_cfg1_serial_under_rocprof.json
cp $(ls $out/stats/*/*_kernel_stats.csv | head -1) $out/${tag}_bench_cfg1_kernel_stats.csv
# the exact-fp32 leg (round 4: halo kernel mode 2, fused Bottleneck in fp32)
FUSG_STREAMS=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/stats32 --output-format csv -- python3 $R/bench.py --steps 5 --warmup 3 --precision f32 --no-cpu-baseline --no-clip > $out/stats32.log 2>&1
grep '^{"metric"' $out/stats32.log > $out/${tag}_bench_f32_serial_under_rocprof.json
cp $(ls $out/stats32/*/*_kernel_stats.csv | head -1) $out/${tag}_bench_f32_kernel_stats.csv
echo "rocprof stats done"
# HBM traffic: separate PMC passes (FETCH_SIZE, WRITE_SIZE)
FUSG_STREAMS=0 timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_f --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --precision f16x3 --no-cpu-baseline --no-clip --no-prof > $out/pmc_f.log 2>&1
FUSG_STREAMS=0 timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_w --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --precision f16x3 --no-cpu-baseline --no-clip --no-prof > $out/pmc_w.log 2>&1
# matrix-pipe utilisation
FUSG_STREAMS=0 timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --kernel-trace -d $out/pmc_m --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --settle-s 0 --precision f16x3 --no-cpu-baseline --no-clip --no-prof > $out/pmc_m.log 2>&1
cd $R
python3 tools/hbm_traffic.py $out/pmc_f $out/pmc_w > $out/${tag}_hbm_traffic_cfg1_f16x3.json
cp profiles/hbm_traffic_latest.json $out/hbm_traffic_latest.json
python3 tools/mfma_util.py $out/pmc_m > $out/${tag}_mfma_util.txt
echo "pmc done"
timeout -k 10 200 python tools/bneck_exp.py 2>&1 | grep -v amdgpu.ids > $out/${tag}_bneck_fused_vs_3_launches.txt
timeout -k 10 200 python tools/issue_time.py 2>&1 | grep -v amdgpu.ids > $out/${tag}_replay_issue_vs_done.jsonl
FUSG_STREAMS=0 timeout -k 10 300 python tools/layer_profile.py --top 90 2>&1 | grep -v amdgpu.ids > $out/${tag}_layer_profile.txt
FUSG_STREAMS=0 FUSG_PRECISION=f32 timeout -k 10 300 python tools/layer_profile.py --top 60 2>&1 | grep -v amdgpu.ids > $out/${tag}_layer_profile_f32.txt
timeout -k 10 300 python tools/small_exp.py 32 2>&1 | grep -v amdgpu.ids > $out/${tag}_small_image_kernel_per_layer.txt
FUSG_STREAMS=0 timeout -k 10 300 python tools/layer_profile.py --batch 1 --top 40 2>&1 | grep -v amdgpu.ids > $out/${tag}_layer_profile_b1.txt
timeout -k 10 400 python tools/halo_exp.py 2>&1 | grep -v amdgpu.ids > $out/${tag}_halo_layers_sustained.txt
rm -rf $out/stats $out/stats32 $out/pmc_f $out/pmc_w $out/pmc_m
echo "all done"
