#!/bin/bash
# GPU box: bf16 evidence of the round's final kernels (three waves per SIMD) + the three bench lines that carry a bf16 leg
R=$GRAFT_REPO_ROOT
cd $R
bash tools/collect_bf16.sh r04 > gpurun_out/r04_bf16_collect.log 2>&1; tail -3 gpurun_out/r04_bf16_collect.log
out=$R/gpurun_out/r04
mkdir -p $out
timeout -k 10 600 python bench.py > $out/bench_cfg1.log 2>&1 && grep '^{"metric"' $out/bench_cfg1.log > $out/r04_bench_cfg1.json
timeout -k 10 500 python bench.py --inpaint --no-cpu-baseline > $out/bench_cfg2.log 2>&1 && grep '^{"metric"' $out/bench_cfg2.log > $out/r04_bench_cfg2_inpaint.json
timeout -k 10 300 python bench.py --res 512 --batch 16 --precision bf16 --no-cpu-baseline --no-clip > $out/bench_512_bf16.log 2>&1 && grep '^{"metric"' $out/bench_512_bf16.log > $out/r04_bench_512_b16_bf16.json
python - <<'PY'
import json
for f in ('r04_bench_cfg1','r04_bench_cfg2_inpaint','r04_bench_512_b16_bf16'):
    d=json.loads(open('gpurun_out/r04/%s.json'%f).read().strip().splitlines()[-1])
    print(f, {k:(v['value'], v['roofline']['frac'], v['roofline']['conv_ms_per_step']) for k,v in d['precision_legs'].items()}, d.get('clip_frame_mode',{}).get('ms_per_clip'))
PY
