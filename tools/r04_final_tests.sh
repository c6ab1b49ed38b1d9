#!/bin/bash
# GPU box: the whole GPU suite as the driver runs it, then one default bench line
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r04t
mkdir -p $out
cd $R
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu > $out/r04_gputest_full.log 2>&1; echo "gpu suite rc=$?"; tail -5 $out/r04_gputest_full.log
cp gpurun_out/parity_observed.json $out/r04_parity.json 2>/dev/null
timeout -k 10 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04t/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_executed'], d['roofline']['conv_ms_per_step'], d['roofline']['launches_per_step'])
for k,v in d['precision_legs'].items(): print(k, v['value'], v['roofline']['frac'], v['roofline']['conv_ms_per_step'], v['roofline']['launches_per_step'])
for k in ('cpu_baseline','clip_mode','frame_mode','clip_frame_mode','configs0_vunet_forward_b1','ssim_vs_cpu_ref','kp_idx_exact'):
    print(k, d.get(k))
PY
