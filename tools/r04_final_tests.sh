#!/bin/bash
# GPU box: smoke(), the whole GPU suite as the driver runs it, then one default bench line
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r04t
mkdir -p $out
cd $R
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu > $out/r04_gputest_full.log 2>&1; echo "gpu suite rc=$?"; tail -5 $out/r04_gputest_full.log
cp gpurun_out/parity_observed.json $out/r04_parity.json 2>/dev/null
t0=$(date +%s)
timeout -k 10 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$? in $(( $(date +%s) - t0 )) s"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04t/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_executed'], d['roofline']['conv_ms_per_step'], d['roofline']['launches_per_step'], d['roofline']['traffic'])
for k,v in d['precision_legs'].items(): print(k, v['value'], v['roofline']['frac'], v['roofline']['conv_ms_per_step'], v['roofline']['launches_per_step'])
for k in ('clip_mode','frame_mode','clip_frame_mode','configs0_vunet_forward_b1','ssim_vs_cpu_ref','kp_idx_exact'):
    print(k, str(d.get(k))[:300])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
