#!/bin/bash
# GPU box: exact-fp32 tap-unit stems + later-frame replay: parity, then the f32 leg and the clip figure
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r04k
mkdir -p $out
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "tapunit or f32_halo or pointwise or conv_vs_torch" > $out/ops_tests.log 2>&1; echo "ops tests rc=$?"; tail -5 $out/ops_tests.log
timeout -k 10 900 python -m pytest tests/test_gpu_frame.py tests/test_gpu_nets.py -x -q -m gpu -k "later or icn or edgeconnect or fp32 or hourglass" > $out/net_tests.log 2>&1; echo "net tests rc=$?"; tail -5 $out/net_tests.log
for arm in "FUSG_NO_F32_HALO=1" "base" "base"; do
  if [ "$arm" = base ]; then e=""; else e="$arm"; fi
  env $e timeout -k 10 300 python bench.py --precision f32 --no-cpu-baseline --steps 10 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('f32 $arm', d['value'], r['frac'], r['conv_ms_per_step'], r['launches_per_step'], 'clip_frame', d.get('clip_frame_mode',{}).get('ms_per_clip'), 'frame', d.get('frame_mode',{}).get('ms_per_frame'))"
done
timeout -k 10 300 python bench.py --precision f16x3 --no-cpu-baseline --steps 20 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('f16x3', d['value'], r['frac'], r['conv_ms_per_step'], r['launches_per_step'], 'clip_frame', d.get('clip_frame_mode'), 'frame', d.get('frame_mode',{}).get('ms_per_frame'))"
