#!/bin/bash
# GPU box: recorded-plan replay with one issuing host thread per stream (FUSG_PLAN_MT=1) vs one thread
R=$GRAFT_REPO_ROOT
cd $R
FUSG_PLAN_MT=1 timeout -k 10 600 python -m pytest tests/test_gpu_nets.py tests/test_gpu_frame.py -x -q -m gpu -k "compiled or replay or frames or later" 2>&1 | tail -4
for mt in 0 1 0 1; do
  echo "== FUSG_PLAN_MT=$mt"
  FUSG_PLAN_MT=$mt timeout -k 10 300 python tools/small_batch.py 1 2 4 8 2>&1 | grep -v amdgpu | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['batch'], 'replay ms', d['replay']['ms_per_pass'], 'crops/s', d['replay']['crops_per_s'], 'host issue ms', d['replay']['host_issue_ms'])"
done
for mt in 0 1; do
  echo "== frame driver FUSG_PLAN_MT=$mt"
  FUSG_PLAN_MT=$mt timeout -k 10 300 python tools/frame_time.py 2>&1 | grep -v amdgpu | head -3
done
