#!/bin/bash
# GPU box: later frames with one frame in flight (run_later_frames): parity + the clip figure
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_frame.py -x -q -m gpu -k "later or clip" 2>&1 | tail -5
timeout -k 10 300 python bench.py --precision f16x3 --no-cpu-baseline --steps 5 --warmup 3 --no-prof 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('frame_mode',{}).get('ms_per_frame'), d.get('clip_frame_mode'))"
