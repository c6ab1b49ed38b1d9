#!/bin/bash
# GPU box: where a B = 8 pass spends its time: per-queue busy time / overlap of the multi-stream pass, and the per-network serial sums
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/tl8 --output-format csv -- python3 $R/bench.py --batch 8 --steps 8 --warmup 4 --precision f16x3 --no-prof --no-cpu-baseline --no-clip > /tmp/tl8.log 2>&1
python3 $R/tools/timeline.py /tmp/tl8
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/tl8r --output-format csv -- python3 $R/bench.py --batch 8 --steps 8 --warmup 4 --precision f16x3 --no-prof --no-cpu-baseline --no-clip --replay > /tmp/tl8r.log 2>&1
echo "== replay"; python3 $R/tools/timeline.py /tmp/tl8r
cd $R
FUSG_STREAMS=0 timeout -k 10 300 python tools/layer_profile.py --batch 8 --top 25 2>&1 | grep -v amdgpu.ids
