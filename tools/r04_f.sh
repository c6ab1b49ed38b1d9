#!/bin/bash
# GPU box: who issues the __amd_rocclr_copyBuffer launches of a pass?  hip-trace + memory-copy-trace of 10 passes, stats only
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r04f
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --hip-trace --memory-copy-trace --kernel-trace --stats -d /tmp/prof_r04f -o onepass -- python3 $R/tools/one_pass.py 32 10 > $out/onepass.log 2>&1; echo "rocprof rc=$?"
find /tmp/prof_r04f -name "*stats*" | head -20
for f in $(find /tmp/prof_r04f -name "*hip_api_stats.csv") $(find /tmp/prof_r04f -name "*memory_copy_stats.csv"); do echo "== $f"; head -30 $f | cut -c1-180; cp $f $out/; done
f=$(find /tmp/prof_r04f -name "*kernel_stats.csv" | head -1); grep -i "copyBuffer\|fillBuffer" $f | cut -c1-200
f=$(find /tmp/prof_r04f -name "*memory_copy_trace.csv" | head -1); echo "copies: $(wc -l < $f)"; head -3 $f; python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
c = collections.Counter()
for r in rows:
    c[(r.get("Direction", r.get("direction", "?")), r.get("Size", r.get("size", "?")))] += 1
for k, v in c.most_common(20):
    print(v, k)
PY
cd $R
timeout -k 10 300 python bench.py --precision f16x3 --no-cpu-baseline --steps 20 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('f16x3 nhwc inputs', d['value'], r['frac'], r['conv_ms_per_step'], r['launches_per_step'], d.get('frame_mode',{}).get('ms_per_frame'), d.get('clip_frame_mode',{}).get('ms_per_clip'))"
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > $out/ops_tests.log 2>&1; echo "ops tests rc=$?"; tail -4 $out/ops_tests.log
