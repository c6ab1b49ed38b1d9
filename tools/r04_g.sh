#!/bin/bash
# GPU box: cache-blocking experiment (FUSG_VU_SUBBATCH), ring-9 on the small kernel (FUSG_UP2_RING9), copy attribution
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r04g
mkdir -p $out
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > $out/ops_tests.log 2>&1; echo "ops tests rc=$?"; tail -6 $out/ops_tests.log
FUSG_VU_SUBBATCH=1 timeout -k 10 600 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -k "vunet and not full_size" > $out/vunet_tests_sub1.log 2>&1; echo "vunet tests (subbatch 1) rc=$?"; tail -3 $out/vunet_tests_sub1.log
FUSG_UP2_RING9=1 timeout -k 10 600 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -k "icn and not full_size" > $out/icn_tests_ring9.log 2>&1; echo "icn tests (ring9) rc=$?"; tail -3 $out/icn_tests_ring9.log
for arm in "base" "FUSG_VU_SUBBATCH=8" "FUSG_VU_SUBBATCH=4" "FUSG_UP2_RING9=1" "base" "FUSG_VU_SUBBATCH=8" "FUSG_UP2_RING9=1"; do
  if [ "$arm" = base ]; then e=""; else e="$arm"; fi
  env $e timeout -k 10 300 python bench.py --precision f16x3 --no-cpu-baseline --no-clip --steps 20 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$arm', d['value'], r['frac'], r['conv_ms_per_step'], r['launches_per_step'])"
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --hip-trace --memory-copy-trace --kernel-trace --stats --output-format csv -d /tmp/prof_r04g -o onepass -- python3 $R/tools/one_pass.py 32 10 > $out/onepass.log 2>&1; echo "rocprof rc=$?"
for f in $(find /tmp/prof_r04g -name "*hip_api_stats.csv") $(find /tmp/prof_r04g -name "*memory_copy_stats.csv"); do echo "== $f"; head -25 "$f" | cut -c1-180; cp "$f" $out/; done
f=$(find /tmp/prof_r04g -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then grep -i "copyBuffer\|fillBuffer" "$f" | cut -c1-200; fi
f=$(find /tmp/prof_r04g -name "*memory_copy_trace.csv" | head -1)
if [ -n "$f" ]; then echo "copies: $(wc -l < "$f")"; head -2 "$f"; python3 -c "
import csv, sys, collections
rows = list(csv.DictReader(open('$f')))
c = collections.Counter()
for r in rows:
    c[(r.get('Direction', '?'), r.get('Size', r.get('Bytes', '?')))] += 1
for k, v in c.most_common(15):
    print(v, k)
"; fi
