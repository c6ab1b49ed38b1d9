#!/bin/bash
# GPU box: A/B/A of one environment switch on the f16x3 leg of bench.py:  tools/ab_bench_env.sh NAME=VALUE [extra bench args]
sw=$1; shift
for cfg in "" "$sw" "" "$sw"; do
  env $cfg python bench.py --precision f16x3 --no-cpu-baseline --no-clip "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); v=d['precision_legs']['f16x3']; print('[$cfg]', v['value'], 'crops/s', v['ms_per_step'], 'ms/step; conv', v['roofline']['conv_ms_per_step'], 'ms, frac', v['roofline']['frac'])"
done
