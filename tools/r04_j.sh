#!/bin/bash
# GPU box: small batches (the 8-vehicles-per-GPU shape of configs[3], the reference's batch 1): which switches help there?
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r04j
mkdir -p $out
cd $R
for arm in "base" "FUSG_BNECK_MINHW=16" "FUSG_BNECK_MINHW=32" "FUSG_SMALL_KSPLIT=1" "FUSG_BNECK_MINHW=16 FUSG_SMALL_KSPLIT=1" "base"; do
  if [ "$arm" = base ]; then e=""; else e="$arm"; fi
  env $e timeout -k 10 200 python tools/small_batch.py 1 2 4 8 16 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    d=json.loads(ln); print('$arm', 'B', d['batch'], 'replay ms', d['replay']['ms_per_pass'], 'crops/s', d['replay']['crops_per_s'], 'eager', d['eager']['crops_per_s'])"
done
