#!/bin/bash
# lab: time the fused lookup+convc1 kernel with parts switched off (library built with -DRC_LAB_FUSED)
set -e
mkdir -p gpurun_out
: > gpurun_out/fused_ablate.txt
for ab in ${ABS:-0 1 2 3 8 9 10 11}; do
  for bt in 8 64; do
    RC_LAB_FUSED=$ab timeout -k 10 120 python bench.py --profile --batch $bt > gpurun_out/ab.log 2>&1
    python - "$ab" "$bt" <<'PY' >> gpurun_out/fused_ablate.txt
import json, sys
for l in open('gpurun_out/ab.log'):
    if l.startswith('{'):
        j = json.loads(l)
        print('ablate', sys.argv[1], 'pairs', sys.argv[2], 'lookup_us', j['lookup_us'], 'fused_us', j['fused_lookup_convc1_us'])
PY
  done
done
