#!/bin/bash
# A/B of two builds of the library on bench.py's timed step (run ON THE GPU BOX): tools/lab_ab.sh <other .so> [bench args]
set -o pipefail
other=$1; shift
for rep in 1 2 3; do
  for which in tree other; do
    if [ $which = other ]; then export RAFTCORR_LIB=$other; else unset RAFTCORR_LIB; fi
    timeout -k 10 200 python bench.py --profile "$@" > gpurun_out/ab_$which.log 2>&1 || { tail -5 gpurun_out/ab_$which.log; exit 1; }
    echo "$which $* $(grep -o '"build_ms": [0-9.]*' gpurun_out/ab_$which.log) $(grep -o '"lookup_us": [0-9.]*' gpurun_out/ab_$which.log) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab_$which.log)"
  done
done
