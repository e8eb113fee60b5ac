#!/bin/bash
# Run ON THE GPU BOX: the fused adjoint's duration, call by call, in bench.py's sceneflow_train run for each lab
# build (tools/lab_adj.sh).  The first 25 calls are the timed step (22 gradient tensors, 365 MB, that nobody has
# touched since they were made, read while the caches still hold the forward's freshly written lookup outputs;
# with RC_BENCH_SHARED_GOUT=1 one tensor for all lookups, as the bench had it before), the later ones the
# fused-convc1 training legs (22 tensors that rc_convc1_bwd has just written).
set -o pipefail
for name in "$@"; do
  if [ "$name" = tree ]; then unset RAFTCORR_LIB; else export RAFTCORR_LIB=$PWD/tools/lab_adj_$name.so; fi
  echo "== $name"
  timeout -k 10 240 bash tools/kt_quick.sh adj_$name --workload sceneflow_train --steps 20 --warmup 5 > /dev/null || exit 1
  python3 tools/kt_calls.py gpurun_out/prof_adj_$name adjoint_dense | python3 -c "
import sys, re, statistics as st
v = [float(x.split('(')[0]) for x in sys.stdin.readline().split()]
print(f'timed step (calls 1-25): median {st.median(v[:25]):.1f} us   fused-convc1 legs (calls 26-): median {st.median(v[25:]):.1f} us, min {min(v[25:]):.0f}, max {max(v[25:]):.0f}')"
  rm -rf gpurun_out/prof_adj_$name
done
