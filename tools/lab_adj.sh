#!/bin/bash
# Lab builds of the library that differ in the fused adjoint only (run HERE; the GPU box gets the .so files):
#   tools/lab_adj.sh <name>=<-D flags> ...     e.g.  tools/lab_adj.sh a1="-DRC_LAB_ADJ_ABLATE=1" pd8="-DRC_LAB_ADJ_PD=8"
# writes tools/lab_adj_<name>.so (git-ignored); use it with RAFTCORR_LIB=tools/lab_adj_<name>.so
set -e
cd "$(dirname "$0")/../raft-stereo_amd/csrc"
mkdir -p /tmp/lab_adj
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c lookup.hip -o /tmp/lab_adj/lookup_$name.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/lab_adj_$name.so alt.o build.o build_bwd.o core.o fused.o fused_bwd.o /tmp/lab_adj/lookup_$name.o sampler.o update.o &&
    echo "built tools/lab_adj_$name.so" ) &
done
wait
