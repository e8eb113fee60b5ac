#!/bin/bash
# Lab builds of the library that differ in ONE source file (run HERE; the GPU box gets the .so files):
#   tools/lab_so.sh <file.hip> <name>=<-D flags> ...   e.g.  tools/lab_so.sh build.hip st0="-DRC_LAB_BUILD_STORE_AUX=0"
# writes tools/lab_adj_<name>.so (git-ignored); use it with RAFTCORR_LIB=$PWD/tools/lab_adj_<name>.so
set -e
cd "$(dirname "$0")/../raft-stereo_amd/csrc"
make -j8 > /dev/null
src=$1; shift
base=${src%.hip}
mkdir -p /tmp/lab_so
extra=""
case $src in fused.hip|alt.hip) extra="-mllvm -amdgpu-mfma-vgpr-form";; esac
others=$(ls *.o | grep -v "^$base.o$")
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra $flags -c $src -o /tmp/lab_so/${base}_$name.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/lab_adj_$name.so $others /tmp/lab_so/${base}_$name.o &&
    echo "built tools/lab_adj_$name.so" ) &
done
wait
