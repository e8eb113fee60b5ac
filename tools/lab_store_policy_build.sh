#!/bin/bash
# lab: builds of the library whose lookup / fused-lookup output stores carry a fixed cache policy
# (tools/lab_libs/<name>/libraftcorr_hip.so; tools/lab_consumer.py times them with their consumers)
set -e
cd "$(dirname "$0")/../raft-stereo_amd/csrc"
make -j8 > /dev/null
build() {  # name, policy string
  mkdir -p ../../tools/lab_libs/$1
  for f in lookup fused; do
    extra=""; [ $f = fused ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra "-DRC_LAB_STORE_POLICY=\"$2\"" -c $f.hip -o /tmp/lab_$f.o
  done
  objs=$(ls *.o | grep -v -e '^lookup.o$' -e '^fused.o$')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/lab_libs/$1/libraftcorr_hip.so $objs /tmp/lab_lookup.o /tmp/lab_fused.o
}
build plain ""
build nt "nt"
build sc1 "sc1"
build sc0sc1 "sc0 sc1"
build sc1nt "sc1 nt"
