#!/bin/bash
# lookup output stores: always non-temporal (tools/lab_libs/lk_nt) against always agent-scope (lk_sc1), by batch,
# for fp32 taps (lookup_us) and 16-bit taps (lookup_16bit_out_us); run ON THE GPU BOX
# the two libraries (built in the container, from raft-stereo_amd/csrc after `make`):
#   for v in "0:nt" "(1ll<<60):sc1"; do val=${v%%:*}; name=${v##*:}; mkdir -p ../../tools/lab_libs/lk_$name
#     hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "-DRC_LAB_LK_NT_BYTES=$val" -c lookup.hip -o /tmp/lab_lk_$name.o
#     hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/lab_libs/lk_$name/libraftcorr_hip.so \
#           $(ls *.o | grep -v '^lookup.o$') /tmp/lab_lk_$name.o; done
set -o pipefail
for b in 16 24 32 40 48 64; do
  for rep in 1 2; do
    for lib in lk_sc1 lk_nt; do
      RAFTCORR_LIB=tools/lab_libs/$lib/libraftcorr_hip.so timeout -k 10 200 python bench.py --profile --batch $b --steps 10 > gpurun_out/nt.log 2>&1 || { tail -5 gpurun_out/nt.log; exit 1; }
      echo "batch $b $lib $(grep -o '"lookup_us": [0-9.]*' gpurun_out/nt.log) $(grep -o '"lookup_16bit_out_us": [0-9.]*' gpurun_out/nt.log)"
    done
  done
done
