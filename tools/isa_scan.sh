#!/bin/bash
# Run HERE (no GPU needed): device assembly of every source of the library into /tmp/rc_isa/, then the load / wait
# pattern of the kernels whose name contains $1 (default: the bf16, radius-4 instantiations).
#   tools/isa_scan.sh [name substring]
set -e
cd "$(dirname "$0")/../raft-stereo_amd/csrc"
mkdir -p /tmp/rc_isa
for f in *.hip; do
  extra=""
  case $f in fused.hip|alt.hip) extra="-mllvm -amdgpu-mfma-vgpr-form";; esac
  [ /tmp/rc_isa/${f%.hip}.s -nt $f ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $extra -S --cuda-device-only -o /tmp/rc_isa/${f%.hip}.s $f 2>/dev/null &
done
wait
for f in /tmp/rc_isa/*.s; do python3 ../../tools/isa_scan.py $f "${1:-IDF16b}"; done
