#!/bin/bash
# Per-kernel register / scratch / LDS / occupancy table from the compiler (runs without a GPU).
cd "$(dirname "$0")/../raft-stereo_amd/csrc" || exit 1
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c --cuda-device-only \
      -Rpass-analysis=kernel-resource-usage -o /dev/null "$f" 2>&1 |
    grep -E "Function Name|SGPRs:|VGPRs:|ScratchSize|Occupancy|LDS Size" |
    sed -e 's/ \[-Rpass-analysis=kernel-resource-usage\]//' -e 's/.*remark: *//' |
    awk -v f="$f" '/Function Name/ {if (n) print line; n=1; cmd="echo " $3 " | c++filt 2>/dev/null"; cmd | getline d; close(cmd); line=f " | " substr(d,1,90) " |"; next} {line=line " " $0 ";"} END {print line}'
done
