#!/bin/bash
# Run ON THE GPU BOX: tools/profile_gpu.sh + tools/pmc_summarize.py, keeping only the summaries
# (kernel-trace stats CSV, per-kernel PMC JSON, the pass log) under gpurun_out/.
#   tools/profile_and_sum.sh <tag> [bench.py args...]
set -o pipefail
tag=$1
bash tools/profile_gpu.sh "$@" > gpurun_out/${tag}.passes.txt 2>&1 || { cat gpurun_out/${tag}.passes.txt; exit 1; }
python3 tools/pmc_summarize.py gpurun_out/prof_$tag gpurun_out/${tag}_pmc.json > gpurun_out/${tag}_sum.txt || exit 1
cp "$(find gpurun_out/prof_$tag/kt -name '*_kernel_stats.csv' | head -1)" gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/prof_$tag
