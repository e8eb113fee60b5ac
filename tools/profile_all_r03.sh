#!/bin/bash
# Run ON THE GPU BOX: the round's committed profiles (kernel-trace stats + PMC passes per workload) into gpurun_out/;
# copy the *_kernel_stats.csv, *_pmc.json and *.passes.txt files into profiles/r03/ afterwards.
set -o pipefail
bash tools/profile_and_sum.sh kitti_b8 || exit 1
bash tools/profile_and_sum.sh kitti_b64 --batch 64 || exit 1
bash tools/profile_and_sum.sh sceneflow_train_b8 --workload sceneflow_train || exit 1
bash tools/profile_and_sum.sh middlebury_alt_full_b1 --workload middlebury_alt_full || exit 1
bash tools/profile_and_sum.sh kitti_f32_b8 --dtype f32 || exit 1
bash tools/profile_and_sum.sh realtime_b1 --workload realtime || exit 1
