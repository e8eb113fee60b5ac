#!/bin/bash
# Run ON THE GPU BOX (through gpurun): one kernel-trace pass of a bench.py command, then the average
# durations of this library's kernels.   tools/kt_quick.sh <tag> [bench.py args...]
set -o pipefail
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d "$out/kt" -o kt --output-format csv -- python3 bench.py "$@" --no-cpu-baseline > "$out/kt.log" 2>&1 || { echo "kernel trace failed"; tail -5 "$out/kt.log"; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, os, sys
csv.field_size_limit(1 << 30)
for path in glob.glob(os.path.join(sys.argv[1], "kt", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        n = r["Name"]
        if "_GLOBAL__N_" in n or "_kernel" in n and "at::" not in n:
            print(f'{n[:90]:90s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:9.2f} us  min {float(r["MinNs"]) / 1e3:9.2f} us')
PY
