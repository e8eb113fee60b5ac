#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel trace of a bench.py command, top kernels by total time.
#   tools/kt_top.sh <tag> [bench.py args...]
set -o pipefail
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$out/kt" -o kt --output-format csv -- python3 bench.py "$@" --no-cpu-baseline > "$out/kt.log" 2>&1 || { echo "kernel trace failed"; tail -5 "$out/kt.log"; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, os, sys
csv.field_size_limit(1 << 30)
rows = []
for path in glob.glob(os.path.join(sys.argv[1], "kt", "**", "*kernel_stats.csv"), recursive=True):
    rows += list(csv.DictReader(open(path)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print(f'{r["Name"][:86]:86s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"]) / 1e3:9.2f} us  {100 * float(r["TotalDurationNs"]) / tot:5.1f} %')
PY
rm -rf "$out/kt"
