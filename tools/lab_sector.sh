#!/bin/bash
# Run ON THE GPU BOX: tools/lab_sector (partial-line reads), alone and under two counter passes.
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/lab_sector
mkdir -p $out
timeout -k 10 120 tools/lab_sector > $out/timing.txt 2>&1 || { tail -5 $out/timing.txt; exit 1; }
cat $out/timing.txt
for pass in "TCC_EA0_RDREQ_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_DRAM_32B_sum TCC_REQ_sum TCC_MISS_sum" "FETCH_SIZE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -s KILL 120 rocprofv3 --pmc $pass -d $out/$tag -o p --output-format csv -- tools/lab_sector > $out/$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 $out/$tag.log; exit 1; }
done
python3 - $out <<'PY'
import csv, glob, os, sys, collections
csv.field_size_limit(1 << 30)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if "read_part" not in k:
        continue
    # dispatches of one kernel: the counter rows come per dispatch (7 launches each); average them
    print(k[:60], {c: round(sum(v) / len(v)) for c, v in sorted(acc[k].items())})
PY
rm -rf $out/TCC_* $out/FETCH_SIZE
