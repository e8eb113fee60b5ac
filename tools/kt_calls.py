"""Durations of every call of the kernels whose name contains <pattern>, in launch order, from a rocprofv3
kernel trace (tools/kt_quick.sh writes one under gpurun_out/prof_<tag>/kt):  tools/kt_calls.py <dir> <pattern>"""
import csv
import glob
import os
import sys

csv.field_size_limit(1 << 30)
root, pat = sys.argv[1], sys.argv[2]
for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    prev_end = None
    out = []
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if pat in r["Kernel_Name"]:
            out.append(((e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0.0))
        prev_end = e
    print(" ".join(f"{d:.0f}({g:.0f})" for d, g in out))
    print("us per call (gap since the previous kernel ended)")
