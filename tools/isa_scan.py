"""Load / wait pattern of every kernel in `hipcc -S --cuda-device-only` output (tools/isa_scan.sh makes it).
Per kernel, in program-text order:  nL>wK = n vector loads issued, then s_waitcnt vmcnt(K);  |B| = s_barrier;
{loop = a loop header.  A run of `1L>w0 1L>w0 ...` is a kernel that waits for every load before issuing the next
(loads that sit under branches: `v = inside ? p[i] : 0`), `8L>w7 0L>w6 ...` is the healthy shape.
    tools/isa_scan.py file.s [substring of the kernel names to show]"""
import re
import sys

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
name, out, pend = None, [], 0


def flush():
    if name and out and want in name:
        print(name[:110], " ".join(out[:48]))


for line in open(path):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        flush()
        name, out, pend = m.group(1), [], 0
        continue
    if name is None:
        continue
    if re.search(r"\b(global_load|buffer_load|flat_load)", line):
        pend += 1
    m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", line)
    if m:
        out.append(f"{pend}L>w{m.group(1)}")
        pend = 0
    if "s_barrier" in line:
        out.append("|B|")
    if re.match(r"^\.LBB.*Loop Header", line):
        out.append("{loop")
flush()
