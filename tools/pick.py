"""Filter: `python bench.py ... | python3 tools/pick.py` prints the few fields of the JSON line that an A/B run compares."""
import sys, json
for line in sys.stdin:
    line=line.strip()
    if not line.startswith("{"): continue
    d=json.loads(line)
    print({k:d.get(k) for k in ("value","ms_per_step","build_ms","lookup_us","build_gbps")}, d["roofline"].get("avg_launch_us"))
