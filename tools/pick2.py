"""Filter: like pick.py, the fused-lookup fields of a bench.py line."""
import sys, json
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    print({k: d.get(k) for k in ("value", "lookup_us", "fused_lookup_convc1_us", "unfused_lookup_then_convc1_us", "lookup_16bit_out_us", "refine_loop_ms", "refine_loop_hip_graph_ms")})
