#!/usr/bin/env python3
"""Condense the rocprofv3 passes written by tools/profile_gpu.sh into one JSON:
per kernel (short name) the average duration and the per-launch average of every counter.

    python tools/pmc_summarize.py gpurun_out/prof_<tag> profiles/r01/<name>.json
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


OURS = ("lookup_fwd_kernel_vt", "lookup_convc1_kernel", "convc1_pack_kernel", "adjoint_dense_kernel", "lookup_fwd_kernel", "lookup_bwd_kernel", "lookup_adjoint_kernel", "build16g_kernel", "build16_kernel", "build_kernel", "unskew_fold_kernel",
        "gemm_nt_kernel", "gemm_rows_kernel", "alt_fwd_kernel", "to_fragments_kernel", "to_fragments16_kernel", "sampler_fwd_kernel", "sampler_bwd_kernel",
        "naive_corr_kernel", "naive_pool_kernel", "gru_gate_r_kernel", "gru_update_kernel", "gru_gate_r_bwd_kernel",
        "gru_update_bwd_kernel", "upsample_kernel", "upsample_bwd_kernel", "upsample_bwd_flow_kernel",
        "convc1_bwd_taps_mfma_kernel", "convc1_bwd_weight_mfma_kernel", "convc1_bwd_taps_kernel",
        "convc1_bwd_weight_kernel", "convc1_bwd_reduce_kernel", "seqloss_partial_kernel", "seqloss_final_kernel",
        "seqloss_bwd_kernel", "pool2x_kernel", "interp_kernel", "convc1_pack_bwd_kernel")


def short(name):
    """Base kernel name without template arguments (one workload runs one instantiation of each; the lookup
    with 16-bit taps, rc_lookup_fwd_vt, runs beside the fp32 one and is kept apart as lookup_fwd_kernel_vt)."""
    if "lookup_fwd_kernel" in name and not re.search(r"fEEv|float>\s*\(", name):
        return "lookup_fwd_kernel_vt"
    for k in OURS:
        if k in name:
            return k
    m = re.search(r"(\w+_kernel)", name)
    return m.group(1) if m else name[:60]


def main(src, dst):
    out = defaultdict(dict)
    for path in glob.glob(os.path.join(src, "*", "**", "*_kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            k = short(row["Name"])
            out[k]["calls"] = int(row["Calls"])
            out[k]["avg_ns"] = float(row["AverageNs"])
            out[k]["min_ns"] = float(row["MinNs"])
            out[k]["max_ns"] = float(row["MaxNs"])
    for path in glob.glob(os.path.join(src, "*", "**", "*_counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(path)):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
            out[short(row["Kernel_Name"])].setdefault("vgpr", int(row["VGPR_Count"]))
            out[short(row["Kernel_Name"])].setdefault("agpr", int(row["Accum_VGPR_Count"]))
            out[short(row["Kernel_Name"])].setdefault("sgpr", int(row["SGPR_Count"]))
            out[short(row["Kernel_Name"])].setdefault("lds", int(row["LDS_Block_Size"]))
        for k, ctrs in acc.items():
            for c, vals in ctrs.items():
                out[k][c] = sum(vals) / len(vals)
    keep = {k: v for k, v in out.items() if k in OURS}
    for k, v in keep.items():
        # HBM-side traffic per launch, corrected as MI355X_MICROARCH.md (HBM section) prescribes:
        # FETCH_SIZE (KiB units of 64-B requests) reads half of a wide coalesced read on gfx950.
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            v["traffic_bytes_raw"] = (v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0
            v["traffic_bytes_corrected"] = (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0
        # gfx950's size-aware read counter (a 128-byte request counts 4): the bytes read from memory without any
        # correction; tools/lab_sector.hip shows it at 4 per line whatever part of a line was asked for
        if "TCC_EA0_RDREQ_DRAM_32B_sum" in v:
            v["read_bytes_dram_32b"] = 32.0 * v["TCC_EA0_RDREQ_DRAM_32B_sum"]
    json.dump(keep, open(dst, "w"), indent=1, sort_keys=True)
    for k, v in sorted(keep.items()):
        print(k, json.dumps({a: (round(b, 1) if isinstance(b, float) else b) for a, b in sorted(v.items())}))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
