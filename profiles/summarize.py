#!/usr/bin/env python3
"""Turn rocprofv3 outputs (gpurun_out/<run>/...) into the small, committed evidence files of a round.

  python profiles/summarize.py r01 gpurun_out/r01_bf16_trace gpurun_out/r01_bf16_fetch gpurun_out/r01_bf16_write

writes profiles/<tag>_kernel_stats.csv (verbatim rocprofv3 --stats summary), profiles/<tag>_pmc_traffic.json
(per-launch HBM-side bytes of the GEMM kernel families) and prints the family averages quoted in DESIGN.md.
Counter handling per /opt/skills/guides/MI355X_MICROARCH.md (HBM, rocprofv3 PMC): FETCH_SIZE and WRITE_SIZE are
collected in SEPARATE --pmc passes, are in KiB, and FETCH_SIZE is doubled on gfx950 (128-B requests tallied as 64 B).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

FAMILIES = {"gather_gemm": "gg_kernel", "wgrad": "wgrad_", "edge": "tnconv_kernel | ggn_kernel"}


def fam_of(name):
    if "gg_kernel" in name or "ggp_kernel" in name or "ggq_kernel" in name:   # gather-GEMM, its patch variant (conv_patch.hpp), the 4-phase narrow form (conv_phase4.hpp)
        return "gather_gemm"
    if "tnconv_kernel" in name or "ggn_kernel" in name:  # edge layers (edge_conv.hip, conv_narrowk.hpp): HBM-bound
        return "edge"
    if "wgrad_" in name and "reduce" not in name and "dot_wgrad" not in name:   # wgrad_kernel<..>, wgrad_bf16[_dma]_kernel, edge_wgrad_kernel
        return "wgrad"
    return None


def main():
    tag, trace_dir = sys.argv[1], sys.argv[2]
    here = os.path.dirname(os.path.abspath(__file__))
    stats = max(glob.glob(os.path.join(trace_dir, "*", "*kernel_stats.csv")), key=os.path.getmtime)
    shutil.copy(stats, os.path.join(here, f"{tag}_kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats)))
    total = sum(int(r["TotalDurationNs"]) for r in rows)
    out = {"kernel_time_ms_total": total / 1e6, "families": {}}
    for fam in FAMILIES:
        sel = [r for r in rows if fam_of(r["Name"]) == fam]
        calls = sum(int(r["Calls"]) for r in sel)
        ns = sum(int(r["TotalDurationNs"]) for r in sel)
        out["families"][fam] = {"launches": calls, "avg_us": ns / max(calls, 1) / 1e3, "share": ns / total}
    if len(sys.argv) >= 5:
        for key, d in (("fetch_kib", sys.argv[3]), ("write_kib", sys.argv[4])):
            f = max(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)
            acc = collections.defaultdict(lambda: [0, 0.0])
            for r in csv.DictReader(open(f)):
                fam = fam_of(r["Kernel_Name"])
                if fam:
                    acc[fam][0] += 1
                    acc[fam][1] += float(r["Counter_Value"])
            for fam, (n, v) in acc.items():
                out["families"][fam][key + "_per_launch"] = v / n
                out["families"][fam][key + "_launches"] = n
        for fam, v in out["families"].items():
            if "fetch_kib_per_launch" in v:
                v["hbm_bytes_per_launch"] = (2.0 * v["fetch_kib_per_launch"] + v["write_kib_per_launch"]) * 1024.0
    json.dump(out, open(os.path.join(here, f"{tag}_pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
