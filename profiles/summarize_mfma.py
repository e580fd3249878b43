#!/usr/bin/env python3
"""rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass of `bench.py --graph 0` -> profiles/<tag>_mfma_busy.json

  python profiles/summarize_mfma.py r03_bf16 gpurun_out/r03_bf16_mfma

MFMA utilisation as north_star words it ("MFMA utilisation ... from rocprof"), per kernel family and for the whole step:

    mfma_busy_frac = sum SQ_VALU_MFMA_BUSY_CYCLES / (sum GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)

SQ_VALU_MFMA_BUSY_CYCLES counts cycles in which a SIMD's matrix pipe is busy, summed over the chip's SIMDs (16 per
v_mfma_f32_16x16x32_bf16, MI355X_MICROARCH.md cycle constants); GRBM_GUI_ACTIVE is reported as the sum over the 8 XCDs
(same guide, DVFS give-back), so /8 is the dispatch's duration in shader cycles.  This is rocprofv3's own MfmaUtil formula
(100 * SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * CU_NUM * 4), derived_counters.xml of gfx94x) with the XCD sum taken
out.  It is a fraction of the CLOCK-CYCLE peak at the clock the chip actually held, so it reads higher than
roofline.frac, which is priced against 2.5 PFLOP/s = the peak at 2.4 GHz.
Cross-check written next to it: cycles the ALGORITHMIC FLOP of the family need at 1024 FLOP per cycle and SIMD
(16x16x32 bf16: 16384 FLOP in 16 cycles), from bench.py's own FLOP accounting when a bench line is given.
"""
import collections
import csv
import re
import glob
import json
import os
import sys


def fam_of(name):
    if "gg_kernel" in name or "ggp_kernel" in name or "ggq_kernel" in name:
        return "gather_gemm"
    if "tnconv_kernel" in name or "ggn_kernel" in name:
        return "edge"
    if "wgrad_" in name and "reduce" not in name and "dot_wgrad" not in name:
        return "wgrad"
    return "other"


def main():
    tag, d = sys.argv[1], sys.argv[2]
    here = os.path.dirname(os.path.abspath(__file__))
    f = max(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        fam = fam_of(r["Kernel_Name"])
        acc[fam][r["Counter_Name"]] += float(r["Counter_Value"])
        m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", r["Kernel_Name"])
        per_kernel[m.group(1) if m else r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[fam].add(r["Dispatch_Id"])
    out = {"source": os.path.relpath(f, os.path.dirname(here)), "formula": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 * 4)",
           "note": "GRBM_GUI_ACTIVE / 8 reads HIGH on dispatches shorter than ~0.3 ms (MI355X_MICROARCH.md, DVFS give-back): "
                   "27 us launches show 84k 'cycles' = 3.1 GHz, so this fraction reads LOW by about the dispatch ramp; "
                   "busy_cycles_over_alg_cycles is the exact part: matrix-pipe busy cycles per algorithmic FLOP cycle",
           "families": {}, "kernels": {}}
    tot = collections.defaultdict(float)
    for fam, c in acc.items():
        for k, v in c.items():
            tot[k] += v
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        out["families"][fam] = {"launches": len(disp[fam]), "mfma_busy_cycles": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0),
                                "gui_active_cycles_sum_xcd": gui,
                                "mfma_busy_frac": (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * 256 * 4)) if gui else None}
    for k, c in sorted(per_kernel.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0))[:12]:
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        out["kernels"][k] = {"mfma_busy_frac": (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * 256 * 4)) if gui else None,
                             "mfma_busy_cycles": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)}
    gui = tot.get("GRBM_GUI_ACTIVE", 0.0)
    out["whole_run"] = {"mfma_busy_frac": tot.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * 256 * 4) if gui else None}
    if len(sys.argv) > 3:                        # bench line of the same build: algorithmic FLOP per launch of the family
        b = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
        r = b["roofline"]
        flop = r["alg_gflop_per_launch"] * 1e9 * out["families"]["gather_gemm"]["launches"]
        # FLOP per cycle and SIMD: v_mfma_f32_16x16x32_bf16 = 16 384 FLOP in 16 cycles; v_mfma_f32_16x16x4_f32 = 2 048 in 32
        per_cycle = 64.0 if "fp32" in tag else 1024.0
        out["flop_per_cycle_and_simd"] = per_cycle
        out["families"]["gather_gemm"]["alg_flop_cycles"] = flop / per_cycle
        out["families"]["gather_gemm"]["busy_cycles_over_alg_cycles"] = out["families"]["gather_gemm"]["mfma_busy_cycles"] / (flop / per_cycle)
    json.dump(out, open(os.path.join(here, f"{tag}_mfma_busy.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
