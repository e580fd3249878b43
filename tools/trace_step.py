"""Reduce a rocprofv3 --kernel-trace CSV of `bench.py --graph 0` to the dispatch sequence of ONE training step.

    python tools/trace_step.py <dir with *_kernel_trace.csv> <steps in the trace> [out.csv]

Prints, for the last step in the trace: index, kernel (short name), grid, workgroup, LDS, VGPRs, duration (us),
and the per-kernel-family totals per step.
"""
import csv, glob, os, re, sys, collections

def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*$", "", n)
    return n[:60]

def main():
    d, nsteps = sys.argv[1], int(sys.argv[2])
    fs = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in fs:
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    n = len(rows)
    # steps are identical sequences: find the period from the tail
    names = [r["Kernel_Name"] for r in rows]
    per = None
    for p in range(50, n // max(nsteps, 1) + 50):
        if names[n - p:] == names[n - 2 * p:n - p]:
            per = p
            break
    if per is None:
        per = n // nsteps
    last = rows[n - per:]
    t0 = int(last[0]["Start_Timestamp"]); t1 = int(last[-1]["End_Timestamp"])
    out = []
    fam = collections.OrderedDict()
    for i, r in enumerate(last):
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        nm = short(r["Kernel_Name"])
        grid = "x".join(str(int(r[k]) // max(int(r[w]), 1)) for k, w in (("Grid_Size_X", "Workgroup_Size_X"), ("Grid_Size_Y", "Workgroup_Size_Y"), ("Grid_Size_Z", "Workgroup_Size_Z")))
        out.append((i, nm, grid, r["Workgroup_Size_X"], r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""), dur))
        a = fam.setdefault(nm, [0, 0.0]); a[0] += 1; a[1] += dur
    busy = sum(o[-1] for o in out)
    for o in out:
        print("%4d %-60s grid %-14s wg %-4s lds %-7s vgpr %-4s %8.2f us" % o)
    print("---- per step: %d dispatches, kernel time %.1f us, span %.1f us" % (per, busy, (t1 - t0) / 1e3))
    for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print("%-60s %4d %9.1f us %5.1f%%" % (k, c, t, 100 * t / busy))
    if len(sys.argv) > 3:
        with open(sys.argv[3], "w") as f:
            w = csv.writer(f); w.writerow(["idx", "kernel", "grid", "wg", "lds", "vgpr", "us"])
            w.writerows(out)

main()
