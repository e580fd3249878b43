#!/bin/bash
# rocprofv3 kernel durations of tools/bn1_bench.py (one-launch BatchNorm backward vs the three launches), per grid size
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/bn1_trace; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/bn1_trace -- python3 tools/bn1_bench.py > gpurun_out/bn1_trace.log 2>&1
python3 - <<'PY'
import csv, glob, re, collections
f = max(glob.glob("gpurun_out/bn1_trace/*/*kernel_trace.csv"))
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r.get("Grid_Size_Y", 1) or 1)) for r in csv.DictReader(open(f))))
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*$", "", n)[:40]
acc = collections.OrderedDict()
for s, e, n, gx, gy in rows:
    k = (short(n), gx, gy)
    if any(t in k[0] for t in ("col_reduce", "bn_bwd", "bn_act_bwd", "onepass")):
        acc.setdefault(k, []).append((e - s) / 1e3)
for k, v in acc.items():
    v = sorted(v)
    print("%-44s grid %5d x %3d  n %3d  median %7.2f us  min %7.2f" % (k[0], k[1], k[2], len(v), v[len(v) // 2], v[0]))
PY
