#!/bin/bash
# One hipGraph-replayed training step as a launch sequence: start offset, duration, gap before, kernel (rocprofv3 kernel trace).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/seq; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/seq -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-paths $SEQ_ARGS > gpurun_out/seq.log 2>&1
python3 - <<'PY'
import csv, glob, re, collections
f = max(glob.glob("gpurun_out/seq/*/*kernel_trace.csv"))
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", ""), r.get("Workgroup_Size_X", "")) for r in csv.DictReader(open(f))))
marks = [i for i, r in enumerate(rows) if "rng_advance" in r[2] or "step_prologue" in r[2]]
a, b = marks[-3], marks[-2]
seg = rows[a:b]
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*$", "", n)[:48]
t0, end = seg[0][0], seg[0][0]
fam = collections.defaultdict(lambda: [0, 0.0])
with open("gpurun_out/seq_step.txt", "w") as o:
    for i, (s, e, n, g, w) in enumerate(seg):
        o.write(f"{i:4d} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {(s - end) / 1e3:6.1f}  {short(n)}  grid {g} wg {w}\n")
        end = max(end, e)
        fam[short(n)][0] += 1; fam[short(n)][1] += (e - s) / 1e3
    o.write(f"span {(rows[b][0] - t0) / 1e3:.1f} us, {len(seg)} kernels\n")
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        o.write(f"{v[1]:8.1f} us {v[0]:3d}  {k}\n")
print(open("gpurun_out/seq_step.txt").read()[-3000:])
PY
