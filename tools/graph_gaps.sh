cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/gaps; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaps -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-paths > gpurun_out/gaps.log 2>&1
python3 - <<'PY'
import csv, glob
f = max(glob.glob("gpurun_out/gaps/*/*kernel_trace.csv"))
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
# steady state: last 20 steps ~ find rng_advance kernels as step markers
marks = [i for i, r in enumerate(rows) if "rng_advance" in r[2] or "step_prologue" in r[2]]
marks = marks[-20:]
steps = []
for a, b in zip(marks[:-1], marks[1:]):
    seg = rows[a:b]
    span = (rows[b][0] - seg[0][0]) / 1e3
    busy = 0.0; end = seg[0][0]
    for s, e, _ in seg:
        if e > end:
            busy += (e - max(s, end)) / 1e3; end = e
    gaps = sorted(((seg[i + 1][0] - max(x[1] for x in seg[: i + 1])) / 1e3, seg[i][2][:40], seg[i + 1][2][:40]) for i in range(len(seg) - 1))
    steps.append((span, busy, len(seg), gaps[-3:]))
for s in steps[-5:]:
    print(f"step span {s[0]:8.1f} us  busy {s[1]:8.1f} us  idle {s[0]-s[1]:7.1f} us  kernels {s[2]}  largest gaps {[(round(g[0],1), g[1], g[2]) for g in s[3]]}")
PY
