#!/bin/bash
# Per-kernel time per step with and without the BatchNorm-backward epilogue (VG_BNB=1 / 0): rocprofv3 kernel trace of a
# short bench run each, summed per kernel name over the last 10 steps.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for m in 0 1; do
  rm -rf gpurun_out/bnbtr_$m
  export VG_BNB=$m
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/bnbtr_$m -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-paths > gpurun_out/bnbtr_$m.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
res = {}
for m in (0, 1):
    f = max(glob.glob(f"gpurun_out/bnbtr_{m}/*/*kernel_trace.csv"))
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
    marks = [i for i, r in enumerate(rows) if "rng_advance" in r[2]][-11:]
    seg = rows[marks[0]:marks[-1]]
    n = len(marks) - 1
    acc = collections.defaultdict(lambda: [0.0, 0])
    for s, e, k in seg:
        acc[k][0] += (e - s) / 1e3 / n
        acc[k][1] += 1
    res[m] = acc
    print(f"VG_BNB={m}: span/step {(rows[marks[-1]][0] - rows[marks[0]][0]) / 1e3 / n:8.1f} us, kernels/step {len(seg) / n:.0f}")
keys = sorted(set(res[0]) | set(res[1]), key=lambda k: -(res[0].get(k, [0, 0])[0] + res[1].get(k, [0, 0])[0]))
for k in keys[:40]:
    a, b = res[0].get(k, [0, 0]), res[1].get(k, [0, 0])
    print(f"{a[0]:8.1f} us x{a[1] // 10:3d} | {b[0]:8.1f} us x{b[1] // 10:3d} | {b[0] - a[0]:+7.1f} | {k[:110]}")
PY
