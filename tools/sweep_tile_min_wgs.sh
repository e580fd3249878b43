mkdir -p gpurun_out
for cfg in "128 64" "256 32"; do
  set -- $cfg
  for v in 128 256 384 512; do
    VG_TILE_MIN_WGS=$v timeout -k 10 200 python bench.py --size $1 --batch $2 --steps 100 --warmup 10 --no-cpu-baseline --no-extra-paths > gpurun_out/sw_$1_$v.json 2>/dev/null || exit 1
    python - <<PY
import json
d=json.loads(open("gpurun_out/sw_$1_$v.json").read().strip().splitlines()[-1])
print("S=$1 B=$2 MIN_WGS=$v:", d["value"], d["ms_per_step"], "gg frac", d["roofline"]["frac"], flush=True)
PY
  done
done
