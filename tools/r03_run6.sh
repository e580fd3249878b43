mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "edge_layer" > gpurun_out/r03_t6.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -6 gpurun_out/r03_t6.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --size 256 --batch 32 --steps 50 --warmup 5 --no-cpu-baseline --no-extra-paths > gpurun_out/r03_s256_b.json 2>gpurun_out/r03_s256_b.err || { tail -5 gpurun_out/r03_s256_b.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03_s256_b.json").read().strip().splitlines()[-1])
print("S=256 B=32:", d["value"], d["ms_per_step"], "gg frac", d["roofline"]["frac"], d["roofline"]["wgrad"])
PY
