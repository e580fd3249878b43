mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r03_t4.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -8 gpurun_out/r03_t4.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-paths --steps 200 --warmup 20 > gpurun_out/r03_bench4.json 2> gpurun_out/r03_bench4.err || { tail -5 gpurun_out/r03_bench4.err; exit 1; }
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r03_bench4.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline_bn']['us_per_step'], j['roofline_bn']['launches_per_step'])
PY
