mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "graphed_reference or dropin or modules_losses" -s > gpurun_out/r03_t3.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/r03_t3.log
[ $rc -le 1 ] || exit $rc
timeout -k 10 600 python bench.py --no-cpu-baseline > gpurun_out/r03_bench3.json 2> gpurun_out/r03_bench3.err || { tail -5 gpurun_out/r03_bench3.err; exit 1; }
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r03_bench3.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['dropin_path'])
PY
