# first GPU call of round 3: new config-size tests (printouts), calibration numbers, baseline bench line of this box
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_configs.py tests/test_gpu_siblings.py tests/test_gpu_parity.py -k "configs or reseed or teacher or continues or benchmarked" -s -q > gpurun_out/r03_t1.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -5 gpurun_out/r03_t1.log
[ $rc -le 1 ] || exit $rc
timeout -k 10 600 python tools/calibrate_r03.py > gpurun_out/r03_calib.log 2>&1 || { echo "calibration failed"; tail -5 gpurun_out/r03_calib.log; exit 1; }
timeout -k 10 400 python bench.py --no-extra-paths > gpurun_out/r03_bench0.json 2> gpurun_out/r03_bench0.err || { echo "bench failed"; tail -5 gpurun_out/r03_bench0.err; exit 1; }
head -c 1200 gpurun_out/r03_bench0.json
