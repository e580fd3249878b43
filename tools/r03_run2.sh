# full GPU suite on the pruned build + calibration numbers
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_configs.py::test_bf16_state_after_first_iteration_vs_reference_checksums_S256 > gpurun_out/r03_t2.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -8 gpurun_out/r03_t2.log
[ $rc -le 1 ] || exit $rc
timeout -k 10 600 python tools/calibrate_r03.py > gpurun_out/r03_calib.log 2>&1 || { echo "calibration failed"; tail -5 gpurun_out/r03_calib.log; exit 1; }
head -4 gpurun_out/r03_calib.log
