mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r03_t5.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -8 gpurun_out/r03_t5.log
[ $rc -eq 0 ] || exit 1
bash tools/ab_tree.sh scratch/base 3
