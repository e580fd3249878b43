mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r03_tfinal.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -6 gpurun_out/r03_tfinal.log
[ $rc -eq 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_smoke.log 2>&1 || { tail -5 gpurun_out/r03_smoke.log; exit 1; }
tail -1 gpurun_out/r03_smoke.log
bash tools/profile_round.sh r03
