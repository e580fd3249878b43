cd /root/repo
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "stationary" 2>&1 | tail -5
for m in 1 0; do
echo "== VG_GG_STATIONARY=$m"; VG_GG_STATIONARY=$m timeout -k 10 200 python3 tools/layer_bench.py 64 128 bf16 20 "^G4|^D1" 2>/dev/null | sed -e "s/| wgrad.*//" -e 's/GF.*| fprop/| fprop/' | cut -c1-170
done
