cd /root/repo
for m in 0 1 2; do
echo "== VG_GG_NMAJOR=$m"; VG_GG_NMAJOR=$m python3 tools/layer_bench.py 64 128 bf16 20 "^G[0-3]|^D[2-3]|^E[2-4]" 2>/dev/null | sed -e "s/| wgrad.*//" -e 's/GF.*| fprop/| fprop/' | cut -c1-170
done
python -m pytest tests/test_gpu_kernels.py -x -q -k "gather or conv" 2>&1 | tail -2
