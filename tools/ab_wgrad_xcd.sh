cd /root/repo
echo "== xcd on";  python3 tools/layer_bench.py 64 128 bf16 20 "^G[1-5]|^D[0-3]|^E[1-3]" 2>/dev/null | sed -e "s/|.*wgrad/| wgrad/"
echo "== xcd off"; VG_WG_XCD=0 python3 tools/layer_bench.py 64 128 bf16 20 "^G[1-5]|^D[0-3]|^E[1-3]" 2>/dev/null | sed -e "s/|.*wgrad/| wgrad/"
python -m pytest tests/test_gpu_kernels.py -x -q -k wgrad 2>&1 | tail -2
