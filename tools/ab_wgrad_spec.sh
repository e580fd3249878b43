cd /root/repo
for m in 2 1; do VG_WG_SPEC=$m timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "wgrad or conv2d or conv_transpose or workgroup" 2>&1 | tail -1; done
for m in 2 1 0; do
echo "== VG_WG_SPEC=$m"; VG_WG_SPEC=$m timeout -k 10 200 python3 tools/layer_bench.py 64 128 bf16 20 "^G[1-4]|^D[1-3]|^E[1-3]" 2>/dev/null | sed -e "s/|.*wgrad/| wgrad/" | cut -c1-130
done
