cd /root/repo
echo "== product"; python3 tools/layer_bench.py 64 128 bf16 20 "^G[0-5]|^D[0-3]|^E" 2>/dev/null | sed -e "s/|.*wgrad/| wgrad/"
for so in scratch/libvg_no_comp.so scratch/libvg_no_load.so; do
  echo "== $so"; VG_LIB_PATH=$PWD/$so python3 tools/layer_bench.py 64 128 bf16 20 "^G[1-5]|^D[0-3]" 2>/dev/null | sed -e "s/|.*wgrad/| wgrad/"
done
