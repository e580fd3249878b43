cd /root/repo
for so in "" scratch/libvg_wg_noepi.so scratch/libvg_wg_epionly.so scratch/libvg_wg_loop_only.so; do
  echo "== ${so:-product}"; VG_LIB_PATH=${so:+$PWD/$so} python3 tools/layer_bench.py 64 128 bf16 20 "^G[1-4]|^D[1-3]" 2>/dev/null | sed -e "s/|.*wgrad/| wgrad/" | cut -c1-120
done
