#!/bin/bash
# run tools/layer_bench.py on the product library and on every scratch/libvg_*.so ablation build
cd "$(dirname "$0")/.."
F=${1:-'^G[1-4]|^D[1-3]'}
echo "== product"; python3 tools/layer_bench.py 64 128 bf16 20 "$F" 2>/dev/null | cut -c1-22,40-160
for so in scratch/libvg_*.so; do
  echo "== $so"; VG_LIB_PATH=$PWD/$so python3 tools/layer_bench.py 64 128 bf16 20 "$F" 2>/dev/null | cut -c1-22,40-160
done
