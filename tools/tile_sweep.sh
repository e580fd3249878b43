#!/bin/bash
# per-layer sweep of the tile-picker thresholds (kernel-only times from tools/layer_bench.py)
cd "$(dirname "$0")/.."
for mw in 128 256 384 512 768; do
  for p256 in 128 256 1000000; do
    echo "== VG_TILE_MIN_WGS=$mw VG_PATCH256_MIN=$p256"
    VG_TILE_MIN_WGS=$mw VG_PATCH256_MIN=$p256 python3 tools/layer_bench.py 64 128 bf16 15 '^G[1-4]|^D[1-3]' 2>/dev/null | sed -e 's/ | wgrad.*//' | cut -c1-6,40-200
  done
done
