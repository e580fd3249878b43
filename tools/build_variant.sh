#!/bin/bash
# tools/build_variant.sh NAME "-DFLAGS": a full alternative libvaegan_hip.so (all objects rebuilt with FLAGS) -> scratch/libvg_NAME.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); C="$ROOT/vae-gan-based-model-for-image-generation-and-denoising_amd/csrc"
NAME=$1; FLAGS=$2; D="$ROOT/scratch/var_$NAME"; mkdir -p "$D"
for f in conv_gemm wgrad bn_act pointwise pack_adam edge_conv; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$ROOT/include" -I"$C" -Wno-unused-function $FLAGS -c "$C/$f.hip" -o "$D/$f.o" &
done; wait
hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/scratch/libvg_$NAME.so" "$D"/*.o; echo built scratch/libvg_$NAME.so
