#!/bin/bash
# Timing-only ablation builds of libvaegan_hip.so (wrong results by construction; never the product library).
#   tools/build_ablate.sh NAME "-DFLAG1 -DFLAG2"   ->  scratch/libvg_NAME.so   (select with VG_LIB_PATH)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C="$ROOT/vae-gan-based-model-for-image-generation-and-denoising_amd/csrc"
NAME=$1; FLAGS=$2
mkdir -p "$ROOT/scratch/abl_$NAME"
for f in conv_gemm wgrad; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$ROOT/include" -I"$C" -Wno-unused-function $FLAGS -c "$C/$f.hip" -o "$ROOT/scratch/abl_$NAME/$f.o" &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/scratch/libvg_$NAME.so" "$ROOT/scratch/abl_$NAME/conv_gemm.o" "$ROOT/scratch/abl_$NAME/wgrad.o" "$C/bn_act.o" "$C/pointwise.o" "$C/pack_adam.o" "$C/edge_conv.o"
echo built scratch/libvg_$NAME.so
