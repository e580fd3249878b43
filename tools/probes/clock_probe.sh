#!/bin/bash
# in-kernel clock and phase times, diagnostic build (scratch/libvg_stamps.so):  tools/probes/clock_probe.sh [layers]
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
VG_LIB_PATH=$PWD/scratch/libvg_stamps.so timeout -k 10 300 python3 tools/probes/clock_probe.py 2 ${1:-G1,G2,G3,G4,D1,D2} 2>&1 | grep -v amdgpu.ids
