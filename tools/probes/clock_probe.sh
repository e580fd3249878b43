#!/bin/bash
# in-kernel clock and phase times, diagnostic build (scratch/libvg_stamps.so)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
VG_LIB_PATH=$PWD/scratch/libvg_stamps.so timeout -k 10 300 python3 tools/probes/clock_probe.py 2 2>&1 | grep -v amdgpu.ids
