#!/bin/bash
# Builds tools/bin/libaa_safe.so next to the shipped library: BOTH kernel translation units (aa_kernels.hip and the fused
# MFCC kernel's aa_kernels_mfcc.hip, the latter with the Makefile's scheduler flags) with -DAA_SAFE_WAITS=1 - every counted
# s_waitcnt a full wait, and the LDS-DMA issued through the compiler builtin so that the compiler adds its own waits in
# front of LDS accesses.  Run tools/check_counted_waits.py on the GPU box afterwards:
#   bash tools/check_counted_waits.sh && gpurun -- 'python tools/check_counted_waits.py audio_analyzer_rs_amd/libaa_stft.so tools/bin/libaa_safe.so'
set -e
cd "$(dirname "$0")/../audio_analyzer_rs_amd/csrc"
make -s
mkdir -p ../../tools/bin
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -fno-strict-aliasing -DAA_SAFE_WAITS=1"
/opt/rocm/bin/hipcc $FLAGS -c -o build/aa_kernels_safe.o aa_kernels.hip &
/opt/rocm/bin/hipcc $FLAGS -mllvm -amdgpu-sched-strategy=max-ilp -c -o build/aa_kernels_mfcc_safe.o aa_kernels_mfcc.hip &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../tools/bin/libaa_safe.so build/aa_kernels_safe.o build/aa_kernels_mfcc_safe.o build/aa_mel.o build/aa_abi.o build/aa_theory.o build/aa_comm.o -ldl
echo built tools/bin/libaa_safe.so
