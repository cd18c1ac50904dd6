#!/bin/bash
# Builds tools/bin/libaa_safe.so (every counted s_waitcnt a full wait, -DAA_SAFE_WAITS=1, and the LDS-DMA issued through
# the compiler builtin, -DAA_V_DMA_ASM=0, so that the compiler adds its own waits in front of LDS accesses) next to the
# shipped library; run tools/check_counted_waits.py on the GPU box afterwards:
#   bash tools/check_counted_waits.sh && gpurun -- 'python tools/check_counted_waits.py audio_analyzer_rs_amd/libaa_stft.so tools/bin/libaa_safe.so'
set -e
cd "$(dirname "$0")/../audio_analyzer_rs_amd/csrc"
make -s
mkdir -p ../../tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -fno-strict-aliasing -DAA_SAFE_WAITS=1 -DAA_V_DMA_ASM=0 -c -o build/aa_kernels_safe.o aa_kernels.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../tools/bin/libaa_safe.so build/aa_kernels_safe.o build/aa_kernels_mfcc.o build/aa_mel.o build/aa_abi.o build/aa_theory.o build/aa_comm.o -ldl
echo built tools/bin/libaa_safe.so
