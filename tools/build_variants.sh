#!/bin/bash
# Builds one library per set of -D switches into tools/bin/ (A/B runs with tools/ab_sustained.py):
#   tools/build_variants.sh name1 "-DAA_X=1" name2 "-DAA_Y=2 -DAA_Z=0" ...
# aa_kernels.hip, aa_kernels_mfcc.hip and aa_abi.cpp are rebuilt per variant (so that -DAA_EXPERIMENTS=1 also switches the
# plan's path selectors on: AA_MEL_GEMM, AA_MEL_MFMA, AA_MFCC_NO_FUSE, AA_SMALL_PLAIN_MAP, and the launchers' AA_GRID_* /
# AA_LDS_PAD / AA_W2048_RING); the other objects come from the shipped build.
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
C=$REPO/audio_analyzer_rs_amd/csrc
make -C $C -s
mkdir -p $REPO/tools/bin
FLAGS="-O3 -std=c++17 -fPIC -fno-slp-vectorize -fno-strict-aliasing -Wall -Wextra -Wno-unused-parameter"
while [ $# -ge 2 ]; do
  NAME=$1; DEFS=$2; shift 2
  # (the fused MFCC kernels live in a translation unit of their own, built with the max-ILP machine scheduler: csrc/Makefile)
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS $DEFS -Rpass-analysis=kernel-resource-usage -c -o $REPO/tools/bin/k_$NAME.o $C/aa_kernels.hip 2> $REPO/tools/bin/k_$NAME.remarks &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS $DEFS -mllvm -amdgpu-sched-strategy=max-ilp -Rpass-analysis=kernel-resource-usage -c -o $REPO/tools/bin/km_$NAME.o $C/aa_kernels_mfcc.hip 2>> $REPO/tools/bin/k_$NAME.remarks &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS $DEFS -c -o $REPO/tools/bin/abi_$NAME.o $C/aa_abi.cpp &&
    echo "$NAME: $(grep -A12 "${SHOW:-wave1024_kernelILi1ELb1}" $REPO/tools/bin/k_$NAME.remarks | grep -E " VGPRs:|ScratchSize" | sed 's/.*remark: *//; s/ \[-R.*//' | tr '\n' ' ')" &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $REPO/tools/bin/lib_$NAME.so $REPO/tools/bin/k_$NAME.o \
        $REPO/tools/bin/km_$NAME.o $C/build/aa_mel.o $REPO/tools/bin/abi_$NAME.o $C/build/aa_theory.o $C/build/aa_comm.o -ldl && echo built lib_$NAME.so ) &
done
wait
