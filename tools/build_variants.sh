#!/bin/bash
# Builds one library per set of -D switches into tools/bin/ (A/B runs with tools/ab_sustained.py):
#   tools/build_variants.sh name1 "-DAA_X=1" name2 "-DAA_Y=2 -DAA_Z=0" ...
# aa_kernels.hip and aa_kernels_mfcc.hip are rebuilt per variant; the other objects come from the shipped build.
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
    echo "$NAME: $(grep -A12 "${SHOW:-wave1024x2_kernelILi1}" $REPO/tools/bin/k_$NAME.remarks | grep -E " VGPRs:|ScratchSize" | sed 's/.*remark: *//; s/ \[-R.*//' | tr '\n' ' ')" &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $REPO/tools/bin/lib_$NAME.so $REPO/tools/bin/k_$NAME.o \
        $REPO/tools/bin/km_$NAME.o $C/build/aa_mel.o $C/build/aa_abi.o $C/build/aa_theory.o $C/build/aa_comm.o -ldl && echo built lib_$NAME.so ) &
done
wait
