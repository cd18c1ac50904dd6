#!/bin/bash
# Design aid (not product): builds diagnostic copies of the library with parts of the fused mel
# reduction removed (AA_MEL_ABL bits, see aa_kernels.hip) into tools/bin/, and times the log-mel
# geometries with each.  Build here, run the python part on the GPU box:
#   tools/ablate_mel.sh build ; gpurun -- tools/ablate_mel.sh run
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
    mkdir -p tools/bin
    for abl in 0 1 2 4 8 15; do
        hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -fno-strict-aliasing -DAA_MEL_ABL=$abl \
            -I include -shared -o tools/bin/libaa_melabl_$abl.so \
            audio_analyzer_rs_amd/csrc/aa_kernels.hip audio_analyzer_rs_amd/csrc/aa_mel.hip audio_analyzer_rs_amd/csrc/aa_abi.cpp &
    done
    wait
else
    for abl in 0 1 2 4 8 15; do
        echo "== AA_MEL_ABL=$abl"
        AA_LIB_OVERRIDE=tools/bin/libaa_melabl_$abl.so python tools/bench_geometries.py logmel
    done
fi
