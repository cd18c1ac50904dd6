#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of every geometry of tools/bench_geometries.py.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof/geoms
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $REPO/tools/bench_geometries.py > $OUT/kt.log 2>&1
grep -v amdgpu.ids $OUT/kt.log | tail -14
cat $OUT/kt/kt_kernel_stats.csv
