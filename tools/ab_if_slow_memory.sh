#!/bin/bash
# Runs on the GPU box: the boxes come in two classes (complex rows at 1024/256: ~90 us or ~113 us).  Times the
# shipped library once; on a slow-memory box runs the store-policy A/B for complex and magnitudes, else only a short one.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
T=$(AB_OUT=complex python tools/ab_sustained.py tools/bin/libaa_new.so 1 2>/dev/null | awk '{print $(NF)}')
echo "complex 1024/256 on this box: $T us"
R=1; if python -c "import sys; sys.exit(0 if float('$T') > 105 else 1)"; then R=3; echo "slow-memory class: full A/B"; fi
AB_OUT=complex python tools/ab_sustained.py tools/bin/libaa_new.so tools/bin/libaa_plainst.so $R 2>/dev/null
python tools/ab_sustained.py tools/bin/libaa_new.so tools/bin/libaa_plainst.so $R 2>/dev/null
