#!/bin/bash
# (historic: the AA_SMALL_PLAIN_MAP pass needs a library built with -DAA_EXPERIMENTS=1 since r4 - tools/build_variants.sh)
# Runs on the GPU box (via gpurun): the round-2 evidence set of the bench kernel and the other
# configurations.  rocprofv3 kernel trace (+stats) and, in SEPARATE passes without any tracing,
# the PMC counters.  Raw output under gpurun_out/prof/r02/; tools/summarize_r02.py condenses it
# into profiles/r02_*.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof/r02
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
HEAD="python3 $REPO/bench.py --steps 300 --warmup 20 --skip-cpu --skip-configs"
ALL="python3 $REPO/bench.py --steps 100 --warmup 20 --skip-cpu --config-steps 100"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_head -o kt -- $HEAD > $OUT/kt_head.log 2>&1; echo "kt_head rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_all -o kt -- $ALL > $OUT/kt_all.log 2>&1; echo "kt_all rc=$?"
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc$i -o pmc -- python3 $REPO/bench.py --steps 30 --warmup 5 --skip-cpu --skip-configs > $OUT/pmc$i.log 2>&1 || echo "pmc pass $i failed: $(tail -2 $OUT/pmc$i.log)"
  echo "pmc $i done"
done
# every configuration: traffic + MFMA + LDS conflicts per kernel
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc$i -o pmc -- python3 $REPO/bench.py --steps 10 --warmup 3 --skip-cpu --config-steps 20 > $OUT/pmc$i.log 2>&1 || echo "pmc pass $i failed: $(tail -2 $OUT/pmc$i.log)"
  echo "pmc $i done"
done
# 256 / 64: LDS bank conflicts before (plain lane map) and after (relabelled frame slots)
for V in plain new; do
  E=""; [ $V = plain ] && E="AA_SMALL_PLAIN_MAP=1"
  env $E true
  AA_SMALL_PLAIN_MAP=$([ $V = plain ] && echo 1 || echo 0) rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/lds256_$V -o pmc -- python3 $REPO/tools/run_geometry.py 256 64 mag 40 > $OUT/lds256_$V.log 2>&1 || echo "lds256 $V failed"
done
python3 $REPO/tools/summarize_r02.py $OUT
