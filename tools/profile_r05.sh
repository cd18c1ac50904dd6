#!/bin/bash
# Runs on the GPU box (via gpurun): the round-5 evidence set of the bench kernel and the other
# configurations.  rocprofv3 kernel trace (+stats) and, in SEPARATE passes without any tracing,
# the PMC counters.  Raw output under gpurun_out/prof/r05/; tools/summarize_r05.py condenses it
# into profiles/r05_*.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof/r05
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
HEAD="python3 $REPO/bench.py --steps 300 --warmup 20 --skip-cpu --skip-configs"
ALL="python3 $REPO/bench.py --steps 100 --warmup 20 --skip-cpu --config-steps 100 --sustained-s 0.2"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_head -o kt -- $HEAD > $OUT/kt_head.log 2>&1; echo "kt_head rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_all -o kt -- $ALL > $OUT/kt_all.log 2>&1; echo "kt_all rc=$?"
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc$i -o pmc -- python3 $REPO/bench.py --steps 30 --warmup 5 --skip-cpu --skip-configs --sustained-s 0.01 > $OUT/pmc$i.log 2>&1 || echo "pmc pass $i failed: $(tail -2 $OUT/pmc$i.log)"
  echo "pmc $i done"
done
# every configuration: traffic + MFMA + LDS conflicts per kernel
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc$i -o pmc -- python3 $REPO/bench.py --steps 10 --warmup 3 --skip-cpu --config-steps 20 --sustained-s 0.01 > $OUT/pmc$i.log 2>&1 || echo "pmc pass $i failed: $(tail -2 $OUT/pmc$i.log)"
  echo "pmc $i done"
done
# the mel kernels on their own (30 launches each): LDS bank conflicts against LDS-active cycles, MFMA busy
for G in "1024 256 logmel" "1024 256 mfcc" "2048 512 logmel"; do
  T=$(echo $G | tr ' ' '_')
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/mel_$T -o pmc -- python3 $REPO/tools/run_geometry.py $G 30 > $OUT/mel_$T.log 2>&1 || echo "mel $T failed: $(tail -2 $OUT/mel_$T.log)"
done
python3 $REPO/tools/summarize_r05.py $OUT
# the bench line each traced command printed (same process as the trace: its kernel_ms must agree with the trace's average)
grep -h '^{"metric"' $OUT/kt_head.log > $REPO/gpurun_out/r05_profiles/r05_kt_head_bench_line.json
grep -h '^{"metric"' $OUT/kt_all.log > $REPO/gpurun_out/r05_profiles/r05_kt_all_bench_line.json
du -sh $OUT | tail -1; rm -rf $OUT   # the raw traces (tens of MB per pass) stay on the box: gpurun copies at most 64 MiB back
