#!/bin/bash
# a few SQ / LDS / TA counters on the accumulate kernels (one rocprofv3 pass per counter group); prints per-kernel averages
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmcacc; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" "SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d $OUT/p$i -- python $ROOT/bench.py --steps 2 --warmup 1 --cpu-iters 0 --phase-reps 1 > $OUT/log$i.txt 2>&1
  DB=$(ls -t $OUT/p$i/*/*.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python $ROOT/tools/pmc_generic.py $DB >> $OUT/counters.txt 2>&1
  rm -rf $OUT/p$i
done
cat $OUT/counters.txt | tail -80
