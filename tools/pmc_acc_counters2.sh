#!/bin/bash
# scalar-cache / wait / L2 counters on the accumulate and row kernels (one rocprofv3 pass per counter group, --kernel-trace only); per-kernel averages
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmcacc2; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU" "SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD" "SQC_DCACHE_REQ SQC_DCACHE_HITS" "SQC_DCACHE_MISSES SQC_TC_DATA_READ_REQ" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA_RDREQ_sum" "TA_BUSY_avr TA_TA_BUSY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE TCP_GATE_EN1_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --kernel-trace -d $OUT/p$i -- python $ROOT/bench.py --steps 2 --warmup 1 --cpu-iters 0 --phase-reps 1 --repeats 1 > $OUT/log$i.txt 2>&1
  DB=$(ls -t $OUT/p$i/*/*.db 2>/dev/null | head -1)
  if [ -n "$DB" ]; then python $ROOT/tools/pmc_generic.py $DB >> $OUT/counters.txt 2>&1; else echo "# group '$grp': no database (counter not available?)" >> $OUT/counters.txt; tail -2 $OUT/log$i.txt >> $OUT/counters.txt; fi
  rm -rf $OUT/p$i
done
grep -E "k_chol_acc2|k_panel_rows|^#" $OUT/counters.txt
