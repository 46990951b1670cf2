cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for SH in 12,12,1024,256,1,1,0 12,12,256,1024,1,1,1 12,12,256,256,3,4,0; do
  OUT=gpurun_out/pmc2_$(echo $SH | tr ',' '_')
  mkdir -p $OUT
  i=0
  for P in \
   "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
   "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
   "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
   "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
   "TA_BUSY_avr TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum" ; do
    i=$((i+1))
    timeout 120 rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$OUT/p$i" -o pmc -- python3 tools/conv_layer_bench.py --tiles 3 --reps 3 --shape $SH > "$OUT/p$i.log" 2>&1 || echo "pass $i failed/timeout"
  done
  python3 tools/pmc_summary.py "$OUT"/p*/pmc_counter_collection.csv | grep "conv\|kernel |" > "$OUT/summary.md"
  echo "== $SH"; cat "$OUT/summary.md"
done
