cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_wino; mkdir -p $OUT
i=0
for P in \
 "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
 "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS" ; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$OUT/p$i" -o pmc -- python3 tools/conv_layer_bench.py --tiles -1 --split 0 --wino 1 --reps 3 --shape 48,48,256,128,3,1,0 > "$OUT/p$i.log" 2>&1 || echo "pass $i failed/timeout"
done
python3 tools/pmc_summary.py "$OUT"/p*/pmc_counter_collection.csv | grep "wino_conv\|kernel |"
