#!/bin/bash
# PMC deep-dive of ONE conv layer shape.  usage (on the GPU box): tools/pmc_layer.sh <outdir> <tile> <shape> [extra
# conv_layer_bench.py flags, e.g. --math bf16x3]
# e.g. tools/pmc_layer.sh gpurun_out/pmc_l1 3 12,12,1024,256,1,1,0
# Every rocprofv3 run is wrapped in `timeout`: an invalid counter set makes rocprofv3 abort and then hang.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=$1; TILE=$2; SHAPE=$3; shift 3; EXTRA="$@"
mkdir -p "$OUT"
i=0
for P in \
 "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM" \
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum" \
 "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
 "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" ; do
  i=$((i+1))
  timeout 90 rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$OUT/p$i" -o pmc -- python3 tools/conv_layer_bench.py --tiles $TILE --reps 3 --shape $SHAPE $EXTRA > "$OUT/p$i.log" 2>&1 || echo "pass $i failed/timeout"
done
python3 tools/pmc_summary.py "$OUT"/p*/pmc_counter_collection.csv | grep "conv\|pw_\|kernel |" > "$OUT/summary.md"
cat "$OUT/summary.md"
