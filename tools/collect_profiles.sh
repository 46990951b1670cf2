#!/bin/bash
# Collects the round's profiling evidence on the GPU box into gpurun_out/prof_<tag>/ (copy what you want judged into
# profiles/).  usage: tools/collect_profiles.sh r02 [commit]
#   1. rocprofv3 --kernel-trace --stats over the default bench.py command (per-kernel average durations);
#   2. PMC passes over a short bench.py run, one counter set per pass (FETCH_SIZE and WRITE_SIZE cannot share a pass);
#   3. tools/pmc_reduce.py -> <tag>_pmc_traffic.json (HBM bytes per launch with the gfx950 FETCH_SIZE correction, MFMA
#      busy fraction, clock) which bench.py quotes as roofline.traffic / roofline.mfma_busy.
# Every rocprofv3 run sits under `timeout`; --pmc is never combined with any trace domain other than --kernel-trace.
TAG=${1:-r02}; COMMIT=${2:-unknown}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-fast-mode --no-train-step --no-full-image > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/stats.err || echo "stats pass failed"
cp $OUT/stats/*kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv 2>/dev/null
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" \
  "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
  "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-roofline --no-fast-mode > $OUT/p$i.log 2>&1 || echo "pmc pass $i failed/timeout"
done
python3 tools/pmc_summary.py $OUT/p*/pmc_counter_collection.csv > $OUT/${TAG}_pmc_summary.md
python3 tools/pmc_reduce.py $TAG $COMMIT $OUT/p*/pmc_counter_collection.csv > $OUT/${TAG}_pmc_traffic.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ops -o ops -- python3 tools/ops_bench.py > $OUT/${TAG}_ops_bench.json 2> $OUT/ops.err || echo "ops pass failed"
cp $OUT/ops/*kernel_stats.csv $OUT/${TAG}_ops_kernel_stats.csv 2>/dev/null
# the vendor fp32 GEMM on the heavy layers' im2col shapes and this library's per-layer table, at the same commit
timeout 300 python3 tools/gemm_reference.py > $OUT/${TAG}_vendor_gemm_reference.txt 2> $OUT/gemm.err || echo "gemm reference failed"
timeout 300 python3 tools/conv_layer_bench.py --tiles=-1 --split 0 --rounds 3 > $OUT/${TAG}_layers.txt 2> $OUT/layers.err || echo "layer table failed"
cat $OUT/${TAG}_pmc_traffic.json
# r06: the TRAINING step under the same counters (no backward kernel had an MFMA-busy figure before): three passes over
# tools/train_bench.py, reduced with the backward kernels' names included -> <tag>_train_pmc.json
j=0
for P in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
  j=$((j+1))
  timeout 400 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/t$j -o pmc -- python3 tools/train_bench.py --steps 2 --warmup 1 > $OUT/t$j.log 2>&1 || echo "train pmc pass $j failed/timeout"
done
PMC_EXTRA_PREFIXES="wino3_wgrad,wino4_wgrad,thin_,bn_,upconv_gather,act_bias_grad,adam_,clip_adam,seg_,dgrad_pack,relu_bitmask,max_pool" \
PMC_COMMAND="rocprofv3 --pmc <set> --kernel-trace -- python3 tools/train_bench.py --steps 2 --warmup 1 (one counter set per pass)" \
  python3 tools/pmc_reduce.py $TAG $COMMIT $OUT/t*/pmc_counter_collection.csv > $OUT/${TAG}_train_pmc.json
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trainstats -o train -- python3 tools/train_bench.py --steps 4 > $OUT/${TAG}_train_under_rocprof.json 2> $OUT/trainstats.err || echo "train stats failed"
cp $OUT/trainstats/*kernel_stats.csv $OUT/${TAG}_train_kernel_stats.csv 2>/dev/null
timeout 300 python3 tools/full_path_bench.py --images 8 > $OUT/${TAG}_full_path_bench.json 2> $OUT/fullpath.err || echo "full path failed"
timeout 200 python3 tools/rccl_selftest.py > $OUT/${TAG}_rccl_one_rank.json 2> $OUT/rccl.err || echo "rccl selftest failed"
timeout 200 python3 tools/emd_time.py > $OUT/${TAG}_emd_level_culling.txt 2> $OUT/emd.err || echo "emd timing failed"
timeout 300 python3 -m pytest tests/test_hostile_inputs_gpu.py -s -q > $OUT/${TAG}_hostile_inputs.txt 2>&1 || echo "hostile inputs run failed"
# the default bench command, untraced, on the same box as everything above (profiles/<tag>_bench_default_run.json)
timeout 600 python3 bench.py > $OUT/${TAG}_bench_default_run.json 2> $OUT/default.err || echo "default bench failed"
# r06 (late): the training step with its weight gradients on the second stream (default) and on the main stream, the
# both-trunk step, and the framework launches of a step by source line
{ for V in "" "--wgrad-main-stream" "" "--wgrad-main-stream" "--full-image" "--full-image --batch 32"; do
    echo "== train_bench.py --steps 10 --warmup 3 $V"; timeout 300 python3 tools/train_bench.py --steps 10 --warmup 3 $V 2>/dev/null | cut -c1-260
  done; } > $OUT/${TAG}_train_streams_ab.txt
timeout 300 python3 tools/aten_census.py --top 60 > $OUT/${TAG}_aten_census.txt 2> $OUT/census.err || echo "census failed"
