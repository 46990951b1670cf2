cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_c8; mkdir -p $OUT
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-roofline --no-fast-mode > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 tools/pmc_reduce.py r03 c8 $OUT/p*/pmc_counter_collection.csv > $OUT/traffic.json
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/pmc_c8/traffic.json'))
for k,v in d['kernels'].items():
    if 'wino4' in k or 'narrow' in k or 'resize' in k: print(k, v)
print('overall', d.get('hbm_bytes_per_launch'), d.get('mfma_busy'))
PY
