#!/bin/bash
# Evidence that the forward step is POWER-limited on these boards: package power and clock level
# sampled with rocm-smi while the bench step loops, the same kernel on all-zero / post-ReLU / dense activations, and the
# wave-owned F(3x3,3x3) kernel's own cycle count against its launch time.  usage (on the GPU box): bash tools/power_probe.sh
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
echo "# rocm-smi while nothing runs"
rocm-smi --showpower --showmaxpower 2>&1 | grep -iE "power \(W\)|Max Graphics"
echo "# rocm-smi every 2 s while python tools/step_knob_run.py --steps 2500 loops the cfg3 forward + Chamfer step"
python tools/step_knob_run.py --steps 2500 > /tmp/step.log 2>&1 &
PID=$!
sleep 16
for i in 1 2 3 4 5; do rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -iE "power \(W\)|sclk|junction" | tr '\n' ' '; echo; sleep 2; done
wait $PID; tail -1 /tmp/step.log
echo "# the F(3x3,3x3) kernels back to back on all-zero / post-ReLU / dense N(0,1) activations (same filters): us per launch"
for a in "--zero-input" "--relu-input" ""; do echo "activations: ${a:-dense}"; python tools/wino3_forms.py --batches 256 $a --rounds 3 2>&1 | grep "^B"; done
if [ -f abl/w3wtrace.so ]; then
  echo "# cycle stamps of the wave-owned F(3x3,3x3) kernel (-DW3W_TRACE build) against its launch time"
  MPSR_LIB_PATH=abl/w3wtrace.so python tools/wino3w_trace.py --form 1 2>&1 | grep -E "launch|matrix pipe|wg 0 wave 0"
fi
if [ -f abl/w3ztrace.so ]; then
  echo "# ... and of the sixteen-product kernel (-DW3Z_TRACE build)"
  MPSR_LIB_PATH=abl/w3ztrace.so python tools/wino3w_trace.py --form 2 2>&1 | grep -E "launch|matrix pipe|wg 0 wave 0"
fi
