#!/bin/bash
# The bench step (tools/step_ab.py, no knob change) under several library builds (abl/*.so), interleaved over rounds.
# usage: tools/step_libs.sh [rounds] lib1 lib2 ...
R=$1; shift
for r in $(seq 1 $R); do for L in "$@"; do echo "== $L (round $r)"; MPSR_LIB_PATH=abl/$L.so python tools/step_ab.py --knob mpsr_debug_set_conv_pointwise --values=-1 --rounds 3 2>&1 | tail -1; done; done
