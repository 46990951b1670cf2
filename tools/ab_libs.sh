#!/bin/bash
# Times the same conv_layer_bench.py command under several library builds (abl/*.so from tools/build_variant.sh),
# interleaved over rounds.  usage: tools/ab_libs.sh "<bench args>" lib1 lib2 ...
ARGS=$1; shift
for r in 1 2; do for L in "$@"; do echo "== $L (round $r)"; MPSR_LIB_PATH=abl/$L.so python tools/conv_layer_bench.py $ARGS 2>&1 | grep "dec\|custom"; done; done
