#!/bin/bash
# Builds abl/<name>.so = the library with ONE source recompiled under extra flags (A/B-ing a kernel variant or a
# trace build inside one GPU session: pass it through MPSR_LIB_PATH).  usage: tools/build_variant.sh <name> <source.hip> <flags...>
# e.g. tools/build_variant.sh w4trace winograd4.hip -DW4_TRACE
set -e
NAME=$1; SRC=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/monopsr_amd/csrc"
make -s -j8
mkdir -p "$ROOT/abl"
OBJ="$ROOT/abl/$NAME.${SRC%.hip}.o"
NOFMA=""
case "$SRC" in nn_distance.hip|image_ops.hip|crop_grad.hip) NOFMA="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I../../include -I. $NOFMA "$@" -c "$SRC" -o "$OBJ"
OTHERS=$(ls *.o | grep -v "^${SRC%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/abl/$NAME.so" $OTHERS "$OBJ"
echo "built abl/$NAME.so"
