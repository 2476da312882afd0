#!/bin/bash
# Experiment builds of ONE kernel source with phases compiled out (timing only - results are not valid convolutions):
#   scripts/exp_abl.sh <source> <MACRO> <bits> [<bits> ...]      e.g.  scripts/exp_abl.sh conv_seam3 DIR_SEAM3_ABL 1 2 4 8
# -> scripts/_exp/lib_<source>_<bits>.so, used through DIRTORCH_AMD_LIB by the scripts/exp_*_time.py drivers.
# (Replaces the one-off exp_ring.sh / exp_sim.sh / exp_fill.sh / exp_fence.sh of rounds 2-3: same recipe, parameterised.)
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
C="$HERE/../deep-image-retrieval_amd/csrc"
O="$HERE/_exp"
SRC="$1"; MACRO="$2"; shift 2
mkdir -p "$O"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-but-set-variable -Wno-unused-variable"
OBJS=$(ls "$C/_build"/*.o | grep -v "/$SRC.o")
for bits in "$@"; do
  /opt/rocm/bin/hipcc $F -D$MACRO=$bits -c "$C/$SRC.hip" -o "$O/${SRC}_$bits.o" &
done
wait
for bits in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -ldl -o "$O/lib_${SRC}_$bits.so" "$O/${SRC}_$bits.o" $OBJS
  echo built "$O/lib_${SRC}_$bits.so"
done
