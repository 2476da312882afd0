#!/bin/bash
# Experiment builds of the library with phases compiled out or a macro flipped (timing only - results are NOT valid):
#   scripts/exp_abl.sh <source> <MACRO> <value> [<value> ...]     one kernel source rebuilt with -D<MACRO>=<value>
#   scripts/exp_abl.sh all <MACRO> <value>                         EVERY source rebuilt with it
# -> scripts/_exp/lib_<source>_<value>.so, used through DIRTORCH_AMD_LIB by bench.py and the scripts/exp_*_time.py drivers.
# Macros in the tree:  conv_seam3  DIR_SEAM3_ABL  (1 no weight DMA, 2 no t2/residual DMA, 4 no MFMAs, 8 no epilogues, 16 no stores)
#                      conv_ring   DIR_RING_ABL   (1 no pixel DMA, 2 no weight DMA, 4 no MFMAs, 8 no epilogue)
#                      sim_split   DIR_SIM_ABL    (1 no database DMA, 2 no query DMA, 4 consumers only take the barriers)
#                      conv_igemm  DIR_EXP_FILL_ONLY / DIR_EXP_NO_FILL (value 1): the LDS-DMA ring alone / the MFMAs alone
#                      all         DIR_EXP_NO_RING_FENCE (value 1): ring_barrier() as a raw s_barrier - what the fences cost
# (Replaces the one-off exp_ring.sh / exp_sim.sh / exp_fill.sh / exp_fence.sh of rounds 2-3: same recipe, parameterised.)
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
C="$HERE/../deep-image-retrieval_amd/csrc"
O="$HERE/_exp"
SRC="$1"; MACRO="$2"; shift 2
mkdir -p "$O"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-but-set-variable -Wno-unused-variable"
if [ "$SRC" = all ]; then
  V="$1"; T="$O/all_$MACRO"; mkdir -p "$T"
  for f in "$C"/*.hip; do /opt/rocm/bin/hipcc $F -D$MACRO=$V -c "$f" -o "$T/$(basename "$f" .hip).o" & done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -ldl -o "$O/lib_all_${MACRO}.so" "$T"/*.o
  rm -rf "$T"
  echo built "$O/lib_all_${MACRO}.so"
  exit 0
fi
OBJS=$(ls "$C/_build"/*.o | grep -v "/$SRC.o")
for bits in "$@"; do
  /opt/rocm/bin/hipcc $F -D$MACRO=$bits -c "$C/$SRC.hip" -o "$O/${SRC}_$bits.o" &
done
wait
for bits in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -ldl -o "$O/lib_${SRC}_$bits.so" "$O/${SRC}_$bits.o" $OBJS
  rm -f "$O/${SRC}_$bits.o"
  echo built "$O/lib_${SRC}_$bits.so"
done
