#!/bin/bash
# Builds libdir_engine.so for gfx950 (cross-compiles without a GPU). Usage: csrc/build.sh [-j N]
#   DIR_EXPERIMENTS=1 csrc/build.sh   an EXPERIMENTS build: + conv_ring.hip (128x256_ring1x1: ties conv_persist.hip inside the
#                                     network) and conv_seam3.hip (layer3's conv3 -> conv1 seam: loses to the two kernels it
#                                     replaces, profiles/r04_seam3_ablation.txt), compiled with -DDIR_EXPERIMENTS into
#                                     dirtorch_amd/libdir_engine_exp.so - select it with DIRTORCH_AMD_LIB=<that path> and
#                                     DIRTORCH_AMD_EXPERIMENTS=1.  The default library ships neither kernel.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-but-set-variable"
SRCS="conv_f32 conv_pair conv_igemm conv_small conv_patch conv_patchlc conv_patchw conv_patchs2 conv_persist conv_persistlc conv_wreg conv_wregd conv_c3c1 conv_c3c1lc stem_pool stem_u8 pointwise resize gemm_f32 sim_split ranking comm engine c_api"
if [ -n "${DIR_EXPERIMENTS:-}" ]; then
  OUT="$HERE/../dirtorch_amd/libdir_engine_exp.so"; OBJ="$HERE/_build_exp"; FLAGS="$FLAGS -DDIR_EXPERIMENTS"; SRCS="$SRCS conv_ring conv_seam3"
else
  OUT="$HERE/../dirtorch_amd/libdir_engine.so"; OBJ="$HERE/_build"
fi
mkdir -p "$OBJ"
pids=()
objs=()
for src in $SRCS; do
  objs+=("$OBJ/$src.o")
  if [ ! -f "$OBJ/$src.o" ] || [ -n "$(find "$HERE" -maxdepth 1 \( -name '*.h' -o -name "$src.hip" \) -newer "$OBJ/$src.o")" ] \
     || [ "$HERE/../../include/dir_engine.h" -nt "$OBJ/$src.o" ]; then
    $HIPCC $FLAGS -c "$HERE/$src.hip" -o "$OBJ/$src.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -ldl -o "$OUT" "${objs[@]}"
echo "built $OUT"
