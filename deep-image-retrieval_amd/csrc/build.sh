#!/bin/bash
# Builds libdir_engine.so for gfx950 (cross-compiles without a GPU). Usage: csrc/build.sh [-j N]
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../dirtorch_amd/libdir_engine.so"
OBJ="$HERE/_build"
mkdir -p "$OBJ"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-but-set-variable"
pids=()
for src in conv_f32 conv_pair conv_igemm conv_patch conv_patchlc conv_patchw conv_persist conv_ring conv_wreg conv_c3c1 conv_seam3 stem_pool pointwise resize gemm_f32 sim_split ranking comm engine c_api; do
  if [ ! -f "$OBJ/$src.o" ] || [ -n "$(find "$HERE" -maxdepth 1 \( -name '*.h' -o -name "$src.hip" \) -newer "$OBJ/$src.o")" ] \
     || [ "$HERE/../../include/dir_engine.h" -nt "$OBJ/$src.o" ]; then
    $HIPCC $FLAGS -c "$HERE/$src.hip" -o "$OBJ/$src.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -ldl -o "$OUT" "$OBJ"/conv_f32.o "$OBJ"/conv_pair.o "$OBJ"/conv_igemm.o "$OBJ"/conv_patch.o "$OBJ"/conv_patchlc.o "$OBJ"/conv_patchw.o "$OBJ"/conv_persist.o "$OBJ"/conv_ring.o "$OBJ"/conv_wreg.o "$OBJ"/conv_c3c1.o "$OBJ"/conv_seam3.o "$OBJ"/stem_pool.o "$OBJ"/pointwise.o "$OBJ"/resize.o "$OBJ"/gemm_f32.o "$OBJ"/sim_split.o "$OBJ"/ranking.o "$OBJ"/comm.o "$OBJ"/engine.o "$OBJ"/c_api.o
echo "built $OUT"
