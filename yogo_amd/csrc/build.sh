#!/usr/bin/env bash
# Build libyogo_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
#   bash build.sh        -> yogo_amd/lib/libyogo_hip.so       (the product: no diagnostic code is compiled in)
#   bash build.sh diag   -> yogo_amd/lib/libyogo_hip_diag.so  (-DYOGO_DIAG: ablation bits + s_memtime phase stamps in
#                           conv_bf16_kernel, driven by yogo_diag_conv_bf16(); tools/ only, never loaded by the package)
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
OBJ="$HERE/obj"
LIBNAME="libyogo_hip.so"
DEFS=()
if [[ "${1:-}" == "diag" ]]; then OBJ="$HERE/obj_diag"; LIBNAME="libyogo_hip_diag.so"; DEFS=(-DYOGO_DIAG -DYOGO_DIAG_PHASES); fi
# diag-coarse: start / loop / epilogue / end stamps only (the per-phase sums of the ping-pong loop cost it scalar registers)
if [[ "${1:-}" == "diag-coarse" ]]; then OBJ="$HERE/obj_diagc"; LIBNAME="libyogo_hip_diag.so"; DEFS=(-DYOGO_DIAG); fi
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# variant TAG FILE [-DNAME=VALUE ...]: libyogo_hip_TAG.so = the product objects with FILE.hip recompiled under the given macros
# (in-process A/B of a compile-time choice: tools/ab_variants.py loads several such libraries side by side).  Build the product first.
if [[ "${1:-}" == "variant" ]]; then
  TAG="$2"; FILE="$3"; shift 3
  VOBJ="$HERE/obj_var"; mkdir -p "$VOBJ" "$OUT"
  extra=(); case "$FILE" in nms|decode_loss) extra=(-ffp-contract=off) ;; esac
  "$HIPCC" -O3 --offload-arch=gfx950 -fPIC -std=c++17 -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -I"$HERE" -I"$HERE/../../include" "$@" "${extra[@]}" -c "$HERE/$FILE.hip" -o "$VOBJ/${TAG}_$FILE.o"
  objs=(); for o in "$HERE/obj"/*.o; do [[ "$(basename "$o")" == "$FILE.o" ]] || objs+=("$o"); done
  "$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" "$VOBJ/${TAG}_$FILE.o" -ldl -o "$OUT/libyogo_hip_$TAG.so"
  echo "built $OUT/libyogo_hip_$TAG.so"
  exit 0
fi
mkdir -p "$OUT" "$OBJ"
COMMON=(-O3 --offload-arch=gfx950 -fPIC -std=c++17 -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -I"$HERE" -I"$HERE/../../include" "${DEFS[@]}")
pids=()
for f in "$HERE"/*.hip; do
  base="$(basename "$f" .hip)"
  extra=()
  case "$base" in
    nms|decode_loss) extra=(-ffp-contract=off) ;;
  esac
  stale=0
  for dep in "$f" "$HERE"/*.h "$HERE/../../include"/*.h "$HERE"/"$base"_*.inc; do
    [[ -f "$dep" && ( ! -f "$OBJ/$base.o" || "$dep" -nt "$OBJ/$base.o" ) ]] && stale=1
  done
  if [[ "$stale" == 1 ]]; then
    "$HIPCC" "${COMMON[@]}" "${extra[@]}" -c "$f" -o "$OBJ/$base.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "$OBJ"/*.o -ldl -o "$OUT/$LIBNAME"
echo "built $OUT/$LIBNAME"
