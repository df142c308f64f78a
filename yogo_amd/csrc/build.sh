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
  extra=(); case "$FILE" in nms|decode_loss) extra=(-ffp-contract=off) ;; conv_bf16_ws|conv_bf16) extra=(-fno-slp-vectorize) ;; esac
  "$HIPCC" -O3 --offload-arch=gfx950 -fPIC -std=c++17 -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -I"$HERE" -I"$HERE/../../include" "$@" "${extra[@]}" -c "$HERE/$FILE.hip" -o "$VOBJ/${TAG}_$FILE.o"
  objs=(); for f in "$HERE"/*.hip; do b="$(basename "$f" .hip)"; [[ "$b" == "$FILE" ]] || objs+=("$HERE/obj/$b.o"); done
  "$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" "$VOBJ/${TAG}_$FILE.o" -ldl -o "$OUT/libyogo_hip_$TAG.so"
  echo "built $OUT/libyogo_hip_$TAG.so"
  exit 0
fi
mkdir -p "$OUT" "$OBJ"
# (EXTRA_DEFS="-DNAME=VALUE ...": extra macros for an experiment, e.g. a diagnostic build of an ablation)
COMMON=(-O3 --offload-arch=gfx950 -fPIC -std=c++17 -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -I"$HERE" -I"$HERE/../../include" "${DEFS[@]}" ${EXTRA_DEFS:-})
pids=()
for f in "$HERE"/*.hip; do
  base="$(basename "$f" .hip)"
  extra=()
  case "$base" in
    nms|decode_loss) extra=(-ffp-contract=off) ;;
    conv_bf16_ws) extra=(-save-temps=obj -fno-slp-vectorize) ;;   # the assembly is audited below (asm-owned accumulator registers); no SLP packing: v_pk_*_f32 beside MFMAs costs more than it saves
    conv_bf16) extra=(-fno-slp-vectorize) ;;   # the same for the tiled kernels: the merged-epilogue forward instantiations -5 ... -9.5 % in the same-box A/B (gpurun_out/r4_abnoslp.log); the training step's launches take the lean epilogue with its explicit packed math and do not change
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
# conv_bf16_ws.hip keeps its accumulators in AGPRs that only its asm statements name (cdna_hip_programming.md 5.7 item 4): the
# compiler must not spill, use scratch, or touch an accumulator register outside those statements
WS_S="$OBJ/conv_bf16_ws-hip-amdgcn-amd-amdhsa-gfx950.s"
if [[ -f "$WS_S" ]]; then
  bad=$(awk '/;;#ASMSTART/{a=1} /;;#ASMEND/{a=0} { if (!a && $0 ~ /v_accvgpr_|scratch_/ && $0 !~ /^[ \t]*;/) n++ } END { print n+0 }' "$WS_S")
  spill=$(grep -E "\.vgpr_spill_count:|\.private_segment_fixed_size:" "$WS_S" | awk '{ s += $2 } END { print s+0 }')
  if [[ "$bad" != 0 || "$spill" != 0 ]]; then
    echo "conv_bf16_ws audit FAILED: $bad compiler accumulator / scratch instructions outside the asm statements, spill / scratch total $spill" >&2
    # (the diagnostic build's stamps cost registers: its numbers are timings, not results -- a warning there, an error in the product)
    if [[ "${#DEFS[@]}" == 0 ]]; then rm -f "$OBJ/conv_bf16_ws.o"; exit 1; fi
  else
    echo "conv_bf16_ws audit ok (no compiler v_accvgpr_* / scratch outside the asm statements, no spills)"
  fi
fi
objs=(); for f in "$HERE"/*.hip; do objs+=("$OBJ/$(basename "$f" .hip).o"); done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -ldl -o "$OUT/$LIBNAME"
echo "built $OUT/$LIBNAME"
