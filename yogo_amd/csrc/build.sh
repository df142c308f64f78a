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
# The wavefront-specialised kernels keep their accumulators in AGPRs that only their asm statements name (cdna_hip_programming.md 5.7
# item 4): outside those statements the compiler must not spill, use scratch, move through v_accvgpr_*, or give an AGPR to any
# operand (gfx950 lets the allocator use AV-class registers for DS / VMEM data).  audit_ws FILE.s STRICT(1|0)
audit_ws() {
  local S="$1" strict="$2" bad scr spill
  [[ -f "$S" ]] || return 0
  bad=$(awk '/;;#ASMSTART/{a=1} /;;#ASMEND/{a=0} { if (!a && $0 !~ /^[ \t]*[;.]/ && ($0 ~ /v_accvgpr_/ || $0 ~ /[ ,\[]a[0-9]+([ ,\]:]|$)/ || $0 ~ /[ ,]a\[[0-9]+:[0-9]+\]/)) n++ } END { print n+0 }' "$S")
  scr=$(awk '/;;#ASMSTART/{a=1} /;;#ASMEND/{a=0} { if (!a && $0 !~ /^[ \t]*[;.]/ && $0 ~ /scratch_/) n++ } END { print n+0 }' "$S")
  spill=$(grep -E "\.vgpr_spill_count:|\.private_segment_fixed_size:" "$S" | awk '{ s += $2 } END { print s+0 }')
  if [[ "$bad" != 0 || "$scr" != 0 || "$spill" != 0 ]]; then
    echo "$(basename "$S") audit: $bad compiler accumulator / AGPR-operand instructions outside the asm statements, $scr scratch instructions, spill / scratch total $spill" >&2
    [[ "$strict" == 1 && ( "$bad" != 0 || "$WS_SPILL_OK" != 1 ) ]] && { echo "audit FAILED" >&2; return 1; }
  else
    echo "$(basename "$S" -hip-amdgcn-amd-amdhsa-gfx950.s) audit ok (no compiler v_accvgpr_* / AGPR operands / scratch outside the asm statements, no spills)"
  fi
  return 0
}
WS_FILES="conv_bf16_ws conv_bf16_ws3 conv_bf16_ws16"
WS_SPILL_OK="${WS_SPILL_OK:-0}"   # (experiments: WS_SPILL_OK=1 turns a spill into a warning)
# variant TAG FILE [-DNAME=VALUE ...]: libyogo_hip_TAG.so = the product objects with FILE.hip recompiled under the given macros
# (in-process A/B of a compile-time choice: tools/ab_variants.py loads several such libraries side by side).  Build the product first.
if [[ "${1:-}" == "variant" ]]; then
  TAG="$2"; FILE="$3"; shift 3
  VOBJ="$HERE/obj_var"; mkdir -p "$VOBJ" "$OUT"
  extra=(); case "$FILE" in nms|decode_loss|conv_bf16_head) extra=(-ffp-contract=off) ;; conv_bf16_ws|conv_bf16_ws3|conv_bf16_ws16) extra=(-save-temps=obj -fno-slp-vectorize) ;; conv_bf16) extra=(-fno-slp-vectorize) ;; esac
  "$HIPCC" -O3 --offload-arch=gfx950 -fPIC -std=c++17 -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -I"$HERE" -I"$HERE/../../include" "$@" "${extra[@]}" -c "$HERE/$FILE.hip" -o "$VOBJ/${TAG}_$FILE.o"
  audit_ws "$VOBJ/${TAG}_$FILE-hip-amdgcn-amd-amdhsa-gfx950.s" 0   # (ablation variants are timings, not results: a warning)
  objs=(); for f in "$HERE"/*.hip; do b="$(basename "$f" .hip)"; [[ "$b" == "$FILE" ]] || objs+=("$HERE/obj/$b.o"); done
  "$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" "$VOBJ/${TAG}_$FILE.o" -ldl -o "$OUT/libyogo_hip_$TAG.so"
  echo "built $OUT/libyogo_hip_$TAG.so"
  exit 0
fi
HOBJ="$HERE/obj_hooks"; HOOKED="conv_bf16 conv_first conv_first_mfma"
mkdir -p "$OUT" "$OBJ" "$HOBJ"
# (EXTRA_DEFS="-DNAME=VALUE ...": extra macros for an experiment, e.g. a diagnostic build of an ablation)
COMMON=(-O3 --offload-arch=gfx950 -fPIC -std=c++17 -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -I"$HERE" -I"$HERE/../../include" "${DEFS[@]}" ${EXTRA_DEFS:-})
pids=()
for f in "$HERE"/*.hip; do
  base="$(basename "$f" .hip)"
  extra=()
  case "$base" in
    nms|decode_loss|conv_bf16_head) extra=(-ffp-contract=off) ;;
    conv_bf16_ws|conv_bf16_ws3|conv_bf16_ws16) extra=(-save-temps=obj -fno-slp-vectorize) ;;   # the assembly is audited below (asm-owned accumulator registers); no SLP packing: v_pk_*_f32 beside MFMAs costs more than it saves
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
  # the test-hooks objects (see the end of this script) compile beside the product's
  if [[ "${#DEFS[@]}" == 0 ]]; then case " $HOOKED " in *" $base "*)
    hstale=0
    for dep in "$f" "$HERE"/*.h "$HERE/../../include"/*.h "$HERE"/"$base"_*.inc; do
      [[ -f "$dep" && ( ! -f "$HOBJ/$base.o" || "$dep" -nt "$HOBJ/$base.o" ) ]] && hstale=1
    done
    if [[ "$hstale" == 1 ]]; then
      hextra=(); [[ "$base" == conv_bf16 ]] && hextra=(-fno-slp-vectorize)
      "$HIPCC" "${COMMON[@]}" -DYOGO_TEST_HOOKS "${hextra[@]}" -c "$f" -o "$HOBJ/$base.o" &
      pids+=($!)
    fi ;; esac; fi
done
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && wait "$p"; done
# (the diagnostic build's stamps cost registers: its numbers are timings, not results -- a warning there, an error in the product)
for wsf in $WS_FILES; do
  if ! audit_ws "$OBJ/$wsf-hip-amdgcn-amd-amdhsa-gfx950.s" "$([[ "${#DEFS[@]}" == 0 ]] && echo 1 || echo 0)"; then rm -f "$OBJ/$wsf.o"; exit 1; fi
  # ... and no compiler instruction reads a register whose LDS read an asm statement left in flight (tools/audit_inflight.py)
  if [[ "${#DEFS[@]}" == 0 && -f "$OBJ/$wsf-hip-amdgcn-amd-amdhsa-gfx950.s" ]]; then
    if ! python3 "$HERE/../../tools/audit_inflight.py" "$OBJ/$wsf-hip-amdgcn-amd-amdhsa-gfx950.s"; then rm -f "$OBJ/$wsf.o"; exit 1; fi
  fi
done
objs=(); for f in "$HERE"/*.hip; do objs+=("$OBJ/$(basename "$f" .hip).o"); done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -ldl -o "$OUT/$LIBNAME"
echo "built $OUT/$LIBNAME"
# libyogo_hip_hooks.so: the product objects with the three files that own a plan switch recompiled under -DYOGO_TEST_HOOKS (the
# yogo_hook_* entry points: tests/ and tools/ only -- the package never loads it; the kernels are the product's, same flags)
if [[ "${#DEFS[@]}" == 0 ]]; then
  objs=(); for f in "$HERE"/*.hip; do b="$(basename "$f" .hip)"; case " $HOOKED " in *" $b "*) objs+=("$HOBJ/$b.o") ;; *) objs+=("$OBJ/$b.o") ;; esac; done
  "$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -ldl -o "$OUT/libyogo_hip_hooks.so"
  echo "built $OUT/libyogo_hip_hooks.so"
fi
