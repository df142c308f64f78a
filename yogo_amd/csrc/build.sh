#!/usr/bin/env bash
# Build libyogo_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT" "$HERE/obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
COMMON=(-O3 --offload-arch=gfx950 -fPIC -std=c++17 -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -I"$HERE" -I"$HERE/../../include")
pids=()
for f in "$HERE"/*.hip; do
  base="$(basename "$f" .hip)"
  extra=()
  case "$base" in
    nms|decode_loss) extra=(-ffp-contract=off) ;;
  esac
  if [[ ! -f "$HERE/obj/$base.o" || "$f" -nt "$HERE/obj/$base.o" || "$HERE/common.h" -nt "$HERE/obj/$base.o" ]]; then
    "$HIPCC" "${COMMON[@]}" "${extra[@]}" -c "$f" -o "$HERE/obj/$base.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "$HERE"/obj/*.o -o "$OUT/libyogo_hip.so"
echo "built $OUT/libyogo_hip.so"
