#!/usr/bin/env bash
# extra rocprofv3 counter passes of the training step (run through gpurun from the repo root):
#   LDS bank conflicts / LDS utilisation, and the instruction mix per kernel  ->  gpurun_out/<tag>/{lds,insts}
# summarised by tools/summarize_prof.py into profiles/<tag>_lds_insts.txt
set -uo pipefail
TAG="${1:-r03}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT="gpurun_out/$TAG"
mkdir -p "$OUT"
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-inference"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc LDSBankConflict LdsUtil --output-format csv -d "$OUT/lds" -- python3 bench.py $ARGS > "$OUT/lds.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES --output-format csv -d "$OUT/insts" -- python3 bench.py $ARGS > "$OUT/insts.log" 2>&1
tail -n 2 "$OUT/lds.log" | cut -c1-200
tail -n 2 "$OUT/insts.log" | cut -c1-200
find "$OUT/lds" "$OUT/insts" -name "*.csv" | wc -l
# issue-slot occupancy per instruction kind (cycles a wavefront of the SIMD had an instruction of that kind in flight)
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/active" -- python3 bench.py $ARGS > "$OUT/active.log" 2>&1
tail -n 1 "$OUT/active.log" | cut -c1-200
