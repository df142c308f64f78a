#!/usr/bin/env bash
# Collects the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats, FETCH_SIZE / WRITE_SIZE (separate passes), MFMA-busy / wave counters  ->  gpurun_out/<tag>/{stats,fetch,write,mfma}
#   and the same for the inference path (tools/infer_profile.py)                           ->  gpurun_out/<tag>/infer_{e2e,post,fetch,write}
# then `python tools/summarize_prof.py gpurun_out/<tag> <tag>` (locally) writes profiles/<tag>_*.txt and profiles/traffic.json.
set -uo pipefail
TAG="${1:-r04}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT="gpurun_out/$TAG"
mkdir -p "$OUT"
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-inference"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py $ARGS > "$OUT/stats.log" 2>&1
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-inference"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 bench.py $ARGS > "$OUT/fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 bench.py $ARGS > "$OUT/write.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/mfma" -- python3 bench.py $ARGS > "$OUT/mfma.log" 2>&1
# the `yogo infer` path (configs[4], batch 256 bf16): kernel stats of the end-to-end pass and of the post-process alone, and the
# post-process kernels' HBM bytes (separate passes)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/infer_e2e" -- python3 tools/infer_profile.py e2e 5 > "$OUT/infer_e2e.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/infer_post" -- python3 tools/infer_profile.py post 10 > "$OUT/infer_post.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/infer_fetch" -- python3 tools/infer_profile.py post 3 > "$OUT/infer_fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/infer_write" -- python3 tools/infer_profile.py post 3 > "$OUT/infer_write.log" 2>&1
grep -h '"metric"\|"which"' "$OUT"/*.log | cut -c1-300
find "$OUT" -name "*.csv" | wc -l
