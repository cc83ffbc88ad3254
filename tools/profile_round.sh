#!/bin/bash
# All rocprofv3 passes behind profiles/<tag>_*: run on the GPU box from the repository root.
#   tools/profile_round.sh r02 [extra bench.py flags]
# 1. kernel trace + stats of the DEFAULT bench command (every leg)   -> <tag>_rocprofv3_kernel_stats_raw.csv
# 2. two PMC passes (FETCH_SIZE, WRITE_SIZE; counters in their own runs) of a short bench run
# 3. one PMC pass with 8 SQ counters                                    -> <tag>_sq_stalls.csv
# then tools/summarize_profile.py / summarize_sq.py condense them into profiles/.
set -u
TAG=${1:-r02}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
P=/tmp/prof_$TAG          # raw rocprofv3 output stays on the box (tens of MB); only the summaries travel back
rm -rf "$P"; mkdir -p "$P" profiles gpurun_out
SHORT="--steps 20 --warmup 5 --no-cpu --no-check --spade-iters 1 --spade-warmup 1 --render-iters 5 --render-warmup 2 --graph-iters 5 --large-batches= $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$P/trace" -o vae -- python bench.py "$@" > "$P/bench_traced.json" 2> "$P/trace.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$P/fetch" -o vae -- python bench.py $SHORT > /dev/null 2> "$P/fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$P/write" -o vae -- python bench.py $SHORT > /dev/null 2> "$P/write.err"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --output-format csv -d "$P/sq" -o sq -- python bench.py $SHORT > /dev/null 2> "$P/sq.err"
for d in trace fetch write sq; do f=$(find "$P/$d" -name '*.csv' | head -1); [ -n "$f" ] && for g in $(find "$P/$d" -name '*.csv'); do mv "$g" "$P/$d/" 2>/dev/null; done; done
cp "$P/trace/vae_kernel_stats.csv" "profiles/${TAG}_rocprofv3_kernel_stats_raw.csv"
python tools/summarize_profile.py "$P" "profiles/${TAG}"
python tools/summarize_sq.py "$P/sq/sq_counter_collection.csv" "profiles/${TAG}"
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* "$P/bench_traced.json" gpurun_out/profiles_$TAG/
tail -c 600 "$P/bench_traced.json"
