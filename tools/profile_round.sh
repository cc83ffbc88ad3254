#!/bin/bash
# All rocprofv3 passes behind profiles/<tag>_*: run on the GPU box from the repository root.
#   tools/profile_round.sh r03
# Per leg (vae = the 64-graph training step, render = 16 rooms forward + backward, spade = batch 32) a run that executes THAT
# workload only, so that a kernel's avg_us is the duration on that leg's shapes:
#   1. kernel trace + stats                                             -> <tag>_<leg>_rocprofv3_kernel_stats_raw.csv
#   2. two PMC passes (FETCH_SIZE, WRITE_SIZE; counters in their own runs, no tracing domains)
#   3. one PMC pass with 8 SQ counters                                    -> <tag>_<leg>_sq_stalls.csv
# plus, for the render leg, a second kernel trace with SLN_SCENE_NO_SIDE=1 (the two chains of the scene backward on ONE stream:
# every kernel's duration is its own)                                     -> <tag>_render_noside_kernel_stats.csv
# and the kernel trace of the DEFAULT bench command (all legs)            -> <tag>_rocprofv3_kernel_stats_raw.csv
# tools/summarize_profile.py / summarize_sq.py condense them into profiles/.
set -u
TAG=${1:-r06}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
P=/tmp/prof_$TAG          # raw rocprofv3 output stays on the box (tens of MB); only the summaries travel back
rm -rf "$P"; mkdir -p "$P" profiles gpurun_out
FAILED=""
COMMON="--no-cpu --no-check --no-dropin --large-batches= --no-graph-build --no-refine --no-sampling"
declare -A FLAGS
FLAGS[vae]="$COMMON --no-render --no-spade --steps 100 --warmup 10"
# (render / spade: --legs-only - no VAE step runs in those traces; rounds 3-5 ran 125 of them in front of the leg, and the `pct` column of
#  the leg's table was a share of the wrong total)
FLAGS[render]="$COMMON --legs-only --no-spade --prof-steps 0 --render-iters 40 --render-warmup 5"
FLAGS[spade]="$COMMON --legs-only --no-render --no-colorize --prof-steps 0 --spade-iters 6 --spade-warmup 2"
declare -A SHORT
SHORT[vae]="$COMMON --no-graph --no-render --no-spade --steps 12 --warmup 3 --prof-steps 0"
SHORT[render]="$COMMON --legs-only --no-spade --prof-steps 0 --render-iters 4 --render-warmup 2"
SHORT[spade]="$COMMON --legs-only --no-render --no-colorize --prof-steps 0 --spade-iters 1 --spade-warmup 1"
flatten() { for g in $(find "$1" -name '*.csv'); do mv "$g" "$1/" 2>/dev/null; done; }
for leg in ${LEGS:-vae render spade}; do
  D="$P/$leg"; mkdir -p "$D"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/trace" -o vae -- python bench.py ${FLAGS[$leg]} > "$D/bench.json" 2> "$D/trace.err"
  for pass in fetch:FETCH_SIZE write:WRITE_SIZE; do
    sub=${pass%%:*}; ctr=${pass##*:}
    for try in 1 2; do                       # a PMC pass that yields no counter rows is retried once, then the leg FAILS (below)
      rm -rf "$D/$sub"
      timeout 600 rocprofv3 --pmc $ctr --output-format csv -d "$D/$sub" -o vae -- python bench.py ${SHORT[$leg]} > /dev/null 2> "$D/$sub.err"
      echo "rc=$? try=$try" >> "$D/$sub.err"
      flatten "$D/$sub"
      [ -s "$D/$sub/vae_counter_collection.csv" ] && [ "$(grep -c $ctr "$D/$sub/vae_counter_collection.csv")" -gt 0 ] && break
      echo "profile_round: $leg: the $ctr pass gave no counter rows (try $try)" >&2
    done
  done
  timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    --output-format csv -d "$D/sq" -o sq -- python bench.py ${SHORT[$leg]} > /dev/null 2> "$D/sq.err"
  for d in trace fetch write sq; do flatten "$D/$d"; done
  cp "$D/trace/vae_kernel_stats.csv" "profiles/${TAG}_${leg}_rocprofv3_kernel_stats_raw.csv"
  python tools/summarize_profile.py --require-traffic "$D" "profiles/${TAG}_${leg}" || { FAILED="$FAILED $leg"; for s in fetch write; do tail -c 1500 "$D/$s.err" > "gpurun_out/profile_${TAG}_${leg}_$s.err"; done; }
  python tools/summarize_sq.py "$D/sq/sq_counter_collection.csv" "profiles/${TAG}_${leg}"
done
if [ -n "${LEGS:-}" ]; then ls -la profiles/${TAG}_*; [ -n "$FAILED" ] && exit 4; exit 0; fi
# render, one stream
D="$P/render_noside"; mkdir -p "$D"
SLN_SCENE_NO_SIDE=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/trace" -o vae -- python bench.py ${FLAGS[render]} > "$D/bench.json" 2> "$D/trace.err"
flatten "$D/trace"
python tools/summarize_profile.py "$D" "profiles/${TAG}_render_noside"
# the default command, all legs
D="$P/all"; mkdir -p "$D"
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/trace" -o vae -- python bench.py --no-cpu "$@" > "$D/bench.json" 2> "$D/trace.err"
flatten "$D/trace"
cp "$D/trace/vae_kernel_stats.csv" "profiles/${TAG}_rocprofv3_kernel_stats_raw.csv"
# the JSON line of the plain default command (reads the traffic columns written above); written BEFORE the copy below, which used
# to put the committed file of an earlier run over a fresh one
python bench.py "$@" > "profiles/${TAG}_bench.json" 2> "$P/bench_default.err"
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* gpurun_out/profiles_$TAG/
for leg in vae render spade render_noside all; do cp "$P/$leg/bench.json" gpurun_out/profiles_$TAG/bench_$leg.json 2>/dev/null; tail -c 300 "$P/$leg/trace.err" > gpurun_out/profiles_$TAG/err_$leg.txt 2>/dev/null; done
ls -la profiles/${TAG}_*
if [ -n "$FAILED" ]; then
  echo "profile_round: HBM-traffic passes FAILED for:$FAILED - the previous profiles/${TAG}_<leg>_kernel_stats.csv (if any) were left in place; see gpurun_out/profile_${TAG}_*.err" >&2
  exit 4
fi
