#!/bin/bash
# Launch-by-launch timeline of ONE replayed VAE training step on the GPU box: kernel, duration, gap to the previous kernel's end.
#   tools/step_timeline.sh [out.txt]
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_t; mkdir -p /tmp/prof_t
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -o e -- python bench.py --steps 12 --warmup 4 --no-spade --no-render --no-graph-build --no-refine --no-sampling --no-cpu --no-check --no-dropin --large-batches= > /dev/null 2>&1
python - "${1:-/dev/stdout}" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/prof_t/**/e_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# a step starts at the N(0,1) draw / first kernel after the adam kernel: split on the Adam kernel
names = [r["Kernel_Name"] for r in rows]
ends = [i for i, n in enumerate(names) if "step_prologue_kernel(" in n or "randn_kernel(" in n]   # first kernel of an iteration (draw + input assembly)
out = open(sys.argv[1], "w")
if len(ends) < 3:
    print("no step boundary found", file=out); sys.exit(0)
a, b = ends[-3], ends[-2]                    # one whole step in the middle of the timed region
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"]); prev_end = None; tot = 0.0; gaps = 0.0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
    d = (e - s) / 1e3; tot += d; gaps += max(gap, 0.0)
    g = "%sx%sx%s" % (r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?"))
    print("%8.1f  %6.2f  %+6.2f  %-14s %s" % ((s - t0) / 1e3, d, gap, g, r["Kernel_Name"][:110]), file=out)
    prev_end = e
print("# launches %d  sum of durations %.1f us  sum of gaps %.1f us  span %.1f us" % (len(step), tot, gaps, (prev_end - t0) / 1e3), file=out)
PY
