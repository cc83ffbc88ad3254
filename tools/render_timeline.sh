#!/bin/bash
# Launch-by-launch timeline of ONE 16-room render iteration (forward + backward) on the GPU box: kernel, duration, gap.
#   tools/render_timeline.sh [out.txt]          (SLN_SCENE_NO_SIDE=1 in the environment serialises the two backward chains,
#                                                so that every kernel's duration is its own)
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_rt; mkdir -p /tmp/prof_rt
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_rt -o e -- python bench.py --steps 3 --warmup 2 --no-spade --no-graph-build --no-refine --no-sampling --no-cpu --no-check --no-dropin --large-batches= --render-iters 12 --render-warmup 4 > /dev/null 2>&1
python - "${1:-/dev/stdout}" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/prof_rt/**/e_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if "project_faces" in n and "backward" not in n.lower() and "bwd" not in n.lower()]
if len(starts) < 6:
    starts = [i for i, n in enumerate(names) if "scene_init_stats_kernel" in n]
out = open(sys.argv[1], "w")
if len(starts) < 6:
    print("no iteration boundary found", file=out); sys.exit(0)
a, b = starts[-4], starts[-3]
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"]); prev_end = None; tot = 0.0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
    d = (e - s) / 1e3; tot += d
    g = "%sx%sx%s" % (r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?"))
    print("%8.1f  %7.2f  %+7.2f  %-16s %s" % ((s - t0) / 1e3, d, gap, g, r["Kernel_Name"][:100]), file=out)
    prev_end = max(prev_end or 0, e)
print("# launches %d  sum of durations %.1f us  span %.1f us" % (len(step), tot, (prev_end - t0) / 1e3), file=out)
PY
