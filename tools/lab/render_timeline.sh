#!/bin/bash
# one iteration of the 16-room render batch (forward + backward, side stream on) as a timeline: start offset, duration, queue of every
# kernel, from a rocprofv3 kernel trace.   tools/lab/render_timeline.sh [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf /tmp/rt; mkdir -p /tmp/rt
env "$@" rocprofv3 --kernel-trace --output-format csv -d /tmp/rt -o k -- python bench.py --no-cpu --no-check --no-dropin --large-batches= --no-graph-build --no-refine --no-sampling --no-spade --steps 3 --warmup 2 --prof-steps 0 --render-iters 40 --render-warmup 5 > /dev/null 2> /tmp/rt/err
f=$(find /tmp/rt -name 'k_kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# the last complete iteration: from the last-but-one project_faces_kernel to the last one
idx = [i for i, n in enumerate(names) if n.startswith('project_faces_kernel') or 'project_faces_kernel(' in n]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]['Start_Timestamp'])
prev_end = t0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%8.1f us  +%7.1f us  q%-3s gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), (s - prev_end) / 1e3, r['Kernel_Name'][:70]))
    prev_end = max(prev_end, e)
print("iteration: %.1f us" % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3))
PY
