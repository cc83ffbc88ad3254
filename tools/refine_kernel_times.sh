#!/bin/bash
# kernel statistics of the refinement leg alone (one room, 60 iterations): which launches an iteration is made of
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_rf; mkdir -p /tmp/prof_rf
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rf -o e -- python bench.py --steps 3 --warmup 2 --no-spade --no-render --no-graph-build --no-cpu --no-check --no-dropin --large-batches= --prof-steps 0 --refine-iters 60 > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/prof_rf/**/e_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = 0.0
for r in rows[:70]:
    c = int(r["Calls"])
    if c % 63 == 0 or c % 60 == 0 or c >= 60:
        print("%-95s %6d %9.1f %8.2f" % (r["Name"][:95], c, float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3))
PY
