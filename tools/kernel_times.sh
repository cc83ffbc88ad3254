#!/bin/bash
# per-kernel durations of the VAE step on the GPU box: rocprofv3 kernel trace of a short bench run
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p /tmp/prof_k
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -o e -- python bench.py --steps 30 --warmup 5 --no-spade --no-render --no-graph-build --no-refine --no-sampling --no-cpu --no-check --no-dropin --large-batches= > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("/tmp/prof_k/e_kernel_stats.csv")))
for r in rows[:${1:-16}]:
    print("%-90s %6s %9.1f %8.2f"%(r["Name"][:90], r["Calls"], float(r["TotalDurationNs"])/1e3, float(r["AverageNs"])/1e3))
PY
