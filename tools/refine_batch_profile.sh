#!/bin/bash
# kernel trace of the 16-room refinement loop (tools/refine_batch_time.py): per-kernel totals of one run of ITERS iterations
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
R=${1:-16}
rm -rf /tmp/rbp; ITERS=${ITERS:-30} timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rbp -o rb -- python tools/refine_batch_time.py $R > /tmp/rbp.out 2> /tmp/rbp.err
f=$(find /tmp/rbp -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms" % (tot / 1e6))
for r in rows[:45]:
    print("%8d calls %9.1f us total %8.2f us avg %5.1f%%  %s" % (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["Percentage"]), r["Name"][:110]))
PY
tail -3 /tmp/rbp.out
