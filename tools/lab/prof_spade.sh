cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_s; mkdir -p /tmp/prof_s
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o e -- python bench.py --steps 10 --warmup 3 --no-render --no-graph-build --no-refine --no-cpu --no-check --large-batches= > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("/tmp/prof_s/e_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print("%-95s %6s %9.2f ms %8.1f us %5.1f%%"%(r["Name"][:95], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
