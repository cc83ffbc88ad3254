cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_r; mkdir -p /tmp/prof_r
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r -o e -- python bench.py --steps 5 --warmup 2 --no-spade --no-graph-build --no-refine --no-cpu --no-check --large-batches= --render-iters 100 --render-warmup 20 > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("/tmp/prof_r/e_kernel_stats.csv")))
for r in rows:
    c=int(r["Calls"])
    if c>=100 and c<=800:
        print("%-100s %6s %9.2f ms %8.1f us"%(r["Name"][:100], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
PY
