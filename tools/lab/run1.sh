cd /tmp && export TMPDIR=/tmp; cd /root/repo
python tools/lab/spade_b1.py 2>&1 | grep -v amdgpu
rm -rf /tmp/sb; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sb -o sb -- python tools/lab/spade_b1.py > /dev/null 2>&1
f=$(find /tmp/sb -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms over %d launches" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
for r in rows[:28]:
    print("%6d calls %9.1f us total %8.2f us avg %5.1f%%  %s" % (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["Percentage"]), r["Name"][:105]))
PY
