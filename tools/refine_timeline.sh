#!/bin/bash
# Launch-by-launch timeline of ONE eager refinement iteration (between two refine_sgd_kernel launches) on the GPU box.
#   tools/refine_timeline.sh [out.txt]
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_rt; mkdir -p /tmp/prof_rt
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_rt -o e -- python tools/refine_profile.py eager > /dev/null 2>&1
python - "${1:-/dev/stdout}" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/prof_rt/**/e_kernel_trace.csv", recursive=True)[0]
rows = [dict(r, kind="k") for r in csv.DictReader(open(f))]
mc = glob.glob("/tmp/prof_rt/**/e_memory_copy_trace.csv", recursive=True)
for m in mc:
    for r in csv.DictReader(open(m)):
        rows.append({"Start_Timestamp": r["Start_Timestamp"], "End_Timestamp": r["End_Timestamp"], "Kernel_Name": "<memcpy %s %s B>" % (r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?"))), "kind": "m"})
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "refine_sgd_kernel" in r["Kernel_Name"]]
out = open(sys.argv[1], "w")
a, b = ends[-3] + 1, ends[-2] + 1
t0 = int(rows[a]["Start_Timestamp"]); prev = None; tot = 0.0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0.0 if prev is None else (s - prev) / 1e3
    tot += (e - s) / 1e3
    print("%8.1f  %6.2f  %+6.2f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, r["Kernel_Name"][:120]), file=out)
    prev = e
print("# launches %d  sum of durations %.1f us  span %.1f us" % (b - a, tot, (prev - t0) / 1e3), file=out)
PY
