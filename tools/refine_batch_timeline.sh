#!/bin/bash
# timeline of ONE iteration of the R-room refinement loop (eager): kernels in start order with queue, start offset, duration and the gap
# to the previous kernel's end on the same queue.  GPU box.
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
R=${1:-16}
rm -rf /tmp/rbt; ITERS=${ITERS:-20} timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/rbt -o rb -- python tools/refine_batch_time.py $R > /tmp/rbt.out 2> /tmp/rbt.err
f=$(find /tmp/rbt -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# iterations are delimited by refine_sgd_rooms_kernel; take one from the middle of the first eager run
idx = [i for i, r in enumerate(rows) if "refine_sgd_rooms" in r["Kernel_Name"]]
a, b = idx[10] + 1, idx[11] + 1
it = rows[a:b]
t0 = int(it[0]["Start_Timestamp"])
last_end = {}
main_q = it[0]["Queue_Id"]
busy = {}
print("iteration: %d kernels, %.1f us from first start to last end" % (len(it), (max(int(r["End_Timestamp"]) for r in it) - t0) / 1e3))
for r in it:
    q = r["Queue_Id"]; s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    busy[q] = busy.get(q, 0) + (e - s)
    print("q%-3s %9.1f us  dur %7.1f  gap %6.1f  %s" % (q, (s - t0) / 1e3, (e - s) / 1e3, gap, r["Kernel_Name"][:90]))
for q, v in busy.items():
    print("queue %s busy %.1f us" % (q, v / 1e3))
PY
