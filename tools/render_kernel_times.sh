#!/bin/bash
# per-kernel durations of the render leg on the GPU box
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/prof_r
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r -o r -- python bench.py --steps 5 --warmup 2 --no-spade --no-graph-build --no-cpu --render-iters 20 > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_r/r_kernel_stats.csv")))
for r in rows[:80]:
    n=r["Name"]
    if any(k in n for k in ("raster","scene","pixel_map","pmb_","depth_backward","texture","fill_ones","indexing")):
        print("%-80s %6s %9.1f %8.2f"%(n[:80], r["Calls"], float(r["TotalDurationNs"])/1e3, float(r["AverageNs"])/1e3))
PY
