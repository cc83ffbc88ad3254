#!/bin/bash
# per-kernel durations of the drop-in renderer legs (one room per call) for builds tools/lab/lib_<X>.so
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp 3d_sln_amd/libsln_hip.so /tmp/lib_keep.so
for v in "$@"; do
  cp tools/lab/lib_$v.so 3d_sln_amd/libsln_hip.so
  rm -rf /tmp/ks_$v; mkdir -p /tmp/ks_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$v -o k -- python bench.py --steps 3 --warmup 2 --no-spade --no-graph-build --no-refine --no-sampling --no-cpu --no-check --large-batches= --prof-steps 0 --render-iters 2 --render-warmup 1 > /dev/null 2> /tmp/ks_$v/err
  f=$(find /tmp/ks_$v -name 'k_kernel_stats.csv' | head -1)
  echo "== $v"; python - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:60]:
    if re.search("raster|pixel_map|depth_backward|texture|project", r['Name']): print("%-78s %6d calls  avg %8.2f us  total %8.2f ms" % (r['Name'][:78], int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
done
cp /tmp/lib_keep.so 3d_sln_amd/libsln_hip.so
