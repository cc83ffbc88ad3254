#!/bin/bash
# per-kernel average durations of the VAE leg for builds of libsln_hip.so (tools/lab/lib_<X>.so), same box:  tools/lab/kstats_libs.sh C E
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp 3d_sln_amd/libsln_hip.so /tmp/lib_keep.so
for v in "$@"; do
  cp tools/lab/lib_$v.so 3d_sln_amd/libsln_hip.so
  rm -rf /tmp/ks_$v; mkdir -p /tmp/ks_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$v -o k -- python bench.py --no-cpu --no-check --no-dropin --large-batches= --no-graph-build --no-refine --no-render --no-spade --steps 60 --warmup 10 > /dev/null 2> /tmp/ks_$v/err
  f=$(find /tmp/ks_$v -name 'k_kernel_stats.csv' | head -1)
  echo "== $v"; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:26]:
    print("%-62s %6d calls  avg %8.2f us" % (r['Name'][:62], int(r['Calls']), float(r['AverageNs']) / 1e3))
PY
done
cp /tmp/lib_keep.so 3d_sln_amd/libsln_hip.so
