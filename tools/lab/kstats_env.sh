#!/bin/bash
# per-kernel average durations of the VAE leg under environment settings, same box:  tools/lab/kstats_env.sh "" "SLN_EMB_RPB=32" ...
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
i=0
for v in "$@"; do
  i=$((i+1)); rm -rf /tmp/ke_$i; mkdir -p /tmp/ke_$i
  env $v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ke_$i -o k -- python bench.py --no-cpu --no-check --no-dropin --large-batches= --no-graph-build --no-refine --no-render --no-spade --steps 60 --warmup 10 > /dev/null 2> /tmp/ke_$i/err
  f=$(find /tmp/ke_$i -name 'k_kernel_stats.csv' | head -1)
  echo "== [$v]"; python - "$f" "${KSTATS_FILTER:-.}" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:40]:
    if re.search(sys.argv[2], r['Name']): print("%-62s %6d calls  avg %8.2f us" % (r['Name'][:62], int(r['Calls']), float(r['AverageNs']) / 1e3))
PY
done
