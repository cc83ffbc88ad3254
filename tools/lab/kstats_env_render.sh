#!/bin/bash
# per-kernel average durations of the render leg (16 rooms, one stream) under environment settings, same box:
#   tools/lab/kstats_env_render.sh "" "SLN_PMB_ABL=1" ...
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
i=0
for v in "$@"; do
  i=$((i+1)); rm -rf /tmp/kr_$i; mkdir -p /tmp/kr_$i
  env SLN_SCENE_NO_SIDE=1 $v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kr_$i -o k -- python bench.py --no-cpu --no-check --no-dropin --large-batches= --no-graph-build --no-refine --no-sampling --no-spade --steps 3 --warmup 2 --prof-steps 0 --render-iters 40 --render-warmup 5 > /dev/null 2> /tmp/kr_$i/err
  f=$(find /tmp/kr_$i -name 'k_kernel_stats.csv' | head -1)
  echo "== [$v]"; python - "$f" "${KSTATS_FILTER:-raster|scene|pixel_map|depth_backward|project}" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:60]:
    if re.search(sys.argv[2], r['Name']): print("%-62s %6d calls  avg %8.2f us" % (r['Name'][:62], int(r['Calls']), float(r['AverageNs']) / 1e3))
PY
done
