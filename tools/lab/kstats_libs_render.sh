#!/bin/bash
# per-kernel average durations of the render leg (16 rooms) for builds of libsln_hip.so (tools/lab/lib_<X>.so), same box, one stream
# and with the side stream:  tools/lab/kstats_libs_render.sh A B
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp 3d_sln_amd/libsln_hip.so /tmp/lib_keep.so
for v in "$@"; do
  cp tools/lab/lib_$v.so 3d_sln_amd/libsln_hip.so
  for side in 1 0; do
    rm -rf /tmp/ks_$v; mkdir -p /tmp/ks_$v
    if [ $side = 0 ]; then export SLN_SCENE_NO_SIDE=1; else unset SLN_SCENE_NO_SIDE; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$v -o k -- python bench.py --no-cpu --no-check --no-dropin --large-batches= --no-graph-build --no-refine --no-sampling --no-spade --steps 3 --warmup 2 --prof-steps 0 --render-iters 40 --render-warmup 5 > /dev/null 2> /tmp/ks_$v/err
    f=$(find /tmp/ks_$v -name 'k_kernel_stats.csv' | head -1)
    echo "== $v side=$side"; python - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = 0
for r in rows[:40]:
    if re.search("raster|scene|pixel_map|depth_backward|project", r['Name']):
        print("%-62s %6d calls  avg %8.2f us" % (r['Name'][:62], int(r['Calls']), float(r['AverageNs']) / 1e3)); tot += float(r['AverageNs']) / 1e3
print("sum of averages %.1f us" % tot)
PY
  done
done
unset SLN_SCENE_NO_SIDE
cp /tmp/lib_keep.so 3d_sln_amd/libsln_hip.so
