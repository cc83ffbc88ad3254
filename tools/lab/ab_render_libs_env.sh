#!/bin/bash
# same-box A/B of builds (tools/lab/lib_<X>.so) x environment settings on the render leg:  tools/lab/ab_render_libs_env.sh "A D" "SLN_X=0" "SLN_SCENE_NO_SIDE=1"
libs=$1; shift
cp 3d_sln_amd/libsln_hip.so /tmp/lib_keep.so
for rep in 1 2; do
for v in $libs; do
  cp tools/lab/lib_$v.so 3d_sln_amd/libsln_hip.so
  for e in "$@"; do
  env $e timeout 300 python bench.py --steps 5 --warmup 2 --no-spade --no-graph-build --no-refine --no-sampling --no-dropin --no-cpu --large-batches= 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['render']; print('$v [$e]', r['renders_per_s'], r['ms_per_batch_p10_p50_p90'], r['scene_forward']['avg_ms_per_batch'], r['scene_backward']['avg_ms_per_batch'])"
  done
done
done
cp /tmp/lib_keep.so 3d_sln_amd/libsln_hip.so
