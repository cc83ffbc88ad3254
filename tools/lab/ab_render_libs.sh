#!/bin/bash
# same-box A/B of builds of libsln_hip.so (tools/lab/lib_<X>.so): render leg of the bench (16 rooms), alternating
#   tools/lab/ab_render_libs.sh A B [A B ...]
cp 3d_sln_amd/libsln_hip.so /tmp/lib_keep.so
for v in "$@"; do
  cp tools/lab/lib_$v.so 3d_sln_amd/libsln_hip.so
  timeout 300 python bench.py --steps 5 --warmup 2 --no-spade --no-graph-build --no-refine --no-sampling --no-dropin --no-cpu --large-batches= 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['render']; print('$v', r['renders_per_s'], r['ms_per_batch_p10_p50_p90'], r['scene_forward']['avg_ms_per_batch'], r['scene_backward']['avg_ms_per_batch'], r['parity'])"
done
cp /tmp/lib_keep.so 3d_sln_amd/libsln_hip.so
