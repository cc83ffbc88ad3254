#!/bin/bash
# same-box A/B of builds of libsln_hip.so (tools/lab/lib_<X>.so): refinement leg of the bench (one room, 16 and 64 rooms in flight), alternating
cp 3d_sln_amd/libsln_hip.so /tmp/lib_keep.so
for v in "$@"; do
  cp tools/lab/lib_$v.so 3d_sln_amd/libsln_hip.so
  timeout 300 python bench.py --steps 5 --warmup 2 --no-spade --no-graph-build --no-render --no-cpu --no-sampling --no-dropin --large-batches= 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['refine']; s=d['summary']; print('$v', 'one room', r['ms_per_iteration'], '16 rooms', s.get('refine16_ms'), '64 rooms', s.get('refine64_ms'), 'parity', s.get('refine_err'))"
done
cp /tmp/lib_keep.so 3d_sln_amd/libsln_hip.so
