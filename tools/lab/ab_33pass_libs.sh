#!/bin/bash
# same-box A/B of builds (tools/lab/lib_<X>.so): the 33-pass drop-in renderer leg (one room through nr.Renderer), alternating
cp 3d_sln_amd/libsln_hip.so /tmp/lib_keep.so
for v in "$@"; do
  cp tools/lab/lib_$v.so 3d_sln_amd/libsln_hip.so
  timeout 300 python bench.py --steps 5 --warmup 2 --no-spade --no-graph-build --no-refine --no-sampling --no-cpu --no-check --large-batches= 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['render']['render_33pass']; print('$v', {k: (v['ms_per_render'] if isinstance(v, dict) else v) for k, v in r.items() if k != 'workload'})"
done
cp /tmp/lib_keep.so 3d_sln_amd/libsln_hip.so
