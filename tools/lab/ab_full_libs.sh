#!/bin/bash
# same-box A/B of builds of libsln_hip.so (tools/lab/lib_<X>.so): the DEFAULT bench sequence (all legs, no CPU baselines)
cp 3d_sln_amd/libsln_hip.so /tmp/lib_keep.so
for v in "$@"; do
  cp tools/lab/lib_$v.so 3d_sln_amd/libsln_hip.so
  timeout 400 python bench.py --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'vae', d['ms_per_step'], 'render', d['render']['ms_per_batch_p10_p50_p90'][1], 'spade', d['spade']['ms_per_batch'], 'refine', d['refine']['ms_per_iteration'], 'graph', d['graph_build']['us_per_batch'])"
done
cp /tmp/lib_keep.so 3d_sln_amd/libsln_hip.so
