#!/bin/bash
# same-box A/B of two builds of libsln_hip.so (tools/lab/lib_A.so, lib_B.so): SPADE leg of the bench, alternating
for v in A B A B; do
  cp tools/lab/lib_$v.so 3d_sln_amd/libsln_hip.so
  echo "== $v $*"
  env "$@" timeout 300 python bench.py --no-render --no-graph-build --no-refine --no-cpu --no-check --large-batches= --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['spade']; print(s['ms_per_batch'], s['conv_kernels']['ms'])"
done
