#!/bin/bash
# same-box A/B of builds of libsln_hip.so (tools/lab/lib_<X>.so): VAE leg of the bench (64 graphs), alternating
#   tools/lab/ab_vae.sh A B [A B ...]
cp 3d_sln_amd/libsln_hip.so /tmp/lib_keep.so
for v in "$@"; do
  cp tools/lab/lib_$v.so 3d_sln_amd/libsln_hip.so
  timeout 300 python bench.py --no-render --no-spade --no-graph-build --no-refine --no-cpu --large-batches= --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print('$v', d['ms_per_step'], d['parity']['bench_batch_loss_rel_err'], {n:(k[n]['ms_per_step'], k[n]['avg_us']) for n in k})"
done
cp /tmp/lib_keep.so 3d_sln_amd/libsln_hip.so
