#!/bin/bash
# same-box A/B of builds of libsln_hip.so (tools/lab/lib_<X>.so) on the large-batch points of the bench
#   tools/lab/ab_large.sh "1024,4096" A B A B
S=$1; shift
cp 3d_sln_amd/libsln_hip.so /tmp/lib_keep.so
for v in "$@"; do
  cp tools/lab/lib_$v.so 3d_sln_amd/libsln_hip.so
  timeout 600 python bench.py --no-render --no-spade --no-graph-build --no-refine --no-sampling --no-cpu --no-check --no-dropin --large-batches=$S --steps 20 --warmup 5 --prof-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], {k:(v['ms_per_step'], v['frac_mfma_whole_step']) for k,v in d['vae_large_batch'].items()})"
done
cp /tmp/lib_keep.so 3d_sln_amd/libsln_hip.so
