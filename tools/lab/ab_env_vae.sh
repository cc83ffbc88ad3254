#!/bin/bash
# same-box A/B of environment settings on the VAE leg of the bench (64 graphs):  tools/lab/ab_env_vae.sh "SLN_NO_DEFER=1" "" "SLN_TN_MULTI_ROWS=512" ...
for v in "$@"; do
  env $v timeout 300 python bench.py --no-render --no-spade --no-graph-build --no-refine --no-cpu --large-batches= --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print('[$v]', d['ms_per_step'], d['ms_per_step_p10_p50_p90'] if 'ms_per_step_p10_p50_p90' in d else '', d['parity']['bench_batch_loss_rel_err'], {n:(k[n]['ms_per_step'], k[n]['avg_us']) for n in k})"
done
