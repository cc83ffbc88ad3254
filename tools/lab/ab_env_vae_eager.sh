#!/bin/bash
# same-box A/B of environment settings on the VAE leg, bench flags after "--":  tools/lab/ab_env_vae_eager.sh "SLN_TN_SIDE=1" ... -- --no-graph
envs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
for v in "${envs[@]}"; do
  env $v timeout 300 python bench.py --no-render --no-spade --no-graph-build --no-refine --no-sampling --no-dropin --no-cpu --large-batches= --steps 200 --warmup 20 --prof-steps 0 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v $*]', d['ms_per_step'], d['ms_per_step_p10_p50_p90'], d['parity']['bench_batch_loss_rel_err'])"
done
