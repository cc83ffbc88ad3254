#!/bin/bash
# as ab_env_vae.sh without the parity check (ablations that make the results garbage):  tools/lab/ab_env_vae_nocheck.sh "SLN_EDGE_ABL=7" ...
for v in "$@"; do
  env $v timeout 300 python bench.py --no-check --no-render --no-spade --no-graph-build --no-refine --no-sampling --no-cpu --no-dropin --large-batches= --steps 200 --warmup 20 --prof-steps 0 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', d['ms_per_step'], d['ms_per_step_p10_p50_p90'])
except Exception as e: print('[$v] failed', e)"
done
