#!/bin/bash
# same-box A/B of environment settings on the SPADE leg of the bench (batch 32):  tools/lab/ab_env_spade.sh "SLN_X=0" "SLN_SPADE_XCD=1" ...
for v in "$@"; do
  env $v timeout 600 python bench.py --steps 3 --warmup 2 --legs-only --no-render --no-graph-build --no-refine --no-sampling --no-cpu --no-dropin --no-colorize --large-batches= --prof-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['spade']; print('[$v]', r.get('images_per_s'), r.get('ms_per_batch'), r.get('parity'))"
done
