#!/bin/bash
# same-box A/B of the refinement leg (one room): env settings given as arguments are applied to the B runs
for v in A B A B; do
  if [ $v = A ]; then E=""; else E="$*"; fi
  env $E timeout 300 python bench.py --steps 5 --warmup 2 --no-spade --no-graph-build --no-render --no-cpu --large-batches= 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['refine']; print('$v $E', r['ms_per_iteration'])"
done
