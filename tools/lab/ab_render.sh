#!/bin/bash
# same-box A/B of the render leg: env settings given as arguments are applied to the B runs (e.g. SLN_SCENE_NO_SIDE=1)
for v in A B A B; do
  if [ $v = A ]; then E=""; else E="$*"; fi
  echo "== $v $E"
  env $E timeout 300 python bench.py --steps 5 --warmup 2 --no-spade --no-graph-build --no-refine --no-cpu --large-batches= 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['render']; print(r['renders_per_s'], r['ms_per_batch_p10_p50_p90'], r['scene_forward']['avg_ms_per_batch'], r['scene_backward']['avg_ms_per_batch'], r['parity'])"
done
