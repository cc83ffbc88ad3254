#!/bin/bash
# same-box A/B of environment settings on the render leg of the bench (16 rooms):  tools/lab/ab_env_render.sh "SLN_X=1" "SLN_COMPOSE_VARIANT=1" ...
for v in "$@"; do
  env $v timeout 300 python bench.py --steps 3 --warmup 2 --no-spade --no-graph-build --no-refine --no-sampling --no-cpu --no-dropin --large-batches= --prof-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['render']; print('[$v]', r['ms_per_batch_p10_p50_p90'], r['renders_per_s'], r.get('scene_forward',{}).get('avg_ms_per_batch'), r.get('scene_backward',{}).get('avg_ms_per_batch'), r['parity']['face_index_pixels_differing'])"
done
