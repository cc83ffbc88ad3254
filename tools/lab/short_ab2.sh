run() { python bench.py "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['render']; print('[$*]', d['ms_per_step'], 'c3 mean', r['renders_per_s'], 'median', r['renders_per_s_median'], r['ms_per_batch_p10_p50_p90'], r['wall_mean_ms_per_batch'])"; }
run --steps 20 --warmup 5 --no-spade --no-graph-build --no-refine --no-sampling --no-dropin --large-batches=
run --steps 20 --warmup 5 --no-cpu --no-spade --no-graph-build --no-refine --no-sampling --no-dropin --large-batches=
run --steps 20 --warmup 5 --no-check --no-spade --no-graph-build --no-refine --no-sampling --no-dropin --large-batches=
run --steps 100 --warmup 5 --no-spade --no-graph-build --no-refine --no-sampling --no-dropin --large-batches=
run --steps 200 --warmup 20 --no-spade --no-graph-build --no-refine --no-sampling --no-dropin --large-batches=
run --steps 20 --warmup 5 --prof-steps 0 --no-spade --no-graph-build --no-refine --no-sampling --no-dropin --large-batches=
