for v in "A=1" "OMP_WAIT_POLICY= GOMP_SPINCOUNT=30000" "OMP_WAIT_POLICY= GOMP_SPINCOUNT=300000" "OMP_WAIT_POLICY=ACTIVE GOMP_SPINCOUNT="; do
  env $v python bench.py --no-spade --no-graph-build --no-refine --no-sampling --no-dropin --large-batches= 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['render']; print('[$v] cpu vae', d['cpu_baseline']['value'], 'c1 cpu', d.get('c1',{}).get('cpu_baseline',{}).get('value'), 'render cpu', r['cpu_baseline']['value'], 'render mean/median', r['mean_over_median'], r['renders_per_s'], 'c2', d['value'])"
done
