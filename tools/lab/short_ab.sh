for v in "A=1" "GOMP_SPINCOUNT=0 OMP_WAIT_POLICY=PASSIVE" "A=2" "GOMP_SPINCOUNT=300000"; do
  env $v python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['summary']; print('[$v]', s['c2_ms_step'], d['ms_per_step_p10_p50_p90'], 'survey', d['survey_protocol_20_100']['ms_per_step_p10_p50_p90'], 'c3', s['c3_renders_per_s'], s['c3_renders_per_s_median'], 'cpu', s['cpu_graphs_per_s'])"
done
