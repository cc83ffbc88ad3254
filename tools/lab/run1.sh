cd /root/repo
timeout 900 python -m pytest tests/test_refine_gpu.py tests/test_raster_gpu.py -x -q 2>&1 | tail -3
timeout 600 python tools/refine_batch_time.py 16 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-120
timeout 600 python bench.py --no-cpu --no-check --no-dropin --large-batches= --no-graph-build --no-sampling --no-refine --no-spade --steps 3 --warmup 2 --prof-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['render']; print({k:r[k] for k in r if k in ('renders_per_s','ms_per_batch_p50','ms_per_batch','value')}, r.get('roofline',{}).get('frac'))"
