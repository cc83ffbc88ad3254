cd /root/repo
timeout 900 python -m pytest tests/test_refine_gpu.py tests/test_raster_gpu.py -x -q 2>&1 | tail -2
echo "8,16,16"; timeout 900 python tools/refine_batch_time.py 8,16,16 2>&1 | grep "eager\|graph:" | cut -c1-60
pick='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["refine"]; print(sys.argv[1], d["value"], r["rooms_16"]["ms_per_iteration"], r["rooms_64"]["ms_per_iteration"], r["ms_per_iteration"], d["render"]["renders_per_s"])'
python bench.py --large-batches= --no-cpu --no-check --no-graph-build --no-sampling --no-spade 2>/dev/null | python -c "$pick" "bench with dropin + render:"
python bench.py --large-batches= --no-cpu --no-check --no-graph-build --no-sampling --no-spade --no-dropin 2>/dev/null | python -c "$pick" "bench without dropin:"
GPU_MAX_HW_QUEUES=2 python bench.py --large-batches= --no-cpu --no-check --no-graph-build --no-sampling --no-spade --no-dropin 2>/dev/null | python -c "$pick" "2 hw queues:"
