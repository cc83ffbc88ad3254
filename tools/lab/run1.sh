cd /root/repo
timeout 900 python -m pytest tests/test_refine_gpu.py tests/test_raster_gpu.py -x -q 2>&1 | tail -5
timeout 300 python tools/refine_batch_time.py 16 2>&1 | tail -2
