cd /root/repo
timeout 900 python -m pytest tests/test_refine_gpu.py -x -q 2>&1 | tail -3
timeout 600 python tools/refine_batch_time.py 16 2>&1 | grep "eager" | cut -c1-70
SLN_REFINE_POOL_ONES=1 timeout 600 python tools/refine_batch_time.py 16 2>&1 | grep "eager" | cut -c1-70
