cd /root/repo
timeout 900 python -m pytest tests/test_refine_gpu.py -x -q 2>&1 | tail -5
timeout 300 python tools/refine_batch_time.py 16 2>&1 | tail -4
SLN_REFINE_ALL_PLANES=1 timeout 300 python tools/refine_batch_time.py 16 2>&1 | tail -4
