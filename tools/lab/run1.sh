cd /root/repo
timeout 1500 python -m pytest tests/test_refine_gpu.py -x -q 2>&1 | tail -3
SLN_REFINE_SETUP_LOG=1 ITERS=5 timeout 900 python tools/refine_batch_time.py 16 2>&1 | grep "set-up:" | tail -6
timeout 900 python tools/refine_batch_time.py 16,64 2>&1 | grep "eager" | cut -c1-130
