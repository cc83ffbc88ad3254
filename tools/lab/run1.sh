cd /root/repo
timeout 900 python -m pytest tests/test_refine_gpu.py -x -q 2>&1 | tail -5
timeout 300 python tools/refine_batch_time.py 16 2>&1 | tail -2
SLN_REFINE_SEPARATE_SGD=1 timeout 300 python tools/refine_batch_time.py 16 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_vae_gpu.py tests/test_train_gpu.py -x -q 2>&1 | tail -3
