cd /root/repo
timeout 900 python tools/refine_batch_time.py 1,2,4,8,32 2>&1 | grep "eager" | cut -c1-75
