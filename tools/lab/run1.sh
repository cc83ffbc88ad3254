cd /root/repo
timeout 300 python tools/refine_batch_time.py 16,64 2>&1 | tail -8 | cut -c1-100
