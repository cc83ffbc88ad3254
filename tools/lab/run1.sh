cd /root/repo
echo "default"; timeout 600 python tools/refine_batch_time.py 16 2>&1 | grep "eager" | cut -c1-70
echo "all planes"; SLN_REFINE_ALL_PLANES=1 timeout 600 python tools/refine_batch_time.py 16 2>&1 | grep "eager" | cut -c1-70
echo "all planes, separate sgd, separate head"; SLN_REFINE_ALL_PLANES=1 SLN_REFINE_SEPARATE_SGD=1 SLN_REFINE_SEPARATE_HEAD=1 timeout 600 python tools/refine_batch_time.py 16 2>&1 | grep "eager" | cut -c1-70
