cd /root/repo
for r in 1 2 4; do
echo "R=$r side"; timeout 600 python tools/refine_batch_time.py $r 2>&1 | grep "eager\|graph:" | cut -c1-70
echo "R=$r no side"; SLN_GROUP_NO_SIDE=1 timeout 600 python tools/refine_batch_time.py $r 2>&1 | grep "eager\|graph:" | cut -c1-70
done
