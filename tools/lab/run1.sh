cd /root/repo
timeout 2500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
python bench.py > gpurun_out/bench_now.json 2>/dev/null; echo rc=$?
