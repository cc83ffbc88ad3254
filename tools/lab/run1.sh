cd /root/repo
timeout 900 python -m pytest tests/test_spade_gpu.py -x -q 2>&1 | tail -3
python tools/lab/spade_b1.py 2>&1 | grep -v amdgpu
