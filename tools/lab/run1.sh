cd /root/repo
SLN_BENCH_DEBUG=1 timeout 1500 python bench.py > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err; echo bench rc=$?
grep "refine debug" gpurun_out/bench_now.err | tail -2
