cd /root/repo
timeout 2400 bash tools/round_extras.sh r05 > gpurun_out/round_extras.log 2>&1; echo extras rc=$?
timeout 900 python bench.py > gpurun_out/profiles_r05/r05_bench.json 2> gpurun_out/bench_final.err; echo bench rc=$?
tail -2 gpurun_out/bench_final.err
