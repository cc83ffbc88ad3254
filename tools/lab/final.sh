cd /root/repo
mkdir -p gpurun_out/profiles_r05
bash tools/profile_round.sh r05 > gpurun_out/profile_round.log 2>&1; echo profile rc=$?
timeout 2400 bash tools/round_extras.sh r05 > gpurun_out/round_extras.log 2>&1; echo extras rc=$?
timeout 900 python bench.py > gpurun_out/profiles_r05/r05_bench.json 2> gpurun_out/bench_final.err; echo bench rc=$?
