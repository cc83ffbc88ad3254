#!/bin/bash
# The two small text files of profiles/<tag>_*: cost of SLN_DETERMINISTIC=1 and the data-parallel code path on one GPU.
#   tools/round_extras.sh r03        (GPU box, repository root)
TAG=${1:-r06}
V="--no-render --no-spade --no-graph-build --no-refine --no-sampling --no-cpu --no-dropin --large-batches= --steps 200 --warmup 20"
R="--no-spade --no-graph-build --no-refine --no-sampling --no-cpu --no-check --no-dropin --large-batches= --steps 3 --warmup 2 --prof-steps 0"
pick_vae='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["kernels"]; print("%-18s %.4f  %s   gemm_tn %.3f ms per step (%d launches)" % (sys.argv[1], d["ms_per_step"], d["ms_per_step_p10_p50_p90"], k["gemm_tn"]["ms_per_step"], k["gemm_tn"]["launches_per_step"]))'
pick_rnd='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])["render"]; print("%-18s %s   forward %.3f, backward %.3f" % (sys.argv[1], d["ms_per_batch_p10_p50_p90"], d["scene_forward"]["avg_ms_per_batch"], d["scene_backward"]["avg_ms_per_batch"]))'
{
  echo "# Cost of SLN_DETERMINISTIC=1 on one MI355X, same box, same build (tools/round_extras.sh)"
  echo "# scene-graph VAE, fused training step, 64 graphs x (32 objects, 64 triples), hipGraph replay, ms per step [p10, p50, p90]"
  python bench.py $V 2>/dev/null | python -c "$pick_vae" default
  SLN_DETERMINISTIC=1 python bench.py $V 2>/dev/null | python -c "$pick_vae" SLN_DETERMINISTIC
  echo "# fused scene pass, 16 rooms x 2k triangles x 256^2, forward + backward, ms per batch [p10, p50, p90]"
  python bench.py $R 2>/dev/null | python -c "$pick_rnd" default
  SLN_DETERMINISTIC=1 python bench.py $R 2>/dev/null | python -c "$pick_rnd" SLN_DETERMINISTIC
  echo "# SPADEGenerator4, batch 32, 256x256, ms per batch (tools/spade_time.py; deterministic: LayerNorm / pooling sums by fixed-order kernels)"
  python tools/spade_time.py 6 2>/dev/null | grep "ms per batch"
  SLN_DETERMINISTIC=1 python tools/spade_time.py 6 2>/dev/null | grep "ms per batch"
  echo "# bit-identity: tests/test_train_gpu.py::test_deterministic_mode_makes_fused_steps_bit_identical[feedforward|recurrent],"
  echo "#               tests/test_spade_gpu.py::test_deterministic_mode_makes_the_generator_bit_identical_run_to_run,"
  echo "#               tests/test_raster_gpu.py::test_deterministic_mode_makes_the_scene_pass_bit_identical[1|9],"
  echo "#               tests/test_train_gpu.py::test_eager_steps_on_batches_of_changing_shape_need_no_host_sync"
} > profiles/${TAG}_deterministic.txt
pick_dp='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({k: (d[k] if k in d else d["config"].get(k)) for k in ("value","ms_per_step","ms_per_step_p10_p50_p90","n_gpus","collective","allreduce_us_standalone")}))'
{
  echo "# SLN_BENCH_FORCE_DP=1 python bench.py --no-cpu --no-render --no-spade --no-graph-build --no-refine --no-sampling --no-dropin --large-batches=   (one MI355X, world of one: the collective path of the 8-GPU run)"
  for e in "SLN_X=0" "SLN_BENCH_FORCE_DP=1" "SLN_BENCH_FORCE_DP=1 SLN_DP_OVERLAP=1"; do
    echo "## $e"
    env $e python bench.py --no-cpu --no-render --no-spade --no-graph-build --no-refine --no-sampling --no-dropin --large-batches= 2>/dev/null | python -c "$pick_dp"
  done
  echo "# batches of changing shape, eager launches (tools/varshape_time.py)"
  python tools/varshape_time.py 2>/dev/null | tail -4
} > profiles/${TAG}_force_dp.txt
# the R-rooms-in-flight refinement loop (round 5): per-kernel totals of a 16-room run, one iteration's timeline, ms per iteration by R
{
  echo "# tools/refine_batch_profile.sh 16: rocprofv3 kernel trace of tools/refine_batch_time.py 16 (VAE over-fitted to 64 rooms first: its"
  echo "# training kernels are in the totals; the refinement's are the *_multi / *_rooms / scene / loss rows; 540 iterations)"
  bash tools/refine_batch_profile.sh 16 2>&1 | grep -v amdgpu.ids
} > profiles/${TAG}_refine_batch_kernel_stats.txt
{
  echo "# tools/refine_batch_timeline.sh 16: ONE iteration of the 16-room loop under rocprofv3 --kernel-trace, kernels in start order"
  echo "# (queue, start offset, duration, gap to the previous kernel of the same queue); q2 = caller's stream, the others = side streams"
  bash tools/refine_batch_timeline.sh 16 2>&1 | grep -v amdgpu.ids
} > profiles/${TAG}_refine_batch_timeline.txt
{
  echo "# tools/refine_batch_time.py 1,2,4,8,16,32,64: ms per iteration (slope between runs of 60 and 120 iterations), eager and hipGraph replay;"
  echo "# second block: the same with every plane processed, the stand-alone SGD step and separate head launches (the round-5 switches off)"
  python tools/refine_batch_time.py 1,2,4,8,16,32,64 2>&1 | grep -v amdgpu.ids
  echo "## SLN_REFINE_ALL_PLANES=1 SLN_REFINE_SEPARATE_SGD=1 SLN_REFINE_SEPARATE_HEAD=1"
  SLN_REFINE_ALL_PLANES=1 SLN_REFINE_SEPARATE_SGD=1 SLN_REFINE_SEPARATE_HEAD=1 python tools/refine_batch_time.py 16 2>&1 | grep -v amdgpu.ids
} > profiles/${TAG}_refine_batch_by_rooms.txt
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_deterministic.txt profiles/${TAG}_force_dp.txt profiles/${TAG}_refine_batch_*.txt gpurun_out/profiles_$TAG/
cat profiles/${TAG}_deterministic.txt profiles/${TAG}_force_dp.txt
