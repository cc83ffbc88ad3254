hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -w -I include -I 3d_sln_amd/csrc tools/lab/gemm_lab.hip -o /tmp/gemm_lab
echo "== default"; /tmp/gemm_lab 2>&1 | grep "^M=" | cut -c1-230
echo "== HIP_FORCE_DEV_KERNARG=1"; HIP_FORCE_DEV_KERNARG=1 /tmp/gemm_lab 2>&1 | grep "^M=" | cut -c1-230
echo "== HIP_FORCE_DEV_KERNARG=0"; HIP_FORCE_DEV_KERNARG=0 /tmp/gemm_lab 2>&1 | grep "^M=" | cut -c1-230
bash tools/lab/ab_env_vae.sh "" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "" "HIP_FORCE_DEV_KERNARG=1"
