#!/bin/bash
# ablation of the scheduled NT loop (GPU box): tools/lab/abl.sh 0 1 2 4 8 15 ...   (bit mask, see SLN_ABL in gemm_bodies.h)
for v in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -w -DSLN_ABL=$v $SLN_LAB_EXTRA -I include -I 3d_sln_amd/csrc tools/lab/gemm_lab.hip -o /tmp/gemm_lab_$v 2>&1 | grep -i error
  echo "== SLN_ABL=$v $SLN_LAB_EXTRA"
  /tmp/gemm_lab_$v 2>&1 | grep -E "^M=|16x16|BatchNorm operand|sums touched|helper wavefronts|prologue pieces|kernarg|since the block|without column" | cut -c1-260
done
