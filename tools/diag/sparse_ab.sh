#!/bin/bash
# A/B of the sparse (neighbour-table) convolution kernels on shapes like the step's: gathered kernel (RD_WS=1) vs k_gemm_b3f<.., TABLE> (RD_WS=2)
cd ${GRAFT_REPO_ROOT:-.}
for shape in "8 64 64 0.52 256 256" "8 64 64 0.78 256 256" "8 128 128 0.20 128 128" "8 128 128 0.57 128 128" "8 256 256 0.057 64 64" "8 256 256 0.32 64 64"; do
  for ws in 1 2; do
    RD_WS=$ws python tools/diag/sparse_micro.py $shape --iters 50 2>&1 | tail -1
  done
done
