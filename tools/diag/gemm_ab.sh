#!/bin/bash
# A/B of the 1-tap GEMM kernels on the step's shapes: gathered kernel (RD_WS=1) vs fragment-major weights from L2 (RD_WS=2)
cd ${GRAFT_REPO_ROOT:-.}
for shape in "8192 256 1024" "8192 1024 256" "8192 256 2304" "8192 2304 256" "32768 512 256" "32768 256 512" "2048 1024 256" "2048 256 1024"; do
  for ws in 1 2; do
    RD_WS=$ws python tools/diag/gemm_micro.py $shape --iters 50 2>&1 | tail -1
  done
done
