#!/bin/bash
# A/B of the dense 3x3 kernels on the step's shapes: LDS-staged weights (RD_WS=1) vs fragment-major weights from L2 (RD_WS=2)
cd ${GRAFT_REPO_ROOT:-.}
for shape in "8 64 64 256 256" "8 64 64 512 256" "8 32 32 256 256" "8 64 64 64 2688" "8 64 64 2688 64" "8 64 64 128 256" "8 128 128 256 256"; do
  for ws in 1 2; do
    RD_WS=$ws python tools/diag/d3_micro.py $shape --iters 50 2>&1 | tail -1
  done
done
