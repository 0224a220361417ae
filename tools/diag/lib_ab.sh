#!/bin/bash
# A/B of two builds of the library inside ONE GPU call (boxes differ by a few percent): tools/diag/micro/librdamd_prev.so against the tree's
cd ${GRAFT_REPO_ROOT:-.}
cp radardistill_amd/csrc/librdamd.so /tmp/new.so
run() {
  for shape in "8 64 64 256 256" "8 64 64 512 256" "8 128 128 256 256" "8 32 32 256 256" "8 32 32 128 128" "8 32 32 384 128"; do
    RD_WS=2 python tools/diag/d3_micro.py $shape --iters 50 2>&1 | tail -1 | cut -c1-90 | sed "s/^/$1 /"
  done
  for shape in "8192 256 1024" "8192 1024 256" "32768 512 256" "32768 256 512"; do
    RD_WS=2 python tools/diag/gemm_micro.py $shape --iters 50 2>&1 | tail -1 | cut -c1-90 | sed "s/^/$1 /"
  done
  for shape in "8 64 64 0.78 256 256" "8 128 128 0.57 128 128"; do
    RD_WS=2 python tools/diag/sparse_micro.py $shape --iters 50 2>&1 | tail -1 | cut -c1-120 | sed "s/^/$1 /"
  done
}
for rep in 1 2; do
  cp /tmp/new.so radardistill_amd/csrc/librdamd.so; run new
  cp tools/diag/micro/librdamd_prev.so radardistill_amd/csrc/librdamd.so; run old
done
cp /tmp/new.so radardistill_amd/csrc/librdamd.so
