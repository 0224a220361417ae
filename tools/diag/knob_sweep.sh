#!/bin/bash
# Diagnostic (GPU box): one bench run per "VAR=value" argument (plus the defaults first and last); prints samples/s and ms per step.
#   bash tools/diag/knob_sweep.sh RD_BIG_TILES=256 RD_GEMM_TILE64=0 ...
run() {
    env "$@" python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --other-math-steps 0 2>/dev/null |
        python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'])"
}
run RD_NOP=1
for kv in "$@"; do run $kv; done
run RD_NOP=1
