#!/bin/bash
# GPU box: regenerate the round's evidence under gpurun_out/prof_refresh/ (copy into profiles/ afterwards):
#   the driver's bench command, rocprofv3 kernel statistics of the same command, the two HBM-traffic PMC passes and the SQ pass.
# rocprofv3 runs get the program itself after `--` (python3 bench.py ...), PMC passes are separate from --stats.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_refresh
mkdir -p $O
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.log 2>&1 || exit 1
echo "driver command done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_stdout.log 2>&1 || exit 2
echo "kernel stats done"
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --other-math-steps 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o f -- $B > $O/pmc_f.log 2>&1 || exit 3
echo "FETCH_SIZE pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o w -- $B > $O/pmc_w.log 2>&1 || exit 4
echo "WRITE_SIZE pass done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES \
    --output-format csv -d $O/sq -o s -- $B > $O/sq.log 2>&1 || exit 5
echo "SQ pass done"
cd $R
python tools/diag/pmc_to_json.py $O/pmc_f $O/pmc_w $O/pmc_hbm_traffic.json || exit 6
python tools/diag/pmc_sq_to_json.py $O/sq $O/pmc_mfma_busy.json || exit 7
# the counter CSVs are large: keep the summaries only
rm -rf $O/pmc_f $O/pmc_w $O/sq
find $O/stats -name "*kernel_trace.csv" -delete
ls -la $O $O/stats
