"""Two data-parallel ranks sharing the one GPU of the test box (gloo backend): the flat-buffer gradient all-reduce of the fused
optimizer vs DistributedDataParallel.  The 8-GPU RCCL run itself is the driver's; this covers the code path it takes."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("overlap,bucket_mb", [("0", "25"), ("1", "25"), ("1", "2")])
def test_flat_allreduce_matches_ddp_two_ranks(overlap, bucket_mb):
    """overlap 0: one pack + one all-reduce after backward; overlap 1: bucketed (4 buckets at 25 MB, ~40 at 2 MB), each bucket's
    all-reduce started from the hook of its last gradient while backward continues -- all three must equal the per-tensor all-reduce
    of the same gradients bit for bit."""
    env = dict(os.environ, RD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", RD_DDP_OVERLAP=overlap,
               RD_DDP_BUCKET_MB=bucket_mb)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_flat_check.py")]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DIST_FLAT_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_sync_batchnorm_two_ranks_equal_one_process_on_all_rows():
    """--sync_bn (tools/train.py:144-145): tests/dist_syncbn_check.py -- two ranks holding different shares of the rows, BatchNorm
    synchronised through rd_bn_train_fwd_sync / rd_bn_bwd_reduce / rd_bn_bwd_apply / rd_vfe_backward_{reduce,weight}, equal one process
    on the concatenated rows (outputs, running statistics, input gradients, summed parameter gradients); then two optimizer steps of
    the converted PillarNet keep the ranks' parameters bit-identical."""
    env = dict(os.environ, RD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_syncbn_check.py")]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DIST_SYNCBN_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.parametrize("overlap", ["0", "1"])
def test_rccl_one_rank_rehearsal_of_the_data_parallel_step(overlap):
    """RD_DP_REHEARSE=1 (dist.rehearsal): the bench's training loop with the complete data-parallel path -- RCCL process group, parameter
    broadcast, flat-buffer exchange (unbucketed by default, bucketed + overlapped with RD_DDP_OVERLAP=1), presence mask, clip + Adam from
    the flat buffer -- in a world of ONE rank on this GPU (two RCCL ranks cannot share a device).  In a world of one the exchange is
    the identity, so the loss after three optimizer steps must equal the plain loop's on the same batches; what the test is for is
    that every RCCL call, stream hand-over and work handle of the N > 1 path has executed on the hardware."""
    import json
    losses = {}
    for rehearse in ("0", "1"):
        env = dict(os.environ, RD_DP_REHEARSE=rehearse, RD_DDP_OVERLAP=overlap, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(_free_port()), RD_BENCH_NO_HOOKS="1")
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
            env.pop(k, None)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--batch", "2", "--steps", "3", "--warmup", "0", "--no-cpu-baseline",
               "--other-math-steps", "0", "--amp-steps", "0"]
        r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert r.returncode == 0 and lines, (r.stdout[-1500:], r.stderr[-3000:])
        out = json.loads(lines[-1])
        assert out["n_gpus"] == 1 and out["steps"] == 3
        losses[rehearse] = out["config"]["final_loss"]
    import math
    assert math.isfinite(losses["1"]) and abs(losses["1"] - losses["0"]) <= 2e-3 * abs(losses["0"]), losses
