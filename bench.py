#!/usr/bin/env python
"""bench.py -- RadarDistill training hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (config.workload): BASELINE.json configs[3] -- the full RadarDistill training step (frozen LiDAR teacher forward,
radar student forward + backward incl. CMA / AFD / PFD / CenterHead losses, grad clip + Adam) on synthetic nuScenes-shaped
sweeps: 512 x 512 BEV (0.2 m pillars), 35k LiDAR + 2k radar points and 30 boxes per sample, 8 samples per GPU, fp32.
A "step" = one such training iteration on one batch; inputs are resident in HBM before the timed region.  One process per
GPU; N > 1 shards samples over ranks (weak scaling) with DDP gradient all-reduce over RCCL.
Prints ONE JSON line on rank 0 (metric: samples/sec, whole job).

`--gpus N` with N > 1 and no torchrun environment starts the N rank processes itself (child processes, one per GPU, before this
process touches the GPU -- the role of tools/scripts/dist_train.sh:10 / torch_train.sh:17 of the reference) and relays rank 0's
JSON line; under torchrun WORLD_SIZE must equal --gpus (anything else is an error, never a silent 1-GPU run).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 matrix peak (no xf32 on gfx950)
PEAK_HBM_GBS = 8000.0             # same guide: HBM3E ~8 TB/s
PEAK_BF16_MFMA_TFLOPS = 2500.0    # same guide: dense bf16 matrix peak (the 5 PF headline figure includes 2:1 sparsity)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="samples per GPU (BATCH_SIZE_PER_GPU of the reference yaml)")
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sync-bn", action="store_true", help="the reference's --sync_bn (tools/train.py:34,144-145): synchronised BatchNorm "
                    "statistics over the ranks (off in the headline, as in the reference's default)")
    ap.add_argument("--amp-steps", type=int, default=-1, help="steps timed under the reference's --use_amp arithmetic (train.autocast: bf16 "
                    "products, fp32 accumulate + AmpScaler loss scaling) after the other legs, reported as 'amp' BESIDE the fp32-class "
                    "headline (-1 = the same count as --steps, 0 = skip)")
    ap.add_argument("--math", choices=["f32", "bf16x3"], default="bf16x3",
                    help="arithmetic of the implicit-GEMM conv kernels: bf16x3 = fp32 operands split into bf16 hi+lo, three bf16 MFMAs per "
                         "product, fp32 accumulate (~4e-6 relative error, inside the 1e-3 parity bound; parity-tested in "
                         "tests/test_gpu_model.py::test_bf16x3_conv_math_parity); f32 = exact fp32 MFMA")
    ap.add_argument("--other-math-steps", type=int, default=-1, help="steps timed in the other arithmetic mode after the timed region "
                    "(reported as 'other_math'; -1 = the same count as --steps, 0 = skip)")
    ap.add_argument("--cpu-baseline-grid", type=int, default=512)
    ap.add_argument("--cpu-baseline-batch", type=int, default=1)
    return ap.parse_args()


def spawn_ranks(n, argv):
    """Start `n` rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, as torchrun sets them),
    wait for all of them and return the worst exit code.  The parent never initialises the GPU; rank 0 inherits stdout and prints
    the JSON line, every rank inherits stderr.  A rank that dies takes the others down (they would wait in a collective forever)."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if port is None:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc, alive = 0, list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in alive:          # the exact children started above, by handle
                    q.terminate()
        time.sleep(0.05)
    return rc


def check_world(args_gpus, world):
    """--gpus is a contract with the caller: the run must use exactly that many ranks."""
    if args_gpus != world:
        raise SystemExit(f"bench.py: --gpus {args_gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args_gpus} "
                         f"(or without torchrun: bench.py starts the ranks itself)")


def build(cfg_path, grid, device):
    from radardistill_amd.data import SyntheticDistillDataset
    from radardistill_amd.pcdet.config import AttrDict, cfg_from_yaml_file
    from radardistill_amd.pcdet.models import build_network
    from radardistill_amd.synthetic import bench_geometry
    cfg = cfg_from_yaml_file(cfg_path, AttrDict())
    pc_range, voxel, gs = bench_geometry(grid)
    cfg.DATA_CONFIG.POINT_CLOUD_RANGE = pc_range
    cfg.MODEL.RADAR_BACKBONE_2D.POINT_CLOUD_RANGE = pc_range
    ds = SyntheticDistillDataset.from_cfg(cfg)
    torch.manual_seed(0)
    model = build_network(model_cfg=cfg.MODEL, num_class=len(cfg.CLASS_NAMES), dataset=ds)
    # non-trivial eval-mode BatchNorm for the frozen teacher: mean N(0, 0.1), var U(0.5, 1.5)
    g = torch.Generator().manual_seed(1)
    for name, buf in model.named_buffers():
        if name.endswith("running_mean"):
            buf.copy_(torch.randn(buf.shape, generator=g) * 0.1)
        elif name.endswith("running_var"):
            buf.copy_(torch.rand(buf.shape, generator=g) + 0.5)
    return model.to(device), cfg, (pc_range, voxel, gs)


def device_batch(batch, device):
    out = {}
    for k, v in batch.items():
        if isinstance(v, np.ndarray):
            out[k] = torch.from_numpy(v).float().to(device)
        else:
            out[k] = v
    out["gt_boxes_host"] = batch["gt_boxes"]
    if torch.device(device).type == "cuda":
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        out["_inputs_ready"] = ev              # inputs are resident: the detector's geometry prelude need not wait for the main stream
    return out


def _median_time(fn, warm=3, reps=10, budget_s=40.0):
    """SURVEY 8(d) / BASELINE.md section 3 protocol: `warm` untimed calls, then the median of `reps` timed ones.  `budget_s` bounds
    the whole measurement (the default bench run must end within minutes): when it runs out the repetitions stop early, never below
    3 timed calls; what was actually run is reported."""
    t_start = time.perf_counter()
    for _ in range(warm):
        fn()
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > budget_s and len(times) >= 3:
            break
    return float(np.median(times)), len(times)


def cpu_baseline(grid, B=1):
    """The CPU oracle (kind "port": the reference itself cannot run here -- spconv / torch_scatter / DCN are absent) timed
    on this host's cores: (1) the C4-shaped training step at B samples -- teacher forward, student forward + backward, clip +
    decoupled-decay Adam -- and (2) BASELINE configs[0] (C1): radar-only PillarNet forward, 1000 radar points, 128 x 128 BEV.
    Warm-up 3 + median of 10 each.  Reported, never the target."""
    from oracle import optim as ooptim
    from oracle import pillarnet as opn
    from radardistill_amd.synthetic import make_batch
    cores = min(len(os.sched_getaffinity(0)), 16)      # the GPU box gives a 16-core share; cpu_count() reports the whole host
    torch.set_num_threads(cores)
    model, cfg, (pc_range, voxel, gs) = build(os.path.join(ROOT, "tools/cfgs/radar_distill/bench_512.yaml"), grid, "cpu")
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    names = [k for k, p in model.named_parameters() if p.requires_grad]
    for k in names:
        state[k].requires_grad_(True)
    params = [state[k] for k in names]
    m1 = [torch.zeros_like(p) for p in params]; m2 = [torch.zeros_like(p) for p in params]
    b = make_batch(batch_size=B, n_lidar=35000, n_radar=2000, n_boxes=30, grid=grid, seed=0)
    ob = {"points": torch.from_numpy(b["points"]), "radar_points": torch.from_numpy(b["radar_points"]),
          "gt_boxes": torch.from_numpy(b["gt_boxes"]), "batch_size": B}
    it = [0]

    def train_step():
        it[0] += 1
        for p in params:
            p.grad = None
        loss, _, _ = opn.forward_train(state, ob, pc_range, voxel, gs, run_teacher_head=True)
        loss.mean().backward()
        _, clipped = ooptim.clip_grad_norm([p.grad for p in params], 10.0)
        lr, mom = ooptim.one_cycle(it[0], 1000)
        with torch.no_grad():
            ooptim.adam_true_wd_step([p.data for p in params], clipped, m1, m2, it[0], lr, mom)

    dt, n = _median_time(train_step)
    out = {"value": round(B / dt, 4), "unit": "samples/sec", "cores": cores, "kind": "port",
           "sample": f"C4 shape at B={B}: full training step (teacher fwd incl. head, student fwd/bwd, clip + Adam) at {grid}x{grid} BEV, 35k LiDAR + 2k "
                     f"radar points, 30 boxes; 3 warm-up + median of {n} steps = {dt:.2f} s per step"}
    # C1: radar-only PillarNet forward (eval), G = 128, 1000 radar points, B = 1
    model1, _, (pcr1, vox1, gs1) = build(os.path.join(ROOT, "tools/cfgs/radar_distill/bench_512.yaml"), 128, "cpu")
    st1 = {k: v.detach() for k, v in model1.state_dict().items()}
    b1 = make_batch(batch_size=1, n_lidar=16, n_radar=1000, n_boxes=1, grid=128, seed=0)
    rp = torch.from_numpy(b1["radar_points"])
    with torch.no_grad():
        dt1, n1 = _median_time(lambda: opn.forward_radar_only(st1, rp, 1, pcr1, vox1, gs1), budget_s=15.0)
    out["c1_radar_only_forward"] = {"value": round(1.0 / dt1, 3), "unit": "samples/sec", "ms": round(dt1 * 1e3, 2),
                                    "sample": f"BASELINE configs[0]: radar-only PillarNet forward, 1000 radar points, 128x128 BEV, B=1; 3 warm-up + median of {n1}"}
    return out


def kernel_names(b3):
    """instantiation tag (kernels._kernel_tag / conv_wgrad) -> (description, rocprofv3 kernel-name prefix)."""
    names = {
        "d3f_128": ("k_conv_d3f_b3<8,16,128> (dense stride-1 3x3 conv, forward and data gradient: 8x16-pixel halo double-buffered in LDS, weight "
                    "fragments straight from L2 in fragment-major split format, one barrier per 32-channel chunk)", "k_conv_d3f_b3<8, 16, 128"),
        "d3f_64": ("k_conv_d3f_b3<8,16,64> (the same kernel with a 64-channel column tile: the 8192-row maps)", "k_conv_d3f_b3<8, 16, 64"),
        "gemmf_128": ("k_gemm_b3f<128,128> (1-tap GEMM: nn.Linear / 1x1 conv / DCNv2 column GEMM; activations split once per 64-channel chunk in "
                      "LDS, weight fragments straight from L2)", "k_gemm_b3f<128, 128, false"),
        "sparsef_128": ("k_gemm_b3f<128,*,table> (sparse convolution over a neighbour table: gathered 64-channel chunks split once in LDS, weight "
                        "fragments straight from L2)", "k_gemm_b3f<128, 128, true"),
        "sparsef_64": ("k_gemm_b3f<64,*,table> (sparse convolution over a neighbour table, 64-row tiles)", "k_gemm_b3f<64, 128, true"),
        "gemmf_64": ("k_gemm_b3f<64,*> (1-tap GEMM, 64-row tiles)", "k_gemm_b3f<64, 128, false"),
        "d3_128": ("k_conv_d3_b3<8,16,128> (dense stride-1 3x3 conv, forward and data gradient: halo-staged 8x16-pixel tile)", "k_conv_d3_b3<8, 16, 128"),
        "d3_16x64": ("k_conv_d3_b3<8,16,64> (dense stride-1 3x3 conv, halo-staged 8x16-pixel x 64-channel tile: the 8192-row maps)", "k_conv_d3_b3<8, 16, 64"),
        "d3_64": ("k_conv_d3_b3<8,8,64> (dense stride-1 3x3 conv, halo-staged 8x8-pixel x 64-channel tile)", "k_conv_d3_b3<8, 8, 64"),
        128: ("k_conv_igemm%s<128,128> (gathered implicit-GEMM conv: sparse / 1x1 / strided / transposed, forward and data gradient)" % ("_b3" if b3 else ""),
              "k_conv_igemm_b3<128, 128" if b3 else "k_conv_igemm<128, 128"),
        64: ("k_conv_igemm%s<64,64> (gathered implicit-GEMM conv, 64x64 tiles)" % ("_b3" if b3 else ""), "k_conv_igemm_b3<64, 64" if b3 else "k_conv_igemm<64, 64"),
        "64x128_table": ("k_conv_igemm_b3<64,128,..,table> (gathered implicit-GEMM conv of the sparse layers, neighbour-table geometry, 64-row x "
                         "128-channel tiles)", "k_conv_igemm_b3<64, 128, false, false, 2"),
        "64x128_dense": ("k_conv_igemm_b3<64,128,..,dense> (stride-2 transposed convs, one output line per tile)", "k_conv_igemm_b3<64, 128, false, false, 3"),
        "64_table": ("k_conv_igemm_b3<64,64,..,table> (gathered implicit-GEMM conv, sparse layers, 64x64 tiles)", "k_conv_igemm_b3<64, 64, false, false, 2"),
        "64_dense": ("k_conv_igemm_b3<64,64,..,dense> (1x1 projections, strided convs: dense geometry, 64x64 tiles)", "k_conv_igemm_b3<64, 64, false, false, 3"),
        "128_table": ("k_conv_igemm_b3<128,128,..,table> (gathered implicit-GEMM conv, large sparse layers)", "k_conv_igemm_b3<128, 128, false, false, 2"),
        "128_dense": ("k_conv_igemm_b3<128,128,..,dense> (large dense-geometry layers)", "k_conv_igemm_b3<128, 128, false, false, 3"),
        "128x64_table": ("k_conv_igemm_b3<128,64,..,table> (large sparse layers with <= 64 output channels)", "k_conv_igemm_b3<128, 64, false, false, 2"),
        "128x64_dense": ("k_conv_igemm_b3<128,64,..,dense>", "k_conv_igemm_b3<128, 64, false, false, 3"),
        "wgrad_d3": ("k_conv_wgrad_d3_b3 (weight gradient of dense stride-1 3x3 convs: 8x8-pixel grad_out tile + 10x10 input halo staged once, "
                     "all 9 taps per staged tile, ds_read_b64_tr_b16 fragments, 8 waves x (32 co x 32 ci x 9 taps))", "k_conv_wgrad_d3_b3"),
        "wgrad_b3_128": ("k_conv_wgrad_tr_b3<128> (gathered weight gradient GEMM: M = Cout tile 128, N = Cin tile 128 of one tap, K = rows; "
                         "row-major LDS images read with ds_read_b64_tr_b16; row chunks combined with fp32 atomics)", "k_conv_wgrad_tr_b3<128"),
        "wgrad_b3_64": ("k_conv_wgrad_tr_b3<64> (gathered weight gradient GEMM, Cin tile 64)", "k_conv_wgrad_tr_b3<64"),
        "wgrad_b3_deform_128": ("k_conv_wgrad_b3<true,128> (DCNv2 weight gradient: input rows blended from 4 bilinear corners while staged)", "k_conv_wgrad_b3<true, 128"),
        "wgrad_f32_128": ("k_conv_wgrad<false,128> (weight gradient GEMM, exact fp32 MFMA)", "k_conv_wgrad<false, 128"),
        "wgrad_f32_64": ("k_conv_wgrad<false,64> (weight gradient GEMM, exact fp32 MFMA, Cin tile 64)", "k_conv_wgrad<false, 64"),
        "wgrad_f32_deform_128": ("k_conv_wgrad<true,128> (DCNv2 weight gradient, exact fp32 MFMA)", "k_conv_wgrad<true, 128"),
        "wgrad_f32_deform_64": ("k_conv_wgrad<true,64> (DCNv2 weight gradient, exact fp32 MFMA)", "k_conv_wgrad<true, 64"),
    }
    return names


def _flops_of(p):
    _, _, pairs, f, _ = p
    return f if pairs is None else float(pairs.item()) * f


def mfma_roofline(timed, table, table_steps, prof_steps, math, limited_tags=()):
    """Roofline entry of the dominant MFMA convolution instantiation.  `table`: events of EVERY MFMA launch of `table_steps` steps
    (ranks the instantiations by summed time); `timed`: events of the launches inside the timed region (`prof_steps` instrumented
    steps).  achieved = ALGORITHMIC flops (dense 2 k^2 Cin Cout rows, sparse 2 pairs Cin Cout) / HIP-event duration."""
    b3 = math == "bf16x3"
    all_ms = [a.elapsed_time(b) for a, b, _, _, _ in table]
    all_flops = [_flops_of(p) for p in table]
    by_kern, fl_kern = {}, {}
    for i, p in enumerate(table):
        tag = p[4][5]
        by_kern[tag] = by_kern.get(tag, 0.0) + all_ms[i]
        fl_kern[tag] = fl_kern.get(tag, 0.0) + all_flops[i]
    limited = {t: by_kern[t] for t in limited_tags if t in by_kern}
    ranked = {k: v for k, v in by_kern.items() if k not in limited} or by_kern
    dom = max(ranked, key=lambda k: ranked[k]) if ranked else 128
    sel = [p for p in timed if p[4][5] == dom]
    kernel_ms = [a.elapsed_time(b) for a, b, _, _, _ in sel]
    flops = [_flops_of(p) for p in sel]
    n = len(sel)
    avg_ms = sum(kernel_ms) / max(n, 1)
    achieved = (sum(flops) / max(n, 1)) / (avg_ms * 1e-3) / 1e12 if n else 0.0
    kname, pmc_key = kernel_names(b3).get(dom, (str(dom), str(dom)))
    peak = PEAK_BF16_MFMA_TFLOPS if b3 else PEAK_F32_MFMA_TFLOPS
    arith = ("fp32 operands split to bf16 hi+lo, 3 x v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate" if b3 else
             "exact fp32 on v_mfma_f32_32x32x2_f32")
    entry = {"bound": "mfma", "kernel": kname + "; " + arith, "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
             "frac": round(achieved / peak, 4),
             "flops": "algorithmic (SURVEY 8(d)): dense 2*k*k*Cin*Cout*rows, sparse 2*pairs*Cin*Cout",
             "algorithmic_flops_per_launch": round(sum(flops) / max(n, 1)),
             # what `traffic` (PMC, all launches of this kernel in the profiled command) compares with: input + output +
             # weights touched once, averaged over the same launches (shape tuple = rows_in, Cin, Cout, taps)
             "algorithmic_bytes_per_launch": round(sum(4.0 * (p[4][0] * p[4][1] + p[4][0] * p[4][2] + p[4][3] * p[4][1] * p[4][2])
                                                       for p in sel) / max(n, 1)),
             "mfma_issue_frac": round((3.0 if b3 else 1.0) * achieved / peak, 4),
             "launches_per_step": n // max(prof_steps, 1), "avg_launch_ms": round(avg_ms, 4),
             "ms_per_step_by_kernel": {str(k): round(v / table_steps, 3) for k, v in sorted(by_kern.items(), key=lambda kv: -kv[1])},
             "algorithmic_tflops_by_kernel": {str(k): round(fl_kern[k] / (by_kern[k] * 1e-3) / 1e12, 1) for k in sorted(by_kern, key=lambda k: -by_kern[k])}}
    return entry, dom, pmc_key, sum(kernel_ms), sum(all_ms), limited, fl_kern


def pmc_traffic(math, pmc_key):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes of this command (rocprofv3 counters cannot be
    collected from inside the run).  The profile records a hash of csrc/ (tools/diag/pmc_to_json.py): when the kernels have changed
    since, the figure is stale and `traffic` is null -- never a number from other code."""
    from radardistill_amd import native
    cur = native.csrc_sha()
    for rnd in ("round3", "round2", "round1"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_hbm_traffic_{math}.json")
        try:
            pm = json.load(open(path))
        except Exception:
            continue
        try:
            val = next(v["hbm_bytes_per_launch_corrected"] for k, v in pm["kernels"].items() if pmc_key in k)
        except StopIteration:
            continue
        src = os.path.relpath(path, ROOT)
        if pm.get("csrc_sha") != cur:
            return None, f"{src} was collected from other kernel sources (csrc hash {str(pm.get('csrc_sha'))[:12]} != {cur[:12]}): stale, not reported"
        return val, src
    return None, "no PMC pass committed for this arithmetic mode"


def cu_limited():
    """Instantiation tags whose launches the library restricts to a share of the CUs (conv_wgrad_d3.hip, RD_WGRAD_D3_RES)."""
    from radardistill_amd import autograd as A, kernels as K
    return {"wgrad_d3"} if (A.WGRAD_STREAM[0] and K.WGRAD_D3_MAX_CUS < 256) else set()


def main():
    args = parse()
    if args.other_math_steps < 0:
        args.other_math_steps = args.steps
    if args.amp_steps < 0:
        args.amp_steps = args.steps
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher (no GPU call has been made yet)
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    from radardistill_amd import dist as D
    world, rank, local_rank = D.env_world()
    check_world(args.gpus, world)
    backend = os.environ.get("RD_DIST_BACKEND", "nccl")
    if os.environ.get("RD_BENCH_DRY_RUN"):
        # launcher / rank bookkeeping rehearsal without a GPU (tests/test_cpu_host.py): rendezvous, barrier, max-over-ranks, one line
        D.init_distributed(backend="gloo")
        D.barrier()
        dt = D.max_over_ranks(0.001 * (rank + 1))
        if rank == 0:
            print(json.dumps({"metric": "samples/sec", "dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "max_over_ranks_s": dt, "global_batch": args.batch * world}), flush=True)
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit(f"bench.py: {world} ranks but {torch.cuda.device_count()} visible GPU(s): RCCL needs one GPU per rank "
                         "(RD_DIST_BACKEND=gloo lets ranks share a GPU to rehearse the path)")
    dev_index = local_rank % torch.cuda.device_count()                # as tools/train.py:176 of the reference
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # "nccl" is RCCL on ROCm.  RD_DIST_BACKEND=gloo lets two ranks share ONE GPU to rehearse the DDP path on a 1-GPU box
    D.init_distributed(backend=backend, device=device)
    from radardistill_amd import kernels as K
    from radardistill_amd import native
    from radardistill_amd.pcdet.models import model_fn_decorator
    from radardistill_amd.synthetic import make_batch
    from radardistill_amd.train import build_optimizer, build_scheduler
    if native.lib().rd_device_ok() != 1:
        raise SystemExit("bench.py: no gfx950 device visible to librdamd.so")

    K.set_conv_math(args.math)
    model, cfg, geom = build(os.path.join(ROOT, "tools/cfgs/radar_distill/bench_512.yaml"), args.grid, device)
    if args.sync_bn:
        from radardistill_amd.train import convert_sync_batchnorm
        model = convert_sync_batchnorm(model)
    model.train()
    optimizer = build_optimizer(model, cfg.OPTIMIZATION)
    total_steps = args.steps + args.warmup
    sched, _ = build_scheduler(optimizer, max(total_steps, 10), 1, -1, cfg.OPTIMIZATION)
    run_model = D.data_parallel(model, optimizer, dev_index)       # flat-buffer gradient all-reduce (RD_DDP=torch: DistributedDataParallel)
    model_func = model_fn_decorator()
    # a few distinct batches per rank, resident in HBM before timing; sample sharding: seed depends on the rank
    batches = [device_batch(make_batch(batch_size=args.batch, n_lidar=35000, n_radar=2000, n_boxes=30, grid=args.grid,
                                       seed=D.shard_seed(rank, i)), device) for i in range(2)]

    host_t = []
    pending = {}                      # batch dicts whose teacher branch was prefetched during the previous step (PillarNet.prefetch_teacher)
    raw_model = run_model.module if hasattr(run_model, "module") else run_model

    def step(it):
        t0 = time.perf_counter()
        sched.step(it)
        optimizer.zero_grad()
        loss, tb, _ = model_func(run_model, pending.pop(it, None) or dict(batches[it % len(batches)]))
        t1 = time.perf_counter()
        loss.backward()
        t2 = time.perf_counter()
        # the NEXT batch's teacher forward is enqueued here (every step runs exactly one teacher forward, one student forward + backward)
        pending[it + 1] = raw_model.prefetch_teacher(dict(batches[(it + 1) % len(batches)]))
        optimizer.step()
        host_t.append((t1 - t0, t2 - t1, time.perf_counter() - t2))
        return loss

    run_model.train()            # once, as train_one_epoch does (tools/train_utils/train_utils.py): the recursive mode switch costs ~1.5 ms
    from radardistill_amd.train import use_training_stream
    use_training_stream(device, int(os.environ.get("RD_MAIN_PRIO", "-1")))          # the loop's stream: high priority (RD_MAIN_PRIO=0: torch's default stream)
    def barrier():
        D.barrier()
        torch.cuda.synchronize()

    # Roofline measurement.  Per-launch HIP events around ALL ~300 MFMA launches of a step cost ~1.3 ms of it (host time and lost
    # overlap), so the work is split: the LAST warm-up step (untimed) carries events on every MFMA launch and ranks the
    # instantiations by summed time; in the timed region only the launches of the dominant instantiation (and the >= 16 MB
    # BatchNorm launches of the HBM figure) carry events, in every step.  With --warmup 0 there is no ranking step: every MFMA
    # launch of every `every`-th timed step is instrumented instead (at least 3 steps).
    K.prefill_event_pool(min(30000, 900 * 4 + 260 * (args.steps + 3)))      # timing events (and their HIP handles) created outside the timed region
    rank_prof = None
    for it in range(args.warmup):
        if it == args.warmup - 1 and not os.environ.get("RD_BENCH_NO_HOOKS"):
            K.CONV_PROFILE, K.WGRAD_PROFILE = [], []
        step(it)
    if K.CONV_PROFILE is not None:
        torch.cuda.synchronize()
        rank_prof, K.CONV_PROFILE, K.WGRAD_PROFILE = K.CONV_PROFILE + K.WGRAD_PROFILE, None, None
        by = {}
        for a_, b_, _, _, shp in rank_prof:
            by[shp[5]] = by.get(shp[5], 0.0) + a_.elapsed_time(b_)
        for tag in cu_limited():          # kernels deliberately kept to a share of the chip do not compete for "dominant" (see below)
            by.pop(tag, None)
        K.PROFILE_TAGS = {max(by, key=lambda k: by[k])} if by else None
    # an event pair costs the stream ~2 x 4 us of marker packets, so even the dominant kernel's ~25 launches are timed in every SECOND
    # step only (>= 3 steps), and the BatchNorm launches of the HBM figure in the first two instrumented steps
    every = (2 if args.steps >= 6 else 1) if rank_prof is not None else max(1, min(int(os.environ.get("RD_BENCH_PROFILE_EVERY", "4")), args.steps // 3 or 1))
    hooked = [it for it in range(args.warmup, args.warmup + args.steps) if (it - args.warmup) % every == 0]
    if os.environ.get("RD_BENCH_NO_HOOKS"):          # diagnostic: cost of the per-launch HIP events themselves
        hooked = []
    prof, bnprof, wprof = [], [], []
    barrier()
    t0 = time.perf_counter()
    for it in range(args.warmup, args.warmup + args.steps):
        on = it in hooked
        K.CONV_PROFILE, K.BN_PROFILE, K.WGRAD_PROFILE = (prof, bnprof, wprof) if on else (None, None, None)
        if on and rank_prof is not None and it not in hooked[:2]:
            K.BN_PROFILE = None
        if on and K.PROFILE_TAGS is not None:          # only the dominant kernel's family pays for its tag computation
            if str(next(iter(K.PROFILE_TAGS))).startswith("wgrad"):
                K.CONV_PROFILE = None
            else:
                K.WGRAD_PROFILE = None
        loss = step(it)
    barrier()
    dt = time.perf_counter() - t0
    K.PROFILE_TAGS = None
    K.CONV_PROFILE = K.BN_PROFILE = K.WGRAD_PROFILE = None
    last_loss = float(loss.detach())
    # The timed steps overlap the teacher's kernels (side stream) with the student's, so a launch's event-to-event duration includes
    # time shared with the other stream.  Two extra steps with the overlap switched off give the kernel's isolated duration.
    iso_prof = iso_bn = None
    from radardistill_amd import autograd as A
    if os.environ.get("RD_TEACHER_STREAM", "1") != "0" or A.WGRAD_STREAM[0]:
        prev_env, prev_w = os.environ.get("RD_TEACHER_STREAM", "1"), A.WGRAD_STREAM[0]
        os.environ["RD_TEACHER_STREAM"] = "0"
        A.WGRAD_STREAM[0] = False
        K.CONV_PROFILE = []
        K.BN_PROFILE = []
        K.WGRAD_PROFILE = []
        for it in range(args.warmup + args.steps, args.warmup + args.steps + 2):
            step(it)
        torch.cuda.synchronize()
        iso_prof, K.CONV_PROFILE = K.CONV_PROFILE + K.WGRAD_PROFILE, None
        K.WGRAD_PROFILE = None
        iso_bn, K.BN_PROFILE = K.BN_PROFILE, None
        os.environ["RD_TEACHER_STREAM"] = prev_env
        A.WGRAD_STREAM[0] = prev_w
    # the same step in the other arithmetic mode (outside the timed region), measured the same way: barrier-bracketed steps, max over
    # ranks, and HIP events on its dominant MFMA kernel -- with --math bf16x3 this leg is the exact-fp32 result (`roofline_f32`)
    other = None
    o_rank = o_timed = None
    o_hooked = 0
    if args.other_math_steps > 0:
        om = "f32" if args.math == "bf16x3" else "bf16x3"
        K.set_conv_math(om)
        it0 = args.warmup + args.steps + 2
        hooks = not os.environ.get("RD_BENCH_NO_HOOKS")
        K.prefill_event_pool(min(30000, 900 * 2 + 260 * (args.other_math_steps + 1)))
        if hooks:
            K.CONV_PROFILE, K.WGRAD_PROFILE = [], []          # ranking step of this mode: events on every MFMA launch, untimed
        step(it0)
        torch.cuda.synchronize()
        if hooks:
            o_rank, K.CONV_PROFILE, K.WGRAD_PROFILE = K.CONV_PROFILE + K.WGRAD_PROFILE, None, None
            oby = {}
            for a_, b_, _, _, shp in o_rank:
                oby[shp[5]] = oby.get(shp[5], 0.0) + a_.elapsed_time(b_)
            for tag in cu_limited():
                oby.pop(tag, None)
            K.PROFILE_TAGS = {max(oby, key=lambda k: oby[k])} if oby else None
        oprof, owprof = [], []
        o_every = 2 if args.other_math_steps >= 6 else 1
        barrier()
        t1 = time.perf_counter()
        for it in range(it0 + 1, it0 + 1 + args.other_math_steps):
            on = hooks and K.PROFILE_TAGS is not None and (it - it0 - 1) % o_every == 0
            K.CONV_PROFILE, K.WGRAD_PROFILE = (oprof, owprof) if on else (None, None)
            if on:
                o_hooked += 1
                if str(next(iter(K.PROFILE_TAGS))).startswith("wgrad"):
                    K.CONV_PROFILE = None
                else:
                    K.WGRAD_PROFILE = None
            step(it)
        barrier()
        odt = D.max_over_ranks(time.perf_counter() - t1, device)
        K.PROFILE_TAGS = None
        K.CONV_PROFILE = K.WGRAD_PROFILE = None
        o_timed = oprof + owprof
        K.set_conv_math(args.math)
        other = {"conv_math": om, "value": round(args.batch * world * args.other_math_steps / odt, 3), "unit": "samples/sec",
                 "ms_per_step": round(odt / args.other_math_steps * 1e3, 3), "steps": args.other_math_steps}
    # the reference's `--use_amp` iteration (tools/train_utils/train_utils.py:57-64): autocast arithmetic (bf16 products, fp32 accumulate;
    # train.autocast covers forward AND backward) + dynamic loss scaling, same barrier / max-over-ranks bracket.  Reported beside the headline.
    amp = None
    if args.amp_steps > 0:
        from radardistill_amd.train import AmpScaler, autocast
        scaler = AmpScaler(device)
        it0 = args.warmup + args.steps + 3 + args.other_math_steps

        def amp_step(it):
            sched.step(it)
            optimizer.zero_grad()
            with autocast(True):
                loss_, _, _ = model_func(run_model, dict(batches[it % len(batches)]))
                scaler.scale(loss_).backward()
            scaler.step(optimizer)
            scaler.update()
            return loss_

        prev_math = K.get_conv_math()
        amp_step(it0)
        barrier()
        t2 = time.perf_counter()
        for it in range(it0 + 1, it0 + 1 + args.amp_steps):
            amp_loss = amp_step(it)
        barrier()
        adt = D.max_over_ranks(time.perf_counter() - t2, device)
        K.set_conv_math(prev_math)
        amp = {"arithmetic": "train.autocast: MFMA products of bf16-rounded operands (1 of bf16x3's 3 terms), fp32 accumulate, fp32 storage and "
                             "master weights; AmpScaler dynamic loss scaling (GradScaler semantics on the device)",
               "value": round(args.batch * world * args.amp_steps / adt, 3), "unit": "samples/sec",
               "ms_per_step": round(adt / args.amp_steps * 1e3, 3), "steps": args.amp_steps, "final_loss": float(amp_loss.detach()),
               "loss_scale": scaler.get_scale()}
    if rank_prof is not None:
        roofline_note = (f"HIP events around every launch of this kernel in {len(hooked)} of the {args.steps} steps of the timed region ("
                         f"{'every step' if every == 1 else 'every second step'}: an event pair costs the stream two marker packets); it was chosen, and the per-kernel table below measured, "
                         "with events on every MFMA launch of the last warm-up step")
    else:
        roofline_note = (f"HIP events around every MFMA launch in {len(hooked)} of the {args.steps} steps of the timed region "
                         f"(every {every}th step: the events themselves cost ~1 ms per instrumented step)")
    prof_steps = max(len(hooked), 1)
    dt = D.max_over_ranks(dt, device)

    if rank == 0:
        print(f"[bench] {args.steps} steps in {dt:.3f} s on {world} GPU(s); peak HBM allocated {torch.cuda.max_memory_allocated(device) / 2**30:.1f} GiB, "
              f"reserved {torch.cuda.memory_reserved(device) / 2**30:.1f} GiB", file=sys.stderr, flush=True)
        ht = np.array(host_t[args.warmup:args.warmup + args.steps]) * 1e3
        print(f"[bench] host enqueue time per step (ms, no device sync): forward+loss {ht[:, 0].mean():.1f}  backward {ht[:, 1].mean():.1f}  "
              f"optimizer {ht[:, 2].mean():.1f}", file=sys.stderr, flush=True)
        samples = args.batch * world * args.steps
        # every MFMA convolution launch of the timed region: forward, data gradient (CONV_PROFILE) and weight gradient (WGRAD_PROFILE),
        # each tagged with the instantiation the C dispatch picks (kernels._kernel_tag / conv_wgrad)
        timed = list(prof) + list(wprof or [])
        table = rank_prof if rank_prof is not None else timed          # the table over ALL instantiations: ranking step, or the timed region
        rank_steps = 1 if rank_prof is not None else prof_steps
        if os.environ.get("RD_BENCH_SHAPES"):
            agg = {}
            for p_ in table:
                a = agg.setdefault(p_[4], [0, 0.0, 0.0]); a[0] += 1; a[1] += p_[0].elapsed_time(p_[1]); a[2] += _flops_of(p_)
            for shape, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                print(f"[shape in_rows,Cin,Cout,taps,mode,kernel={shape}] launches/step {n / rank_steps:.1f} ms/step {ms / rank_steps:.3f} "
                      f"TF/s {fl / (ms * 1e-3) / 1e12:.1f}", file=sys.stderr)
        # roofline of the DOMINANT kernel = the MFMA instantiation with the largest summed time, weight gradients included.
        # Kernels the library deliberately keeps to a share of the chip (the halo weight-gradient kernel runs on <= 80 CUs of its side
        # stream so that the main stream can start work beside it: 19.4 -> 18.7 ms per step) have launch durations set by that share,
        # not by their code; they are listed under `cu_limited_kernels` and the dominant kernel is chosen among the others.
        step_ms = dt / args.steps * 1e3
        roof, dom, pmc_key, dom_ms, all_ms_sum, limited, fl_kern = mfma_roofline(timed, table, rank_steps, prof_steps, args.math, cu_limited())
        peak = roof["peak"]
        iso = None
        if iso_prof:
            isel = [p_ for p_ in iso_prof if p_[4][5] == dom]
            if isel:
                ims = sum(a.elapsed_time(b) for a, b, _, _, _ in isel) / len(isel)
                ifl = sum(_flops_of(p_) for p_ in isel) / len(isel)
                iso = (ims, ifl / (ims * 1e-3) / 1e12)
        roof["traffic"], roof["traffic_source"] = pmc_traffic(args.math, pmc_key)
        if args.math == "bf16x3":
            # what a register-only bf16 MFMA loop sustains on this chip on random operands (the clock held under MFMA load is well under the
            # 2.4 GHz behind the spec peak: MI355X_MICROARCH.md, DVFS give-back), measured in this run after the timed region
            ceil = K.probe_mfma_bf16()
            roof["mfma_loop_ceiling"] = {"tflops": round(ceil, 1), "frac_of_spec_peak": round(ceil / peak, 4),
                                         "issued_tflops_of_this_kernel": round(3.0 * roof["achieved"], 1),
                                         "issued_frac_of_ceiling": round(3.0 * roof["achieved"] / ceil, 4) if ceil > 0 else None,
                                         "note": "rd_probe_mfma_bf16: v_mfma_f32_32x32x16_bf16 only, operands in registers, two waves per SIMD on every "
                                                 "CU, random data; `frac` above stays priced against the 2.5 PF spec peak"}
        roof["isolated"] = None if iso is None else {
            "note": "same launches in 2 extra steps with the stream overlaps (teacher || student, wgrad || dgrad) off: in the timed region a "
                    "launch shares the GPU with the other streams' kernels, which lengthens it while shortening the step",
            "avg_launch_ms": round(iso[0], 4), "achieved": round(iso[1], 3), "frac": round(iso[1] / peak, 4)}
        roof["measured"] = roofline_note
        roof["cu_limited_kernels"] = {t: {"max_cus": K.WGRAD_D3_MAX_CUS, "of": 256, "ms_per_step": round(v / rank_steps, 3),
                                          "algorithmic_tflops": round(fl_kern[t] / (v * 1e-3) / 1e12, 1),
                                          "note": "side stream, overlapped with the main stream; launched on a share of the chip by design, "
                                                  "so its launch duration reflects that share"} for t, v in limited.items()}
        roof["time_share_of_step"] = round(dom_ms / prof_steps / step_ms, 4)
        roof["all_mfma_conv_share_of_step"] = round(all_ms_sum / rank_steps / step_ms, 4)
        b3 = args.math == "bf16x3"
        out = {
            "metric": "samples/sec", "value": round(samples / dt, 3), "unit": "samples/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(step_ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16x3 (fp32 storage and accumulate)" if b3 else "f32", "data": "synthetic",
            "config": {"workload": "RadarDistill full training step (BASELINE configs[3]): frozen LiDAR teacher fwd + radar student "
                                   "fwd/bwd (VFE, SparseEnc, CMA+DCNv2, DenseEnc, CenterHead, AFD+PFD+detection losses) + clip + Adam",
                       "bev": f"{args.grid}x{args.grid}", "pillar_m": 0.2, "lidar_pts": 35000, "radar_pts": 2000, "boxes": 30,
                       "conv_math": args.math, "batch_per_gpu": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}" + ("" if world == 1 else ("/ddp" if os.environ.get("RD_DDP", "flat") == "torch" else "/flat-allreduce")) + ("+sync_bn" if args.sync_bn else ""),
                       "teacher_head": "computed (unused by the loss, as in the reference)", "final_loss": last_loss},
            "roofline": roof,
        }
        # the strict-precision result of the same run: the exact-fp32 leg's dominant kernel against the fp32 MFMA peak
        if other is not None and o_rank:
            om = other["conv_math"]
            oroof, odom, okey, odom_ms, oall_ms, _, _ = mfma_roofline(o_timed, o_rank, 1, max(o_hooked, 1), om, cu_limited())
            oroof["traffic"], oroof["traffic_source"] = pmc_traffic(om, okey)
            oroof["samples_per_sec"], oroof["ms_per_step"] = other["value"], other["ms_per_step"]
            oroof["measured"] = (f"HIP events around every launch of this kernel in {o_hooked} of the {other['steps']} steps of the {om} leg "
                                 "(timed after the headline region of the same run, same barrier / max-over-ranks bracket); the per-kernel "
                                 "table comes from the leg's untimed first step with events on every MFMA launch")
            oroof["time_share_of_step"] = round(odom_ms / max(o_hooked, 1) / other["ms_per_step"], 4)
            out["roofline_" + om] = oroof
        # HBM side of the metric ("HBM GB/s vs peak"): the streaming train-mode BatchNorm forward (normalise + affine + residual + ReLU,
        # one launch per layer), algorithmic bytes = x (+ residual) read once, y written once; launches of >= 16 MB only (smaller maps
        # are launch-latency bound and L2 / Infinity-Cache resident).
        big = [(a.elapsed_time(b), by) for a, b, by, _ in (bnprof or []) if by >= 16e6]
        if big:
            ms_tot, by_tot = sum(t for t, _ in big), sum(b for _, b in big)
            gbs = by_tot / (ms_tot * 1e-3) / 1e9
            out["roofline_hbm"] = {"bound": "hbm", "kernel": "k_bn_train_fwd (train-mode BatchNorm + residual + ReLU over rows, launches moving >= 16 MB)",
                                   "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                                   "traffic": None, "traffic_note": "the PMC passes average ALL k_bn_train_fwd dispatches (45 per step, most of them the small "
                                   "sparse-stage ones), this figure only the >= 16 MB launches: no comparable per-launch counter value",
                                   "launches_per_step": len(big) // max(min(prof_steps, 2) if rank_prof is not None else prof_steps, 1),
                                   "avg_launch_ms": round(ms_tot / len(big), 4), "algorithmic_bytes_per_launch": int(by_tot / len(big)),
                                   "measured": "HIP events around every such launch inside the timed region (other streams' kernels run concurrently)"}
            ibig = [(a.elapsed_time(b), by) for a, b, by, _ in (iso_bn or []) if by >= 16e6]
            if ibig:
                igbs = sum(b for _, b in ibig) / (sum(t for t, _ in ibig) * 1e-3) / 1e9
                out["roofline_hbm"]["isolated"] = {"achieved": round(igbs, 1), "frac": round(igbs / PEAK_HBM_GBS, 4),
                                                   "avg_launch_ms": round(sum(t for t, _ in ibig) / len(ibig), 4),
                                                   "note": "same launches in the 2 extra steps with the stream overlaps off"}
        if other is not None:
            out["other_math"] = other
        if amp is not None:
            out["amp"] = amp
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.cpu_baseline_grid, args.cpu_baseline_batch)
            except Exception as e:                      # the baseline is a reported extra; never lose the GPU number to it
                out["cpu_baseline"] = {"value": None, "unit": "samples/sec", "cores": None, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    if torch.distributed.is_available() and torch.distributed.is_initialized():          # (world > 1, or the one-rank rehearsal)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
