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
    ap.add_argument("--math", choices=["f32", "bf16x3"], default="bf16x3",
                    help="arithmetic of the implicit-GEMM conv kernels: bf16x3 = fp32 operands split into bf16 hi+lo, three bf16 MFMAs per "
                         "product, fp32 accumulate (~4e-6 relative error, inside the 1e-3 parity bound; parity-tested in "
                         "tests/test_gpu_model.py::test_bf16x3_conv_math_parity); f32 = exact fp32 MFMA")
    ap.add_argument("--other-math-steps", type=int, default=5, help="extra steps timed in the other arithmetic mode after the timed region "
                    "(reported as 'other_math'; 0 = skip)")
    ap.add_argument("--cpu-baseline-grid", type=int, default=512)
    ap.add_argument("--cpu-baseline-batch", type=int, default=4)
    return ap.parse_args()


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


def cpu_baseline(grid, B=4):
    """The CPU oracle (a port: the reference itself cannot run here) timed on this host: one full training
    forward + backward on a bounded sample (B samples) of the bench geometry.  Reported, never the target."""
    from oracle import pillarnet as opn
    model, cfg, (pc_range, voxel, gs) = build(os.path.join(ROOT, "tools/cfgs/radar_distill/bench_512.yaml"), grid, "cpu")
    from radardistill_amd.synthetic import make_batch
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for k, p in model.named_parameters():
        if p.requires_grad:
            state[k].requires_grad_(True)
    b = make_batch(batch_size=B, n_lidar=35000, n_radar=2000, n_boxes=30, grid=grid, seed=0)
    ob = {"points": torch.from_numpy(b["points"]), "radar_points": torch.from_numpy(b["radar_points"]),
          "gt_boxes": torch.from_numpy(b["gt_boxes"]), "batch_size": B}
    cores = min(len(os.sched_getaffinity(0)), 16)      # the GPU box gives a 16-core share; cpu_count() reports the host
    torch.set_num_threads(cores)
    t0 = time.time()
    loss, _, _ = opn.forward_train(state, ob, pc_range, voxel, gs)
    loss.mean().backward()
    dt = time.time() - t0
    return {"value": round(B / dt, 4), "unit": "samples/sec", "cores": cores, "kind": "port",
            "sample": f"1 training step (teacher fwd + student fwd/bwd, no optimizer) at B={B}, {grid}x{grid} BEV, 35k+2k points: {dt:.1f} s"}


def main():
    args = parse()
    from radardistill_amd import dist as D
    world, rank, local_rank = D.env_world()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    dev_index = local_rank % torch.cuda.device_count()                # as tools/train.py:176 of the reference
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # "nccl" is RCCL on ROCm.  RD_DIST_BACKEND=gloo lets two ranks share ONE GPU to rehearse the DDP path on a 1-GPU box
    D.init_distributed(backend=os.environ.get("RD_DIST_BACKEND", "nccl"), device=device)
    from radardistill_amd import kernels as K
    from radardistill_amd import native
    from radardistill_amd.pcdet.models import model_fn_decorator
    from radardistill_amd.synthetic import make_batch
    from radardistill_amd.train import build_optimizer, build_scheduler
    if native.lib().rd_device_ok() != 1:
        raise SystemExit("bench.py: no gfx950 device visible to librdamd.so")

    K.set_conv_math(args.math)
    model, cfg, geom = build(os.path.join(ROOT, "tools/cfgs/radar_distill/bench_512.yaml"), args.grid, device)
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
    main_prio = int(os.environ.get("RD_MAIN_PRIO", "0"))
    if main_prio:
        torch.cuda.set_stream(torch.cuda.Stream(device, priority=main_prio))
    for it in range(args.warmup):
        step(it)

    def barrier():
        D.barrier()
        torch.cuda.synchronize()

    K.prefill_event_pool(min(30000, 900 * (args.steps + 3)))       # timing events created (and their HIP handles) outside the timed region
    K.CONV_PROFILE = []
    K.BN_PROFILE = []
    if os.environ.get("RD_BENCH_NO_HOOKS"):          # diagnostic: cost of the per-launch HIP events themselves
        K.CONV_PROFILE = K.BN_PROFILE = None
    K.WGRAD_PROFILE = [] if os.environ.get("RD_BENCH_SHAPES") else None
    barrier()
    t0 = time.perf_counter()
    for it in range(args.warmup, args.warmup + args.steps):
        loss = step(it)
    barrier()
    dt = time.perf_counter() - t0
    prof, K.CONV_PROFILE = (K.CONV_PROFILE or []), None
    bnprof, K.BN_PROFILE = (K.BN_PROFILE or []), None
    wprof, K.WGRAD_PROFILE = K.WGRAD_PROFILE, None
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
        for it in range(args.warmup + args.steps, args.warmup + args.steps + 2):
            step(it)
        torch.cuda.synchronize()
        iso_prof, K.CONV_PROFILE = K.CONV_PROFILE, None
        iso_bn, K.BN_PROFILE = K.BN_PROFILE, None
        os.environ["RD_TEACHER_STREAM"] = prev_env
        A.WGRAD_STREAM[0] = prev_w
    # the same step in the other arithmetic mode, for reference (outside the timed region)
    other = None
    if args.other_math_steps > 0:
        om = "f32" if args.math == "bf16x3" else "bf16x3"
        K.set_conv_math(om)
        it0 = args.warmup + args.steps
        step(it0)
        barrier()
        t1 = time.perf_counter()
        for it in range(it0 + 1, it0 + 1 + args.other_math_steps):
            step(it)
        barrier()
        odt = D.max_over_ranks(time.perf_counter() - t1, device)
        K.set_conv_math(args.math)
        other = {"conv_math": om, "value": round(args.batch * world * args.other_math_steps / odt, 3), "unit": "samples/sec",
                 "ms_per_step": round(odt / args.other_math_steps * 1e3, 3), "steps": args.other_math_steps}
    roofline_note = "HIP events around every launch of the kernel inside the timed region"
    prof_steps = args.steps
    dt = D.max_over_ranks(dt, device)

    if rank == 0:
        print(f"[bench] {args.steps} steps in {dt:.3f} s on {world} GPU(s); peak HBM allocated {torch.cuda.max_memory_allocated(device) / 2**30:.1f} GiB, "
              f"reserved {torch.cuda.memory_reserved(device) / 2**30:.1f} GiB", file=sys.stderr, flush=True)
        ht = np.array(host_t[-args.steps:]) * 1e3
        print(f"[bench] host enqueue time per step (ms, no device sync): forward+loss {ht[:, 0].mean():.1f}  backward {ht[:, 1].mean():.1f}  "
              f"optimizer {ht[:, 2].mean():.1f}", file=sys.stderr, flush=True)
        samples = args.batch * world * args.steps
        all_ms = [a.elapsed_time(b) for a, b, _, _, _ in prof]
        all_flops = [(f if pairs is None else float(pairs.item()) * f) for _, _, pairs, f, _ in prof]
        if os.environ.get("RD_BENCH_SHAPES"):
            agg = {}
            for ms, fl, (_, _, _, _, shape) in zip(all_ms, all_flops, prof):
                a = agg.setdefault(shape, [0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += fl
            for shape, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                print(f"[shape in_rows,Cin,Cout,taps,mode,tile={shape}] launches/step {n / prof_steps:.1f} ms/step {ms / prof_steps:.3f} "
                      f"TF/s {fl / (ms * 1e-3) / 1e12:.1f}", file=sys.stderr)
            wagg = {}
            for e0, e1, pairs, f, shape in (wprof or []):
                a = wagg.setdefault(shape, [0, 0.0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += (f if pairs is None else float(pairs.item()) * f)
            for shape, (n, ms, fl) in sorted(wagg.items(), key=lambda kv: -kv[1][1]):
                print(f"[wgrad in_rows,Cin,Cout,taps,mode={shape}] launches/step {n / prof_steps:.1f} ms/step {ms / prof_steps:.3f} "
                      f"TF/s {fl / (ms * 1e-3) / 1e12:.1f}", file=sys.stderr)
        # roofline of the dominant kernel = the forward / data-gradient convolution instantiation with the largest summed time in the
        # timed region (kernels._kernel_tag mirrors the C dispatch): the halo-staged dense 3x3 kernel or the gathered implicit GEMM
        b3 = args.math == "bf16x3"
        by_kern, fl_kern = {}, {}
        for i, p in enumerate(prof):
            if p[4][4] < 10:                                                       # mode >= 10: exact-fp32 transposed-weight data gradient
                by_kern[p[4][5]] = by_kern.get(p[4][5], 0.0) + all_ms[i]
                fl_kern[p[4][5]] = fl_kern.get(p[4][5], 0.0) + all_flops[i]
        # three instantiations sit within a few percent of each other in summed time (~5 ms per step each), so the plain maximum flips
        # from run to run: among those within 10 % of the largest time, name the one that does the most work
        top = max(by_kern.values()) if by_kern else 0.0
        dom = max((k for k in by_kern if by_kern[k] >= 0.9 * top), key=lambda k: fl_kern[k]) if by_kern else 128
        sel = [i for i, p in enumerate(prof) if p[4][5] == dom and p[4][4] < 10]
        kernel_ms = [all_ms[i] for i in sel]
        flops = [all_flops[i] for i in sel]
        n_launch = len(sel)
        avg_ms = sum(kernel_ms) / max(n_launch, 1)
        achieved = (sum(flops) / max(n_launch, 1)) / (avg_ms * 1e-3) / 1e12 if n_launch else 0.0
        iso = None
        if iso_prof:
            isel = [p for p in iso_prof if p[4][5] == dom and p[4][4] < 10]
            if isel:
                ims = sum(a.elapsed_time(b) for a, b, _, _, _ in isel) / len(isel)
                ifl = sum((f if pairs is None else float(pairs.item()) * f) for _, _, pairs, f, _ in isel) / len(isel)
                iso = (ims, ifl / (ims * 1e-3) / 1e12)
        names = {
            "d3_128": ("k_conv_d3_b3<8,16,128> (dense stride-1 3x3 conv, forward and data gradient: 8x16-pixel halo staged and split to bf16 hi+lo once "
                       "per 32-channel chunk, 9 taps by LDS offset, pre-split weights, 3 x v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate)",
                       "k_conv_d3_b3<8, 16, 128, true>"),
            "d3_64": ("k_conv_d3_b3<8,8,64> (dense stride-1 3x3 conv on the halo-staged bf16x3 kernel, 8x8-pixel x 64-channel tiles)", "k_conv_d3_b3<8, 8, 64, true>"),
            128: (("k_conv_igemm_b3<128,128,false> (gathered implicit-GEMM conv: sparse / 1x1 / strided / transposed; fp32 activations split to bf16 hi+lo in LDS, "
                   "3 x v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate)", "k_conv_igemm_b3<128, 128, false, false>") if b3 else
                  ("k_conv_igemm<128,128,2,2,false> (gathered implicit-GEMM conv: sparse + dense 3x3 / 1x1 / transposed, exact fp32 MFMA)", "k_conv_igemm<128, 128, 2, 2, false")),
            64: (("k_conv_igemm_b3<64,64,false> (gathered implicit-GEMM conv, 64x64 tiles, bf16x3)", "k_conv_igemm_b3<64, 64, false, false>") if b3 else
                 ("k_conv_igemm<64,64,2,2,false> (gathered implicit-GEMM conv, 64x64 tiles, exact fp32 MFMA)", "k_conv_igemm<64, 64, 2, 2, false")),
        }
        kname, pmc_key = names[dom]
        peak = PEAK_BF16_MFMA_TFLOPS if b3 else PEAK_F32_MFMA_TFLOPS
        if b3:
            # every algorithmic multiply-add is three bf16 MFMA products (a_lo*b_hi + a_hi*b_lo + a_hi*b_hi): price the kernel
            # against the dense bf16 MFMA peak with the flops it really issues
            algorithmic, achieved = achieved, 3.0 * achieved
        else:
            algorithmic = achieved
        traffic = None          # HBM bytes per launch from the committed PMC passes (cannot be collected live inside bench.py)
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", f"round1_pmc_hbm_traffic_{args.math}.json")))["kernels"]
            traffic = next(v["hbm_bytes_per_launch_corrected"] for k, v in pm.items() if pmc_key in k)
        except Exception:
            pass
        out = {
            "metric": "samples/sec", "value": round(samples / dt, 3), "unit": "samples/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16x3 (fp32 storage and accumulate)" if b3 else "f32", "data": "synthetic",
            "config": {"workload": "RadarDistill full training step (BASELINE configs[3]): frozen LiDAR teacher fwd + radar student "
                                   "fwd/bwd (VFE, SparseEnc, CMA+DCNv2, DenseEnc, CenterHead, AFD+PFD+detection losses) + clip + Adam",
                       "bev": f"{args.grid}x{args.grid}", "pillar_m": 0.2, "lidar_pts": 35000, "radar_pts": 2000, "boxes": 30,
                       "conv_math": args.math, "batch_per_gpu": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}" + ("" if world == 1 else ("/ddp" if os.environ.get("RD_DDP", "flat") == "torch" else "/flat-allreduce")),
                       "teacher_head": "computed (unused by the loss, as in the reference)", "final_loss": last_loss},
            "roofline": {"bound": "mfma", "kernel": kname,
                         "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic, "algorithmic_tflops": round(algorithmic, 3),
                         "isolated": None if iso is None else {
                             "note": "same launches in 2 extra steps with the stream overlaps (teacher || student, wgrad || dgrad) off: in the timed region a "
                                     "launch shares the GPU with the other streams' kernels, which lengthens it while shortening the step",
                             "avg_launch_ms": round(iso[0], 4), "achieved": round(iso[1] * (3.0 if b3 else 1.0), 3),
                             "frac": round(iso[1] * (3.0 if b3 else 1.0) / peak, 4)},
                         "launches_per_step": n_launch // max(prof_steps, 1), "measured": roofline_note, "avg_launch_ms": round(avg_ms, 4),
                         "ms_per_step_by_kernel": {str(k): round(v / prof_steps, 3) for k, v in sorted(by_kern.items(), key=lambda kv: -kv[1])},
                         "time_share_of_step": round(sum(kernel_ms) / prof_steps / (dt / args.steps * 1e3), 4),
                         "all_mfma_conv_fwd_dgrad_share_of_step": round(sum(all_ms) / prof_steps / (dt / args.steps * 1e3), 4)},
        }
        # HBM side of the metric ("HBM GB/s vs peak"): the streaming train-mode BatchNorm forward (normalise + affine + residual + ReLU,
        # one launch per layer), algorithmic bytes = x (+ residual) read once, y written once; launches of >= 16 MB only (smaller maps
        # are launch-latency bound and L2 / Infinity-Cache resident).
        big = [(a.elapsed_time(b), by) for a, b, by, _ in (bnprof or []) if by >= 16e6]
        if big:
            ms_tot, by_tot = sum(t for t, _ in big), sum(b for _, b in big)
            gbs = by_tot / (ms_tot * 1e-3) / 1e9
            out["roofline_hbm"] = {"bound": "hbm", "kernel": "k_bn_train_fwd (train-mode BatchNorm + residual + ReLU over rows, launches moving >= 16 MB)",
                                   "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                                   "traffic": None, "launches_per_step": len(big) // max(prof_steps, 1),
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
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.cpu_baseline_grid, args.cpu_baseline_batch)
            except Exception as e:                      # the baseline is a reported extra; never lose the GPU number to it
                out["cpu_baseline"] = {"value": None, "unit": "samples/sec", "cores": None, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
