"""Diagnostic (GPU box): statistical profile of the HOST side of the training step across ALL threads (cProfile does not see the
autograd engine's thread, where every backward function runs): a sampler thread reads sys._current_frames() every ~0.3 ms and
counts (a) the innermost frame and (b) every radardistill_amd frame on the stack (cumulative)."""
import collections
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench as B


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    device = torch.device("cuda", 0)
    from radardistill_amd import kernels as K
    from radardistill_amd.pcdet.models import model_fn_decorator
    from radardistill_amd.synthetic import make_batch
    from radardistill_amd.train import build_optimizer, build_scheduler
    K.set_conv_math("bf16x3")
    model, cfg, geom = B.build(os.path.join(ROOT, "tools/cfgs/radar_distill/bench_512.yaml"), 512, device)
    model.train()
    opt = build_optimizer(model, cfg.OPTIMIZATION)
    sched, _ = build_scheduler(opt, 1000, 1, -1, cfg.OPTIMIZATION)
    fn = model_fn_decorator()
    batches = [B.device_batch(make_batch(batch_size=batch, n_lidar=35000, n_radar=2000, n_boxes=30, grid=512, seed=i), device) for i in range(2)]

    def step(it):
        sched.step(it)
        opt.zero_grad()
        loss, tb, _ = fn(model, dict(batches[it % 2]))
        loss.backward()
        opt.step()

    for it in range(5):
        step(it)
    torch.cuda.synchronize()
    inner, cum = collections.Counter(), collections.Counter()
    stop = [False]
    me = []
    n_samples = [0]

    def sampler():
        me.append(threading.get_ident())
        while not stop[0]:
            for tid, fr in sys._current_frames().items():
                if tid == me[0]:
                    continue
                f = fr
                first = True
                seen = set()
                depth = 0
                while f is not None and depth < 60:
                    co = f.f_code
                    key = f"{co.co_filename.split('/repo/')[-1].split('dist-packages/')[-1]}:{co.co_name}"
                    if first:
                        inner[key + f":{f.f_lineno}"] += 1
                        first = False
                    if "radardistill_amd" in co.co_filename and key not in seen:
                        cum[key] += 1
                        seen.add(key)
                    f = f.f_back
                    depth += 1
            n_samples[0] += 1
            time.sleep(0.0003)

    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    n = 60
    t0 = time.perf_counter()
    for it in range(5, 5 + n):
        step(it)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop[0] = True
    th.join()
    tot = sum(inner.values())
    print(f"{n} steps, {dt / n * 1e3:.1f} ms/step under sampling, {n_samples[0]} sampling rounds, {tot} thread-samples")
    print("---- innermost frame (share of thread-samples; idle threads sit in their wait frame)")
    for k, c in inner.most_common(45):
        print(f"  {100.0 * c / n_samples[0]:6.2f} %  {k}")
    print("---- cumulative, radardistill_amd frames (share of sampling rounds)")
    for k, c in cum.most_common(60):
        print(f"  {100.0 * c / n_samples[0]:6.2f} %  {k}")


if __name__ == "__main__":
    main()
