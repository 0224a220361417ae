"""A13 pinned: fixture g8_optim was produced by the REFERENCE's own build_optimizer / OptimWrapper / OneCycle / clip_grad_norm_
loop (tests/golden/make_golden.py::g8_optim) on the small problem of tests/golden/optim_case.py.  CPU: the oracle restatement,
the host-side schedule and parameter numbering, and the checkpoint layout against it.  GPU: the fused HIP optimizer against it.

Tolerances: lr / momentum are Python floats -> equal to 1e-15; parameters after each of the 8 steps agree to 2e-6 relative
(+1e-8 absolute): fp32 with a different but equivalent operation order (torch 2.10's Adam uses lerp_, the oracle mul_/add_ like
the torch 1.10 the reference was written for, the kernel fuses the chain)."""
import numpy as np
import pytest
import torch

from oracle import optim as ooptim
from tests.golden import optim_case as OC
from tests.seeded import seeded_fill_


def _model(device="cpu"):
    torch.manual_seed(0)
    m = OC.OptimCaseNet()
    sd = m.state_dict(); seeded_fill_(sd, seed=18); m.load_state_dict(sd)
    return m.to(device)


def _flat(model):
    return torch.cat([p.detach().reshape(-1).cpu() for _, p in OC.trainable(model)]).numpy()


def _moments_close(got, ref):
    """Moments are sums of signed terms: elements that cancel to near zero carry the rounding of the large ones."""
    np.testing.assert_allclose(got, ref, rtol=2e-6, atol=1e-6 * float(np.abs(ref).max()))


def test_oracle_optimizer_vs_reference_fixture(golden_dir):
    g = np.load(f"{golden_dir}/g8_optim.npz")
    m = _model()
    named = OC.trainable(m)
    assert [n for n, _ in named] == list(g["names"])
    params = [p.detach() for _, p in named]
    mom1 = [torch.zeros_like(p) for p in params]
    mom2 = [torch.zeros_like(p) for p in params]
    steps = [0] * len(params)
    total_steps = OC.TOTAL_ITERS_EACH_EPOCH * OC.TOTAL_EPOCHS
    for it in range(OC.N_STEPS):
        lr, mom = ooptim.one_cycle(it, total_steps, OC.OPTIM_CFG["LR"], OC.OPTIM_CFG["MOMS"], OC.OPTIM_CFG["DIV_FACTOR"], OC.OPTIM_CFG["PCT_START"])
        assert abs(lr - g["lr"][it]) <= 1e-15 and abs(mom - g["mom"][it]) <= 1e-15
        OC.assign_grads(m, it)
        total, clipped = ooptim.clip_grad_norm([p.grad for _, p in named], OC.OPTIM_CFG["GRAD_NORM_CLIP"])
        np.testing.assert_allclose(float(total), g["total_norm"][it], rtol=1e-6)
        ooptim.adam_true_wd_step(params, clipped, mom1, mom2, steps, lr, mom, wd=OC.OPTIM_CFG["WEIGHT_DECAY"])
        np.testing.assert_allclose(_flat(m), g["params_after"][it], rtol=2e-6, atol=1e-8, err_msg=f"step {it}")
    for i, (n, _) in enumerate(named):
        assert bool(g[f"has_state_{n}"]) == (steps[i] > 0), n
        if steps[i]:
            assert steps[i] == int(g[f"step_{n}"]), n
            _moments_close(mom1[i].numpy(), g[f"exp_avg_{n}"])
            _moments_close(mom2[i].numpy(), g[f"exp_avg_sq_{n}"])


def test_host_schedule_and_parameter_numbering_vs_reference_fixture(golden_dir):
    from radardistill_amd.train import OneCycle, reference_param_groups
    g = np.load(f"{golden_dir}/g8_optim.npz")

    class Knobs:
        lr = mom = None
    k = Knobs()
    sched = OneCycle(k, OC.TOTAL_ITERS_EACH_EPOCH * OC.TOTAL_EPOCHS, OC.OPTIM_CFG["LR"], OC.OPTIM_CFG["MOMS"], OC.OPTIM_CFG["DIV_FACTOR"],
                     OC.OPTIM_CFG["PCT_START"])
    assert abs(k.lr - float(g["lr0"])) <= 1e-15 and abs(k.mom - float(g["mom0"])) <= 1e-15
    for it in range(OC.N_STEPS):
        sched.step(it)
        assert abs(k.lr - g["lr"][it]) <= 1e-15 and abs(k.mom - g["mom"][it]) <= 1e-15
    m = _model()
    params, groups = reference_param_groups(m)
    names = {id(p): n for n, p in m.named_parameters()}
    assert [names[id(params[i])] for grp in groups for i in grp] == list(g["ref_param_order"])
    assert [len(grp) for grp in groups] == list(g["ref_group_sizes"])


def _reference_optimizer_state(g, model):
    """The reference checkpoint's `optimizer_state` rebuilt from the fixture (what torch.optim.Adam.state_dict() returned there)."""
    order = list(g["ref_param_order"])
    state = {}
    for idx, n in enumerate(order):
        if bool(g[f"has_state_{n}"]):
            state[idx] = {"step": int(g[f"step_{n}"]), "exp_avg": torch.from_numpy(g[f"exp_avg_{n}"]),
                          "exp_avg_sq": torch.from_numpy(g[f"exp_avg_sq_{n}"])}
    sizes = list(g["ref_group_sizes"])
    groups, base = [], 0
    for s in sizes:
        groups.append({"lr": float(g["lr"][-1]), "betas": (float(g["mom"][-1]), 0.99), "eps": 1e-8, "weight_decay": 0, "amsgrad": False,
                       "params": list(range(base, base + int(s)))})
        base += int(s)
    return {"state": state, "param_groups": groups}


def test_optimizer_state_interchanges_with_the_reference_layout(golden_dir):
    """load_state_dict takes a reference `optimizer_state`; state_dict gives the same structure back (host logic only: no launch)."""
    from radardistill_amd.train import build_optimizer
    from radardistill_amd.pcdet.config import AttrDict
    g = np.load(f"{golden_dir}/g8_optim.npz")
    m = _model()
    opt = build_optimizer(m, AttrDict(OC.OPTIM_CFG))
    ref_sd = _reference_optimizer_state(g, m)
    opt.load_state_dict(ref_sd)
    assert opt.step_count == 8 and abs(opt.lr - g["lr"][-1]) < 1e-15 and abs(opt.mom - g["mom"][-1]) < 1e-15
    names = [n for n, _ in OC.trainable(m)]
    own = {n: opt.step_count - int(opt.skipped[i]) for i, n in enumerate(names)}
    assert own["late.weight"] == 7 and own["early.bias"] == 3 and own["unused.weight"] == 0 and own["body.0.weight"] == 8
    back = opt.state_dict()
    assert set(back["state"]) == set(ref_sd["state"]) and [gr["params"] for gr in back["param_groups"]] == [gr["params"] for gr in ref_sd["param_groups"]]
    for idx, st in ref_sd["state"].items():
        assert back["state"][idx]["step"] == st["step"]
        assert torch.equal(back["state"][idx]["exp_avg"], st["exp_avg"]) and torch.equal(back["state"][idx]["exp_avg_sq"], st["exp_avg_sq"])
    # the structure is torch.optim.Adam's own: a plain Adam over the same numbering accepts it
    order = list(g["ref_param_order"])
    byname = dict(m.named_parameters())
    sizes = [int(s) for s in g["ref_group_sizes"]]
    plain = torch.optim.Adam([{"params": [byname[n] for n in order[:sizes[0]]]}, {"params": [byname[n] for n in order[sizes[0]:]]}])
    plain.load_state_dict(back)
    assert int(plain.state[byname["late.weight"]]["step"]) == 7
    with pytest.raises(RuntimeError):
        opt.load_state_dict({"foo": 1})


def test_student_init_checkpoint_synthesis(tmp_path):
    """ckpt.py:9-23: every teacher key is duplicated under 'radar_' + key, bookkeeping entries are carried over; loading it into the
    distillation model fills teacher and student and skips the radar VFE's 15-column Linear (shape test)."""
    from radardistill_amd import ckpt as C
    teacher = {"vfe.pfn_layers.0.linear.weight": torch.randn(32, 14), "backbone_3d.conv1.0.conv1.weight": torch.randn(32, 3, 3, 32),
               "dense_head.shared_conv.0.bias": torch.randn(64), "global_step": torch.tensor([7])}
    src = {"epoch": 20, "it": 1234, "optimizer_state": None, "version": "pcdet+0.5.2", "model_state": teacher}
    torch.save(src, tmp_path / "lidar.pth")
    C.main([str(tmp_path / "lidar.pth"), str(tmp_path / "init.pth")])
    out = torch.load(tmp_path / "init.pth", weights_only=True)
    assert (out["epoch"], out["it"], out["optimizer_state"], out["version"]) == (20, 1234, None, "pcdet+0.5.2")
    keys = list(out["model_state"])
    assert keys == [k for t in teacher for k in (t, "radar_" + t)]
    for t, v in teacher.items():
        assert torch.equal(out["model_state"][t], v) and torch.equal(out["model_state"]["radar_" + t], v)


@pytest.mark.gpu
def test_fused_optimizer_vs_reference_fixture(golden_dir):
    """The HIP optimizer (rd_grad_norm + rd_adam_step through FusedAdamOneCycle) reproduces the reference loop step by step,
    including clipped steps, the two parameter kinds, and parameters that sit steps out (gradient None)."""
    from radardistill_amd.pcdet.config import AttrDict
    from radardistill_amd.train import build_optimizer, build_scheduler
    g = np.load(f"{golden_dir}/g8_optim.npz")
    m = _model("cuda")
    cfg = AttrDict(OC.OPTIM_CFG)
    opt = build_optimizer(m, cfg)
    sched, _ = build_scheduler(opt, OC.TOTAL_ITERS_EACH_EPOCH, OC.TOTAL_EPOCHS, -1, cfg)
    for it in range(OC.N_STEPS):
        sched.step(it)
        opt.zero_grad()
        OC.assign_grads(m, it)
        norm = opt.step()
        np.testing.assert_allclose(float(norm[0]), g["total_norm"][it], rtol=2e-6)
        np.testing.assert_allclose(_flat(m), g["params_after"][it], rtol=2e-6, atol=1e-8, err_msg=f"step {it}")
    sd = opt.state_dict()
    order = list(g["ref_param_order"])
    for idx, n in enumerate(order):
        assert (idx in sd["state"]) == bool(g[f"has_state_{n}"]), n
        if idx in sd["state"]:
            assert sd["state"][idx]["step"] == int(g[f"step_{n}"]), n
            _moments_close(sd["state"][idx]["exp_avg"].cpu().numpy(), g[f"exp_avg_{n}"])
            _moments_close(sd["state"][idx]["exp_avg_sq"].cpu().numpy(), g[f"exp_avg_sq_{n}"])
    # resume: a second optimizer restored from that state continues identically
    m2 = _model("cuda")
    m2.load_state_dict(m.state_dict())
    opt2 = build_optimizer(m2, cfg)
    opt2.load_state_dict(sd)
    for o, mm in ((opt, m), (opt2, m2)):
        o.lr, o.mom = 5e-4, 0.9
        o.zero_grad()
        OC.assign_grads(mm, 1)
        o.step()
    assert np.array_equal(_flat(m), _flat(m2))


@pytest.mark.gpu
def test_amp_loss_scaling_matches_torch_gradscaler():
    """The `--use_amp` control flow (train_utils.py:23,57-64): loss scaling, unscale + clip + step, skipped step on overflow, scale
    back-off and growth -- AmpScaler + the fused optimizer against torch.cuda.amp.GradScaler driving the same update rule (decoupled
    decay + torch.optim.Adam, i.e. OptimWrapper.step) on the same gradients."""
    from radardistill_amd.train import AmpScaler, FusedAdamOneCycle
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    shapes = [(300,), (64, 3, 3, 32), (4097,)]
    init = [torch.randn(s, generator=g) for s in shapes]
    ps = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    rs = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    opt = FusedAdamOneCycle(ps, wd=0.01, grad_clip=10.0)
    ref = torch.optim.Adam(rs, lr=1e-3, betas=(0.9, 0.99), eps=1e-8)
    mine = AmpScaler(dev, init_scale=2.0 ** 16, growth_interval=2)
    theirs = torch.amp.GradScaler("cuda", init_scale=2.0 ** 16, growth_interval=2)
    lr, mom, wd = 1e-3, 0.9, 0.01
    for it in range(6):
        grads = [torch.randn(s, generator=g) * (30.0 if it == 4 else 0.5) for s in shapes]
        if it == 2:
            grads[1][0, 0, 0, 0] = float("inf")                       # overflow: both must skip the step and halve the scale
        theirs.scale(torch.ones((), device=dev))                       # GradScaler creates its device-side state on the first scale()
        s_mine, s_ref = mine.get_scale(), theirs.get_scale()
        assert s_mine == s_ref, (it, s_mine, s_ref)
        for p, r, gg in zip(ps, rs, grads):
            p.grad = (gg * s_mine).to(dev)
            r.grad = (gg * s_ref).to(dev)
        opt.lr, opt.mom = lr, mom
        norm = mine.step(opt)
        mine.update()
        theirs.unscale_(ref)
        total = torch.nn.utils.clip_grad_norm_(rs, 10.0)
        for grp in ref.param_groups:
            grp["lr"], grp["betas"] = lr, (mom, 0.99)
        before = [r.detach().clone() for r in rs]
        theirs.step(ref)
        skipped = all(torch.equal(a, b.detach()) for a, b in zip(before, rs))
        if not skipped:                                               # OptimWrapper.step: decay happens with the step, not when it is skipped
            with torch.no_grad():
                for r, b in zip(rs, before):
                    r.copy_(b * (1 - wd * lr) + (r - b))
        theirs.update()
        assert skipped == (it == 2)
        if it != 2:
            np.testing.assert_allclose(float(norm[0]), float(total), rtol=1e-5)
        for p, r in zip(ps, rs):
            np.testing.assert_allclose(p.detach().cpu().numpy(), r.detach().cpu().numpy(), rtol=2e-5, atol=1e-6, err_msg=f"step {it}")
    assert mine.get_scale() == theirs.get_scale()
    # checkpoint / resume through an overflow-skipped step: the stored Adam `step` of every parameter is free of the skipped step (5, as
    # in torch's own state), and an optimizer restored from it continues exactly like torch's (bias correction would shift otherwise)
    sd = opt.state_dict()
    for idx, r in enumerate(rs):
        assert sd["state"][idx]["step"] == int(ref.state[r]["step"]) == 5
    ps2 = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt2 = FusedAdamOneCycle(ps2, wd=0.01, grad_clip=10.0)
    opt2.load_state_dict(sd)
    grads = [torch.randn(s, generator=g) * 0.5 for s in shapes]
    for o, pp in ((opt, ps), (opt2, ps2)):
        for p, gg in zip(pp, grads):
            p.grad = gg.to(dev).clone()
        o.lr, o.mom = lr, mom
        o.step()
    with torch.no_grad():
        for r, gg in zip(rs, grads):
            r.grad = gg.to(dev).clone()
        torch.nn.utils.clip_grad_norm_(rs, 10.0)
        before = [r.detach().clone() for r in rs]
    ref.step()
    with torch.no_grad():
        for r, b in zip(rs, before):
            r.copy_(b * (1 - wd * lr) + (r - b))
    for p, p2, r in zip(ps, ps2, rs):
        np.testing.assert_allclose(p2.detach().cpu().numpy(), r.detach().cpu().numpy(), rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(p.detach().cpu().numpy(), r.detach().cpu().numpy(), rtol=2e-5, atol=1e-6)
