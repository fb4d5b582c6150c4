"""FlatAdam (one fneus_adam launch per step) against torch.optim.Adam, the reference's optimiser (exp_runner.py:108)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _params(seed, flat):
    """a weight-normalised layer carved out of one flat buffer (adjacent views, like the fused MLPs) + loose tensors"""
    rs = np.random.RandomState(seed)
    shapes = [(256,), (256, 1), (256, 39), (3,), ()]
    vals = [torch.from_numpy(np.asarray(rs.standard_normal(s), dtype=np.float32)) for s in shapes]
    if not flat:
        return [torch.nn.Parameter(v.clone().to(DEV)) for v in vals]
    n = sum(v.numel() for v in vals[:3])
    buf, gbuf = torch.empty(n, device=DEV), torch.zeros(n, device=DEV)
    ps, off = [], 0
    for v in vals[:3]:
        p = torch.nn.Parameter(torch.empty(0, device=DEV))
        p.data = buf[off: off + v.numel()].view(v.shape)
        p.data.copy_(v)
        p.grad = gbuf[off: off + v.numel()].view(v.shape)
        ps.append(p)
        off += v.numel()
    for v in vals[3:]:
        p = torch.nn.Parameter(v.clone().to(DEV))
        p.grad = torch.zeros_like(p)
        ps.append(p)
    return ps


def _grads(seed, step, ps):
    rs = np.random.RandomState(1000 * seed + step)
    return [torch.from_numpy(np.asarray(rs.standard_normal(tuple(p.shape)) * (10.0 ** rs.uniform(-4, 0)), dtype=np.float32)).to(DEV)
            for p in ps]


def test_flat_adam_matches_torch_adam_and_clears_grads():
    from fneus.optim import FlatAdam
    ref_ps, ps = _params(1, False), _params(1, True)
    ref = torch.optim.Adam(ref_ps, lr=5e-4)
    opt = FlatAdam(ps, lr=5e-4)
    for step in range(12):
        if step == 6:                       # learning-rate schedule, reference style (exp_runner.py:206-215)
            for g in ref.param_groups:
                g["lr"] = 2e-4
            opt.param_groups[0]["lr"] = 2e-4
        for p, q, g in zip(ref_ps, ps, _grads(1, step, ps)):
            p.grad = g.clone()
            q.grad.add_(g)                  # accumulate into the persistent (cleared) buffer, like autograd does
        ref.step()
        opt.step()
        for q in ps:
            assert q.grad.abs().max().item() == 0.0, "the step must leave cleared gradients behind"
    assert opt.n_segments == 3              # the three adjacent views are ONE segment
    for p, q in zip(ref_ps, ps):
        assert (p - q).abs().max().item() <= 2e-6
        for key in ("exp_avg", "exp_avg_sq"):       # same recurrences in fp32: agreement to rounding
            a, b = ref.state[p][key], opt.state[q][key]
            assert (a - b).abs().max().item() <= 1e-5 * (a.abs().max().item() + 1e-30), key
        assert float(opt.state[q]["step"]) == 12.0


def test_flat_adam_state_dict_roundtrip_continues_identically():
    from fneus.optim import FlatAdam
    a_ps, b_ps = _params(2, True), _params(2, True)
    a, b = FlatAdam(a_ps, lr=1e-3), FlatAdam(b_ps, lr=1e-3)
    for step in range(4):
        for p, g in zip(a_ps, _grads(2, step, a_ps)):
            p.grad.add_(g)
        a.step()
    sd = a.state_dict()
    assert set(sd["state"][0].keys()) >= {"step", "exp_avg", "exp_avg_sq"}          # torch.optim.Adam's format
    with torch.no_grad():
        for p, q in zip(a_ps, b_ps):
            q.copy_(p)
    b.load_state_dict(sd)
    for step in range(4, 8):
        gs = _grads(2, step, a_ps)
        for p, q, g in zip(a_ps, b_ps, gs):
            p.grad.add_(g)
            q.grad.add_(g)
        a.step()
        b.step()
    for p, q in zip(a_ps, b_ps):
        assert torch.equal(p.detach(), q.detach())
    assert float(b.state[b_ps[0]]["step"]) == 8.0


@pytest.mark.parametrize("n_outside", [0, 8], ids=["wmask", "womask"])
def test_optimizer_state_is_interchangeable_with_the_reference_parameter_list(n_outside):
    """The reference hands Adam nerf_outside (24), sdf_network, deviation_network, color_network, refColor_network, in that
    order, whether or not the background NeRF is evaluated (exp_runner.py:89-96); torch maps optimiser state onto
    parameters by POSITION.  So (a) a torch.optim.Adam over tensors of the reference's list must load this trainer's
    optimizer.state_dict() with every moment on the right shape, and (b) FlatAdam must load that Adam's state_dict back and
    keep stepping; (c) a state_dict of another order (round 1's: sdf first) must be refused, not silently mis-assigned."""
    import copy
    from fneus import ops, synth
    from fneus.trainer import Stage1Trainer, WMASK_MODEL
    conf = copy.deepcopy(WMASK_MODEL)
    conf["neus_renderer"] = dict(n_samples=16, n_importance=16, n_outside=n_outside, up_sample_steps=4, perturb=0.0)
    tr = Stage1Trainer(DEV, model_conf=conf, prec=ops.PREC_PARITY, seed=3, use_graph=False)
    names = [f"{tag}.{k}" for tag, m in zip(("nerf", "sdf", "var", "color", "ref"), tr.modules) for k, _ in m.named_parameters()]
    assert len(names) == len(tr.params) == 24 + 27 + 1 + 15 + 20 and names[0].startswith("nerf.") and names[24].startswith("sdf.")
    data = torch.from_numpy(synth.ray_batch(32, seed=2, n_miss=2)).to(DEV)
    bg = torch.ones(1, 3, device=DEV) if n_outside else None
    for _ in range(2):
        tr.train_step(data, cos_anneal_ratio=0.5, background_rgb=bg)
    sd = tr.optimizer.state_dict()
    assert sd["param_groups"][0]["params"] == list(range(len(tr.params)))
    with_state = set(sd["state"].keys())
    assert with_state == (set(range(len(tr.params))) if n_outside else set(range(24, len(tr.params))))   # Adam skipped the unused NeRF
    # (a) a plain torch Adam over the reference's parameter list
    plain = [torch.nn.Parameter(p.detach().clone()) for p in tr.params]
    adam = torch.optim.Adam(plain, lr=5e-4)
    adam.load_state_dict(copy.deepcopy(sd))
    for i, p in enumerate(plain):
        if i in with_state:
            assert adam.state[p]["exp_avg"].shape == p.shape and float(adam.state[p]["step"]) == 2.0
            p.grad = torch.randn_like(p) * 1e-3
    adam.step()
    sd2 = adam.state_dict()
    # (b) back into a fresh trainer's FlatAdam
    tr2 = Stage1Trainer(DEV, model_conf=conf, prec=ops.PREC_PARITY, seed=3, use_graph=False)
    tr2.optimizer.load_state_dict(sd2)
    tr2.train_step(data, cos_anneal_ratio=0.5, background_rgb=bg)
    torch.cuda.synchronize()
    k = 24                                                    # sdf.lin0.bias
    st = tr2.optimizer.state[tr2.params[k]]
    assert float(st["step"]) == 4.0 and st["exp_avg"].shape == tr2.params[k].shape
    assert torch.isfinite(st["exp_avg"]).all() and st["exp_avg"].abs().max() > 0
    # (c) round 1's order (sdf, deviation, color, refColor[, nerf]) is refused
    order = list(range(24, len(tr.params))) + list(range(24))
    bad = {"state": {j: sd2["state"][i] for j, i in enumerate(order) if i in sd2["state"]},
           "param_groups": copy.deepcopy(sd2["param_groups"])}
    with pytest.raises(ValueError):
        tr2.optimizer.load_state_dict(bad)
