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
