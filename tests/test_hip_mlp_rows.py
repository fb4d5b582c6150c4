"""The trained plain MLPs of stages 2 / 3 on the fneus_mlp_* kernels (csrc/mlp_rows_kernels.hip) against the same nn.Sequential in
float64 on the CPU (reference: models/fields.py:338-413 Lvis / IndirectLight, models/inverRender.py:451-598 BRDF auto-encoder,
net_cs).  fp32 products and sums on both sides of the comparison that matters -- torch's own fp32 GPU result is measured against the
same float64 values and the kernels must not be further away than 4 x that (+ 1e-6 of the scale)."""
import os
import sys

import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))

pytestmark = pytest.mark.gpu


def _net(kind):
    if kind == "lvis":            # fields.py:338-369
        dims, act, last = [90, 256, 256, 256, 256, 1], nn.ReLU, nn.Sigmoid
    elif kind == "indi":          # fields.py:372-413
        dims, act, last = [63, 512, 512, 512, 512, 144], nn.ReLU, None
    elif kind == "brdf_enc":      # inverRender.py:474-480
        dims, act, last = [63, 512, 512, 512, 512, 32], lambda: nn.LeakyReLU(0.2), None
    elif kind == "brdf_dec":      # inverRender.py:482-486
        dims, act, last = [32, 128, 128, 4], lambda: nn.LeakyReLU(0.2), None
    elif kind == "net_cs":        # inverRender.py:488-498
        dims, act, last = [90, 256, 256, 256, 256, 1], lambda: nn.LeakyReLU(0.2), nn.Sigmoid
    else:                         # a single Linear with a sigmoid: the top layer is the bottom layer
        dims, act, last = [7, 5], nn.ReLU, nn.Sigmoid
    mods = []
    for i in range(len(dims) - 1):
        mods.append(nn.Linear(dims[i], dims[i + 1]))
        if i < len(dims) - 2:
            mods.append(act())
    if last is not None:
        mods.append(last())
    return nn.Sequential(*mods)


class _Owner:
    direct_grads = False


def _run(seq, x, cot, owner, engine):
    from models import fields
    from fneus import ops
    saved = ops.MLP_ROWS
    ops.MLP_ROWS = engine
    try:
        for p in seq.parameters():
            if not owner.direct_grads:
                p.grad = None
        x = x.clone().requires_grad_(True)
        y = fields._seq_direct(seq, x, owner)
        (y * cot).sum().backward()
        return y.detach(), x.grad.detach(), [p.grad.detach().clone() for p in seq.parameters()]
    finally:
        ops.MLP_ROWS = saved


@pytest.mark.parametrize("kind,rows", [("lvis", 2048), ("lvis", 37), ("indi", 512), ("brdf_enc", 512), ("brdf_enc", 24),
                                       ("brdf_dec", 512), ("brdf_dec", 1), ("net_cs", 500), ("one", 70)])
def test_mlp_rows_match_float64(kind, rows):
    torch.manual_seed(7)
    seq = _net(kind)
    n_in, n_out = seq[0].in_features, [m for m in seq if isinstance(m, nn.Linear)][-1].out_features
    x = torch.randn(rows, n_in)
    cot = torch.randn(rows, n_out)
    ref_seq = _net(kind).double()
    ref_seq.load_state_dict({k: v.double() for k, v in seq.state_dict().items()})
    xr = x.double().requires_grad_(True)
    yr = ref_seq(xr)
    (yr * cot.double()).sum().backward()
    ref = [yr.detach(), xr.grad] + [p.grad for p in ref_seq.parameters()]
    seq = seq.cuda()
    owner = _Owner()
    got = _run(seq, x.cuda(), cot.cuda(), owner, True)
    lib = _run(seq, x.cuda(), cot.cuda(), owner, False)
    got = [got[0], got[1]] + got[2]
    lib = [lib[0], lib[1]] + lib[2]
    names = ["y", "dx"] + [k for k, _ in seq.named_parameters()]
    worst = 0.0
    for name, r, g, t in zip(names, ref, got, lib):
        scale = float(r.abs().max()) + 1e-30
        eg = float((g.double().cpu() - r).abs().max()) / scale
        et = float((t.double().cpu() - r).abs().max()) / scale
        worst = max(worst, eg)
        assert eg <= 4.0 * et + 1e-6, (kind, rows, name, eg, et)
    print(f"[mlp_rows] {kind} rows={rows}: worst relative error {worst:.2e}")


def test_mlp_rows_direct_gradients_and_frozen_layers():
    """the trainers' route: parameter gradients written into persistent .grad buffers (overwritten, not accumulated);
    a frozen network (no parameter gradient asked for) still hands back the input gradient"""
    torch.manual_seed(3)
    seq = _net("brdf_dec").cuda()
    x, cot = torch.randn(300, 32, device="cuda"), torch.randn(300, 4, device="cuda")
    owner = _Owner()
    _, dx_a, grads_a = _run(seq, x, cot, owner, True)
    for p in seq.parameters():
        p.grad = torch.full_like(p, 123.0)            # stale contents: the kernels overwrite
    owner.direct_grads = True
    _, dx_b, grads_b = _run(seq, x, cot, owner, True)
    assert torch.equal(dx_a, dx_b)
    for a, b in zip(grads_a, grads_b):
        assert torch.equal(a, b)                       # the same launches, bit for bit
    owner.direct_grads = False
    for p in seq.parameters():
        p.requires_grad_(False)
        p.grad = None
    from models import fields
    xg = x.clone().requires_grad_(True)
    (fields._seq_direct(seq, xg, owner) * cot).sum().backward()
    assert torch.equal(xg.grad, dx_a)
    assert all(p.grad is None for p in seq.parameters())


def test_mlp_rows_reproducible_and_no_rows():
    torch.manual_seed(5)
    from models import fields
    seq = _net("net_cs").cuda()
    owner = _Owner()
    x, cot = torch.randn(512, 90, device="cuda"), torch.randn(512, 1, device="cuda")
    a = _run(seq, x, cot, owner, True)
    b = _run(seq, x, cot, owner, True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and all(torch.equal(p, q) for p, q in zip(a[2], b[2]))
    e = _run(seq, x[:0], cot[:0], owner, True)
    assert e[0].shape == (0, 1) and e[1].shape == (0, 90) and all(float(g.abs().max()) == 0.0 for g in e[2])
    with torch.no_grad():                                # frozen use (stage 3's IndirectLight): forward only
        y = fields._seq_direct(seq, x, owner)
    assert torch.equal(y, a[0])


def test_mlp_group_equals_one_by_one():
    """networks of different depth and row count in lockstep (stage 2: Lvis + IndirectLight; stage 3: BRDF encoder + net_cs):
    the same launches' arithmetic, bit for bit, as each network on its own"""
    torch.manual_seed(11)
    from models import fields
    owner = _Owner()
    kinds = [("lvis", 2048), ("indi", 512), ("brdf_dec", 300), ("one", 0)]
    seqs = [_net(k).cuda() for k, _ in kinds]
    xs = [torch.randn(r, s[0].in_features, device="cuda") for (_, r), s in zip(kinds, seqs)]
    xs[2].requires_grad_(True)
    single = []
    for s, x in zip(seqs, xs):
        y = fields._seq_direct(s, x, owner)
        y.square().sum().backward()
        single.append((y.detach(), [p.grad.clone() for p in s.parameters()], None if x.grad is None else x.grad.clone()))
        for p in s.parameters():
            p.grad = None
        x.grad = None
    ys = fields.seq_group([(s, x, owner) for s, x in zip(seqs, xs)])
    sum(y.square().sum() for y in ys).backward()
    for (y1, g1, dx1), y, s, x in zip(single, ys, seqs, xs):
        assert torch.equal(y1, y.detach())
        for a, p in zip(g1, s.parameters()):
            assert torch.equal(a, p.grad)
        assert (dx1 is None) == (x.grad is None) and (dx1 is None or torch.equal(dx1, x.grad))


def test_mlp_rows_rejects_missing_arguments():
    from fneus import ops
    w = torch.randn(4, 8, device="cuda")
    with pytest.raises(RuntimeError, match="fneus_mlp"):
        ops.mlp_forward([dict(weight=w, rows=3, n_in=8, n_out=4)])            # no x, no y
    with pytest.raises(RuntimeError, match="fneus_mlp"):
        ops.mlp_backward_params([dict(weight=w, x=torch.randn(3, 8, device="cuda"), rows=3, n_in=8, n_out=4)])
    x = torch.randn(3, 8, device="cuda")
    with pytest.raises(RuntimeError, match="2 GiB"):            # 32-bit operand offsets: refused, not wrapped around
        ops.mlp_forward([dict(x=x, weight=w, y=torch.empty(3, 4, device="cuda"), rows=2 ** 27, n_in=8, n_out=4)])
