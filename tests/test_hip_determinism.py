"""Repeated launches of the chain kernels must be bit-identical (no atomics on these paths): a guard against data races
between the wavefronts of a workgroup (tests/checkers/dbg_race.py found one in the opt-in tensor-parallel K2 this way)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _net():
    from fneus import ops, synth
    from oracle import ref_torch as R
    dev = torch.device("cuda:0")
    sd = {k: torch.from_numpy(v) for k, v in synth.sdf_state_dict(20).items()}
    p = R.sdf_params_from_state_dict(sd)
    net = ops.PackedNet("sdf", dev)
    net.set_raw_from_effective([w.to(dev) for w in p["W"]], [b.to(dev) for b in p["b"]])
    net.pack()
    return net, dev


@pytest.mark.parametrize("n", [8192, 40000])          # tensor-parallel small launch / one wave per tile
def test_sdf_fwd_is_reproducible(n):
    from fneus import ops
    net, dev = _net()
    x = (torch.rand(n, 3, device=dev) * 2 - 1).contiguous()
    ref = ops.sdf_fwd(net.blob, n, 3, pts=x).clone()
    for _ in range(40):
        assert torch.equal(ops.sdf_fwd(net.blob, n, 3, pts=x), ref)


def test_sdf_fwd_grad_and_bwd_are_reproducible():
    from fneus import ops
    net, dev = _net()
    n = 32768
    x = (torch.rand(n, 3, device=dev) * 2 - 1).contiguous()
    ds, df, dn = torch.randn(n, device=dev), torch.randn(n, 256, device=dev), torch.randn(n, 3, device=dev)

    def run():
        st = ops.SdfStash(n, dev, 3, True)
        bufs = ops.SdfBwdBufs(n, dev, 3)
        for t in (st.h, st.a, bufs.adj, bufs.zbar):      # slot 3 of these planes has 224 valid columns: define the rest
            t.zero_()
        sdf, feat, nrm = ops.sdf_fwd_grad(net.blob, n, 3, st, True, pts=x)
        ops.sdf_bwd(net.blob, n, 3, st, bufs, ds, df, dn, pts=x)
        torch.cuda.synchronize()
        return [t.clone() for t in (sdf, feat, nrm, st.h, st.a, bufs.adj, bufs.zbar)]

    ref = run()
    for _ in range(12):
        for a, b, name in zip(run(), ref, ("sdf", "feat", "normal", "h", "a", "adj", "zbar")):
            assert torch.equal(a, b), name


@pytest.mark.parametrize("prec", [3, 1])
def test_color_forward_backward_are_reproducible(prec):
    """K4 in its tensor-parallel form reads its B operands from LDS like K2: same guard"""
    from fneus import ops, synth
    dev = torch.device("cuda:0")
    n = 65536
    cnet = ops.PackedNet("color", dev).load_state_dict({k: torch.from_numpy(v) for k, v in synth.color_state_dict(21).items()})
    cnet.pack()
    g = torch.Generator(device=dev).manual_seed(3)
    x = (torch.rand(n, 3, device=dev, generator=g) * 2 - 1).contiguous()
    d = torch.nn.functional.normalize(torch.randn(n, 3, device=dev, generator=g), dim=-1).contiguous()
    nrm = torch.randn(n, 3, device=dev, generator=g)
    feat = (torch.randn(n, 256, device=dev, generator=g) * 0.3).contiguous()
    c = torch.randn(n, 3, device=dev, generator=g)
    st = ops.ColStash(n, dev, prec)

    def run():
        rgb = ops.color_fwd(cnet.blob, n, prec, nrm, feat, st, True, pts=x, dirs=d)
        d_feat, d_normal = ops.color_bwd(cnet.blob, n, prec, c, rgb, st)
        torch.cuda.synchronize()
        return [t.clone() for t in (rgb, d_feat, d_normal, st.u.view(torch.int16), st.zbar.view(torch.int16), st.mask,
                                    st.side.view(torch.int16))]

    # `side` = the encodings of the side inputs as the kernel stored them: round 4 saw them differ from run to run in the units a
    # workgroup encodes BETWEEN two passes (units >= 256 of this launch) with an SLP-packed fp32 sincos (fneus_common.h fn_sincos;
    # tools/experiments/r04/col_repro_dbg.py is this loop)
    ref = run()
    for it in range(60):
        for a, b, name in zip(run(), ref, ("rgb", "d_feat", "d_normal", "u", "zbar", "mask", "side")):
            assert torch.equal(a, b), (name, it, int((a != b).sum()))


@pytest.mark.parametrize("use_graph", [False, True])
def test_whole_train_step_is_bit_reproducible_in_deterministic_mode(use_graph):
    """FNEUS_DETERMINISTIC=1 / ops.set_deterministic(True): the weight-gradient GEMM adds its split-K partials in split order
    (fneus_dw_gemm_pp_det) instead of with fp32 atomics -- the only order-dependent sums of the step.  Two trainers with the
    same seed on the same batches then hold bit-identical parameters after every step (the reference's CPU loop,
    exp_runner.py:179-181, is deterministic as well); in the default mode they differ in the last bits."""
    from fneus import ops
    from fneus.trainer import Stage1Trainer, synthetic_batches
    dev = torch.device("cuda:0")
    batches = synthetic_batches(6, 512, dev, seed0=77)

    def run(det):
        ops.set_deterministic(det)
        torch.manual_seed(1234)                       # the depth jitter of render() draws from torch's generator
        try:
            tr = Stage1Trainer(dev, prec=ops.PREC_PARITY, seed=31, use_graph=use_graph)
            out = []
            for b in batches:
                tr.train_step(b)
                torch.cuda.synchronize()
                out.append(torch.cat([p.detach().reshape(-1) for p in tr.params]).clone())
            return out
        finally:
            ops.set_deterministic(None)

    a, b = run(True), run(True)
    for i, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x, y), f"step {i}: {(x != y).sum().item()} of {x.numel()} parameters differ"
    # the deterministic sums equal the atomic ones to rounding: same trajectory to ~1e-6 relative after the first step
    c = run(False)
    d = (a[0] - c[0]).abs().max().item()
    print(f"deterministic vs atomic mode after one step: max |difference| {d:.2e}")
    assert d <= 5e-6


def test_deterministic_gemm_equals_the_atomic_one():
    from fneus import ops
    net, dev = _net()
    n = 20000
    x = (torch.rand(n, 3, device=dev) * 2 - 1).contiguous()
    ds, df, dn = torch.randn(n, device=dev), torch.randn(n, 256, device=dev), torch.randn(n, 3, device=dev)
    st, bufs = ops.SdfStash(n, dev, 3, True), ops.SdfBwdBufs(n, dev, 3)
    for t in (st.h, st.a, bufs.adj, bufs.zbar):
        t.zero_()
    ops.sdf_fwd_grad(net.blob, n, 3, st, True, pts=x)
    ops.sdf_bwd(net.blob, n, 3, st, bufs, ds, df, dn, pts=x)
    grads = []
    for det in (False, True, True):
        ops.set_deterministic(det)
        try:
            g = torch.zeros(net.n_params, dtype=torch.float32, device=dev)
            ops.sdf_dw_jobs(net, st, bufs, g, n).run()
            torch.cuda.synchronize()
            grads.append(g)
        finally:
            ops.set_deterministic(None)
    assert torch.equal(grads[1], grads[2])
    scale = grads[0].abs().max().item()
    assert (grads[0] - grads[1]).abs().max().item() <= 2e-6 * scale
