"""GPU parity: the fragment-plane weight-gradient GEMM (fneus_dw_gemm_pp) vs fp64 matmul of the same bf16 operands.

The reference computes these products as torch autograd's addmm backward (fields.py:86) in fp32; here the operands are
the bf16 planes the chain kernels write, so the check is exact arithmetic on identical inputs: fp32 accumulation error only."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(rs, n, w, dev, scale=1.0):
    return torch.from_numpy((rs.standard_normal((n, w)) * scale).astype(np.float32)).to(dev)


@pytest.mark.parametrize("gprec", [1, 3])
@pytest.mark.parametrize("n", [32, 1000, 4096 + 17])
def test_gemm_pp_two_terms(gprec, n):
    from fneus import ops, pp
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(n + gprec)
    P = 2 if gprec == 3 else 1
    Z, U, A, D = _mk(rs, n, 256, dev), _mk(rs, n, 256, dev), _mk(rs, n, 224, dev), _mk(rs, n, 48, dev)
    zp, up, ap, dp = pp.pack(Z, 16, P), pp.pack(U, 16, P), pp.pack(A, 16, P), pp.pack(D, 4, P)
    T = pp.n_tiles(n)
    # the value the kernel sees (bf16 hi, or hi + lo)
    Zv, Uv, Av, Dv = (pp.value(x, n).double() for x in (zp, up, ap, dp))
    grad = torch.zeros(256 * 256 + 256 + 217 * 39 + 300, dtype=torch.float32, device=dev)
    base = grad.data_ptr()
    jobs = ops.GemmPPJobs(dev, "test", target_wgs=64)
    # job 0: 256 x 256, two terms, bias;  job 1: 217 x 39 (ragged rows / columns), one term, scale, different ldc
    jobs.add(ops.PPOperand(zp, 0, 8), ops.PPOperand(up, 0, 8), base, 256, 256, 256,
             A2=ops.PPOperand(ap, 0, 8), B2=ops.PPOperand(zp, 0, 8), bias_ptr=base + 4 * 65536)
    off1 = 65536 + 256
    jobs.add(ops.PPOperand(ap, 0, 7), ops.PPOperand(dp, 0, 2), base + 4 * off1, 39, 217, 39, scale=0.5)
    jobs.finalize(T).run(gprec=gprec)
    torch.cuda.synchronize()
    A256 = torch.zeros(n, 256, dtype=torch.float64, device=dev)
    A256[:, :224] = Av[:, :224]
    ref0 = Zv.T @ Uv + A256.T @ Zv
    refb = Zv.sum(0)
    ref1 = 0.5 * (Av[:, :217].T @ Dv[:, :39])
    got0 = grad[:65536].view(256, 256).double()
    gotb = grad[65536:65536 + 256].double()
    got1 = grad[off1:off1 + 217 * 39].view(217, 39).double()
    tol = 2e-6 if gprec == 3 else 2e-6          # operands are identical: only fp32 accumulation differs
    for name, g, r in (("C0", got0, ref0), ("bias", gotb, refb), ("C1", got1, ref1)):
        e = ((g - r).norm() / r.norm()).item()
        print(f"gemm_pp gprec={gprec} n={n} {name} rel {e:.2e}")
        assert e <= (tol if gprec == 1 else 2e-5), name        # gprec 3 drops the lo*lo term: 2^-16 relative
    assert grad[off1 + 217 * 39:].abs().max().item() == 0.0      # nothing written past the ragged job


def test_gemm_pp_constant_block_and_offsets():
    """a constant A2 block (every tile the same: the implicit ones row of the sdf output) and operands that start at a
    fragment offset inside their block"""
    from fneus import ops, pp
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(5)
    n = 2048 + 5
    T = pp.n_tiles(n)
    H, ADJ = _mk(rs, n, 256, dev), _mk(rs, n, 256, dev)
    Zs = torch.zeros(n, 32, device=dev)
    Zs[:, 0] = _mk(rs, n, 1, dev)[:, 0]
    hp, adjp, zsp = pp.pack(H, 16), pp.pack(ADJ, 16), pp.pack(Zs, 2)
    ones = torch.zeros(32, 32, device=dev)
    ones[:, 0] = 1.0
    onesp = pp.pack(ones, 2)                                   # [1, 1, 2, 64, 8]: one block, reused for every tile
    grad = torch.zeros(257 * 256 + 16, dtype=torch.float32, device=dev)
    jobs = ops.GemmPPJobs(dev, "t", target_wgs=16)
    jobs.add(ops.PPOperand(zsp, 0, 1), ops.PPOperand(hp, 0, 8), grad.data_ptr(), 256, 1, 256,
             A2=ops.PPOperand(onesp, 0, 1, const=True), B2=ops.PPOperand(adjp, 0, 8), bias_ptr=grad.data_ptr() + 4 * 257 * 256)
    # columns 64..191 of H against columns 32..95 of ADJ (fragment offsets 4 and 2)
    jobs.add(ops.PPOperand(hp, 4, 4), ops.PPOperand(adjp, 2, 2), grad.data_ptr() + 4 * 256, 256, 128, 64)
    jobs.finalize(T).run(gprec=1)
    torch.cuda.synchronize()
    Hv, ADJv, Zv = pp.value(hp, n).double(), pp.value(adjp, n).double(), pp.value(zsp, n).double()
    ref_row = Zv[:, 0] @ Hv + ADJv.sum(0)          # NOTE: the ragged tile's padding rows are zero in ADJ, so the ones block adds nothing
    got_row = grad[:256].double()
    assert ((got_row - ref_row).norm() / ref_row.norm()).item() <= 2e-6
    assert abs(grad[257 * 256].item() - Zv[:, 0].sum().item()) <= 1e-3
    ref2 = Hv[:, 64:192].T @ ADJv[:, 32:96]
    got2 = grad[256:256 + 128 * 256].view(128, 256)[:, :64].double()
    assert ((got2 - ref2).norm() / ref2.norm()).item() <= 2e-6
