"""Round 6: weight-gradient error of the cotangent chains on bf16 activations (FNEUS_BWD_XHI=1, FNEUS_COLB_XHI=1) beside hi + lo
activations (=0), on launches that take the resident-weight kernels (>= 1024 sample tiles):
 (a) the SDF network's double backward at 65 536 points with RANDOM cotangents against fp64 autograd of the oracle (the worst
     case for rounding: no cancellation against the sum), per layer;
 (b) the golden gradients of the 512-ray fixture (the reference's own loss), per tensor of the SDF and colour networks."""
import os, sys
root = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "factored-neus_amd")); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
T = lambda a: torch.from_numpy(np.asarray(a))
from fneus import ops, synth
from oracle import ref_torch as R
dev = torch.device("cuda:0")
SW = [s for s in os.environ.get("XHI_SWITCHES", "FNEUS_BWD_XHI,FNEUS_COLB_XHI").split(",") if s]

def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()

def part_a(n=65536):
    sd = {k: T(v) for k, v in synth.sdf_state_dict(20).items()}
    sp = R.sdf_params_from_state_dict(sd)
    net = ops.PackedNet("sdf", dev)
    net.set_raw_from_effective([w.to(dev) for w in sp["W"]], [b.to(dev) for b in sp["b"]]); net.pack()
    rs = np.random.RandomState(6)
    x = T(rs.uniform(-1.1, 1.1, size=(n, 3)).astype(np.float32))
    c_s, c_f, c_n = T(rs.standard_normal((n, 1)).astype(np.float32)), T((rs.standard_normal((n, 256)) * 0.05).astype(np.float32)), T(rs.standard_normal((n, 3)).astype(np.float32))
    p64 = {"W": [w.double().requires_grad_(True) for w in sp["W"]], "b": [b.double().requires_grad_(True) for b in sp["b"]], "scale": 1.0}
    tot = 0
    for i in range(0, n, 8192):
        sl = slice(i, i + 8192)
        s_r, f_r, n_r, _ = R.sdf_value_feature_normal(x[sl].double(), p64)
        ((s_r * c_s[sl].double()).sum() + (f_r * c_f[sl].double()).sum() + (n_r * c_n[sl].double()).sum()).backward()
    xd = x.to(dev).contiguous()
    for gprec in (1, 3):
        for v in ("1", "0"):
            for s in SW: os.environ[s] = v
            stash = ops.SdfStash(n, dev, 3, train=True, gprec=gprec)
            ops.sdf_fwd_grad(net.blob, n, 3, stash, True, pts=xd)
            bufs = ops.SdfBwdBufs(n, dev, 3, gprec=gprec)
            ops.sdf_bwd(net.blob, n, 3, stash, bufs, c_s.to(dev).reshape(-1).contiguous(), c_f.to(dev).contiguous(), c_n.to(dev).contiguous(), pts=xd)
            grad = torch.zeros(net.n_params, dtype=torch.float32, device=dev)
            ops.sdf_dw_jobs(net, stash, bufs, grad, n).run()
            torch.cuda.synchronize()
            dWs, dbs = net.split_flat(grad)
            print(f"(a) n={n} gprec={gprec} XHI={v}: dW " + " ".join(f"{rel(dWs[l], p64['W'][l].grad):.2e}" for l in range(9)))
            print(f"                          db " + " ".join(f"{rel(dbs[l], p64['b'][l].grad):.2e}" for l in range(9)))
            if gprec == 3: break

def part_b(name="render_wmask_b512_n64"):
    import test_hip_render as TR
    g = TR.load(os.path.join(root, "tests", "golden"), name)
    for gprec in (2, 3):
        for v in ("1", "0"):
            for s in SW: os.environ[s] = v
            out, nets, _ = TR.run(g, 3, teacher_z=True, fused_loss=True, gprec=gprec)
            out["losses"]["loss"].backward()
            rows = []
            for key in g:
                if not key.startswith("grad_norm/"): continue
                pname = key[len("grad_norm/"):]; net, rest = pname.split(".", 1)
                if nets.get(net) is None or net not in ("sdf", "color", "var"): continue
                prm = dict(nets[net].named_parameters())[rest]
                ref_norm, ref_sub = float(g[key]), g["grad_sub/" + pname]
                sub = prm.grad.detach().cpu().reshape(-1)[::997].numpy()
                scale = max(ref_norm / np.sqrt(prm.numel()), np.abs(ref_sub).max(), 1e-7)
                rows.append((pname, np.abs(sub - ref_sub).max() / scale, abs(prm.grad.double().norm().item() - ref_norm) / (ref_norm + 1e-12)))
            ws = max(r[1] for r in rows if r[0].startswith("sdf")); wn = max(r[2] for r in rows if r[0].startswith("sdf"))
            cs = max(r[1] for r in rows if r[0].startswith("color")); cn = max(r[2] for r in rows if r[0].startswith("color"))
            print(f"(b) {name} gprec={gprec} XHI={v}: sdf worst sub {ws:.2e} norm {wn:.2e} | colour worst sub {cs:.2e} norm {cn:.2e}")
            print("      sdf e_sub by tensor: " + " ".join(f"{r[1]:.1e}" for r in rows if r[0].startswith("sdf")))
            if gprec == 3: break

part_a(); part_b()
