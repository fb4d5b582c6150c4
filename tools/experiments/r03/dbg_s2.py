import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/factored-neus_amd'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
import test_hip_stage2 as t
g = t.load('/root/repo/tests/golden', 'lvis_render_room_b24_n32')
for env in ({"FNEUS_K1_W8_BIG":"31","FNEUS_K1_W8_SMALL":"2"}, {"FNEUS_K1_W8_BIG":"0","FNEUS_K1_W8_SMALL":"0"}):
    os.environ.update(env)
    tr = t.build(g)
    data = t.T(g["data"]).to(t.DEV)
    out = tr.renderer.lvis_render(data[:, :3].contiguous(), data[:, 3:6].contiguous(), t.T(g["near"]).to(t.DEV), t.T(g["far"]).to(t.DEV),
                                  u_theta=t.T(g["step0/u_theta"]).to(t.DEV), u_z=t.T(g["step0/u_z"]).to(t.DEV))
    m = t.T(g["out/sdf_mask"])
    r = (out["gt_trace_radiance"] - out["pre_trace_radiance"]).detach().cpu()[m]
    rr = (t.T(g["out/gt_trace_radiance"]) - t.T(g["out/pre_trace_radiance"]))[m]
    flips = (torch.sign(r) != torch.sign(rr)).sum().item()
    print(env, "min |gt-pre| trace", r.abs().min().item(), "sign flips vs reference", flips, "of", r.numel(),
          "max |gt diff|", (out["gt_trace_radiance"].detach().cpu() - t.T(g["out/gt_trace_radiance"])).abs().max().item())
    r2 = (out["gt_lvis"] - out["pre_lvis"]).detach().cpu()[m]; rr2 = (t.T(g["out/gt_lvis"]) - t.T(g["out/pre_lvis"]))[m]
    print("   lvis flips", (torch.sign(r2) != torch.sign(rr2)).sum().item())
