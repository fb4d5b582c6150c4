"""exp_runner.Runner end to end on the GPU with the synthetic dataset: train a few steps (graph replay), write a
checkpoint in the reference's format, resume from it, render a validation image and an SDF grid."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _conf(tmp_path):
    src = open(os.path.join(ROOT, "factored-neus_amd", "confs", "wmask.conf")).read()
    src = src.replace("./exp/CASE_NAME/", str(tmp_path) + "/exp/CASE_NAME/")
    path = os.path.join(tmp_path, "wmask_test.conf")
    open(path, "w").write(src)
    return path


def test_runner_train_checkpoint_resume_validate(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
    import exp_runner
    dev = torch.device("cuda:0")
    conf = _conf(str(tmp_path))
    r = exp_runner.Runner(conf, mode="train", case="synth", type="synthetic", device=dev)
    r.batch_size = 128
    r.save_freq, r.report_freq = 5, 3
    r.train(max_steps=5)
    assert r.iter_step == 5 and r.trainer.iter_step == 5
    assert len(r.trainer._graphs) == 1                     # wmask.conf: anneal_end = 0 -> the step is replayed as a graph
    ckpts = sorted(os.listdir(os.path.join(r.base_exp_dir, "checkpoints")))
    assert ckpts == ["ckpt_000005.pth"]
    ck = torch.load(os.path.join(r.base_exp_dir, "checkpoints", ckpts[0]), map_location="cpu")
    assert set(ck.keys()) >= {"nerf", "sdf_network_fine", "variance_network_fine", "color_network_fine", "refColor_network",
                              "optimizer", "iter_step"}                         # exp_runner.py:266-278
    assert "lin0.weight_g" in ck["sdf_network_fine"] and "net_cd.0.weight" in ck["refColor_network"]
    before = {k: v.clone() for k, v in r.trainer.sdf_network.state_dict().items()}
    step_before = float(r.trainer.optimizer.state[r.trainer.sdf_network.lin0.bias]["step"])
    # resume in a fresh runner: weights, optimiser moments and the step counter come back
    r2 = exp_runner.Runner(conf, mode="train", case="synth", is_continue=True, type="synthetic", device=dev)
    assert r2.iter_step == 5
    for k, v in r2.trainer.sdf_network.state_dict().items():
        assert torch.equal(v.cpu(), before[k].cpu()), k
    r2.batch_size = 128
    r2.train(max_steps=2)
    assert r2.iter_step == 7
    assert float(r2.trainer.optimizer.state[r2.trainer.sdf_network.lin0.bias]["step"]) == step_before + 2
    moved = max((a.cpu() - before[k].cpu()).abs().max().item() for k, a in r2.trainer.sdf_network.state_dict().items())
    assert 0.0 < moved < 1e-2
    img = r2.validate_image(idx=0, resolution_level=8)
    assert os.path.exists(img)
    from models.mesh import read_ply
    ply = r2.validate_mesh(resolution=48)
    assert ply.endswith("meshes/{:0>8d}.ply".format(r2.iter_step))  # exp_runner.py:530
    v, f = read_ply(ply)
    assert len(v) > 100 and len(f) > 100 and f.max() < len(v)
    # every vertex lies on the zero level set of the trained SDF (to the grid's interpolation error)
    with torch.no_grad():
        sdf_v = r2.trainer.sdf_network.sdf(torch.from_numpy(v).float().to(dev))
    assert sdf_v.abs().max().item() < 5e-3


@pytest.mark.parametrize("conf_name", ["wmask.conf", "womask.conf"])
def test_runner_on_a_dtu_format_case(tmp_path, conf_name):
    """--type dtu: the file loader (image/*.png, mask/*.png, cameras_sphere.npz) feeds the same step; womask.conf adds the
    background NeRF++ (K7), the white background and a ramping cos_anneal_ratio -- still one graph"""
    sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
    import exp_runner
    from models.dataset import SyntheticDataset, export_dtu_scene
    dev = torch.device("cuda:0")
    data_root = os.path.join(str(tmp_path), "data")
    export_dtu_scene(SyntheticDataset(n_images=4, H=48, W=64, device=torch.device("cpu"), seed=2),
                     os.path.join(data_root, "toy"), scale=20.0, offset=(1.0, 2.0, 3.0))
    src = open(os.path.join(ROOT, "factored-neus_amd", "confs", conf_name)).read()
    src = src.replace("./exp/CASE_NAME/", str(tmp_path) + "/exp/CASE_NAME/")
    import re
    src = re.sub(r"data_dir\s*=\s*\S+", "data_dir = " + data_root + "/CASE_NAME/", src)
    path = os.path.join(str(tmp_path), conf_name)
    open(path, "w").write(src)
    r = exp_runner.Runner(path, mode="train", case="toy", type="dtu", device=dev)
    assert r.dataset.n_images == 4 and (r.dataset.H, r.dataset.W) == (48, 64)
    r.batch_size = 128
    r.save_freq = r.report_freq = r.val_freq = r.val_mesh_freq = 10 ** 9
    r.train(max_steps=6)
    assert r.iter_step == 6 and len(r.trainer._graphs) == 1
    for p in r.trainer.params:
        assert torch.isfinite(p).all()
    img = r.validate_image(idx=1, resolution_level=4)
    assert os.path.exists(img)


def test_three_stage_pipeline_through_the_runners(tmp_path):
    """the reference's three command lines in sequence on one case (sh_dtu.sh): exp_runner.py (stage 1) -> lvis.py (stage 2,
    loads the stage-1 checkpoint from base_exp_dir_geo) -> mateIllu.py (stage 3, loads the stage-2 checkpoint), each writing
    checkpoints with the reference's keys; stage 2 and 3 resume from their own checkpoints"""
    sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
    import exp_runner
    import lvis
    import mateIllu
    dev = torch.device("cuda:0")
    conf = _conf(str(tmp_path))
    r1 = exp_runner.Runner(conf, mode="train", case="synth", type="synthetic", device=dev)
    r1.batch_size = 128
    r1.save_freq, r1.report_freq = 4, 10 ** 9
    r1.train(max_steps=4)
    # stage 2
    r2 = lvis.Runner(conf, mode="train", case="synth", type="synthetic", device=dev)
    for k, v in r1.trainer.sdf_network.state_dict().items():
        assert torch.equal(v, r2.trainer.sdf_network.state_dict()[k]), k          # geometry came from the stage-1 checkpoint
    r2.batch_size, r2.save_freq, r2.report_freq = 128, 3, 10 ** 9
    out = r2.train(max_steps=3)
    assert r2.iter_step == 3 and out is not None and bool(torch.isfinite(out["loss"]))
    ck = torch.load(os.path.join(r2.base_exp_dir_lvis, "checkpoints", "ckpt_000003.pth"), map_location="cpu")
    assert set(ck.keys()) == {"sdf_network_fine", "variance_network_fine", "color_network_fine", "refColor_network",
                              "lvis_network", "indiLgt_network", "optimizer", "iter_step"}          # lvis.py:255-266
    assert "lvis.0.weight" in ck["lvis_network"] and "indi.8.bias" in ck["indiLgt_network"]
    r2b = lvis.Runner(conf, mode="train", case="synth", is_continue=True, type="synthetic", device=dev)
    assert r2b.iter_step == 3
    assert torch.equal(r2b.trainer.lvis_network.state_dict()["lvis.4.weight"], r2.trainer.lvis_network.state_dict()["lvis.4.weight"])
    # stage 3
    r3 = mateIllu.Runner(conf, mode="train", case="synth", type="synthetic", device=dev)
    assert torch.equal(r3.trainer.lvis_network.state_dict()["lvis.4.weight"], r2.trainer.lvis_network.state_dict()["lvis.4.weight"])
    r3.batch_size, r3.save_freq, r3.report_freq = 128, 3, 10 ** 9
    out = r3.train(max_steps=3)
    assert r3.iter_step == 3 and out is not None and bool(torch.isfinite(out["loss"]))
    ck = torch.load(os.path.join(r3.base_exp_dir_mateIllu, "checkpoints", "ckpt_000003.pth"), map_location="cpu")
    assert set(ck.keys()) == {"sdf_network_fine", "variance_network_fine", "refColor_network", "lvis_network", "indiLgt_network",
                              "mateIllu_network", "optimizer", "iter_step"}                          # mateIllu.py:269-281
    assert set(ck["mateIllu_network"].keys()) >= {"lgtSGs", "brdf_encoder_layer.0.weight", "brdf_decoder_layer.4.bias",
                                                  "net_cs.8.weight"}
    assert "specular_reflectance" not in ck["mateIllu_network"]           # a plain attribute in the reference, not state
    r3b = mateIllu.Runner(conf, mode="train", case="synth", is_continue=True, type="synthetic", device=dev)
    assert r3b.iter_step == 3
    assert torch.equal(r3b.trainer.mateIllu_network.lgtSGs, r3.trainer.mateIllu_network.lgtSGs)


def test_runner_on_a_shiny_blender_format_case(tmp_path):
    """--type shiny with womask.conf (BASELINE config 5's loader): transforms_train.json + PNG + disparity TIFFs"""
    sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
    import re
    import exp_runner
    from models.dataset import SyntheticDataset, export_shiny_scene
    dev = torch.device("cuda:0")
    data_root = os.path.join(str(tmp_path), "data")
    export_shiny_scene(SyntheticDataset(n_images=4, H=48, W=64, device=torch.device("cpu"), seed=2), os.path.join(data_root, "toy"))
    src = open(os.path.join(ROOT, "factored-neus_amd", "confs", "womask.conf")).read()
    src = src.replace("./exp/CASE_NAME/", str(tmp_path) + "/exp/CASE_NAME/")
    src = re.sub(r"data_dir\s*=\s*\S+", "data_dir = " + data_root + "/CASE_NAME/", src)
    path = os.path.join(str(tmp_path), "womask.conf")
    open(path, "w").write(src)
    r = exp_runner.Runner(path, mode="train", case="toy", type="shiny", device=dev)
    assert r.dataset.n_images == 4 and (r.dataset.H, r.dataset.W) == (48, 64)
    r.batch_size = 128
    r.save_freq = r.report_freq = r.val_freq = r.val_mesh_freq = 10 ** 9
    r.train(max_steps=5)
    assert r.iter_step == 5 and all(bool(torch.isfinite(p).all()) for p in r.trainer.params)


def test_stage2_and_stage3_runners_under_torchrun_with_two_ranks(tmp_path):
    """`torchrun --nproc-per-node 2 lvis.py / mateIllu.py`: rays sharded by rank, the step replayed as a chain of graphs cut at
    the collectives, rank 0 writes the checkpoints (two ranks share the test box's GPU, so gloo instead of RCCL)"""
    import re
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
    import exp_runner
    conf = _conf(str(tmp_path))
    src = open(conf).read()
    src = re.sub(r"save_freq = \d+", "save_freq = 4", src)
    src = src.replace("lvis { batch_size = 512", "lvis { batch_size = 128").replace("metaIllu { batch_size = 512", "metaIllu { batch_size = 128")
    open(conf, "w").write(src)
    r1 = exp_runner.Runner(conf, mode="train", case="synth", type="synthetic", device=torch.device("cuda:0"))
    r1.batch_size = 128
    r1.train(max_steps=4)
    assert os.path.exists(os.path.join(r1.base_exp_dir, "checkpoints", "ckpt_000004.pth"))
    env = dict(os.environ, FNEUS_DIST_BACKEND="gloo")
    for i, (script, sub) in enumerate((("lvis.py", "lvis"), ("mateIllu.py", "mateIllu"))):
        port = 29400 + (os.getpid() + 61 * i) % 150
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "factored-neus_amd", script), "--conf", conf, "--case", "synth",
               "--type", "synthetic", "--max_steps", "8"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=env, cwd=str(tmp_path))
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        ck_dir = os.path.join(str(tmp_path), "exp", "synth", "wmask", sub, "checkpoints")
        assert sorted(os.listdir(ck_dir)) == ["ckpt_000004.pth", "ckpt_000008.pth"], os.listdir(ck_dir)
        ck = torch.load(os.path.join(ck_dir, "ckpt_000008.pth"), map_location="cpu")
        assert int(ck["iter_step"]) == 8


def test_stage1_runner_under_torchrun_with_two_ranks(tmp_path):
    """`torchrun --nproc-per-node 2 exp_runner.py`: every rank draws its own rays (seeded by rank), the gradients are summed,
    rank 0 writes the checkpoint (gloo: the two ranks share the test box's GPU)"""
    import re
    import subprocess
    conf = _conf(str(tmp_path))
    src = re.sub(r"save_freq = \d+", "save_freq = 6", open(conf).read()).replace("batch_size = 512\n", "batch_size = 128\n", 1)
    open(conf, "w").write(src)
    port = 29250 + os.getpid() % 140
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "factored-neus_amd", "exp_runner.py"), "--conf", conf, "--case", "synth",
           "--type", "synthetic", "--max_steps", "6"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=dict(os.environ, FNEUS_DIST_BACKEND="gloo"),
                       cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    ck_dir = os.path.join(str(tmp_path), "exp", "synth", "wmask", "geometry", "checkpoints")
    assert os.listdir(ck_dir) == ["ckpt_000006.pth"]
    ck = torch.load(os.path.join(ck_dir, "ckpt_000006.pth"), map_location="cpu")
    assert int(ck["iter_step"]) == 6 and all(torch.isfinite(v).all() for v in ck["sdf_network_fine"].values())
