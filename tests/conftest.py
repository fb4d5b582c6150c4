import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "factored-neus_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def write_shiny_case(case_dir, g, ball):
    """the files of the Shiny-Blender case held by tests/golden/raygen_shiny.npz (PNG and float TIFF are lossless: the repo's
    loader then reads the same bytes the reference's DatasetShiny read when the fixture was made; tests/golden/gen_golden.py
    write_shiny_case)"""
    from PIL import Image
    os.makedirs(case_dir, exist_ok=True)
    with open(os.path.join(case_dir, "transforms_train.json"), "w") as fp:
        fp.write(str(g["meta"]))
    for i in range(g["png"].shape[0]):
        Image.fromarray(g["png"][i]).save(os.path.join(case_dir, "r_%d.png" % i))
        if ball:
            Image.fromarray(g["alpha"][i]).save(os.path.join(case_dir, "r_%d_alpha.png" % i))
        else:
            Image.fromarray(g["disp"][i]).save(os.path.join(case_dir, "r_%d_disp.tiff" % i))
    return case_dir
