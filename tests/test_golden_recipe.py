"""The golden recipe itself: tests/golden/gen_golden.py must run at HEAD against the REFERENCE (not against this repo's
own `models` package, which once shadowed it) and reproduce committed fixtures bit for bit.  Needs /root/reference, i.e.
the build container; skipped on the GPU box."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree only exists in the build container")
def test_generator_imports_the_reference_and_reproduces_committed_fixtures(tmp_path):
    gen = os.path.join(ROOT, "tests", "golden", "gen_golden.py")
    # two small cases: a render fixture (one Adam step) and the stage-2/3 entry; --check compares every committed key
    r = subprocess.run([sys.executable, gen, "--out", str(tmp_path), "--only", "render_wmask_b16_n16_c0,lvis_util_b24_n32,raygen_dtu,raygen_shiny",
                        "--check"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "committed fixtures reproduced bit for bit" in r.stdout
    assert "render_wmask_b16_n16_c0: " in r.stdout and " 0 differ" in r.stdout


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree only exists in the build container")
def test_generator_refuses_the_repo_package():
    """with the repo's package directory on sys.path the import guard must still pick the reference's modules"""
    code = (f"import sys; sys.path.insert(0, {os.path.join(ROOT, 'factored-neus_amd')!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests', 'golden')!r});"
            "import gen_golden as g; e, f, r = g.import_reference()[:3];"
            "assert f.__file__.startswith('/root/reference/'), f.__file__; print('ok', f.__file__)")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok /root/reference/models/fields.py" in r.stdout, r.stdout + r.stderr[-1500:]
