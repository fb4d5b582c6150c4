"""CPU-side checks of the C-ABI shared library: it loads and exports every symbol include/fneus.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "fneus.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fneus_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_all_declared_symbols():
    from fneus import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 6
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/fneus.h but not exported"
    assert lib.fneus_version() >= 100


def test_layout_query_matches_host_description():
    from fneus import netdesc
    for which, ins, outs in ((0, netdesc.SDF_IN, netdesc.SDF_OUT), (1, netdesc.COL_IN, netdesc.COL_OUT)):
        ly = netdesc.query_layout(which)
        assert ly.n_layers == len(ins)
        for l in range(ly.n_layers):
            ksf, ntf, ksr, ntr = ly.geom[l]
            assert ntf * 32 >= outs[l] and ksf * 16 >= ins[l]
            assert ksr == 2 * ntf or (which == 0 and l == 3 and ksr == 14) or ksr * 16 >= outs[l]
        assert ly.total > 0 and ly.total % 16 == 0


def test_pack_job_tables():
    import numpy as np
    from fneus import netdesc
    for d in (netdesc.build_sdf_jobs(), netdesc.build_color_jobs()):
        jobs, maps = d["jobs"], d["maps"]
        assert jobs.dtype.itemsize == 64
        assert (np.diff(jobs["unit_base"]) > 0).all()
        # every parameter is referenced by at least one forward fragment map
        assert maps.max() < max(max(d["ins"]), max(d["outs"]))
        assert d["units"] == int(jobs["unit_base"][-1]) + int(jobs["nt"][-1])
        rows = d["rows"]
        assert rows.dtype.itemsize == 16 and len(rows) == sum(d["outs"])
        assert d["n_raw"] == sum(o * i + 2 * o for i, o in zip(d["ins"], d["outs"]))
