"""The LAMMPS style adapters (adapters/lammps: the classes interfaceToLammps/style_user.h:43-74 registers) EXECUTED once:
tests/adapters/lammps_host/adapter_host.cpp builds the LAMMPS-side objects (tests/adapters/lammps_min declarations) from a
case of tests/golden/reference_pins.json, instantiates PairGranHertzFixHistoryAmd / FixCoheAmd / FixFluidDragAmd /
PairLubricatePolyAmd, calls settings / init / init_list / compute / post_force in LAMMPS' order, and the f / torque / touch /
shear the adapter leaves in LAMMPS' own arrays are compared with what the reference's lines computed on the same inputs
(pair_gran_hertzFix_history.cpp:45-287, fix_cohesive.cpp:138-263, fix_fluid_drag.cpp:114-164, pair_lubricate_poly.cpp:65-444).
Tolerances as tests/test_reference_pins_gpu.py (1e-12; lubricate 1e-11)."""
import glob
import os
import subprocess

import numpy as np
import pytest

from tests import dem_cases as dc
from tests.test_reference_pins import LUB_KEY, PINS, unhex

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GRAN_KEY, COHE_KEY, FDRAG_KEY = "pair_gran_hertzFix_history.cpp:109-286", "fix_cohesive.cpp:161-262", "fix_fluid_drag.cpp:143-163"


def build_host(outdir):
    """g++ only (the adapters are what LAMMPS' host compiler would see: no HIP headers), linked against the C-ABI library"""
    exe = os.path.join(str(outdir), "adapter_host")
    srcs = [os.path.join(ROOT, "tests", "adapters", "lammps_host", "adapter_host.cpp")] + \
        sorted(glob.glob(os.path.join(ROOT, "adapters", "lammps", "*.cpp")))
    libdir = os.path.join(ROOT, "sedifoam_amd")
    r = subprocess.run(["g++", "-std=gnu++98", "-O1", "-Wall", "-Wno-unused-variable",
                        "-I", os.path.join(ROOT, "tests", "adapters", "lammps_min"), "-I", os.path.join(ROOT, "include"),
                        "-I", os.path.join(ROOT, "adapters", "lammps")] + srcs +
                       ["-L", libdir, "-lsedifoam_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    return build_host(tmp_path_factory.mktemp("adapter_host"))


def write_case(path, I, extra=None):
    rows = {}
    for k, v in dict(I, **(extra or {})).items():
        if k in ("firstneigh", "touch", "shear") or v is None or isinstance(v, (dict, str)):
            continue
        rows[k] = np.asarray(v, dtype=np.float64).reshape(-1)
    if "firstneigh" in I:
        nl = I["nlocal"]
        rows["numneigh"] = np.array([len(I["firstneigh"][i]) for i in range(nl)], dtype=np.float64)
        rows["jlist"] = np.array([j for i in range(nl) for j in I["firstneigh"][i]], dtype=np.float64)
        if "touch" in I:
            rows["touch"] = np.array([t for i in range(nl) for t in I["touch"][i]], dtype=np.float64)
            rows["shear"] = np.array([s for i in range(nl) for s in I["shear"][i]], dtype=np.float64).reshape(-1)
    with open(path, "w") as f:
        for k, a in rows.items():
            f.write("%s %d %s\n" % (k, a.size, " ".join(repr(float(x)) for x in a)))


def read_out(path):
    out = {}
    for line in open(path):
        w = line.split()
        vals = w[2:2 + int(w[1])]
        out[w[0]] = np.array([float.fromhex(x) if ("x" in x or "nan" in x or "inf" in x) else float(x) for x in vals])
    return out


def run(host, kind, tmp_path, I, extra=None):
    case, res = str(tmp_path / "case.txt"), str(tmp_path / "out.txt")
    write_case(case, I, extra)
    r = subprocess.run([host, kind, case, res], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stderr)
    return read_out(res)


@pytest.mark.parametrize("k", range(len(PINS[GRAN_KEY])))
def test_pair_gran_hertzfix_history_adapter_compute(host, tmp_path, k):
    c = PINS[GRAN_KEY][k]
    I, O = c["inp"], c["out"]
    got = run(host, "pair_gran", tmp_path, I)
    n, nl = I["n"], I["nlocal"]
    assert dc.rel_err(got["f"].reshape(n, 3), unhex(O["f"])) <= 1e-12
    assert dc.rel_err(got["torque"].reshape(n, 3), unhex(O["torque"])) <= 1e-12
    ref_touch = np.array([t for i in range(nl) for t in O["touch"][i]], dtype=np.int64)
    ref_shear = np.array([float.fromhex(s) for i in range(nl) for s in O["shear"][i]])
    assert np.array_equal(got["touch"].astype(np.int64), ref_touch)   # back in FixShearHistory's pages
    if ref_shear.size:
        assert dc.rel_err(got["shear"], ref_shear) <= 1e-12
    if "mass_rigid" in I:   # (:68-86: the body masses went to the ghosts through the pair's forward communication)
        assert int(got["forward_comm_pair_calls"][0]) == 1


@pytest.mark.parametrize("k", range(len(PINS[COHE_KEY])))
def test_fix_cohesive_adapter_post_force(host, tmp_path, k):
    c = PINS[COHE_KEY][k]
    I, O = c["inp"], c["out"]
    got = run(host, "fix_cohesive", tmp_path, I)
    assert dc.rel_err(got["f"].reshape(I["n"], 3), unhex(O["f"])) <= 1e-12


@pytest.mark.parametrize("k", range(len(PINS[FDRAG_KEY])))
def test_fix_fdrag_adapter_post_force(host, tmp_path, k):
    c = PINS[FDRAG_KEY][k]
    I, O = c["inp"], c["out"]
    got = run(host, "fix_fdrag", tmp_path, I)
    assert dc.rel_err(got["f"].reshape(I["n"], 3), unhex(O["f"])) <= 1e-13
    assert np.array_equal(got["vOld"].reshape(I["n"], 3), unhex(O["vOld"]))
    assert int(got["exchange_ok"][0]) == 1


@pytest.mark.parametrize("k", range(len(PINS[LUB_KEY])))
def test_pair_lubricate_poly_adapter_compute(host, tmp_path, k):
    c = PINS[LUB_KEY][k]
    I, O = c["inp"], c["out"]
    extra = {q: float.fromhex(O[q]) for q in ("R0", "RT0", "RS0")}
    got = run(host, "pair_lubricate", tmp_path, I, extra)
    assert dc.rel_err_nan(got["f"].reshape(I["n"], 3), unhex(O["f"])) <= 1e-11
    assert dc.rel_err_nan(got["torque"].reshape(I["n"], 3), unhex(O["torque"])) <= 1e-11
