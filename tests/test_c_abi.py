"""include/sedifoam_amd.h used from C and from C++ with the reference's own function names
(SEDIFOAM_AMD_LAMMPS_NAMES): tests/c_abi/dropin_driver.c is written like a caller of interfaceToLammps/library.h.
CPU: it compiles and links against libsedifoam_amd.so in both languages.  GPU: it runs."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "dropin_driver.c")
LIBDIR = os.path.join(ROOT, "sedifoam_amd")


def _build(tmp_path, compiler, extra):
    exe = str(tmp_path / ("dropin_" + compiler))
    cmd = [compiler] + extra + ["-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
                                "-L", LIBDIR, "-lsedifoam_amd", "-Wl,-rpath," + LIBDIR, "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.parametrize("compiler,extra", [("gcc", ["-std=c99"]), ("g++", ["-x", "c++", "-std=c++11"])])
def test_header_compiles_and_links_as_c_and_cxx(tmp_path, compiler, extra):
    if not os.path.exists(os.path.join(LIBDIR, "libsedifoam_amd.so")):
        pytest.skip("library not built")
    _build(tmp_path, compiler, extra)


@pytest.mark.gpu
def test_c_caller_runs_a_bed_through_the_library_names(tmp_path):
    d = 5.0e-4
    pts = [(0.5 * d + ix * 1.02 * d, 0.5 * d + iy * 1.02 * d, 0.5 * d + iz * 1.02 * d)
           for iy in range(4) for ix in range(6) for iz in range(6)]
    data = tmp_path / "bed.in"
    with open(data, "w") as f:
        f.write(" sphere data\n\n %d atoms\n 1 atom types\n\n 0.0 %g xlo xhi\n 0.0 %g ylo yhi\n 0.0 %g zlo zhi\n\nAtoms\n\n"
                % (len(pts), 6.12 * d, 8.0 * d, 6.12 * d))
        for k, p in enumerate(pts):
            f.write(" %d 1 %g 2650 %.12g %.12g %.12g\n" % (k + 1, d, p[0], p[1], p[2]))
    exe = _build(tmp_path, "gcc", ["-std=c99"])
    r = subprocess.run([exe, str(data)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    tok = r.stdout.strip().split()
    assert tok[0] == "OK" and int(tok[1]) == len(pts)
    assert float(tok[3]) > float(tok[2])          # net upward force 2 m g - m g: the bed rises
