"""adapters/: the reference's plug-in registrations (LAMMPS PairStyle / FixStyle, OpenFOAM dragModel) forwarding to
the C ABI.  LAMMPS and OpenFOAM are not installed here, so this checks what can be checked without them: every adapter
is valid C++98 against minimal declarations of the interfaces it touches (tests/adapters/*_min), registers the
reference's style names, and calls only symbols that include/sedifoam_amd.h declares."""
import glob
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAMMPS = sorted(glob.glob(os.path.join(ROOT, "adapters", "lammps", "*.cpp")))
FOAM = sorted(glob.glob(os.path.join(ROOT, "adapters", "openfoam", "*.C")))


@pytest.mark.parametrize("src", LAMMPS, ids=os.path.basename)
def test_lammps_adapter_compiles(src):
    r = subprocess.run(["g++", "-std=c++98", "-fsyntax-only", "-Wall", "-Werror", "-Wno-unused-variable",
                        "-I", os.path.join(ROOT, "tests", "adapters", "lammps_min"), "-I", os.path.join(ROOT, "include"),
                        "-I", os.path.join(ROOT, "adapters", "lammps"), src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.parametrize("src", FOAM, ids=os.path.basename)
def test_openfoam_adapter_compiles(src):
    r = subprocess.run(["g++", "-std=c++98", "-x", "c++", "-fsyntax-only", "-Wall", "-Werror",
                        "-I", os.path.join(ROOT, "tests", "adapters", "openfoam_min"), "-I", os.path.join(ROOT, "include"),
                        "-I", os.path.join(ROOT, "adapters", "openfoam"), src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_adapters_register_the_reference_style_names():
    """interfaceToLammps/style_user.h:43-50,65-74 and ErgunWenYu.C:35-42"""
    text = "".join(open(p).read() for p in glob.glob(os.path.join(ROOT, "adapters", "*", "*.[hH]")))
    for reg in ("PairStyle(gran/hertzFix/history,", "PairStyle(lubricate/poly,", "FixStyle(fdrag,",
                "FixStyle(cohesive,", 'TypeName("ErgunWenYu")'):
        assert reg in text, reg
    src = open(os.path.join(ROOT, "adapters", "openfoam", "ErgunWenYuAmd.C")).read()
    assert "addToRunTimeSelectionTable(dragModel, ErgunWenYuAmd, dictionary)" in src


def test_adapters_call_only_declared_c_abi_symbols():
    from sedifoam_amd import _lib
    declared = set(_lib.exported_symbols())
    used = set()
    for p in glob.glob(os.path.join(ROOT, "adapters", "*", "*")):
        if os.path.isfile(p):
            used |= set(re.findall(r"\b(sf[k]?_[a-z_0-9]+)\s*\(", open(p).read()))
    used -= {"sf_last_error"}
    assert used and used <= declared, used - declared
