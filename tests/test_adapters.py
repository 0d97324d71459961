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


@pytest.mark.parametrize("mpi", ["fake_mpi_int", "fake_mpi_ptr"])
@pytest.mark.parametrize("src", FOAM, ids=os.path.basename)
def test_openfoam_adapter_compiles(src, mpi):
    """(enhancedCloudAmd owns the LAMMPS object through include/lammps_shim like softParticleCloud.C:57-62 does: it is
    compiled against both MPI_Comm ABI families, tests/c_abi/fake_mpi_*)"""
    r = subprocess.run(["g++", "-std=c++98", "-x", "c++", "-fsyntax-only", "-Wall", "-Werror",
                        "-I", os.path.join(ROOT, "tests", "adapters", "openfoam_min"),
                        "-I", os.path.join(ROOT, "tests", "c_abi", mpi), "-I", os.path.join(ROOT, "include", "lammps_shim"),
                        "-I", os.path.join(ROOT, "include"),
                        "-I", os.path.join(ROOT, "adapters", "openfoam"), src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_enhanced_cloud_adapter_has_the_reference_class_surface():
    """enhancedCloud.H:183-249 + what lammpsFoam.C's include files call on the cloud (createParticles.H:7-20,
    moveParticles.H:3-4, liftDragCoeffs.H:16-18, pEqn.H:22, writeCPUTime.H:7-18)"""
    h = open(os.path.join(ROOT, "adapters", "openfoam", "enhancedCloudAmd.H")).read()
    flat = re.sub(r"\s+", " ", h)
    assert "class enhancedCloud" in flat
    ctor = ("enhancedCloud ( const volVectorField& U, const volScalarField& p, volVectorField& Ue, const volVectorField& Uf, "
            "const volVectorField& DDtUf, dimensionedScalar nu, volScalarField& alpha, IOdictionary& cloudDict, "
            "IOdictionary& transDict, scalar diffusionBandWidth, label diffusionSteps );")
    assert ctor in flat
    for member in ("void calcTcFields();", "void evolve();", "const volScalarField& Omega() const",
                   "const volVectorField& Asrc() const", "label particleCount() const",
                   "const scalarList& diffusionTimeCount() const", "const scalar& particleMoveTime() const",
                   "scalarList cpuTimeSplit()", "void dragInfo();", "void averageInfo();"):
        assert member in flat, member
    c = open(os.path.join(ROOT, "adapters", "openfoam", "enhancedCloudAmd.C")).read()
    # serial: the whole step in the library; -parallel: the sf_cloud_phase pieces around the collective lammps_step
    for call in ("sf_cloud_evolve(cloud_)", "sf_cloud_calc_tc_fields(cloud_)", "phase(0)", "phase(1)", "phase(2)", "phase(3)",
                 "phase(4)", "phase(5)", "phase(6)", "lammps_step(lmp_, subSteps_)", "new LAMMPS_NS::LAMMPS(0, NULL, commLammps)"):
        assert call in c, call


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
