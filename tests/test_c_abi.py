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


# ---- the LAMMPS OBJECT shim: how softParticleCloud.C really owns LAMMPS (new LAMMPS / input->one / delete) ----
SHIM_SRC = os.path.join(ROOT, "tests", "c_abi", "shim_driver.cpp")


def _build_shim(tmp_path, std, mpi_flavour):
    exe = str(tmp_path / ("shim_%s_%s" % (std.replace("+", "x"), mpi_flavour)))
    cmd = ["g++", "-std=" + std, "-O1", "-Wall", "-Werror", "-Wno-unused-variable",
           "-I", os.path.join(ROOT, "tests", "c_abi", "fake_mpi_" + mpi_flavour),
           "-I", os.path.join(ROOT, "include", "lammps_shim"), "-I", os.path.join(ROOT, "include"), SHIM_SRC, "-o", exe,
           "-L", LIBDIR, "-lsedifoam_amd", "-Wl,-rpath," + LIBDIR]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.parametrize("std,mpi_flavour", [("c++98", "int"), ("c++98", "ptr"), ("c++17", "int"), ("c++17", "ptr")])
def test_lammps_object_shim_compiles_for_both_mpi_abis(tmp_path, std, mpi_flavour):
    """include/lammps_shim/{lammps,input,atom,library}.h stand where LAMMPS' headers stand for
    lammpsFoam/include/LammpsCollection.H; MPI_Comm as an int (MPICH) and as a pointer (Open MPI); C++98 (OpenFOAM 2.3)
    and C++17."""
    if not os.path.exists(os.path.join(LIBDIR, "libsedifoam_amd.so")):
        pytest.skip("library not built")
    _build_shim(tmp_path, std, mpi_flavour)


def _build_shim_real_mpich(tmp_path):
    """the same driver against the image's real MPICH (mpi.h + libmpi of /opt/conda) instead of a stand-in header"""
    if not (os.path.exists("/opt/conda/include/mpi.h") and os.path.exists("/opt/conda/lib/libmpi.so")):
        pytest.skip("no MPICH in this image")
    exe = str(tmp_path / "shim_real_mpich")
    cmd = ["g++", "-std=c++98", "-O1", "-Wall", "-Werror", "-Wno-unused-variable", "-Wno-long-long",
           "-DSHIM_DRIVER_REAL_MPI", "-I", "/opt/conda/include",
           "-I", os.path.join(ROOT, "include", "lammps_shim"), "-I", os.path.join(ROOT, "include"), SHIM_SRC, "-o", exe,
           "-L", LIBDIR, "-lsedifoam_amd", "-Wl,-rpath," + LIBDIR, "/opt/conda/lib/libmpi.so",
           "-Wl,-rpath,/usr/lib/x86_64-linux-gnu", "-Wl,-rpath,/opt/conda/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_lammps_object_shim_compiles_against_a_real_mpi(tmp_path):
    """MPICH 3.3 of the image: MPI_Comm is its real int handle, mpi.h its real header (C++98, as OpenFOAM 2.3 builds)"""
    if not os.path.exists(os.path.join(LIBDIR, "libsedifoam_amd.so")):
        pytest.skip("library not built")
    _build_shim_real_mpich(tmp_path)


def _bed_script(tmp_path, nx=6, nz=6, processors=None, velocity=None):
    d = 5.0e-4
    pts = [(0.5 * d + ix * 1.02 * d, 0.5 * d + iy * 1.02 * d, 0.5 * d + iz * 1.02 * d)
           for iy in range(4) for ix in range(nx) for iz in range(nz)]
    data = tmp_path / "bed.in"
    with open(data, "w") as f:
        f.write(" sphere data\n\n %d atoms\n 1 atom types\n\n 0.0 %g xlo xhi\n 0.0 %g ylo yhi\n 0.0 %g zlo zhi\n\nAtoms\n\n"
                % (len(pts), 1.02 * nx * d, 8.0 * d, 1.02 * nz * d))
        for k, p in enumerate(pts):
            f.write(" %d 1 %g 2650 %.12g %.12g %.12g\n" % (k + 1, d, p[0], p[1], p[2]))
    script = tmp_path / "in.lammps"
    with open(script, "w") as f:
        f.write("# bed of the reference's kind, read line by line through lmp->input->one\n"
                "atom_style sphere\nboundary pp ff pp\nnewton off\ncommunicate single vel yes\n")
        if processors:
            f.write("processors %s\n" % processors)   # (expMueller09/in.lammps:7 `processors  2 1 1`)
        f.write("read_data %s\n\nneighbor 1.0e-4 bin\nneigh_modify delay 0\n"
                "pair_style gran/hertzFix/history 1.0e7 NULL 0.5 NULL 0.4 1\npair_coeff * *\ntimestep 1e-6\n" % data)
        if velocity:
            f.write("velocity all set %s\n" % velocity)
        f.write("fix 1 all nve/sphere\nfix 2 all gravity 9.8 vector 0 -1 0\nfix 3 all fdrag\n"
                "fix ywall all wall/granFix 1.0e7 NULL 0.5 NULL 0.4 1 yplane 0.0 0.004\nthermo 1000\n")
    return script, len(pts)


@pytest.mark.gpu
@pytest.mark.parametrize("mpi_flavour", ["int", "ptr"])
def test_lammps_object_shim_runs_the_call_sequence_of_soft_particle_cloud(tmp_path, mpi_flavour):
    script, n = _bed_script(tmp_path)
    exe = _build_shim(tmp_path, "c++98", mpi_flavour)
    r = subprocess.run([exe, str(script)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    tok = r.stdout.strip().split()
    assert tok[0] == "OK" and int(tok[1]) == n
    assert float(tok[3]) > float(tok[2])          # net upward force 2 m g - m g: the bed rises
    assert int(tok[4]) == n + 1 - 2               # one particle created, two deleted


@pytest.mark.gpu
def test_lammps_object_shim_runs_on_the_real_mpich(tmp_path):
    """the same call sequence with MPI_Init / MPI_Comm_dup / MPI_Finalize of a real MPI library (singleton start)"""
    script, n = _bed_script(tmp_path)
    exe = _build_shim_real_mpich(tmp_path)
    r = subprocess.run([exe, str(script)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    tok = r.stdout.strip().split()
    assert tok[0] == "OK" and int(tok[1]) == n and int(tok[4]) == n + 1 - 2


@pytest.mark.gpu
def test_lammps_object_shim_aborts_like_lammps_on_a_bad_script_line(tmp_path):
    """LAMMPS ends the run on an input error (error->all); the shim prints the engine's message and aborts"""
    script = tmp_path / "in.lammps"
    script.write_text("atom_style sphere\npair_style gran/hertzFix/history 1.0e7 NULL\n")
    exe = _build_shim(tmp_path, "c++98", "int")
    r = subprocess.run([exe, str(script)], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "pair_style" in (r.stdout + r.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("world,processors,nx,nz", [(2, None, 12, 12), (2, "2 1 1", 12, 12), (4, "2 1 2", 12, 12),
                                                    (4, "* 1 *", 12, 12), (3, "1 1 3", 6, 9), (4, "4 1 1", 6, 6)])
def test_lammps_object_shim_runs_in_parallel_from_the_reference_calls_alone(tmp_path, world, processors, nx, nz):
    """`mpirun -np N lammpsFoam -parallel` as the reference does it (softParticleCloud.C:60-62,106,119-201,893-914): N
    ranks each create `new LAMMPS(0, NULL, comm)` on the duplicated world communicator and feed it the SAME script
    lines; the engine decomposes itself from `processors px py pz` (or chooses the grid like LAMMPS), every rank keeps
    the atoms of its brick of the `read_data` file, lammps_get_global_n / _get_initial_np are the global / per-rank
    counts of library.cpp:94-131, lammps_step is collective, atoms migrate between the ranks (the bed is thrown
    sideways through the periodic faces: ~10 rebuilds in the 65 sub-steps), create / delete are collective.  The
    driver (tests/c_abi/shim_driver.cpp, the image's real MPICH) contains no sf_* call; its N-rank result must equal
    its 1-rank result.  The ranks share the box's one GPU, the wire is the stand-in (tests/c_abi/standin_rccl.cpp).
    `4 1 1` on the narrow bed: bricks thinner than twice the ghost cutoff (an atom is a ghost of BOTH x-neighbours:
    more send directions than the sub-step kernel writes itself -> the pack kernel stays in the loop)."""
    import shutil
    from tests.test_halo_gpu import _standin_rccl
    script, n = _bed_script(tmp_path, nx=nx, nz=nz, processors=processors, velocity="8.0 0.0 5.0")
    exe = _build_shim_real_mpich(tmp_path)
    mpirun = shutil.which("mpirun") or "/opt/conda/bin/mpirun"
    one = subprocess.run([exe, str(script)], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, SHIM_DUMP=str(tmp_path / "one")))
    assert one.returncode == 0 and one.stdout.strip().splitlines()[-1].startswith("OK"), one.stdout + one.stderr
    ref = one.stdout.strip().splitlines()[-1].split()
    env = dict(os.environ, SF_RCCL_LIB=_standin_rccl(tmp_path), SHIM_DUMP=str(tmp_path / "many"))
    r = subprocess.run([mpirun, "-np", str(world), exe, str(script)], capture_output=True, text=True, timeout=600,
                       env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    ok = [ln for ln in r.stdout.splitlines() if ln.startswith("OK")]
    assert len(ok) == 1, r.stdout[-3000:]
    tok = ok[0].split()
    assert int(tok[1]) == int(ref[1]) == n and int(tok[4]) == int(ref[4]) == n + 1 - 2 and int(tok[7]) == world
    assert int(tok[8]) >= 10, "no atom changed its rank: the run did not exercise the migration"
    assert float(tok[2]) == pytest.approx(float(ref[2]), rel=1e-12)
    for k in (3, 5, 6):        # mean height, tag-weighted checksums of positions and velocities
        assert float(tok[k]) == pytest.approx(float(ref[k]), rel=1e-9), (k, tok, ref)
    # ... and atom by atom (before the create / delete calls): every rank dumped tag, wrapped position, velocity
    import numpy as np
    a = np.loadtxt(str(tmp_path / "one.0"))
    b = np.concatenate([np.loadtxt(str(tmp_path / ("many.%d" % q)), ndmin=2) for q in range(world)])
    assert len(a) == len(b) == n and np.array_equal(np.sort(a[:, 0]), np.sort(b[:, 0]))
    a, b = a[np.argsort(a[:, 0])], b[np.argsort(b[:, 0])]
    scale_x, scale_v = np.max(np.abs(a[:, 1:4])), np.max(np.abs(a[:, 4:7]))
    Lx, Lz = (float(t) for t in open(str(tmp_path / "one.0")).readline().split()[1:3])
    dx = np.abs(a[:, 1:4] - b[:, 1:4])
    dx[:, 0] = np.minimum(dx[:, 0], np.abs(dx[:, 0] - Lx))   # (an atom on the periodic face may be wrapped either way)
    dx[:, 2] = np.minimum(dx[:, 2], np.abs(dx[:, 2] - Lz))
    assert np.max(dx) <= 1e-9 * scale_x, np.max(dx)
    assert np.max(np.abs(a[:, 4:7] - b[:, 4:7])) <= 1e-9 * scale_v
