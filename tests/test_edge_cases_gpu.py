"""Edge cases of the product path on the GPU: empty / tiny / ragged systems, capacity and slot growth,
input-script errors with the reference's wording, lost atoms, particle injection and removal."""
import numpy as np
import pytest

from sedifoam_amd import synthetic
from tests import dem_cases as dc
import tests.test_dem_gpu as T

pytestmark = pytest.mark.gpu


def test_single_particle_free_fall_and_wall_bounce():
    from sedifoam_amd import Lammps
    lmp = Lammps()
    lmp.set_box([0, 0, 0], [0.01, 0.01, 0.01])
    lmp.create_atoms([[0.005, 0.00051, 0.005]], [1e-3], [2650.0])
    lmp.commands("""
        atom_style sphere
        boundary f f f
        newton off
        neighbor 2.5e-4 bin
        pair_style gran/hertzFix/history 1e7 NULL 0.5 NULL 0.4 1
        pair_coeff * *
        timestep 1e-6
        fix 1 all nve/sphere
        fix 2 all gravity 9.81 vector 0 -1 0
        fix 3 all fdrag
        fix w all wall/granFix 1e7 NULL 0.5 NULL 0.4 1 yplane 0.0 0.01
    """)
    r = 0.5e-3; m = 4 / 3 * np.pi * r ** 3 * 2650.0
    orc = dc.ob.OracleDem([[0.005, 0.00051, 0.005]], [r], [m], [0, 0, 0], [0.01, 0.01, 0.01])
    orc.pair_gran("hertz", 1e7, None, 0.5, None, 0.4, 1); orc.fix_gravity(9.81, 0, -1, 0); orc.fix_fdrag(0.0)
    orc.fix_wall(1, 0.0, 0.01, 1e7, None, 0.5, None, 0.4, 1); orc.neighbor(2.5e-4); orc.timestep(1e-6)
    lmp.setup(); orc.setup()
    for _ in range(4):
        lmp.step(500); orc.run(500)
        a, b = lmp.get_state(), orc.get()
        assert np.max(np.abs(a["x"] - b["x"])) < 1e-14 and dc.rel_err(a["v"], b["v"]) < 1e-9
    assert lmp.info().nghost == 0 and lmp.get_local_n() == 1
    assert np.allclose(lmp.wall_shear(0), orc.wall_shear(0), atol=1e-18)


def test_empty_engine_steps_and_reports_nothing():
    from sedifoam_amd import Lammps
    lmp = Lammps()
    lmp.set_box([0, 0, 0], [1, 1, 1])
    lmp.commands("atom_style sphere\nneighbor 0.1 bin\npair_style gran/hooke/history 1e3 NULL 0 NULL 0.1 0\n"
                 "timestep 1e-5\nfix 1 all nve/sphere\nfix 3 all fdrag")
    lmp.step(10)
    assert lmp.get_local_n() == 0 and lmp.get_local_info()["x"].shape == (0, 3)


@pytest.mark.parametrize("n_side", [(3, 3, 3), (4, 3, 5)])     # 108 and 240 atoms: not multiples of 64
def test_ragged_sizes(n_side):
    bed = T._bed(n_side, periodic=True, seed=8)
    T._run_case(bed, T.BASE, steps=(1, 20))


def test_neighbor_slots_grow_on_overflow():
    # neigh_modify one 4: far fewer slots than the 12 neighbours of the lattice -> the engine must widen its
    # slot-major arrays at the first build (and keep the history correct afterwards)
    from sedifoam_amd import Lammps
    bed = T._bed((5, 5, 5), periodic=True, seed=17, vmax=0.4)
    cfg = dict(T.BASE, skin=0.05e-3, walls=T._walls(bed))
    lmp = Lammps()
    lmp.set_box(bed["boxlo"], bed["boxhi"])
    lmp.command("neigh_modify delay 0 one 4")
    lmp.create_atoms(bed["x"], bed["diameter"], bed["density"], v=bed["v"])
    for line in dc.script_lines(bed, cfg):
        if not line.startswith("neigh_modify"):
            lmp.command(line)
    orc = dc.make_oracle(bed, cfg)
    lmp.setup(); orc.setup()
    assert lmp.info().max_neigh_cap >= 12 and lmp.info().max_neigh_used >= 12
    lmp.step(80); orc.run(80)
    T._compare(lmp, orc)
    assert lmp.info().nbuilds >= 2


@pytest.mark.parametrize("ghost_free", [False, True])
def test_ghosts_outgrow_the_capacity_of_a_narrow_periodic_column(ghost_free, monkeypatch):
    """A tall column 3 lattice cells (4.2 d) wide, periodic in x and z: every grain has 1.5 periodic images on average
    (those of the x faces, then the z images of grains AND of x images), which is more than the slack the per-atom
    arrays are created with.  The ghost kernels keep their counts on the device and create nothing once the capacity
    would be exceeded; the host grows the arrays and repeats (DemEngine::make_periodic_ghosts).
    ghost_free: the same column through the list build that walks its stencil around the box instead of making ghost
    atoms (round 6, the default of a single-domain run): 3.37 cutoffs per periodic length, the narrowest box it takes --
    every atom's stencil reaches around the box in x AND z; the images LAMMPS would have made are still counted."""
    monkeypatch.setenv("SF_GHOST_FREE", "1" if ghost_free else "0")
    bed = T._bed((3, 600, 3), periodic=True, seed=3, vmax=0.1)
    assert bed["n"] == 21600
    cfg = dict(T.BASE, walls=T._walls(bed))
    lmp = dc.make_hip(bed, cfg)
    cap0 = lmp.info().capacity
    orc = dc.make_oracle(bed, cfg)
    lmp.setup(); orc.setup()
    info = lmp.info()
    assert info.nghost == orc.nghost and info.nlocal + info.nghost > cap0
    assert (info.capacity == cap0) if ghost_free else (info.capacity > cap0)
    T._compare(lmp, orc, tol_f=5e-12)
    lmp.step(5); orc.run(5)
    T._compare(lmp, orc, tol_f=5e-12)


def test_script_errors_use_reference_wording():
    from sedifoam_amd import Lammps, SfError
    lmp = Lammps()
    lmp.set_box([0, 0, 0], [1, 1, 1])
    lmp.command("boundary p f p")
    with pytest.raises(SfError, match="Illegal pair_style command"):
        lmp.command("pair_style gran/hertzFix/history 1e7 NULL 0.5 NULL 0.4")          # 5 args (needs 6)
    with pytest.raises(SfError, match="Illegal pair_style command"):
        lmp.command("pair_style gran/hertzFix/history -1 NULL 0.5 NULL 0.4 1")
    with pytest.raises(SfError, match="Illegal fix cohesive command"):
        lmp.command("fix c all cohesive 1e-20 1e-7 1e-9 1e-4")                           # 7 args (needs 8)
    with pytest.raises(SfError, match="invalid option for cohesive force model"):
        lmp.command("fix c all cohesive 1e-20 1e-7 1e-9 1e-4 2")
    with pytest.raises(SfError, match="Cannot use wall in periodic dimension"):
        lmp.command("fix w all wall/granFix 1e7 NULL 0.5 NULL 0.4 1 xplane 0 1")
    lmp.command("pair_style gran/hertzFix/history 1e7 NULL 0.5 NULL 0.4 1")
    with pytest.raises(SfError, match="incompatible with Pair style"):
        lmp.command("fix w all wall/gran 1e7 NULL 0.5 NULL 0.4 1 yplane 0 1")           # stock wall/gran + hertzFix
    with pytest.raises(SfError, match="Unknown fix style"):
        lmp.command("fix q all rigid/small molecule")
    with pytest.raises(SfError, match="Unknown command"):
        lmp.command("bond_style harmonic")
    with pytest.raises(SfError, match="no `fix ... fdrag`"):
        lmp.put_local_info(np.zeros((0, 3)), np.zeros(0, np.int32))
    # what the fused kernel cannot do the way LAMMPS would is refused, not silently done differently
    for bad in ("neigh_modify delay 5", "neigh_modify every 10 check no", "velocity all set 0.1 NULL 0.0"):
        with pytest.raises(SfError, match="supported"):
            lmp.command(bad)
    lmp.command("neigh_modify delay 0 every 1 check yes one 40")
    lmp.command("group bottom type 2")
    with pytest.raises(SfError, match="fix cohesive on a group"):
        lmp.command("fix c bottom cohesive 1e-20 1e-7 1e-9 1e-4 1")
    lmp.command("fix 4 bottom freeze")
    # force fixes may follow fix freeze (the reference's bed scripts put wall/gran after it; they then act on frozen
    # grains too); only fix cohesive, which the pair loop adds before every post_force fix, must come first
    lmp.command("fix 2 all gravity 9.8 vector 0 -1 0")
    with pytest.raises(SfError, match="after fix freeze"):
        lmp.command("fix c all cohesive 1e-20 1e-7 1e-9 1e-4 1")


def test_put_local_info_rejects_foreign_tag():
    from sedifoam_amd import SfError
    bed = T._bed((3, 3, 3), periodic=True, seed=2)
    lmp = dc.make_hip(bed, dict(T.BASE, walls=T._walls(bed)))
    lmp.setup()
    tag = lmp.get_local_info()["tag"].copy()
    tag[0] = 10 ** 6
    with pytest.raises(SfError, match="incoming tag not owned"):
        lmp.put_local_info(np.zeros((len(tag), 3)), tag)


def test_lost_atom_is_an_error_not_silent():
    from sedifoam_amd import Lammps, SfError
    lmp = Lammps()
    lmp.set_box([0, 0, 0], [0.01, 0.01, 0.01])
    lmp.create_atoms([[0.005, 0.005, 0.005]], [1e-3], [2650.0], v=[[50.0, 0, 0]])
    lmp.commands("atom_style sphere\nboundary f f f\nneighbor 2.5e-4 bin\n"
                 "pair_style gran/hertzFix/history 1e7 NULL 0.5 NULL 0.4 1\ntimestep 1e-5\nfix 1 all nve/sphere\nfix 3 all fdrag")
    with pytest.raises(SfError, match="Lost atoms"):
        for _ in range(10):
            lmp.step(100)


def test_create_and_delete_particles():
    # lammps_create_particle / lammps_delete_particle (library.cpp:406-621): counts, tags, mass with the
    # reference's pi literal, neighbours rebuilt, history of the survivors kept
    bed = T._bed((4, 4, 4), periodic=True, seed=6)
    cfg = dict(T.BASE, walls=T._walls(bed))
    lmp = dc.make_hip(bed, cfg)
    lmp.setup()
    lmp.step(10)
    n0 = lmp.get_local_n()
    h0 = lmp.history()
    top = float(bed["x"][:, 1].max())
    pos = np.array([[1.0e-3, top + 2.0e-3, 1.0e-3], [3.0e-3, top + 2.0e-3, 2.0e-3]])
    lmp.create_particle(pos, [n0 + 1, n0 + 2], 0.8e-3, 2000.0, 1, [0.0, -0.1, 0.0])
    assert lmp.get_local_n() == n0 + 2 and lmp.get_global_n() == n0 + 2
    ii = lmp.get_initial_info()
    k = int(np.nonzero(ii["tag"] == n0 + 1)[0][0])
    assert ii["diam"][k] == pytest.approx(0.8e-3) and ii["rho"][k] == pytest.approx(2000.0, rel=1e-12)
    assert np.allclose(ii["v"][k], [0.0, -0.1, 0.0])
    lmp.step(5)
    h1 = lmp.history()
    assert set(h0) <= set(h1) or len(set(h0) - set(h1)) < 5        # old contacts (and their history) survive
    dead = [1, 5, n0 + 2]
    lmp.delete_particle(dead)
    assert lmp.get_local_n() == n0 - 1
    tags = set(lmp.get_local_info()["tag"].tolist())
    assert not (tags & set(dead)) and (n0 + 1) in tags
    lmp.step(20)
    st = lmp.get_state()
    assert np.isfinite(st["x"]).all() and len(st["tag"]) == n0 - 1
    assert all(t not in dead for pair in lmp.history() for t in pair)


def test_created_particles_join_group_active():
    """library.cpp:452-455: lammps_create_particle puts new atoms into `all` AND group "active" -- in the reference's
    bed scripts nve/sphere, gravity and fdrag act on `active`, so an injected particle must be integrated and fall."""
    bed = T._bed((4, 4, 4), periodic=True, seed=6)
    bed["type"] = np.where(bed["x"][:, 1] < 0.9e-3, 2, 1).astype(np.int32)
    bed["v"][bed["type"] == 2] = 0.0
    cfg = dict(T.BASE, walls=T._walls(bed), frozen_types=[2])
    lmp = dc.make_hip(bed, cfg)
    lmp.setup()
    lmp.step(5)
    n0 = lmp.get_local_n()
    top = float(bed["x"][:, 1].max())
    pos = np.array([[1.0e-3, top + 3.0e-3, 1.0e-3]])
    lmp.create_particle(pos, [n0 + 1], 0.8e-3, 2000.0, 1, [0.0, 0.0, 0.0])
    lmp.step(100)
    st = lmp.get_state()
    k = int(np.nonzero(st["tag"] == n0 + 1)[0][0])
    # free fall over 100 sub-steps of 1 us: v = -g t (to the half-kick bookkeeping), y below the start
    assert st["v"][k, 1] == pytest.approx(-9.81 * 100e-6, rel=0.02) and st["x"][k, 1] < pos[0, 1]


def test_bed_script_with_groups_read_data_and_comments(tmp_path):
    """The command set of the reference's bed cases through the parser end to end: `read_data` of a sphere data file
    with two atom types, `boundary pp ff pp`, `group .. type`, `group .. subtract`, fixes on a group, trailing `#`
    comments, ignored thermo/dump lines.  Atoms outside the `active` group are neither integrated nor pulled by
    gravity; the others fall."""
    from sedifoam_amd import Lammps
    d = 5.0e-4
    rng = np.random.default_rng(4)
    pts = []
    for iy in range(4):
        for ix in range(6):
            for iz in range(6):
                pts.append((0.5 * d + ix * 1.05 * d, 0.5 * d + iy * 1.05 * d, 0.5 * d + iz * 1.05 * d))
    pts = np.array(pts)
    types = np.where(pts[:, 1] < d, 2, 1)
    data = tmp_path / "bed.in"
    with open(data, "w") as f:
        f.write(" sphere data\n\n   %d    atoms\n   2    atom types\n\n 0.0 %g  xlo xhi\n 0.0 %g  ylo yhi\n 0.0 %g  zlo zhi\n\nAtoms\n\n"
                % (len(pts), 6.3 * d, 8.0 * d, 6.3 * d))
        for k, (p, t) in enumerate(zip(pts, types)):
            f.write(" %d %d %g 2500 %.12g %.12g %.12g\n" % (7 * (k + 1), t, d, p[0], p[1], p[2]))
    lmp = Lammps()
    lmp.commands("""
        atom_style   sphere
        atom_modify  map array
        boundary     pp ff pp
        newton       off
        communicate single vel yes
        read_data    %s
        neighbor     1.0e-4 bin
        neigh_modify delay 0
        pair_style   gran/hooke/history 2000.0 NULL 50.0 NULL 0.4 0
        pair_coeff   * *
        timestep     1e-6
        group        bottom type 2
        group        active subtract all bottom
        velocity     all set 0.0 0.0 0.0 units box
        fix   1 active nve/sphere
        fix   2 active gravity 9.8 vector 0 -1 0   # spherical 90.0 -180.0
        fix   3 active fdrag
        fix   ywall all wall/gran 2000.0 NULL 50.0 NULL 0.4 0 yplane 0.0 0.004
        thermo_style one   # ignored
        thermo       2000
        thermo_modify lost error
        dump  id all custom 100 snapshot id type x y z
    """ % data)
    info0 = lmp.get_initial_info()
    assert sorted(info0["tag"]) == [7 * (k + 1) for k in range(len(pts))]
    x0 = lmp.get_state()
    lmp.step(200)
    x1 = lmp.get_state()
    t_sorted = types[np.argsort(7 * (np.arange(len(pts)) + 1))]
    bottom = t_sorted == 2
    assert np.array_equal(x1["x"][bottom], x0["x"][bottom]) and np.all(x1["v"][bottom] == 0.0)
    assert np.all(x1["v"][~bottom][:, 1] < 0.0)          # free fall / settling under gravity
    with pytest.raises(Exception, match="Could not find group ID"):
        lmp.command("fix 9 nosuchgroup nve/sphere")


@pytest.mark.parametrize("e", [0.5, 0.9])
def test_hertz_head_on_collision_restitution(e):
    """binary head-on collision through the HIP engine: coefficient of restitution = gamman (see the oracle twin in
    tests/test_oracle_known_answers.py)"""
    from sedifoam_amd import Lammps
    R, rho, u = 0.5e-3, 2500.0, 0.2
    lmp = Lammps()
    lmp.set_box([0, 0, 0], [4e-3, 4e-3, 4e-3])
    lmp.create_atoms([[1.0e-3, 2.0e-3, 2.0e-3], [2.2e-3, 2.0e-3, 2.0e-3]], [2 * R, 2 * R], [rho, rho],
                     v=[[u / 2, 0, 0], [-u / 2, 0, 0]])
    lmp.commands("""
        atom_style sphere
        boundary f f f
        newton off
        communicate single vel yes
        neighbor 0.3e-3 bin
        neigh_modify delay 0
        pair_style gran/hertzFix/history 1.0e7 NULL %.17g NULL 0.0 1
        pair_coeff * *
        timestep 2.0e-8
        fix 1 all nve/sphere
        fix 3 all fdrag
    """ % e)
    lmp.step(int((0.2e-3 / u + 4.0e-4) / 2.0e-8))
    st = lmp.get_state()
    assert st["x"][1, 0] - st["x"][0, 0] > 2 * R
    assert (st["v"][1, 0] - st["v"][0, 0]) / u == pytest.approx(e, abs=2e-3)
