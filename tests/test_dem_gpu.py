"""GPU parity of the DEM hot path: HIP engine (through the lammps_* C-ABI) vs the CPU oracle on the same
seeded inputs.  FP64 tolerances: forces / torques / shear after one evaluation
rel <= 1e-12, positions / velocities after 50 sub-steps rel <= 1e-9 (summation order and FMA contraction
differ; nothing else)."""
import numpy as np
import pytest

from sedifoam_amd import synthetic
from tests import dem_cases as dc

pytestmark = pytest.mark.gpu

BASE = dict(pair="hertz", kn=1.0e7, gamman=0.5, xmu=0.4, g=9.81, dt=1.0e-6, skin=0.25e-3)


def _bed(ncells, periodic, seed=12345, **kw):
    bed = synthetic.fcc_bed(ncells, seed=seed, **kw)
    if not periodic:
        # closed box: walls on every side, lattice shifted one radius inside
        bed["periodic"] = (0, 0, 0)
        bed["x"][:, 0] += 0.3e-3
        bed["x"][:, 2] += 0.3e-3
        bed["boxhi"][0] += 0.6e-3
        bed["boxhi"][2] += 0.6e-3
    return bed


def _walls(bed):
    w = [(1, float(bed["boxlo"][1]), float(bed["boxhi"][1]))]
    if not bed["periodic"][0]:
        w.append((0, float(bed["boxlo"][0]), float(bed["boxhi"][0])))
    if not bed["periodic"][2]:
        w.append((2, float(bed["boxlo"][2]), float(bed["boxhi"][2])))
    return w


def _compare(lmp, orc, tol_f=1e-12, tol_x=1e-9, check_force=True):
    a = lmp.get_state()
    b = orc.get()
    assert (a["tag"] == b["tag"]).all()
    if check_force:
        assert dc.rel_err(a["f"], b["f"]) <= tol_f
        assert dc.rel_err(a["torque"], b["torque"]) <= tol_f
    # positions: relative to the particle diameter scale, velocities to max |v|
    assert np.max(np.abs(a["x"] - b["x"])) <= tol_x * 1e-3
    assert dc.rel_err(a["v"], b["v"]) <= tol_x
    assert dc.rel_err(a["omega"], b["omega"]) <= max(tol_x, 1e-9) or np.max(np.abs(b["omega"])) == 0.0
    ha, hb = lmp.history(), orc.history()
    assert set(ha) == set(hb)
    if ha:
        sa = np.array([ha[k] for k in sorted(ha)])
        sb = np.array([hb[k] for k in sorted(hb)])
        assert dc.rel_err(sa, sb) <= max(tol_x, tol_f)


def _run_case(bed, cfg, steps=(1, 49), fdrag_seed=7, tol_f=1e-12, walls=None):
    cfg = dict(cfg)
    cfg["walls"] = _walls(bed) if walls is None else walls
    lmp = dc.make_hip(bed, cfg)
    orc = dc.make_oracle(bed, cfg)
    lmp.setup()
    orc.setup()
    assert lmp.get_local_n() == orc.nlocal
    _compare(lmp, orc, tol_f=tol_f)           # setup forces (shearupdate = 0)
    rng = np.random.default_rng(fdrag_seed)
    st = orc.get()
    fd = rng.normal(scale=1e-6, size=st["x"].shape)
    perm = rng.permutation(len(fd))      # rows in arbitrary order: matched by tag (library.cpp:344-366)
    lmp.put_local_info(fd[perm], st["tag"][perm])
    orc.put_fdrag(fd[perm], st["tag"][perm])
    for n in steps:
        lmp.step(n)
        orc.run(n)
        _compare(lmp, orc, tol_f=tol_f)
    return lmp, orc


@pytest.mark.parametrize("motion", [{"wiggle": (1, 0.05e-3, 2.0e-4)}, {"wiggle": (0, 0.05e-3, 2.0e-4)},
                                    {"shear": (0, 0.5)}])
def test_closed_box_with_moving_floor(motion):
    """fix wall/granFix ... yplane lo hi wiggle|shear (fix_wall_granFix.cpp:117-141, :255-264): the floor and lid
    oscillate along their normal (positions and velocity follow the step count) or along x (velocity only), or slide
    along x; 120 sub-steps = 0.6 of a wiggle period."""
    bed = _bed((5, 5, 5), periodic=False, seed=31)
    walls = _walls(bed)
    walls[0] = walls[0] + (motion,)
    lmp, orc = _run_case(bed, BASE, steps=(1, 60, 59), walls=walls, tol_f=5e-12)   # (torque sums cancel to ~1e-3 of their terms)
    wa, wb = lmp.wall_shear(0), orc.wall_shear(0)
    assert (np.abs(wb).sum(axis=1) > 0).sum() >= 10 and dc.rel_err(wa, wb) <= 1e-9
    # the second run continues the wall's clock (time_origin is set once, at setup)
    a, b = lmp.get_state(), orc.get()
    assert dc.rel_err(a["v"], b["v"]) <= 1e-9


@pytest.mark.parametrize("rotate", [False, True])
def test_bed_in_a_z_cylinder(rotate):
    """fix wall/granFix ... zcylinder R [shear x v] (fix_wall_granFix.cpp:107-112, :309-322): radial contact with the
    cylinder about the origin; sheared about x or y the wall rotates."""
    bed = synthetic.fcc_bed((6, 5, 6), seed=41)
    d = float(bed["diameter"][0])
    x = bed["x"].copy()
    ctr = 0.5 * (x.min(axis=0) + x.max(axis=0))
    x[:, 0] -= ctr[0]
    x[:, 1] -= ctr[1]
    rr = np.hypot(x[:, 0], x[:, 1])
    keep = rr < 2.1e-3
    R = float(rr[keep].max() + 0.495 * d)                  # the outermost grains press up to 0.5 % of d into the wall
    n = int(keep.sum())
    assert n > 100 and (rr[keep] > R - 0.5 * d).sum() >= 4
    for k in ("diameter", "density", "v"):
        bed[k] = np.ascontiguousarray(np.asarray(bed[k])[keep])
    bed["x"] = np.ascontiguousarray(x[keep])
    bed["n"] = n
    bed["periodic"] = (0, 0, 0)
    zlo, zhi = float(bed["x"][:, 2].min() - 0.495 * d), float(bed["x"][:, 2].max() + 0.495 * d)
    bed["boxlo"] = np.array([-R - d, -R - d, zlo]); bed["boxhi"] = np.array([R + d, R + d, zhi])
    extra = {"cyl": R}
    if rotate:
        extra["shear"] = (0, 0.3)
    walls = [(3, None, None, extra), (2, zlo, zhi)]
    lmp, orc = _run_case(bed, BASE, steps=(1, 40), walls=walls, tol_f=5e-12)
    f = lmp.get_state()["f"]
    assert np.isfinite(f).all() and np.abs(f).max() > 0.0
    # the cylinder is really touched, and its history (tangential displacement) agrees
    wa, wb = lmp.wall_shear(0), orc.wall_shear(0)
    assert (np.abs(wb).sum(axis=1) > 0).sum() >= 4
    assert dc.rel_err(wa, wb) <= 1e-9


def test_closed_box_hertz():
    bed = _bed((5, 5, 5), periodic=False)
    lmp, orc = _run_case(bed, BASE)
    assert lmp.info().nghost == 0


def test_periodic_hertz_bed():
    bed = _bed((8, 6, 8), periodic=True)
    lmp, orc = _run_case(bed, BASE)
    info = lmp.info()
    assert info.nghost > 0 and info.nghost == orc.nghost
    # full list = 2 x the oracle's half list (owned-owned pairs twice, owned-ghost pairs once each side)
    assert info.npairs_full == 2 * (orc.npairs - 0) - 0 or info.npairs_full > orc.npairs


@pytest.mark.parametrize("ghost_free", ["0", "1"])
@pytest.mark.parametrize("box", ["xz", "xyz"])
def test_periodic_images_as_ghost_atoms_and_around_the_box(ghost_free, box, monkeypatch):
    """The two ways a single-domain rebuild treats periodic images -- ghost atoms per dimension as LAMMPS makes them
    (SF_GHOST_FREE=0) and the list build that walks its cell stencil around the box (the default since round 6) -- on a
    bed periodic in x and z (walls in y) and on a fully periodic one, hot enough to rebuild several times: the same
    pairs, forces, histories as the oracle, the same number of rebuilds, the same count of images."""
    monkeypatch.setenv("SF_GHOST_FREE", ghost_free)
    bed = _bed((7, 6, 8), periodic=True, seed=19, vmax=0.5)
    cfg = dict(BASE, skin=0.05e-3)
    walls = None
    if box == "xyz":
        bed["boxhi"][1] = 6 * bed["edge"]
        bed["x"][:, 1] %= bed["boxhi"][1]
        bed["periodic"] = (1, 1, 1)
        cfg["g"] = 0.0
        cfg["skin"] = 0.02e-3   # (no walls to run into: the grains only move skin / 2 often enough against a thinner skin)
        walls = []
    lmp, orc = _run_case(bed, cfg, steps=(1, 150), walls=walls, tol_f=5e-12)   # (g = 0: the forces are a few mN, measured 2e-12)
    info = lmp.info()
    assert info.nbuilds >= 3 and info.nbuilds == orc.nbuilds
    assert info.nghost > 0 and info.nghost == orc.nghost


def test_periodic_hooke_bed():
    cfg = dict(BASE, pair="hooke", kn=2.0e3, gamman=50.0, dampflag=1)
    bed = _bed((6, 5, 6), periodic=True, seed=4)
    _run_case(bed, cfg)


def test_periodic_plain_hooke_bed_with_wall_and_rebuilds():
    """`pair_style gran/hooke` [3P] + the wall's HOOKE branch (fix_wall_granFix.cpp:219-220, :333-335, :347-437): no
    shear history anywhere; fast grains and a thin skin so that lists are rebuilt while contacts are open."""
    cfg = dict(BASE, pair="hooke_plain", kn=2.0e3, gamman=50.0, dampflag=1, skin=0.05e-3)
    bed = _bed((6, 5, 6), periodic=True, seed=41, vmax=0.4)
    lines = dc.script_lines(bed, dict(cfg, walls=_walls(bed)))
    assert any(l.startswith("pair_style gran/hooke ") for l in lines)
    # (torques here are sums of velocity-damping forces alone, ~1e-8 N m out of contact forces 1e3 x larger: after 100
    # sub-steps of 0.4 m/s grains the worst component sits at 1e-11 of the largest)
    lmp, orc = _run_case(bed, cfg, steps=(40, 60), tol_f=5e-11)
    assert lmp.info().nbuilds >= 2 and orc.nbuilds == lmp.info().nbuilds
    hist = lmp.history()
    assert len(hist) > 0 and all(np.all(s == 0.0) for s in hist.values())   # contacts counted, history slots zero


def test_rebuild_with_history_carry_over():
    # fast particles + thin skin: several neighbour rebuilds inside the run, shear history must survive
    bed = _bed((6, 6, 6), periodic=True, seed=99, vmax=0.5)
    cfg = dict(BASE, skin=0.05e-3)
    lmp, orc = _run_case(bed, cfg, steps=(60, 60))
    assert lmp.info().nbuilds >= 3 and orc.nbuilds >= 3


def test_mid_size_100k_bed_with_rebuilds():
    """BASELINE config C3's size (100 k grains), the DEM leg; the coupled step at this size is
    tests/test_cloud_gpu.py::test_config_c3_100k_coupled_50_substeps.  The launch shapes of the large-N path (XCD remap,
    several rebuilds with history carry-over) against the oracle, which still finishes in seconds here."""
    bed = _bed((29, 29, 30), periodic=True, seed=23, vmax=0.5)
    assert bed["n"] >= 100000
    # net force = 12 contact forces ~1e3 x larger that nearly cancel: the worst of 3e5 components sits at 2e-12 of
    # max |f| (1e-12 holds for the 1e3-particle cases)
    lmp, orc = _run_case(bed, dict(BASE, skin=0.05e-3), steps=(1, 110), tol_f=5e-12)
    assert lmp.info().nbuilds >= 3 and orc.nbuilds == lmp.info().nbuilds


@pytest.mark.parametrize("pair,fdrag_group", [("hertz", "all"), ("hooke", "active")])
def test_frozen_bottom_layer_groups(pair, fdrag_group):
    """The bed set-up of the reference's example cases (cases/example-cases/*/in.lammps): `group bottom type 2`,
    `group active subtract all bottom`, nve/sphere + gravity (+ fdrag) on `active`, `fix 4 bottom freeze`.  Frozen
    atoms do not move, feel no force, and count as infinitely heavy in the contact law
    (pair_gran_hertzFix_history.cpp:188-189)."""
    bed = _bed((6, 5, 6), periodic=True, seed=29, vmax=0.3)
    bed["type"] = np.where(bed["x"][:, 1] < 0.9e-3, 2, 1).astype(np.int32)     # the two bottom lattice layers
    assert 0 < (bed["type"] == 2).sum() < bed["n"] // 2
    bed["v"][bed["type"] == 2] = 0.0
    cfg = dict(BASE, pair=pair, skin=0.08e-3, frozen_types=[2], fdrag_group=fdrag_group)
    if pair == "hooke":
        cfg.update(kn=2.0e3, gamman=50.0)
    lmp, orc = _run_case(bed, cfg, steps=(1, 80), tol_f=5e-12)
    a = lmp.get_state()
    bottom = bed["type"][a["tag"] - 1] == 2
    assert np.array_equal(a["x"][bottom], bed["x"][a["tag"] - 1][bottom])      # frozen atoms never move
    assert np.all(a["f"][bottom] == 0.0) and np.all(a["torque"][bottom] == 0.0) and np.all(a["v"][bottom] == 0.0)
    assert np.abs(a["f"][~bottom]).max() > 0.0
    assert lmp.info().nbuilds >= 2


def test_reference_bed_script_order_wall_after_freeze():
    """cases/example-cases/transport-bedload/in.lammps:25-31 verbatim in their own order: `fix 1 all nve/sphere`,
    `fix 2 all gravity`, `fix 3 all fdrag`, `fix 4 bottom freeze`, THEN `fix ywall all wall/gran ... yplane`.  [3P]
    Modify::post_force runs the fixes in script order, so the wall still pushes frozen grains (and nve/sphere, on
    `all`, moves them): the bottom layer overlaps the wall here and must lift off exactly as in the oracle."""
    bed = _bed((6, 5, 6), periodic=True, seed=31, vmax=0.2)
    bed["type"] = np.where(bed["x"][:, 1] < 1.5e-3, 2, 1).astype(np.int32)        # the two bottom lattice layers
    bed["v"][bed["type"] == 2] = 0.0
    lowest = bed["x"][:, 1].min()
    walls = [(1, float(lowest - 0.45e-3), float(bed["boxhi"][1]))]     # lowest grains overlap the wall by ~0.05 mm
    cfg = dict(BASE, pair="hooke", kn=2.0e3, gamman=50.0, skin=0.08e-3, frozen_types=[2], freeze_first=True,
               nve_all=True, walls=walls)
    lines = dc.script_lines(bed, cfg)
    k = [i for i, l in enumerate(lines) if l.startswith("fix")]
    assert [lines[i].split()[3] for i in k] == ["nve/sphere", "gravity", "fdrag", "freeze", "wall/gran"]
    lmp, orc = _run_case(bed, cfg, steps=(1, 60), tol_f=5e-12, walls=walls)
    a = lmp.get_state()
    bottom = bed["type"][a["tag"] - 1] == 2
    touching = bottom & (bed["x"][a["tag"] - 1][:, 1] < lowest + 1e-9 + 0.02e-3)
    assert touching.any()
    # frozen grains in contact with the wall: only the wall force is left, and it has moved them (up, off the wall)
    assert np.all(a["x"][touching][:, 1] > bed["x"][a["tag"] - 1][touching][:, 1])
    # frozen grains that do not reach the wall: no force at all
    free_frozen = bottom & ~touching & (bed["x"][a["tag"] - 1][:, 1] > lowest + 0.3e-3)
    assert free_frozen.any() and np.all(a["f"][free_frozen] == 0.0)


def test_random_dilute_gas_collisions():
    """Not a lattice: 1500 spheres at random non-overlapping positions (volume fraction ~0.2) with 1 m/s random
    velocities in a periodic box with y walls -- neighbour counts from 0 to ~8, contacts that open and close
    (touch flag set / cleared, history created and dropped), a rebuild every ~25 sub-steps."""
    rng = np.random.default_rng(77)
    d = 1.0e-3
    L = np.array([18e-3, 18e-3, 18e-3])
    pts = []
    cell = {}
    while len(pts) < 1500:
        p = rng.uniform(0.6 * d, L - 0.6 * d)
        key = tuple((p // d).astype(int))
        ok = True
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dz in (-1, 0, 1):
                    for q in cell.get((key[0] + dx, key[1] + dy, key[2] + dz), ()):
                        if np.sum((pts[q] - p) ** 2) < (1.02 * d) ** 2:
                            ok = False
        if ok:
            cell.setdefault(key, []).append(len(pts))
            pts.append(p)
    x = np.array(pts)
    n = len(x)
    bed = dict(x=x, v=rng.uniform(-1.0, 1.0, size=(n, 3)), diameter=np.full(n, d), density=np.full(n, 2650.0),
               boxlo=np.zeros(3), boxhi=L.copy(), periodic=(1, 0, 1), n=n)
    lmp, orc = _run_case(bed, dict(BASE, skin=0.1e-3, g=0.0), steps=(1, 149), tol_f=1e-11)
    assert lmp.info().nbuilds >= 4 and orc.nbuilds == lmp.info().nbuilds
    h = lmp.history()
    assert 0 < len(h) < n       # some contacts are open at the end, far fewer than in a packed bed


def test_carrier_rho_added_mass_term():
    bed = _bed((4, 4, 4), periodic=True, seed=5)
    _run_case(bed, dict(BASE, carrier_rho=1000.0), steps=(1, 20))


def test_polydisperse_cohesive_lubricate():
    # d in [0.85, 1.0] mm on a lattice of spacing 0.95 mm: about half of the neighbours touch.
    # cut_inner >= the largest contact distance (as LAMMPS requires), lubrication acts in (1.001, 1.1] mm
    bed = _bed((5, 5, 5), periodic=True, seed=11, poly=(0.85e-3, 1.0e-3), spacing=0.95)
    cfg = dict(BASE, skin=0.2e-3, cohesive=(1.0e-13, 1.0e-7, 1.0e-7, 1.0e-4, 1),
               lub=(1.0e-3, 1, 1, 1.001e-3, 1.1e-3, 1, 1))
    _run_case(bed, cfg, steps=(1, 30))


def test_mid_size_polydisperse_cohesive_lubricate_32k():
    """BASELINE config C5's physics (polydisperse grains, fix cohesive + pair lubricate/poly overlay) at 32 k grains,
    through a rebuild"""
    bed = _bed((20, 20, 20), periodic=True, seed=13, poly=(0.85e-3, 1.0e-3), spacing=0.95, vmax=0.5)
    cfg = dict(BASE, skin=0.06e-3, cohesive=(1.0e-13, 1.0e-7, 1.0e-7, 1.0e-4, 1),
               lub=(1.0e-3, 1, 1, 1.001e-3, 1.1e-3, 1, 1))
    lmp, orc = _run_case(bed, cfg, steps=(1, 70), tol_f=5e-12)
    assert lmp.info().nbuilds >= 2 and orc.nbuilds == lmp.info().nbuilds


# SURVEY.md 8(d), config C5 as written: d ~ U(0.5, 1.5) mm (size ratio 3), fix cohesive ah = 1e-20 lam = 1e-7 smin = 1e-9
# smax = 0.1 d opt = 1, lubricate/poly mu = 1e-3 flaglog = 1 flagfld = 0, inner / outer cutoff 1.001 / 1.1 -- LAMMPS takes the
# two cutoffs as lengths: 1.001 / 1.1 of the LARGEST pair (d_max), so that every overlapping pair is inside the inner cutoff
# (beyond it the reference takes log(h_sep) of a negative gap, pair_lubricate_poly.cpp:286-300).  flagHI, flagVF: defaults 1 1
C5_WIDE = dict(cohesive=(1.0e-20, 1.0e-7, 1.0e-9, 1.0e-4, 1), lub=(1.0e-3, 1, 0, 1.001 * 1.5e-3, 1.1 * 1.5e-3, 1, 1))
# (a Hamaker constant the forces can see, with the floor of the narrow C5 case: at smin = 1e-9 the force of a touching pair is
# ~ 1 / del^2 of a gap del = r - (ri + rj) that cancels to 1e-6 of r -- conditioned 1e-10, not a 1e-11 comparison)
C5_WIDE_STRONG = dict(C5_WIDE, cohesive=(1.0e-13, 1.0e-7, 1.0e-7, 1.0e-4, 1))


@pytest.mark.parametrize("params", [C5_WIDE, C5_WIDE_STRONG], ids=["survey_8d", "strong_cohesion"])
def test_c5_wide_polydisperse_32k_through_a_rebuild(params):
    """Config C5 on the size distribution SURVEY.md 8(d) fixes: 32 k grains d ~ U(0.5, 1.5) mm in a dense disordered
    periodic bed (grown, synthetic.grown_poly_bed), Hertz history + fix cohesive + lubricate/poly with beta0 = rj / ri
    anywhere in [1/3, 3] (pair_lubricate_poly.cpp:299-324, fix_cohesive.cpp:236-245), cells of 2 r_max + skin holding up to
    27 x more small grains than large ones, rows of very different length in one wave: setup + 1 + 70 sub-steps through a
    rebuild, HIP vs oracle."""
    bed = synthetic.grown_poly_bed(32000, seed=17, vmax=0.5)
    ratio = bed["diameter"].max() / bed["diameter"].min()
    assert ratio > 2.9
    cfg = dict(BASE, skin=0.06e-3, g=0.0, **params)
    lmp, orc = _run_case(bed, cfg, steps=(1, 70), tol_f=2e-11, walls=[])   # (measured 5.2e-12: the log series at beta0 = 3)
    info = lmp.info()
    assert info.nbuilds >= 2 and orc.nbuilds == info.nbuilds
    assert info.npairs_full / info.nlocal > 14      # (a full list twice as long as the narrow C5 bed's)


def test_cohesive_opt0():
    bed = _bed((4, 4, 4), periodic=True, seed=12)
    cfg = dict(BASE, cohesive=(1.0e-13, 1.0e-7, 1.0e-7, 1.0e-4, 0))
    _run_case(bed, cfg, steps=(1, 20))


def test_deterministic_rerun():
    bed = _bed((6, 5, 6), periodic=True, seed=3)
    cfg = dict(BASE, walls=_walls(bed))
    outs = []
    for _ in range(2):
        lmp = dc.make_hip(bed, cfg)
        lmp.setup()
        lmp.step(25)
        outs.append(lmp.get_state())
    for k in ("x", "v", "omega", "f", "torque"):
        assert np.array_equal(outs[0][k], outs[1][k]), k


@pytest.mark.parametrize("env", [{"SF_LDS": "1", "SF_TILE": "4", "SF_SUB": "1"},
                                 {"SF_TILE": "0", "SF_XCD_REMAP": "0", "SF_SUB": "1"},
                                 {"SF_TILE": "8", "SF_SUB": "2"}, {"SF_TILE": "4", "SF_SUB": "3"},
                                 {"SF_LPA": "1", "SF_NT_POLICY": "0"}, {"SF_LPA": "1", "SF_NT_POLICY": "1"},
                                 {"SF_LPA": "1", "SF_NT_POLICY": "2", "SF_HIST_COPIES": "2"},
                                 {"SF_LPA": "1", "SF_NT_POLICY": "3"}, {"SF_LPA": "2"}, {"SF_LPA": "4"},
                                 {"SF_SCAN_MIN": "1"}, {"SF_ROCPRIM_SCAN": "1"},
                                 {"SF_HIST_IN_PLACE": "0"}, {"SF_HIST_IN_PLACE": "0", "SF_BUILD_LDS": "0"},
                                 {"SF_BUILD_QUAD": "0"}, {"SF_RANK_PERMUTE": "0"},
                                 {"SF_PARK_MARGIN": "-6"}, {"SF_PARK_MARGIN": "-6", "SF_BUILD_QUAD": "0"}])
def test_kernel_variants_agree_with_oracle(env, monkeypatch):
    """The LDS-staged tile kernel (k_substep_lds), the plain / tiled orderings of the gathering kernel, its cache
    policies (non-temporal rows or not, chosen by system size in production), the lanes per atom and the scan of the
    cell histograms (the engine's tile scan, used above 64 k cells in production, forced on for these small beds; or
    rocPRIM's; one workgroup below that in production), and the list build's three forms (old list read in place + parked
    candidates in LDS -- production on a single domain --, staged partner tags + LDS, staged + parked in memory) are speed
    options only: every one must reproduce the oracle, through rebuilds too.  SF_PARK_MARGIN=-6 gives the list build six
    LDS parking rows fewer than the previous list's longest row: every rebuild overflows them and is built again with as many
    rows as list slots (F_PARK_OVER)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    bed = _bed((6, 6, 6), periodic=True, seed=99, vmax=0.5)
    lmp, orc = _run_case(bed, dict(BASE, skin=0.05e-3), steps=(60, 60))
    assert lmp.info().nbuilds >= 3
    bed = _bed((5, 5, 5), periodic=True, seed=11, poly=(0.85e-3, 1.0e-3), spacing=0.95)
    _run_case(bed, dict(BASE, skin=0.2e-3, cohesive=(1.0e-13, 1.0e-7, 1.0e-7, 1.0e-4, 1),
                        lub=(1.0e-3, 1, 1, 1.001e-3, 1.1e-3, 1, 1)), steps=(1, 30))


def test_list_build_with_four_lanes_per_atom_gives_the_same_bits(monkeypatch):
    """k_build_neigh_quad (four or eight lanes per atom on as many consecutive records of a row, single domain without a ghost pass)
    builds the list k_build_neigh builds, word for word: a hot loose periodic bed (ghost-free build, touching neighbours
    first), a closed packed one and a polydisperse periodic one with both extra arms end in the same bits after several
    rebuilds."""
    cases = ((_bed((7, 6, 8), periodic=True, seed=23, vmax=0.5, jitter=0.3, spacing=1.1), dict(BASE, skin=0.03e-3), 150),
             (_bed((5, 5, 5), periodic=False, seed=5, vmax=0.5), dict(BASE, skin=0.03e-3), 120),
             (_bed((5, 5, 5), periodic=True, seed=11, poly=(0.85e-3, 1.0e-3), spacing=0.95, vmax=0.3),
              dict(BASE, skin=0.03e-3, cohesive=(1.0e-13, 1.0e-7, 1.0e-7, 1.0e-4, 1), lub=(1.0e-3, 1, 1, 1.001e-3, 1.1e-3, 1, 1)), 90))
    for bed, cfg, steps in cases:
        outs = []
        for q in ("0", "2", "4", "8"):
            monkeypatch.setenv("SF_BUILD_QUAD", q)
            lmp = dc.make_hip(bed, dict(cfg, walls=_walls(bed)))
            lmp.setup()
            lmp.step(steps)
            st = lmp.get_state()
            st["hist"] = lmp.history()
            st["nbuilds"] = lmp.info().nbuilds
            outs.append(st)
        assert outs[0]["nbuilds"] >= 2
        for o in outs[1:]:
            assert o["nbuilds"] == outs[0]["nbuilds"]
            for k in ("x", "v", "omega", "f", "torque"):
                assert np.array_equal(outs[0][k], o[k]), k
            assert set(o["hist"]) == set(outs[0]["hist"])
            assert all(np.array_equal(o["hist"][k], outs[0]["hist"][k]) for k in o["hist"])


def test_queueing_up_to_the_predicted_rebuild_changes_nothing(monkeypatch):
    """Sub-steps are queued up to where the next rebuild is expected instead of to the end of the run
    (RebuildPredictor): which sub-step triggers is still decided on the device, so a hot bed that rebuilds every few
    sub-steps must end in bitwise the same state with the prediction on and off, after the same number of rebuilds."""
    bed = _bed((6, 6, 6), periodic=True, seed=5, vmax=0.8)
    cfg = dict(BASE, skin=0.04e-3, walls=_walls(bed))
    outs = []
    for on in ("1", "0"):
        monkeypatch.setenv("SF_QUEUE_PREDICT", on)
        lmp = dc.make_hip(bed, cfg)
        lmp.setup()
        for n in (60, 35, 60):
            lmp.step(n)
        outs.append((lmp.get_state(), lmp.info().nbuilds))
    assert outs[0][1] == outs[1][1] and outs[0][1] >= 6
    for k in ("x", "v", "omega", "f", "torque"):
        assert np.array_equal(outs[0][0][k], outs[1][0][k]), k


@pytest.mark.parametrize("arms", [False, True])
def test_half_wave_gather_against_the_plain_gather(arms, monkeypatch):
    """The half-wave gather (sf_dem_kernels.h, sf_coop_variant: lanes l and l + 32 read a neighbour's record together,
    v_permlane32_swap puts the halves back) is chosen for beds beyond the memory-side cache (non-temporal policy 1 / 2);
    it moves the same bytes into the same operations in the same order.  A small bed forced through it -- closed box
    (neighbour counts from 3 to 12 inside one wave, so lanes wait at the wave's largest count), an atom count that
    leaves the last wave partly empty, rebuilds, and the kernel with both the cohesive and the lubrication arm -- must
    give the bits of the plain gather (policy 0) after the first sub-step, and after 105 sub-steps and two or three rebuilds
    differ from it by last bits only: the two are different instantiations of the kernel template (measured: one pair
    in two thousand gets a different last bit of its history per sub-step, tests/micro/debug_coop_bits2.py) -- 1e-12
    here against the 1e-9 of the parity tests."""
    monkeypatch.setenv("SF_LPA", "1")
    monkeypatch.setenv("SF_TOUCH_PREFETCH", "0")
    if arms:
        bed = _bed((5, 5, 5), periodic=True, seed=11, poly=(0.85e-3, 1.0e-3), spacing=0.95, vmax=0.3)
        cfg = dict(BASE, skin=0.08e-3, cohesive=(1.0e-13, 1.0e-7, 1.0e-7, 1.0e-4, 1),
                   lub=(1.0e-3, 1, 1, 1.001e-3, 1.1e-3, 1, 1), walls=_walls(bed))
    else:
        bed = _bed((7, 5, 6), periodic=False, seed=21, vmax=0.6)
        cfg = dict(BASE, skin=0.05e-3, walls=_walls(bed))
    assert len(bed["x"]) % 64 != 0
    first, last = [], []
    for policy in ("0", "2"):
        monkeypatch.setenv("SF_NT_POLICY", policy)
        lmp = dc.make_hip(bed, cfg)
        lmp.setup()
        lmp.step(1)
        first.append(lmp.get_state())
        for n in (39, 25, 40):
            lmp.step(n)
        last.append((lmp.get_state(), lmp.info().nbuilds, lmp.history()))
    for k in ("x", "v", "omega", "f", "torque"):
        assert np.array_equal(first[0][k], first[1][k]), k
    assert last[0][1] == last[1][1] and last[0][1] >= (2 if arms else 3)
    assert (last[0][0]["tag"] == last[1][0]["tag"]).all()
    assert np.max(np.abs(last[0][0]["x"] - last[1][0]["x"])) <= 1e-12 * 1e-3
    for k in ("v", "omega", "f", "torque"):
        assert dc.rel_err(last[1][0][k], last[0][0][k]) <= 1e-12, k
    assert last[0][2].keys() == last[1][2].keys()
