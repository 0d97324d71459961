"""The oracle against numbers computed by THE REFERENCE'S OWN SOURCE LINES (tests/golden/reference_pins.json, made by
tests/golden/make_reference_pins.py in the build container: the ii / jj loop of PairGranHertzFixHistory::compute, FixCohe::
post_force, FixFluidDrag::post_force and the two dragModel::Jd bodies, read from /root/reference at generation time,
transliterated statement by statement and executed on seeded LAMMPS-shaped inputs).  No golden vector of the reference's
own test suite touches these closures; this is their pin: a typo in oracle/orc_*.c -- a constant, an operator, the order
of a product -- fails here.  CPU only; the HIP path is held to the oracle by the -m gpu tests."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from oracle import binding as ob

HERE = os.path.dirname(os.path.abspath(__file__))
PINS = json.load(open(os.path.join(HERE, "golden", "reference_pins.json")))


def unhex(a):
    return np.array([[float.fromhex(v) for v in row] for row in a]) if a and isinstance(a[0], list) else \
        np.array([float.fromhex(v) for v in a])


def csr(firstneigh, nlocal):
    first = np.zeros(nlocal + 1, dtype=np.int32)
    for i in range(nlocal):
        first[i + 1] = first[i] + len(firstneigh[i])
    jl = np.array([j for i in range(nlocal) for j in firstneigh[i]] or [0], dtype=np.int32)
    return first, jl


def ulps(a, b):
    """largest distance in units of the last place of the larger magnitude"""
    a, b = np.asarray(a, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()
    # (NaN where the reference's lines give NaN -- the log of a negative lubrication gap -- and nowhere else)
    if np.isnan(b).any() or np.isnan(a).any():
        if not np.array_equal(np.isnan(a), np.isnan(b)):
            return float("inf")
        a, b = a[~np.isnan(b)], b[~np.isnan(b)]
    scale = np.maximum(np.spacing(np.maximum(np.abs(a), np.abs(b))), 5e-324)
    return float(np.max(np.abs(a - b) / scale)) if a.size else 0.0


@pytest.mark.parametrize("k", range(len(PINS["pair_gran_hertzFix_history.cpp:109-286"])))
def test_hertzfix_history_compute_equals_the_reference_lines(k):
    c = PINS["pair_gran_hertzFix_history.cpp:109-286"][k]
    I, O = c["inp"], c["out"]
    L = ob.lib()
    p = ob.GranParams()
    # settings(): kt and gammat as the reference derives them are inputs of the loop, passed through unchanged
    assert L.orc_gran_settings(C.byref(p), I["kn"], 0, I["kt"], I["gamman"], 1, 0.0, I["xmu"], 1, 1.0) == 0
    n, nlocal = I["n"], I["nlocal"]
    first, jl = csr(I["firstneigh"], nlocal)
    touch = np.array([t for i in range(nlocal) for t in I["touch"][i]] or [0], dtype=np.int32)
    shear = np.array([s for i in range(nlocal) for s in I["shear"][i]] or [0.0])
    ilist = np.arange(nlocal, dtype=np.int32)
    nl = ob.NeighList(nlocal, ob.P(ilist), ob.P(first), ob.P(jl), ob.P(touch), ob.P(shear))
    f, tq = np.zeros((n, 3)), np.zeros((n, 3))
    # the fix_rigid branch (:182-185) replaces mi / mj of an atom of a rigid body by the body's mass and nothing else: the
    # oracle's mass argument IS the mass the pair law sees
    mass = ob.f64(I["rmass"])
    if "mass_rigid" in I:
        mr = ob.f64(I["mass_rigid"])
        assert 5 < np.count_nonzero(mr) < n
        mass = np.where(mr > 0.0, mr, mass)
    L.orc_pair_gran_hertzfix_history(C.byref(p), I["dt"], I["shearupdate"], nlocal, ob.P(ob.f64(I["x"])),
                                     ob.P(ob.f64(I["v"])), ob.P(ob.f64(I["omega"])), ob.P(ob.f64(I["radius"])),
                                     ob.P(mass), ob.P(ob.i32(I["mask"])), I["freeze_group_bit"],
                                     C.byref(nl), ob.P(f), ob.P(tq))
    ref_touch = [t for i in range(nlocal) for t in O["touch"][i]]
    ref_shear = np.array([float.fromhex(s) for i in range(nlocal) for s in O["shear"][i]])
    assert touch[:len(ref_touch)].tolist() == ref_touch
    assert sum(ref_touch) >= 10 and len(ref_touch) - sum(ref_touch) >= 3       # both branches were exercised
    # the pair terms are identical to the last bit; f and torque are sums over pairs in the same order
    assert ulps(shear[:ref_shear.size], ref_shear) <= 1.0
    assert ulps(f, unhex(O["f"])) <= 1.0 and ulps(tq, unhex(O["torque"])) <= 1.0


@pytest.mark.parametrize("k", range(len(PINS["fix_cohesive.cpp:161-262"])))
def test_fix_cohesive_post_force_equals_the_reference_lines(k):
    c = PINS["fix_cohesive.cpp:161-262"][k]
    I, O = c["inp"], c["out"]
    L = ob.lib()
    n, nlocal = I["n"], I["nlocal"]
    first, jl = csr(I["firstneigh"], nlocal)
    ilist = np.arange(nlocal, dtype=np.int32)
    nl = ob.NeighList(nlocal, ob.P(ilist), ob.P(first), ob.P(jl), None, None)
    f = np.zeros((n, 3))
    assert L.orc_fix_cohesive(I["ah"], I["lam"], I["smin"], I["smax"], I["opt"], nlocal, I["newton_pair"],
                              ob.P(ob.f64(I["x"])), ob.P(ob.f64(I["radius"])), ob.P(ob.i32(I["mask"])), I["groupbit"],
                              C.byref(nl), ob.P(f)) == 0
    ref = unhex(O["f"])
    assert np.count_nonzero(ref) > 20
    assert ulps(f, ref) <= 1.0


@pytest.mark.parametrize("k", range(len(PINS["fix_fluid_drag.cpp:143-163"])))
def test_fix_fluid_drag_post_force_equals_the_reference_lines(k):
    c = PINS["fix_fluid_drag.cpp:143-163"][k]
    I, O = c["inp"], c["out"]
    L = ob.lib()
    f, vOld = ob.f64(I["f"]).copy(), ob.f64(I["vOld"]).copy()
    L.orc_fix_fluid_drag(I["n"], I["dt"], I["carrier_rho"], ob.P(ob.f64(I["v"])), ob.P(ob.f64(I["rmass"])),
                         ob.P(ob.f64(I["radius"])), ob.P(ob.i32(I["mask"])), I["groupbit"],
                         ob.P(ob.f64(I["ffluiddrag"])), ob.P(ob.f64(I["DuDt"])), ob.P(vOld), ob.P(f))
    assert ulps(f, unhex(O["f"])) <= 1.0
    assert np.array_equal(vOld, unhex(O["vOld"]))


@pytest.mark.parametrize("key,fn", [("ErgunWenYu.C:104-132", "orc_ergun_wenyu_jd"),
                                    ("SyamlalOBrien.C:105-143", "orc_syamlal_obrien_jd")])
def test_drag_model_jd_equals_the_reference_lines(key, fn):
    L = ob.lib()
    for c in PINS[key]:
        I, O = c["inp"], c["out"]
        jd = np.zeros(I["n"])
        getattr(L, fn)(I["n"], ob.P(ob.f64(I["Ur"])), ob.P(ob.f64(I["alpha"])), ob.P(ob.f64(I["pd"])), I["nuf"],
                       I["rhof"], ob.P(jd))
        ref = unhex(O["Jd"])
        assert np.all(np.isfinite(ref))
        assert ulps(jd, ref) <= 1.0, (key, ulps(jd, ref))


WALL_KEY = "fix_wall_granFix.cpp:286-344,361-436,446-553,563-678"


@pytest.mark.parametrize("k", range(len(PINS[WALL_KEY])))
def test_fix_wall_granfix_post_force_equals_the_reference_lines(k):
    """plane walls: the post_force loop and the hooke / hooke_history / hertz_history laws, as the reference's lines compute
    them; pairstyle enum of the reference (HOOKE 0, HOOKE_HISTORY 1, HERTZ_HISTORY 2) -> the oracle's 3 / 1 / 2"""
    c = PINS[WALL_KEY][k]
    I, O = c["inp"], c["out"]
    L = ob.lib()
    p = ob.GranParams()
    assert L.orc_gran_settings(C.byref(p), I["kn"], 0, I["kt"], I["gamman"], 0, I["gammat"], I["xmu"], 1, 1.0) == 0
    n = I["n"]
    shear = ob.f64(I["shear"]).copy()
    f, tq = np.zeros((n, 3)), np.zeros((n, 3))
    L.orc_fix_wall_gran(C.byref(p), {0: 3, 1: 1, 2: 2}[I["pairstyle"]], I["wallstyle"], I["lo"], I["hi"], I["dt"],
                        I["shearupdate"], n, ob.P(ob.f64(I["x"])), ob.P(ob.f64(I["v"])), ob.P(ob.f64(I["omega"])),
                        ob.P(ob.f64(I["radius"])), ob.P(ob.f64(I["rmass"])), ob.P(ob.i32(I["mask"])), I["groupbit"],
                        ob.P(shear), ob.P(f), ob.P(tq))
    ref_f = unhex(O["f"])
    assert np.count_nonzero(ref_f[:, I["wallstyle"]]) >= 20
    assert ulps(f, ref_f) <= 1.0 and ulps(tq, unhex(O["torque"])) <= 1.0
    if I["pairstyle"] != 0:
        assert ulps(shear, unhex(O["shear"])) <= 1.0


LUB_KEY = "pair_lubricate_poly.cpp:193-407,539-559"


@pytest.mark.parametrize("k", range(len(PINS[LUB_KEY])))
def test_lubricate_poly_equals_the_reference_lines(k):
    """init_style's volume-fraction constants R0 / RT0 / RS0 and the compute loop (log series or the 1 / h term, FLD
    terms, the reference's 100 (ri + rj) gap below the inner cutoff, full list: only i is updated)"""
    c = PINS[LUB_KEY][k]
    I, O = c["inp"], c["out"]
    L = ob.lib()
    n, nlocal = I["n"], I["nlocal"]
    p = ob.LubParams(I["mu"], I["flaglog"], I["flagfld"], I["flagHI"], I["flagVF"], I["cut_inner"], I["cut_global"],
                     0.0, 0.0, 0.0, 1.0)
    L.orc_lubricate_init(C.byref(p), n, ob.P(ob.f64(I["radius"])), I["vol_T"])
    for name in ("R0", "RT0", "RS0"):
        assert ulps([getattr(p, name)], [float.fromhex(O[name])]) <= 1.0, name
    first, jl = csr(I["firstneigh"], nlocal)
    ilist = np.arange(nlocal, dtype=np.int32)
    nl = ob.NeighList(nlocal, ob.P(ilist), ob.P(first), ob.P(jl), None, None)
    f, tq = np.zeros((n, 3)), np.zeros((n, 3))
    L.orc_pair_lubricate_poly(C.byref(p), nlocal, ob.P(ob.f64(I["x"])), ob.P(ob.f64(I["v"])), ob.P(ob.f64(I["omega"])),
                              ob.P(ob.f64(I["radius"])), C.byref(nl), ob.P(f), ob.P(tq))
    ref_f = unhex(O["f"])
    assert np.count_nonzero(ref_f) > 60 and O["overlaps"] >= 1
    if I["cut_inner"] < 1.0e-3:   # the case with overlapping pairs BEYOND the inner cutoff: log(h_sep < 0), :286-300
        assert 5 < np.isnan(ref_f).any(axis=1).sum() < n
    assert ulps(f, ref_f) <= 1.0 and ulps(tq, unhex(O["torque"])) <= 1.0


# ---- the OpenFOAM-side loops of enhancedCloud.C (A6, A8, A9): particleToEulerianField, updateParticleAlpha / Ur, the force
# assembly of updateDragOnParticles with the history force and the inlet override, calcTcFields -- executed line by line by
# tests/golden/make_reference_pins.py through a three-component vector with OpenFOAM's operators ----
CLOUD_KEY = "enhancedCloud.C:41-108,129-311,318-439,913-979"


def _vunhex(rows):
    return np.array([[float.fromhex(v) for v in r] for r in rows])


def replay_cloud_with_the_oracle(I, check):
    """walk the oracle through the CFD steps of one pin case; check(step, name, value) sees every output of a recorded step"""
    L = ob.lib()
    n, (nx, ny, nz) = I["n"], I["mesh_n"]
    ncells = nx * ny * nz
    fl = ob.CloudFlags()
    F = I["flags"]
    fl.particleDrag = int(F.get("particleDrag", 1)); fl.particlePressureGrad = int(F.get("particlePressureGrad", 1))
    fl.particleBuoyancy = int(F.get("particleBuoyancy", 0)); fl.particleAddedMass = int(F.get("particleAddedMass", 0))
    fl.particleLift = int(F.get("particleLift", 0)); fl.lubricationForce = int(F.get("lubricationForce", 0))
    fl.gravity = (C.c_double * 3)(*I["gravity"]); fl.rhob = I["rhob"]; fl.nub = I["nub"]; fl.deltaT = I["deltaT"]
    hist = bool(F.get("particleHistoryForce", 0))
    cell = ob.i32(I["cell"]); pos = ob.f64(I["pos"]); d = ob.f64(I["d"])
    V = np.full(ncells, I["dx"] ** 3)
    Uf, UfOld = ob.f64(I["Uf"]), ob.f64(I["UfOld"])
    DDtUf, gradp, curlU = ob.f64(I["DDtUf"]), ob.f64(I["gradp"]), ob.f64(I["curlU"])
    sumFb, n0 = np.zeros((n, 3)), np.zeros(n)
    for step in range(1, I["n_steps"] + 1):
        U, UOld = ob.f64(I["U"][step]), ob.f64(I["U"][step - 1])
        gamma, Ue = np.zeros(ncells), np.zeros((ncells, 3))
        L.orc_particle_to_eulerian(n, ob.P(cell), ob.P(d), ob.P(U), ncells, ob.P(V), ob.P(gamma), ob.P(Ue))
        Uri, mag, Jd = np.zeros((n, 3)), np.zeros(n), np.zeros(n)
        pDrag, pDuDt = np.zeros((n, 3)), np.zeros((n, 3))
        L.orc_drag_on_particles_hist(C.byref(fl), I["model"], n, ob.P(cell), ob.P(pos), ob.P(d), ob.P(U), ob.P(UOld),
                                     ob.P(gamma), ob.P(Uf), ob.P(gradp), ob.P(DDtUf), ob.P(curlU), step if hist else -1,
                                     ob.P(UfOld), ob.P(sumFb), ob.P(n0), ob.P(Uri), ob.P(mag), ob.P(Jd), ob.P(pDrag),
                                     ob.P(pDuDt))
        if I["inlet"]:
            J = I["inlet"]
            L.orc_inlet_force_override(J["addParticleOption"], ob.P(ob.f64(J["inletForce"])), ob.P(ob.f64(J["inletBox"])),
                                       ob.P(ob.f64(J["eccentricity"])), I["deltaT"], n, ob.P(pos), ob.P(ob.f64(I["mass"])),
                                       ob.P(U), ob.P(pDrag))
        Asrc, Omega = np.zeros((ncells, 3)), np.ones(ncells)
        L.orc_calc_tc_fields(n, ob.P(cell), ob.P(d), ob.P(U), ob.P(Jd), ncells, ob.P(V), ob.P(gamma), ob.P(Uf), ob.P(Asrc),
                             ob.P(Omega))
        for name, val in (("gamma", gamma), ("Ue", Ue), ("Uri", Uri), ("magUri", mag), ("Jd", Jd), ("pDrag", pDrag),
                          ("pDuDt", pDuDt), ("sumDeltaFb", sumFb), ("n0", n0), ("Asrc", Asrc), ("Omega", Omega)):
            check(step, name, val.copy())


@pytest.mark.parametrize("k", range(len(PINS[CLOUD_KEY])))
def test_cloud_loops_equal_the_reference_lines(k):
    """orc_cloud.c (A6 drag assembly incl. added-mass cap, lift, wall lubrication, the Basset history force through its
    window reset, both inlet regions; A8 scatter; A9 Asrc) against the reference's own statements, output by output"""
    c = PINS[CLOUD_KEY][k]
    want = {o["step"]: o for o in c["out"]}
    seen = []
    worst = {}

    def check(step, name, val):
        if step not in want:
            return
        ref = _vunhex(want[step][name]) if isinstance(want[step][name][0], list) else unhex(want[step][name])
        seen.append((step, name))
        worst[name] = max(worst.get(name, 0.0), ulps(val, ref))
    replay_cloud_with_the_oracle(c["inp"], check)
    assert len(seen) == 11 * len(want)
    # sums over the particles of a cell and products of four factors: the oracle keeps the reference's order
    for name, u in worst.items():
        assert u <= 2.0, (name, u, worst)
    F = c["inp"]["flags"]
    if F.get("particleHistoryForce"):
        last = c["out"][-1]
        assert any(float.fromhex(v) > 0.0 for v in last["n0"]), "the history window never reset: the else branch is not pinned"
    if c["inp"]["inlet"]:
        # the override replaced the assembled force of the particles inside the region, and only theirs
        I = c["inp"]
        m, U = np.array(I["mass"]), np.array(I["U"][1])
        over = m[:, None] * (np.array(I["inlet"]["inletForce"])[None, :] - U) / I["deltaT"]
        got = _vunhex(c["out"][0]["pDrag"])
        inside = np.all(np.abs(got - over) <= 1e-12 * np.abs(over).max(), axis=1)
        assert 0 < inside.sum() < I["n"]
