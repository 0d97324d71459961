"""Known-answer tests for the parts of the oracle that NO reference golden vector reaches
("parity unpinned" in oracle/sedifoam_oracle.h): gran/hertzFix/history, fix cohesive, pair lubricate/poly,
ErgunWenYu, fix fdrag's added-mass term.  Expected values are derived by hand from the formulas in the cited
reference lines (closed forms for special configurations), not from running the oracle."""
import ctypes as C

import numpy as np
import pytest

from oracle import binding as ob

L = ob.lib()


def _pair(x, v, w, r, m, p, dt=1e-6, shear0=None, hertz=True, shearupdate=1):
    x = ob.f64(x); v = ob.f64(v); w = ob.f64(w); r = ob.f64(r); m = ob.f64(m)
    mask = ob.i32([1, 1])
    ilist = ob.i32([0, 1]); first = ob.i32([0, 1, 1]); jlist = ob.i32([1])
    touch = ob.i32([1 if shear0 is not None else 0])
    shear = ob.f64(shear0 if shear0 is not None else [0, 0, 0])
    nl = ob.NeighList(2, ob.P(ilist), ob.P(first), ob.P(jlist), ob.P(touch), ob.P(shear))
    f = np.zeros((2, 3)); t = np.zeros((2, 3))
    fn = L.orc_pair_gran_hertzfix_history if hertz else L.orc_pair_gran_hooke_history
    fn(C.byref(p), dt, shearupdate, 2, ob.P(x), ob.P(v), ob.P(w), ob.P(r), ob.P(m), ob.P(mask), 0,
       C.byref(nl), ob.P(f), ob.P(t))
    return f, t, shear, touch


def _params(kn=1e7, gamman=0.5, xmu=0.4):
    p = ob.GranParams()
    assert L.orc_gran_settings(C.byref(p), kn, 1, 0.0, gamman, 1, 0.0, xmu, 1, 1.0) == 0
    return p


def test_hertz_static_normal_force_closed_form():
    # two equal spheres at rest, overlap delta: only the Hertz spring acts
    # F = polyhertz * (4/5.46) kn * delta, polyhertz = sqrt(delta * R/2)   (pair_gran_hertzFix_history.cpp:199-200)
    R, delta, kn = 0.5e-3, 2.0e-5, 1e7
    p = _params(kn)
    x = [[0, 0, 0], [2 * R - delta, 0, 0]]
    f, t, sh, touch = _pair(x, np.zeros((2, 3)), np.zeros((2, 3)), [R, R], [1e-6, 1e-6], p)
    F = np.sqrt(delta * R * R / (2 * R)) * 4.0 / 5.46 * kn * delta
    assert f[0, 0] == pytest.approx(-F, rel=1e-14) and f[1, 0] == pytest.approx(F, rel=1e-14)
    assert np.all(f[:, 1:] == 0) and np.all(t == 0) and touch[0] == 1 and np.all(sh == 0)


def test_hertz_normal_damping_uses_gamman_as_restitution():
    # approaching at speed u along the line of centres: extra repulsion sqrt(sn meff) * 2 sqrt(5/6) beta u
    # with beta = -ln(e)/sqrt(ln(e)^2 + pi^2), sn = (2/1.82) kn polyhertz   (:192-200)
    R, delta, kn, e, m, u = 0.5e-3, 2.0e-5, 1e7, 0.5, 2e-6, 0.1
    p = _params(kn, gamman=e)
    x = [[0, 0, 0], [2 * R - delta, 0, 0]]
    v = [[u / 2, 0, 0], [-u / 2, 0, 0]]
    f, _, _, _ = _pair(x, v, np.zeros((2, 3)), [R, R], [m, m], p)
    poly = np.sqrt(delta * R / 2)
    beta = -np.log(e) / np.sqrt(np.log(e) ** 2 + np.pi ** 2)
    spring = poly * 4.0 / 5.46 * kn * delta
    damping = np.sqrt(2.0 / 1.82 * kn * poly * (m / 2)) * 2 * np.sqrt(5.0 / 6.0) * beta * u
    assert f[1, 0] == pytest.approx(spring + damping, rel=1e-13)
    assert f[0, 0] == -f[1, 0]


def test_hertz_tangential_spring_and_coulomb_cap():
    # pure sliding in y with an existing shear displacement s: fs = -polyhertz (8/8.84) kt s  (:234) unless
    # |fs| > xmu |Fn| where it is capped and the stored shear rescaled (:243-253)
    R, delta, kn, xmu = 0.5e-3, 2.0e-5, 1e7, 0.4
    p = _params(kn, xmu=xmu)
    kt = kn * 2.0 / 7.0
    x = [[0, 0, 0], [2 * R - delta, 0, 0]]
    poly = np.sqrt(delta * R / 2)
    Fn = poly * 4.0 / 5.46 * kn * delta
    s_small = 1e-8
    f, t, sh, _ = _pair(x, np.zeros((2, 3)), np.zeros((2, 3)), [R, R], [1e-6] * 2, p, shear0=[0, s_small, 0],
                        shearupdate=1)
    fs = -poly * 8.0 / 8.84 * kt * s_small
    assert abs(fs) < xmu * Fn
    assert f[0, 1] == pytest.approx(fs, rel=1e-13) and f[1, 1] == pytest.approx(-fs, rel=1e-13)
    # torque = -R * (del x fs)/r on both : del = (-(2R-delta),0,0), fs along y -> z torque
    r = 2 * R - delta
    assert t[0, 2] == pytest.approx(-R * (-(r) * fs) / r, rel=1e-13) and t[1, 2] == pytest.approx(t[0, 2], rel=1e-13)
    s_big = 1e-4
    f, t, sh, _ = _pair(x, np.zeros((2, 3)), np.zeros((2, 3)), [R, R], [1e-6] * 2, p, shear0=[0, s_big, 0])
    assert abs(f[0, 1]) == pytest.approx(xmu * Fn, rel=1e-13)
    assert sh[1] == pytest.approx(xmu * Fn / (poly * 8.0 / 8.84 * kt), rel=1e-13)   # rescaled displacement


def test_non_touching_pair_resets_history():
    p = _params()
    f, t, sh, touch = _pair([[0, 0, 0], [1.01e-3, 0, 0]], np.zeros((2, 3)), np.zeros((2, 3)), [0.5e-3] * 2,
                            [1e-6] * 2, p, shear0=[1e-6, 2e-6, 3e-6])
    assert touch[0] == 0 and np.all(sh == 0) and np.all(f == 0)


def _half_list():
    ilist = ob.i32([0, 1]); first = ob.i32([0, 1, 1]); jlist = ob.i32([1])
    return ilist, first, jlist, ob.NeighList(2, ob.P(ilist), ob.P(first), ob.P(jlist), None, None)


def test_cohesive_opt1_hamaker_sphere_sphere():
    # ccel = -ah R^6 / (6 del^2 (r+R)^2 r^3), R = ri+rj, del = r-R  (fix_cohesive.cpp:239-241)
    ah, ri, rj, gap, smin, smax = 1e-19, 0.4e-3, 0.6e-3, 5e-6, 1e-9, 1e-4
    keep = _half_list()
    x = ob.f64([[0, 0, 0], [0, ri + rj + gap, 0]]); rad = ob.f64([ri, rj]); mask = ob.i32([1, 1])
    f = np.zeros((2, 3))
    assert L.orc_fix_cohesive(ah, 1e-7, smin, smax, 1, 2, 0, ob.P(x), ob.P(rad), ob.P(mask), 1, C.byref(keep[3]),
                              ob.P(f)) == 0
    R = ri + rj; r = R + gap
    ccel = -ah * R ** 6 / 6.0 / gap ** 2 / (r + R) ** 2 / r ** 3
    # force on i = del_vec * ccel / r with del_vec = xi - xj = (0,-r,0): attraction pulls i towards +y
    assert f[0, 1] == pytest.approx(-ccel, rel=1e-12) and f[0, 1] > 0 and f[1, 1] == -f[0, 1]
    # below smin the gap is clamped (:242-244)
    x2 = ob.f64([[0, 0, 0], [0, R + 0.5e-9, 0]]); f2 = np.zeros((2, 3))
    L.orc_fix_cohesive(ah, 1e-7, smin, smax, 1, 2, 0, ob.P(x2), ob.P(rad), ob.P(mask), 1, C.byref(keep[3]), ob.P(f2))
    cc = -ah * R ** 6 / 6.0 / smin ** 2 / (smin + 2 * R) ** 2 / (smin + R) ** 3
    assert f2[0, 1] == pytest.approx(-cc, rel=1e-12)
    # beyond smax nothing; invalid option is an error (:262)
    x3 = ob.f64([[0, 0, 0], [0, R + 2e-4, 0]]); f3 = np.zeros((2, 3))
    L.orc_fix_cohesive(ah, 1e-7, smin, smax, 1, 2, 0, ob.P(x3), ob.P(rad), ob.P(mask), 1, C.byref(keep[3]), ob.P(f3))
    assert np.all(f3 == 0)
    assert L.orc_fix_cohesive(ah, 1e-7, smin, smax, 2, 2, 0, ob.P(x), ob.P(rad), ob.P(mask), 1, C.byref(keep[3]),
                              ob.P(f)) == -1


def test_cohesive_opt0_three_branches():
    ah, lam, smin, smax, R = 1e-19, 1e-7, 1e-9, 1e-4, 1e-3
    keep = _half_list()
    rad = ob.f64([R / 2, R / 2]); mask = ob.i32([1, 1])
    for gap in (5e-8, 2e-8, 5e-10):      # > lam/pi ; (smin, lam/pi] ; < smin
        x = ob.f64([[0, 0, 0], [R + gap, 0, 0]]); f = np.zeros((2, 3))
        L.orc_fix_cohesive(ah, lam, smin, smax, 0, 2, 0, ob.P(x), ob.P(rad), ob.P(mask), 1, C.byref(keep[3]), ob.P(f))
        if gap > lam / np.pi:
            cc = -ah * R * lam * (6.4988e-3 - 4.5316e-4 * lam / gap + 1.1326e-5 * lam ** 2 / gap ** 2) / gap ** 3
        else:
            s = max(gap, smin)
            cc = -ah * (lam + 22.242 * s) * R * lam / 24.0 / (lam + 11.121 * s) ** 2 / s ** 2
        assert f[0, 0] == pytest.approx(-cc, rel=1e-12) and f[1, 0] == -f[0, 0]


def test_lubricate_squeeze_equal_spheres_and_inner_cutoff_edit():
    # flaglog = 0: F = 6 pi mu ri (beta0^2/beta1^2/h) vn with h = gap/ri, beta0 = 1 -> 6 pi mu ri /(4 h) * vn
    mu, ri, gap, u = 1e-3, 0.5e-3, 2e-5, 0.05
    lp = ob.LubParams(); lp.mu = mu; lp.flaglog = 0; lp.flagfld = 0; lp.flagHI = 1; lp.flagVF = 0
    lp.cut_inner = 1.001e-3; lp.cut_global = 1.2e-3; lp.vxmu2f = 1.0
    ilist = ob.i32([0, 1]); first = ob.i32([0, 1, 2]); jlist = ob.i32([1, 0])
    nl = ob.NeighList(2, ob.P(ilist), ob.P(first), ob.P(jlist), None, None)
    rad = ob.f64([ri, ri])
    x = ob.f64([[0, 0, 0], [2 * ri + gap, 0, 0]]); v = ob.f64([[u, 0, 0], [0, 0, 0]]); w = np.zeros((2, 3))
    f = np.zeros((2, 3)); t = np.zeros((2, 3))
    L.orc_pair_lubricate_poly(C.byref(lp), 2, ob.P(x), ob.P(v), ob.P(w), ob.P(rad), C.byref(nl), ob.P(f), ob.P(t))
    a_sq = 6 * np.pi * mu * ri * (1.0 / 4.0 / (gap / ri))
    assert f[0, 0] == pytest.approx(-a_sq * u, rel=1e-12)      # resists the approach
    assert f[1, 0] == pytest.approx(+a_sq * u, rel=1e-12)      # full list: j computes its own (opposite) force
    # closer than cut_inner: the reference's edit sets h_sep = 100 (ri + rj) (pair_lubricate_poly.cpp:294-295)
    x2 = ob.f64([[0, 0, 0], [2 * ri + 0.5e-6, 0, 0]]); f2 = np.zeros((2, 3)); t2 = np.zeros((2, 3))
    L.orc_pair_lubricate_poly(C.byref(lp), 2, ob.P(x2), ob.P(v), ob.P(w), ob.P(rad), C.byref(nl), ob.P(f2), ob.P(t2))
    a2 = 6 * np.pi * mu * ri * (1.0 / 4.0 / (100 * 2 * ri / ri))
    assert f2[0, 0] == pytest.approx(-a2 * u, rel=1e-12)
    # isotropic FLD terms and the volume-fraction constants (:213-220, :551-559)
    lp.flagfld = 1; lp.flagHI = 0; lp.flagVF = 1
    L.orc_lubricate_init(C.byref(lp), 2, ob.P(rad), 1e-6)
    vol_f = 2 * 4.0 / 3.0 * np.pi * ri ** 3 / 1e-6
    assert lp.R0 == pytest.approx(6 * np.pi * mu * (1 + 2.16 * vol_f), rel=1e-14)
    assert lp.RT0 == pytest.approx(8 * np.pi * mu, rel=1e-14)
    f3 = np.zeros((2, 3)); t3 = np.zeros((2, 3)); w3 = ob.f64([[0, 0, 2.0], [0, 0, 0]])
    L.orc_pair_lubricate_poly(C.byref(lp), 2, ob.P(x), ob.P(v), ob.P(w3), ob.P(rad), C.byref(nl), ob.P(f3), ob.P(t3))
    assert f3[0, 0] == pytest.approx(-lp.R0 * ri * u, rel=1e-14)
    assert t3[0, 2] == pytest.approx(-lp.RT0 * ri ** 3 * 2.0, rel=1e-14)


def test_ergun_wenyu_branches():
    nu, rho = 1e-6, 1000.0
    Ur = ob.f64([0.1, 0.1, 5.0]); alpha = ob.f64([0.1, 0.5, 0.1]); d = ob.f64([1e-3, 1e-3, 1e-3]); out = np.zeros(3)
    L.orc_ergun_wenyu_jd(3, ob.P(Ur), ob.P(alpha), ob.P(d), nu, rho, ob.P(out))
    # Wen-Yu, beta = 0.9 > 0.8, Re = 90 (ErgunWenYu.C:104-118)
    beta = 0.9; Re = beta * 0.1 * 1e-3 / nu
    Cd = 24 * (1 + 0.15 * Re ** 0.687) / Re
    assert out[0] == pytest.approx(0.75 * Cd * rho * 0.1 * beta ** -2.65 / 1e-3, rel=1e-13)
    # Ergun, beta = 0.5 <= 0.8 (:124-131)
    assert out[1] == pytest.approx(150 * 0.5 * nu * rho / (0.5e-3) ** 2 + 1.75 * rho * 0.1 / 0.5e-3, rel=1e-13)
    # Re = 4500 > 1000 -> Cd = 0.44 (:111-114)
    assert out[2] == pytest.approx(0.75 * 0.44 * rho * 5.0 * 0.9 ** -2.65 / 1e-3, rel=1e-13)


def test_fix_fdrag_added_mass_and_mistyped_pi():
    # f += ffluiddrag + carrier_rho/rho * 0.5 m (DuDt - (v - vOld)/dt), rho from the mistyped pi (:147-157)
    n = 1; dt = 1e-6; r = 0.5e-3; m = 1.4e-6
    v = ob.f64([[0.2, 0, 0]]); vOld = ob.f64([[0.1, 0, 0]]); fd = ob.f64([[1e-6, 2e-6, 3e-6]]); du = ob.f64([[5.0, 0, 0]])
    f = np.zeros((1, 3)); mask = ob.i32([1])
    L.orc_fix_fluid_drag(n, dt, 1000.0, ob.P(v), ob.P(ob.f64([m])), ob.P(ob.f64([r])), ob.P(mask), 1, ob.P(fd), ob.P(du),
                         ob.P(vOld), ob.P(f))
    rho_p = 3.0 * m / (4.0 * 3.14159265358917323846 * r ** 3)
    acc = (0.2 - 0.1) / dt
    assert f[0, 0] == pytest.approx(1e-6 + 1000.0 / rho_p * 0.5 * m * (5.0 - acc), rel=1e-14)
    assert f[0, 1] == pytest.approx(2e-6, rel=1e-14) and vOld[0, 0] == 0.2
    assert rho_p != 3.0 * m / (4.0 * np.pi * r ** 3)        # the typo is preserved (2e-13 relative)


def test_smooth_field_neumann_eigenmode():
    """orc_smooth_field (enhancedCloud.C:790-907): cos(pi*m*(i+1/2)/n) is an eigenvector of the zero-gradient
    7-point Laplacian, so `steps` implicit-Euler steps to tau = b^2/4 scale it by (1 - dtau*lambda)^-steps."""
    n = np.array([16, 5, 3], np.int32); dx = np.array([1e-3, 2e-3, 3e-3]); D = np.array([1.0, 0.5, 2.0])
    band, steps, m = 4e-3, 3, 2
    i = np.arange(n[0])
    mode = np.cos(np.pi * m * (i + 0.5) / n[0])
    f = np.tile(mode, n[1] * n[2]) + 0.75                 # x fastest; the constant is the lambda = 0 mode
    f0 = f.copy()
    L = ob.lib()
    L.orc_smooth_field(ob.P(n), ob.P(dx), ob.P(D), band, steps, 1, ob.P(f))
    dtau = band ** 2 / 4.0 / steps
    lam = -(2.0 - 2.0 * np.cos(np.pi * m / n[0])) * D[0] / dx[0] ** 2
    want = 0.75 + (f0 - 0.75) * (1.0 - dtau * lam) ** (-steps)
    assert np.max(np.abs(f - want)) < 1e-13
    # band width 0 / zero steps: untouched (diffusionRunTime_ never loops)
    g = f0.copy()
    L.orc_smooth_field(ob.P(n), ob.P(dx), ob.P(D), 0.0, steps, 1, ob.P(g))
    assert np.array_equal(g, f0)


def test_smooth_field_cyclic_axis_eigenmodes():
    """Cyclic patch pair along x (the reference's channel cases: blockMeshDict `cyclic`, `boundary pp ff pp`): the
    operator is circulant there, cos(2 pi k i / n) AND sin(2 pi k i / n) are eigenvectors with eigenvalue
    (2 - 2 cos(2 pi k / n)) D / dx^2; along the zero-gradient y axis the Neumann cosine still is.  A point source next
    to the cyclic face spreads across it; with zero-gradient it does not."""
    n = np.array([12, 6, 1], np.int32); dx = np.array([1e-3, 2e-3, 3e-3]); D = np.array([1.0, 0.5, 2.0])
    per = np.array([1, 0, 0], np.int32)
    band, steps, k, m = 4e-3, 3, 2, 1
    ix = np.arange(n[0]); iy = np.arange(n[1])
    fx = np.cos(2 * np.pi * k * ix / n[0]) + 0.5 * np.sin(2 * np.pi * k * ix / n[0])
    fy = np.cos(np.pi * m * (iy + 0.5) / n[1])
    f0 = (fy[:, None] * fx[None, :]).reshape(-1) + 0.25
    f = f0.copy()
    L = ob.lib()
    L.orc_smooth_field_periodic(ob.P(n), ob.P(dx), ob.P(D), band, steps, 1, ob.P(f), ob.P(per))
    dtau = band ** 2 / 4.0 / steps
    lam = (2 - 2 * np.cos(2 * np.pi * k / n[0])) * D[0] / dx[0] ** 2 + (2 - 2 * np.cos(np.pi * m / n[1])) * D[1] / dx[1] ** 2
    want = 0.25 + (f0 - 0.25) * (1.0 + dtau * lam) ** (-steps)
    assert np.max(np.abs(f - want)) < 1e-13
    src = np.zeros(int(n.prod())); src[0] = 1.0                       # cell (0, 0, 0), on the cyclic face
    a = src.copy(); L.orc_smooth_field_periodic(ob.P(n), ob.P(dx), ob.P(D), band, steps, 1, ob.P(a), ob.P(per))
    b = src.copy(); L.orc_smooth_field(ob.P(n), ob.P(dx), ob.P(D), band, steps, 1, ob.P(b))
    assert a[n[0] - 1] == pytest.approx(a[1], rel=1e-12) and a[n[0] - 1] > 10 * b[n[0] - 1]
    assert a.sum() == pytest.approx(1.0, rel=1e-12) and b.sum() == pytest.approx(1.0, rel=1e-12)
    # graded solver with uniform widths and the cyclic pair: the same system
    w = [np.full(n[q], dx[q]) for q in range(3)]
    wp = (ob.dp * 3)(ob.P(w[0]), ob.P(w[1]), ob.P(w[2]))
    c = src.copy(); L.orc_smooth_field_graded_periodic(ob.P(n), ob.P(dx), wp, ob.P(D), band, steps, 1, ob.P(c), ob.P(per))
    assert np.max(np.abs(c - a)) < 1e-13


@pytest.mark.parametrize("e", [0.3, 0.5, 0.9])
def test_hertz_head_on_collision_restitution_equals_gamman(e):
    """Physics-level check of the hertzFix damping (pair_gran_hertzFix_history.cpp:192-200): with
    beta = -ln(e)/sqrt(ln^2 e + pi^2) and the damping 2 sqrt(5/6) beta sqrt(sn meff) v_n the coefficient of restitution
    of a binary head-on collision is e, independent of the impact speed (Antypov & Elliott 2011) -- here through
    the oracle's full DEM loop (nve/sphere + list + history), two impact speeds."""
    R, rho, kn = 0.5e-3, 2500.0, 1.0e7
    m = 4.0 / 3.0 * np.pi * R ** 3 * rho
    for u in (0.05, 0.4):
        x = np.array([[1.0e-3, 2.0e-3, 2.0e-3], [2.2e-3, 2.0e-3, 2.0e-3]])
        v = np.array([[u / 2, 0, 0], [-u / 2, 0, 0]])
        dem = ob.OracleDem(x, [R, R], [m, m], [0, 0, 0], [4e-3, 4e-3, 4e-3], periodic=(0, 0, 0), v=v)
        dem.pair_gran("hertz", kn, None, e, None, 0.0, 1)
        dem.fix_gravity(0.0, 0.0, -1.0, 0.0)
        dem.fix_fdrag(0.0)
        dem.neighbor(0.3e-3)
        dem.timestep(2.0e-8)
        dem.setup()
        gap = 0.2e-3
        steps = int((gap / u + 4.0e-4) / 2.0e-8)          # approach + a generous contact time
        dem.run(steps)
        st = dem.get()
        assert st["x"][1, 0] - st["x"][0, 0] > 2 * R          # separated again
        e_meas = (st["v"][1, 0] - st["v"][0, 0]) / u
        assert e_meas == pytest.approx(e, abs=2e-3), (u, e_meas)


def _wall(x, v, r, p, wallstyle, lo=0.0, hi=1.0, cyl=0.0, wiggle=0, shear=0, axis=0, amp=0.0, period=1.0, vshear=0.0,
          steps=0, dt=1e-6, hertz=True):
    x = ob.f64([x]); v = ob.f64([v]); w = np.zeros((1, 3)); rr = ob.f64([r]); m = ob.f64([1.0e-6])
    mask = ob.i32([1]); sh = np.zeros((1, 3)); f = np.zeros((1, 3)); t = np.zeros((1, 3))
    L.orc_fix_wall_gran_moving(C.byref(p), 2 if hertz else 1, wallstyle, lo, hi, cyl, wiggle, shear, axis, amp, period,
                               vshear, steps, dt, 1, 1, ob.P(x), ob.P(v), ob.P(w), ob.P(rr), ob.P(m), ob.P(mask), 1,
                               ob.P(sh), ob.P(f), ob.P(t))
    return f[0], t[0], sh[0]


def test_wall_wiggle_moves_the_plane_and_gives_it_a_velocity():
    """fix_wall_granFix.cpp:257-263: wlo = lo + A - A cos(omega t), vwall[axis] = A omega sin(omega t), t = steps dt."""
    p = _params(gamman=1.0 - 1e-12, xmu=0.0)      # no damping (beta -> 0), no friction: pure Hertz spring
    r, A, T, dt = 0.5e-3, 0.1e-3, 1.0e-3, 1e-6
    y = r + 0.5 * A                                # half an amplitude above contact with the resting floor
    f0, _, _ = _wall([0, y, 0], [0, 0, 0], r, p, 1, lo=0.0, hi=1.0, wiggle=1, axis=1, amp=A, period=T, steps=0, dt=dt)
    assert np.all(f0 == 0.0)                       # t = 0: the floor is at lo
    # half a period later the floor stands at lo + 2 A: overlap 1.5 A, wall at rest (sin = 0 up to rounding)
    fh, _, _ = _wall([0, y, 0], [0, 0, 0], r, p, 1, lo=0.0, hi=1.0, wiggle=1, axis=1, amp=A, period=T, steps=500, dt=dt)
    ov = 1.5 * A
    fn = np.sqrt(ov * r) * 4.0 / 5.46 * 1e7 * ov   # hertzFix spring of the wall twin (:602-608): polyhertz (4/5.46) kn delta
    assert fh[1] == pytest.approx(fn, rel=1e-9) and abs(fh[0]) < 1e-12 * fn
    # a quarter period: floor at lo + A moving up at A omega; with damping the approaching wall pushes harder
    pd = _params(gamman=0.5, xmu=0.0)
    fq, _, _ = _wall([0, y, 0], [0, 0, 0], r, pd, 1, lo=0.0, hi=1.0, wiggle=1, axis=1, amp=A, period=T, steps=250, dt=dt)
    fs, _, _ = _wall([0, y - A, 0], [0, 0, 0], r, pd, 1, lo=0.0, hi=1.0)     # same overlap on a wall at rest
    fv, _, _ = _wall([0, y - A, 0], [0, -A * 2 * np.pi / T, 0], r, pd, 1, lo=0.0, hi=1.0)   # ... and the same closing speed
    assert fq[1] > fs[1] > 0.0 and fq[1] == pytest.approx(fv[1], rel=1e-12)


def test_wall_shear_drags_the_particle_along_and_is_capped_by_coulomb():
    """:264 vwall[axis] = vshear; the tangential spring builds up with the relative velocity v - vwall (:457-459)."""
    p = _params(gamman=0.5, xmu=0.4)
    r = 0.5e-3
    f, t, sh = _wall([0, r * 0.98, 0], [0, 0, 0], r, p, 1, lo=0.0, hi=1.0, shear=1, axis=0, vshear=1.0e-3)
    assert f[0] > 0.0 and f[1] > 0.0              # dragged along +x, pushed off the floor
    assert sh[0] == pytest.approx(-1.0e-3 * 1e-6, rel=1e-9)   # shear += (v - vwall) dt, below the Coulomb limit
    assert t[2] > 0.0                              # force +x at the contact point -r y: torque +z
    # a fast wall saturates at mu * Fn
    f2, _, _ = _wall([0, r * 0.98, 0], [0, 0, 0], r, p, 1, lo=0.0, hi=1.0, shear=1, axis=0, vshear=2.0e3)
    fn_only, _, _ = _wall([0, r * 0.98, 0], [0, 0, 0], r, p, 1, lo=0.0, hi=1.0)
    assert f2[0] == pytest.approx(0.4 * fn_only[1], rel=1e-9)


def test_z_cylinder_contact_is_radial_and_rotation_is_tangential():
    """:309-322: del = -(R - |xy|)/|xy| (x, y, 0); sheared about x or y the wall rotates, vwall = v (y, -x, 0)/|xy|."""
    p = _params(gamman=0.5, xmu=0.4)
    r, R = 0.5e-3, 5.0e-3
    ov = 0.02 * r
    rad = R - r + ov
    ang = 0.7
    x = [rad * np.cos(ang), rad * np.sin(ang), 0.3e-3]
    f, _, _ = _wall(x, [0, 0, 0], r, p, 3, cyl=R)
    plane, _, _ = _wall([0, r - ov, 0], [0, 0, 0], r, p, 1, lo=0.0, hi=1.0)
    assert np.hypot(f[0], f[1]) == pytest.approx(plane[1], rel=1e-9) and f[2] == 0.0
    assert f[0] * x[0] + f[1] * x[1] < 0.0 and abs(f[0] * x[1] - f[1] * x[0]) < 1e-9 * plane[1] * rad   # inward, radial
    far, _, _ = _wall([0.1e-3, 0, 0], [0, 0, 0], r, p, 3, cyl=R)
    assert np.all(far == 0.0)
    # rotating cylinder: the tangential force points along the wall velocity (y, -x, 0)/|xy| * v
    g, _, _ = _wall(x, [0, 0, 0], r, p, 3, cyl=R, shear=1, axis=0, vshear=1.5)
    tang = np.array([x[1], -x[0]]) / rad
    assert g[0] * tang[0] + g[1] * tang[1] > 0.0
    # shear along z slides the wall along its axis instead
    h, _, _ = _wall(x, [0, 0, 0], r, p, 3, cyl=R, shear=1, axis=2, vshear=1.5)
    assert h[2] > 0.0 and abs(h[0] * tang[0] + h[1] * tang[1]) < 1e-9 * abs(h[2])


def test_graded_block_cell_owner_and_smoothing():
    """Graded (simpleGrading) block: the owner is the cell whose face interval [f_i, f_i+1) holds the point; the
    volume-weighted smoothing solver reproduces the uniform one when the widths are uniform, conserves sum(V phi) and
    never widens the range on a graded block."""
    n = np.array([6, 5, 4], np.int32)
    dx = np.array([1e-3, 1.2e-3, 0.8e-3]); origin = np.array([0.0, -1e-3, 2e-3])
    yf = origin[1] + np.array([0.0, 0.5, 1.3, 2.5, 4.1, 6.0]) * 1e-3
    faces = (ob.dp * 3)(None, ob.P(yf), None)
    pts = np.array([[0.5e-3, yf[0], 2.1e-3],            # on the first face: inside, cell 0
                    [0.5e-3, yf[2], 2.1e-3],            # on an inner face: the upper cell
                    [0.5e-3, np.nextafter(yf[2], -1), 2.1e-3],
                    [0.5e-3, yf[5], 2.1e-3],            # on the last face: outside
                    [5.5e-3, 0.5 * (yf[3] + yf[4]), 2e-3 + 3.1 * 0.8e-3]])
    cell = np.zeros(len(pts), np.int32)
    L.orc_cell_owner_graded(len(pts), ob.P(pts), ob.P(origin), ob.P(dx), ob.P(n), faces, ob.P(cell))
    assert list(cell) == [0, 0 + 6 * 2, 0 + 6 * 1, -1, 5 + 6 * (3 + 5 * 3)]
    D = np.array([1.0, 0.5, 2.0])
    rng = np.random.default_rng(0)
    f = rng.uniform(size=(int(n.prod()), 3))
    a = f.copy(); L.orc_smooth_field(ob.P(n), ob.P(dx), ob.P(D), 3e-3, 3, 3, ob.P(a.reshape(-1)))
    w = [np.full(n[k], dx[k]) for k in range(3)]
    wp = (ob.dp * 3)(ob.P(w[0]), ob.P(w[1]), ob.P(w[2]))
    b = f.copy(); L.orc_smooth_field_graded(ob.P(n), ob.P(dx), wp, ob.P(D), 3e-3, 3, 3, ob.P(b.reshape(-1)))
    assert np.abs(a - b).max() <= 1e-13 * np.abs(a).max()
    w[1] = np.ascontiguousarray(np.diff(yf))
    wp = (ob.dp * 3)(None, ob.P(w[1]), None)
    V = (w[0][:, None, None] * w[1][None, :, None] * w[2][None, None, :]).transpose(2, 1, 0).reshape(-1)
    c = f.copy(); L.orc_smooth_field_graded(ob.P(n), ob.P(dx), wp, ob.P(D), 3e-3, 3, 3, ob.P(c.reshape(-1)))
    assert np.allclose((V[:, None] * c).sum(0), (V[:, None] * f).sum(0), rtol=1e-13)
    assert c.min() >= f.min() and c.max() <= f.max() and c.std() < 0.5 * f.std()
    # a field that is constant stays constant (zero-gradient walls, row sums of the operator vanish)
    one = np.full(int(n.prod()), 3.25)
    L.orc_smooth_field_graded(ob.P(n), ob.P(dx), wp, ob.P(D), 3e-3, 3, 1, ob.P(one))
    assert np.allclose(one, 3.25, rtol=1e-14)


def test_inlet_override_known_answers():
    """enhancedCloud.C:249-257 with softParticleCloud::pointInRegion (:1354-1417): inside the region the force becomes
    m (inletForce - U) / deltaT, outside it stays; option 1 = box with its faces, option 2 = between two cylinders around
    the axis (x1,y1,z1)-(x2,y2,z2); off (option 0 or zero inletForce) nothing changes."""
    import ctypes as C
    L = ob.lib()
    pos = np.array([[0.5, 0.5, 0.5], [1.0, 0.2, 0.3], [1.5, 0.5, 0.5], [0.5, 0.5, 2.0]])   # inside, on a face, outside x, outside z
    U = np.array([[0.1, 0.0, 0.0], [0.0, 0.2, 0.0], [0.0, 0.0, 0.3], [0.1, 0.1, 0.1]])
    m = np.array([2.0, 3.0, 4.0, 5.0])
    F = np.array([0.5, -0.25, 0.125]); dT = 1.0e-3
    box = np.array([0.0, 1.0, 0.0, 1.0, 0.0, 1.0, 0.0, 0.0, 0.0]); ecc = np.zeros(3)
    base = np.arange(12, dtype=float).reshape(4, 3) + 1.0
    p = base.copy()
    L.orc_inlet_force_override(1, ob.P(F), ob.P(box), ob.P(ecc), dT, 4, ob.P(pos), ob.P(m), ob.P(U), ob.P(p))
    for i in (0, 1):
        assert np.allclose(p[i], m[i] * (F - U[i]) / dT, rtol=1e-15)
    assert np.array_equal(p[2:], base[2:])
    for opt, f in ((0, F), (1, np.zeros(3))):     # switched off
        p = base.copy()
        L.orc_inlet_force_override(opt, ob.P(f), ob.P(box), ob.P(ecc), dT, 4, ob.P(pos), ob.P(m), ob.P(U), ob.P(p))
        assert np.array_equal(p, base)
    # hollow cylinder along z from z = 0 to 1 around (0.5, 0.5): r1 = 0.1, r2 = 0.4
    cyl = np.array([0.5, 0.5, 0.5, 0.5, 0.0, 1.0, 0.1, 0.4, 0.0])
    pts = np.array([[0.5, 0.5, 0.5], [0.7, 0.5, 0.5], [0.95, 0.5, 0.5], [0.7, 0.5, 1.2]])   # on the axis, in the shell, beyond r2, beyond the end
    p = base.copy()
    L.orc_inlet_force_override(2, ob.P(F), ob.P(cyl), ob.P(ecc), dT, 4, ob.P(pts), ob.P(m), ob.P(U), ob.P(p))
    assert np.allclose(p[1], m[1] * (F - U[1]) / dT, rtol=1e-15)
    assert np.array_equal(p[[0, 2, 3]], base[[0, 2, 3]])
    # the eccentricity shifts the inner cylinder only: with it at (0.2, 0, 0) the point at x = 0.7 sits on its axis
    p = base.copy()
    L.orc_inlet_force_override(2, ob.P(F), ob.P(cyl), ob.P(np.array([0.2, 0.0, 0.0])), dT, 4, ob.P(pts), ob.P(m), ob.P(U),
                               ob.P(p))
    assert np.array_equal(p[1], base[1]) and not np.array_equal(p[0], base[0])


def test_fix_freeze_acts_where_it_stands_in_the_fix_list():
    """[3P] Modify::post_force runs the fixes in script order: a wall registered AFTER fix freeze still pushes a frozen
    grain (cases/example-cases/transport-bedload/in.lammps:28-31), one registered before it does not."""
    from tests import dem_cases as dc
    d = 1.0e-3
    bed = dict(x=np.array([[1.5e-3, 0.45e-3, 1.5e-3], [1.5e-3, 2.0e-3, 1.5e-3]]), v=np.zeros((2, 3)),
               diameter=np.full(2, d), density=np.full(2, 2650.0), boxlo=np.zeros(3), boxhi=np.full(3, 3.0e-3),
               periodic=(1, 0, 1), n=2, type=np.array([2, 1], dtype=np.int32))
    base = dict(pair="hooke", kn=2.0e3, gamman=0.0, xmu=0.0, g=9.81, dt=1.0e-6, skin=0.25e-3,
                walls=[(1, 0.0, 3.0e-3)], frozen_types=[2], nve_all=True)
    f = {}
    for first in (False, True):
        orc = dc.make_oracle(bed, dict(base, freeze_first=first))
        orc.setup()
        f[first] = orc.get()["f"]
    m = 4.0 * np.pi / 3.0 * (0.5 * d) ** 3 * 2650.0
    assert np.all(f[False][0] == 0.0)                                        # freeze last: nothing left
    assert f[True][0][1] == pytest.approx(2.0e3 * 0.05e-3, rel=1e-12)        # freeze first: the wall spring kn * overlap only
    assert f[True][0][0] == 0.0 and f[True][0][2] == 0.0                     # (no gravity: it came before the freeze)
    for k in (False, True):
        assert f[k][1][1] == pytest.approx(-m * 9.81, rel=1e-12)             # the free grain just falls


def test_plain_hooke_pair_and_wall_closed_forms():
    """`pair_style gran/hooke` [3P] and FixWallGranFix::hooke (fix_wall_granFix.cpp:347-437): Hookean spring + normal
    damping along the line of centres, tangential force = velocity damping meff gammat vrel capped by mu |Fn|, no
    history.  gammat = gamman / 2 (NULL), meff = m/2 between equal grains, m against a wall."""
    R, delta, kn, gn, mu, m = 0.5e-3, 2.0e-5, 1e7, 50.0, 0.4, 2e-6
    p = _params(kn, gamman=gn, xmu=mu)
    x = ob.f64([[0, 0, 0], [2 * R - delta, 0, 0]])
    w = np.zeros((2, 3)); r = ob.f64([R, R]); mm = ob.f64([m, m]); mask = ob.i32([1, 1])
    ilist = ob.i32([0, 1]); first = ob.i32([0, 1, 1]); jlist = ob.i32([1])

    def pair(v, p=p):
        v = ob.f64(v)
        touch = ob.i32([0]); shear = np.zeros(3)
        nl = ob.NeighList(2, ob.P(ilist), ob.P(first), ob.P(jlist), ob.P(touch), ob.P(shear))
        f = np.zeros((2, 3)); t = np.zeros((2, 3))
        L.orc_pair_gran_hooke(C.byref(p), 2, ob.P(x), ob.P(v), ob.P(w), ob.P(r), ob.P(mm), ob.P(mask), 0,
                              C.byref(nl), ob.P(f), ob.P(t))
        assert np.all(shear == 0.0)
        return f, t

    # at rest: the spring alone, kn delta
    f, t = pair(np.zeros((2, 3)))
    assert f[1, 0] == pytest.approx(kn * delta, rel=1e-13) and f[0, 0] == -f[1, 0] and np.all(t == 0)
    # closing at speed u: + meff gamman u
    u = 0.05
    f, _ = pair([[u / 2, 0, 0], [-u / 2, 0, 0]])
    fn = kn * delta + 0.5 * m * gn * u
    assert f[1, 0] == pytest.approx(fn, rel=1e-13)
    # sliding slowly past each other at relative speed s along y: damping force meff (gamman/2) s, opposing the motion
    sl = 1.0e-3
    f, t = pair([[0, sl / 2, 0], [0, -sl / 2, 0]])
    ft = 0.5 * m * (0.5 * gn) * sl
    assert ft < mu * kn * delta
    assert f[0, 1] == pytest.approx(-ft, rel=1e-12) and f[1, 1] == pytest.approx(ft, rel=1e-12)
    # torque on grain 0: -radi * rinv (del x fs), del = x0 - x1 = (-(2R - delta), 0, 0), fs = (0, -ft, 0)
    assert t[0, 2] == pytest.approx(-R * ft, rel=1e-12) and t[1, 2] == pytest.approx(-R * ft, rel=1e-12)
    # sliding fast on a nearly frictionless contact: capped at mu |Fn|
    f, _ = pair([[0, 50.0, 0], [0, -50.0, 0]], _params(kn, gamman=gn, xmu=1e-6))
    assert 0.5 * m * (0.5 * gn) * 100.0 > 1e-6 * kn * delta
    assert f[1, 1] == pytest.approx(1e-6 * kn * delta, rel=1e-12)

    # the wall twin: meff = m, radius in place of radsum; a floor at y = 0
    xs = ob.f64([[0, R - delta, 0]]); rr = ob.f64([R]); m1 = ob.f64([m]); mk = ob.i32([1])
    for vel, want_y, want_x in (([0, 0, 0], kn * delta, 0.0),
                                ([0, -u, 0], kn * delta + m * gn * u, 0.0),
                                ([sl, 0, 0], kn * delta, -m * (0.5 * gn) * sl)):
        v1 = ob.f64([vel]); sh = np.full((1, 3), 7.0); f = np.zeros((1, 3)); t = np.zeros((1, 3))
        L.orc_fix_wall_gran(C.byref(p), 3, 1, 0.0, 1.0, 1e-6, 1, 1, ob.P(xs), ob.P(v1), ob.P(np.zeros((1, 3))),
                            ob.P(rr), ob.P(m1), ob.P(mk), 1, ob.P(sh), ob.P(f), ob.P(t))
        assert f[0, 1] == pytest.approx(want_y, rel=1e-13)
        assert f[0, 0] == pytest.approx(want_x, rel=1e-12, abs=1e-30)
        assert np.all(sh == 7.0)      # the plain law keeps no shear array (:327: `if (pairstyle != HOOKE)`)


def gaussian_point_source_case(N=57, cells_per_band=8, band=6e-3):
    """the reference's own derivation (documentation/diffusionEqn/diffusionEqn.tex, section 2): a point source of unit
    solid volume diffused to tau = b^2 / 4 IS the Gaussian kernel of band width b,
    K(r, tau) = (4 pi tau)^(-3/2) exp(-r^2 / (4 tau)).  Fine uniform mesh (b / 8 per cell), source in the centre cell."""
    h = band / cells_per_band
    n = np.array([N, N, N], np.int32)
    c = N // 2
    ax = (np.arange(N) - c) * h
    Z, Y, X = np.meshgrid(ax, ax, ax, indexing="ij")      # field[z, y, x], x fastest
    tau = band * band / 4.0
    G = np.exp(-(X ** 2 + Y ** 2 + Z ** 2) / (4.0 * tau)) / (4.0 * np.pi * tau) ** 1.5
    f0 = np.zeros(N ** 3)
    f0[(c * N + c) * N + c] = 1.0 / h ** 3
    return n, np.array([h, h, h]), f0, G, X, tau


def check_against_the_gaussian(smooth, band=6e-3):
    """`smooth(n, dx, f0, band, steps)` -> field.  What must hold for the reference's six implicit steps
    (cloudProperties diffusionSteps, enhancedCloud.C:836-905) and how the document's Gaussian is approached:
      * the solid volume is conserved (1e-12) -- the conservation property the document puts first;
      * the variance per axis is EXACTLY 2 tau for any number of implicit-Euler steps (sum x^2 L f = 2 sum f for the
        3-point Laplacian): 0.5 % here, what the +-3.5 b of the mesh cut off;
      * six steps give the implicit-Euler kernel (1 + tau k^2 / 6)^-6, peakier than exp(-tau k^2): within 0.46 of the
        Gaussian's peak (measured 0.455); the difference falls first order in 1 / steps -- 0.10 at 24 steps, 0.035 at 96
        (what remains is the one-cell width of the source)."""
    n, dx, f0, G, X, tau = gaussian_point_source_case(band=band)
    h3 = float(dx[0]) ** 3
    errs = {}
    for steps in (6, 24, 96):
        f = smooth(n, dx, f0.copy(), band, steps).reshape(G.shape)
        assert abs(f.sum() * h3 - 1.0) < 1e-12
        assert abs((f * X ** 2).sum() * h3 / (2.0 * tau) - 1.0) < 5e-3
        assert f.min() > -1e-12 * f.max()
        errs[steps] = float(np.abs(f - G).max() / G.max())
    assert errs[6] < 0.46 and errs[24] < 0.10 and errs[96] < 0.035
    assert errs[24] < errs[6] / 3.5 and errs[96] < errs[24] / 2.5
    return errs


def test_smooth_field_point_source_is_the_documents_gaussian():
    """N1 pinned to the reference's own documentation instead of a hand-derived answer (see check_against_the_gaussian)"""
    L = ob.lib()
    D = np.ones(3)

    def smooth(n, dx, f, band, steps):
        L.orc_smooth_field(ob.P(n), ob.P(dx), ob.P(D), band, steps, 1, ob.P(f))
        return f
    check_against_the_gaussian(smooth)
