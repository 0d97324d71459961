"""Pin the oracle against the reference's OWN golden vectors.

The reference's regression suite (cases/auto-testing) only holds whole-case curves.  Two of them
can be reproduced by the particle hot path alone with the fluid held at its (known, uniform)
initial state -- one-way coupling -- because the particles are dilute (alpha < 0.4 %):

  * xiaocase3: one d = 83 um sphere released at rest in a uniform 0.05 m/s stream, g = 0.
    golden: data/lammps08.dat (t, vx, vy, vz) and data/xiaoCase3.dat (t, vy benchmark curve).
  * multiParticlesCollideRho / Dia: four spheres settling in still water (hydrostatic pressure
    gradient = buoyancy); particles 1 and 4 never touch anything before the last dump row.
    golden: data/origin/p[14].dat dump rows every 1000 DEM steps.

They exercise: SyamlalOBrien::Jd, the drag assembly of enhancedCloud::updateDragOnParticles
(drag + pressure-gradient force), particleToEulerianField (alpha), adjustLampTimestep, the
lammps_put_local_info -> fix fdrag -> nve/sphere loop, fix gravity and gran/hooke/history + wall/gran
being silent for non-touching particles.  Tolerances reflect what one-way coupling neglects
(fluid response to the particle, diffusion smoothing of alpha), not oracle freedom.
"""
import os

import numpy as np
import pytest

from oracle import binding as ob

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _coupled_oneway(x0, d, rho_p, box, mesh_n, Uf, gradp, deltaT, dt_in, sub_cycles, n_cfd,
                    pair, walls, gravity, drag_model=1, rhob=1000.0, nub=1e-6, skin=5e-4,
                    sample_every=1):
    """enhancedCloud::evolve() with a frozen uniform fluid; returns (t, x, v) histories."""
    L = ob.lib()
    n = len(d)
    d = ob.f64(d)
    r = 0.5 * d
    # read_data for atom_style sphere: mass = 4/3 pi r^3 rho (true pi) [3P]
    m = 4.0 * np.pi / 3.0 * r ** 3 * ob.f64(rho_p)
    dem = ob.OracleDem(x0, r, m, box[0], box[1])
    dem.pair_gran(*pair)
    dem.fix_gravity(*gravity)
    dem.fix_fdrag(0.0)
    for w in walls:
        dem.fix_wall(*w)
    dem.neighbor(skin)
    # adjustLampTimestep (softParticleCloud.C:209-261)
    import ctypes as C
    dtadj = C.c_double(); steps = C.c_int(); sc = C.c_int(); ss = C.c_int()
    rc = L.orc_adjust_timestep(deltaT, dt_in, sub_cycles, C.byref(dtadj), C.byref(steps),
                               C.byref(sc), C.byref(ss))
    assert rc == 0
    dem.timestep(dtadj.value)
    dem.setup()   # lammps_step(0) at construction (softParticleCloud.C:189)

    origin = ob.f64(box[0]); ncell = ob.i32(mesh_n)
    dx = (ob.f64(box[1]) - origin) / ncell
    ncells = int(np.prod(ncell))
    V = np.full(ncells, float(np.prod(dx)))
    Uf_c = np.tile(ob.f64(Uf), (ncells, 1))
    gp_c = np.tile(ob.f64(gradp), (ncells, 1))
    zeros_c = np.zeros((ncells, 3))
    fl = ob.CloudFlags()
    fl.particleDrag = 1; fl.particlePressureGrad = 1
    fl.rhob = rhob; fl.nub = nub; fl.deltaT = deltaT
    gamma = np.zeros(ncells); Ue = np.zeros((ncells, 3))
    cell = np.zeros(n, dtype=np.int32)

    def scatter(st):
        L.orc_cell_owner(n, ob.P(st["x"]), ob.P(origin), ob.P(dx), ob.P(ncell), ob.P(cell))
        L.orc_particle_to_eulerian(n, ob.P(cell), ob.P(d), ob.P(st["v"]), ncells, ob.P(V),
                                   ob.P(gamma), ob.P(Ue))

    st = dem.get()
    scatter(st)   # constructor: particleToEulerianField (enhancedCloud.C:635)
    ts, xs, vs = [0.0], [st["x"].copy()], [st["v"].copy()]
    Uri = np.zeros((n, 3)); mag = np.zeros(n); Jd = np.zeros(n)
    pDrag = np.zeros((n, 3)); pDuDt = np.zeros((n, 3))
    UOld = st["v"].copy()
    for it in range(n_cfd):
        for k in range(sc.value):
            L.orc_cell_owner(n, ob.P(st["x"]), ob.P(origin), ob.P(dx), ob.P(ncell), ob.P(cell))
            L.orc_drag_on_particles(C.byref(fl), drag_model, n, ob.P(cell), ob.P(st["x"]),
                                    ob.P(d), ob.P(st["v"]), ob.P(UOld), ob.P(gamma), ob.P(Uf_c),
                                    ob.P(gp_c), ob.P(zeros_c), ob.P(zeros_c), ob.P(Uri),
                                    ob.P(mag), ob.P(Jd), ob.P(pDrag), ob.P(pDuDt))
            dem.put_fdrag(pDrag, st["tag"])
            dem.run(ss.value)
            UOld = st["v"]
            st = dem.get()
            if k == 0:
                scatter(st)   # enhancedCloud.C:773-776
        if (it + 1) % sample_every == 0:
            ts.append((it + 1) * deltaT); xs.append(st["x"].copy()); vs.append(st["v"].copy())
    return np.array(ts), np.array(xs), np.array(vs)


def test_xiaocase3_single_particle_entrainment():
    # cases/auto-testing/test-cases/xiaocase3: IC_uniform.in, in.lammps, system/controlDict,
    # constant/{cloudProperties,transportProperties}, 0/Ub
    box = ([0.0, 0.0, 0.0], [4e-3, 4e-3, 5e-4])
    walls = [(0, 0.0, 4e-3, 5000.0, None, 11200.0, None, 0.1, 0),
             (1, 0.0, 4e-3, 5000.0, None, 11200.0, None, 0.1, 0),
             (2, 0.0, 5e-4, 5000.0, None, 11200.0, None, 0.1, 0)]
    t, x, v = _coupled_oneway(
        x0=[[2e-3, 1.9e-3, 2.5e-4]], d=[8.3e-5], rho_p=[2000.0], box=box, mesh_n=[10, 10, 1],
        Uf=[0.0, 0.05, 0.0], gradp=[0.0, 0.0, 0.0], deltaT=2e-5, dt_in=2e-7, sub_cycles=1,
        n_cfd=250, pair=("hooke", 5000.0, None, 11200.0, None, 0.1, 0), walls=walls,
        gravity=(0.0, 0.0, -1.0, 0.0))
    vy = v[:, 0, 1]
    # benchmark curve the reference overlays its result on (13 rows, digitised: +-1 % noise,
    # ends at 0.0504 > 0.05 because the real channel flow accelerates slightly at the axis)
    bench = np.loadtxt(os.path.join(GOLD, "xiaocase3_xiaoCase3.dat"))
    checked = 0
    for tt, vv in bench:
        if tt < 2e-4 or tt > 5e-3:
            continue
        assert np.interp(tt, t, vy) == pytest.approx(vv, rel=0.05), (tt, vv)
        checked += 1
    assert checked >= 9
    # "lammps08 code" rows (t vx vy vz).  The t = 5e-4 row (0.02575) contradicts the benchmark
    # curve of the same case (0.0315 at that time) by 20 %, so it is only bounded loosely.
    gold = np.loadtxt(os.path.join(GOLD, "xiaocase3_lammps08.dat"))
    for row in gold[1:]:
        got = np.interp(row[0], t, vy)
        tol = 0.25 if row[0] < 1e-3 else 0.04
        assert got == pytest.approx(row[2], rel=tol), (row[0], got, row[2])
    # terminal value 0.0500031: the particle ends up riding with the stream
    assert vy[-1] == pytest.approx(0.0500031, rel=2e-3)
    # x and z stay put
    assert abs(x[-1, 0, 0] - 2e-3) < 1e-9 and abs(x[-1, 0, 2] - 2.5e-4) < 1e-9


@pytest.mark.parametrize("case,ds,rhos,x0", [
    ("Rho", [1.5e-3] * 4, [4650.0, 3650.0, 2650.0, 1650.0],
     [[5e-2, 7.5e-2, 5e-2], [9e-2, 8.5e-2, 5e-2], [9.1e-2, 8.5e-2, 5e-2], [1.7e-1, 7.5e-2, 5e-2]]),
    ("Dia", [3.5e-3, 3.0e-3, 2.5e-3, 2.0e-3], [2650.0] * 4,
     [[5e-2, 7.5e-2, 5e-2], [9e-2, 8.5e-2, 5e-2], [9.2e-2, 8.5e-2, 5e-2], [1.7e-1, 6.5e-2, 5e-2]]),
])
def test_multi_particles_settling(case, ds, rhos, x0):
    # cases/auto-testing/test-cases/multiParticlesCollide{Rho,Dia}: deltaT 1e-3, timestep 1e-5,
    # subCycles 2, gravity 9.8 (in.lammps) / g (0 -9.8 0) for the fluid => grad p = rho_f g
    box = ([0.0, 0.0, 0.0], [0.2, 0.1, 0.1])
    wp = (4910.0, None, 0.0, None, 0.0, 0)
    walls = [(0, 0.0, 0.2) + wp, (1, 0.0, 0.1) + wp, (2, 0.0, 0.1) + wp]
    t, x, v = _coupled_oneway(
        x0=x0, d=ds, rho_p=rhos, box=box, mesh_n=[40, 20, 1], Uf=[0.0, 0.0, 0.0],
        gradp=[0.0, -9.8 * 1000.0, 0.0], deltaT=1e-3, dt_in=1e-5, sub_cycles=2, n_cfd=200,
        pair=("hooke", 4910.0, None, 0.0, None, 0.15, 0), walls=walls,
        gravity=(9.8, 0.0, -1.0, 0.0), skin=0.02, sample_every=10)
    for pid in (1, 2, 3, 4):
        gold = np.loadtxt(os.path.join(GOLD, "multiParticlesCollide%s_p%d.dat" % (case, pid)))
        nrow = min(len(gold), len(t))
        # mass column of the dump pins read_data's density->mass conversion
        m = 4.0 * np.pi / 3.0 * (0.5 * ds[pid - 1]) ** 3 * rhos[pid - 1]
        assert m == pytest.approx(gold[0, 3], rel=2e-6)
        for k in range(2, nrow):
            assert v[k, pid - 1, 1] == pytest.approx(gold[k, 8], rel=0.05), (pid, k)
            assert x[k, pid - 1, 1] == pytest.approx(gold[k, 5], abs=1.0e-3), (pid, k)
            # particles 2 and 3 start overlapped (0.5 / 0.75 mm) and are thrown apart along x by
            # gran/hooke/history: the x they coast to pins the contact impulse (they travel
            # 40-75 mm; two-way fluid coupling accounts for the remaining few mm)
            assert x[k, pid - 1, 0] == pytest.approx(gold[k, 4], abs=3.5e-3), (pid, k)
    # terminal settling velocity of the isolated heavy sphere (Rho case: -0.31418 m/s)
    if case == "Rho":
        assert v[-1, 0, 1] == pytest.approx(-0.314177, rel=3e-3)
