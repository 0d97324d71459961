"""Seeded synthetic packings used by the benchmark and the parity tests:
monodisperse (or uniformly polydisperse) spheres on an FCC lattice with nearest-neighbour spacing
`spacing * d` (0.98 => every one of the 12 neighbours overlaps by 2 %), uniform jitter and small random
velocities.  Pure numpy host code; identical inputs feed the GPU engine and the CPU oracle."""
import numpy as np


def fcc_cells_for(n_target):
    """(ncx, ncy, ncz) with 4*ncx*ncy*ncz >= n_target, as cubic as possible."""
    c = max(1, int(round((n_target / 4.0) ** (1.0 / 3.0))))
    best = None
    for a in range(max(1, c - 2), c + 3):
        for b in range(max(1, c - 2), c + 3):
            for e in range(max(1, c - 2), c + 3):
                n = 4 * a * b * e
                if n >= n_target and (best is None or n < best[0]):
                    best = (n, a, b, e)
    return best[1:]


def fcc_bed(ncells, d=1.0e-3, spacing=0.98, jitter=0.005, vmax=0.01, seed=12345, rho=2650.0,
            poly=None, y_gap_top=None):
    """FCC bed periodic in x and z, resting on a wall at y = 0.

    Returns dict(x, v, diameter, density, boxlo, boxhi, periodic).  `poly=(dmin, dmax)` draws diameters
    uniformly (lattice spacing then follows dmax)."""
    ncx, ncy, ncz = ncells
    rng = np.random.default_rng(seed)
    dmax = d if poly is None else poly[1]
    a = spacing * dmax              # nearest-neighbour distance
    edge = a * np.sqrt(2.0)         # conventional cubic cell
    basis = np.array([[0, 0, 0], [0.5, 0.5, 0], [0.5, 0, 0.5], [0, 0.5, 0.5]])
    ii, jj, kk = np.meshgrid(np.arange(ncx), np.arange(ncy), np.arange(ncz), indexing="ij")
    cells = np.stack([ii.ravel(), jj.ravel(), kk.ravel()], axis=1).astype(np.float64)
    x = (cells[:, None, :] + basis[None, :, :]).reshape(-1, 3) * edge
    n = x.shape[0]
    x[:, 0] += 0.25 * edge
    x[:, 2] += 0.25 * edge
    x[:, 1] += 0.5 * spacing * dmax   # bottom layer overlaps the y = 0 wall like a neighbour would
    x += rng.uniform(-jitter * dmax, jitter * dmax, size=(n, 3))
    v = rng.uniform(-vmax, vmax, size=(n, 3))
    diam = np.full(n, d) if poly is None else rng.uniform(poly[0], poly[1], size=n)
    dens = np.full(n, rho)
    top = ncy * edge
    gap = y_gap_top if y_gap_top is not None else 0.25 * top + 4 * dmax
    boxlo = np.array([0.0, 0.0, 0.0])
    boxhi = np.array([ncx * edge, top + gap, ncz * edge])
    return dict(x=x, v=v, diameter=diam, density=dens, boxlo=boxlo, boxhi=boxhi,
                periodic=(1, 0, 1), n=n, edge=edge)


def hertz_script(bed, kn=1.0e7, gamman=0.5, xmu=0.4, dt=1.0e-6, skin_d=0.25, g=9.81, d=None,
                 pair="gran/hertzFix/history", wall="wall/granFix", extra=()):
    """The in.lammps-style command list of the synthetic Hertz bed."""
    d = d if d is not None else float(np.max(bed["diameter"]))
    p = bed["periodic"]
    lines = [
        "atom_style sphere",
        "boundary %s %s %s" % tuple("p" if q else "f" for q in p),
        "newton off",
        "communicate single vel yes",
        "neighbor %.17g bin" % (skin_d * d),
        "neigh_modify delay 0",
        "pair_style %s %.17g NULL %.17g NULL %.17g 1" % (pair, kn, gamman, xmu),
        "pair_coeff * *",
        "timestep %.17g" % dt,
        "fix 1 all nve/sphere",
        "fix 2 all gravity %.17g vector 0 -1 0" % g,
        "fix 3 all fdrag",
        "fix ywall all %s %.17g NULL %.17g NULL %.17g 1 yplane %.17g %.17g"
        % (wall, kn, gamman, xmu, bed["boxlo"][1], bed["boxhi"][1]),
    ]
    lines.extend(extra)
    return lines


def grown_poly_bed(n_target, dlo=0.5e-3, dhi=1.5e-3, phi=0.58, seed=15, vmax=0.05, rho=2650.0, growth=1.02,
                   relax_steps=150, final_steps=600, verbose=False):
    """A DENSE disordered polydisperse bed in a fully periodic box: d ~ U(dlo, dhi) (seeded; SURVEY.md 8d: C5 is
    d ~ U(0.5, 1.5) mm, size ratio 3), solid fraction `phi`.

    A lattice cannot hold such a bed (a site spacing that fits the largest grains leaves the small ones floating, one that
    fits the mean makes every fourth pair overlap by a quarter of a diameter), so the bed is GROWN: grains start on
    jittered FCC sites at a common scale factor at which nothing overlaps and are inflated by `growth` per stage to their
    full size; every stage is relaxed by the HIP engine itself (frictionless Hertz contacts at restitution 0.05 -- gran/hertzFix's
    `gamman` IS the restitution coefficient --, no gravity,
    velocities zeroed between stages) -- a Lubachevsky-Stillinger-style compression.  The engine is bit-reproducible, so
    the bed is a deterministic function of its arguments.  Runs on the GPU (no CPU fallback exists): for `bench.py` and the
    `-m gpu` tests, which hand the SAME finished bed to the engine and to the CPU oracle.

    Returns the dictionary of `fcc_bed` (+ `stages`: how many growth stages ran)."""
    from .lammps import Lammps
    ncx, ncy, ncz = fcc_cells_for(n_target)
    rng = np.random.default_rng(seed)
    n = 4 * ncx * ncy * ncz
    diam = rng.uniform(dlo, dhi, size=n)
    vol = float(np.sum(np.pi / 6.0 * diam ** 3))
    boxvol = vol / phi
    edge = (boxvol / (ncx * ncy * ncz)) ** (1.0 / 3.0)   # conventional cubic cell
    a = edge / np.sqrt(2.0)                              # nearest-neighbour distance of the sites
    basis = np.array([[0, 0, 0], [0.5, 0.5, 0], [0.5, 0, 0.5], [0, 0.5, 0.5]])
    ii, jj, kk = np.meshgrid(np.arange(ncx), np.arange(ncy), np.arange(ncz), indexing="ij")
    cells = np.stack([ii.ravel(), jj.ravel(), kk.ravel()], axis=1).astype(np.float64)
    x = ((cells[:, None, :] + basis[None, :, :]).reshape(-1, 3) + 0.25) * edge
    x += rng.uniform(-0.04 * a, 0.04 * a, size=(n, 3))
    v = rng.uniform(-vmax, vmax, size=(n, 3))
    boxlo = np.zeros(3)
    boxhi = np.array([ncx, ncy, ncz], dtype=np.float64) * edge
    dens = np.full(n, rho)
    tag = np.arange(1, n + 1, dtype=np.int32)
    s = 0.9 * a / dhi          # (largest pair: 0.9 a apart at most 0.08 a closer than a: nothing overlaps)
    stage = 0
    while True:
        s = min(1.0, s * growth)
        last = s >= 1.0
        lmp = Lammps()
        lmp.set_box(boxlo, boxhi)
        lmp.create_atoms(x, diam * s, dens, v=np.zeros((n, 3)), tag=tag)
        for line in ["atom_style sphere", "boundary p p p", "newton off", "communicate single vel yes",
                     "neighbor %.17g bin" % (0.1 * dhi), "neigh_modify delay 0",
                     "pair_style gran/hertzFix/history 1e7 NULL 0.05 NULL 0.0 1", "pair_coeff * *", "timestep 1e-6",
                     "fix 1 all nve/sphere"]:
            lmp.command(line)
        lmp.setup()
        lmp.step(final_steps if last else relax_steps)
        st = lmp.get_state()
        x = np.mod(st["x"] - boxlo, boxhi - boxlo) + boxlo
        if verbose:
            print("grown_poly_bed: stage %d scale %.4f max|v| %.3g" % (stage, s, float(np.abs(st["v"]).max())))
        lmp.close()
        stage += 1
        if last:
            break
    return dict(x=x, v=v, diameter=diam, density=dens, boxlo=boxlo, boxhi=boxhi, periodic=(1, 1, 1), n=n, edge=edge,
                stages=stage)
