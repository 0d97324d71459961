"""Shared set-up for the DEM parity tests: the same script-level configuration drives the HIP engine
(through the lammps_* C-ABI, sedifoam_amd.Lammps) and the CPU oracle (oracle/binding.py)."""
import numpy as np

from oracle import binding as ob


def make_oracle(bed, cfg):
    r = 0.5 * bed["diameter"]
    m = 4.0 * np.pi / 3.0 * r ** 3 * bed["density"]
    dem = ob.OracleDem(bed["x"], r, m, bed["boxlo"], bed["boxhi"], periodic=bed["periodic"], v=bed["v"],
                       omega=bed.get("omega"))
    style = cfg.get("pair", "hertz")
    dem.pair_gran(style, cfg["kn"], None, cfg["gamman"], None, cfg["xmu"], cfg.get("dampflag", 1))
    if cfg.get("lub"):
        dem.pair_lubricate(*cfg["lub"])
    dem.fix_gravity(cfg["g"], 0.0, -1.0, 0.0)
    dem.fix_fdrag(cfg.get("carrier_rho", 0.0))
    for (dim, lo, hi) in cfg["walls"]:
        dem.fix_wall(dim, lo, hi, cfg["kn"], None, cfg["gamman"], None, cfg["xmu"], cfg.get("dampflag", 1))
    if cfg.get("cohesive"):
        dem.fix_cohesive(*cfg["cohesive"])
    dem.neighbor(cfg["skin"])
    dem.timestep(cfg["dt"])
    return dem


def script_lines(bed, cfg):
    style = {"hertz": "gran/hertzFix/history", "hooke": "gran/hooke/history"}[cfg.get("pair", "hertz")]
    wall = "wall/granFix" if cfg.get("pair", "hertz") == "hertz" else "wall/gran"
    gran = "%s %.17g NULL %.17g NULL %.17g %d" % (style, cfg["kn"], cfg["gamman"], cfg["xmu"],
                                                   cfg.get("dampflag", 1))
    if cfg.get("lub"):
        mu, flaglog, flagfld, cin, cgl, fhi, fvf = cfg["lub"]
        pair = "pair_style hybrid/overlay %s lubricate/poly %.17g %d %d %.17g %.17g %d %d" % (
            gran, mu, flaglog, flagfld, cin, cgl, fhi, fvf)
    else:
        pair = "pair_style " + gran
    p = bed["periodic"]
    lines = ["atom_style sphere", "boundary %s %s %s" % tuple("p" if q else "f" for q in p), "newton off",
             "communicate single vel yes", "neighbor %.17g bin" % cfg["skin"], "neigh_modify delay 0", pair,
             "pair_coeff * *", "timestep %.17g" % cfg["dt"], "fix 1 all nve/sphere",
             "fix 2 all gravity %.17g vector 0 -1 0" % cfg["g"]]
    cr = cfg.get("carrier_rho", 0.0)
    lines.append("fix 3 all fdrag" + (" %d" % int(cr) if cr else ""))
    for k, (dim, lo, hi) in enumerate(cfg["walls"]):
        lines.append("fix w%d all %s %.17g NULL %.17g NULL %.17g %d %splane %s %s" % (
            k, wall, cfg["kn"], cfg["gamman"], cfg["xmu"], cfg.get("dampflag", 1), "xyz"[dim],
            "NULL" if lo is None else "%.17g" % lo, "NULL" if hi is None else "%.17g" % hi))
    if cfg.get("cohesive"):
        lines.append("fix coh all cohesive %.17g %.17g %.17g %.17g %d" % tuple(cfg["cohesive"]))
    return lines


def make_hip(bed, cfg):
    from sedifoam_amd import Lammps
    lmp = Lammps()
    lmp.set_box(bed["boxlo"], bed["boxhi"])
    lmp.create_atoms(bed["x"], bed["diameter"], bed["density"], v=bed["v"], omega=bed.get("omega"))
    for line in script_lines(bed, cfg):
        lmp.command(line)
    return lmp


def rel_err(a, b):
    """max |a-b| / max |b| (a global scale, so near-zero components do not blow up)"""
    scale = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (scale if scale > 0 else 1.0))
