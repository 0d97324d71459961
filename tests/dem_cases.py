"""Shared set-up for the DEM parity tests: the same script-level configuration drives the HIP engine
(through the lammps_* C-ABI, sedifoam_amd.Lammps) and the CPU oracle (oracle/binding.py)."""
import numpy as np

from oracle import binding as ob


def subset(bed, mine):
    """the atoms of `bed` selected by the boolean mask `mine` (one slab of a decomposed run); tags = global index + 1"""
    sub = dict(bed)
    for k in ("x", "v", "diameter", "density", "omega", "type"):
        if bed.get(k) is not None:
            sub[k] = np.asarray(bed[k])[mine]
    sub["tag"] = (np.nonzero(mine)[0] + 1).astype(np.int32)
    sub["n"] = int(np.count_nonzero(mine))
    return sub


def slab_mask(bed, rank, world):
    lo, hi = float(bed["boxlo"][0]), float(bed["boxhi"][0])
    w = (hi - lo) / world
    return (bed["x"][:, 0] >= lo + rank * w) & ((bed["x"][:, 0] < lo + (rank + 1) * w) | (rank == world - 1))


def make_oracle(bed, cfg):
    r = 0.5 * bed["diameter"]
    m = 4.0 * np.pi / 3.0 * r ** 3 * bed["density"]
    dem = ob.OracleDem(bed["x"], r, m, bed["boxlo"], bed["boxhi"], periodic=bed["periodic"], v=bed["v"],
                       omega=bed.get("omega"), tag=bed.get("tag"))
    style = cfg.get("pair", "hertz")
    dem.pair_gran(style, cfg["kn"], None, cfg["gamman"], None, cfg["xmu"], cfg.get("dampflag", 1))
    if cfg.get("lub"):
        dem.pair_lubricate(*cfg["lub"])
    dem.fix_gravity(cfg["g"], 0.0, -1.0, 0.0)
    dem.fix_fdrag(cfg.get("carrier_rho", 0.0))
    frozen = cfg.get("frozen_types") is not None
    if frozen and cfg.get("freeze_first"):
        dem.fix_freeze(2)   # the reference's order: fix 4 bottom freeze, THEN fix ywall all wall/gran
    for wall in cfg["walls"]:
        dim, lo, hi = wall[:3]
        extra = wall[3] if len(wall) > 3 else {}
        dem.fix_wall(min(dim, 2), lo, hi, cfg["kn"], None, cfg["gamman"], None, cfg["xmu"], cfg.get("dampflag", 1))
        if dim == 3:
            dem.wall_cylinder(extra["cyl"])
        if "wiggle" in extra:
            dem.wall_motion("wiggle", *extra["wiggle"])
        if "shear" in extra:
            dem.wall_motion("shear", *extra["shear"])
    if cfg.get("cohesive"):
        dem.fix_cohesive(*cfg["cohesive"])
    if cfg.get("frozen_types") is not None:
        # group bottom type 2 / group active subtract all bottom (bits 2 and 4), as in the reference's bed cases
        t = np.asarray(bed["type"])
        bottom = np.isin(t, cfg["frozen_types"])
        dem.set_mask(1 + 2 * bottom + 4 * (~bottom))
        nve = 1 if cfg.get("nve_all") else 4   # the reference's bed cases: fix 1 all nve/sphere
        dem.set_groups(nve=nve, gravity=nve, fdrag=1 if cfg.get("fdrag_group", "all") == "all" else 4, wall=1,
                       cohesive=1, freeze=0 if cfg.get("freeze_first") else 2)
    dem.neighbor(cfg["skin"])
    dem.timestep(cfg["dt"])
    return dem


def script_lines(bed, cfg):
    style = {"hertz": "gran/hertzFix/history", "hooke": "gran/hooke/history",
             "hooke_plain": "gran/hooke"}[cfg.get("pair", "hertz")]
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
    frozen = cfg.get("frozen_types") is not None
    act = "active" if frozen and not cfg.get("nve_all") else "all"
    lines = ["atom_style sphere", "boundary %s %s %s" % tuple("p" if q else "f" for q in p), "newton off",
             "communicate single vel yes", "neighbor %.17g bin" % cfg["skin"], "neigh_modify delay 0", pair,
             "pair_coeff * *", "timestep %.17g" % cfg["dt"]]
    if frozen:   # cases/example-cases/*/in.lammps: a bed resting on a layer of fixed particles
        lines += ["group bottom type " + " ".join(str(t) for t in cfg["frozen_types"]),
                  "group active subtract all bottom"]
    lines += ["fix 1 %s nve/sphere" % act, "fix 2 %s gravity %.17g vector 0 -1 0" % (act, cfg["g"])]
    cr = cfg.get("carrier_rho", 0.0)
    lines.append("fix 3 %s fdrag" % (act if cfg.get("fdrag_group", "all") != "all" else "all")
                 + (" %d" % int(cr) if cr else ""))
    if frozen and cfg.get("freeze_first"):
        lines.append("fix 4 bottom freeze")
    for k, wl in enumerate(cfg["walls"]):
        dim, lo, hi = wl[:3]
        extra = wl[3] if len(wl) > 3 else {}
        # walls: (dim, lo, hi[, extra]); dim 3 = zcylinder extra["cyl"]; extra["wiggle"] = (axis, amplitude, period),
        # extra["shear"] = (axis, vshear)  (fix_wall_granFix.cpp:83-141)
        geom = ("zcylinder %.17g" % extra["cyl"]) if dim == 3 else "%splane %s %s" % (
            "xyz"[dim], "NULL" if lo is None else "%.17g" % lo, "NULL" if hi is None else "%.17g" % hi)
        line = "fix w%d all %s %.17g NULL %.17g NULL %.17g %d %s" % (
            k, wall, cfg["kn"], cfg["gamman"], cfg["xmu"], cfg.get("dampflag", 1), geom)
        if "wiggle" in extra:
            line += " wiggle %s %.17g %.17g" % ("xyz"[extra["wiggle"][0]], extra["wiggle"][1], extra["wiggle"][2])
        if "shear" in extra:
            line += " shear %s %.17g" % ("xyz"[extra["shear"][0]], extra["shear"][1])
        lines.append(line)
    if cfg.get("cohesive"):
        lines.append("fix coh all cohesive %.17g %.17g %.17g %.17g %d" % tuple(cfg["cohesive"]))
    if frozen and not cfg.get("freeze_first"):
        lines.append("fix 4 bottom freeze")
    return lines


def make_hip(bed, cfg):
    from sedifoam_amd import Lammps
    lmp = Lammps()
    lmp.set_box(bed["boxlo"], bed["boxhi"])
    lmp.create_atoms(bed["x"], bed["diameter"], bed["density"], v=bed["v"], omega=bed.get("omega"),
                     type_=bed.get("type"), tag=bed.get("tag"))
    for line in script_lines(bed, cfg):
        lmp.command(line)
    return lmp


def rel_err(a, b):
    """max |a-b| / max |b| (a global scale, so near-zero components do not blow up)"""
    scale = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (scale if scale > 0 else 1.0))


def rel_err_nan(a, b):
    """rel_err for outputs the reference itself leaves NaN in places (lubricate/poly: the log of a negative gap): the NaN
    pattern must be the reference's exactly, the finite entries are compared as usual"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if not np.array_equal(np.isnan(a), np.isnan(b)):
        return float("inf")
    ok = ~np.isnan(b)
    return rel_err(a[ok], b[ok])
