"""ctypes binding of the CPU oracle (oracle/libsedifoam_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package sedifoam_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsedifoam_oracle.so")

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)


def build(force=False):
    """Compile the oracle with gcc (a few seconds)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


class GranParams(C.Structure):
    _fields_ = [("kn", C.c_double), ("kt", C.c_double), ("gamman", C.c_double),
                ("gammat", C.c_double), ("xmu", C.c_double), ("dampflag", C.c_int)]


class NeighList(C.Structure):
    _fields_ = [("inum", C.c_int), ("ilist", ip), ("first", ip), ("jlist", ip),
                ("touch", ip), ("shear", dp)]


class LubParams(C.Structure):
    _fields_ = [("mu", C.c_double), ("flaglog", C.c_int), ("flagfld", C.c_int),
                ("flagHI", C.c_int), ("flagVF", C.c_int), ("cut_inner", C.c_double),
                ("cut_global", C.c_double), ("R0", C.c_double), ("RT0", C.c_double),
                ("RS0", C.c_double), ("vxmu2f", C.c_double)]


class CloudFlags(C.Structure):
    _fields_ = [("particleDrag", C.c_int), ("particlePressureGrad", C.c_int),
                ("particleBuoyancy", C.c_int), ("particleAddedMass", C.c_int),
                ("particleLift", C.c_int), ("lubricationForce", C.c_int),
                ("gravity", C.c_double * 3), ("rhob", C.c_double), ("nub", C.c_double),
                ("deltaT", C.c_double)]


class Smooth(C.Structure):
    _fields_ = [("n", C.c_int * 3), ("dx", C.c_double * 3), ("D", C.c_double * 3), ("band", C.c_double),
                ("steps", C.c_int), ("UfSmooth", C.c_int), ("UpSmooth", C.c_int), ("dragSmooth", C.c_int),
                ("alphaSmooth", C.c_int), ("w", C.POINTER(C.c_double) * 3), ("periodic", C.c_int * 3)]


def _declare(L):
    L.orc_dem_create.restype = C.c_void_p
    L.orc_dem_create.argtypes = [C.c_int, dp, dp, dp, dp, dp, ip, dp, dp, ip]
    L.orc_dem_destroy.argtypes = [C.c_void_p]
    L.orc_dem_pair_gran.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_double,
                                    C.c_double, C.c_int, C.c_double, C.c_double, C.c_int]
    L.orc_dem_pair_lubricate.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_double,
                                         C.c_double, C.c_int, C.c_int]
    L.orc_dem_fix_cohesive.argtypes = [C.c_void_p] + [C.c_double] * 4 + [C.c_int]
    L.orc_dem_fix_gravity.argtypes = [C.c_void_p] + [C.c_double] * 4
    L.orc_dem_fix_fdrag.argtypes = [C.c_void_p, C.c_double]
    L.orc_dem_fix_freeze.argtypes = [C.c_void_p, C.c_int]
    L.orc_dem_fix_wall.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double,
                                   C.c_double, C.c_int, C.c_double, C.c_double, C.c_int,
                                   C.c_double, C.c_double, C.c_int]
    L.orc_dem_wall_cylinder.argtypes = [C.c_void_p, C.c_double]
    L.orc_dem_wall_motion.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double]
    L.orc_dem_set_mask.argtypes = [C.c_void_p, ip]
    L.orc_dem_set_groups.argtypes = [C.c_void_p] + [C.c_int] * 6
    L.orc_dem_neighbor.argtypes = [C.c_void_p, C.c_double]
    L.orc_dem_timestep.argtypes = [C.c_void_p, C.c_double]
    L.orc_dem_threads.argtypes = [C.c_void_p, C.c_int]
    L.orc_dem_setup.argtypes = [C.c_void_p]
    L.orc_dem_run.argtypes = [C.c_void_p, C.c_int]
    for name in ("orc_dem_nlocal", "orc_dem_nghost", "orc_dem_nbuilds"):
        getattr(L, name).argtypes = [C.c_void_p]
        getattr(L, name).restype = C.c_int
    L.orc_dem_npairs.argtypes = [C.c_void_p]
    L.orc_dem_npairs.restype = C.c_long
    L.orc_dem_get.argtypes = [C.c_void_p, dp, dp, dp, dp, dp, ip]
    L.orc_dem_put_fdrag.argtypes = [C.c_void_p, C.c_int, dp, ip]
    L.orc_dem_get_history.argtypes = [C.c_void_p, C.c_int, ip, ip, dp]
    L.orc_dem_get_history.restype = C.c_int
    L.orc_dem_get_wall_shear.argtypes = [C.c_void_p, C.c_int, dp]
    vp = C.c_void_p
    L.orc_dem_set_subdomain.argtypes = [vp, C.c_double, C.c_double]
    L.orc_dem_local_particle_volume.restype = C.c_double
    L.orc_dem_local_particle_volume.argtypes = [vp]
    L.orc_dem_set_global_particle_volume.argtypes = [vp, C.c_double]
    L.orc_dem_local_max_radius.restype = C.c_double
    L.orc_dem_local_max_radius.argtypes = [vp]
    L.orc_dem_set_global_max_radius.argtypes = [vp, C.c_double]
    for name in ("orc_dem_run_begin", "orc_dem_ext_setup", "orc_dem_rebuild_begin", "orc_dem_rebuild_sort",
                 "orc_dem_rebuild_finish", "orc_dem_ghost_forward_local"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = None
    L.orc_dem_substep.argtypes = [vp, C.c_int]
    for name in ("orc_dem_need_rebuild", "orc_dem_max_partners", "orc_dem_migrate_record_doubles"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = C.c_int
    L.orc_dem_migrate_set_slots.argtypes = [vp, C.c_int]
    L.orc_dem_migrate_pack.argtypes = [vp, C.c_int, C.c_double, vp, C.c_long]
    L.orc_dem_migrate_pack.restype = C.c_long
    L.orc_dem_migrate_unpack.argtypes = [vp, vp, C.c_long]
    L.orc_dem_border_pack.argtypes = [vp, C.c_int, C.c_double, vp, C.c_long]
    L.orc_dem_border_pack.restype = C.c_long
    L.orc_dem_border_unpack.argtypes = [vp, C.c_int, vp, C.c_long]
    L.orc_dem_forward_pack.argtypes = [vp, C.c_int, C.c_double, vp]
    L.orc_dem_forward_pack.restype = C.c_long
    L.orc_dem_forward_unpack.argtypes = [vp, C.c_int, vp, C.c_long]
    L.orc_dem_forward_unpack.restype = C.c_int
    L.orc_gran_settings.argtypes = [C.POINTER(GranParams), C.c_double, C.c_int, C.c_double,
                                    C.c_double, C.c_int, C.c_double, C.c_double, C.c_int,
                                    C.c_double]
    L.orc_gran_settings.restype = C.c_int
    pair_args = [C.POINTER(GranParams), C.c_double, C.c_int, C.c_int, dp, dp, dp, dp, dp, ip,
                 C.c_int, C.POINTER(NeighList), dp, dp]
    L.orc_pair_gran_hertzfix_history.argtypes = pair_args
    L.orc_pair_gran_hooke_history.argtypes = pair_args
    L.orc_pair_gran_hooke.argtypes = [C.POINTER(GranParams), C.c_int, dp, dp, dp, dp, dp, ip, C.c_int,
                                      C.POINTER(NeighList), dp, dp]
    L.orc_fix_cohesive.argtypes = [C.c_double] * 4 + [C.c_int, C.c_int, C.c_int, dp, dp, ip,
                                                      C.c_int, C.POINTER(NeighList), dp]
    L.orc_fix_cohesive.restype = C.c_int
    L.orc_lubricate_init.argtypes = [C.POINTER(LubParams), C.c_int, dp, C.c_double]
    L.orc_pair_lubricate_poly.argtypes = [C.POINTER(LubParams), C.c_int, dp, dp, dp, dp,
                                          C.POINTER(NeighList), dp, dp]
    L.orc_fix_fluid_drag.argtypes = [C.c_int, C.c_double, C.c_double, dp, dp, dp, ip, C.c_int,
                                     dp, dp, dp, dp]
    L.orc_fix_wall_gran.argtypes = [C.POINTER(GranParams), C.c_int, C.c_int, C.c_double,
                                    C.c_double, C.c_double, C.c_int, C.c_int, dp, dp, dp, dp, dp,
                                    ip, C.c_int, dp, dp, dp]
    L.orc_fix_wall_gran_moving.argtypes = [C.POINTER(GranParams), C.c_int, C.c_int, C.c_double, C.c_double,
                                           C.c_double, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                           C.c_double, C.c_long, C.c_double, C.c_int, C.c_int, dp, dp, dp, dp,
                                           dp, ip, C.c_int, dp, dp, dp]
    L.orc_nve_sphere_initial.argtypes = [C.c_int, C.c_double, dp, dp, dp, dp, dp, dp, dp]
    L.orc_nve_sphere_final.argtypes = [C.c_int, C.c_double, dp, dp, dp, dp, dp, dp]
    L.orc_fix_gravity.argtypes = [C.c_int, C.c_double, dp, dp, dp]
    L.orc_ergun_wenyu_jd.argtypes = [C.c_int, dp, dp, dp, C.c_double, C.c_double, dp]
    L.orc_syamlal_obrien_jd.argtypes = [C.c_int, dp, dp, dp, C.c_double, C.c_double, dp]
    L.orc_no_correction_jd.argtypes = [C.c_int, dp, dp, dp, C.c_double, C.c_double, dp]
    L.orc_cell_owner.argtypes = [C.c_int, dp, dp, dp, ip, ip]
    L.orc_cell_owner_graded.argtypes = [C.c_int, dp, dp, dp, ip, C.POINTER(dp), ip]
    L.orc_inlet_force_override.argtypes = [C.c_int, dp, dp, dp, C.c_double, C.c_int, dp, dp, dp, dp]
    L.orc_inlet_force_override.restype = None
    L.orc_drag_on_particles.argtypes = [C.POINTER(CloudFlags), C.c_int, C.c_int, ip] + [dp] * 14
    L.orc_drag_on_particles_hist.argtypes = ([C.POINTER(CloudFlags), C.c_int, C.c_int, ip] + [dp] * 9 + [C.c_int] +
                                             [dp] * 3 + [dp] * 5)
    L.orc_particle_to_eulerian.argtypes = [C.c_int, ip, dp, dp, C.c_int, dp, dp, dp]
    L.orc_calc_tc_fields.argtypes = [C.c_int, ip, dp, dp, dp, C.c_int, dp, dp, dp, dp, dp]
    L.orc_smooth_field.argtypes = [ip, dp, dp, C.c_double, C.c_int, C.c_int, dp]
    L.orc_smooth_field_graded.argtypes = [ip, dp, C.POINTER(dp), dp, C.c_double, C.c_int, C.c_int, dp]
    L.orc_smooth_field_periodic.argtypes = [ip, dp, dp, C.c_double, C.c_int, C.c_int, dp, ip]
    L.orc_smooth_field_graded_periodic.argtypes = [ip, dp, C.POINTER(dp), dp, C.c_double, C.c_int, C.c_int, dp, ip]
    L.orc_particle_to_eulerian_smooth.argtypes = [C.c_int, ip, dp, dp, C.c_int, dp, C.POINTER(Smooth), dp, dp]
    L.orc_uf_smoothed.argtypes = [C.c_int, dp, dp, C.POINTER(Smooth), dp]
    L.orc_calc_tc_fields_smooth.argtypes = [C.c_int, ip, dp, dp, dp, C.c_int, dp, dp, dp, C.POINTER(Smooth), dp, dp]
    L.orc_adjust_timestep.argtypes = [C.c_double, C.c_double, C.c_int, dp, ip, ip, ip]
    L.orc_adjust_timestep.restype = C.c_int


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def P(a):
    """pointer to a contiguous numpy array (float64 or int32), or NULL."""
    if a is None:
        return None
    if a.dtype == np.float64:
        return a.ctypes.data_as(dp)
    if a.dtype == np.int32:
        return a.ctypes.data_as(ip)
    raise TypeError(a.dtype)


class OracleDem:
    """The oracle's DEM driver = what `lammps_step(n)` does, on the CPU."""

    def __init__(self, x, radius, rmass, boxlo, boxhi, periodic=(0, 0, 0), v=None, omega=None,
                 tag=None):
        L = lib()
        self.L = L
        x = f64(x).reshape(-1, 3)
        n = x.shape[0]
        self.n = n
        v = f64(v).reshape(-1, 3) if v is not None else np.zeros((n, 3))
        omega = f64(omega).reshape(-1, 3) if omega is not None else np.zeros((n, 3))
        tag = i32(tag) if tag is not None else np.arange(1, n + 1, dtype=np.int32)
        self.h = L.orc_dem_create(n, P(x), P(v), P(omega), P(f64(radius)), P(f64(rmass)), P(tag),
                                  P(f64(boxlo)), P(f64(boxhi)), P(i32(periodic)))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_dem_destroy(self.h)
            self.h = None

    def pair_gran(self, style, kn, kt, gamman, gammat, xmu, dampflag):
        """style 'hooke' (gran/hooke/history) | 'hertz' (gran/hertzFix/history) | 'hooke_plain' (gran/hooke, no
        history); kt/gammat None = NULL."""
        st = {"hooke": 1, "hertz": 2, "hooke_plain": 3, None: 0}[style]
        rc = self.L.orc_dem_pair_gran(self.h, st, kn, kt is None, kt or 0.0, gamman,
                                      gammat is None, gammat or 0.0, xmu, dampflag)
        if rc:
            raise ValueError("Illegal pair_style command")

    def pair_lubricate(self, mu, flaglog, flagfld, cut_inner, cut_global, flagHI=1, flagVF=1):
        self.L.orc_dem_pair_lubricate(self.h, mu, flaglog, flagfld, cut_inner, cut_global,
                                      flagHI, flagVF)

    def fix_cohesive(self, ah, lam, smin, smax, opt):
        self.L.orc_dem_fix_cohesive(self.h, ah, lam, smin, smax, opt)

    def fix_gravity(self, mag, gx, gy, gz):
        self.L.orc_dem_fix_gravity(self.h, mag, gx, gy, gz)

    def fix_fdrag(self, carrier_rho=0.0):
        self.L.orc_dem_fix_fdrag(self.h, carrier_rho)

    def fix_freeze(self, groupbit):
        """fix freeze at this place of the fix list (the fixes registered later still act on the frozen atoms)"""
        self.L.orc_dem_fix_freeze(self.h, int(groupbit))

    def fix_wall(self, dim, lo, hi, kn, kt, gamman, gammat, xmu, dampflag):
        self.L.orc_dem_fix_wall(self.h, dim, lo is None, lo or 0.0, hi is None, hi or 0.0, kn,
                                kt is None, kt or 0.0, gamman, gammat is None, gammat or 0.0,
                                xmu, dampflag)

    def wall_cylinder(self, radius):
        """the wall registered last becomes `zcylinder radius`"""
        self.L.orc_dem_wall_cylinder(self.h, float(radius))

    def wall_motion(self, kind, axis, a, b=0.0):
        """the wall registered last: kind "wiggle" (axis, amplitude, period) or "shear" (axis, vshear)"""
        self.L.orc_dem_wall_motion(self.h, 1 if kind == "wiggle" else 2, int(axis), float(a), float(b))

    def set_mask(self, mask):
        """per-atom group bits in creation order (bit 0 = all is always set)"""
        self.L.orc_dem_set_mask(self.h, P(i32(mask)))

    def set_groups(self, nve=1, gravity=1, fdrag=1, wall=1, cohesive=1, freeze=0):
        """group bit of every registered fix kind (call after the fix_* registrations); freeze = 0: no fix freeze"""
        self.L.orc_dem_set_groups(self.h, nve, gravity, fdrag, wall, cohesive, freeze)

    def neighbor(self, skin):
        self.L.orc_dem_neighbor(self.h, skin)

    def timestep(self, dt):
        self.L.orc_dem_timestep(self.h, dt)

    def setup(self):
        self.L.orc_dem_setup(self.h)

    def run(self, n):
        self.L.orc_dem_run(self.h, n)

    @property
    def nlocal(self):
        return self.L.orc_dem_nlocal(self.h)

    @property
    def nghost(self):
        return self.L.orc_dem_nghost(self.h)

    @property
    def nbuilds(self):
        return self.L.orc_dem_nbuilds(self.h)

    @property
    def npairs(self):
        return self.L.orc_dem_npairs(self.h)

    def get(self):
        """dict of local arrays sorted by tag."""
        n = self.nlocal
        out = {k: np.zeros((n, 3)) for k in ("x", "v", "omega", "f", "torque")}
        tag = np.zeros(n, dtype=np.int32)
        self.L.orc_dem_get(self.h, P(out["x"]), P(out["v"]), P(out["omega"]), P(out["f"]),
                           P(out["torque"]), P(tag))
        order = np.argsort(tag, kind="stable")
        res = {k: a[order] for k, a in out.items()}
        res["tag"] = tag[order]
        return res

    def put_fdrag(self, fdrag, tag):
        fdrag = f64(fdrag).reshape(-1, 3)
        self.L.orc_dem_put_fdrag(self.h, fdrag.shape[0], P(fdrag), P(i32(tag)))

    def history(self):
        """{(tag_i, tag_j): shear[3]} with tag_i < tag_j orientation-normalised (sign kept for i->j)."""
        cap = max(self.npairs, 1)
        ti = np.zeros(cap, dtype=np.int32)
        tj = np.zeros(cap, dtype=np.int32)
        sh = np.zeros((cap, 3))
        n = self.L.orc_dem_get_history(self.h, cap, P(ti), P(tj), P(sh))
        out = {}
        for a, b, s in zip(ti[:n], tj[:n], sh[:n]):
            if a < b:
                out[(int(a), int(b))] = s.copy()
            else:
                out[(int(b), int(a))] = -s
        return out

    def wall_shear(self, w):
        n = self.nlocal
        sh = np.zeros((n, 3))
        tag = np.zeros(n, dtype=np.int32)
        self.L.orc_dem_get_wall_shear(self.h, w, P(sh))
        self.L.orc_dem_get(self.h, None, None, None, None, None, P(tag))
        return sh[np.argsort(tag, kind="stable")]


class OracleSlabEngine:
    """Adaptor with the interface sedifoam_amd.halo.SlabDriver expects, backed by the CPU oracle and torch CPU
    tensors -- the stand-in for HipSlabEngine in the gloo (no-GPU) tests of the multi-rank protocol."""

    class _Info:
        pass

    def __init__(self, dem):
        import torch
        self.torch = torch
        self.dem = dem
        self.L = dem.L
        self.h = dem.h
        self.device = torch.device("cpu")
        self._trigger = torch.full((1,), 2 ** 31 - 1, dtype=torch.int32)

    def alloc(self, ndoubles):
        return self.torch.zeros(max(int(ndoubles), 1), dtype=self.torch.float64)

    def info(self):
        i = self._Info()
        i.nlocal = self.dem.nlocal
        i.nghost = self.dem.nghost
        i.max_neigh_used = self.L.orc_dem_max_partners(self.h)
        return i

    def set_subdomain(self, rank, world, lo, hi):
        self.L.orc_dem_set_subdomain(self.h, lo, hi)

    def local_particle_volume(self):
        return self.L.orc_dem_local_particle_volume(self.h)

    def set_global_particle_volume(self, v):
        self.L.orc_dem_set_global_particle_volume(self.h, float(v))

    def local_max_radius(self):
        return self.L.orc_dem_local_max_radius(self.h)

    def set_global_max_radius(self, r):
        self.L.orc_dem_set_global_max_radius(self.h, float(r))

    def setup(self):
        self.L.orc_dem_ext_setup(self.h)

    def run_begin(self):
        self._trigger[0] = 2 ** 31 - 1
        self.L.orc_dem_run_begin(self.h)
        if self.L.orc_dem_need_rebuild(self.h):
            self._trigger[0] = -1

    def substep(self, last):
        self.L.orc_dem_substep(self.h, int(last))

    def need_rebuild(self):
        return self.L.orc_dem_need_rebuild(self.h)

    # batch mode of the driver: the oracle emulates the device-resident trigger word with a CPU tensor
    @property
    def trigger(self):
        return self._trigger

    def substep_k(self, last, kstep):
        # a sub-step behind the (all-reduced) trigger exits early, like the HIP kernel
        if int(self._trigger[0]) < kstep:
            return
        self.L.orc_dem_substep(self.h, int(last))
        if not last and self.L.orc_dem_need_rebuild(self.h):
            self._trigger[0] = min(int(self._trigger[0]), kstep)

    def batch_end(self, first_k, launched):
        return int(self._trigger[0])

    def rebuild_begin(self):
        self.L.orc_dem_rebuild_begin(self.h)

    def rebuild_sort(self):
        self.L.orc_dem_rebuild_sort(self.h)

    def rebuild_finish(self):
        self.L.orc_dem_rebuild_finish(self.h)
        self._trigger[0] = 2 ** 31 - 1

    def migrate_set_slots(self, m):
        self.L.orc_dem_migrate_set_slots(self.h, int(m))

    def migrate_record_doubles(self):
        return self.L.orc_dem_migrate_record_doubles(self.h)

    def migrate_pack(self, side, xshift, buf):
        n = self.L.orc_dem_migrate_pack(self.h, side, xshift, buf.data_ptr(), buf.numel())
        if n < 0:
            raise RuntimeError("migrate buffer too small")
        return n

    def migrate_unpack(self, buf, ndoubles):
        self.L.orc_dem_migrate_unpack(self.h, buf.data_ptr(), int(ndoubles))

    def border_pack(self, side, xshift, buf):
        n = self.L.orc_dem_border_pack(self.h, side, xshift, buf.data_ptr(), buf.numel() // 14)
        if n < 0:
            raise RuntimeError("border buffer too small")
        return n

    def border_unpack(self, side, buf, natoms):
        self.L.orc_dem_border_unpack(self.h, side, buf.data_ptr(), int(natoms))

    def forward_pack(self, side, xshift, buf):
        return self.L.orc_dem_forward_pack(self.h, side, xshift, buf.data_ptr())

    def forward_unpack(self, side, buf, natoms):
        if self.L.orc_dem_forward_unpack(self.h, side, buf.data_ptr(), int(natoms)) != 0:
            raise RuntimeError("forward_unpack: ghost count mismatch")

    def ghost_forward_local(self):
        self.L.orc_dem_ghost_forward_local(self.h)

    # overlapped schedule of the driver, without streams: the boundary part is empty, the interior part is the
    # whole sub-step (the CPU engine has no reason to split), the vote is the one the fused exchange carries
    def enable_overlap(self):
        self.comm_stream = None

    def overlap_begin(self):
        self._trigger[0] = 2 ** 31 - 1

    def substep_part(self, part, last, kstep):
        if part == 1:
            self.substep_k(last, kstep)

    def substep_flip(self, kstep):
        pass

    def overlap_batch_end(self, first_k, launched, last_kstep):
        return int(self._trigger[0])

    # the one-collective forward halo of the driver (header = rebuild trigger, then the records), on CPU tensors
    def index_table(self, values):
        return self.torch.tensor(list(values), dtype=self.torch.int32)

    def forward_pack_fused(self, shift0, off0, shift1, off1, hdr_off, sendbuf):
        sendbuf[hdr_off.long()] = float(int(self._trigger[0]))
        self.forward_pack(0, shift0, sendbuf[int(off0):])
        self.forward_pack(1, shift1, sendbuf[int(off1):])

    def forward_unpack_fused(self, recvbuf, off_l, n_l, off_r, n_r, hdr_off, kstep=-1):
        self._trigger[0] = min(int(self._trigger[0]), int(recvbuf[hdr_off.long()].min().item()))
        self.forward_unpack(0, recvbuf[int(off_l):], n_l)
        self.forward_unpack(1, recvbuf[int(off_r):], n_r)
        self.ghost_forward_local()
