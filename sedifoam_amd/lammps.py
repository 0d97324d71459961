"""Host-side mirror of the patched LAMMPS library interface the reference's OpenFOAM side binds
(interfaceToLammps/library.h:29-63).  Same names, argument meaning and buffers; the work happens in
libsedifoam_amd.so on the GPU."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import SfError, check, dp, ip


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    if a is None:
        return None
    return a.ctypes.data_as(dp if a.dtype == np.float64 else ip)


class Lammps:
    """`lammps_open` ... `lammps_close` (library.h:29-63)."""

    def __init__(self, comm=0):
        self.L = _lib.lib()
        h = C.c_void_p()
        check(self.L.sf_lammps_open(0, None, comm, C.byref(h)))
        self.ptr = h

    def close(self):
        if self.ptr:
            check(self.L.sf_lammps_close(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- library.h:31-32 ----
    def command(self, line):
        err = self.L.sf_lammps_command(self.ptr, line.encode())
        if err is not None:
            raise SfError(err.decode())

    def commands(self, text):
        for line in text.splitlines():
            self.command(line)

    def file(self, path):
        check(self.L.sf_lammps_file(self.ptr, path.encode()))

    # ---- library.h:35-63 ----
    def get_global_n(self):
        return check(self.L.sf_lammps_get_global_n(self.ptr))

    def get_local_n(self):
        return check(self.L.sf_lammps_get_local_n(self.ptr))

    def get_initial_info(self):
        n = self.get_local_n()
        x = np.zeros((n, 3)); v = np.zeros((n, 3)); d = np.zeros(n); rho = np.zeros(n)
        tag = np.zeros(n, np.int32); cpu = np.zeros(n, np.int32); typ = np.zeros(n, np.int32)
        check(self.L.sf_lammps_get_initial_info(self.ptr, _p(x), _p(v), _p(d), _p(rho), _p(tag), _p(cpu), _p(typ)))
        return dict(x=x, v=v, diam=d, rho=rho, tag=tag, lmpCpuId=cpu, type=typ)

    def get_local_domain(self):
        d = np.zeros(6)
        check(self.L.sf_lammps_get_local_domain(self.ptr, _p(d)))
        return d

    def get_local_info(self):
        n = self.get_local_n()
        x = np.zeros((n, 3)); v = np.zeros((n, 3))
        foam = np.zeros(n, np.int32); cpu = np.zeros(n, np.int32); tag = np.zeros(n, np.int32)
        check(self.L.sf_lammps_get_local_info(self.ptr, _p(x), _p(v), _p(foam), _p(cpu), _p(tag)))
        return dict(x=x, v=v, foamCpuId=foam, lmpCpuId=cpu, tag=tag)

    def put_local_info(self, fdrag, tag, DuDt=None, foamCpuId=None):
        fdrag = _f64(fdrag).reshape(-1, 3)
        tag = _i32(tag)
        n = fdrag.shape[0]
        du = _f64(DuDt) if DuDt is not None else np.zeros((n, 3))
        cpu = _i32(foamCpuId) if foamCpuId is not None else np.zeros(n, np.int32)
        check(self.L.sf_lammps_put_local_info(self.ptr, n, _p(fdrag), _p(du), _p(cpu), _p(tag)))

    def step(self, n):
        check(self.L.sf_lammps_step(self.ptr, int(n)))

    def set_timestep(self, dt):
        check(self.L.sf_lammps_set_timestep(self.ptr, float(dt)))

    def get_timestep(self):
        return self.L.sf_lammps_get_timestep(self.ptr)

    def create_particle(self, position, tag, diameter, rho, type_, vel):
        position = _f64(position).reshape(-1, 3)
        t = _f64(tag)
        check(self.L.sf_lammps_create_particle(self.ptr, position.shape[0], _p(position), _p(t), diameter, rho,
                                               type_, _p(_f64(vel))))

    def delete_particle(self, tags):
        t = _i32(tags)
        check(self.L.sf_lammps_delete_particle(self.ptr, _p(t), t.shape[0]))

    # ---- engine extras (sf_dem_*) ----
    def create_atoms(self, x, diameter, density, v=None, omega=None, tag=None, type_=None):
        x = _f64(x).reshape(-1, 3)
        n = x.shape[0]
        check(self.L.sf_dem_create_atoms(
            self.ptr, n, _p(x), _p(_f64(v).reshape(-1, 3)) if v is not None else None,
            _p(_f64(omega).reshape(-1, 3)) if omega is not None else None, _p(_f64(diameter)),
            _p(_f64(density)), _p(_i32(tag)) if tag is not None else None,
            _p(_i32(type_)) if type_ is not None else None))

    def set_box(self, lo, hi):
        check(self.L.sf_dem_set_box(self.ptr, _p(_f64(lo)), _p(_f64(hi))))

    def setup(self):
        check(self.L.sf_dem_setup(self.ptr))

    def info(self):
        out = _lib.DemInfo()
        check(self.L.sf_dem_get_info(self.ptr, C.byref(out)))
        return out

    def device_view(self):
        out = _lib.DemDeviceView()
        check(self.L.sf_dem_device_view_get(self.ptr, C.byref(out)))
        return out

    def set_profiling(self, on=True):
        check(self.L.sf_dem_set_profiling(self.ptr, int(on)))

    def get_profile(self):
        """(launches, summed kernel milliseconds) of the fused sub-step kernel, from HIP events."""
        n = C.c_longlong(); ms = C.c_double()
        check(self.L.sf_dem_get_profile(self.ptr, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def get_rebuild_profile(self):
        """(neighbour rebuilds inside runs while profiling was on, their summed host-clock milliseconds)"""
        n = C.c_longlong(); ms = C.c_double()
        check(self.L.sf_dem_get_rebuild_profile(self.ptr, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def get_state(self):
        """x, v, omega, f, torque of the owned atoms sorted by tag."""
        n = self.get_local_n()
        li = self.get_local_info()
        f = np.zeros((n, 3)); t = np.zeros((n, 3)); w = np.zeros((n, 3)); tag = np.zeros(n, np.int32)
        check(self.L.sf_dem_get_forces(self.ptr, _p(f), _p(t), _p(w), _p(tag)))
        assert (tag == li["tag"]).all()
        o = np.argsort(tag, kind="stable")
        return dict(x=li["x"][o], v=li["v"][o], omega=w[o], f=f[o], torque=t[o], tag=tag[o])

    def history(self):
        """{(tag_i, tag_j): shear[3]} for touching pairs, tag_i < tag_j."""
        cap = max(int(self.info().npairs_full), 1)
        ti = np.zeros(cap, np.int32); tj = np.zeros(cap, np.int32); sh = np.zeros((cap, 3))
        n = check(self.L.sf_dem_get_history(self.ptr, cap, _p(ti), _p(tj), _p(sh)))
        return {(int(a), int(b)): s.copy() for a, b, s in zip(ti[:n], tj[:n], sh[:n])}

    def wall_shear(self, w):
        n = self.get_local_n()
        sh = np.zeros((n, 3))
        check(self.L.sf_dem_get_wall_shear(self.ptr, w, _p(sh)))
        tag = self.get_local_info()["tag"]
        return sh[np.argsort(tag, kind="stable")]
