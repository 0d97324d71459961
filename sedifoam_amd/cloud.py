"""Host-side mirror of the OpenFOAM plug-in surfaces of the hot path:

  dragModel      lammpsFoam/dragModels/dragModel/dragModel.H:55-138, newDragModel.C:31-64
                 (run-time selection by the `dragModel` keyword of constant/cloudProperties)
  enhancedCloud  lammpsFoam/enhancedCloud.H:183-249 (evolve, calcTcFields, Omega, Asrc, ...)

Everything numerical happens in libsedifoam_amd.so on the GPU; this file only marshals."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import SfError, check, dp, ip

_DRAG_TABLE = {"ErgunWenYu": 0, "SyamlalOBrien": 1, "NoCorrection": 2}


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    if a is None:
        return None
    return a.ctypes.data_as(dp if a.dtype == np.float64 else ip)


class dragModel:
    """dragModel::New(cloudDict, transDict, alpha, pd) + Jd(Ur) on device arrays (torch CUDA tensors)."""

    def __init__(self, name, nub, rhob):
        if name not in _DRAG_TABLE:
            # newDragModel.C:45-55: FatalError listing the selection table
            raise SfError("Unknown dragModel type %s\n\nValid dragModel types are :\n%s"
                          % (name, sorted(_DRAG_TABLE)))
        self.name = name
        self.model = _DRAG_TABLE[name]
        self.nuf = float(nub)
        self.rhof = float(rhob)

    @classmethod
    def New(cls, cloudDict, transDict):
        return cls(cloudDict["dragModel"], transDict["nub"], transDict["rhob"])

    def Jd(self, Ur, alpha, pd):
        """Ur, alpha, pd: 1-D float64 CUDA tensors of equal size; returns a new tensor (tmp<scalarField>)."""
        import torch
        if not (Ur.numel() == alpha.numel() == pd.numel()):
            raise SfError("%s::Jd() Inconsistent Ur/Alpha/pd. Ur size: %d Alpha size: %d pd size: %d"
                          % (self.name, Ur.numel(), alpha.numel(), pd.numel()))
        out = torch.empty_like(Ur)
        L = _lib.lib()
        check(L.sfk_drag_model_jd(self.model, Ur.numel(), Ur.data_ptr(), alpha.data_ptr(), pd.data_ptr(),
                                  self.nuf, self.rhof, out.data_ptr(), torch.cuda.current_stream().cuda_stream))
        return out


class enhancedCloud:
    """enhancedCloud(U, p, Ue, Uf, DDtUf, nu, alpha, cloudDict, transDict, ...) on a uniform hex block.

    cloudDict keys (constant/cloudProperties): dragModel, subCycles, particleDrag, particlePressureGrad,
    particleBuoyancy, particleAddedMass, particleLift, particleHistoryForce, lubricationForce, g, maxPossibleAlpha, diffusionBandWidth,
    diffusionSteps, UfSmooth, UpSmooth, dragSmooth, alphaSmooth, smoothDirection.
    transDict keys (constant/transportProperties): rhob, nub."""

    def __init__(self, lammps, mesh_origin, mesh_dx, mesh_n, cloudDict, transDict, deltaT, driver=None,
                 mesh_faces=None, mesh_labels=None, mesh_periodic=None):
        """mesh_faces: (xf, yf, zf) face coordinates of a graded (blockMesh simpleGrading) block, n+1 ascending values
        per axis or None for a uniform axis (then mesh_origin / mesh_dx apply along it).
        mesh_labels: OpenFOAM cell label of every cell of the grid, in grid order ix + nx*(iy + ny*iz) (multi-block
        blockMesh cases number their cells block by block); every field array is then in label order.
        mesh_periodic: (px, py, pz) cyclic patch pairs of the (diffusion) mesh: smoothField couples the first and the
        last cell layer along such an axis instead of closing them with zeroGradient (the reference's channel cases).
        driver: a sedifoam_amd.halo.SlabDriver when the particles are decomposed over several GPUs (lammps is
        then the driver's engine).  Every rank holds the whole mesh; the per-cell sums of gamma, Ue and Asrc are
        added over the ranks (torch.distributed all_reduce on the device arrays) inside evolve()/calcTcFields()."""
        self.L = _lib.lib()
        self.lmp = lammps
        self.driver = driver
        name = cloudDict.get("dragModel", "ErgunWenYu")
        if name not in _DRAG_TABLE:
            raise SfError("Unknown dragModel type %s\n\nValid dragModel types are :\n%s"
                          % (name, sorted(_DRAG_TABLE)))
        pr = _lib.CloudProps()
        pr.dragModel = _DRAG_TABLE[name]
        pr.subCycles = int(cloudDict.get("subCycles", 1))
        pr.particleDrag = int(cloudDict.get("particleDrag", True))                    # enhancedCloud.C:586
        pr.particlePressureGrad = int(cloudDict.get("particlePressureGrad", True))    # :587
        pr.particleBuoyancy = int(cloudDict.get("particleBuoyancy", False))           # :589
        pr.particleAddedMass = int(cloudDict.get("particleAddedMass", False))         # :591
        pr.particleLift = int(cloudDict.get("particleLift", False))                   # :593
        pr.lubricationForce = int(cloudDict.get("lubricationForce", False))           # :597
        pr.particleHistoryForce = int(cloudDict.get("particleHistoryForce", False))   # :595-596
        g = cloudDict.get("g", (0.0, 0.0, 0.0))
        pr.gravity = (C.c_double * 3)(*g)
        pr.rhob = float(transDict["rhob"])
        pr.nub = float(transDict["nub"])
        pr.maxPossibleAlpha = float(cloudDict.get("maxPossibleAlpha", 0.0))
        pr.diffusionBandWidth = float(cloudDict.get("diffusionBandWidth", 0.0))
        pr.diffusionSteps = int(cloudDict.get("diffusionSteps", 0))
        pr.UfSmooth = int(cloudDict.get("UfSmooth", True))          # enhancedCloud.C:573-576
        pr.UpSmooth = int(cloudDict.get("UpSmooth", True))
        pr.dragSmooth = int(cloudDict.get("dragSmooth", True))
        pr.alphaSmooth = int(cloudDict.get("alphaSmooth", True))
        sd = cloudDict.get("smoothDirection", (1.0, 0, 0, 0, 1.0, 0, 0, 0, 1.0))   # :578-583 (tensor)
        if len(sd) == 9:
            if any(abs(sd[k]) > 0 for k in (1, 2, 3, 5, 6, 7)):
                raise SfError("smoothDirection: only diagonal tensors are supported")
            sd = (sd[0], sd[4], sd[8])
        pr.smoothDirection = (C.c_double * 3)(*sd)
        m = _lib.CloudMesh()
        m.origin = (C.c_double * 3)(*mesh_origin)
        m.dx = (C.c_double * 3)(*mesh_dx)
        m.n = (C.c_int * 3)(*mesh_n)
        m.periodic = (C.c_int * 3)(*[int(bool(q)) for q in ((0, 0, 0) if mesh_periodic is None else mesh_periodic)])
        keep_faces = []
        for k in range(3):
            fk = None if mesh_faces is None else mesh_faces[k]
            if fk is None:
                m.faces[k] = C.POINTER(C.c_double)()
            else:
                a = np.ascontiguousarray(fk, dtype=np.float64)
                if a.shape != (int(mesh_n[k]) + 1,):
                    raise SfError("mesh_faces[%d] must hold n + 1 = %d coordinates" % (k, int(mesh_n[k]) + 1))
                keep_faces.append(a)
                m.faces[k] = a.ctypes.data_as(C.POINTER(C.c_double))
        self.ncells = int(np.prod(mesh_n))
        if mesh_labels is not None:
            lab = np.ascontiguousarray(mesh_labels, dtype=np.int32)
            if lab.shape != (self.ncells,):
                raise SfError("mesh_labels must hold one label per cell")
            keep_faces.append(lab)
            m.cell_label = lab.ctypes.data_as(C.POINTER(C.c_int))
        h = C.c_void_p()
        if driver is not None and not driver.is_setup:
            driver.setup()
        check(self.L.sf_cloud_create(lammps.ptr, C.byref(m), C.byref(pr), float(deltaT), C.byref(h)))
        self.ptr = h
        if driver is not None:
            sc = C.c_int(); ss = C.c_int()
            check(self.L.sf_cloud_sub_cycling(self.ptr, C.byref(sc), C.byref(ss)))
            self._sub = (sc.value, ss.value)
            g = _lib.dp(); u = _lib.dp(); a = _lib.dp(); nc = C.c_int()
            check(self.L.sf_cloud_device_fields(self.ptr, C.byref(g), C.byref(u), C.byref(a), C.byref(nc)))
            self._dev = dict(gamma=(C.cast(g, C.c_void_p).value, nc.value),
                             Ue=(C.cast(u, C.c_void_p).value, 3 * nc.value),
                             Asrc=(C.cast(a, C.c_void_p).value, 3 * nc.value))
            # the constructor scattered this rank's particles only: redo it over all ranks
            self._phase(2); self._sum_over_ranks("gamma", "Ue"); self._phase(3); self._phase(6)

    def close(self):
        if getattr(self, "ptr", None):
            check(self.L.sf_cloud_destroy(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def setFluid(self, Uf=None, DDtUf=None, gradp=None, curlU=None):
        arrs = [None if a is None else _f64(a).reshape(self.ncells, 3) for a in (Uf, DDtUf, gradp, curlU)]
        check(self.L.sf_cloud_set_fluid(self.ptr, *[_p(a) for a in arrs]))

    def smoothField(self, field):
        """enhancedCloud::smoothField on a host array [ncells] or [ncells][3]; returns the smoothed copy."""
        f = np.array(field, dtype=np.float64, order="C", copy=True)
        ncomp = 1 if f.ndim == 1 else f.shape[1]
        check(self.L.sf_cloud_smooth_field(self.ptr, _p(f.reshape(-1)), ncomp))
        return f

    def evolve(self):
        if self.driver is None:
            check(self.L.sf_cloud_evolve(self.ptr))
            return
        # enhancedCloud::evolve (enhancedCloud.C:669-787) on a decomposed domain
        self._phase(0)
        sub_cycles, sub_steps = self._sub
        for k in range(sub_cycles):
            self._phase(1)
            self.driver.step(sub_steps)
            if k == 0:
                self._phase(2)
                self._sum_over_ranks("gamma", "Ue")
                self._phase(3)

    def calcTcFields(self):
        if self.driver is None:
            check(self.L.sf_cloud_calc_tc_fields(self.ptr))
            return
        self._phase(4)
        self._sum_over_ranks("Asrc")
        self._phase(5)

    def _phase(self, ph):
        check(self.L.sf_cloud_phase(self.ptr, int(ph)))

    def _sum_over_ranks(self, *names):
        """in-place SUM over the ranks of the named device arrays (views through __cuda_array_interface__)"""
        d = self.driver
        if d.world == 1 and not d.self_comm:
            return
        import torch

        class _View:
            def __init__(self, ptr, n):
                self.__cuda_array_interface__ = dict(shape=(n,), typestr="<f8", data=(ptr, False), version=2)
        for nm in names:
            ptr, n = self._dev[nm]
            t = torch.as_tensor(_View(ptr, n), device=d.e.device)
            if d.transport == "host":
                h = t.cpu()
                d.dist.all_reduce(h)
                t.copy_(h)
            else:
                d.dist.all_reduce(t)

    def _fields(self):
        g = np.zeros(self.ncells); ue = np.zeros((self.ncells, 3))
        a = np.zeros((self.ncells, 3)); om = np.zeros(self.ncells)
        check(self.L.sf_cloud_get_fields(self.ptr, _p(g), _p(ue), _p(a), _p(om)))
        return g, ue, a, om

    def gamma(self):
        return self._fields()[0]

    def Ue(self):
        return self._fields()[1]

    def Asrc(self):
        return self._fields()[2]

    def Omega(self):
        return self._fields()[3]

    def averageInfo(self):
        """enhancedCloud::averageInfo: dict(totalVolume, totalVel[3], averageVel[3])"""
        out = np.zeros(7)
        check(self.L.sf_cloud_average_info(self.ptr, _p(out)))
        return dict(totalVolume=out[0], totalVel=out[1:4].copy(), averageVel=out[4:7].copy())

    def particleCount(self):
        return check(self.L.sf_cloud_particle_count(self.ptr))

    def particles(self):
        n = self.particleCount()
        tag = np.zeros(n, np.int32); cell = np.zeros(n, np.int32)
        pd = np.zeros((n, 3)); jd = np.zeros(n)
        check(self.L.sf_cloud_get_particles(self.ptr, _p(tag), _p(cell), _p(pd), _p(jd)))
        return dict(tag=tag, cell=cell, pDrag=pd, Jd=jd)

    def cpuTimeSplit(self):
        t = _lib.CloudTimers()
        check(self.L.sf_cloud_get_timers(self.ptr, C.byref(t)))
        return {k: getattr(t, k) for k, _ in t._fields_}


def adjustLampTimestep(deltaT, dtLampIn, subCycles):
    """softParticleCloud::adjustLampTimestep (softParticleCloud.C:209-261)."""
    L = _lib.lib()
    dt = C.c_double(); steps = C.c_int(); sc = C.c_int(); ss = C.c_int()
    rc = L.sf_cloud_adjust_timestep(deltaT, dtLampIn, subCycles, C.byref(dt), C.byref(steps), C.byref(sc),
                                    C.byref(ss))
    if rc != 0:
        raise SfError("softParticleCloud::adjustLampTimestep() Time step adjustment error.")
    return dict(dtLampAdj=dt.value, solidStepsPerDt=steps.value, subCycles=sc.value, subSteps=ss.value)
