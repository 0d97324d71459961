"""Host-side mirror of the OpenFOAM plug-in surfaces of the hot path:

  dragModel      lammpsFoam/dragModels/dragModel/dragModel.H:55-138, newDragModel.C:31-64
                 (run-time selection by the `dragModel` keyword of constant/cloudProperties)
  enhancedCloud  lammpsFoam/enhancedCloud.H:183-249 (evolve, calcTcFields, Omega, Asrc, ...)

Everything numerical happens in libsedifoam_amd.so on the GPU; this file only marshals."""
import ctypes as C

import os

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
    particleBuoyancy, particleAddedMass, particleLift, particleHistoryForce, lubricationForce, addParticleOption +
    inletForce + inletBox + eccentricity (the inlet override only), g, maxPossibleAlpha, diffusionBandWidth,
    diffusionSteps, UfSmooth, UpSmooth, dragSmooth, alphaSmooth, smoothDirection.
    transDict keys (constant/transportProperties): rhob, nub."""

    def __init__(self, lammps, mesh_origin, mesh_dx, mesh_n, cloudDict, transDict, deltaT, driver=None,
                 mesh_faces=None, mesh_labels=None, mesh_periodic=None, mesh_partition=False):
        """mesh_faces: (xf, yf, zf) face coordinates of a graded (blockMesh simpleGrading) block, n+1 ascending values
        per axis or None for a uniform axis (then mesh_origin / mesh_dx apply along it).
        mesh_labels: OpenFOAM cell label of every cell of the grid, in grid order ix + nx*(iy + ny*iz) (multi-block
        blockMesh cases number their cells block by block); every field array is then in label order.
        mesh_periodic: (px, py, pz) cyclic patch pairs of the (diffusion) mesh: smoothField couples the first and the
        last cell layer along such an axis instead of closing them with zeroGradient (the reference's channel cases).
        mesh_partition (with driver): the mesh is PARTITIONED by the slab planes of the particle decomposition instead
        of replicated: this rank keeps its nx/world cell layers plus a ghost layer on each side, scatters locally, moves
        the ghost-layer sums to its face neighbours and takes part in a distributed smoothing solve (see _slab_*);
        no collective is as large as the mesh.  mesh_origin / mesh_dx / mesh_n still describe the WHOLE mesh, setFluid
        takes whole-mesh arrays (it keeps this rank's part) and gamma() / Ue() / Asrc() return this rank's owned cells.
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
        # the inlet override of updateDragOnParticles (:249-257): addParticleOption 1 | 2, inletForce, inletBox (the
        # tensor's nine numbers), eccentricity -- softParticleCloud.C:460-486; the add / delete schedules are not built
        pr.addParticleOption = int(cloudDict.get("addParticleOption", 0))
        pr.inletForce = (C.c_double * 3)(*cloudDict.get("inletForce", (0.0, 0.0, 0.0)))
        pr.inletBox = (C.c_double * 9)(*cloudDict.get("inletBox", (0.0,) * 9))
        pr.eccentricity = (C.c_double * 3)(*cloudDict.get("eccentricity", (0.0, 0.0, 0.0)))
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
        self.partition = bool(mesh_partition) and driver is not None
        self.mesh_n_global = tuple(int(k) for k in mesh_n)
        if self.partition:
            W, r = driver.world, driver.rank
            if hasattr(driver, "grid"):
                raise SfError("mesh_partition: the mesh is cut by x-slab planes; over a brick decomposition "
                              "(BrickDriver) keep the whole mesh on every rank")
            if mesh_faces is not None and mesh_faces[0] is not None:
                raise SfError("mesh_partition: the mesh must be uniform along x")
            if mesh_labels is not None:
                raise SfError("mesh_partition: cell labels are not supported")
            if int(mesh_n[0]) % W:
                raise SfError("mesh_partition: %d cell layers along x do not divide into %d slabs" % (mesh_n[0], W))
            self.nxl = int(mesh_n[0]) // W
            x0 = float(mesh_origin[0]) + r * self.nxl * float(mesh_dx[0])
            if abs(x0 - driver.sublo) > 1e-9 * abs(float(mesh_dx[0])):
                raise SfError("mesh_partition: the slab planes of the mesh (%g) and of the particles (%g) differ"
                              % (x0, driver.sublo))
            mesh_origin = (x0 - float(mesh_dx[0]), float(mesh_origin[1]), float(mesh_origin[2]))
            mesh_n = (self.nxl + 2, int(mesh_n[1]), int(mesh_n[2]))
        m = _lib.CloudMesh()
        m.origin = (C.c_double * 3)(*mesh_origin)
        m.dx = (C.c_double * 3)(*mesh_dx)
        m.n = (C.c_int * 3)(*mesh_n)
        m.slab_nx_global = self.mesh_n_global[0] if self.partition else 0
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
            if self.partition:
                self._slab_init(mesh_periodic)
                # the constructor scattered this rank's particles into its slab + ghost layers
                self._slab_halo_add("gamma", "Ue"); self._slab_phase(3); self._slab_phase(6)
            else:
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
        if self.partition:
            arrs = [None if a is None else self._slab_take(_f64(a).reshape(-1, 3)) for a in (Uf, DDtUf, gradp, curlU)]
            check(self.L.sf_cloud_set_fluid(self.ptr, *[_p(a) for a in arrs]))
            if Uf is not None and not self._slab_started:
                self._slab_phase(6)          # UfSmoothed of the initial condition (enhancedCloud.C:641-655)
            return
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
        if self.partition:
            self._slab_started = True
            self._slab_phase(0)
            sub_cycles, sub_steps = self._sub
            for k in range(sub_cycles):
                self._phase(1)
                self.driver.step(sub_steps)
                if k == 0:
                    self._phase(2)
                    self._slab_halo_add("gamma", "Ue")
                    self._slab_phase(3)
            return
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
        if self.partition:
            self._phase(4)
            self._slab_halo_add("Asrc")
            self._slab_phase(5)
            return
        self._phase(4)
        self._sum_over_ranks("Asrc")
        self._phase(5)

    def _phase(self, ph):
        return check(self.L.sf_cloud_phase(self.ptr, int(ph)))

    # ---- mesh partitioned into x-slabs (mesh_partition=True): everything below moves data of LOCAL size only ----
    def _slab_init(self, mesh_periodic):
        import torch
        d = self.driver
        self._slab_started = False
        self._torch = torch
        self._per_x = bool(mesh_periodic[0]) if mesh_periodic is not None else False
        W, r = d.world, d.rank
        self._left = r - 1 if r > 0 else (W - 1 if self._per_x else None)
        self._right = r + 1 if r < W - 1 else (0 if self._per_x else None)
        nxg, ny, nz = self.mesh_n_global
        self._shape = (nz, ny, self.nxl + 2)
        # the exchanges in C++ over the engine's own RCCL communicator (sf_cloud_slab_halo_add / sf_cloud_slab_phase)
        # whenever the C++ slab driver is up; the torch.distributed versions below serve the gloo / host transports
        self._cxx_slab = bool(getattr(d, "_cxx", False)) and d.transport == "rccl" and not hasattr(d, "grid") \
            and os.environ.get("SF_CLOUD_SLAB_PY", "0") != "1"

    def _dev_tensor(self, ptr, shape):
        torch = self._torch
        n = int(np.prod(shape))

        class _View:
            def __init__(self, p, k):
                self.__cuda_array_interface__ = dict(shape=(k,), typestr="<f8", data=(p, False), version=2)
        return torch.as_tensor(_View(ptr, n), device=self.driver.e.device).view(*shape)

    def _field(self, name):
        ptr, n = self._dev[name]
        ncomp = n // int(np.prod(self._shape))
        return self._dev_tensor(ptr, self._shape + ((ncomp,) if ncomp > 1 else ()))

    def _slab_take(self, a):
        """this rank's cells (ghost layers included, wrapped or clamped at the box ends) of a whole-mesh array"""
        nxg, ny, nz = self.mesh_n_global
        g = a.reshape(nz, ny, nxg, -1)
        r = self.driver.rank
        ix = np.arange(r * self.nxl - 1, (r + 1) * self.nxl + 1)
        ix = np.mod(ix, nxg) if self._per_x else np.clip(ix, 0, nxg - 1)
        return np.ascontiguousarray(g[:, :, ix, :]).reshape(-1, a.shape[1])

    def _sendrecv(self, to_left, to_right):
        """two face messages: returns (from_left, from_right); None where there is no neighbour"""
        torch, d = self._torch, self.driver
        dist = d.dist
        if d.world == 1:
            return (to_right if self._left is not None else None, to_left if self._right is not None else None)
        host = d.transport == "host"
        dev = "cpu" if host else d.e.device
        sl, sr = to_left.to(dev).contiguous(), to_right.to(dev).contiguous()
        rl, rr = torch.empty_like(sl), torch.empty_like(sr)
        ops = []
        if self._left is not None:
            ops.append(dist.P2POp(dist.isend, sl, self._left, tag=31))
        if self._right is not None:
            ops.append(dist.P2POp(dist.isend, sr, self._right, tag=32))
        if self._right is not None:
            ops.append(dist.P2POp(dist.irecv, rr, self._right, tag=31))
        if self._left is not None:
            ops.append(dist.P2POp(dist.irecv, rl, self._left, tag=32))
        for q in dist.batch_isend_irecv(ops):
            q.wait()
        odev = d.e.device
        return (rl.to(odev) if self._left is not None else None, rr.to(odev) if self._right is not None else None)

    def _slab_halo_add(self, *names):
        """what this rank's particles deposited in its ghost layers belongs to the neighbours' edge layers: send it,
        add what arrives, then refresh the ghost layers with the neighbours' edge values"""
        if self._cxx_slab:
            mask = sum({"gamma": 1, "Ue": 2, "Asrc": 4}[nm] for nm in names)
            check(self.L.sf_cloud_slab_halo_add(self.ptr, mask))
            return
        for nm in names:
            f = self._field(nm)                      # [nz, ny, nxs(, 3)]
            gl, gr = f[:, :, 0].clone(), f[:, :, -1].clone()
            fl, fr = self._sendrecv(gl, gr)
            if fl is not None:
                f[:, :, 1] += fl
            if fr is not None:
                f[:, :, -2] += fr
            self._slab_refresh(f)

    def _slab_refresh(self, f):
        el, er = f[:, :, 1].clone(), f[:, :, -2].clone()
        fl, fr = self._sendrecv(el, er)
        # from the left neighbour comes ITS right edge = my left ghost layer
        f[:, :, 0] = fl if fl is not None else f[:, :, 1]
        f[:, :, -1] = fr if fr is not None else f[:, :, -2]

    def _a2a(self, send):
        """send[q] -> rank q ; returns recv[p] = what rank p sent to me (equal shapes)"""
        torch, d = self._torch, self.driver
        if d.world == 1:
            return send.clone()
        if d.transport != "host":
            recv = torch.empty_like(send)
            d.dist.all_to_all_single(recv, send.contiguous())
            return recv
        h = send.cpu().contiguous()                  # gloo has no all-to-all: point-to-point messages through the host
        out = torch.empty_like(h)
        ops = []
        for q in range(d.world):
            if q == d.rank:
                out[q] = h[q]
            else:
                ops.append(d.dist.P2POp(d.dist.isend, h[q], q, tag=40))
                ops.append(d.dist.P2POp(d.dist.irecv, out[q], q, tag=40))
        for w in d.dist.batch_isend_irecv(ops):
            w.wait()
        return out.to(send.device)

    def _slab_xsolve(self):
        """the x direction of the implicit diffusion solve: transpose the smoother's work array to complete x-lines
        (each rank a share of the lines), solve, transpose back -- with the two columns next to every slab"""
        torch, d = self._torch, self.driver
        W, r = d.world, d.rank
        wp = _lib.dp(); nf = C.c_int()
        check(self.L.sf_cloud_smooth_work(self.ptr, C.byref(wp), C.byref(nf)))
        nz, ny, nxs = self._shape
        nxg = self.mesh_n_global[0]
        NL = nf.value * nz * ny
        work = self._dev_tensor(C.cast(wp, C.c_void_p).value, (NL, nxs))
        nlq = (NL + W - 1) // W
        send = torch.zeros((W * nlq, self.nxl), dtype=torch.float64, device=work.device)
        send[:NL] = work[:, 1:-1]
        recv = self._a2a(send.view(W, nlq, self.nxl))             # recv[p] = my lines, columns of rank p
        lines = recv.permute(1, 0, 2).reshape(nlq, nxg).contiguous()
        nvalid = max(0, min(nlq, NL - r * nlq))
        check(self.L.sf_cloud_smooth_xsolve(self.ptr, lines.data_ptr(), nvalid, r * nlq))
        cols = []
        for p in range(W):
            ix = np.arange(p * self.nxl - 1, (p + 1) * self.nxl + 1)
            ix = np.mod(ix, nxg) if self._per_x else np.clip(ix, 0, nxg - 1)
            cols.append(lines[:, torch.as_tensor(ix, device=lines.device)])
        back = self._a2a(torch.stack(cols, 0))                    # back[q] = rows of rank q's lines, my columns
        work[:, :] = back.reshape(W * nlq, nxs)[:NL]

    def _slab_phase(self, ph):
        if self._cxx_slab:
            check(self.L.sf_cloud_slab_phase(self.ptr, int(ph)))
            return
        if self._phase(ph) == 1:
            self._slab_xsolve()
            if self._phase(ph) != 0:
                raise SfError("sf_cloud_phase %d did not finish after its x solve" % ph)
        elif ph in (3, 5):
            pass    # (no smoothing configured for this field: the ghost layers were refreshed by _slab_halo_add)

    def owned(self, a):
        """the owned cells [nz*ny*nxl(, 3)] of a local field array returned by gamma() / Ue() / Asrc()"""
        nz, ny, nxs = self._shape
        b = np.asarray(a).reshape(nz, ny, nxs, -1)[:, :, 1:-1, :]
        return b.reshape(nz * ny * self.nxl, -1).squeeze()

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
