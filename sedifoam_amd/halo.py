"""One-process-per-GPU driver of the DEM engine on a 1-D slab decomposition along x.

What LAMMPS' Comm class does for the reference under `mpirun -np N` ([3P] comm.cpp: exchange / borders /
forward_comm), re-designed for MI355X: the pack / unpack halves are HIP kernels on device
buffers inside libsedifoam_amd.so (csrc/sf_dem_halo.hip), the transport is torch.distributed point-to-point
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).  A slab has two face neighbours,
so each GPU talks to two peers over two xGMI links; with `newton off` there is no reverse (force) message.

Per DEM sub-step: one fused force/integrate kernel, one 1-int all-reduce (did any atom move > skin/2?),
one forward halo (72 B per ghost).  On a rebuild: migration of atoms that left the slab (with fix fdrag
arrays, wall and pair shear history), border exchange of ghost atoms, device neighbour build.

The driver only talks to an "engine adaptor" (HipSlabEngine below; the CPU tests plug the oracle in through
the same interface), so the protocol is exercised without a GPU by tests/test_halo_gloo.py.
"""
import ctypes as C
import os

import numpy as np

BORDER_DOUBLES = 14
FORWARD_DOUBLES = 9


class HipSlabEngine:
    """Adaptor: sedifoam_amd.Lammps + torch CUDA buffers -> the calls SlabDriver makes."""

    def __init__(self, lmp):
        import torch
        from . import _lib
        self.torch = torch
        self.lmp = lmp
        self.L = lmp.L
        self.check = _lib.check
        self.device = torch.device("cuda", torch.cuda.current_device())
        # run the engine on torch's current stream: RCCL traffic then orders after the pack kernels and the
        # unpack kernels after the receives without host synchronisation
        self.check(self.L.sf_dem_set_stream(lmp.ptr, torch.cuda.current_stream().cuda_stream))
        # the engine's flag block lives in a torch tensor so that word 0 (the rebuild trigger) can be
        # all-reduced over the ranks on the stream, between two sub-steps, without the host looking at it
        self.flags = torch.zeros(32, dtype=torch.int32, device=self.device)
        self.check(self.L.sf_dem_set_flag_buffer(lmp.ptr, self.flags.data_ptr()))
        self.trigger = self.flags[0:1]

    def alloc(self, ndoubles):
        return self.torch.empty(max(int(ndoubles), 1), dtype=self.torch.float64, device=self.device)

    def info(self):
        return self.lmp.info()

    def set_subdomain(self, rank, world, lo, hi):
        self.check(self.L.sf_dem_set_subdomain(self.lmp.ptr, rank, world, lo, hi))

    def setup(self):
        self.check(self.L.sf_dem_setup(self.lmp.ptr))

    def run_begin(self):
        self.check(self.L.sf_dem_run_begin(self.lmp.ptr))

    def substep(self, last):
        self.check(self.L.sf_dem_substep(self.lmp.ptr, int(last)))

    def need_rebuild(self):
        return self.check(self.L.sf_dem_need_rebuild(self.lmp.ptr))

    def substep_k(self, last, kstep):
        self.check(self.L.sf_dem_substep_k(self.lmp.ptr, int(last), int(kstep)))

    def batch_end(self, first_k, launched):
        t = C.c_int()
        self.check(self.L.sf_dem_batch_end(self.lmp.ptr, int(first_k), int(launched), C.byref(t)))
        return t.value

    def rebuild_begin(self):
        self.check(self.L.sf_dem_rebuild_begin(self.lmp.ptr))

    def rebuild_sort(self):
        self.check(self.L.sf_dem_rebuild_sort(self.lmp.ptr))

    def rebuild_finish(self):
        self.check(self.L.sf_dem_rebuild_finish(self.lmp.ptr))

    def migrate_set_slots(self, m):
        self.check(self.L.sf_dem_migrate_set_slots(self.lmp.ptr, int(m)))

    def migrate_record_doubles(self):
        return self.check(self.L.sf_dem_migrate_record_doubles(self.lmp.ptr))

    def migrate_count(self):
        return self.check(self.L.sf_dem_migrate_count(self.lmp.ptr))

    def local_particle_volume(self):
        v = C.c_double()
        self.check(self.L.sf_dem_local_particle_volume(self.lmp.ptr, C.byref(v)))
        return v.value

    def set_global_particle_volume(self, v):
        self.check(self.L.sf_dem_set_global_particle_volume(self.lmp.ptr, float(v)))

    def local_max_radius(self):
        v = C.c_double()
        self.check(self.L.sf_dem_local_max_radius(self.lmp.ptr, C.byref(v)))
        return v.value

    def set_global_max_radius(self, r):
        self.check(self.L.sf_dem_set_global_max_radius(self.lmp.ptr, float(r)))

    def migrate_pack(self, side, xshift, buf):
        return self.check(self.L.sf_dem_migrate_pack(self.lmp.ptr, side, xshift, buf.data_ptr(), buf.numel()))

    def migrate_unpack(self, buf, ndoubles):
        self.check(self.L.sf_dem_migrate_unpack(self.lmp.ptr, buf.data_ptr(), int(ndoubles)))

    def border_pack(self, side, xshift, buf):
        return self.check(self.L.sf_dem_border_pack(self.lmp.ptr, side, xshift, buf.data_ptr(),
                                                    buf.numel() // BORDER_DOUBLES))

    def border_unpack(self, side, buf, natoms):
        self.check(self.L.sf_dem_border_unpack(self.lmp.ptr, side, buf.data_ptr(), int(natoms)))

    def forward_pack(self, side, xshift, buf):
        return self.check(self.L.sf_dem_forward_pack(self.lmp.ptr, side, xshift, buf.data_ptr()))

    def forward_unpack(self, side, buf, natoms):
        self.check(self.L.sf_dem_forward_unpack(self.lmp.ptr, side, buf.data_ptr(), int(natoms)))

    def ghost_forward_local(self):
        self.check(self.L.sf_dem_ghost_forward_local(self.lmp.ptr))

    def forward_pack2(self, shift0, buf0, shift1, buf1):
        n0 = C.c_longlong(); n1 = C.c_longlong()
        self.check(self.L.sf_dem_forward_pack2(self.lmp.ptr, shift0, buf0.data_ptr(), shift1, buf1.data_ptr(),
                                               C.byref(n0), C.byref(n1)))
        return n0.value, n1.value

    def forward_unpack2(self, buf0, n0, buf1, n1):
        self.check(self.L.sf_dem_forward_unpack2(self.lmp.ptr, buf0.data_ptr(), int(n0), buf1.data_ptr(), int(n1)))

    def index_table(self, values):
        return self.torch.tensor(list(values), dtype=self.torch.int32, device=self.device)

    def forward_pack_fused(self, shift0, off0, shift1, off1, hdr_off, sendbuf):
        self.check(self.L.sf_dem_forward_pack_fused(self.lmp.ptr, shift0, int(off0), shift1, int(off1),
                                                    hdr_off.data_ptr(), hdr_off.numel(), sendbuf.data_ptr()))

    def forward_unpack_fused(self, recvbuf, off_l, n_l, off_r, n_r, hdr_off, kstep=-1):
        self.check(self.L.sf_dem_forward_unpack_fused(self.lmp.ptr, recvbuf.data_ptr(), int(off_l), int(n_l),
                                                      int(off_r), int(n_r), hdr_off.data_ptr(), hdr_off.numel(),
                                                      int(kstep)))

    # ---- the per-sub-step loop in C++ over RCCL (csrc/sf_halo_rccl.hip) ----
    def comm_init(self, dist, rank, world):
        """one RCCL communicator per engine; the 128-byte id travels through the torch process group"""
        ident = [None]
        if rank == 0:
            buf = C.create_string_buffer(128)
            self.check(self.L.sf_dem_comm_unique_id(buf))
            ident[0] = buf.raw
        if world > 1:
            dist.broadcast_object_list(ident, src=0)
        self.check(self.L.sf_dem_comm_init(self.lmp.ptr, ident[0], int(rank), int(world)))

    # ---- the whole driver in C++ (sf_slab_*): Python only forwards ----
    def slab_init(self, dist, rank, world, xlo, xhi, periodic_x):
        ident = [None]
        if rank == 0:
            buf = C.create_string_buffer(128)
            self.check(self.L.sf_dem_comm_unique_id(buf))
            ident[0] = buf.raw
        if world > 1:
            dist.broadcast_object_list(ident, src=0)
        self.check(self.L.sf_slab_init(self.lmp.ptr, ident[0], int(rank), int(world), float(xlo), float(xhi),
                                       int(bool(periodic_x))))

    def brick_init(self, dist, rank, world, grid):
        """3-D processor grid `grid` = (px, py, pz) over the engine's box (sf_brick_init); setup / step / rebuild are
        then the slab_* calls"""
        ident = [None]
        if rank == 0:
            buf = C.create_string_buffer(128)
            self.check(self.L.sf_dem_comm_unique_id(buf))
            ident[0] = buf.raw
        if world > 1:
            dist.broadcast_object_list(ident, src=0)
        self.check(self.L.sf_brick_init(self.lmp.ptr, ident[0], int(rank), int(world), int(grid[0]), int(grid[1]),
                                        int(grid[2])))

    def slab_setup(self):
        self.check(self.L.sf_slab_setup(self.lmp.ptr))

    def slab_rebuild(self):
        self.check(self.L.sf_slab_rebuild(self.lmp.ptr))

    def slab_step(self, n):
        self.check(self.L.sf_slab_step(self.lmp.ptr, int(n)))

    def slab_rebuild_count(self):
        return self.check(self.L.sf_slab_rebuild_count(self.lmp.ptr))

    def halo_run(self, first_k, n, lay):
        from . import _lib
        t = C.c_int()
        if "_c" in lay:                      # the ctypes image of a cached layout
            self.check(self.L.sf_dem_halo_run(self.lmp.ptr, int(first_k), int(n), C.byref(lay["_c"][0]), C.byref(t)))
            return t.value
        W = len(lay["in_split"])
        LL = C.c_longlong * W
        so, ro = [0] * W, [0] * W
        for p in range(1, W):
            so[p] = so[p - 1] + lay["in_split"][p - 1]
            ro[p] = ro[p - 1] + lay["out_split"][p - 1]
        keep = (LL(*so), LL(*lay["in_split"]), LL(*ro), LL(*lay["out_split"]))
        h = _lib.HaloLayout()
        h.world = W
        h.shift_left, h.shift_right = lay["shift_left"], lay["shift_right"]
        h.soff_l, h.soff_r = lay["soff_l"], lay["soff_r"]
        h.roff_l, h.n_from_left = lay["roff_l"], lay["n_from_left"]
        h.roff_r, h.n_from_right = lay["roff_r"], lay["n_from_right"]
        h.send_off, h.send_cnt, h.recv_off, h.recv_cnt = keep
        h.dev_shdr, h.dev_rhdr = lay["shdr"].data_ptr(), lay["rhdr"].data_ptr()
        h.dev_tx, h.dev_rx = lay["tx"].data_ptr(), lay["rx"].data_ptr()
        lay["_c"] = (h, keep)
        self.check(self.L.sf_dem_halo_run(self.lmp.ptr, int(first_k), int(n), C.byref(h), C.byref(t)))
        return t.value

    # ---- overlapped halo: boundary atoms, then [exchange on comm_stream || interior atoms on the main stream] ----
    def enable_overlap(self):
        torch = self.torch
        ncu = int(os.environ.get("SF_HALO_COMM_CUS", "2"))
        if ncu > 0:
            # the engine creates a CU-partitioned stream pair; torch work of the driver runs on the same two streams
            m = C.c_void_p(); c = C.c_void_p()
            self.check(self.L.sf_dem_partition_streams(self.lmp.ptr, ncu, C.byref(m), C.byref(c)))
            self.main_stream = torch.cuda.ExternalStream(m.value, device=self.device)
            self.comm_stream = torch.cuda.ExternalStream(c.value, device=self.device)
            torch.cuda.current_stream().synchronize()
            torch.cuda.set_stream(self.main_stream)
        else:
            self.comm_stream = torch.cuda.Stream(device=self.device, priority=-1)
        self.ev_boundary = torch.cuda.Event()
        self.ev_halo = torch.cuda.Event()
        self.check(self.L.sf_dem_set_overlap(self.lmp.ptr, 1, self.comm_stream.cuda_stream))

    def overlap_begin(self):
        self.check(self.L.sf_dem_overlap_begin(self.lmp.ptr))

    def substep_part(self, part, last, kstep):
        self.check(self.L.sf_dem_substep_part(self.lmp.ptr, int(part), int(last), int(kstep)))

    def substep_flip(self, kstep):
        self.check(self.L.sf_dem_substep_flip(self.lmp.ptr, int(kstep)))

    def overlap_batch_end(self, first_k, launched, last_kstep):
        t = C.c_int()
        self.check(self.L.sf_dem_overlap_batch_end(self.lmp.ptr, int(first_k), int(launched), int(last_kstep),
                                                   C.byref(t)))
        return t.value


class SlabDriver:
    """lammps_step() for one slab of an x-decomposed domain.  All methods are collective over the ranks."""

    def __init__(self, eng, dist, rank, world, xlo, xhi, periodic_x=True, halo_atoms=None, transport="direct",
                 overlap=None):
        import torch
        # "direct": P2P on the engine's own (device) buffers = RCCL over xGMI.  "host": stage through CPU tensors
        # (lets a gloo process group carry the halo of GPU engines, e.g. two ranks sharing one GPU in a test).
        self.transport = transport
        self.torch = torch
        self.e = eng
        self.dist = dist
        self.rank, self.world = rank, world
        self.L = float(xhi - xlo)
        self.periodic_x = bool(periodic_x)
        self.left = rank - 1 if rank > 0 else (world - 1 if periodic_x else None)
        self.right = rank + 1 if rank < world - 1 else (0 if periodic_x else None)
        # shift applied to what goes out through the global box faces
        self.shift_left = self.L if (rank == 0 and periodic_x) else 0.0
        self.shift_right = -self.L if (rank == world - 1 and periodic_x) else 0.0
        w = self.L / world
        self.sublo, self.subhi = xlo + rank * w, xlo + (rank + 1) * w
        if rank == world - 1:
            self.subhi = float(xhi)
        eng.set_subdomain(rank, world, self.sublo, self.subhi)
        # SF_HALO_SELF_COMM=1: a single rank sends its periodic images to itself through the process group instead
        # of handing the buffers over locally (exercises the RCCL path on a 1-GPU box)
        self.self_comm = (world == 1 and periodic_x and dist is not None
                          and os.environ.get("SF_HALO_SELF_COMM", "0") == "1")
        # one all-to-all per sub-step (halo + rebuild vote) instead of an all-reduce and a P2P group
        self.fused = hasattr(eng, "forward_pack_fused") and os.environ.get("SF_HALO_FUSED", "1") != "0"
        # exchange under the interior kernel (see sf_dem_set_overlap in include/sedifoam_amd.h)
        if overlap is None:
            overlap = os.environ.get("SF_HALO_OVERLAP", "0") == "1"
        self.overlap = bool(overlap) and self.fused and hasattr(eng, "enable_overlap") and (world > 1 or self.self_comm)
        if self.overlap:
            eng.enable_overlap()
        # transport "rccl": the WHOLE driver runs in C++ on the engine's own RCCL communicator (sf_slab_*: sub-step
        # loop, rebuild-time migration / border exchanges, global reductions); this class only forwards.  A failure
        # to bring RCCL up is an error -- a silent fall-back would benchmark the Python loop -- unless
        # SF_HALO_ALLOW_FALLBACK=1
        self._cxx = False
        if self.transport == "rccl":
            if not (self.fused and hasattr(eng, "slab_init") and (world > 1 or self.self_comm)):
                self.transport = "direct"      # (single slab without self-communication: nothing to exchange)
            else:
                try:
                    eng.slab_init(dist, rank, world, xlo, xhi, periodic_x)
                    self._cxx = True
                except Exception as ex:      # e.g. librccl not loadable: the same on every rank
                    if os.environ.get("SF_HALO_ALLOW_FALLBACK", "0") != "1":
                        raise RuntimeError("transport='rccl': the C++ RCCL driver is unavailable (%s); set "
                                           "SF_HALO_ALLOW_FALLBACK=1 to run the torch.distributed loop instead" % ex)
                    import warnings
                    warnings.warn("RCCL driver unavailable (%s); using torch.distributed from Python" % ex)
                    self.transport = "direct"
        self._cap_atoms = int(halo_atoms) if halo_atoms else max(eng.info().nlocal, 4096)
        self._bufs = {}
        self._nrecv = [0, 0]
        self._n_rebuilds = 0
        self._lay, self._lay_key = None, -1
        self.is_setup = False

    @property
    def n_rebuilds(self):
        return self.e.slab_rebuild_count() if self._cxx else self._n_rebuilds

    @n_rebuilds.setter
    def n_rebuilds(self, v):
        self._n_rebuilds = v

    # ---- plumbing ----
    def _buf(self, name, ndoubles):
        b = self._bufs.get(name)
        if b is None or b.numel() < ndoubles:
            b = self.e.alloc(int(ndoubles * 1.25) + 64)
            self._bufs[name] = b
        return b

    def _allreduce_max(self, v):
        if self.world == 1 and not self.self_comm:
            return int(v)
        t = self.torch.tensor([int(v)], dtype=self.torch.int64,
                              device=self.e.device if self.transport != "host" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return int(t.item())

    def _allreduce_max2(self, a, b):
        if self.world == 1 and not self.self_comm:
            return int(a), int(b)
        t = self.torch.tensor([int(a), int(b)], dtype=self.torch.int64,
                              device=self.e.device if self.transport != "host" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return int(t[0].item()), int(t[1].item())

    def _exchange(self, send_l, n_l, send_r, n_r, known=None):
        """send_l[:n_l] goes to the left neighbour, send_r[:n_r] to the right one.
        Returns (from_left, n_from_left, from_right, n_from_right) in doubles.  `known` = receive sizes
        when both sides already know them (forward halo), else they are exchanged first."""
        torch, dist = self.torch, self.dist
        if self.world == 1 and not self.self_comm:
            # one slab: my own images.  What leaves through the left face arrives from the right.
            if self.periodic_x:
                return send_r, n_r, send_l, n_l
            return send_l, 0, send_r, 0
        if self.left is None:
            n_l = 0
        if self.right is None:
            n_r = 0
        tdev = self.e.device if self.transport != "host" else torch.device("cpu")
        if known is None:
            cs_l = torch.tensor([n_l], dtype=torch.int64, device=tdev)
            cs_r = torch.tensor([n_r], dtype=torch.int64, device=tdev)
            cr_l = torch.zeros(1, dtype=torch.int64, device=tdev)
            cr_r = torch.zeros(1, dtype=torch.int64, device=tdev)
            ops = []
            if self.left is not None:
                ops.append(dist.P2POp(dist.isend, cs_l, self.left, tag=10))
            if self.right is not None:
                ops.append(dist.P2POp(dist.isend, cs_r, self.right, tag=11))
            if self.right is not None:
                ops.append(dist.P2POp(dist.irecv, cr_r, self.right, tag=10))   # leftward traffic comes from my right
            if self.left is not None:
                ops.append(dist.P2POp(dist.irecv, cr_l, self.left, tag=11))
            for r in dist.batch_isend_irecv(ops):
                r.wait()
            m_l, m_r = int(cr_l.item()), int(cr_r.item())
        else:
            m_l, m_r = known
        recv_l = self._buf("recv_l", m_l)
        recv_r = self._buf("recv_r", m_r)
        if self.transport != "host":
            tx_l, tx_r, rx_l, rx_r = send_l, send_r, recv_l, recv_r
        else:
            if self.e.device.type == "cuda":
                torch.cuda.synchronize()
            tx_l, tx_r = send_l[:n_l].cpu(), send_r[:n_r].cpu()
            rx_l = torch.empty(max(m_l, 1), dtype=torch.float64)
            rx_r = torch.empty(max(m_r, 1), dtype=torch.float64)
        ops = []
        if self.left is not None and n_l:
            ops.append(dist.P2POp(dist.isend, tx_l[:n_l], self.left, tag=20))
        if self.right is not None and n_r:
            ops.append(dist.P2POp(dist.isend, tx_r[:n_r], self.right, tag=21))
        if self.right is not None and m_r:
            ops.append(dist.P2POp(dist.irecv, rx_r[:m_r], self.right, tag=20))
        if self.left is not None and m_l:
            ops.append(dist.P2POp(dist.irecv, rx_l[:m_l], self.left, tag=21))
        if ops:
            for r in dist.batch_isend_irecv(ops):
                r.wait()
        if self.transport == "host":
            recv_l[:m_l].copy_(rx_l[:m_l])
            recv_r[:m_r].copy_(rx_r[:m_r])
        return recv_l, m_l, recv_r, m_r

    # ---- the three halo operations ----
    def rebuild(self):
        e = self.e
        if self._cxx:
            e.slab_rebuild()
            return
        e.rebuild_begin()
        # one all-reduce carries the history slots a migrating atom needs (max over ranks of max_neigh_used) and, in
        # the bits above, whether any rank has an atom outside its slab: the usual rebuild migrates nothing and then
        # skips the pack / count exchange / data exchange / unpack round
        crossed = e.migrate_count() if hasattr(e, "migrate_count") else 1
        # (one collective, two values, MAX each: the slot count over ALL ranks whoever migrates)
        v, c = self._allreduce_max2(int(e.info().max_neigh_used), 1 if crossed else 0)
        e.migrate_set_slots(v)
        if c:
            rec = e.migrate_record_doubles()
            nmax = max(self._cap_atoms // 8, 1024)
            b0 = self._buf("mig_l", nmax * rec)
            b1 = self._buf("mig_r", nmax * rec)
            n0 = e.migrate_pack(0, self.shift_left, b0)
            n1 = e.migrate_pack(1, self.shift_right, b1)
            rl, ml, rr, mr = self._exchange(b0, n0, b1, n1)
            if self.world == 1 and not self.periodic_x and (n0 or n1):
                raise RuntimeError("Lost atoms: an atom left the non-periodic box in x")
            e.migrate_unpack(rl, ml)
            e.migrate_unpack(rr, mr)
        e.rebuild_sort()
        cap = max(self._cap_atoms, e.info().nlocal)
        s0 = self._buf("bor_l", cap * BORDER_DOUBLES)
        s1 = self._buf("bor_r", cap * BORDER_DOUBLES)
        a0 = e.border_pack(0, self.shift_left, s0)
        a1 = e.border_pack(1, self.shift_right, s1)
        self._nsend = [a0, a1]
        rl, ml, rr, mr = self._exchange(s0, a0 * BORDER_DOUBLES, s1, a1 * BORDER_DOUBLES)
        self._nrecv = [ml // BORDER_DOUBLES, mr // BORDER_DOUBLES]
        e.border_unpack(0, rl, self._nrecv[0])
        e.border_unpack(1, rr, self._nrecv[1])
        e.rebuild_finish()
        self._n_rebuilds += 1

    def _fused_layout(self):
        """Send / receive layout of the one-collective forward halo (valid until the next rebuild, cached): per peer
        rank one header double (rebuild trigger) + the forward records for / from that peer."""
        if self._lay is not None and self._lay_key == self.n_rebuilds:
            return self._lay
        self._lay = self._make_fused_layout()
        self._lay_key = self.n_rebuilds
        return self._lay

    def _make_fused_layout(self):
        F = FORWARD_DOUBLES
        ns, nr = self._nsend, self._nrecv          # [to left, to right], [from left, from right]
        W = self.world
        in_split = [1] * W
        out_split = [1] * W
        if self.left is not None:
            in_split[self.left] += ns[0] * F
            out_split[self.left] += nr[0] * F
        if self.right is not None:
            in_split[self.right] += ns[1] * F
            out_split[self.right] += nr[1] * F
        sbase = [0] * W
        rbase = [0] * W
        for p in range(1, W):
            sbase[p] = sbase[p - 1] + in_split[p - 1]
            rbase[p] = rbase[p - 1] + out_split[p - 1]
        same = self.left is not None and self.left == self.right
        lay = dict(in_split=in_split, out_split=out_split, ntx=sum(in_split), nrx=sum(out_split))
        # left-going records first, then right-going ones, inside the chunk for a peer that is both neighbours;
        # what a peer sent leftwards reaches me from my right, so its chunk holds [from-right part, from-left part]
        # records selected at a face without a neighbour (non-periodic box end) go to scratch behind the chunks
        ntx = lay["ntx"]
        scratch = 0
        if self.left is not None:
            lay["soff_l"] = sbase[self.left] + 1
        else:
            lay["soff_l"] = ntx + scratch
            scratch += ns[0] * F
        if self.right is not None:
            lay["soff_r"] = sbase[self.right] + 1 + (ns[0] * F if same else 0)
        else:
            lay["soff_r"] = ntx + scratch
            scratch += ns[1] * F
        lay["roff_r"] = rbase[self.right] + 1 if self.right is not None else 0
        lay["roff_l"] = (rbase[self.left] + 1 + (nr[1] * F if same else 0)) if self.left is not None else 0
        lay["shift_left"], lay["shift_right"] = self.shift_left, self.shift_right
        lay["n_from_left"], lay["n_from_right"] = nr[0], nr[1]
        lay["shdr"] = self.e.index_table(sbase)
        lay["rhdr"] = self.e.index_table(rbase)
        lay["tx"] = self._buf("a2a_tx", lay["ntx"] + scratch)
        lay["rx"] = self._buf("a2a_rx", lay["nrx"])
        return lay

    def _all_to_all(self, lay):
        """rx[chunk p] <- what rank p put into its chunk for me."""
        torch, dist = self.torch, self.dist
        tx, rx = lay["tx"][:lay["ntx"]], lay["rx"][:lay["nrx"]]
        if self.transport != "host":
            dist.all_to_all_single(rx, tx, lay["out_split"], lay["in_split"])
            return
        # gloo has no all-to-all: the same chunks as point-to-point messages through host memory
        if self.e.device.type == "cuda":
            torch.cuda.synchronize()
        htx = tx.cpu()
        hrx = torch.empty(lay["nrx"], dtype=torch.float64)
        so = ro = 0
        ops = []
        for p in range(self.world):
            a, b = lay["in_split"][p], lay["out_split"][p]
            if p == self.rank:
                hrx[ro:ro + b] = htx[so:so + a]
            else:
                ops.append(dist.P2POp(dist.isend, htx[so:so + a], p))
                ops.append(dist.P2POp(dist.irecv, hrx[ro:ro + b], p))
            so += a
            ro += b
        if ops:
            for r in dist.batch_isend_irecv(ops):
                r.wait()
        rx.copy_(hrx)

    def forward_fused(self, lay, kstep=-1):
        """rebuild vote + forward halo of one sub-step: pack kernel, ONE collective, unpack kernel."""
        e = self.e
        e.forward_pack_fused(self.shift_left, lay["soff_l"], self.shift_right, lay["soff_r"], lay["shdr"], lay["tx"])
        self._all_to_all(lay)
        e.forward_unpack_fused(lay["rx"], lay["roff_l"], self._nrecv[0], lay["roff_r"], self._nrecv[1], lay["rhdr"],
                               kstep)

    def _exchange_overlapped(self, lay, kstep):
        """the exchange that follows sub-step kstep, on the communication stream, after the boundary kernel"""
        e = self.e
        cs = getattr(e, "comm_stream", None)
        if cs is None:                      # CPU stand-in engine: no streams
            self.forward_fused(lay, kstep)
            return
        torch = self.torch
        e.ev_boundary.record(torch.cuda.current_stream())
        with torch.cuda.stream(cs):
            cs.wait_event(e.ev_boundary)
            self.forward_fused(lay, kstep)
            e.ev_halo.record(cs)

    def _wait_halo(self):
        e = self.e
        if getattr(e, "comm_stream", None) is not None:
            self.torch.cuda.current_stream().wait_event(e.ev_halo)

    def _step_overlapped(self, n):
        e = self.e
        e.run_begin()
        k = 0
        while k < n:
            lay = self._fused_layout()
            self._exchange_overlapped(lay, k - 1)          # ghosts + vote before sub-step k
            for s in range(k, n):
                last = s == n - 1
                self._wait_halo()
                e.substep_part(2, last, s)                 # boundary atoms (read the ghosts of exchange s-1)
                # (the event recorded inside _exchange_overlapped sits between the two parts on the main stream)
                if getattr(e, "comm_stream", None) is not None:
                    e.ev_boundary.record(self.torch.cuda.current_stream())
                e.substep_part(1, last, s)                 # interior atoms, under the exchange of sub-step s
                e.substep_flip(s)
                self._exchange_after_boundary(lay, s)
            self._wait_halo()
            trig = e.overlap_batch_end(k, n - k, n - 1)
            if trig >= n:
                break
            k = trig + 1
            self.rebuild()
            e.overlap_begin()

    def _exchange_after_boundary(self, lay, kstep):
        """like _exchange_overlapped, but the boundary event has already been recorded (before the interior kernel)"""
        e = self.e
        cs = getattr(e, "comm_stream", None)
        if cs is None:
            self.forward_fused(lay, kstep)
            return
        torch = self.torch
        with torch.cuda.stream(cs):
            cs.wait_event(e.ev_boundary)
            self.forward_fused(lay, kstep)
            e.ev_halo.record(cs)

    def forward(self):
        e = self.e
        f0 = self._buf("fwd_l", self._nsend[0] * FORWARD_DOUBLES)
        f1 = self._buf("fwd_r", self._nsend[1] * FORWARD_DOUBLES)
        if hasattr(e, "forward_pack2"):
            a0, a1 = e.forward_pack2(self.shift_left, f0, self.shift_right, f1)
        else:
            a0 = e.forward_pack(0, self.shift_left, f0)
            a1 = e.forward_pack(1, self.shift_right, f1)
        rl, ml, rr, mr = self._exchange(f0, a0 * FORWARD_DOUBLES, f1, a1 * FORWARD_DOUBLES,
                                        known=(self._nrecv[0] * FORWARD_DOUBLES, self._nrecv[1] * FORWARD_DOUBLES))
        if hasattr(e, "forward_unpack2"):
            e.forward_unpack2(rl, ml // FORWARD_DOUBLES, rr, mr // FORWARD_DOUBLES)
        else:
            e.forward_unpack(0, rl, ml // FORWARD_DOUBLES)
            e.forward_unpack(1, rr, mr // FORWARD_DOUBLES)
            e.ghost_forward_local()

    # ---- lammps_* surface ----
    def _allreduce_sum_f64(self, v):
        if self.world == 1 and not self.self_comm:
            return float(v)
        t = self.torch.tensor([float(v)], dtype=self.torch.float64,
                              device=self.e.device if self.transport != "host" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def _allreduce_max_f64(self, v):
        if self.world == 1 and not self.self_comm:
            return float(v)
        t = self.torch.tensor([float(v)], dtype=self.torch.float64,
                              device=self.e.device if self.transport != "host" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def setup(self):
        if self._cxx:
            self.e.slab_setup()
            self.is_setup = True
            return
        # list / ghost cutoff 2 r_max + skin: r_max over ALL ranks ([3P] MPI_Allreduce of maxrad_dynamic)
        if hasattr(self.e, "local_max_radius"):
            self.e.set_global_max_radius(self._allreduce_max_f64(self.e.local_max_radius()))
        self.rebuild()
        # pair lubricate/poly: volume fraction of ALL particles (MPI_Allreduce, pair_lubricate_poly.cpp:540-543)
        if hasattr(self.e, "local_particle_volume"):
            self.e.set_global_particle_volume(self._allreduce_sum_f64(self.e.local_particle_volume()))
        self.e.setup()
        self.is_setup = True

    def _reduce_trigger(self):
        """global rebuild vote: MIN over the ranks of the device-resident trigger word, on the stream."""
        if self.world == 1 and not self.self_comm:
            return
        trig = self.e.trigger
        if self.transport != "host":
            self.dist.all_reduce(trig, op=self.dist.ReduceOp.MIN)
        else:
            h = trig.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.MIN)
            trig.copy_(h)

    def step(self, n):
        """lammps_step(n) = "run n pre no post no" (library.cpp:372-386) on the decomposed domain.

        All n sub-steps (kernel + rebuild vote + forward halo each) are queued without the host waiting for any
        of them: a sub-step whose index lies behind the all-reduced trigger exits early on every rank, the host
        synchronises once per batch, rebuilds if the trigger fired and queues the rest."""
        if not self.is_setup:
            self.setup()
        n = int(n)
        e = self.e
        if self._cxx:
            e.slab_step(n)
            return
        if self.overlap:
            self._step_overlapped(n)
            return
        e.run_begin()
        k = 0
        fused = self.fused and (self.world > 1 or self.self_comm)
        while k < n:
            if fused:
                lay = self._fused_layout()
                for s in range(k, n):
                    self.forward_fused(lay)
                    e.substep_k(s == n - 1, s)
            else:
                for s in range(k, n):
                    self._reduce_trigger()
                    self.forward()
                    e.substep_k(s == n - 1, s)
            trig = e.batch_end(k, n - k)
            if trig >= n:
                break
            k = trig + 1          # sub-steps k..trig ran (trig = -1: the list was stale for sub-step 0)
            self.rebuild()

    def info(self):
        return self.e.info()

    def set_profiling(self, on=True):
        self.e.lmp.set_profiling(on)

    def get_profile(self):
        return self.e.lmp.get_profile()

    def get_rebuild_profile(self):
        """(rebuilds since setup, host ms spent in them) -- C++ driver only"""
        if not getattr(self, "_cxx", False):
            return 0, 0.0
        import ctypes as C
        n, ms = C.c_longlong(0), C.c_double(0.0)
        self.e.check(self.e.L.sf_slab_rebuild_profile(self.e.lmp.ptr, C.byref(n), C.byref(ms)))
        return int(n.value), float(ms.value)

    def comm_info(self):
        """{"rccl_ranks", "rccl_version", "rccl_library"} of the C++ driver's communicator (ncclCommCount, ncclGetVersion, the
        shared object its nccl* symbols come from); None when the Python loop drives the halo"""
        if not getattr(self, "_cxx", False):
            return None
        import ctypes as C
        n, v = C.c_int(0), C.c_int(0)
        path = C.create_string_buffer(256)
        self.e.check(self.e.L.sf_slab_comm_info(self.e.lmp.ptr, C.byref(n), C.byref(v), path, 256))
        return {"rccl_ranks": int(n.value), "rccl_version": int(v.value), "rccl_library": path.value.decode()}

    def get_exchange_profile(self):
        """(sampled forward exchanges, their summed ms) since profiling was switched on -- C++ driver only"""
        if not getattr(self, "_cxx", False):
            return 0, 0.0
        import ctypes as C
        n, ms = C.c_longlong(0), C.c_double(0.0)
        self.e.check(self.e.L.sf_slab_exchange_profile(self.e.lmp.ptr, C.byref(n), C.byref(ms)))
        return int(n.value), float(ms.value)

    @classmethod
    def from_global_bed(cls, bed, script, dist, rank, world, transport=None, grid=None):
        """bench.py, strong scaling (BASELINE config C4): ONE bed for all ranks, every rank owns the atoms of its
        x slab (tags = global index + 1) -- or, with grid = (px, py, pz), of its brick (BrickDriver)."""
        if grid is not None and tuple(grid) != (world, 1, 1):
            return BrickDriver.from_global_bed(bed, script, dist, rank, world, grid)
        from . import Lammps
        if transport is None:
            transport = os.environ.get("SF_HALO_TRANSPORT", "rccl")
        lo, hi = float(bed["boxlo"][0]), float(bed["boxhi"][0])
        w = (hi - lo) / world
        x = np.asarray(bed["x"])
        mine = (x[:, 0] >= lo + rank * w) & ((x[:, 0] < lo + (rank + 1) * w) | (rank == world - 1))
        lmp = Lammps()
        lmp.set_box(bed["boxlo"], bed["boxhi"])
        lmp.create_atoms(x[mine], np.asarray(bed["diameter"])[mine], np.asarray(bed["density"])[mine],
                         v=np.asarray(bed["v"])[mine], tag=(np.nonzero(mine)[0] + 1).astype(np.int64))
        for line in script:
            lmp.command(line)
        return cls(HipSlabEngine(lmp), dist, rank, world, lo, hi, periodic_x=bool(bed["periodic"][0]),
                   transport=transport)

    @classmethod
    def from_bed(cls, bed, script, dist, rank, world, transport=None):
        """bench.py: every rank owns one copy of `bed` (its own seed), laid side by side along x."""
        from . import Lammps
        if transport is None:
            transport = os.environ.get("SF_HALO_TRANSPORT", "rccl")
        lx = float(bed["boxhi"][0] - bed["boxlo"][0])
        x = np.array(bed["x"], copy=True)
        x[:, 0] += rank * lx
        lo = np.array(bed["boxlo"], dtype=np.float64)
        hi = np.array(bed["boxhi"], dtype=np.float64)
        hi[0] = lo[0] + world * lx
        lmp = Lammps()
        lmp.set_box(lo, hi)
        n = x.shape[0]
        lmp.create_atoms(x, bed["diameter"], bed["density"], v=bed["v"],
                         tag=np.arange(1, n + 1, dtype=np.int64) + rank * n)
        for line in script:
            lmp.command(line)
        return cls(HipSlabEngine(lmp), dist, rank, world, lo[0], hi[0], periodic_x=bool(bed["periodic"][0]),
                   transport=transport)


def brick_grid(world, bed=None):
    """processor grid of the brick driver for `world` ranks: as cubic as the factorisation allows, never cutting a
    non-periodic (wall) dimension of `bed` before the periodic ones are cut (8 -> 2x2x2, 4 -> 2x2x1 / 2x1x2, 2 -> 2x1x1)"""
    periodic = tuple(bed["periodic"]) if bed is not None else (1, 1, 1)
    best = None
    for px in range(1, world + 1):
        if world % px:
            continue
        for py in range(1, world // px + 1):
            if (world // px) % py:
                continue
            g = (px, py, world // px // py)
            walls_cut = sum(1 for k in range(3) if g[k] > 1 and not periodic[k])
            key = (walls_cut, max(g) - min(g), -g[0], -g[2])
            if best is None or key < best[0]:
                best = (key, g)
    return best[1]


def brick_mask(bed, rank, grid):
    """atoms of `bed` inside the brick of `rank` (x fastest in the rank numbering, like sf_brick_init)"""
    c = (rank % grid[0], (rank // grid[0]) % grid[1], rank // (grid[0] * grid[1]))
    x = np.asarray(bed["x"])
    m = np.ones(len(x), dtype=bool)
    for k in range(3):
        lo, hi = float(bed["boxlo"][k]), float(bed["boxhi"][k])
        w = (hi - lo) / grid[k]
        m &= (x[:, k] >= lo + c[k] * w) if c[k] > 0 else True
        m &= (x[:, k] < lo + (c[k] + 1) * w) if c[k] < grid[k] - 1 else True
    return m


class BrickDriver(SlabDriver):
    """lammps_step() for one brick of a domain cut by a 3-D processor grid.  The whole driver is C++ over RCCL
    (sf_brick_init, then sf_slab_setup / _step / _rebuild: csrc/sf_halo_rccl.hip); this class only forwards."""

    def __init__(self, eng, dist, rank, world, grid):
        import torch
        self.torch, self.e, self.dist = torch, eng, dist
        self.rank, self.world, self.grid = rank, world, tuple(int(g) for g in grid)
        self.transport = "rccl"
        self.self_comm = False
        self.overlap = False
        self.fused = True
        eng.brick_init(dist, rank, world, self.grid)
        self._cxx = True
        self._n_rebuilds = 0
        self.is_setup = False

    @classmethod
    def from_global_bed(cls, bed, script, dist, rank, world, grid):
        from . import Lammps
        mine = brick_mask(bed, rank, grid)
        x = np.asarray(bed["x"])
        lmp = Lammps()
        lmp.set_box(bed["boxlo"], bed["boxhi"])
        lmp.create_atoms(x[mine], np.asarray(bed["diameter"])[mine], np.asarray(bed["density"])[mine],
                         v=np.asarray(bed["v"])[mine], tag=(np.nonzero(mine)[0] + 1).astype(np.int64))
        for line in script:
            lmp.command(line)
        return cls(HipSlabEngine(lmp), dist, rank, world, grid)
