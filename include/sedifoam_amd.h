/* sedifoam_amd.h -- C-ABI of libsedifoam_amd.so: the MI355X-native (gfx950, HIP) replacement of
 * sediFoam's per-step CFD-DEM particle hot path.
 *
 * Three groups of entry points, all `extern "C"`, plain pointers and sizes, int status returns
 * (0 = ok, <0 = error; text via sf_last_error()).  Paths cited are relative to the reference
 * checkout (xiaoh/sediFoam).
 *
 *  (1) sf_lammps_*  : the patched LAMMPS C library interface the OpenFOAM side binds
 *                     (interfaceToLammps/library.h:29-63).  Same argument meaning, same AoS
 *                     host buffers (xyz-interleaved doubles, int32 ids/tags, caller allocates).
 *                     With SEDIFOAM_AMD_LAMMPS_NAMES defined before including this header the
 *                     reference's own names (lammps_open, lammps_step, ...) are provided as
 *                     inline forwards so softParticleCloud.C compiles against it unchanged.
 *  (2) sfk_*        : per-kernel entry points on DEVICE pointers, one per reference
 *                     PairStyle / FixStyle / dragModel / cloud method on the hot path.
 *  (3) sf_cloud_*   : the enhancedCloud surface (lammpsFoam/enhancedCloud.H:183-249) on
 *                     device-resident fields, driving (1) without host marshalling.
 *  (4) sf_dem_*     : fine-grained stepping + halo pack/unpack used by the one-process-per-GPU
 *                     driver (ghost-particle exchange over RCCL, sedifoam_amd/halo.py).
 *
 * All device work is issued on the engine's own HIP stream; functions that hand data to the
 * host synchronise that stream before returning.  One engine per process per GPU; calls on one
 * engine must come from one host thread at a time (same rule as the reference: single-threaded
 * per MPI rank).
 */
#ifndef SEDIFOAM_AMD_H
#define SEDIFOAM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char *sf_last_error(void);
/* 0 if a HIP device is usable; never falls back to a CPU path */
int sf_device_check(void);
const char *sf_version(void);

/* ------------------------------------------------------------------------------------------
 * (1) LAMMPS library surface -- interfaceToLammps/library.h:29-63, library.cpp:40-621
 * ---------------------------------------------------------------------------------------- */

/* library.h:29 lammps_open(int, char**, MPI_Comm, void**): `comm` is carried opaquely (this
 * library needs no MPI: one engine per process, ranks are wired up by sf_dem_* below). */
int sf_lammps_open(int argc, char **argv, intptr_t comm, void **ptr);
/* The same on rank `rank` of `world` ranks: what `new LAMMPS(0, NULL, commLammps)` is when lammpsFoam runs
 * `mpirun -np N lammpsFoam -parallel` (lammpsFoam/softParticleCloud.C:60-62).  The library links no MPI, so the
 * caller -- include/lammps_shim/sedifoam_lammps_shim.h does it with the application's own mpi.h -- passes its rank,
 * the size and a 128-byte communicator id (sf_dem_comm_unique_id on rank 0, MPI_Bcast).  With world > 1 the engine
 * decomposes itself as LAMMPS does: `processors px py pz` (default: the grid of least surface area, [3P]
 * ProcMap::onelevel_grid) cuts the box into bricks when read_data creates it, every rank keeps the atoms of its
 * brick, and sf_lammps_step / _get_global_n / _get_initial_np / _create_particle / _delete_particle become
 * collective exactly where interfaceToLammps/library.cpp:94-131,372-386,470-473 is.  One process per GPU: the device
 * is rank % (visible devices) unless SF_DEVICE names one. */
int sf_lammps_open_world(int argc, char **argv, intptr_t comm, int rank, int world, const char *id128, void **ptr);
/* the processor grid a `processors px py pz` line (user[k] = 0 for `*`) resolves to on `world` ranks for a box [lo, hi):
 * [3P] LAMMPS 1Feb14 ProcMap::onelevel_grid -- of the factorisations that agree with the given entries, visited with px
 * slowest, the first of least sub-domain surface xy / (px py) + xz / (px pz) + yz / (py pz).  Host logic only. */
int sf_procgrid_choose(int world, const double lo[3], const double hi[3], const int user[3], int out[3]);
/* library.h:30 */
int sf_lammps_close(void *ptr);
/* library.h:31  run every line of an input script (library.cpp:63-67) */
int sf_lammps_file(void *ptr, const char *path);
/* library.h:32  one input-script command (library.cpp:73-77).  Understood commands are the ones
 * the reference's in.lammps files use: units, atom_style sphere, atom_modify, boundary, newton,
 * communicate, processors, read_data (or sf_dem_create_atoms), neighbor, neigh_modify, pair_style {gran/hertzFix/history,
 * gran/hooke/history, lubricate/poly, hybrid/overlay}, pair_coeff, timestep, velocity all set,
 * fix {nve/sphere, gravity, fdrag, freeze, cohesive, wall/gran and wall/granFix ({x,y,z}plane | zcylinder, wiggle | shear)},
 * group {type, subtract, union, intersect}, run; thermo*, dump and restart are accepted without effect.
 * Returns NULL like LAMMPS, or an error string. */
const char *sf_lammps_command(void *ptr, const char *line);
/* library.h:34 (debug barrier) -- a stream synchronise here */
int sf_lammps_sync(void *ptr);
/* library.h:35 */
int sf_lammps_get_global_n(void *ptr);
/* library.h:38  np_[nprocs]: atoms owned by each rank (this rank's slot filled; caller reduces) */
int sf_lammps_get_initial_np(void *ptr, int *np_);
/* library.h:40-42  diam = 2 r; rho = 3 m / (4 pi' r^3) with the reference's pi' (library.cpp:200) */
int sf_lammps_get_initial_info(void *ptr, double *coords, double *velos, double *diam,
                               double *rho_, int *tag_, int *lmpCpuId_, int *type_);
/* library.h:45 */
int sf_lammps_get_local_n(void *ptr);
/* library.h:48  {xlo,xhi,ylo,yhi,zlo,zhi} of this rank's sub-domain */
int sf_lammps_get_local_domain(void *ptr, double *domain_);
/* library.h:51-52 */
int sf_lammps_get_local_info(void *ptr, double *coords, double *velos_, int *foamCpuId_,
                             int *lmpCpuId_, int *tag_);
/* library.h:55-56  rows matched to atoms by tag (library.cpp:344-366); DuDt accepted and
 * ignored exactly as the reference does (library.cpp:314-367 never reads it) */
int sf_lammps_put_local_info(void *ptr, int nLocalIn, const double *fdrag, const double *DuDt,
                             const int *foamCpuIdIn, const int *tagIn);
/* library.h:58  "run n pre no post no" (library.cpp:372-386) */
int sf_lammps_step(void *ptr, int n);
/* library.h:59-60 */
int sf_lammps_set_timestep(void *ptr, double dt_i);
double sf_lammps_get_timestep(void *ptr);
/* library.h:61-63 (particle injection / removal; tag[] is double in the reference) */
int sf_lammps_create_particle(void *ptr, int npAdd, const double *position, const double *tag,
                              double diameter, double rho, int type, const double *vel);
int sf_lammps_delete_particle(void *ptr, const int *deleteList, int nDelete);

/* Atoms without a data file: what `read_data` would have loaded (atom_style sphere rows:
 * tag type diameter density x y z).  mass = 4/3 pi r^3 rho like LAMMPS read_data. */
int sf_dem_create_atoms(void *ptr, int n, const double *x /*3n*/, const double *v /*3n|NULL*/,
                        const double *omega /*3n|NULL*/, const double *diameter,
                        const double *density, const int *tag /*n|NULL*/, const int *type /*n|NULL*/);
int sf_dem_set_box(void *ptr, const double lo[3], const double hi[3]);

/* ------------------------------------------------------------------------------------------
 * (4) fine-grained DEM stepping, device access and ghost-particle halo
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int nlocal, nghost, capacity, max_neigh_used, max_neigh_cap;
  long long nbuilds, nsteps;
  long long npairs_full; /* sum of numneigh over owned atoms (full list) */
} sf_dem_info;
int sf_dem_get_info(void *ptr, sf_dem_info *out);
/* device pointers into the engine's state (valid until the next rebuild / step):
 *   xr, vm, om : double4 records (x,y,z,radius) (vx,vy,vz,rmass) (wx,wy,wz,0)
 *   fdrag      : component-major [3][capacity] doubles ; tag/type : int32 [capacity] */
typedef struct {
  void *xr, *vm, *om, *force, *torque;
  double *fdrag, *DuDt, *vOld;
  int *tag, *type, *foamCpuId;
  int nlocal, nghost, capacity;
  void *stream; /* hipStream_t */
} sf_dem_device_view;
int sf_dem_device_view_get(void *ptr, sf_dem_device_view *out);
/* HIP-event timing of every launch of the fused sub-step kernel since profiling was switched on
 * (events recorded on the engine's stream): number of launches and their summed duration */
int sf_dem_set_profiling(void *ptr, int on);
int sf_dem_get_profile(void *ptr, long long *launches, double *kernel_ms);
/* neighbour rebuilds ([3P] Neighbor::build + the re-sort, ghosts and history carry-over around it) that sf_dem_step /
 * lammps_step ran while profiling was on: how many, and their summed host-clock duration (synchronised) */
int sf_dem_get_rebuild_profile(void *ptr, long long *rebuilds, double *ms);
/* forces/torques of owned atoms to host AoS (3n each, engine order) with tags */
int sf_dem_get_forces(void *ptr, double *f, double *torque, double *omega, int *tag);
/* touching pairs (tag_i < tag_j, shear oriented i->j) ; returns count or <0 */
long long sf_dem_get_history(void *ptr, long long max, int *tag_i, int *tag_j, double *shear);
/* per-atom wall shear of wall fix w, AoS 3n in engine order */
int sf_dem_get_wall_shear(void *ptr, int w, double *shear);

/* sub-domain decomposition (1-D slabs along x; rank r owns [sublo, subhi) of the global box).
 * left/right < 0 : no neighbour on that side (wall or self-periodic handled internally). */
int sf_dem_set_subdomain(void *ptr, int rank, int nranks, double sublo, double subhi);
/* step phases (what sf_lammps_step does internally, exposed for the multi-rank driver) */
int sf_dem_run_begin(void *ptr);            /* initial_integrate with the stored forces */
int sf_dem_substep(void *ptr, int last);    /* fused force + final [+ next initial] kernel */
int sf_dem_need_rebuild(void *ptr);         /* 1 if any owned atom moved > skin/2 (synchronises) */
/* Batch mode (no host synchronisation per sub-step).  The engine keeps an int32 "trigger" word on the device:
 * the smallest sub-step index whose new positions left the skin/2 sphere (INT_MAX: none); a sub-step whose
 * index is larger than the trigger exits immediately.  sf_dem_set_flag_buffer places the 32-int flag block
 * (trigger = word 0) in caller-owned device memory so the driver can all-reduce(MIN) it over the ranks with RCCL
 * on the same stream between two sub-steps; sf_dem_batch_end synchronises once, returns the trigger and repairs
 * the ping-pong parity for the sub-steps that exited early. */
int sf_dem_substep_k(void *ptr, int last, int kstep);
int sf_dem_batch_end(void *ptr, int first_k, int launched, int *trigger);
int sf_dem_set_flag_buffer(void *ptr, void *dev_ints32);
int sf_dem_setup(void *ptr);                /* first run's setup: forces with shearupdate = 0 */
int sf_dem_rebuild_begin(void *ptr);        /* shear history -> partner tags; forget ghosts */
int sf_dem_rebuild_sort(void *ptr);         /* pbc + sort owned atoms by bin (after migration) */
int sf_dem_rebuild_finish(void *ptr);       /* periodic ghosts (y,z), bins, full list, history */
/* halo: side 0 = towards -x neighbour, 1 = towards +x.  Buffers are DEVICE doubles.
 * pack: side = the face the atoms sit at / leave through; unpack: side = the neighbour the data came FROM.
 * border records (on rebuild)  : 13 doubles / atom (x y z r vx vy vz m wx wy wz tag type)
 * forward records (every step) : 9 doubles / atom  (x y z vx vy vz wx wy wz)
 * migrate records (on rebuild) : sf_dem_migrate_record_doubles() doubles / atom (26 + 3 nwalls + 4 slots:
 *                                state, fix fdrag arrays, wall shear, partner tags + shear history) */
long long sf_dem_border_pack(void *ptr, int side, double xshift, double *dev_buf, long long max_atoms);
int sf_dem_border_unpack(void *ptr, int side, const double *dev_buf, long long natoms);
long long sf_dem_forward_pack(void *ptr, int side, double xshift, double *dev_buf);
int sf_dem_forward_unpack(void *ptr, int side, const double *dev_buf, long long natoms);
/* both faces in one launch each (per-sub-step path); unpack2 also refreshes the local y/z images */
int sf_dem_forward_pack2(void *ptr, double shift0, double *buf0, double shift1, double *buf1, long long *n0,
                         long long *n1);
int sf_dem_forward_unpack2(void *ptr, const double *buf0, long long n0, const double *buf1, long long n1);
/* One all-to-all per sub-step instead of a P2P halo plus an all-reduce: the send buffer holds, per peer rank, one
 * header double (this rank's rebuild trigger) at dev_hdr_off[peer] followed by the forward records for that peer
 * (left-going records at off0, right-going ones at off1, offsets in doubles).  The unpack side takes the MIN of the
 * nhdr received headers into the device trigger word and refreshes the ghosts that came from the left / right. */
int sf_dem_forward_pack_fused(void *ptr, double shift0, long long off0, double shift1, long long off1,
                              const int *dev_hdr_off, int nhdr, double *dev_sendbuf);
int sf_dem_forward_unpack_fused(void *ptr, const double *dev_recvbuf, long long off_from_left,
                                long long n_from_left, long long off_from_right, long long n_from_right,
                                const int *dev_hdr_off, int nhdr, int kstep);
/* Overlapped halo (opt-in per engine, before the first rebuild): a sub-step is split into the BOUNDARY atoms (those
 * the forward halo sends or whose list holds an atom of another GPU) and the INTERIOR atoms.  Per sub-step k:
 *   main stream:  substep_part(2,k) -> [event A] -> substep_part(1,k) -> substep_flip(k)
 *   comm stream:  wait A -> forward_pack_fused -> all-to-all -> forward_unpack_fused(.., k) -> [event B]
 *   main stream:  wait B -> substep_part(2,k+1) ...
 * so the exchange runs under the interior kernel.  The rebuild vote of exchange k (MIN over all ranks) decides
 * whether sub-step k+1 runs; an interior atom that crosses skin/2 in sub-step k is voted one exchange later, i.e.
 * sub-step k+1 still runs on every rank and the rebuild follows it.  The neighbour list is built with a 10 % larger
 * skin in this mode so that it stays a superset of the interacting pairs for that extra sub-step (checked:
 * sf_dem_overlap_batch_end fails if any atom moved more than 5 % of the skin in one sub-step). */
int sf_dem_set_overlap(void *ptr, int on, void *comm_stream);
/* Compute-unit partition for the overlapped halo: creates (once) two streams with disjoint CU masks -- the exchange
 * gets comm_cus_per_xcd (1 or 2) CUs of each of the 8 XCDs, the sub-step kernels the other 240-248 -- makes them the
 * engine's main and communication stream and returns both so the caller can enqueue its own work on them. */
int sf_dem_partition_streams(void *ptr, int comm_cus_per_xcd, void **main_stream, void **comm_stream);
int sf_dem_overlap_begin(void *ptr);
int sf_dem_substep_part(void *ptr, int part, int last, int kstep);
int sf_dem_substep_flip(void *ptr, int kstep);
int sf_dem_overlap_batch_end(void *ptr, int first_k, int launched, int last_kstep, int *trigger);
int sf_dem_boundary_count(void *ptr);

/* The per-sub-step loop of a decomposed domain queued from C++ over RCCL (replaces LAMMPS' Comm::forward_comm +
 * the MPI_Allreduce of Neighbor::decide inside Verlet::run).  sf_dem_comm_unique_id on one rank, the 128 bytes to
 * every rank by any means (MPI_Bcast, torch.distributed), sf_dem_comm_init collectively.  sf_dem_halo_run(first_k, n)
 * queues sub-steps first_k .. n-1 (fused pack, one grouped ncclSend/ncclRecv, fused unpack, kernels; overlapped if
 * sf_dem_set_overlap is on), synchronises once and returns the voted rebuild trigger like sf_dem_batch_end. */
typedef struct {
  int world;
  double shift_left, shift_right;            /* periodic shift applied to what leaves through the box faces */
  long long soff_l, soff_r;                  /* send buffer: where the left- / right-going records start (doubles) */
  long long roff_l, n_from_left;             /* receive buffer: records that came from the left neighbour */
  long long roff_r, n_from_right;
  const long long *send_off, *send_cnt;      /* per peer rank, in doubles; header word first in every chunk */
  const long long *recv_off, *recv_cnt;
  const int *dev_shdr, *dev_rhdr;            /* device tables [world]: header offsets in the send / receive buffer */
  double *dev_tx, *dev_rx;
} sf_halo_layout;
int sf_dem_comm_unique_id(char *id128);
int sf_dem_comm_init(void *ptr, const char *id128, int rank, int world);
int sf_dem_halo_run(void *ptr, int first_k, int n, const sf_halo_layout *lay, int *trigger);
/* The whole slab driver in C++ over RCCL -- what an MPI / C++ host (lammpsFoam under mpirun, one rank per GPU) calls
 * instead of lammps_step on a decomposed domain; Comm::exchange / Comm::borders / Comm::forward_comm and the
 * MPI_Allreduce's of Neighbor::decide, PairGranHookeHistory::init_one (max radius) and PairLubricatePoly::init_style
 * (particle volume, pair_lubricate_poly.cpp:540-543) of the reference's LAMMPS [3P].  1-D slabs along x:
 *   sf_dem_comm_unique_id on rank 0 -> the 128 bytes to every rank (MPI_Bcast) ->
 *   sf_slab_init(ptr, id, rank, world, xlo, xhi, periodic_x)   collective: communicator, sub-domain of this rank
 *   sf_slab_setup(ptr)      collective: global max radius, first rebuild (migration, borders, list), global particle
 *                           volume, setup forces
 *   sf_slab_step(ptr, n)    collective: "run n pre no post no"; every sub-step = fused pack, ONE grouped
 *                           ncclSend/ncclRecv (ghost records to the two face neighbours + the rebuild vote to every
 *                           rank), fused unpack, fused sub-step kernel; one host synchronisation per batch; on a
 *                           voted trigger: sf_slab_rebuild and the remaining sub-steps
 *   sf_slab_rebuild(ptr)    collective: migration of leavers with all per-atom state (fix fdrag arrays,
 *                           fix_fluid_drag.cpp:211-243, wall and pair history), size pre-exchange, border exchange,
 *                           device list build
 * Particles are handed over per rank with sf_dem_create_atoms / read_data before sf_slab_setup (every rank its own
 * slab's atoms).  The rebuild-time exchanges use ncclSend/ncclRecv on device buffers; nothing goes through Python. */
int sf_slab_init(void *ptr, const char *id128, int rank, int world, double xlo, double xhi, int periodic_x);
/* 3-D brick decomposition ([3P] `processors px py pz`; the reference's parallel cases cut two dimensions,
 * cases/example-cases/transport-bedload/system/decomposeParDict: n (14 1 6)): px * py * pz = world bricks over the
 * engine's box (set_box / `boundary` before this call), rank r owns brick (r % px, (r / px) % py, r / (px py)).
 * Collective, instead of sf_slab_init; sf_slab_setup / sf_slab_step / sf_slab_rebuild then drive the bricks:
 * migration staged over the dimensions, every ghost sent straight by its owner to the (up to 26) neighbour bricks --
 * ONE grouped ncclSend/ncclRecv per sub-step --, the rebuild vote in a header word to every rank.  Dimensions the
 * grid does not cut keep their periodic images local. */
int sf_brick_init(void *ptr, const char *id128, int rank, int world, int px, int py, int pz);
/* the exchange pattern sf_brick_init sets up for `rank`, host logic only (no device needed): blocks sent -- (peer,
 * direction code (dx+1) + 3 (dy+1) + 9 (dz+1)) in send order, blocks received -- (peer, the sender's direction code) in
 * receive order (arrays of 26), and the face neighbours [dim][low, high] used by the staged migration (-1: none) */
int sf_brick_pattern(int rank, int px, int py, int pz, const int *periodic, int *nsend, int *send_peer, int *send_code,
                     int *nrecv, int *recv_peer, int *recv_code, int *face_nbr);
int sf_slab_setup(void *ptr);
int sf_slab_rebuild(void *ptr);
int sf_slab_step(void *ptr, int n);
/* 1 when sf_slab_init / sf_brick_init has made this engine one domain of a decomposed run: sf_lammps_step(ptr, n)
 * and the `run` command then ARE sf_slab_step(ptr, n) (interfaceToLammps/library.cpp:372-386 is the same call on
 * one rank and on N) */
int sf_slab_active(void *ptr);
/* How the forward halo of this (brick) engine travels.  0: the RCCL exchange (one grouped send / receive + an unpack kernel
 * per sub-step).  1: DIRECT GHOST WRITES -- the sub-step kernel writes the border atoms' records straight into the
 * neighbour ranks' receive areas (IPC mappings of fine-grained device memory, opened at sf_brick_init / at a rebuild) and
 * one kernel per exchange publishes and awaits a flag word per rank.  2: GHOST SLOTS -- the records go straight into the
 * ghost range of the neighbours' record arrays (IPC mappings of the arrays themselves), the sub-step kernel's last wave
 * publishes vote and flag and the next sub-step kernel waits for the flags at its gate: NO kernel between two sub-step
 * kernels.  SF_HALO_DIRECT unset or 0: RCCL; 1 / 2: that transport, required (sf_brick_init or the rebuild fails on every
 * rank otherwise); "auto" / "auto2": tried with a flag round between all ranks, RCCL exchange when it does not come up on
 * every rank or is lost at a later rebuild.  A flag that does not arrive within SF_HALO_DIRECT_TIMEOUT seconds (20) is an
 * error of sf_slab_step, never a hang. */
int sf_slab_direct_halo(void *ptr);
long long sf_slab_rebuild_count(void *ptr);
/* values[0..n) summed over the ranks of the engine's communicator, in place on the host (MPI_Allreduce(MPI_SUM) of
 * interfaceToLammps/library.cpp:112-131,470-473; counts stay exact below 2^53).  Collective; after sf_slab_init /
 * sf_brick_init. */
int sf_slab_allreduce_sum(void *ptr, double *values, int n);
/* the communicator behind this engine: ranks as RCCL counts them (ncclCommCount), ncclGetVersion, and the shared
 * object the nccl* symbols were loaded from */
int sf_slab_comm_info(void *ptr, int *comm_ranks, int *rccl_version, char *lib_path, int lib_path_len);
/* while sf_dem_set_profiling is on: HIP-event time of the sampled forward exchanges (vote + ghosts: RCCL kernel and
 * unpack, measured on this rank's stream from the end of the sub-step kernel before); returns and resets the sums */
int sf_slab_exchange_profile(void *ptr, long long *exchanges, double *ms);
/* rebuilds since sf_slab_setup (its own, which allocates, not counted) and the host wall time spent in them */
int sf_slab_rebuild_profile(void *ptr, long long *rebuilds, double *ms);
int sf_slab_layout_get(void *ptr, sf_halo_layout *out);   /* the forward-halo layout of the current list (tests) */
/* owned atoms that left the slab through either face (>= 0; < 0: error): when it is 0 on every rank the migration
 * exchange of this rebuild can be skipped */
long long sf_dem_migrate_count(void *ptr);
long long sf_dem_migrate_pack(void *ptr, int side, double xshift, double *dev_buf, long long max_doubles);
int sf_dem_migrate_unpack(void *ptr, const double *dev_buf, long long ndoubles);
int sf_dem_migrate_record_doubles(void *ptr);
/* history slots carried by a migrating atom: the max over all ranks of sf_dem_info.max_neigh_used */
int sf_dem_migrate_set_slots(void *ptr, int mrec);
/* pair lubricate/poly: its isotropic resistances R0 / RT0 / RS0 depend on the volume fraction of ALL particles
 * (MPI_Allreduce of volP, pair_lubricate_poly.cpp:540-543).  On a decomposed domain the driver sums
 * sf_dem_local_particle_volume over the ranks and passes the total to every rank before sf_dem_setup
 * (sf_dem_setup fails on a decomposed domain with lubricate/poly when it has not been set). */
int sf_dem_local_particle_volume(void *ptr, double *volP);
int sf_dem_set_global_particle_volume(void *ptr, double volP);
/* the neighbour / ghost cutoff 2 r_max + skin needs the largest radius of ALL ranks ([3P] the MPI_Allreduce of
 * maxrad_dynamic in PairGranHookeHistory::init_one): MAX-reduce the local value, set it before the first rebuild */
int sf_dem_local_max_radius(void *ptr, double *rmax);
int sf_dem_set_global_max_radius(void *ptr, double rmax);
/* refresh the periodic y/z images (also of ghosts received from other GPUs) after a forward unpack */
int sf_dem_ghost_forward_local(void *ptr);
/* issue all engine work on a caller-owned hipStream_t (NULL = the legacy default stream, which is torch's
 * default; (void*)-1 = back to the engine's own stream) */
int sf_dem_set_stream(void *ptr, void *stream);

/* ------------------------------------------------------------------------------------------
 * (2) per-kernel entry points (device pointers; stream = hipStream_t or NULL)
 * ---------------------------------------------------------------------------------------- */
/* Device memory for callers that are compiled by a host compiler without the HIP headers (the LAMMPS PairStyle /
 * FixStyle and OpenFOAM dragModel adapters under adapters/): plain hipMalloc / hipMemcpyAsync / hipMemsetAsync /
 * hipStreamSynchronize.  sf_dev_alloc returns NULL on failure (see sf_last_error). */
void *sf_dev_alloc(size_t bytes);
int sf_dev_free(void *dev);
int sf_dev_upload(void *dev_dst, const void *host_src, size_t bytes, void *stream);
int sf_dev_download(void *host_dst, const void *dev_src, size_t bytes, void *stream);   /* synchronises the stream */
int sf_dev_zero(void *dev, size_t bytes, void *stream);
int sf_dev_sync(void *stream);
typedef struct {
  double kn, kt, gamman, gammat, xmu;
  int dampflag;
} sfk_gran_params;
/* PairGranHertzFixHistory::settings  pair_gran_hertzFix_history.cpp:293-317 */
int sfk_gran_settings(sfk_gran_params *p, double kn, int kt_null, double kt, double gamman,
                      int gammat_null, double gammat, double xmu, int dampflag, double nktv2p);

/* PairGranHertzFixHistory::compute  pair_gran_hertzFix_history.cpp:45-287 on a LAMMPS-shaped
 * half list in CSR form (rows = owned atoms ilist[ii]; first[inum+1]; jlist; touch; shear[3*])
 * and AoS atom arrays (double[n][3]) -- what a PairStyle adapter holds after flattening
 * NeighList.  hertz = 1: gran/hertzFix/history, 0: gran/hooke/history.  f/torque are
 * accumulated (+=) with FP64 atomics for the j side when j < nlocal. */
int sfk_pair_gran_history_compute(int hertz, const sfk_gran_params *p, double dt, int shearupdate,
                                  int nlocal, int inum, const int *ilist, const int *first,
                                  const int *jlist, int *touch, double *shear, const double *x,
                                  const double *v, const double *omega, const double *radius,
                                  const double *rmass, const int *mask, int freeze_group_bit,
                                  double *f, double *torque, void *stream);
/* the same with the fix_rigid branch of the reference (pair_gran_hertzFix_history.cpp:72-86, 182-185): mass_rigid[n] is the
 * per-atom array the pair style fills from FixRigid's "body" / "masstotal" (the mass of the atom's rigid body, 0 for a free
 * atom; NULL: no fix rigid) -- an atom with mass_rigid > 0 enters the effective mass with it instead of rmass.  [3P] fix
 * rigid itself (the body integration) is LAMMPS' and outside this library. */
int sfk_pair_gran_history_compute_rigid(int hertz, const sfk_gran_params *p, double dt, int shearupdate,
                                        int nlocal, int inum, const int *ilist, const int *first,
                                        const int *jlist, int *touch, double *shear, const double *x,
                                        const double *v, const double *omega, const double *radius,
                                        const double *rmass, const int *mask, int freeze_group_bit,
                                        double *f, double *torque, const double *mass_rigid, void *stream);
/* FixCohe::post_force  fix_cohesive.cpp:138-263 (half list, CSR) */
int sfk_fix_cohesive_post_force(double ah, double lam, double smin, double smax, int opt,
                                int nlocal, int newton_pair, const int *ilist, const int *first,
                                const int *jlist, const double *x, const double *radius,
                                const int *mask, int groupbit, double *f, void *stream);
/* PairLubricatePoly::compute  pair_lubricate_poly.cpp:65-444 (full list, CSR, no shearing) */
typedef struct {
  double mu;
  int flaglog, flagfld, flagHI, flagVF;
  double cut_inner, cut_global, R0, RT0, RS0, vxmu2f;
} sfk_lub_params;
int sfk_pair_lubricate_poly_compute(const sfk_lub_params *p, int inum, const int *ilist,
                                    const int *first, const int *jlist, const double *x,
                                    const double *v, const double *omega, const double *radius,
                                    double *f, double *torque, void *stream);
/* FixWallGranFix::post_force  fix_wall_granFix.cpp:247-345 with the laws of :361-678, a wall at rest (AoS [n][3];
 * pairstyle: the reference's enum 0 hooke / 1 hooke/history / 2 hertz/history; wallstyle 0 / 1 / 2: x / y / z plane pair
 * [lo, hi], 3: z cylinder of radius cylradius; shear[n][3]: the fix's per-atom history, read and updated; f / torque +=) */
int sfk_fix_wall_granfix_post_force(int pairstyle, const sfk_gran_params *p, int wallstyle, double lo, double hi,
                                    double cylradius, double dt, int shearupdate, int nlocal, const double *x,
                                    const double *v, const double *omega, const double *radius,
                                    const double *rmass, const int *mask, int groupbit, double *shear, double *f,
                                    double *torque, void *stream);
/* FixFluidDrag::post_force  fix_fluid_drag.cpp:114-164 (AoS [n][3]) */
int sfk_fix_fluid_drag_post_force(int nlocal, double dt, double carrier_rho, const double *v,
                                  const double *rmass, const double *radius, const int *mask,
                                  int groupbit, const double *ffluiddrag, const double *DuDt,
                                  double *vOld, double *f, void *stream);
/* dragModel::Jd  ErgunWenYu.C:86-145 (model 0) / SyamlalOBrien.C:85-144 (model 1) / NoCorrection.C:85-146 (model 2) */
int sfk_drag_model_jd(int model, int n, const double *Ur, const double *alpha, const double *pd,
                      double nuf, double rhof, double *Jd, void *stream);
/* cell owner of a uniform blockMesh hex block (the result of softParticle::move tracking,
 * softParticle.C:102-151): cell = ix + nx*(iy + ny*iz), -1 outside.  x AoS [n][3] */
int sfk_cell_owner(int n, const double *x, const double origin[3], const double dx[3],
                   const int ncell[3], int *cell, void *stream);
/* the same on a graded (rectilinear) block: dev_faces[k] = DEVICE array of ncell[k]+1 face coordinates, or NULL =
 * uniform along k (origin/dx) */
int sfk_cell_owner_graded(int n, const double *x, const double origin[3], const double dx[3],
                          const int ncell[3], const double *const dev_faces[3], int *cell, void *stream);

/* ------------------------------------------------------------------------------------------
 * (3) enhancedCloud surface -- lammpsFoam/enhancedCloud.H:183-249, enhancedCloud.C
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  /* constant/cloudProperties + transportProperties keys read at enhancedCloud.C:573-608 */
  int dragModel;          /* 0 ErgunWenYu, 1 SyamlalOBrien, 2 NoCorrection  (cloudProperties: dragModel) */
  int subCycles;
  int particleDrag, particlePressureGrad, particleBuoyancy, particleAddedMass, particleLift,
      lubricationForce;
  double gravity[3];      /* cloudProperties: g */
  double rhob, nub;       /* transportProperties */
  double maxPossibleAlpha;
  /* diffusion-based coarse graining (enhancedCloud::smoothField, enhancedCloud.C:790-907; keys read at
   * :564-583 and createFields.H:126-149): band width b (tau = b^2/4), implicit steps, which fields, and the
   * diagonal of smoothDirection.  diffusionBandWidth <= 0 or diffusionSteps <= 0: no smoothing. */
  double diffusionBandWidth;
  int diffusionSteps;
  int UfSmooth, UpSmooth, dragSmooth, alphaSmooth;
  double smoothDirection[3];
  int particleHistoryForce;   /* reduced-order Basset history force, enhancedCloud.C:197-233 */
  /* the inlet override of updateDragOnParticles (enhancedCloud.C:249-257; keys read at :600-608 and
   * softParticleCloud.C:460-486): with addParticleOption 1 (box) or 2 (hollow cylinder) and a non-zero inletForce, a
   * particle inside inletBox gets pDrag = m (inletForce - U) / deltaT instead of the assembled force.  inletBox holds
   * the tensor's nine components x1 x2 y1 y2 z1 z2 r1 r2 (unused) as pointInRegion reads them
   * (softParticleCloud.C:1354-1417); eccentricity shifts the inner cylinder.  The add / delete schedules that
   * addParticleOption also switches on in the reference are not part of this library. */
  int addParticleOption;
  double inletForce[3];
  double inletBox[9];
  double eccentricity[3];
} sf_cloud_props;
/* uniform hex block mesh (blockMeshDict: hex (...) (nx ny nz) simpleGrading (1 1 1)) */
typedef struct {
  double origin[3], dx[3];
  int n[3];
  /* blockMesh simpleGrading: faces[k] = the n[k]+1 ascending face coordinates along axis k (host pointer, copied), or
   * NULL = uniform cells origin[k] + i dx[k].  The mesh stays a tensor product: cell (ix, iy, iz) = ix + nx*(iy + ny*iz)
   * spans faces[0][ix..ix+1] x faces[1][iy..iy+1] x faces[2][iz..iz+1]. */
  const double *faces[3];
  /* multi-block blockMesh cases: OpenFOAM numbers the cells block by block.  cell_label[ix + nx*(iy + ny*iz)] = the
   * OpenFOAM label of that cell of the merged rectilinear grid (host pointer to a permutation of 0..ncells-1, copied),
   * NULL = identity.  Every host array that crosses this interface (set_fluid, get_fields, get_particles' cell,
   * smooth_field) is then in label order; sf_cloud_device_fields stays in grid order. */
  const int *cell_label;
  /* cyclic patch pairs of the diffusion mesh: periodic[k] != 0 couples the first and the last cell layer along axis k
   * in smoothField (the reference's channel cases: blockMeshDict `cyclic` patches, in.lammps `boundary pp ff pp`);
   * 0 = zeroGradient at both ends (enhancedCloud.C:800-815 default patch type) */
  int periodic[3];
  /* Mesh partitioned into x-slabs by the planes of the particle decomposition (one slab per GPU): this block is then
   * ONE slab -- n[0] = its owned cell layers + one ghost layer on each side, origin[0] = the low face of the low ghost
   * layer, uniform along x -- and slab_nx_global (> 0) is the number of cell layers of the WHOLE mesh along x
   * (periodic[0] its cyclic pair).  0: the block is the whole mesh.  Ghost layers receive what this rank's particles
   * deposit just outside its slab (they drift by up to skin/2 between rebuilds); the caller adds them to the neighbour's
   * edge layers and refreshes them (sedifoam_amd/cloud.py: SlabCloud). */
  int slab_nx_global;
} sf_cloud_mesh;

int sf_cloud_create(void *lmp, const sf_cloud_mesh *mesh, const sf_cloud_props *props,
                    double deltaT, void **cloud);
int sf_cloud_destroy(void *cloud);
/* fluid-side inputs, host AoS [ncells][3] (Uf, DDtUf, gradp = fvc::grad(p), curlU = fvc::curl(Uf));
 * any pointer may be NULL = keep previous (zero initially) */
int sf_cloud_set_fluid(void *cloud, const double *Uf, const double *DDtUf, const double *gradp,
                       const double *curlU);
/* enhancedCloud::evolve()  enhancedCloud.C:669-787 */
int sf_cloud_evolve(void *cloud);
/* enhancedCloud::calcTcFields()  enhancedCloud.C:316-441 */
int sf_cloud_calc_tc_fields(void *cloud);
/* stand-alone smoothField on a host field [ncells][ncomp] (ncomp 1 or 3), in place */
int sf_cloud_smooth_field(void *cloud, double *field, int ncomp);
/* evolve() / calcTcFields() in pieces, for a decomposed domain (one engine per GPU, every rank holds the whole mesh):
 *   phase 0 next time step + UfSmoothed (phase 6: UfSmoothed only, for initialisation) ; per sub-cycle: phase 1 drag on this rank's particles -> the caller runs subSteps DEM
 *   sub-steps through the halo driver -> (first sub-cycle) phase 2 per-cell sums of this rank's particles -> the
 *   caller adds gamma [ncells] and Ue [ncells][3] over the ranks in place -> phase 3 smoothing + Ue/gamma.
 *   calcTcFields: phase 4 (alpha cap, local Asrc sums) -> add Asrc [ncells][3] over the ranks -> phase 5.
 * sf_cloud_device_fields returns the device arrays to reduce (MPI_Allreduce / ncclAllReduce / torch.distributed). */
int sf_cloud_phase(void *cloud, int phase);   /* 0: done ; 1 (slab mesh): paused for the x solve, see below */
/* Slab mesh: the implicit diffusion solve of smoothField is global along x.  A smoothing phase (0 / 6, 3, 5) does the
 * local half (transforms along y and z) and returns 1; the caller transposes the planar work array
 * [nfields][nz][ny][n[0]] (owned columns 1 .. n[0]-2) to complete x-lines -- line number = (field * nz + iz) * ny + iy,
 * an all-to-all of local size -- runs sf_cloud_smooth_xsolve on its share of the lines ([nlines][slab_nx_global],
 * first_line = the number of its first line), transposes back so that every rank also gets the two columns next to
 * its slab (its ghost layers), and calls the same phase again. */
int sf_cloud_smooth_work(void *cloud, double **dev_work, int *nfields);
/* The two exchanges of a slab mesh over the engine's own RCCL communicator (the one sf_slab_init made), on the engine's
 * stream -- what the host otherwise does between the sf_cloud_phase calls with MPI:
 *   sf_cloud_slab_halo_add(cloud, fields)  fields = bit 0 gamma | bit 1 Ue | bit 2 Asrc: what this rank's particles
 *       deposited in its two ghost layers is sent to the face neighbours and added to their edge layers, then the ghost
 *       layers are refreshed with the neighbours' edge values (two face messages each way per field);
 *   sf_cloud_slab_phase(cloud, phase)      sf_cloud_phase(phase); when it pauses for the x solve: the all-to-all that
 *       turns the planar work array into complete x-lines, sf_cloud_smooth_xsolve on this rank's share, the all-to-all
 *       back (every rank also gets the two columns next to its slab), and the phase again.
 * Cells are partitioned by the same planes as the particles (mesh.n[0] - 2 owned layers per rank). */
int sf_cloud_slab_halo_add(void *cloud, int fields);
int sf_cloud_slab_phase(void *cloud, int phase);
int sf_cloud_slab_info(void *cloud, void **lammps, int *n3, int *nx_global, int *periodic_x, int *smoothing);
int sf_cloud_smooth_xsolve(void *cloud, double *dev_lines, long long nlines, long long first_line);
int sf_cloud_sub_cycling(void *cloud, int *subCycles, int *subSteps);
int sf_cloud_device_fields(void *cloud, double **gamma, double **Ue, double **Asrc, int *ncells);
/* accessors: gamma (alpha) [ncells], Ue [ncells][3], Asrc [ncells][3], Omega [ncells] to host */
int sf_cloud_get_fields(void *cloud, double *gamma, double *Ue, double *Asrc, double *Omega);
/* per-particle results of the last evolve sub-cycle in tag order (n = particle count):
 * cell [n], pDrag [n][3], Jd [n] ; any may be NULL */
int sf_cloud_get_particles(void *cloud, int *tag, int *cell, double *pDrag, double *Jd);
int sf_cloud_particle_count(void *cloud);
/* enhancedCloud::averageInfo (enhancedCloud.C:1341-1370): out[0] total particle volume, out[1..3] sum of Vol*U,
 * out[4..6] volume-averaged particle velocity (of this engine's particles; a decomposed run adds out[0..3] over ranks) */
int sf_cloud_average_info(void *cloud, double *out7);
/* softParticleCloud::adjustLampTimestep  softParticleCloud.C:209-261 (done by sf_cloud_create;
 * exposed for tests) */
int sf_cloud_adjust_timestep(double deltaT, double dtLampIn, int subCycles_in, double *dtLampAdj,
                             int *solidStepsPerDt, int *subCycles, int *subSteps);
/* timers in the reference's buckets (writeCPUTime.H:1-19): seconds since creation */
typedef struct {
  double evolve, calcTc, dragOnParticles, lammps, particleMove, scatter;
} sf_cloud_timers;
int sf_cloud_get_timers(void *cloud, sf_cloud_timers *t);

#ifdef __cplusplus
}
#endif

#ifdef SEDIFOAM_AMD_LAMMPS_NAMES
/* drop-in spelling of interfaceToLammps/library.h for softParticleCloud.C */
static inline void lammps_open(int a, char **b, intptr_t c, void **p) { sf_lammps_open(a, b, c, p); }
static inline void lammps_close(void *p) { sf_lammps_close(p); }
static inline void lammps_file(void *p, char *s) { sf_lammps_file(p, s); }
static inline char *lammps_command(void *p, char *s) { return (char *)sf_lammps_command(p, s); }
static inline void lammps_sync(void *p) { sf_lammps_sync(p); }
static inline int lammps_get_global_n(void *p) { return sf_lammps_get_global_n(p); }
static inline void lammps_get_initial_np(void *p, int *n) { sf_lammps_get_initial_np(p, n); }
static inline void lammps_get_initial_info(void *p, double *c, double *v, double *d, double *r,
                                           int *t, int *l, int *ty)
{ sf_lammps_get_initial_info(p, c, v, d, r, t, l, ty); }
static inline int lammps_get_local_n(void *p) { return sf_lammps_get_local_n(p); }
static inline void lammps_get_local_domain(void *p, double *d) { sf_lammps_get_local_domain(p, d); }
static inline void lammps_get_local_info(void *p, double *c, double *v, int *f, int *l, int *t)
{ sf_lammps_get_local_info(p, c, v, f, l, t); }
static inline void lammps_put_local_info(void *p, int n, double *fd, double *du, int *f, int *t)
{ sf_lammps_put_local_info(p, n, fd, du, f, t); }
static inline void lammps_step(void *p, int n) { sf_lammps_step(p, n); }
static inline void lammps_set_timestep(void *p, double dt) { sf_lammps_set_timestep(p, dt); }
static inline double lammps_get_timestep(void *p) { return sf_lammps_get_timestep(p); }
static inline void lammps_create_particle(void *p, int n, double *x, double *t, double d,
                                          double r, int ty, double *v)
{ sf_lammps_create_particle(p, n, x, t, d, r, ty, v); }
static inline void lammps_delete_particle(void *p, int *l, int n) { sf_lammps_delete_particle(p, l, n); }
#endif

#endif /* SEDIFOAM_AMD_H */
