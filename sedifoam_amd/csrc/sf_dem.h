// sf_dem.h -- device-resident DEM engine: what `lammps_step(n)` does to the particles, on one GPU.
//
// State layout in HBM (all FP64, capacity `cap` particle slots = owned atoms then ghost atoms):
//   xr[2], vm[2], om[2] : double4 records (x,y,z,radius) (vx,vy,vz,rmass) (wx,wy,wz,0), ping-pong:
//                         sub-step k reads buffer k&1 (own row coalesced, neighbour rows gathered) and
//                         writes buffer (k+1)&1, so force, final_integrate(k) and initial_integrate(k+1)
//                         fuse into ONE kernel with no read/write hazard.
//   neigh   [M][cap] int32 : full neighbour list, slot-major (lane i reads consecutive addresses);
//                            bit 30 of an entry = "touching" flag (LAMMPS keeps it in a separate
//                            touch[] array, pair_gran_hertzFix_history.cpp:135,212)
//   shear[2][M][3][cap] f64: per-slot shear history, same slot-major layout, ONE copy per contact (see kOwnBit),
//                            ping-pong like the records: the partner side gathers the owner's previous value
//   fdrag/DuDt/vOld [3][cap], wall shear [nwall][3][cap], xhold [3][cap], force/torque double4.
// The list is FULL (every owned atom lists all its neighbours; each contact is evaluated from both
// sides, bit-for-bit antisymmetric) so no atomics are needed and results are run-to-run
// deterministic.  This is the reference's own `newton off` semantics for owned-ghost pairs
// (pair_gran_hertzFix_history.cpp:273) applied to every pair.  The shear HISTORY of a pair of two
// atoms of this GPU is nevertheless kept once, like the reference's half list does: by the atom with the
// lower index (the "owner" of the pair; its list word carries kOwnBit).  The other side ("partner") reads
// the owner's previous value, negates it and repeats the same update in registers without storing it.
// Pairs with a periodic image or with a ghost of another GPU keep a copy on each side, as in the reference.
#pragma once
#include <map>
#include <string>
#include <vector>
#include <cstdint>
#include <string>
#include <vector>

#include "sf_common.h"
#include "sf_physics.h"

namespace sf {

// neighbour word: bit 30 = the reference's touch[] flag, bit 31 = this side stores the pair's history (kOwnBit; a
// partner-side word carries the owner's slot in bits 25-29 instead of an image code).  Root mode (default): bits 0-24 = index of the ROOT atom (an
// owned atom, or a ghost owned by another GPU), bits 25-29 = periodic image code (sx+1) + 3(sy+1) + 9(sz+1) of the
// neighbour relative to its root, 13 = the root itself: the kernel gathers the root's record and adds the shift, so
// periodic images are never materialised between rebuilds (no forward copy per sub-step).  Index mode (LDS-staged
// kernel): bits 0-29 = index of the owned or ghost atom.
constexpr int kNeighMask = 0x3FFFFFFF;
constexpr int kTouchBit = 0x40000000;
constexpr int kOwnBit = (int)0x80000000;   // this side stores the pair's shear history
constexpr int kIdxBits = 25;
constexpr int kIdxMask = (1 << kIdxBits) - 1;
constexpr int kNoShift = 13;
__host__ __device__ __forceinline__ int neigh_index(int word, int roots)
{
  return roots ? (word & kIdxMask) : (word & kNeighMask);
}
constexpr int kMaxWalls = 6;

enum Flag {
  F_TRIGGER = 0,    // smallest sub-step index whose new positions exceeded skin/2 (INT_MAX: none)
  F_NEIGH_OVER,     // max numneigh wanted when the per-atom slot count M overflowed
  F_GHOST_COUNT,    // ghost atoms appended so far in this rebuild
  F_GHOST_OVER,     // capacity overflow during ghost creation
  F_MAXNEIGH,       // max numneigh of the built list
  F_LOST,           // atoms outside a non-periodic box (bins clamp them; reported)
  F_SEND_COUNT,     // scratch counter for halo packing
  F_SEND_COUNT2,
  F_STAGE_MAX,      // largest number of atoms any tile stages in LDS
  // overlapped halo (decomposed domain): kernels add their trigger to F_TRIG_LOCAL, the exchange of sub-step s
  // publishes the MIN over all ranks in F_VOTE0 + (s & 1); the kernels of sub-step s+1 test that word
  F_TRIG_LOCAL,
  F_VOTE0,
  F_VOTE1,
  F_MARGIN_FAIL,    // an atom moved more than half the list margin in one sub-step (overlap mode)
  F_PART_SLOTS,     // would-be partner sides of the last list build ...
  F_PART_COAL,      // ... and how many of them gather coalesced (k_partner_coalescing)
  F_LIST_SLOTS,     // listed neighbours of the owned atoms (same kernel) ...
  F_LIST_TOUCH,     // ... and how many of them touch
  F_MIG_TRUNC,      // a migrating atom had more history slots than the migrate record carries (an error, never truncated)
  F_GHOST_BEFORE,   // + dim: ghosts that existed before the images of periodic dimension `dim` were made (3 words)
  F_GHOST_BEFORE_Z = F_GHOST_BEFORE + 2,
  F_HALO_TIMEOUT,   // direct ghost writes: a peer's "exchange done" flag did not arrive in time (an error, never a hang)
  F_HALO_TIMEOUT_PEER,   // ... which rank's ...
  F_HALO_TIMEOUT_SEEN,   // ... and the value its flag had (adjacent words: brick_direct_probe clears the three together)
  F_PARK_OVER,      // list build: most accepted candidates of an atom when they did not fit the parking rows in LDS (BuildParams::P)
  F_ARRIVAL = 31,   // host side only: "the copy of this block has landed" (DemEngine::flags_copy_wait); never written on the device
  F_NFLAGS = 32
};

struct WallParams {
  int dim;         // 0/1/2: plane normal ; 3: z cylinder (fix_wall_granFix.cpp:107-112)
  int bit;         // group of the fix (mask bit, 1 = all)
  int post_freeze; // the fix line follows `fix freeze` in the script: it still acts on frozen atoms ([3P] Modify runs
                   // post_force in script order; cases/example-cases/transport-*/in.lammps: freeze, then wall/gran)
  double lo, hi;   // planes: the positions for THIS sub-step (a wiggling wall moves them, :259-262)
  double cylradius;
  double vwall[3]; // wall velocity for this sub-step (wiggle :263, shear :264)
  double vrot;     // z cylinder sheared about x or y: the wall ROTATES, vwall = vrot (y, -x, 0)/|xy| (:316-320)
  GranParams gp;
};

// how a wall moves (host side; WallParams carries the per-sub-step result)
struct WallMotion {
  int wiggle = 0, shear = 0, axis = 0;
  double amplitude = 0.0, period = 1.0, vshear = 0.0;
  double lo0 = 0.0, hi0 = 0.0;   // the positions of the fix command
};

constexpr int kForwardDoubles = 9;   // forward halo record: x | v | omega
constexpr int kBrickSlots = 7;       // directions that can send one atom (3 faces + 3 edges + 1 corner)
constexpr int kBlkShift = 26;        // record slot = (block << kBlkShift) | offset in doubles inside the block
constexpr int kBlkMask = (1 << kBlkShift) - 1;
// the rebuild vote in the one-double header of a forward chunk: an int in the slot's first four bytes (so that the
// kernel that finds an atom beyond skin/2 can lower it with an integer atomicMin)
__host__ __device__ inline int* header_vote_ptr(double* slot) { return reinterpret_cast<int*>(slot); }
__host__ __device__ inline int header_vote(const double* slot) { return *reinterpret_cast<const int*>(slot); }

struct DemPtrs {
  const double4* xr_in;
  const double4* vm_in;
  const double4* om_in;
  double4* xr_out;
  double4* vm_out;
  double4* om_out;
  double4* force;
  double4* torque;
  int* neigh;
  int* numneigh;
  const double* shear_in;        // history written by the previous sub-step (or the list build)
  double* shear_out;             // history after this sub-step (owner slots only)
  double* fdrag;
  double* DuDt;
  double* vOld;
  double* wshear;          // [nwall][3][cap]
  unsigned char* wtouch;   // [cap] bit w = touching wall w
  const double* xhold;     // [3][cap]
  const int* mask;
  int* flags;
  // forward halo written by the sub-step itself (StepParams::tx_fused): position of an atom in the send list of the
  // left / right face (-1: not sent) and the two send blocks (component-major, see k_forward_pack_fused)
  const int* sendslot[2];
  double* tx[2];
  double* tx_sendbuf;           // vote headers: the 8-byte slot at tx_sendbuf + tx_hdr_off[p] holds an int (header_vote)
  const int* bslot;             // brick driver (tx_fused == 2): [kBrickSlots][cap] where an atom's forward records go:
                                // (send block q << kBlkShift) | index of the record in that block (-1: no further
                                // direction sends this atom)
  double* const* tx_blkptr;     // [directions] where block q of the exchange that follows this sub-step starts: in the
                                // local send buffer, or -- direct ghost writes -- in the NEIGHBOUR's receive area (an
                                // IPC mapping; two areas, used alternately)
  const size_t* tx_blkcnt;      // [directions] records in block q.  A block is component-major, [kForwardDoubles][count]:
                                // consecutive border atoms of a wave write consecutive doubles -- whole 64-byte lines
                                // instead of one 8-byte word per 72-byte record, which is what a write over xGMI into a
                                // neighbour's uncached area is made of
  const int* tx_hdr_off;
  int* xcd_time;                // StepParams::xcd_time: [64 x + 0] first start, [64 x + 32] last end of XCD x (100 MHz clock)
  int* pq_head;                 // persistent tiles (k_substep_persist): [2][8][32] -- per launch parity and XCD one head word
                                // on a line of its own: the next tile of that XCD's range nobody has taken yet
  // ghost slots (StepParams::gs_on, sf_halo_rccl.hip, sf_dem_gs.h): the neighbours' sub-step kernels write the records of
  // this rank's ghosts straight into the ghost range of xr / vm / om.  On the sending side tx_blkptr is then [3][kMaxDirs]:
  // where block q's first x | v | omega record goes in the NEIGHBOUR's arrays (the buffer its launch of the next number
  // reads), tx_blkshift[3 q ..] the periodic shift the sender adds.  gs_sync: who to wait for and who to tell (GsSync),
  // gs_count: completion counters.
  const double* tx_blkshift;
  const struct GsSync* gs_sync;
  int* gs_my_sync;              // GsSync::my_sync (the flag / vote lines the gate polls)
  int* gs_count;                // [8][32] (eight 128-byte lines): one line per XCD, a 64-bit word each: workgroups done + 2^32 x those that triggered
  // LDS-staged tiles (k_substep_lds)
  const unsigned short* nloc;   // [M][cap] position of the neighbour in its tile's staged copy
  const int* tile_first;        // [ntiles] owned-atom range of a tile
  const int* tile_last;
  const int* stage_start;       // [ntiles+1] range of the tile in stage_idx
  const int* stage_idx;         // atom indices (owned or ghost) to stage, bin by bin
};

// ghost slots: the flag / vote lines of the ranks (device copy, set up once per communicator)
struct GsSync {
  int world, rank;
  long long max_ticks;     // 100 MHz clock: how long a wave waits for a peer's flag
  int* my_sync;            // for sender r the line [kGsStride r]: ONE 64-bit word, (flag << 32) | vote
  int* peer_sync[32];      // every rank's area as mapped here
};
constexpr int kGsStride = 32;   // ints: one 128-byte line per sending rank (one writer per line)

struct StepParams {
  int nlocal, cap, mode;   // mode 0: force + final + next initial ; 1: last (force + final, store f) ; 2: setup
  int kstep;
  int nslots;        // rows of the slot-major list arrays (M)
  int roots;         // neighbour words hold (root, image code), see kIdxMask
  double prd[3];     // box lengths (image shift = code component * prd)
  int part, nb;      // 0: every owned atom ; 1: interior atoms [n_lo, n_hi) ; 2: boundary atoms [0, n_lo) + [n_hi, nlocal)
  int n_lo, n_hi;
  int trig_test;     // flag word whose value < kstep means "list stale, do nothing"
  int trig_set;      // flag word that receives atomicMin(kstep + trig_add) when an atom exceeds skin/2
  int trig_add;
  double margin_sq;  // > 0: flag F_MARGIN_FAIL when one sub-step moves an atom by more than sqrt(margin_sq)
  double dt, trigger_sq;
  GranParams gran;
  CoheParams cohe;
  LubParams lub;
  int nwalls;
  int stage_cap;   // LDS slots per workgroup in k_substep_lds
  int xcd_remap;   // blockIdx -> contiguous chunk per XCD (8 XCDs, block b runs on XCD b % 8); 2: chunks of UNEQUAL
                   // size -- XCD x works on the xcd_count[x] blocks from xcd_first[x], the grid is 8 x the largest count
                   // and a workgroup beyond its XCD's count exits at once
  int xcd_first[8], xcd_count[8];
  int sweep_rev;   // walk each XCD's range backwards (every other sub-step)
  int xcd_time;    // this launch records when each XCD starts and ends (DemPtrs::xcd_time): the engine balances the shares
  int pq_par;      // persistent tiles: which of the two sets of head words this launch pulls from (it zeroes the other one)
  WallParams wall[kMaxWalls];
  int have_gravity;
  double gacc[3];
  int have_fdrag;
  double carrier_rho;
  int have_nve;
  // LAMMPS groups: every fix acts on the atoms whose mask has the fix's group bit (bit 0 = all).  use_groups = 0:
  // every fix is on `all`, the mask is not even read.  freeze_bit != 0: fix freeze; frozen atoms carry omega.w = 1
  int use_groups, nve_bit, grav_bit, fdrag_bit, cohe_bit, freeze_bit;
  int post_freeze;   // bit 0: fix gravity, bit 1: fix fdrag come AFTER fix freeze in the script (walls: WallParams)
  // mode 0 only: atoms with x < tx_xlo or x >= tx_xhi look their send slots up and write their new x (+ tx_shift of the
  // face), v, omega into DemPtrs::tx -- the pack kernel of the forward halo, fused
  int tx_fused, tx_nhdr;   // tx_nhdr > 0 (any part, mode 0): a trigger also lowers the tx_nhdr vote headers
                           // tx_fused == 2: brick driver -- an atom beyond tx_lo3 / tx_hi3 in some dimension looks its (up
                           // to kBrickSlots) record positions up in DemPtrs::bslot and writes its new x, v, omega there,
                           // unshifted (the receiver adds the periodic shift of the block)
  double tx_lo3[3], tx_hi3[3];
  // ghost slots: gs_on -- the ghost records of this launch's input buffers were written by the neighbours' kernels; gs_seq --
  // the number of this launch: its waves wait until every rank's flag says gs_seq ("my records for launch gs_seq are in
  // your arrays", gs_wait = 0: nobody to wait for) and read the vote that travels in the same word; the wave that stays behind
  // (sf_dem_gs.h) stores this rank's vote and the flag gs_seq + 1 into every rank's line
  int gs_on, gs_seq, gs_wait;
  int gs_world, gs_rank;   // (copies of GsSync::world / rank: the gate reads them from the kernel arguments)
  int tx_n[2];             // atoms in the left / right send list (a face's block is [kForwardDoubles][tx_n])
  double tx_xlo, tx_xhi, tx_shift[2];
};

struct BinGrid {
  double lo[3], inv[3];
  int n[3];
  int stencil;     // neighbour search radius in cells (cells may be finer than the cutoff: cell = cut / stencil)
  int nbins;       // key space: tiles * tile^3 (>= n[0]*n[1]*n[2])
  int tile;        // bins are numbered tile by tile (tile x tile x tile bins) so that particles that are
  int nt[3];       // close in space are close in memory in all three directions; tile <= 1: plain x-fastest
  int xslow;       // 1: z fastest, x slowest (decomposed domain: the atoms next to the two x faces of the slab are
                   // then a prefix and a suffix of the sorted array = the boundary part of the overlapped halo)
  int wrap[3];     // 1: the cells of this dimension tile the periodic box exactly and the list build walks its stencil
                   // AROUND the box (candidate = atom of the wrapped cell + the box length): no ghost atoms are made for
                   // the periodic images of a single-domain run (DemEngine::ghost_free_)
};

// bin coordinates -> sort key / cell index
__host__ __device__ __forceinline__ int bin_key(const BinGrid& g, int cx, int cy, int cz)
{
  if (g.xslow) return cz + g.n[2] * (cy + g.n[1] * cx);
  if (g.tile <= 1) return cx + g.n[0] * (cy + g.n[1] * cz);
  const int T = g.tile;
  const int tx = cx / T, ty = cy / T, tz = cz / T;
  const int lx = cx - tx * T, ly = cy - ty * T, lz = cz - tz * T;
  return ((tx + g.nt[0] * (ty + g.nt[1] * tz)) * T + lz) * T * T + ly * T + lx;
}

// Block-level aggregation of counters: same-address global atomics from different XCDs are resolved at the memory
// side at ~11 ns each, so per-wave atomics of a 1 M-atom kernel cost hundreds of microseconds.  1024-thread blocks,
// one global atomic per block.
__device__ __forceinline__ int block_sum_int_1024(int v)
{
  __shared__ int ws[16];
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) ws[w] = v;
  __syncthreads();
  int t = 0;
  if (threadIdx.x == 0)
    for (int k = 0; k < (int)(blockDim.x >> 6); k++) t += ws[k];
  return t;   // valid in thread 0
}

// Sub-steps are queued speculatively: the host learns that the list went stale only when it synchronises, and every
// sub-step queued behind the trigger is a wasted launch (4-5 us; on a decomposed domain a wasted forward exchange
// too, ~24 us).  Rebuilds come at regular intervals in a bed that does not change its temperature, so the queue is cut
// short of where the next trigger is expected, then fed in small pieces until it has happened.  Which sub-step
// triggers is decided on the device as before: only launch counts change, never results.  The same numbers on every
// rank (rebuilds are global events), so all ranks of a decomposed run queue the same exchanges.
struct RebuildPredictor {
  long long last = 0;      // absolute sub-step index of the last rebuild
  double interval = 0.0;   // sub-steps between rebuilds (mean of the last two estimates); 0: no history yet
  bool on = true;
  // Queue THROUGH the predicted trigger instead of stopping short of it.  On a large single domain a sub-step queued
  // behind a trigger is one early-exit launch (~5 us) while every piece ends in a host read (~25 us of idle GPU): a hot
  // bed that rebuilds every ~10 sub-steps paid two or three reads per rebuild for stopping short and creeping up in
  // small pieces.  Not on small systems, where an early-exit launch costs a whole launch period (10 k grains: -5 %),
  // not on a decomposed domain, where the sub-step behind a trigger still runs its halo exchange, and not when rebuilds
  // are rare (interval > 32: the pieces around the trigger are 16 long, as dear as the reads they would save).
  bool overshoot = false;
  void rebuilt(long long step)
  {
    if (step > last) {
      const double d = (double)(step - last);
      interval = interval > 0.0 ? 0.5 * (interval + d) : d;
    }
    last = step;
  }
  // a rebuild outside the stepping loop (particles created / deleted, an external sf_slab_rebuild, setup): the
  // displacements start from zero again, the interval estimate stands
  void external(long long step) { last = step; }
  // how many of the `remaining` sub-steps of a run to queue now; `step` = absolute index of the first of them
  int chunk(long long step, int remaining) const
  {
    if (!on || interval <= 0.0 || remaining <= 4) return remaining;
    int small = (int)(interval / 8.0);
    small = small < 4 ? 4 : (small > 16 ? 16 : small);
    const double left = interval - (double)(step - last);   // predicted sub-steps until the trigger
    int c;
    if (left >= (double)(remaining + small)) c = remaining;           // not expected within this run
    else if (overshoot && interval <= 32.0 && left >= -(double)small)
      c = (int)(left > 0.0 ? left : 0.0) + small;                     // hot bed: run through it
    else if (left > 2.0 * small) c = (int)left - small;               // stop short of it
    else if (left >= -(double)small) c = small;                       // around it: small pieces
    else c = (int)(-left / 2.0) > small ? (int)(-left / 2.0) : small; // overdue (the bed calmed down): lengthen again
    return c < remaining ? (c < 1 ? 1 : c) : remaining;
  }
};

class DemEngine {
 public:
  DemEngine();
  ~DemEngine();

  // ---- configuration (input-script surface) ----
  void set_box(const double lo[3], const double hi[3]);
  void set_periodic(int px, int py, int pz);
  void create_atoms(int n, const double* x, const double* v, const double* omega,
                    const double* diameter, const double* density, const int* tag, const int* type);
  void set_pair_gran(int style, double kn, bool kt_null, double kt, double gamman, bool gammat_null,
                     double gammat, double xmu, int dampflag);
  void set_pair_lubricate(double mu, int flaglog, int flagfld, double cut_inner, double cut_global,
                          int flagHI, int flagVF);
  void set_cohesive(double ah, double lam, double smin, double smax, int opt, int groupbit = 1);
  void set_gravity(double mag, double gx, double gy, double gz, int groupbit = 1);
  void set_fdrag(double carrier_rho, int groupbit = 1);
  void set_freeze(int groupbit);                      // [3P] fix freeze
  // [3P] group command: styles type / subtract / union / intersect (what the reference's cases use)
  int group_bit(const std::string& name) const;
  void group_type(const std::string& name, int op, int v1, int v2, const std::vector<int>& list);
  void group_combine(const std::string& name, int mode, const std::vector<std::string>& args);
  void set_velocity_group(int groupbit, double vx, double vy, double vz);
  // the wall registered last: `zcylinder radius` instead of a plane pair / wiggle (kind 1: axis amplitude period) or
  // shear (kind 2: axis vshear)
  void wall_cylinder(double radius);
  void wall_motion(int kind, int axis, double a, double b);
  void add_wall(int dim, bool lo_null, double lo, bool hi_null, double hi, double kn, bool kt_null,
                double kt, double gamman, bool gammat_null, double gammat, double xmu, int dampflag,
                bool granfix, int groupbit = 1, bool cylinder = false);
  void set_nve_sphere(int groupbit = 1)
  {
    have_nve_ = true;
    nve_bit_ = groupbit;
    use_groups_ = use_groups_ || groupbit != 1;
  }
  void set_skin(double s) { skin_ = s; }
  void set_timestep(double dt) { dt_ = dt; }
  double timestep() const { return dt_; }
  void set_max_neigh(int m);
  void set_velocity_all(double vx, double vy, double vz);
  void set_subdomain(int rank, int nranks, double sublo, double subhi);
  void set_subdomain3(int rank, int nranks, const double lo[3], const double hi[3], const int ext[3]);

  // ---- stepping ----
  void setup();            // first run: build list, forces with shearupdate = 0
  // pair lubricate/poly's volume fraction needs the particle volume of ALL ranks (pair_lubricate_poly.cpp:540-543)
  double local_particle_volume();
  void set_global_particle_volume(double v) { global_volP_ = v; }
  // the list / ghost cutoff 2 r_max + skin uses the largest radius of ALL ranks ([3P] MPI_Allreduce of maxrad_dynamic
  // in PairGranHookeHistory::init_one)
  double local_max_radius() const { return rmax_; }
  void set_global_max_radius(double r) { rmax_ = r > rmax_ ? r : rmax_; }
  void run(int nsteps);    // "run n pre no post no"
  void run_begin();
  void substep(bool last);
  // batch mode of the multi-rank driver: sub-step with an explicit index for the stale-list early exit, and the
  // end-of-batch bookkeeping (returns the trigger index, INT_MAX if none; fixes the ping-pong parity)
  void substep_k(bool last, int kstep);
  int batch_end(int first_k, int launched);
  // overlapped halo (decomposed domain, see sf_dem_halo.hip): boundary atoms first, the exchange of their new
  // records runs on comm_stream while the interior atoms are advanced on the main stream
  void set_overlap(bool on, hipStream_t comm_stream);
  void make_partitioned_streams(int comm_cus_per_xcd, hipStream_t* main_out, hipStream_t* comm_out);
  void overlap_begin();                         // after run_begin: reset the local / voted trigger words
  void substep_part(int part, bool last, int kstep);   // part 2 = boundary, 1 = interior; same buffers for both
  void substep_flip(int kstep);                 // images of owned atoms + buffer parity, after both parts
  int overlap_batch_end(int first_k, int launched, int last_kstep);
  int boundary_count() const { return nb_; }
  bool overlap() const { return overlap_; }
  hipStream_t comm_stream() const { return comm_stream_; }
  void set_flag_buffer(int* dev);
  bool need_rebuild();
  void rebuild_begin();
  void rebuild_sort();
  void rebuild_finish();
  void set_in_run(bool on) { in_run_ = on; }   // the driver's stepping loop brackets its rebuilds with it
  // ... through this guard: a rebuild that throws (lost atoms, list overflow, an RCCL error) must not leave the engine
  // marked "inside a run" -- later out-of-run rebuilds would skip the force / torque permutation
  struct InRunGuard {
    explicit InRunGuard(DemEngine& e) : e_(&e) { e.in_run_ = true; }
    ~InRunGuard() { release(); }
    void release()
    {
      if (e_) e_->in_run_ = false;
      e_ = nullptr;
    }
    InRunGuard(const InRunGuard&) = delete;
    InRunGuard& operator=(const InRunGuard&) = delete;

   private:
    DemEngine* e_;
  };

  // ---- data exchange ----
  int nlocal() const { return nlocal_; }
  int nghost();   // (ghost-free list build: the periodic images are counted on demand, see ghost_free_)
  int rank() const { return rank_; }
  int nranks() const { return nranks_; }
  void sublo_hi(double out[6]) const;
  void box(double lo[3], double hi[3], int periodic[3]) const
  {
    for (int k = 0; k < 3; k++) {
      lo[k] = boxlo_[k];
      hi[k] = boxhi_[k];
      periodic[k] = periodic_[k];
    }
  }
  void get_local_info(double* x, double* v, int* foamCpuId, int* tag);
  void get_initial_info(double* x, double* v, double* diam, double* rho, int* tag, int* type);
  void put_local_info(int n, const double* fdrag, const int* foamCpuId, const int* tagIn);
  void get_forces(double* f, double* torque, double* omega, int* tag);
  long long get_history(long long max, int* tag_i, int* tag_j, double* shear);
  void get_wall_shear(int w, double* shear);
  void create_particles(int np, const double* pos, const double* tag, double diameter, double rho,
                        int type, const double* vel);
  void delete_particles(const int* tags, int n);

  // halo (device buffers)
  long long border_pack(int side, double xshift, double* buf, long long max_atoms);
  void border_pack_both(double xshift0, double* buf0, double xshift1, double* buf1, long long max_atoms, long long* n0,
                        long long* n1);
  void border_unpack(int side, const double* buf, long long natoms);
  long long forward_pack(int side, double xshift, double* buf);
  void forward_unpack(int side, const double* buf, long long natoms);
  void forward_pack2(double shift0, double* buf0, double shift1, double* buf1, long long* n0, long long* n1);
  void forward_unpack2(const double* buf0, long long n0, const double* buf1, long long n1);
  void forward_pack_fused(double shift0, long long off0, double shift1, long long off1, const int* hdr_off, int nhdr,
                          double* sendbuf);
  void forward_unpack_fused(const double* recvbuf, long long off0, long long n0, long long off1, long long n1,
                            const int* hdr_off, int nhdr, int kstep);
  // The sub-step kernel writes the forward records of the border atoms itself (no pack kernel between a sub-step and
  // the exchange that follows it): tx0 / tx1 = send buffers of the left / right face in border-list order.  Valid
  // until the next rebuild.  forward_tx_written(): the last sub-step launched did so.
  // sendbuf + hdr_off[p] (p < nhdr): the vote header of the chunk for rank p (see header_vote); a kernel whose atom
  // moves beyond skin/2 lowers the headers together with its trigger word.
  void set_forward_tx(double* tx0, double shift0, double* tx1, double shift1, double* sendbuf, const int* hdr_off,
                      int nhdr);
  bool forward_tx_written() const { return tx_written_; }
  bool profiling() const { return profiling_; }
  // ---- 3-D brick decomposition (sf_brick_*, csrc/sf_brick_rccl.hip): up to 26 send directions instead of two x faces.
  // A direction d = (dx, dy, dz), components in {-1, 0, 1}, names the neighbour brick the atoms go to; an owned atom
  // belongs to it when it lies within the ghost cutoff of EVERY face d points through (a corner atom is sent in 7
  // directions).  shift = what is added to its position so that it lands in the receiver's frame (a box length across
  // a periodic face of the global box).  All ghosts come straight from their owners: one exchange stage per sub-step.
  static constexpr int kMaxDirs = 26;
  struct BrickBlocks {             // blocks of one exchange, passed by value to the pack / unpack kernels
    int n;
    int first[kMaxDirs + 1];       // send: positions in the concatenated send list; receive: ghost slots after nlocal
    long long off[kMaxDirs];       // where the block's records start in the send / receive buffer (doubles)
    double shift[kMaxDirs][3];     // send only
  };
  void brick_set_dirs(int ndir, const int* d3, const double* shift3);
  void brick_border_select(long long* counts);             // fills the send lists; counts[ndir]
  void brick_border_pack(double* buf);                     // every block in direction order, kBorderDoubles per atom
  void brick_ghost_unpack(const double* buf, long long natoms);   // border records -> external ghosts (appended)
  void brick_forward_pack(const BrickBlocks& snd, double* sendbuf, const int* hdr_off, int nhdr);
  // the sub-step kernel writes the forward records itself from now on (until the next rebuild): record positions of
  // every sent atom from the blocks' offsets in the send buffer.  direct_blk != null: [2][kMaxDirs] block starts in the
  // neighbours' receive areas (parity of the exchange), the vote headers are then not the kernel's business
  void brick_set_forward_tx(const BrickBlocks& snd, double* sendbuf, const int* hdr_off, int nhdr,
                            double* const* direct_blk = nullptr);
  void brick_forward_unpack(const BrickBlocks& rcv, const double* recvbuf, const int* hdr_off, int nhdr);
  // direct ghost writes (sf_halo_rccl.hip): which of the two receive areas the exchange after the next sub-step uses
  void set_tx_parity(int par) { tx_par_ = par & 1; }
  bool tx_direct() const { return tx_direct_; }
  int halo_timeout() const { return h_flags_[F_HALO_TIMEOUT]; }   // (after a synchronising flag read: batch_end)
  int halo_timeout_peer() const { return h_flags_[F_HALO_TIMEOUT_PEER]; }
  int halo_timeout_seen() const { return h_flags_[F_HALO_TIMEOUT_SEEN]; }
  // one kernel instead of {RCCL send/recv, unpack}: tell every rank "my records of exchange `seq` are in your area"
  // (vote first, then the flag, system-scope release), wait for the same word from every rank (bounded: F_HALO_TIMEOUT),
  // lower the trigger word to the smallest vote and move the received records into the ghost slots
  static constexpr int kSyncStride = 32;   // ints: one 128-byte line per sending rank
  struct DirectSync {
    int world, rank, seq, par;   // (seq: the exchange number as the flag words hold it -- 32 bits, compared as such)
    int* my_sync;          // this rank's area (fine-grained): for sender r the line [kSyncStride r]: flag, vote[2]
    int* peer_sync[32];    // every rank's area, mapped (own entry unused)
    long long max_ticks;   // 100 MHz clock
  };
  void brick_direct_unpack(const BrickBlocks& rcv, const double* recvarea, const DirectSync& D);
  bool brick_direct_probe(const BrickBlocks& none, const DirectSync& D);   // bring-up: one flag round, no records; synchronises;
                                                                            // false = a peer's flag did not arrive in time
  // ---- ghost slots (SF_HALO_DIRECT=2, sf_halo_rccl.hip): NO kernel between two sub-step kernels.  The neighbours'
  // sub-step kernels write the records of this rank's ghosts straight into the ghost range of its record arrays (IPC
  // mappings of xr / vm / om themselves); a sub-step kernel waits at its gate for every rank's flag (DemPtrs::gs_sync)
  // and one wave that stays behind publishes this rank's vote and flag.  Launches are numbered by gs_seq(): the same
  // number on every rank (all ranks queue the same launches).
  void gs_configure(const GsSync& sync, long long first_seq);   // once per communicator: device copy of the sync table,
                                                                // counters; first_seq: the number of the first launch
  // the record arrays a launch whose number has parity `par` READS (what the neighbours must write the ghosts of that
  // launch into), as things stand now: valid until the next rebuild (a rebuild may swap allocations, a trigger shifts the
  // parity of the buffers against the launch numbers)
  void gs_input_arrays(int par, void** x, void** v, void** w) const;
  // after every rebuild: per send block q and launch parity, where its first x | v | omega record goes in the neighbour's
  // arrays: blk6[(par * 3 + a) * kMaxDirs + q]
  void brick_set_forward_gs(const BrickBlocks& snd, double4* const* blk6);
  // false when the buffers were flipped since then by something that is not a numbered launch (the setup evaluation):
  // the neighbours' tables then point at the wrong buffer of this rank and must be made again
  bool gs_mapping_valid() const { return gs_ready_ && ((cur_ ^ (int)(gs_seq_ & 1)) == gs_map_base_); }
  void gs_off();                                     // back to the other transports (areas gone)
  // can the sub-step kernel write the border records itself (what ghost slots rest on)?  Not in a brick thinner than twice
  // the ghost cutoff (brick_set_forward_tx) and not with SF_HALO_FUSED_PACK=0
  bool brick_fused_pack_possible() const;
  bool gs_on() const { return gs_ready_; }
  long long gs_seq() const { return gs_seq_; }       // number of the NEXT sub-step launch
  // the records of every border atom as they are now, for launch gs_seq() (the start of a run, the first launch after a
  // rebuild: no sub-step kernel has written them), and this rank's flag
  void gs_pack();
  // end of a piece: wait for the flags of launch gs_seq() and fold the votes into the trigger word, unless a trigger
  // before sub-step `kstep_end` is known already (then some rank may never publish that flag)
  void gs_close(int kstep_end);
  const BrickBlocks& brick_send_blocks() const { return bsend_blocks_; }
  long long migrate_count3();      // owned atoms outside the brick in any external dimension
  long long migrate_pack_dim(int dim, int side, double shift, double* buf, long long max_doubles);
  bool brick() const { return brick_; }
  long long migrate_pack(int side, double xshift, double* buf, long long max_doubles);
  void migrate_unpack(const double* buf, long long ndoubles);
  int migrate_record_doubles() const;
  void migrate_set_slots(int mrec);
  void ghost_forward_local();

  // Per-atom state of a client (the cloud's previous velocity and Basset-history sums: softParticle.H:95-107) that must
  // stay with its atom: up to kMaxExtra rows of doubles that are permuted by every re-sort and travel in the migrate
  // record, like a LAMMPS fix's per-atom arrays do through copy_arrays / pack_exchange.  init[r] is what a row holds
  // for an atom that did not exist before (created later).  Returns the first row index.
  static constexpr int kMaxExtra = 8;
  int register_extra(int nrows, const double* init);
  void unregister_extra(int first, int nrows);   // any order: the rows are tracked in a bitmap
  int nextra() const { return nextra_; }
  double* d_extra() const { return extra_.as<double>(); }
  // device view for the cloud
  hipStream_t stream() const { return stream_; }
  // run on a caller-owned stream (e.g. torch's current stream, so RCCL traffic orders after the pack
  // kernels without host synchronisation)
  void set_stream(hipStream_t s);
  size_t capacity() const { return cap_; }
  const double4* d_xr() const { return xr_[cur_].as<double4>(); }
  const double4* d_vm() const { return vm_[cur_].as<double4>(); }
  const double4* d_om() const { return om_[cur_].as<double4>(); }
  double4* d_force() const { return force_.as<double4>(); }
  double4* d_torque() const { return torque_.as<double4>(); }
  double* d_fdrag() const { return fdrag_.as<double>(); }
  double* d_DuDt() const { return DuDt_.as<double>(); }
  double* d_vOld() const { return vOld_.as<double>(); }
  int* d_tag() const { return tag_.as<int>(); }
  int* d_type() const { return type_.as<int>(); }
  int* d_foamCpuId() const { return foamCpuId_.as<int>(); }
  int max_tag() const { return max_tag_; }

  long long nbuilds() const { return nbuilds_; }
  long long nsteps() const { return nsteps_; }
  int max_neigh_used() const { return max_neigh_used_; }
  int max_neigh_cap() const { return M_; }
  long long npairs_full();
  bool is_setup() const { return setup_done_; }
  double cutneighmax() const;
  double last_substep_ms() const { return last_substep_ms_; }
  // per-launch HIP-event timing of the fused sub-step kernel (bench.py roofline leg)
  void set_profiling(bool on);
  void get_profile(long long* launches, double* kernel_ms);
  // neighbour rebuilds inside runs while profiling was on: how many, and their summed duration on the host clock, from
  // the trigger read to the end of the last kernel of the rebuild (synchronised: only while profiling)
  void get_rebuild_profile(long long* rebuilds, double* ms);

 private:
  void ensure_capacity(size_t need);
  void alloc_all(size_t cap);
  void grow_neigh(int newM);
  DemPtrs ptrs(int in_buf) const;
  StepParams step_params(int mode, int kstep) const;
  void launch_substep(int in_buf, int mode, int kstep, int part = 0);
public:
  void launch_ghost_forward(int buf, int kstep, int phase = 0, int trig_word = F_TRIGGER, hipStream_t s = nullptr);
private:
  void launch_initial_integrate();
  void rebuild();          // rebuild_begin + rebuild_sort + rebuild_finish
  // rows = false: the slot-major history rows (numneigh, partner tags, shear) stay where they are and the list build
  // that follows reads them through hist_perm_ (a copy of perm) -- they are 3/4 of the bytes a re-sort would move
  struct RankJob {   // k_rank_permute: the sort keys, the first sorted position of every cell, the arrival order inside the cells
    const unsigned* keys;
    const int* first;
    const int* arrival;
  };
  void permute_locals(const int* perm, int n_new, bool rows = true, const RankJob* rank = nullptr);
  void migrate_compact();
  void compute_partner_tags();
  int select_locals(int mode, double bound, DevArray& list, int dim = 0);
 public:
  long long migrate_count();   // owned atoms outside [sublo, subhi): what the two migrate_pack calls would send
 private:
  void make_periodic_ghosts();
  void bin_and_build();
  void read_flags();
  void reset_flag(int idx, int value);
  void reset_flags(int idx, int count, int value);   // `count` adjacent flags, one launch
  void flags_copy_begin();   // flag words -> pinned host block, asynchronously ...
  void flags_copy_wait();    // ... and the host waits for that copy (not for the stream)
  void set_flags3(int i0, int v0, int i1, int v1, int i2, int v2);   // three (flag, value) pairs, one launch
  void compute_grid();
  double max_radius();
  void sync() const { SF_HIP(hipStreamSynchronize(stream_)); }

  hipStream_t stream_ = nullptr;
  hipStream_t own_stream_ = nullptr;
  bool external_stream_ = false;
  size_t cap_ = 0;
  int nlocal_ = 0, nghost_ = 0, next_ghost_ = 0;   // next_ghost_: external ghosts appended by border_unpack
                                                   // nghost_ < 0 (inside a rebuild only): the count is still on the device --
                                                   // make_periodic_ghosts did not wait for it, bin_and_build reads it with its flags
  bool ghost_sync_ = false;                        // bin_and_build: a deferred count overflowed the capacity -- make the ghosts
                                                   // again and wait for the count
  int cur_ = 0;
  int M_ = 32;
  int max_neigh_used_ = 0;
  int max_tag_ = 0;
  bool setup_done_ = false, have_list_ = false;
  long long nbuilds_ = 0, nsteps_ = 0;
  double last_substep_ms_ = 0.0;

  // domain
  double boxlo_[3] = {0, 0, 0}, boxhi_[3] = {1, 1, 1};
  int periodic_[3] = {0, 0, 0};
  int rank_ = 0, nranks_ = 1;
  // sub-domain of this GPU and the dimensions whose halo is EXTERNAL (ghosts come from other GPUs through the driver;
  // every other periodic dimension gets its images locally, as (root, image code) neighbour words).  Slab driver: x.
  // Brick driver (set_subdomain3): every dimension the processor grid cuts.
  double sublo_[3] = {0.0, 0.0, 0.0}, subhi_[3] = {1.0, 1.0, 1.0};
  bool ext_[3] = {false, false, false};
  bool have_subdomain_ = false;   // true: some halo is external (driven through sf_dem_border_* / sf_brick_*)
  bool brick_ = false;            // set_subdomain3: 3-D processor grid (direction lists instead of two x faces)
  // environment overrides, all measured: SF_SUB sort cells per cutoff length, SF_TILE tile-major
  // sort, SF_XCD_REMAP contiguous block range per XCD, SF_LDS the LDS-staged kernel
  int opt_tile_ = 0, opt_xcd_remap_ = 1, opt_lds_ = 0, opt_sub_ = 2;
  // sort cells per cutoff length chosen from the list statistics (0: opt_sub_): cells of the full cutoff for a loose bed,
  // whose lanes do not gather consecutive records anyway and whose rebuilds are many (choose_kernel)
  int sort_sub_ = 0;
  bool opt_sub_env_ = false;
  double xcd_weight_[8] = {1, 1, 1, 1, 1, 1, 1, 1};   // share of the sorted range each XCD works on (launch_substep)
  bool xcd_weighted_ = false;
  // XCD balance: a launch a few sub-steps after every list build is timed per XCD (two atomics per wave), and the shares
  // follow the measured rates.  Placement only: results do not depend on it.
  bool xcd_auto_ = true;
  int xcd_countdown_ = 0;          // launches until the next timed one (0: none pending)
  bool xcd_sample_pending_ = false;
  int* d_xcd_time_ = nullptr;      // [512 + 512]: the sample (two words per XCD, a cache line apart), then its initial values
  int* h_xcd_time_ = nullptr;      // pinned
  void apply_xcd_sample();
  size_t stamp_last_grid_ = 0;   // workgroups of the last k_substep launch (SF_EXP_STAMP variant builds)
  int mrec_ = 0;                   // history slots per migrating atom (global max over ranks)
  bool migrate_pending_ = false;
  int migrate_leavers_ = 0;        // atoms packed by migrate_pack since the last compaction
  DevArray leave_;                 // per owned atom: 0 stay, 1 leaves to -x, 2 leaves to +x
  double skin_ = 0.0, dt_ = 0.0;
  double global_volP_ = -1.0;      // < 0: not set (single domain: the local sum is the global one)
  double rmax_ = 0.0;

  // styles
  GranParams gran_{};
  CoheParams cohe_{};
  LubParams lub_{};
  int nwalls_ = 0;
  WallParams walls_[kMaxWalls];
  WallMotion wall_motion_[kMaxWalls];
  long long wall_time_origin_ = 0;   // sub-step count at setup (FixWallGranFix::init, :181)
  long long run_base_step_ = 0;      // sub-steps done when the current run / batch sequence started
  bool have_gravity_ = false, have_fdrag_ = false, have_nve_ = false;
  std::map<std::string, int> groups_{{"all", 1}};
  bool use_groups_ = false;
  int nve_bit_ = 1, grav_bit_ = 1, fdrag_bit_ = 1, cohe_bit_ = 1, freeze_bit_ = 0;
  int post_freeze_ = 0;   // StepParams::post_freeze
  int new_group_bit(const std::string& name);
  void mark_frozen();
  double gacc_[3] = {0, 0, 0};
  double carrier_rho_ = 0.0;

  // device arrays (rows x cap)
  DevArray xr_[2], vm_[2], om_[2], force_, torque_;
  DevArray tag_, type_, mask_, foamCpuId_;
  DevArray fdrag_, DuDt_, vOld_, xhold_;
  DevArray extra_;                 // [kMaxExtra][cap] client rows (register_extra)
  int nextra_ = 0;                 // 1 + highest row in use
  unsigned extra_used_ = 0;        // bit r: row r belongs to a client
  double extra_init_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  DevArray wshear_, wtouch_;
  DevArray gsrc_, gshift_;
  DevArray neigh_, numneigh_, shear_[2];
  DevArray neigh_old_, numneigh_old_, ptag_;   // B-side buffers swapped in by permute/build
  int flags_seq_ = 0;                // sequence number of the last flag publication (flags_copy_begin / _wait)
  bool build_flags_clean_ = false;   // the list build's counters were zeroed by k_pbc_keys of this rebuild
  bool trigger_rearmed_ = false;     // F_TRIGGER was set back to INT_MAX by k_back_slots of this rebuild
  int park_rows_ = 0;                // LDS parking rows of the next list build (0: as many as list slots)
  bool hist_in_place_ = false;   // this rebuild's list build reads the old list in place (compute_partner_tags)
  bool build_parks_in_lds() const;
  // During a rebuild shear_[hist_buf_] holds the OLD list's history expanded to a copy per side (what partner tags
  // address: migration, re-injection); the new list's history is built into the other buffer, which then becomes
  // shear_[cur_]
  int hist_buf_ = 0;
  // one history copy per contact (true) or one per side (false): chosen per list build from the measured coalescing
  // of the previous list (SF_HIST_COPIES=1 / 2 pins it)
  bool hist_single_ = true;
  int hist_mode_env_ = 0;
  // v / omega of a neighbour prefetched by touch bit (true) or always (false): template parameter TP of k_substep,
  // from the measured fraction of listed neighbours that touch (SF_TOUCH_PREFETCH=0 / 1 pins it)
  bool touch_prefetch_ = true;
  int touch_prefetch_env_ = -1;
  // touching neighbours in the first slots of a row (k_build_neigh): loose beds only
  bool touch_first_ = false;
  int touch_first_env_ = -1;
  int opt_lpa_ = 0;                          // SF_LPA: lanes per atom pinned (1, 2 or 4; 0 = by size)
  // Single domain, plain keys, (root, image code) words: the periodic images of a rebuild are not materialised as ghost
  // atoms -- the list build wraps its cell stencil around the box (BinGrid::wrap) and tests atom j + box length where
  // LAMMPS tests the ghost copy (the same sum, the same bits).  Twelve launches less per rebuild (ghost selection and
  // creation per dimension, the counting sort of the ghosts).  nghost() -- a diagnostic -- counts the images LAMMPS would
  // have made when somebody asks.  SF_GHOST_FREE=0: the ghost path.
  int opt_ghost_free_ = -1;
  bool ghost_free_ = false;
  int nimages_ = -1;                         // ghost_free_: images counted for nghost() (-1: not counted since the last rebuild)
  int opt_persist_ = -1;                     // SF_PERSIST: persistent tiles (k_substep_persist) off (0), on wherever the kernel
                                             // exists (1), default (-1): where it measured faster (launch_substep)
  int opt_persist_waves_ = 0;                // SF_PERSIST_WAVES: waves per XCD of the persistent launch (0: the resident ones)
  int* d_pq_head_ = nullptr;                 // [2][8][32] head words of the persistent launch (DemPtrs::pq_head)
  int pq_par_ = 0;
  bool in_run_ = false;                      // rebuild() called from the stepping loop of run()
  RebuildPredictor predict_;                 // single-domain run(): how far to queue (SF_QUEUE_PREDICT=0: everything)
  int nt_policy_ = 2, nt_policy_env_ = -1;   // non-temporal policy of the row streams (sf_dem_kernels.h, NTP)
  void measure_list();     // queue k_partner_coalescing on the current list (results with the next flag read)
  bool list_stats_near_a_threshold() const;
  long long list_sampled_ = 0;   // atoms the last measure_list looked at
  void choose_kernel();    // pick touch_prefetch_ from the last measurement
  DevArray nloc_;                      // [M][cap] uint16 (see DemPtrs::nloc)
  int* tile_tab_ = nullptr;            // [2][ntiles] tile_first / tile_last, then [ntiles+1] counts, starts
  size_t tile_alloc_ = 0;
  int* stage_idx_ = nullptr;
  size_t stage_alloc_ = 0;
  int* eoff_ = nullptr;                // [ntiles][(T+2)^3] offset of an extended bin in the tile's staged copy
  size_t eoff_alloc_ = 0;
  int ntiles_ = 0, stage_cap_ = 0;
  bool lds_active_ = false;
  void build_stage_tables();
  DevArray tmp4_, tmpd_, tmpi_;        // gather scratch
  // (one scratch array per permuted array: permute_locals gathers everything in one launch and swaps allocations)
  DevArray tmp4b_, tmp4c_, tmpi_b_, tmpi_c_, tmpi_d_, fdrag_alt_, DuDt_alt_, vOld_alt_, extra_alt_, wtouch_alt_;
  DevArray keys_, keys_alt_, perm_, perm_alt_, keys64_, keys64_alt_;
  int* cell_start_ = nullptr;          // [nbins][4]: owned start/end, ghost start/end of every cell (hipMalloc: 16-byte aligned)
  size_t cell_alloc_ = 0;
  DevArray hist_perm_;        // see permute_locals(rows = false)
  bool hist_indirect_ = false;
  bool row_tables_ = false;   // cell_start_ holds the reversed lower-bound tables (plain keys) instead of cell ranges
  int* tagmap_ = nullptr;
  size_t tagmap_alloc_ = 0;
  long long order_version_ = 0, tagmap_builds_ = -1;   // the tag -> index table is rebuilt when the atom order changed
  struct IoBuf {
    void* p = nullptr;
    size_t n = 0;
  };
  IoBuf io_d_[3], io_i_[2];            // staging of the lammps_put/get_local_info boundary (persistent, grown)
  double* io_doubles(int which, size_t n);
  int* io_ints(int which, size_t n);
  void* sort_tmp_ = nullptr;
  size_t sort_tmp_bytes_ = 0;
  int* d_flags_ = nullptr;
  int* own_flags_ = nullptr;
  unsigned long long* count64_ = nullptr;   // scratch counter of npairs_full()
  int* h_flags_ = nullptr;             // pinned
  std::vector<DevArray*> per_atom_;    // registry for capacity growth
  BinGrid grid_{};
  // halo bookkeeping: send lists for forward comm [side] (device index arrays) and ghost slot ranges
  DevArray sendlist_[2];
  DevArray sendslot_;                  // [2][cap] inverse of sendlist_ (-1: not sent), for the fused forward pack
  double* tx_ptr_[2] = {nullptr, nullptr};
  double tx_shift_[2] = {0.0, 0.0};
  double* tx_sendbuf_ = nullptr;
  const int* tx_hdr_off_ = nullptr;
  int tx_nhdr_ = 0, tx_n_[2] = {0, 0};
  bool tx_ready_ = false, tx_written_ = false;
  bool tx_direct_ = false;             // records go straight into the neighbours' receive areas
  bool gs_ready_ = false;              // ghost slots: the tables below are valid
  GsSync* d_gs_sync_ = nullptr;
  GsSync h_gs_sync_{};
  int* d_gs_count_ = nullptr;          // [8][32] completion counters, one 128-byte line per XCD (zero between launches)
  long long gs_seq_ = 1;
  int gs_map_base_ = 0;                // buffer that launches of EVEN number read, as the neighbours were told
  int tx_par_ = 0;
  double** d_blkptr_ = nullptr;        // [2][kMaxDirs] device table of block starts (both rows equal unless tx_direct_),
                                       // then [kMaxDirs] records per block (size_t)
  double** d_gsblk_ = nullptr;         // ghost slots: [2][3][kMaxDirs] block starts in the neighbours' x | v | omega arrays,
                                       // then [kMaxDirs][3] shifts
  DevArray isb_;                       // (check only, SF_CHECK_BOUNDARY=1) list-derived boundary flags
  // brick decomposition: directions, face masks of the owned atoms, concatenated send lists
  int bndir_ = 0;
  int bdir_[kMaxDirs][3];
  BrickBlocks bsend_blocks_{};
  DevArray bmask_;
  int* bsend_list_ = nullptr;
  DevArray bslot_;                     // [kBrickSlots + 1][cap]: record positions, then a per-atom cursor
  size_t bsend_alloc_ = 0;
  int* d_bcount_ = nullptr;            // [2 * kMaxDirs] device counters / cursors
  int* h_bcount_ = nullptr;            // pinned twin
  int lanes_per_atom(int nwork) const;
  int nb_ = 0;                         // boundary atoms = [0, n_lo_) and [n_hi_, nlocal_) of the x-slowest order
  int n_lo_ = 0, n_hi_ = 0;
  bool overlap_ = false;
  bool roots_ = true;                  // neighbour words are (root, image code); false with the LDS-staged kernel
  hipStream_t comm_stream_ = nullptr;
  hipStream_t masked_main_ = nullptr, masked_comm_ = nullptr;   // CU-partitioned pair (make_partitioned_streams)
  void mark_boundary();
  double lskin() const { return overlap_ ? 1.1 * skin_ : skin_; }   // list skin: +10 % margin in overlap mode
  long long nsend_[2] = {0, 0};
  int recv_first_[2] = {0, 0}, recv_count_[2] = {0, 0};
  hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
  bool hist_clean_ = false;   // the cell histograms of cell_start_ are all zero (their users count them back down)
  long long prof_rebuilds_ = 0;
  double prof_rebuild_ms_ = 0.0;
  hipEvent_t ev_flags_ = nullptr;   // "the flag words have reached the host" (bin_and_build)
  bool profiling_ = false;
  std::vector<hipEvent_t> prof_ev_;   // pairs (start, stop) of launches not yet harvested
  size_t prof_used_ = 0;
  std::vector<int> prof_step_;         // sub-step number of each pair
  bool prof_open_ = false;             // a boundary part recorded its start event, the interior part closes it
  long long prof_launches_ = 0;
  double prof_ms_ = 0.0;
  void harvest_profile(int last_step);
};

// sf_sort.hip
void sort_pairs_u32(void*& tmp, size_t& tmp_bytes, unsigned* keys_in, unsigned* keys_out, int* vals_in,
                    int* vals_out, int n, int end_bit, hipStream_t s);
void exclusive_scan_i32(void*& tmp, size_t& tmp_bytes, const int* in, int* out, int n, hipStream_t s);
void select_zero_keys(void*& tmp, size_t& tmp_bytes, const unsigned* keys, int* out, int* d_count, int n,
                      hipStream_t s);
void sort_pairs_u64(void*& tmp, size_t& tmp_bytes, unsigned long long* keys_in,
                    unsigned long long* keys_out, int* vals_in, int* vals_out, int n, int end_bit,
                    hipStream_t s);

}  // namespace sf
