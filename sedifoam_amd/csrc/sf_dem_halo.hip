// sf_dem_halo.hip -- ghost-particle halo of the 1-D slab decomposition (one GPU per slab along x) and
// particle injection/removal.
//
// This is the device half of what LAMMPS' Comm class does for the reference ([3P] comm.cpp, configured by
// `communicate single vel yes`, `newton off` in every in.lammps): exchange() = atoms that left the slab
// migrate with all their per-atom data (incl. fix fdrag's arrays, fix_fluid_drag.cpp:211-243, the wall shear
// history, fix_wall_granFix.cpp:726-744, and the pair shear history by partner tag, FixShearHistory);
// borders() = atoms within the ghost cutoff of a slab face become ghost atoms of the neighbour slab;
// forward_comm() = ghost x, v, omega refreshed every sub-step (72 B per ghost).  With `newton off` there is
// no reverse (force) communication: both owners evaluate a cross-boundary contact.
// Packing/unpacking run as HIP kernels on device buffers; the transport between GPUs (RCCL send/recv over
// xGMI) is done by the host driver, sedifoam_amd/halo.py.
#include <algorithm>
#include <climits>
#include <vector>

#include "sf_dem.h"
#include "sf_dem_gs.h"

namespace sf {

constexpr int kBorderDoubles = 14;   // x r | v m | omega | tag type mask
constexpr int kMigrateFixed = 26;  // + 3*nwalls + 4*mrec + nextra

// key 0 = selected, 1 = not (a stable compaction then lists the selected atoms, ascending)
// mode 0: x < bound ; mode 1: x >= bound
__global__ __launch_bounds__(256) void k_select_keys(const double4* xr, int n, int mode, double bound, unsigned* keys,
                                                     int dim = 0)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double4 p = xr[i];
  const double x = dim == 0 ? p.x : (dim == 1 ? p.y : p.z);
  keys[i] = (mode == 0 ? (x < bound) : (x >= bound)) ? 0u : 1u;
}

// owned atoms that left the slab [lo, hi) through either face
__global__ __launch_bounds__(1024) void k_count_outside(const double4* xr, int n, double lo, double hi, int* counter)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool out = false;
  if (i < n) {
    const double x = xr[i].x;
    out = x < lo || x >= hi;
  }
  const int t = block_sum_int_1024(out ? 1 : 0);   // one global atomic per block (sf_dem.h)
  if (threadIdx.x == 0 && t) atomicAdd(counter, t);
}

__global__ __launch_bounds__(256) void k_border_pack(const int* list, int n, double xshift, const double4* xr,
                                                     const double4* vm, const double4* om, const int* tag,
                                                     const int* type, const int* mask, double* buf)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int i = list[k];
  const double4 x = xr[i], v = vm[i], w = om[i];
  double* b = buf + (size_t)k * kBorderDoubles;
  b[13] = (double)mask[i];
  b[0] = x.x + xshift; b[1] = x.y; b[2] = x.z; b[3] = x.w;
  b[4] = v.x; b[5] = v.y; b[6] = v.z; b[7] = v.w;
  b[8] = w.x; b[9] = w.y; b[10] = w.z;
  b[11] = (double)tag[i];
  b[12] = (double)type[i];
}

__global__ __launch_bounds__(256) void k_border_unpack(const double* buf, int n, int first, double4* xr, double4* vm,
                                                       double4* om, double4* xr_b, double4* vm_b, double4* om_b,
                                                       int* tag, int* type, int* mask, int* gsrc, int freeze_bit)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const double* b = buf + (size_t)k * kBorderDoubles;
  const int g = first + k;
  xr[g] = {b[0], b[1], b[2], b[3]};
  vm[g] = {b[4], b[5], b[6], b[7]};
  // omega.w = 1 marks an atom of the fix-freeze group (the pair kernel gathers it with omega)
  const int m = (int)b[13];
  const double frozen = (m & freeze_bit) ? 1.0 : 0.0;
  om[g] = {b[8], b[9], b[10], frozen};
  // radius, mass and the frozen mark (.w) must also be valid in the other ping-pong buffer: the forward halo only
  // carries x, v, omega
  xr_b[g] = {b[0], b[1], b[2], b[3]};
  vm_b[g] = {b[4], b[5], b[6], b[7]};
  om_b[g] = {b[8], b[9], b[10], frozen};
  tag[g] = (int)b[11];
  type[g] = (int)b[12];
  mask[g] = m;
  gsrc[g] = -1;  // owned by another GPU
}

__global__ __launch_bounds__(256) void k_forward_pack(const int* list, int n, double xshift, const double4* xr,
                                                      const double4* vm, const double4* om, double* buf)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int i = list[k];
  const double4 x = xr[i], v = vm[i], w = om[i];
  double* b = buf + (size_t)k * kForwardDoubles;
  b[0] = x.x + xshift; b[1] = x.y; b[2] = x.z;
  b[3] = v.x; b[4] = v.y; b[5] = v.z;
  b[6] = w.x; b[7] = w.y; b[8] = w.z;
}

__global__ __launch_bounds__(256) void k_forward_unpack(const double* buf, int n, int first, double4* xr, double4* vm,
                                                        double4* om)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const double* b = buf + (size_t)k * kForwardDoubles;
  const int g = first + k;
  double4 x = xr[g], v = vm[g];
  x.x = b[0]; x.y = b[1]; x.z = b[2];
  v.x = b[3]; v.y = b[4]; v.z = b[5];
  xr[g] = x;   // radius / mass (.w) were set by the border exchange
  vm[g] = v;
  om[g] = {b[6], b[7], b[8], om[g].w};   // .w: frozen mark, set by the border exchange
}

struct MigratePtrs {
  double4 *xr, *vm, *om;
  int *tag, *type, *mask, *foamCpuId, *numneigh, *ptag;
  double *fdrag, *DuDt, *vOld, *wshear, *shear;
  unsigned char* wtouch;
  double* extra;   // client rows (DemEngine::register_extra), nextra of them
  int nextra;
};

__global__ __launch_bounds__(128) void k_migrate_pack(const int* list, int n, double xshift, MigratePtrs P, size_t cap,
                                                      int nwalls, int mrec, int have_list, int rec, double* buf,
                                                      int* leave, int code, int* flags, int dim = 0)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int i = list[k];
  leave[i] = code;
  double* b = buf + (size_t)k * rec;
  const double4 x = P.xr[i], v = P.vm[i], w = P.om[i];
  b[0] = x.x; b[1] = x.y; b[2] = x.z; b[3] = x.w;
  b[dim] += xshift;   // (the migration shift of the face's dimension)
  b[4] = v.x; b[5] = v.y; b[6] = v.z; b[7] = v.w;
  b[8] = w.x; b[9] = w.y; b[10] = w.z;
  b[11] = P.tag[i]; b[12] = P.type[i]; b[13] = P.mask[i]; b[14] = P.foamCpuId[i];
  for (int c = 0; c < 3; c++) {
    b[15 + c] = P.fdrag[(size_t)c * cap + i];
    b[18 + c] = P.DuDt[(size_t)c * cap + i];
    b[21 + c] = P.vOld[(size_t)c * cap + i];
  }
  b[24] = P.wtouch[i];
  for (int c = 0; c < 3 * nwalls; c++) b[25 + c] = P.wshear[(size_t)c * cap + i];
  double* h = b + 25 + 3 * nwalls;
  int nn = have_list ? P.numneigh[i] : 0;
  if (nn > mrec) {   // the record has mrec slots (the global max of max_neigh_used): more would lose history -- an error
    flags[F_MIG_TRUNC] = nn;
    nn = mrec;
  }
  h[0] = nn;
  for (int s = 0; s < mrec; s++) {
    const bool ok = s < nn;
    h[1 + 4 * s] = ok ? (double)P.ptag[(size_t)s * cap + i] : -1.0;
    for (int c = 0; c < 3; c++) h[2 + 4 * s + c] = ok ? P.shear[(size_t)(3 * s + c) * cap + i] : 0.0;
  }
  double* ex = h + 1 + 4 * mrec;
  for (int r = 0; r < P.nextra; r++) ex[r] = P.extra[(size_t)r * cap + i];
}

__global__ __launch_bounds__(128) void k_migrate_unpack(const double* buf, int n, int first, MigratePtrs P, size_t cap,
                                                        int nwalls, int mrec, int rec)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const double* b = buf + (size_t)k * rec;
  const int i = first + k;
  P.xr[i] = {b[0], b[1], b[2], b[3]};
  P.vm[i] = {b[4], b[5], b[6], b[7]};
  P.om[i] = {b[8], b[9], b[10], 0.0};
  P.tag[i] = (int)b[11]; P.type[i] = (int)b[12]; P.mask[i] = (int)b[13]; P.foamCpuId[i] = (int)b[14];
  for (int c = 0; c < 3; c++) {
    P.fdrag[(size_t)c * cap + i] = b[15 + c];
    P.DuDt[(size_t)c * cap + i] = b[18 + c];
    P.vOld[(size_t)c * cap + i] = b[21 + c];
  }
  P.wtouch[i] = (unsigned char)b[24];
  for (int c = 0; c < 3 * nwalls; c++) P.wshear[(size_t)c * cap + i] = b[25 + c];
  const double* h = b + 25 + 3 * nwalls;
  P.numneigh[i] = (int)h[0];
  for (int s = 0; s < mrec; s++) {
    P.ptag[(size_t)s * cap + i] = (int)h[1 + 4 * s];
    for (int c = 0; c < 3; c++) P.shear[(size_t)(3 * s + c) * cap + i] = h[2 + 4 * s + c];
  }
  const double* ex = h + 1 + 4 * mrec;
  for (int r = 0; r < P.nextra; r++) P.extra[(size_t)r * cap + i] = ex[r];
}

__global__ __launch_bounds__(256) void k_stay_keys(const int* leave, int n, unsigned* keys, int* idx)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys[i] = leave[i] ? 1u : 0u;
  idx[i] = i;
}

// ------------------------------------------------------------------------------------------------
int DemEngine::select_locals(int mode, double bound, DevArray& list, int dim)
{
  if (!nlocal_) return 0;
  k_select_keys<<<div_up(nlocal_, 256), 256, 0, stream_>>>(xr_[cur_].as<double4>(), nlocal_, mode, bound,
                                                         keys_.as<unsigned>(), dim);
  select_zero_keys(sort_tmp_, sort_tmp_bytes_, keys_.as<unsigned>(), list.as<int>(), d_flags_ + F_SEND_COUNT,
                   nlocal_, stream_);
  read_flags();
  return h_flags_[F_SEND_COUNT];
}

long long DemEngine::border_pack(int side, double xshift, double* buf, long long max_atoms)
{
  if (side < 0 || side > 1) fail("border_pack: side must be 0 or 1");
  const double cut = cutneighmax();
  const int n = select_locals(side == 0 ? 0 : 1, side == 0 ? sublo_[0] + cut : subhi_[0] - cut, sendlist_[side]);
  if (n > max_atoms) fail("border_pack: %d atoms do not fit the %lld-atom buffer", n, max_atoms);
  nsend_[side] = n;
  if (n)
    k_border_pack<<<div_up(n, 256), 256, 0, stream_>>>(sendlist_[side].as<int>(), n, xshift, xr_[cur_].as<double4>(),
                                                       vm_[cur_].as<double4>(), om_[cur_].as<double4>(),
                                                       tag_.as<int>(), type_.as<int>(), mask_.as<int>(), buf);
  if (!external_stream_) sync();
  return n;
}

// both faces with ONE host read of the two counts (the per-side call above synchronises once per side)
void DemEngine::border_pack_both(double xshift0, double* buf0, double xshift1, double* buf1, long long max_atoms,
                                 long long* n0, long long* n1)
{
  *n0 = *n1 = 0;
  nsend_[0] = nsend_[1] = 0;
  if (!nlocal_) return;
  const double cut = cutneighmax();
  const double bound[2] = {sublo_[0] + cut, subhi_[0] - cut};
  for (int side = 0; side < 2; side++) {
    k_select_keys<<<div_up(nlocal_, 256), 256, 0, stream_>>>(xr_[cur_].as<double4>(), nlocal_, side, bound[side],
                                                           keys_.as<unsigned>());
    select_zero_keys(sort_tmp_, sort_tmp_bytes_, keys_.as<unsigned>(), sendlist_[side].as<int>(),
                     d_flags_ + (side ? F_SEND_COUNT2 : F_SEND_COUNT), nlocal_, stream_);
  }
  read_flags();
  const int n[2] = {h_flags_[F_SEND_COUNT], h_flags_[F_SEND_COUNT2]};
  double* const buf[2] = {buf0, buf1};
  const double shift[2] = {xshift0, xshift1};
  for (int side = 0; side < 2; side++) {
    if (n[side] > max_atoms) fail("border_pack: %d atoms do not fit the %lld-atom buffer", n[side], max_atoms);
    nsend_[side] = n[side];
    if (n[side])
      k_border_pack<<<div_up(n[side], 256), 256, 0, stream_>>>(sendlist_[side].as<int>(), n[side], shift[side],
                                                               xr_[cur_].as<double4>(), vm_[cur_].as<double4>(),
                                                               om_[cur_].as<double4>(), tag_.as<int>(), type_.as<int>(),
                                                               mask_.as<int>(), buf[side]);
  }
  *n0 = n[0];
  *n1 = n[1];
}

void DemEngine::border_unpack(int side, const double* buf, long long natoms)
{
  if (side < 0 || side > 1) fail("border_unpack: side must be 0 or 1");
  const int n = (int)natoms;
  ensure_capacity((size_t)nlocal_ + next_ghost_ + n + 1024);
  const int first = nlocal_ + next_ghost_;
  recv_first_[side] = first;
  recv_count_[side] = n;
  if (n)
    k_border_unpack<<<div_up(n, 256), 256, 0, stream_>>>(buf, n, first, xr_[cur_].as<double4>(),
                                                         vm_[cur_].as<double4>(), om_[cur_].as<double4>(),
                                                         xr_[cur_ ^ 1].as<double4>(), vm_[cur_ ^ 1].as<double4>(),
                                                         om_[cur_ ^ 1].as<double4>(), tag_.as<int>(), type_.as<int>(),
                                                         mask_.as<int>(), gsrc_.as<int>(), freeze_bit_);
  next_ghost_ += n;
  if (!external_stream_) sync();
}

long long DemEngine::forward_pack(int side, double xshift, double* buf)
{
  const int n = (int)nsend_[side];
  if (n)
    k_forward_pack<<<div_up(n, 256), 256, 0, stream_>>>(sendlist_[side].as<int>(), n, xshift,
                                                        xr_[cur_].as<double4>(), vm_[cur_].as<double4>(),
                                                        om_[cur_].as<double4>(), buf);
  if (!external_stream_) sync();
  return n;
}

void DemEngine::forward_unpack(int side, const double* buf, long long natoms)
{
  if (natoms != recv_count_[side])
    fail("forward_unpack: got %lld ghosts from side %d, the border exchange set up %d", natoms, side,
         recv_count_[side]);
  const int n = (int)natoms;
  if (n)
    k_forward_unpack<<<div_up(n, 256), 256, 0, stream_>>>(buf, n, recv_first_[side], xr_[cur_].as<double4>(),
                                                          vm_[cur_].as<double4>(), om_[cur_].as<double4>());
}

void DemEngine::ghost_forward_local() { launch_ghost_forward(cur_, INT_MIN); }

// both faces in one launch each (the per-sub-step path of the multi-rank driver)
__global__ __launch_bounds__(256) void k_forward_pack2(const int* list0, int n0, double shift0, double* buf0,
                                                       const int* list1, int n1, double shift1, double* buf1,
                                                       const double4* xr, const double4* vm, const double4* om)
{
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int* list = list0;
  double shift = shift0;
  double* buf = buf0;
  if (k >= n0) {
    k -= n0;
    if (k >= n1) return;
    list = list1;
    shift = shift1;
    buf = buf1;
  }
  const int i = list[k];
  const double4 x = xr[i], v = vm[i], w = om[i];
  double* b = buf + (size_t)k * kForwardDoubles;
  b[0] = x.x + shift; b[1] = x.y; b[2] = x.z;
  b[3] = v.x; b[4] = v.y; b[5] = v.z;
  b[6] = w.x; b[7] = w.y; b[8] = w.z;
}

__global__ __launch_bounds__(256) void k_forward_unpack2(const double* buf0, int n0, int first0, const double* buf1,
                                                         int n1, int first1, double4* xr, double4* vm, double4* om)
{
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  const double* buf = buf0;
  int first = first0;
  if (k >= n0) {
    k -= n0;
    if (k >= n1) return;
    buf = buf1;
    first = first1;
  }
  const double* b = buf + (size_t)k * kForwardDoubles;
  const int g = first + k;
  double4 x = xr[g], v = vm[g];
  x.x = b[0]; x.y = b[1]; x.z = b[2];
  v.x = b[3]; v.y = b[4]; v.z = b[5];
  xr[g] = x;
  vm[g] = v;
  om[g] = {b[6], b[7], b[8], om[g].w};   // .w: frozen mark, set by the border exchange
}

// Fused forward halo for ONE all-to-all per sub-step: the send buffer holds, for every peer rank, a header word
// (this rank's rebuild trigger, as a double) followed by the halo records meant for that peer, so the same
// exchange carries the ghosts and the global rebuild vote (MIN over the headers on the receiving side).
__global__ __launch_bounds__(256) void k_forward_pack_fused(const int* list0, int n0, double shift0, double* buf0,
                                                            const int* list1, int n1, double shift1, double* buf1,
                                                            const int* hdr_off, int nhdr, double* sendbuf,
                                                            const int* trig_word, const double4* xr,
                                                            const double4* vm, const double4* om)
{
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int* list = list0;
  double shift = shift0;
  double* buf = buf0;
  size_t n = (size_t)n0;
  if (k >= n0) {
    k -= n0;
    if (k >= n1) {
      k -= n1;
      if (k < nhdr) {
        sendbuf[hdr_off[k]] = 0.0;
        *header_vote_ptr(sendbuf + hdr_off[k]) = __atomic_load_n(trig_word, __ATOMIC_RELAXED);
      }
      return;
    }
    list = list1;
    shift = shift1;
    buf = buf1;
    n = (size_t)n1;
  }
  // a face's block is component-major, [kForwardDoubles][n]: consecutive border atoms write (and the receiver reads)
  // consecutive doubles -- also what lets the sub-step kernel write these records itself (StepParams::tx_fused)
  const int i = list[k];
  const double4 x = xr[i], v = vm[i], w = om[i];
  double* b = buf + k;
  b[0] = x.x + shift; b[n] = x.y; b[2 * n] = x.z;
  b[3 * n] = v.x; b[4 * n] = v.y; b[5 * n] = v.z;
  b[6 * n] = w.x; b[7 * n] = w.y; b[8 * n] = w.z;
}

__global__ __launch_bounds__(256) void k_forward_unpack_fused(const double* buf0, int n0, int first0,
                                                              const double* buf1, int n1, int first1,
                                                              const int* hdr_off, int nhdr, const double* recvbuf,
                                                              int* vote_word, double4* xr, double4* vm, double4* om)
{
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  const double* buf = buf0;
  int first = first0;
  size_t n = (size_t)n0;
  if (k >= n0) {
    k -= n0;
    if (k >= n1) {
      k -= n1;
      if (k < nhdr) atomicMin(vote_word, header_vote(recvbuf + hdr_off[k]));
      return;
    }
    buf = buf1;
    first = first1;
    n = (size_t)n1;
  }
  const double* b = buf + k;
  const int g = first + k;
  double4 x = xr[g], v = vm[g];
  x.x = b[0]; x.y = b[n]; x.z = b[2 * n];
  v.x = b[3 * n]; v.y = b[4 * n]; v.z = b[5 * n];
  xr[g] = x;
  vm[g] = v;
  om[g] = {b[6 * n], b[7 * n], b[8 * n], om[g].w};   // .w: frozen mark, set by the border exchange
}

void DemEngine::forward_pack_fused(double shift0, long long off0, double shift1, long long off1, const int* hdr_off,
                                   int nhdr, double* sendbuf)
{
  // overlap mode: on the communication stream, header = this rank's accumulated local trigger
  hipStream_t st = overlap_ ? comm_stream_ : stream_;
  const int tot = (int)(nsend_[0] + nsend_[1]) + nhdr;
  if (tot)
    k_forward_pack_fused<<<div_up(tot, 256), 256, 0, st>>>(
        sendlist_[0].as<int>(), (int)nsend_[0], shift0, sendbuf + off0, sendlist_[1].as<int>(), (int)nsend_[1],
        shift1, sendbuf + off1, hdr_off, nhdr, sendbuf, d_flags_ + (overlap_ ? F_TRIG_LOCAL : F_TRIGGER),
        xr_[cur_].as<double4>(), vm_[cur_].as<double4>(), om_[cur_].as<double4>());
  if (!external_stream_) sync();
}

void DemEngine::forward_unpack_fused(const double* recvbuf, long long off0, long long n0, long long off1,
                                     long long n1, const int* hdr_off, int nhdr, int kstep)
{
  if (n0 != recv_count_[0] || n1 != recv_count_[1])
    fail("forward_unpack: got %lld/%lld ghosts, the border exchange set up %d/%d", n0, n1, recv_count_[0],
         recv_count_[1]);
  hipStream_t st = overlap_ ? comm_stream_ : stream_;
  // overlap mode: the vote of the exchange that follows sub-step kstep goes to F_VOTE0 + (kstep & 1)
  int* vote = d_flags_ + (overlap_ ? F_VOTE0 + (kstep & 1) : F_TRIGGER);
  const int tot = (int)(n0 + n1) + nhdr;
  if (tot)
    k_forward_unpack_fused<<<div_up(tot, 256), 256, 0, st>>>(
        recvbuf + off0, (int)n0, recv_first_[0], recvbuf + off1, (int)n1, recv_first_[1], hdr_off, nhdr, recvbuf,
        vote, xr_[cur_].as<double4>(), vm_[cur_].as<double4>(), om_[cur_].as<double4>());
  if (overlap_)   // images of the received ghosts; images of owned atoms were refreshed by substep_flip
    launch_ghost_forward(cur_, kstep, 2, F_VOTE0 + ((kstep + 1) & 1), comm_stream_);
  else
    launch_ghost_forward(cur_, INT_MIN);   // local y/z images of everything, received ghosts included
}

__global__ __launch_bounds__(256) void k_fill_sendslot(const int* list0, int n0, int* slot0, const int* list1, int n1,
                                                       int* slot1)
{
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n0) slot0[list0[k]] = k;
  else if (k - n0 < n1) slot1[list1[k - n0]] = k - n0;
}

void DemEngine::set_forward_tx(double* tx0, double shift0, double* tx1, double shift1, double* sendbuf,
                               const int* hdr_off, int nhdr)
{
  static const bool off = getenv("SF_HALO_FUSED_PACK") && !atoi(getenv("SF_HALO_FUSED_PACK"));
  tx_ready_ = tx_written_ = false;
  if (off || !nlocal_) return;
  if (sendslot_.cap < cap_) sendslot_.alloc(sizeof(int), 2, cap_, stream_);
  SF_HIP(hipMemsetAsync(sendslot_.ptr, 0xFF, sizeof(int) * 2 * sendslot_.cap, stream_));
  const int tot = (int)(nsend_[0] + nsend_[1]);
  if (tot)
    k_fill_sendslot<<<div_up(tot, 256), 256, 0, stream_>>>(sendlist_[0].as<int>(), (int)nsend_[0], sendslot_.as<int>(),
                                                           sendlist_[1].as<int>(), (int)nsend_[1],
                                                           sendslot_.as<int>() + sendslot_.cap);
  tx_ptr_[0] = tx0;
  tx_ptr_[1] = tx1;
  tx_shift_[0] = shift0;
  tx_shift_[1] = shift1;
  tx_sendbuf_ = sendbuf;
  tx_hdr_off_ = hdr_off;
  tx_nhdr_ = nhdr;
  tx_n_[0] = (int)nsend_[0];
  tx_n_[1] = (int)nsend_[1];
  tx_ready_ = true;
}

void DemEngine::forward_pack2(double shift0, double* buf0, double shift1, double* buf1, long long* n0, long long* n1)
{
  *n0 = nsend_[0];
  *n1 = nsend_[1];
  const int tot = (int)(nsend_[0] + nsend_[1]);
  if (tot)
    k_forward_pack2<<<div_up(tot, 256), 256, 0, stream_>>>(sendlist_[0].as<int>(), (int)nsend_[0], shift0, buf0,
                                                           sendlist_[1].as<int>(), (int)nsend_[1], shift1, buf1,
                                                           xr_[cur_].as<double4>(), vm_[cur_].as<double4>(),
                                                           om_[cur_].as<double4>());
  if (!external_stream_) sync();
}

void DemEngine::forward_unpack2(const double* buf0, long long n0, const double* buf1, long long n1)
{
  if (n0 != recv_count_[0] || n1 != recv_count_[1])
    fail("forward_unpack: got %lld/%lld ghosts, the border exchange set up %d/%d", n0, n1, recv_count_[0],
         recv_count_[1]);
  const int tot = (int)(n0 + n1);
  if (tot)
    k_forward_unpack2<<<div_up(tot, 256), 256, 0, stream_>>>(buf0, (int)n0, recv_first_[0], buf1, (int)n1,
                                                             recv_first_[1], xr_[cur_].as<double4>(),
                                                             vm_[cur_].as<double4>(), om_[cur_].as<double4>());
  launch_ghost_forward(cur_, INT_MIN);   // local y/z images of everything, received ghosts included
}

void DemEngine::migrate_set_slots(int mrec)
{
  mrec_ = std::max(mrec, 0);
  if (mrec_ > M_) grow_neigh(mrec_ + 4);
  max_neigh_used_ = std::max(max_neigh_used_, mrec_);
}

int DemEngine::migrate_record_doubles() const { return kMigrateFixed + 3 * nwalls_ + 4 * mrec_ + nextra_; }

static MigratePtrs mig_ptrs(DevArray& xr, DevArray& vm, DevArray& om, DevArray& tag, DevArray& type, DevArray& mask,
                            DevArray& foam, DevArray& numneigh, DevArray& ptag, DevArray& fdrag, DevArray& DuDt,
                            DevArray& vOld, DevArray& wshear, DevArray& shear, DevArray& wtouch, DevArray& extra,
                            int nextra)
{
  MigratePtrs P;
  P.xr = xr.as<double4>(); P.vm = vm.as<double4>(); P.om = om.as<double4>();
  P.tag = tag.as<int>(); P.type = type.as<int>(); P.mask = mask.as<int>(); P.foamCpuId = foam.as<int>();
  P.numneigh = numneigh.as<int>(); P.ptag = ptag.as<int>();
  P.fdrag = fdrag.as<double>(); P.DuDt = DuDt.as<double>(); P.vOld = vOld.as<double>();
  P.wshear = wshear.as<double>(); P.shear = shear.as<double>();
  P.wtouch = wtouch.as<unsigned char>();
  P.extra = extra.as<double>();
  P.nextra = nextra;
  return P;
}

long long DemEngine::migrate_count()
{
  if (!nlocal_) return 0;
  reset_flag(F_SEND_COUNT, 0);
  k_count_outside<<<div_up(nlocal_, 1024), 1024, 0, stream_>>>(xr_[cur_].as<double4>(), nlocal_, sublo_[0], subhi_[0],
                                                             d_flags_ + F_SEND_COUNT);
  read_flags();
  return h_flags_[F_SEND_COUNT];
}

long long DemEngine::migrate_pack(int side, double xshift, double* buf, long long max_doubles)
{
  return migrate_pack_dim(0, side, xshift, buf, max_doubles);
}

long long DemEngine::migrate_pack_dim(int dim, int side, double xshift, double* buf, long long max_doubles)
{
  if (side < 0 || side > 1 || dim < 0 || dim > 2) fail("migrate_pack: side must be 0 or 1, dim 0..2");
  if (!migrate_pending_ && nlocal_) {
    SF_HIP(hipMemsetAsync(leave_.ptr, 0, sizeof(int) * nlocal_, stream_));
    migrate_pending_ = true;
  }
  DevArray& list = sendlist_[side];  // reused: the border exchange refills it afterwards
  const int n = select_locals(side == 0 ? 0 : 1, side == 0 ? sublo_[dim] : subhi_[dim], list, dim);
  const int rec = migrate_record_doubles();
  if ((long long)n * rec > max_doubles)
    fail("migrate_pack: %d atoms x %d doubles do not fit the %lld-double buffer", n, rec, max_doubles);
  if (n) {
    MigratePtrs P = mig_ptrs(xr_[cur_], vm_[cur_], om_[cur_], tag_, type_, mask_, foamCpuId_, numneigh_, ptag_,
                             fdrag_, DuDt_, vOld_, wshear_, shear_[hist_buf_], wtouch_, extra_, nextra_);
    k_migrate_pack<<<div_up(n, 128), 128, 0, stream_>>>(list.as<int>(), n, xshift, P, cap_, nwalls_, mrec_,
                                                        have_list_ ? 1 : 0, rec, buf, leave_.as<int>(), side + 1,
                                                        d_flags_, dim);
  }
  migrate_leavers_ += n;
  read_flags();
  if (h_flags_[F_MIG_TRUNC]) {
    const int listed = h_flags_[F_MIG_TRUNC];
    reset_flag(F_MIG_TRUNC, 0);   // (not sticky: the caller may widen the record and try again)
    fail("migrate_pack: an atom lists %d neighbours but the migrate record carries %d history slots (call "
         "sf_dem_migrate_set_slots with the maximum of max_neigh_used over ALL ranks)", listed, mrec_);
  }
  return (long long)n * rec;
}

void DemEngine::migrate_compact()
{
  migrate_pending_ = false;
  const int left = migrate_leavers_;
  migrate_leavers_ = 0;
  if (!nlocal_ || !left) return;   // (the usual rebuild: nobody crossed a face)
  // the staying atoms, in their order, move to the front
  k_stay_keys<<<div_up(nlocal_, 256), 256, 0, stream_>>>(leave_.as<int>(), nlocal_, keys_.as<unsigned>(),
                                                         perm_.as<int>());
  select_zero_keys(sort_tmp_, sort_tmp_bytes_, keys_.as<unsigned>(), perm_alt_.as<int>(), d_flags_ + F_SEND_COUNT2,
                   nlocal_, stream_);
  read_flags();
  const int nstay = h_flags_[F_SEND_COUNT2];
  if (nstay + left != nlocal_) fail("migration: %d atoms stay + %d leave != %d owned", nstay, left, nlocal_);
  permute_locals(perm_alt_.as<int>(), nstay);
  nlocal_ = nstay;
}

void DemEngine::migrate_unpack(const double* buf, long long ndoubles)
{
  if (migrate_pending_) migrate_compact();
  const int rec = migrate_record_doubles();
  if (ndoubles % rec) fail("migrate_unpack: %lld doubles is not a multiple of the %d-double record", ndoubles, rec);
  const int n = (int)(ndoubles / rec);
  if (!n) return;
  ensure_capacity((size_t)nlocal_ + n + 1024);
  MigratePtrs P = mig_ptrs(xr_[cur_], vm_[cur_], om_[cur_], tag_, type_, mask_, foamCpuId_, numneigh_, ptag_, fdrag_,
                           DuDt_, vOld_, wshear_, shear_[hist_buf_], wtouch_, extra_, nextra_);
  k_migrate_unpack<<<div_up(n, 128), 128, 0, stream_>>>(buf, n, nlocal_, P, cap_, nwalls_, mrec_, rec);
  nlocal_ += n;
  order_version_++;
  // tags of immigrants may exceed what this rank has seen
  std::vector<double> hb((size_t)n * rec);
  SF_HIP(hipMemcpyAsync(hb.data(), buf, sizeof(double) * hb.size(), hipMemcpyDeviceToHost, stream_));
  sync();
  for (int k = 0; k < n; k++) {
    max_tag_ = std::max(max_tag_, (int)hb[(size_t)k * rec + 11]);
    rmax_ = std::max(rmax_, hb[(size_t)k * rec + 3]);
  }
}

// ------------------------------------------------------------------------------------------------
// 3-D brick decomposition: direction lists ([3P] Comm::borders / forward_comm on a 3-D processor grid, with every
// ghost sent by its owner in ONE stage instead of LAMMPS' three staged dimensions: one exchange latency per sub-step)
// ------------------------------------------------------------------------------------------------
struct BrickSel {
  int ndir;
  int need[DemEngine::kMaxDirs];   // the face bits (1 << 2 dim = low face, 2 << 2 dim = high face) a direction needs
  double lo[3], hi[3];             // x_d < lo[d]: within the cutoff of the low face; x_d >= hi[d]: of the high face
  int ext[3];
};

__device__ __forceinline__ int face_mask(const double4& x, const BrickSel& B)
{
  int m = 0;
  const double c[3] = {x.x, x.y, x.z};
  for (int d = 0; d < 3; d++) {
    if (!B.ext[d]) continue;
    if (c[d] < B.lo[d]) m |= 1 << (2 * d);
    if (c[d] >= B.hi[d]) m |= 2 << (2 * d);
  }
  return m;
}

// face mask of every owned atom + how many atoms every direction sends (one global atomic per block and direction)
__global__ __launch_bounds__(1024) void k_brick_count(const double4* xr, int n, BrickSel B, unsigned char* mask,
                                                      int* counts)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int m = 0;
  if (i < n) {
    m = face_mask(xr[i], B);
    mask[i] = (unsigned char)m;
  }
  if (!__syncthreads_or(m)) return;   // (most blocks hold no border atom)
  for (int q = 0; q < B.ndir; q++) {
    const int t = block_sum_int_1024((m & B.need[q]) == B.need[q] ? 1 : 0);
    if (threadIdx.x == 0 && t) atomicAdd(&counts[q], t);
  }
}

// the send list of every direction (block q of the concatenated list starts at first[q]); one cursor atomic per
// wave and direction
__global__ __launch_bounds__(256) void k_brick_fill(const unsigned char* mask, int n, BrickSel B,
                                                    DemEngine::BrickBlocks blk, int* cursor, int* list)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int m = i < n ? mask[i] : 0;
  if (!__ballot(m != 0)) return;
  const int lane = threadIdx.x & 63;
  for (int q = 0; q < B.ndir; q++) {
    const bool in = m != 0 && (m & B.need[q]) == B.need[q];
    const unsigned long long b = __ballot(in);
    if (!b) continue;
    int base = 0;
    if (lane == __ffsll((long long)b) - 1) base = atomicAdd(&cursor[q], __popcll(b));
    base = __shfl(base, __ffsll((long long)b) - 1, 64);
    if (in) list[blk.first[q] + base + __popcll(b & ((1ull << lane) - 1ull))] = i;
  }
}

__device__ __forceinline__ int block_of(const DemEngine::BrickBlocks& blk, int k)
{
  int b = 0;
  while (b + 1 < blk.n && k >= blk.first[b + 1]) b++;
  return b;
}

__global__ __launch_bounds__(256) void k_brick_border_pack(const int* list, DemEngine::BrickBlocks blk, const double4* xr,
                                                           const double4* vm, const double4* om, const int* tag,
                                                           const int* type, const int* mask, double* buf)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= blk.first[blk.n]) return;
  const int q = block_of(blk, k);
  const int i = list[k];
  const double4 x = xr[i], v = vm[i], w = om[i];
  double* b = buf + (size_t)k * kBorderDoubles;
  b[0] = x.x + blk.shift[q][0]; b[1] = x.y + blk.shift[q][1]; b[2] = x.z + blk.shift[q][2]; b[3] = x.w;
  b[4] = v.x; b[5] = v.y; b[6] = v.z; b[7] = v.w;
  b[8] = w.x; b[9] = w.y; b[10] = w.z;
  b[11] = (double)tag[i];
  b[12] = (double)type[i];
  b[13] = (double)mask[i];
}

// forward halo of one sub-step: the records of every block go to its place in the per-peer chunks, every chunk's
// header carries this rank's rebuild trigger (the MIN over the headers a rank receives is the global vote)
__global__ __launch_bounds__(256) void k_brick_forward_pack(const int* list, DemEngine::BrickBlocks blk,
                                                            const int* hdr_off, int nhdr, double* sendbuf,
                                                            const int* trig_word, const double4* xr, const double4* vm,
                                                            const double4* om, double* const* blkptr)
{
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int tot = blk.first[blk.n];
  if (k >= tot) {
    k -= tot;
    if (k < nhdr) {
      sendbuf[hdr_off[k]] = 0.0;
      *header_vote_ptr(sendbuf + hdr_off[k]) = __atomic_load_n(trig_word, __ATOMIC_RELAXED);
    }
    return;
  }
  const int q = block_of(blk, k);
  const int i = list[k];
  const double4 x = xr[i], v = vm[i], w = om[i];
  double* b = blkptr[q] + (size_t)(k - blk.first[q]);
  const size_t n = (size_t)(blk.first[q + 1] - blk.first[q]);   // (component-major block: [kForwardDoubles][n])
  // (unshifted, like the records the sub-step kernel writes itself: the receiver adds the block's periodic shift)
  b[0] = x.x; b[n] = x.y; b[2 * n] = x.z;
  b[3 * n] = v.x; b[4 * n] = v.y; b[5 * n] = v.z;
  b[6 * n] = w.x; b[7 * n] = w.y; b[8 * n] = w.z;
}

// where the forward records of a sent atom go: slot s of atom i <- position of its record in block q (arrival order)
__global__ __launch_bounds__(256) void k_brick_slots(const int* list, DemEngine::BrickBlocks blk, int* slots, int* cursor,
                                                     size_t cap)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= blk.first[blk.n]) return;
  const int q = block_of(blk, k);
  const int i = list[k];
  const int s = atomicAdd(&cursor[i], 1);   // (< kBrickSlots: brick_set_forward_tx refuses thinner bricks)
  if (s < kBrickSlots) slots[(size_t)s * cap + i] = (q << kBlkShift) | (k - blk.first[q]);
}

__global__ __launch_bounds__(256) void k_brick_forward_unpack(DemEngine::BrickBlocks blk, const int* hdr_off, int nhdr,
                                                              const double* recvbuf, int* vote_word, int nlocal,
                                                              double4* xr, double4* vm, double4* om)
{
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int tot = blk.first[blk.n];
  if (k >= tot) {
    k -= tot;
    if (k < nhdr) atomicMin(vote_word, header_vote(recvbuf + hdr_off[k]));
    return;
  }
  const int q = block_of(blk, k);
  const double* b = recvbuf + blk.off[q] + (size_t)(k - blk.first[q]);
  const size_t n = (size_t)(blk.first[q + 1] - blk.first[q]);
  const int g = nlocal + k;
  double4 x = xr[g], v = vm[g];
  x.x = b[0] + blk.shift[q][0]; x.y = b[n] + blk.shift[q][1]; x.z = b[2 * n] + blk.shift[q][2];
  v.x = b[3 * n]; v.y = b[4 * n]; v.z = b[5 * n];
  xr[g] = x;   // radius / mass (.w) were set by the border exchange
  vm[g] = v;
  om[g] = {b[6 * n], b[7 * n], b[8 * n], om[g].w};
}

void DemEngine::brick_set_forward_tx(const BrickBlocks& snd, double* sendbuf, const int* hdr_off, int nhdr,
                                     double* const* direct_blk)
{
  static const bool off = getenv("SF_HALO_FUSED_PACK") && !atoi(getenv("SF_HALO_FUSED_PACK"));
  tx_ready_ = tx_written_ = false;
  // where the blocks of an exchange start: the local send buffer, or the neighbours' receive areas (two of them)
  tx_direct_ = direct_blk != nullptr;
  tx_sendbuf_ = sendbuf;
  tx_hdr_off_ = hdr_off;
  tx_nhdr_ = nhdr;
  gs_ready_ = false;
  {
    if (!d_blkptr_) SF_HIP(hipMalloc(&d_blkptr_, sizeof(double*) * 3 * kMaxDirs));
    double* h[3 * kMaxDirs];
    for (int par = 0; par < 2; par++)
      for (int q = 0; q < kMaxDirs; q++)
        h[par * kMaxDirs + q] = q >= snd.n ? nullptr : (direct_blk ? direct_blk[par * kMaxDirs + q] : sendbuf + snd.off[q]);
    for (int q = 0; q < kMaxDirs; q++)   // (third row: records per block)
      reinterpret_cast<size_t*>(h)[2 * kMaxDirs + q] = q >= snd.n ? 0 : (size_t)(snd.first[q + 1] - snd.first[q]);
    SF_HIP(hipMemcpyAsync(d_blkptr_, h, sizeof(h), hipMemcpyHostToDevice, stream_));
    SF_HIP(hipStreamSynchronize(stream_));   // (h is on the stack)
  }
  for (int q = 0; q < snd.n; q++)
    if ((long long)(snd.first[q + 1] - snd.first[q]) > (long long)kBlkMask)
      fail("brick_set_forward_tx: a send block of %d atoms does not fit the record slot encoding", snd.first[q + 1] - snd.first[q]);
  if (off || !nlocal_) return;
  // A brick thinner than twice the ghost cutoff in an external dimension has atoms that are ghosts of BOTH neighbours
  // along it: three choices in that dimension instead of two, up to 3 * 2 * 2 - 1 = 11 (17, 26) send directions per
  // atom -- more than the kBrickSlots = 7 records the sub-step kernel writes itself.  Such a rank keeps the stand-alone
  // pack kernel in front of every exchange (same send buffer; the neighbours cannot tell).
  const double cut = cutneighmax();
  for (int d = 0; d < 3; d++)
    if (ext_[d] && subhi_[d] - sublo_[d] < 2.0 * cut) return;
  if (bslot_.cap != cap_) bslot_.alloc(sizeof(int), kBrickSlots + 1, cap_, stream_);   // (row stride = P.cap)
  SF_HIP(hipMemsetAsync(bslot_.ptr, 0xFF, sizeof(int) * kBrickSlots * bslot_.cap, stream_));
  int* cursor = bslot_.as<int>() + (size_t)kBrickSlots * bslot_.cap;
  SF_HIP(hipMemsetAsync(cursor, 0, sizeof(int) * bslot_.cap, stream_));
  const int tot = snd.first[snd.n];
  if (snd.n && (size_t)(snd.off[snd.n - 1] + (long long)tot * kForwardDoubles) >= (size_t)INT_MAX)
    fail("brick_set_forward_tx: send buffer beyond 2^31 doubles");
  if (tot) k_brick_slots<<<div_up(tot, 256), 256, 0, stream_>>>(bsend_list_, snd, bslot_.as<int>(), cursor, bslot_.cap);
  tx_ready_ = true;
}

void DemEngine::brick_set_dirs(int ndir, const int* d3, const double* shift3)
{
  if (ndir < 0 || ndir > kMaxDirs) fail("brick_set_dirs: %d directions", ndir);
  bndir_ = ndir;
  bsend_blocks_.n = ndir;
  for (int q = 0; q < ndir; q++)
    for (int k = 0; k < 3; k++) {
      bdir_[q][k] = d3[3 * q + k];
      if (bdir_[q][k] && !ext_[k]) fail("brick_set_dirs: direction %d points through a face that is not external", q);
      bsend_blocks_.shift[q][k] = shift3[3 * q + k];
    }
  if (!d_bcount_) {
    SF_HIP(hipMalloc(&d_bcount_, sizeof(int) * 2 * kMaxDirs));
    SF_HIP(hipHostMalloc(&h_bcount_, sizeof(int) * 2 * kMaxDirs));
  }
}

void DemEngine::brick_border_select(long long* counts)
{
  BrickSel B;
  B.ndir = bndir_;
  const double cut = cutneighmax();
  for (int d = 0; d < 3; d++) {
    B.ext[d] = ext_[d] ? 1 : 0;
    B.lo[d] = sublo_[d] + cut;
    B.hi[d] = subhi_[d] - cut;
    // every ghost comes from an ADJACENT brick: a brick thinner than the ghost cutoff would need its second neighbours
    if (ext_[d] && subhi_[d] - sublo_[d] < cut)
      fail("brick decomposition: the sub-domain is %.6g wide in dimension %d, less than the ghost cutoff %.6g -- use "
           "fewer bricks along it", subhi_[d] - sublo_[d], d, cut);
  }
  for (int q = 0; q < bndir_; q++) {
    int need = 0;
    for (int d = 0; d < 3; d++) need |= bdir_[q][d] < 0 ? 1 << (2 * d) : (bdir_[q][d] > 0 ? 2 << (2 * d) : 0);
    B.need[q] = need;
  }
  for (int q = 0; q <= bndir_; q++) bsend_blocks_.first[q] = 0;
  tx_ready_ = tx_written_ = false;
  if (nlocal_ && bndir_) {
    SF_HIP(hipMemsetAsync(d_bcount_, 0, sizeof(int) * 2 * kMaxDirs, stream_));
    k_brick_count<<<div_up(nlocal_, 1024), 1024, 0, stream_>>>(xr_[cur_].as<double4>(), nlocal_, B,
                                                             bmask_.as<unsigned char>(), d_bcount_);
    SF_HIP(hipMemcpyAsync(h_bcount_, d_bcount_, sizeof(int) * kMaxDirs, hipMemcpyDeviceToHost, stream_));
    sync();
    for (int q = 0; q < bndir_; q++) bsend_blocks_.first[q + 1] = bsend_blocks_.first[q] + h_bcount_[q];
    const size_t tot = (size_t)bsend_blocks_.first[bndir_];
    if (tot > bsend_alloc_) {
      if (bsend_list_) SF_HIP(hipFree(bsend_list_));
      bsend_alloc_ = tot + tot / 4 + 1024;
      SF_HIP(hipMalloc(&bsend_list_, sizeof(int) * bsend_alloc_));
    }
    if (tot)
      k_brick_fill<<<div_up(nlocal_, 256), 256, 0, stream_>>>(bmask_.as<unsigned char>(), nlocal_, B, bsend_blocks_,
                                                             d_bcount_ + kMaxDirs, bsend_list_);
  }
  for (int q = 0; q < bndir_; q++) counts[q] = bsend_blocks_.first[q + 1] - bsend_blocks_.first[q];
}

void DemEngine::brick_border_pack(double* buf)
{
  const int tot = bsend_blocks_.first[bndir_];
  if (tot)
    k_brick_border_pack<<<div_up(tot, 256), 256, 0, stream_>>>(bsend_list_, bsend_blocks_, xr_[cur_].as<double4>(),
                                                              vm_[cur_].as<double4>(), om_[cur_].as<double4>(),
                                                              tag_.as<int>(), type_.as<int>(), mask_.as<int>(), buf);
}

void DemEngine::brick_ghost_unpack(const double* buf, long long natoms)
{
  const int n = (int)natoms;
  ensure_capacity((size_t)nlocal_ + next_ghost_ + n + 1024);
  const int first = nlocal_ + next_ghost_;
  if (n)
    k_border_unpack<<<div_up(n, 256), 256, 0, stream_>>>(buf, n, first, xr_[cur_].as<double4>(),
                                                         vm_[cur_].as<double4>(), om_[cur_].as<double4>(),
                                                         xr_[cur_ ^ 1].as<double4>(), vm_[cur_ ^ 1].as<double4>(),
                                                         om_[cur_ ^ 1].as<double4>(), tag_.as<int>(), type_.as<int>(),
                                                         mask_.as<int>(), gsrc_.as<int>(), freeze_bit_);
  next_ghost_ += n;
}

void DemEngine::brick_forward_pack(const BrickBlocks& snd, double* sendbuf, const int* hdr_off, int nhdr)
{
  if (tx_direct_) nhdr = 0;   // (the votes travel with the flags of brick_direct_unpack)
  const int tot = snd.first[snd.n] + nhdr;
  if (tot)
    k_brick_forward_pack<<<div_up(tot, 256), 256, 0, stream_>>>(bsend_list_, snd, hdr_off, nhdr, sendbuf,
                                                               d_flags_ + F_TRIGGER, xr_[cur_].as<double4>(),
                                                               vm_[cur_].as<double4>(), om_[cur_].as<double4>(),
                                                               d_blkptr_ + (size_t)tx_par_ * kMaxDirs);
}

// Direct ghost writes: the records of this exchange were written into the receive areas by the NEIGHBOURS' sub-step
// kernels (IPC mappings; sf_halo_rccl.hip).  This kernel is everything that stands between two sub-step kernels:
//   1. workgroup 0 tells every rank that this rank's records and vote of exchange `seq` are in place: the vote into
//      the peer's vote[par][me], then seq into its flag[me] with a system-scope release (the sub-step kernel that wrote
//      the records is complete: stream order);
//   2. every workgroup waits (acquire, bounded by max_ticks) until all ranks have said the same here;
//   3. trigger word <- min of the votes; received records -> ghost slots (+ the block's periodic shift).
__global__ __launch_bounds__(256) void k_brick_direct_unpack(DemEngine::BrickBlocks blk, const double* recvarea,
                                                             DemEngine::DirectSync D, int* flags, int nlocal, double4* xr,
                                                             double4* vm, double4* om)
{
  __shared__ int ok;
  const int W = D.world;
  if (blockIdx.x == 0 && (int)threadIdx.x < W && (int)threadIdx.x != D.rank) {
    // (every sender has a 128-byte line of its own in the receiver's area: {flag, vote[2]}.  Eight ranks' words on ONE
    // line went wrong on eight XCDs: a flag was seen at 61 and later at 57 -- the per-XCD L2s are not coherent with each
    // other, a line has to have one writer)
    int* peer = D.peer_sync[threadIdx.x] + DemEngine::kSyncStride * D.rank;
    __hip_atomic_store(&peer[1 + D.par], __atomic_load_n(&flags[F_TRIGGER], __ATOMIC_RELAXED), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&peer[0], D.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (threadIdx.x == 0) ok = 1;
  __syncthreads();
  if ((int)threadIdx.x < W && (int)threadIdx.x != D.rank) {
    const long long t0 = wall_clock64();
    // (32-bit exchange numbers compared by their difference: the count may wrap)
    while ((int)((unsigned)__hip_atomic_load(&D.my_sync[DemEngine::kSyncStride * threadIdx.x], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - (unsigned)D.seq) < 0) {
      __builtin_amdgcn_s_sleep(4);
      if (wall_clock64() - t0 > D.max_ticks) {
        ok = 0;
        // (which rank's flag was missing and the value it had: two words, the exchange number outgrows any packing)
        flags[F_HALO_TIMEOUT_PEER] = (int)threadIdx.x;
        flags[F_HALO_TIMEOUT_SEEN] = __hip_atomic_load(&D.my_sync[DemEngine::kSyncStride * threadIdx.x], __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
  }
  __syncthreads();
  if (!ok) {
    if (threadIdx.x == 0) flags[F_HALO_TIMEOUT] = D.seq;
    return;
  }
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int tot = blk.first[blk.n];
  if (k >= tot) {
    k -= tot;
    if (k < W && k != D.rank)
      atomicMin(&flags[F_TRIGGER], __hip_atomic_load(&D.my_sync[DemEngine::kSyncStride * k + 1 + D.par], __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_SYSTEM));
    return;
  }
  const int q = block_of(blk, k);
  const double* b = recvarea + blk.off[q] + (size_t)(k - blk.first[q]);
  const size_t n = (size_t)(blk.first[q + 1] - blk.first[q]);
  const int g = nlocal + k;
  double4 x = xr[g], v = vm[g];
  x.x = b[0] + blk.shift[q][0]; x.y = b[n] + blk.shift[q][1]; x.z = b[2 * n] + blk.shift[q][2];
  v.x = b[3 * n]; v.y = b[4 * n]; v.z = b[5 * n];
  xr[g] = x;
  vm[g] = v;
  om[g] = {b[6 * n], b[7 * n], b[8 * n], om[g].w};
}

void DemEngine::brick_direct_unpack(const BrickBlocks& rcv, const double* recvarea, const DirectSync& D)
{
  if (rcv.first[rcv.n] != next_ghost_)
    fail("brick_direct_unpack: %d ghosts in the layout, the border exchange set up %d", rcv.first[rcv.n], next_ghost_);
  if (D.world > 32) fail("brick_direct_unpack: %d ranks (at most 32)", D.world);
  const int tot = rcv.first[rcv.n] + D.world;
  k_brick_direct_unpack<<<div_up(tot, 256), 256, 0, stream_>>>(rcv, recvarea, D, d_flags_, nlocal_,
                                                              xr_[cur_].as<double4>(), vm_[cur_].as<double4>(),
                                                              om_[cur_].as<double4>());
  launch_ghost_forward(cur_, INT_MIN);   // (index mode only: local images of everything, received ghosts included)
}

// ------------------------------------------------------------------------------------------------
// ghost slots: the stand-alone pack (start of a run, first launch after a rebuild), the flag, the end of a piece
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gs_pack(const int* list, DemEngine::BrickBlocks blk, double* const* blk3,
                                                 const double* blkshift, const double4* xr, const double4* vm,
                                                 const double4* om)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= blk.first[blk.n]) return;
  const int q = block_of(blk, k);
  const int i = list[k];
  const double4 x = xr[i], v = vm[i], w = om[i];
  const size_t r = (size_t)(k - blk.first[q]);
  const double* sh = blkshift + 3 * q;
  gs_store(reinterpret_cast<double4*>(blk3[q]) + r, x.x + sh[0], x.y + sh[1], x.z + sh[2], x.w);
  gs_store(reinterpret_cast<double4*>(blk3[DemEngine::kMaxDirs + q]) + r, v.x, v.y, v.z, v.w);
  gs_store(reinterpret_cast<double4*>(blk3[2 * DemEngine::kMaxDirs + q]) + r, w.x, w.y, w.z, w.w);
}

__global__ __launch_bounds__(64) void k_gs_publish(const GsSync* Y, int* flags, int seq)
{
  int vote = 0;
  if (threadIdx.x == 0) vote = atomicMin(&flags[F_TRIGGER], INT_MAX);
  vote = __shfl(vote, 0, 64);
  gs_publish(Y, vote, seq);
}

__global__ __launch_bounds__(64) void k_gs_close(const GsSync* Y, int* flags, int seq, int kstep_end)
{
  if (__atomic_load_n(&flags[F_TRIGGER], __ATOMIC_RELAXED) < kstep_end) return;   // (known: nobody may publish `seq`)
  if (__atomic_load_n(&flags[F_HALO_TIMEOUT], __ATOMIC_RELAXED) != 0) return;
  (void)gs_gate(Y, flags, seq, kstep_end);
}

void DemEngine::gs_configure(const GsSync& sync, long long first_seq)
{
  gs_seq_ = first_seq;
  h_gs_sync_ = sync;
  if (!d_gs_sync_) SF_HIP(hipMalloc(&d_gs_sync_, sizeof(GsSync)));
  if (!d_gs_count_) SF_HIP(hipMalloc(&d_gs_count_, sizeof(int) * 8 * 32));
  SF_HIP(hipMemcpyAsync(d_gs_sync_, &sync, sizeof(GsSync), hipMemcpyHostToDevice, stream_));
  SF_HIP(hipMemsetAsync(d_gs_count_, 0, sizeof(int) * 8 * 32, stream_));
  SF_HIP(hipStreamSynchronize(stream_));   // (sync is the caller's)
}

bool DemEngine::brick_fused_pack_possible() const
{
  if (getenv("SF_HALO_FUSED_PACK") && !atoi(getenv("SF_HALO_FUSED_PACK"))) return false;
  // (the LDS-staged tile kernel neither writes border records nor runs the hand-off: the ranks decide this BEFORE they
  // agree on a transport -- gs_rebuild / direct_rebuild all-reduce it --, whatever build_stage_tables later finds per rank)
  if (opt_lds_) return false;
  const double cut = cutneighmax();
  for (int d = 0; d < 3; d++)
    if (ext_[d] && subhi_[d] - sublo_[d] < 2.0 * cut) return false;
  return true;
}

void DemEngine::gs_off() { gs_ready_ = false; }

void DemEngine::gs_input_arrays(int par, void** x, void** v, void** w) const
{
  // launch gs_seq_ reads buffer cur_; every launch flips both
  const int b = cur_ ^ (int)(((long long)(par & 1) - gs_seq_) & 1);
  *x = xr_[b].ptr;
  *v = vm_[b].ptr;
  *w = om_[b].ptr;
}

void DemEngine::brick_set_forward_gs(const BrickBlocks& snd, double4* const* blk6)
{
  if (!d_gs_sync_) fail("brick_set_forward_gs: gs_configure first");
  // (the record-slot table of the border atoms: as for the other transports)
  brick_set_forward_tx(snd, nullptr, nullptr, 0, nullptr);
  if (!tx_ready_ && nlocal_)
    fail("ghost slots need the sub-step kernel to write the border records itself: a brick thinner than twice the ghost "
         "cutoff (or SF_HALO_FUSED_PACK=0) cannot use SF_HALO_DIRECT=2");
  struct {
    double* ptr[6 * kMaxDirs];
    double shift[3 * kMaxDirs];
  } h;
  for (int k = 0; k < 6; k++)
    for (int q = 0; q < kMaxDirs; q++)
      h.ptr[k * kMaxDirs + q] = q < snd.n ? reinterpret_cast<double*>(blk6[k * kMaxDirs + q]) : nullptr;
  for (int q = 0; q < kMaxDirs; q++)
    for (int k = 0; k < 3; k++) h.shift[3 * q + k] = q < snd.n ? snd.shift[q][k] : 0.0;
  if (!d_gsblk_) SF_HIP(hipMalloc(&d_gsblk_, sizeof(h)));
  SF_HIP(hipMemcpyAsync(d_gsblk_, &h, sizeof(h), hipMemcpyHostToDevice, stream_));
  SF_HIP(hipMemsetAsync(d_gs_count_, 0, sizeof(int) * 8 * 32, stream_));
  SF_HIP(hipStreamSynchronize(stream_));   // (h is on the stack)
  tx_direct_ = true;    // (no vote headers in a send buffer)
  tx_written_ = false;
  gs_map_base_ = cur_ ^ (int)(gs_seq_ & 1);
  gs_ready_ = true;
}

void DemEngine::gs_pack()
{
  if (!gs_ready_) fail("gs_pack: no ghost-slot layout (rebuild first)");
  // A launch number must never be published twice: the last kernel that ran has published gs_seq_ ("my records for launch
  // gs_seq_ are in place") when no early-exited launch followed it -- a trigger in the last sub-step of a piece, then the
  // rebuild, then this pack -- and a neighbour that still sees that flag would pass its gate before these records and
  // this vote have arrived.  Two numbers on: the parity of the buffers against the launch numbers stays what the
  // neighbours were told (every rank packs at the same points).
  gs_seq_ += 2;
  const int par = (int)(gs_seq_ & 1);
  const int tot = bsend_blocks_.first[bsend_blocks_.n];
  if (tot)
    k_gs_pack<<<div_up(tot, 256), 256, 0, stream_>>>(bsend_list_, bsend_blocks_, d_gsblk_ + (size_t)par * 3 * kMaxDirs,
                                                    reinterpret_cast<const double*>(d_gsblk_ + 6 * (size_t)kMaxDirs),
                                                    xr_[cur_].as<double4>(), vm_[cur_].as<double4>(), om_[cur_].as<double4>());
  k_gs_publish<<<1, 64, 0, stream_>>>(d_gs_sync_, d_flags_, (int)gs_seq_);
  tx_written_ = true;
}

void DemEngine::gs_close(int kstep_end)
{
  if (!gs_ready_) fail("gs_close: no ghost-slot layout");
  k_gs_close<<<1, 64, 0, stream_>>>(d_gs_sync_, d_flags_, (int)gs_seq_, kstep_end);
}

// returns false when a peer's flag did not arrive in time; the timeout words are cleared again either way (a probe that
// failed makes the caller fall back to RCCL: the stepping loop must not find a stale F_HALO_TIMEOUT afterwards)
bool DemEngine::brick_direct_probe(const BrickBlocks& none, const DirectSync& D)
{
  read_flags();
  const int keep = h_flags_[F_TRIGGER];
  reset_flag(F_HALO_TIMEOUT, 0);
  k_brick_direct_unpack<<<1, 256, 0, stream_>>>(none, nullptr, D, d_flags_, 0, nullptr, nullptr, nullptr);
  read_flags();
  const bool timed_out = h_flags_[F_HALO_TIMEOUT] != 0;
  reset_flags(F_HALO_TIMEOUT, 3, 0);   // F_HALO_TIMEOUT, _PEER, _SEEN
  reset_flag(F_TRIGGER, keep);         // (the probe's votes carried whatever the word held: put it back)
  return !timed_out;
}

void DemEngine::brick_forward_unpack(const BrickBlocks& rcv, const double* recvbuf, const int* hdr_off, int nhdr)
{
  if (rcv.first[rcv.n] != next_ghost_)
    fail("brick_forward_unpack: %d ghosts in the layout, the border exchange set up %d", rcv.first[rcv.n], next_ghost_);
  const int tot = rcv.first[rcv.n] + nhdr;
  if (tot)
    k_brick_forward_unpack<<<div_up(tot, 256), 256, 0, stream_>>>(rcv, hdr_off, nhdr, recvbuf, d_flags_ + F_TRIGGER,
                                                                 nlocal_, xr_[cur_].as<double4>(),
                                                                 vm_[cur_].as<double4>(), om_[cur_].as<double4>());
  launch_ghost_forward(cur_, INT_MIN);   // (index mode only: local images of everything, received ghosts included)
}

// owned atoms outside the brick in any external dimension
__global__ __launch_bounds__(1024) void k_count_outside3(const double4* xr, int n, BrickSel B, int* counter)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool out = false;
  if (i < n) {
    const double4 x = xr[i];
    const double c[3] = {x.x, x.y, x.z};
    for (int d = 0; d < 3; d++) out = out || (B.ext[d] && (c[d] < B.lo[d] || c[d] >= B.hi[d]));
  }
  const int t = block_sum_int_1024(out ? 1 : 0);
  if (threadIdx.x == 0 && t) atomicAdd(counter, t);
}

long long DemEngine::migrate_count3()
{
  if (!nlocal_) return 0;
  BrickSel B;
  B.ndir = 0;
  for (int d = 0; d < 3; d++) {
    B.ext[d] = ext_[d] ? 1 : 0;
    B.lo[d] = sublo_[d];
    B.hi[d] = subhi_[d];
  }
  reset_flag(F_SEND_COUNT, 0);
  k_count_outside3<<<div_up(nlocal_, 1024), 1024, 0, stream_>>>(xr_[cur_].as<double4>(), nlocal_, B,
                                                              d_flags_ + F_SEND_COUNT);
  read_flags();
  return h_flags_[F_SEND_COUNT];
}

// ------------------------------------------------------------------------------------------------
// lammps_create_particle / lammps_delete_particle (library.cpp:406-621)
// ------------------------------------------------------------------------------------------------
void DemEngine::create_particles(int np, const double* pos, const double* tag, double diameter, double rho, int type,
                                 const double* vel)
{
  if (np <= 0) return;
  ensure_capacity((size_t)nlocal_ + nghost_ + np + 1024);
  const double r = 0.5 * diameter;
  // library.cpp:460 -- the reference's mistyped pi literal is kept
  const double m = 4.0 * kPiTypo / 3.0 * r * r * r * rho;
  std::vector<double4> hx(np), hv(np), hw(np, double4{0, 0, 0, 0}), hz(np, double4{0, 0, 0, 0});
  // library.cpp:452-455: mask = 1 | bitmask of group "active" (the reference indexes bitmask[-1] when the group does
  // not exist; here the new atoms are then in `all` only)
  int newmask = 1;
  {
    auto it = groups_.find("active");
    if (it != groups_.end()) newmask |= it->second;
  }
  std::vector<int> ht(np), hty(np, type), hm(np, newmask), h0(np, 0);
  for (int k = 0; k < np; k++) {
    hx[k] = {pos[3 * k], pos[3 * k + 1], pos[3 * k + 2], r};
    hv[k] = {vel[0], vel[1], vel[2], m};
    ht[k] = (int)tag[k];  // library.cpp:437
    max_tag_ = std::max(max_tag_, ht[k]);
  }
  rmax_ = std::max(rmax_, r);
  // ghosts are dropped (library.cpp:476-480 resets nghost) and re-created by the forced rebuild below
  const size_t o4 = sizeof(double4) * nlocal_, oi = sizeof(int) * nlocal_;
  auto up = [&](DevArray& a, const void* src, size_t bytes, size_t off) {
    SF_HIP(hipMemcpyAsync((char*)a.ptr + off, src, bytes, hipMemcpyHostToDevice, stream_));
  };
  up(xr_[cur_], hx.data(), sizeof(double4) * np, o4);
  up(vm_[cur_], hv.data(), sizeof(double4) * np, o4);
  up(om_[cur_], hw.data(), sizeof(double4) * np, o4);
  up(force_, hz.data(), sizeof(double4) * np, o4);
  up(torque_, hz.data(), sizeof(double4) * np, o4);
  up(tag_, ht.data(), sizeof(int) * np, oi);
  up(type_, hty.data(), sizeof(int) * np, oi);
  up(mask_, hm.data(), sizeof(int) * np, oi);
  up(foamCpuId_, h0.data(), sizeof(int) * np, oi);
  up(numneigh_, h0.data(), sizeof(int) * np, oi);
  std::vector<double> z(np, 0.0);
  for (int c = 0; c < 3; c++) {
    up(fdrag_, z.data(), sizeof(double) * np, sizeof(double) * ((size_t)c * cap_ + nlocal_));
    up(DuDt_, z.data(), sizeof(double) * np, sizeof(double) * ((size_t)c * cap_ + nlocal_));
    up(vOld_, z.data(), sizeof(double) * np, sizeof(double) * ((size_t)c * cap_ + nlocal_));
  }
  std::vector<unsigned char> zb(np, 0);
  up(wtouch_, zb.data(), np, nlocal_);
  for (int r = 0; r < nextra_; r++) {   // client rows: the value registered for atoms that did not exist before
    std::vector<double> iv(np, extra_init_[r]);
    up(extra_, iv.data(), sizeof(double) * np, sizeof(double) * ((size_t)r * cap_ + nlocal_));
    sync();
  }
  sync();
  nlocal_ += np;
  order_version_++;
  nghost_ = 0;
  // next_reneighbor = ntimestep + 1 for every fix (library.cpp:482-486): rebuild before the next force
  if (setup_done_ && !have_subdomain_) rebuild();
}

void DemEngine::delete_particles(const int* tags, int n)
{
  if (n <= 0 || !nlocal_) return;
  std::vector<int> ht(nlocal_), leave(nlocal_, 0);
  SF_HIP(hipMemcpyAsync(ht.data(), tag_.ptr, sizeof(int) * nlocal_, hipMemcpyDeviceToHost, stream_));
  sync();
  std::vector<int> del(tags, tags + n);
  std::sort(del.begin(), del.end());
  int ndel = 0;
  for (int i = 0; i < nlocal_; i++)
    if (std::binary_search(del.begin(), del.end(), ht[i])) {
      leave[i] = 1;
      ndel++;
    }
  if (!ndel) return;
  if (have_list_) {
    // the history of the surviving atoms must outlive the compaction
    compute_partner_tags();
  }
  SF_HIP(hipMemcpyAsync(leave_.ptr, leave.data(), sizeof(int) * nlocal_, hipMemcpyHostToDevice, stream_));
  migrate_leavers_ = ndel;
  migrate_compact();
  nghost_ = 0;
  if (setup_done_ && !have_subdomain_) {
    // list rows are already partner tags: rebuild without recomputing them
    compute_grid();
    next_ghost_ = 0;
    rebuild_sort();
    rebuild_finish();
  }
}

}  // namespace sf
