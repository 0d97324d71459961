// sf_halo_rccl.hip -- the per-sub-step halo loop of a decomposed domain, queued entirely from C++ over RCCL.
//
// Reference counterpart: LAMMPS' Comm::forward_comm (MPI_Sendrecv per swap, [3P] comm.cpp) inside Verlet::run,
// plus the MPI_Allreduce of Neighbor::decide.  Here one grouped ncclSend/ncclRecv per sub-step carries the ghost
// records to the two face neighbours AND one header word (the rebuild trigger) to every rank, on the engine's
// streams, so a whole lammps_step(n) is queued without the host waiting (or a Python interpreter in the loop).
// librccl is dlopen'ed on first use: inside a PyTorch process that resolves to the RCCL torch already loaded
// (one RCCL per process), elsewhere to /opt/rocm/lib.  Engines that never call sf_dem_comm_init do not load it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <map>
#include <array>

#include "../../include/sedifoam_amd.h"
#include "sf_handles.h"
#include "sf_roctx.h"

namespace sf {

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  char path[256] = {0};   // the library these symbols come from (dladdr of ncclSend)
};

static RcclApi& rccl()
{
  static RcclApi api;
  if (api.lib) return api;
  // SF_RCCL_LIB: another build of the library -- or the tests' stand-in that moves the messages of several ranks
  // sharing ONE GPU through host memory (tests/c_abi/standin_rccl.cpp; RCCL itself refuses two ranks on one device)
  if (const char* over = getenv("SF_RCCL_LIB")) {
    api.lib = dlopen(over, RTLD_NOW | RTLD_LOCAL);
    if (!api.lib) fail("SF_RCCL_LIB=%s: %s", over, dlerror());
  }
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    if (api.lib) break;
    api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
  }
  if (!api.lib) fail("cannot load librccl.so.1 (%s)", dlerror());
  auto sym = [&](const char* n) {
    void* p = dlsym(api.lib, n);
    if (!p) fail("librccl: symbol %s not found", n);
    return p;
  };
  api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
  api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
  api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
  api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
  api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
  api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
  api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
  api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
  api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
  api.CommCount = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
  api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(sym("ncclGetVersion"));
  Dl_info di;
  if (dladdr(reinterpret_cast<void*>(api.Send), &di) && di.dli_fname) snprintf(api.path, sizeof(api.path), "%s", di.dli_fname);
  return api;
}

#define SF_NCCL(call)                                                                              \
  do {                                                                                             \
    ncclResult_t r_ = (call);                                                                      \
    if (r_ != ncclSuccess) ::sf::fail("RCCL error %s at %s:%d", ::sf::rccl().GetErrorString(r_), __FILE__, __LINE__); \
  } while (0)

// SF_HALO_FAKE_DELAY_US (development knob): one wave spins for that long on the exchange's stream, standing in
// for the xGMI transfer time of a real neighbour when a single GPU sends its periodic images to itself
__global__ void k_fake_link_delay(long long ticks)
{
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// a device buffer of doubles that only grows
struct GrowBuf {
  double* p = nullptr;
  size_t n = 0;
  double* need(size_t count)
  {
    if (count > n) {
      if (p) (void)hipFree(p);
      n = count + count / 4 + 1024;
      SF_HIP(hipMalloc(&p, sizeof(double) * n));
    }
    return p;
  }
  ~GrowBuf()
  {
    if (p) (void)hipFree(p);
  }
};

// ---- 3-D brick decomposition (sf_brick_init): processor grid P, direct exchange with up to 26 neighbour bricks ----
struct BrickState {
  int P[3] = {1, 1, 1}, c[3] = {0, 0, 0};
  double lo[3], hi[3];
  int periodic[3];
  bool ext[3];
  struct SDir {   // a send direction: the neighbour brick at offset d
    int d[3], code, peer;
    double shift[3];
  };
  struct RDir {   // a block this rank receives: what the neighbour at offset s sends in ITS direction -s
    int peer, sender_code;
  };
  std::vector<SDir> sdirs;   // sorted by (peer, code): the blocks for one peer are contiguous, in the sender's order
  std::vector<RDir> rdirs;   // sorted by (peer, sender's code): the same order seen from the receiving side
  int nbr[3][2];             // face neighbours per dimension (-1: none), for the staged migration
  double mshift[3][2];
  std::vector<long long> nsend, nrecv;   // atoms per block
  DemEngine::BrickBlocks snd{}, rcv{};
  long long* d_cnt = nullptr;   // [2 * 26] block sizes: mine, then the neighbours'
  long long* h_cnt = nullptr;
  GrowBuf btx, brx;
  ~BrickState()
  {
    if (d_cnt) (void)hipFree(d_cnt);
    if (h_cnt) (void)hipHostFree(h_cnt);
  }
};

// ---- direct ghost writes (brick driver): the sub-step kernel of a rank writes the forward records of its border atoms
// straight into the receive area of the neighbour's process -- an IPC mapping of fine-grained device memory -- and one
// kernel per exchange publishes / awaits a flag word per rank (DemEngine::brick_direct_unpack).  No RCCL kernel, no
// separate pack or unpack pass between two sub-steps.  RCCL still carries everything rare: the rebuild-time migration
// and border exchange, the reductions, and the IPC handles themselves.
struct DirectHalo {
  bool on = false;
  bool must = false;                    // SF_HALO_DIRECT asked for it without "auto": losing it is an error
  // 1: receive areas + one unpack kernel per exchange; 2: GHOST SLOTS -- the neighbours' sub-step kernels write whole
  // records into the ghost range of this rank's record arrays (IPC mappings of the arrays themselves): no kernel between
  // two sub-step kernels (gs_rebuild, DemEngine::brick_set_forward_gs, sf_dem_gs.h)
  int mode = 1;
  int* my_sync = nullptr;               // one 128-byte line per sending rank: {flag, vote[2]}; peers write, this rank polls
  std::vector<int*> peer_sync;          // every rank's area as mapped here
  double* rx[2] = {nullptr, nullptr};   // receive areas (exchange parity); neighbours write, this rank unpacks
  size_t rx_cap = 0;                    // doubles (mode 1) / records per array (mode 2)
  long long rx_gen = 0;                 // bumped when the areas are re-allocated: the neighbours re-open their mappings
  struct Peer {
    long long gen_seen = -1;
    void* map[2] = {nullptr, nullptr};  // that rank's receive areas as mapped here
    long long remote_off = 0;           // where this rank's chunk starts in them (doubles)
    // ghost slots: that rank's record arrays as mapped here, by IPC handle (an allocation is opened once; the owner may
    // retire it -- the mapping then only keeps the memory alive until this object goes)
    std::map<std::array<char, 64>, void*> opened;
  };
  std::vector<Peer> peers;
  long long xseq = 0;                   // exchanges done so far (the same number on every rank)
  long long* d_msg = nullptr;           // [2][world][kMsg] handle / offset messages (device, for the RCCL exchange)
  long long* h_msg = nullptr;
  static constexpr int kMsg = 50;       // generation, two 64-byte IPC handles, chunk offset; ghost slots: six handles, first ghost
  ~DirectHalo()
  {
    for (Peer& p : peers) {
      for (void* m : p.map)
        if (m && m != rx[0] && m != rx[1]) (void)hipIpcCloseMemHandle(m);
      for (auto& kv : p.opened) (void)hipIpcCloseMemHandle(kv.second);
    }
    for (size_t r = 0; r < peer_sync.size(); r++)
      if (peer_sync[r] && peer_sync[r] != my_sync) (void)hipIpcCloseMemHandle(peer_sync[r]);
    if (my_sync) (void)hipFree(my_sync);
    for (double* b : rx)
      if (b) (void)hipFree(b);
    if (d_msg) (void)hipFree(d_msg);
    if (h_msg) (void)hipHostFree(h_msg);
  }
};

struct HaloComm {
  ncclComm_t comm = nullptr;
  BrickState* brick = nullptr;
  DirectHalo* direct = nullptr;
  int rank = 0, world = 1;
  hipEvent_t ev_boundary = nullptr, ev_halo = nullptr;
  // ---- the slab driver (sf_slab_*): what sedifoam_amd/halo.py SlabDriver does, in C++ ----
  bool slab = false, periodic_x = true, is_setup = false;
  double box_lo = 0.0, box_len = 1.0;
  int left = -1, right = -1;                 // face neighbours (-1: none, a non-periodic box end)
  double shift_left = 0.0, shift_right = 0.0;
  long long nsend[2] = {0, 0}, nrecv[2] = {0, 0};   // border atoms to / from the left, right neighbour
  long long n_rebuilds = 0;
  GrowBuf mig[2], bor[2], rx[2], a2a_tx, a2a_rx;
  long long* d_counts = nullptr;             // [4] device: counts to left, to right ; from left, from right
  long long* h_counts = nullptr;             // pinned twin
  double* d_red = nullptr;                   // [4] all-reduce scratch (in, out)
  double* h_red = nullptr;
  int* d_hdr = nullptr;                      // [2 * world] header offsets: send, receive
  int* h_hdr = nullptr;                      // pinned twin
  std::vector<long long> send_off, send_cnt, recv_off, recv_cnt;
  sf_halo_layout lay{};
  bool lay_valid = false;
  // what the forward exchange costs on this rank's stream (profiling on: an event pair around every 8th exchange of
  // the serial loop, from the end of the sub-step kernel before it to the end of the unpack kernel)
  std::vector<hipEvent_t> xev;
  size_t xev_used = 0;
  long long x_count = 0;
  double x_ms = 0.0;
  RebuildPredictor predict;  // how far slab_step queues (every sub-step queued behind a trigger is a wasted exchange)
  bool pre_exchanged = false;   // the exchange in front of the next sub-step has already run (end of the last piece)
  double rebuild_ms = 0.0;   // host wall time inside slab_rebuild (it ends synchronised), rebuilds after the setup's
  long long rebuilds_at_setup = 0;
  void harvest_exchange_profile()
  {
    for (size_t q = 0; 2 * q + 1 < xev_used; q++) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, xev[2 * q], xev[2 * q + 1]) == hipSuccess) {
        x_ms += ms;
        x_count++;
      }
    }
    xev_used = 0;
  }
  ~HaloComm()
  {
    delete direct;
    delete brick;
    if (comm) (void)rccl().CommDestroy(comm);
    if (ev_boundary) (void)hipEventDestroy(ev_boundary);
    if (ev_halo) (void)hipEventDestroy(ev_halo);
    for (hipEvent_t e : xev) (void)hipEventDestroy(e);
    if (d_counts) (void)hipFree(d_counts);
    if (h_counts) (void)hipHostFree(h_counts);
    if (d_red) (void)hipFree(d_red);
    if (h_red) (void)hipHostFree(h_red);
    if (d_hdr) (void)hipFree(d_hdr);
    if (h_hdr) (void)hipHostFree(h_hdr);
  }
  // rx[chunk p] <- what rank p put into its chunk for this rank (an all-to-all with per-peer counts)
  void all_to_all(const sf_halo_layout& L, hipStream_t st)
  {
    RcclApi& a = rccl();
    SF_NCCL(a.GroupStart());
    for (int p = 0; p < world; p++) {
      if (L.send_cnt[p])
        SF_NCCL(a.Send(L.dev_tx + L.send_off[p], (size_t)L.send_cnt[p], ncclDouble, p, comm, st));
      if (L.recv_cnt[p])
        SF_NCCL(a.Recv(L.dev_rx + L.recv_off[p], (size_t)L.recv_cnt[p], ncclDouble, p, comm, st));
    }
    SF_NCCL(a.GroupEnd());
    link_delay(st);
  }
  static void link_delay(hipStream_t st)
  {
    static const int fake_us = getenv("SF_HALO_FAKE_DELAY_US") ? atoi(getenv("SF_HALO_FAKE_DELAY_US")) : 0;
    if (fake_us > 0) k_fake_link_delay<<<1, 64, 0, st>>>((long long)fake_us * 100);   // wall_clock64: 100 MHz
  }
};

static void halo_deleter(void* p) { delete static_cast<HaloComm*>(p); }

// ------------------------------------------------------------------------------------------------
// The slab driver in C++: rebuild-time exchanges ([3P] Comm::exchange / Comm::borders, the migration payload of
// fix_fluid_drag.cpp:211-243 + wall and pair history) and the lammps_step loop over RCCL, so that an MPI / C++ host
// (lammpsFoam) drives N GPUs through sf_slab_* alone.  Same protocol as sedifoam_amd/halo.py (which the gloo tests
// run against the oracle twin).
// ------------------------------------------------------------------------------------------------
constexpr int kBorderDoublesC = 14, kForwardDoublesC = 9;

static void slab_scratch(HaloComm& hc)
{
  if (hc.d_counts) return;
  SF_HIP(hipMalloc(&hc.d_counts, sizeof(long long) * 4));
  SF_HIP(hipHostMalloc(&hc.h_counts, sizeof(long long) * 4));
  SF_HIP(hipMalloc(&hc.d_red, sizeof(double) * 4));
  SF_HIP(hipHostMalloc(&hc.h_red, sizeof(double) * 4));
  SF_HIP(hipMalloc(&hc.d_hdr, sizeof(int) * 2 * hc.world));
  SF_HIP(hipHostMalloc(&hc.h_hdr, sizeof(int) * 2 * hc.world));
}

// one value reduced over the ranks (rebuild / setup time only: synchronises)
static double slab_allreduce(HaloComm& hc, hipStream_t st, double v, ncclRedOp_t op)
{
  if (hc.world == 1) return v;
  hc.h_red[0] = v;
  SF_HIP(hipMemcpyAsync(hc.d_red, hc.h_red, sizeof(double), hipMemcpyHostToDevice, st));
  SF_NCCL(rccl().AllReduce(hc.d_red, hc.d_red + 1, 1, ncclDouble, op, hc.comm, st));
  SF_HIP(hipMemcpyAsync(hc.h_red + 1, hc.d_red + 1, sizeof(double), hipMemcpyDeviceToHost, st));
  SF_HIP(hipStreamSynchronize(st));
  return hc.h_red[1];
}

// two values, each reduced with MAX (one collective)
static void slab_allreduce_max2(HaloComm& hc, hipStream_t st, double& a, double& b)
{
  if (hc.world == 1) return;
  hc.h_red[0] = a;
  hc.h_red[1] = b;
  SF_HIP(hipMemcpyAsync(hc.d_red, hc.h_red, 2 * sizeof(double), hipMemcpyHostToDevice, st));
  SF_NCCL(rccl().AllReduce(hc.d_red, hc.d_red + 2, 2, ncclDouble, ncclMax, hc.comm, st));
  SF_HIP(hipMemcpyAsync(hc.h_red + 2, hc.d_red + 2, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
  SF_HIP(hipStreamSynchronize(st));
  a = hc.h_red[2];
  b = hc.h_red[3];
}

// send[0][:n0] goes to the left neighbour, send[1][:n1] to the right one (doubles); returns what arrived from the
// left / right in hc.rx[0] / hc.rx[1] and their sizes.  known = receive sizes when both sides already know them.
// Message order with one peer on both sides (2 ranks, periodic): sends left then right, receives from-right then
// from-left -- what I sent leftwards reaches my left neighbour from its right.
static void pair_exchange(HaloComm& hc, hipStream_t st, int left, int right, const double* s0, long long n0,
                          const double* s1, long long n1, long long& m0, long long& m1);
static void slab_exchange(HaloComm& hc, hipStream_t st, const double* s0, long long n0, const double* s1, long long n1,
                          long long& m0, long long& m1)
{
  pair_exchange(hc, st, hc.left, hc.right, s0, n0, s1, n1, m0, m1);
}

// (left / right: the two neighbours along one dimension)
static void pair_exchange(HaloComm& hc, hipStream_t st, int left, int right, const double* s0, long long n0,
                          const double* s1, long long n1, long long& m0, long long& m1)
{
  RcclApi& a = rccl();
  if (left < 0) n0 = 0;
  if (right < 0) n1 = 0;
  hc.h_counts[0] = n0;
  hc.h_counts[1] = n1;
  hc.h_counts[2] = hc.h_counts[3] = 0;
  SF_HIP(hipMemcpyAsync(hc.d_counts, hc.h_counts, sizeof(long long) * 4, hipMemcpyHostToDevice, st));
  SF_NCCL(a.GroupStart());
  if (left >= 0) SF_NCCL(a.Send(hc.d_counts + 0, 1, ncclInt64, left, hc.comm, st));
  if (right >= 0) SF_NCCL(a.Send(hc.d_counts + 1, 1, ncclInt64, right, hc.comm, st));
  if (right >= 0) SF_NCCL(a.Recv(hc.d_counts + 3, 1, ncclInt64, right, hc.comm, st));   // leftward traffic comes from my right
  if (left >= 0) SF_NCCL(a.Recv(hc.d_counts + 2, 1, ncclInt64, left, hc.comm, st));
  SF_NCCL(a.GroupEnd());
  SF_HIP(hipMemcpyAsync(hc.h_counts, hc.d_counts, sizeof(long long) * 4, hipMemcpyDeviceToHost, st));
  SF_HIP(hipStreamSynchronize(st));
  m0 = hc.h_counts[2];
  m1 = hc.h_counts[3];
  double* r0 = hc.rx[0].need((size_t)m0 + 1);
  double* r1 = hc.rx[1].need((size_t)m1 + 1);
  SF_NCCL(a.GroupStart());
  if (left >= 0 && n0) SF_NCCL(a.Send(s0, (size_t)n0, ncclDouble, left, hc.comm, st));
  if (right >= 0 && n1) SF_NCCL(a.Send(s1, (size_t)n1, ncclDouble, right, hc.comm, st));
  if (right >= 0 && m1) SF_NCCL(a.Recv(r1, (size_t)m1, ncclDouble, right, hc.comm, st));
  if (left >= 0 && m0) SF_NCCL(a.Recv(r0, (size_t)m0, ncclDouble, left, hc.comm, st));
  SF_NCCL(a.GroupEnd());
}

// send / receive layout of the one-collective forward halo (valid until the next rebuild): per peer rank one header
// double (rebuild trigger) + the forward records for / from that peer
static void slab_layout(HaloComm& hc, hipStream_t st)
{
  const int W = hc.world, F = kForwardDoublesC;
  hc.send_cnt.assign(W, 1);
  hc.recv_cnt.assign(W, 1);
  hc.send_off.assign(W, 0);
  hc.recv_off.assign(W, 0);
  if (hc.left >= 0) {
    hc.send_cnt[hc.left] += hc.nsend[0] * F;
    hc.recv_cnt[hc.left] += hc.nrecv[0] * F;
  }
  if (hc.right >= 0) {
    hc.send_cnt[hc.right] += hc.nsend[1] * F;
    hc.recv_cnt[hc.right] += hc.nrecv[1] * F;
  }
  for (int p = 1; p < W; p++) {
    hc.send_off[p] = hc.send_off[p - 1] + hc.send_cnt[p - 1];
    hc.recv_off[p] = hc.recv_off[p - 1] + hc.recv_cnt[p - 1];
  }
  const long long ntx = hc.send_off[W - 1] + hc.send_cnt[W - 1], nrx = hc.recv_off[W - 1] + hc.recv_cnt[W - 1];
  const bool same = hc.left >= 0 && hc.left == hc.right;
  // records selected at a face without a neighbour (non-periodic box end) go to scratch behind the chunks
  long long scratch = 0;
  sf_halo_layout& L = hc.lay;
  L.world = W;
  L.shift_left = hc.shift_left;
  L.shift_right = hc.shift_right;
  if (hc.left >= 0) L.soff_l = hc.send_off[hc.left] + 1;
  else {
    L.soff_l = ntx + scratch;
    scratch += hc.nsend[0] * F;
  }
  if (hc.right >= 0) L.soff_r = hc.send_off[hc.right] + 1 + (same ? hc.nsend[0] * F : 0);
  else {
    L.soff_r = ntx + scratch;
    scratch += hc.nsend[1] * F;
  }
  L.roff_r = hc.right >= 0 ? hc.recv_off[hc.right] + 1 : 0;
  L.roff_l = hc.left >= 0 ? hc.recv_off[hc.left] + 1 + (same ? hc.nrecv[1] * F : 0) : 0;
  L.n_from_left = hc.nrecv[0];
  L.n_from_right = hc.nrecv[1];
  L.send_off = hc.send_off.data();
  L.send_cnt = hc.send_cnt.data();
  L.recv_off = hc.recv_off.data();
  L.recv_cnt = hc.recv_cnt.data();
  // (h_hdr is pinned and stays: the copy may read it after this function has returned; the stream is synchronised
  // long before the next layout overwrites it)
  for (int p = 0; p < W; p++) {
    hc.h_hdr[p] = (int)hc.send_off[p];
    hc.h_hdr[W + p] = (int)hc.recv_off[p];
  }
  SF_HIP(hipMemcpyAsync(hc.d_hdr, hc.h_hdr, sizeof(int) * 2 * W, hipMemcpyHostToDevice, st));
  L.dev_shdr = hc.d_hdr;
  L.dev_rhdr = hc.d_hdr + W;
  L.dev_tx = hc.a2a_tx.need((size_t)(ntx + scratch) + 1);
  L.dev_rx = hc.a2a_rx.need((size_t)nrx + 1);
  hc.lay_valid = true;
}

// (after slab_layout: the sub-step kernels write the forward records of the border atoms straight into the chunks)
static void slab_fused_pack(SfLammps& S, HaloComm& hc)
{
  const sf_halo_layout& L = hc.lay;
  S.eng.set_forward_tx(L.dev_tx + L.soff_l, L.shift_left, L.dev_tx + L.soff_r, L.shift_right, L.dev_tx, L.dev_shdr,
                       L.world);
}

static void slab_rebuild(SfLammps& S, HaloComm& hc)
{
  Range r("neighbor rebuild");
  const auto t_begin = std::chrono::steady_clock::now();
  DemEngine& e = S.eng;
  hipStream_t st = e.stream();
  static const bool dbgt = getenv("SF_DEBUG_REBUILD_PHASES") != nullptr;
  auto tprev = t_begin;
  auto lap = [&](const char* what) {
    if (!dbgt) return;
    (void)hipStreamSynchronize(st);
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[rebuild] %-28s %8.1f us\n", what, std::chrono::duration<double, std::micro>(t - tprev).count());
    tprev = t;
  };
  e.rebuild_begin();
  lap("rebuild_begin (partner tags)");
  // one all-reduce (two values, MAX each) carries the history slots a migrating atom needs -- max_neigh_used over
  // ALL ranks, whoever migrates -- and whether any rank has an atom outside its slab: the usual rebuild migrates
  // nothing and skips that round
  const long long crossed = e.migrate_count();
  double slots = (double)e.max_neigh_used(), any_crossed = crossed ? 1.0 : 0.0;
  slab_allreduce_max2(hc, st, slots, any_crossed);
  lap("allreduce");
  e.migrate_set_slots((int)slots);
  if (any_crossed != 0.0) {
    const int rec = e.migrate_record_doubles();
    const size_t cap = (size_t)(crossed + 1) * rec;
    double* b0 = hc.mig[0].need(cap);
    double* b1 = hc.mig[1].need(cap);
    const long long n0 = e.migrate_pack(0, hc.shift_left, b0, (long long)cap);
    const long long n1 = e.migrate_pack(1, hc.shift_right, b1, (long long)cap);
    if ((hc.left < 0 && n0) || (hc.right < 0 && n1)) fail("Lost atoms: an atom left the non-periodic box in x");
    long long m0 = 0, m1 = 0;
    slab_exchange(hc, st, b0, n0, b1, n1, m0, m1);
    e.migrate_unpack(hc.rx[0].p, m0);
    e.migrate_unpack(hc.rx[1].p, m1);
  }
  lap("migration");
  e.rebuild_sort();
  lap("rebuild_sort");
  const size_t bcap = (size_t)e.nlocal() + 1;
  double* s0 = hc.bor[0].need(bcap * kBorderDoublesC);
  double* s1 = hc.bor[1].need(bcap * kBorderDoublesC);
  long long a0 = 0, a1 = 0;
  e.border_pack_both(hc.shift_left, s0, hc.shift_right, s1, (long long)bcap, &a0, &a1);
  hc.nsend[0] = a0;
  hc.nsend[1] = a1;
  long long m0 = 0, m1 = 0;
  lap("border_pack x2");
  slab_exchange(hc, st, s0, a0 * kBorderDoublesC, s1, a1 * kBorderDoublesC, m0, m1);
  lap("border exchange");
  hc.nrecv[0] = m0 / kBorderDoublesC;
  hc.nrecv[1] = m1 / kBorderDoublesC;
  e.border_unpack(0, hc.rx[0].p, hc.nrecv[0]);
  e.border_unpack(1, hc.rx[1].p, hc.nrecv[1]);
  lap("border_unpack x2");
  e.rebuild_finish();
  lap("rebuild_finish (ghosts, list)");
  hc.n_rebuilds++;
  slab_layout(hc, st);
  slab_fused_pack(S, hc);
  lap("layout + send slots");
  hc.rebuild_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
}

static int slab_halo_run(SfLammps& S, HaloComm& hc, int first_k, int end_k, int n);

static void slab_step(SfLammps& S, HaloComm& hc, int n)
{
  Range r("lammps");
  DemEngine& e = S.eng;
  e.run_begin();
  hc.pre_exchanged = false;
  int k = 0;
  while (k < n) {
    // (the overlapped schedule opens every call with an exchange of its own: it queues the whole run as before)
    const int end = e.overlap() ? n : k + hc.predict.chunk(e.nsteps(), n - k);
    const int trig = slab_halo_run(S, hc, k, end, n);
    if (trig >= end) {
      k = end;
      continue;
    }
    k = trig + 1;   // sub-steps k..trig ran (trig = -1: the list was stale for sub-step 0)
    hc.pre_exchanged = false;
    {
      DemEngine::InRunGuard guard(e);
      slab_rebuild(S, hc);
    }
    hc.predict.rebuilt(e.nsteps());
    if (e.overlap()) e.overlap_begin();
  }
}

// queue sub-steps first_k .. end_k - 1 of a run of n (per sub-step: one grouped ncclSend/ncclRecv, fused unpack, the
// kernel that also writes the next exchange's records; overlapped if sf_dem_set_overlap is on), synchronise once,
// return the voted rebuild trigger
static int halo_run_layout(SfLammps& S, HaloComm& hcr, int first_k, int end_k, int n, const sf_halo_layout& layr)
{
  HaloComm* hc = &hcr;
  const sf_halo_layout* lay = &layr;
  DemEngine& e = S.eng;
  hipStream_t main = e.stream();
  int trigger = 0;
  // (the slab driver's own layout: the sub-step kernel that integrated the border atoms has written their forward
  // records and, if one of its atoms moved beyond skin/2, the vote headers -- DemEngine::set_forward_tx)
  auto exchange = [&](int kstep, hipStream_t st) {
    if (!e.forward_tx_written())
      e.forward_pack_fused(lay->shift_left, lay->soff_l, lay->shift_right, lay->soff_r, lay->dev_shdr, lay->world,
                           lay->dev_tx);
    hc->all_to_all(*lay, st);
    e.forward_unpack_fused(lay->dev_rx, lay->roff_l, lay->n_from_left, lay->roff_r, lay->n_from_right, lay->dev_rhdr,
                           lay->world, kstep);
  };
  const int launched = end_k - first_k;
  if (!e.overlap()) {
    for (int s = first_k; s < end_k; s++) {
      if (s == first_k && hc->pre_exchanged) {   // (ghosts and votes in front of this sub-step are in place)
        hc->pre_exchanged = false;
        e.substep_k(s == n - 1, s);
        continue;
      }
      const bool timed = e.profiling() && s % 8 == 4 && s > first_k;   // (not the sub-steps the kernel profile times)
      if (timed) {
        if (hc->xev_used + 2 > hc->xev.size())
          for (int k = 0; k < 2; k++) {
            hipEvent_t ev;
            SF_HIP(hipEventCreate(&ev));
            hc->xev.push_back(ev);
          }
        SF_HIP(hipEventRecord(hc->xev[hc->xev_used], main));
      }
      exchange(-1, main);
      if (timed) {
        SF_HIP(hipEventRecord(hc->xev[hc->xev_used + 1], main));
        hc->xev_used += 2;
      }
      e.substep_k(s == n - 1, s);
    }
    // A piece that stops before the end of the run closes with the exchange that belongs in front of the next
    // sub-step: the ranks learn of a trigger in the LAST sub-step of the piece only through its votes, and every rank
    // must take the same decision here (rebuild or go on) -- found by the 2-rank coupled test over the stand-in wire.
    if (end_k < n) {
      exchange(-1, main);
      hc->pre_exchanged = true;
    }
    trigger = e.batch_end(first_k, launched);
    hc->harvest_exchange_profile();   // (batch_end has synchronised)
  } else {
    hipStream_t cs = e.comm_stream();
    SF_HIP(hipEventRecord(hc->ev_boundary, main));
    SF_HIP(hipStreamWaitEvent(cs, hc->ev_boundary, 0));
    exchange(first_k - 1, cs);                       // ghosts + vote before sub-step first_k
    SF_HIP(hipEventRecord(hc->ev_halo, cs));
    for (int s = first_k; s < end_k; s++) {
      const bool last = s == n - 1;
      SF_HIP(hipStreamWaitEvent(main, hc->ev_halo, 0));
      e.substep_part(2, last, s);                    // boundary atoms: need the ghosts of exchange s-1
      SF_HIP(hipEventRecord(hc->ev_boundary, main));
      e.substep_part(1, last, s);                    // interior atoms, under the exchange of sub-step s
      e.substep_flip(s);
      SF_HIP(hipStreamWaitEvent(cs, hc->ev_boundary, 0));
      exchange(s, cs);
      SF_HIP(hipEventRecord(hc->ev_halo, cs));
    }
    SF_HIP(hipStreamWaitEvent(main, hc->ev_halo, 0));
    trigger = e.overlap_batch_end(first_k, launched, end_k - 1);
  }
  return trigger;
}

static int slab_halo_run(SfLammps& S, HaloComm& hc, int first_k, int end_k, int n)
{
  if (!hc.lay_valid) fail("sf_slab_step: no halo layout (rebuild first)");
  return halo_run_layout(S, hc, first_k, end_k, n, hc.lay);
}

// ------------------------------------------------------------------------------------------------
// The brick driver: [3P] Comm on a 3-D processor grid (LAMMPS' `processors Px Py Pz`; the reference's parallel cases
// decompose in two dimensions, cases/example-cases/transport-bedload/system/decomposeParDict: n (14 1 6)).
//   * migration at a rebuild: staged, dimension by dimension, like Comm::exchange (an atom that left through an edge
//     is forwarded by the rank it reaches first);
//   * ghosts: every owned atom within the ghost cutoff of a face / edge / corner goes STRAIGHT to the brick behind
//     it (up to 26 directions), so the forward halo of a sub-step is ONE grouped ncclSend/ncclRecv with the
//     neighbours -- LAMMPS' three staged swaps would be three exchange latencies per sub-step;
//   * the rebuild vote: a header word in every chunk, to every rank (a rank that is not a neighbour gets the header
//     alone): MIN over what a rank receives = Neighbor::decide's MPI_Allreduce, exact on any grid.
// Dimensions the grid does not cut (P_d = 1) keep their periodic images local ((root, image code) neighbour words).
// ------------------------------------------------------------------------------------------------
static int brick_rank(const BrickState& B, const int c[3]) { return c[0] + B.P[0] * (c[1] + B.P[1] * c[2]); }

// the brick at offset d from coordinates c: its rank (-1: beyond a non-periodic box face) and the shift that takes a
// position of this brick's frame into that brick's
static int brick_neighbour(const BrickState& B, const int c[3], const int d[3], double shift[3])
{
  int n[3];
  for (int k = 0; k < 3; k++) {
    shift[k] = 0.0;
    n[k] = c[k] + d[k];
    if (n[k] < 0 || n[k] >= B.P[k]) {
      if (!B.periodic[k]) return -1;
      shift[k] = n[k] < 0 ? (B.hi[k] - B.lo[k]) : -(B.hi[k] - B.lo[k]);
      n[k] = (n[k] + B.P[k]) % B.P[k];
    }
  }
  return brick_rank(B, n);
}

static void brick_directions(BrickState& B)
{
  B.sdirs.clear();
  B.rdirs.clear();
  for (int dz = -1; dz <= 1; dz++)
    for (int dy = -1; dy <= 1; dy++)
      for (int dx = -1; dx <= 1; dx++) {
        const int d[3] = {dx, dy, dz};
        if (!dx && !dy && !dz) continue;
        bool ok = true;
        for (int k = 0; k < 3; k++) ok = ok && (B.ext[k] || d[k] == 0);
        if (!ok) continue;
        BrickState::SDir sd;
        sd.peer = brick_neighbour(B, B.c, d, sd.shift);
        if (sd.peer < 0) continue;
        for (int k = 0; k < 3; k++) sd.d[k] = d[k];
        sd.code = (dx + 1) + 3 * (dy + 1) + 9 * (dz + 1);
        B.sdirs.push_back(sd);
        // the neighbour at offset d also SENDS to this rank, in its direction -d
        BrickState::RDir rd;
        rd.peer = sd.peer;
        rd.sender_code = (-dx + 1) + 3 * (-dy + 1) + 9 * (-dz + 1);
        B.rdirs.push_back(rd);
      }
  std::sort(B.sdirs.begin(), B.sdirs.end(), [](const BrickState::SDir& a, const BrickState::SDir& b) {
    return a.peer != b.peer ? a.peer < b.peer : a.code < b.code;
  });
  std::sort(B.rdirs.begin(), B.rdirs.end(), [](const BrickState::RDir& a, const BrickState::RDir& b) {
    return a.peer != b.peer ? a.peer < b.peer : a.sender_code < b.sender_code;
  });
  for (int k = 0; k < 3; k++)
    for (int side = 0; side < 2; side++) {
      int d[3] = {0, 0, 0};
      d[k] = side ? 1 : -1;
      double sh[3];
      B.nbr[k][side] = B.ext[k] ? brick_neighbour(B, B.c, d, sh) : -1;
      B.mshift[k][side] = sh[k];
    }
}

static void brick_topology(HaloComm& hc, DemEngine& e)
{
  BrickState& B = *hc.brick;
  brick_directions(B);
  std::vector<int> d3;
  std::vector<double> s3;
  for (const auto& sd : B.sdirs)
    for (int k = 0; k < 3; k++) {
      d3.push_back(sd.d[k]);
      s3.push_back(sd.shift[k]);
    }
  e.brick_set_dirs((int)B.sdirs.size(), d3.data(), s3.data());
  B.nsend.assign(B.sdirs.size(), 0);
  B.nrecv.assign(B.rdirs.size(), 0);
  if (!B.d_cnt) {
    SF_HIP(hipMalloc(&B.d_cnt, sizeof(long long) * 2 * DemEngine::kMaxDirs));
    SF_HIP(hipHostMalloc(&B.h_cnt, sizeof(long long) * 2 * DemEngine::kMaxDirs));
  }
}

// one grouped exchange of per-peer segments: segment p of `send` (sizes scnt, contiguous in peer order) to rank p,
// segment p of `recv` from rank p
template <class T>
static void brick_segments(HaloComm& hc, hipStream_t st, const T* send, const std::vector<long long>& soff,
                           const std::vector<long long>& scnt, T* recv, const std::vector<long long>& roff,
                           const std::vector<long long>& rcnt, ncclDataType_t ty)
{
  RcclApi& a = rccl();
  SF_NCCL(a.GroupStart());
  for (int p = 0; p < hc.world; p++) {
    if (scnt[p]) SF_NCCL(a.Send(send + soff[p], (size_t)scnt[p], ty, p, hc.comm, st));
    if (rcnt[p]) SF_NCCL(a.Recv(recv + roff[p], (size_t)rcnt[p], ty, p, hc.comm, st));
  }
  SF_NCCL(a.GroupEnd());
}

// ------------------------------------------------------------------------------------------------
// direct ghost writes: bring-up, rebuild-time handle exchange
// ------------------------------------------------------------------------------------------------
static void* alloc_fine_grained(size_t bytes)
{
  void* p = nullptr;
  SF_HIP(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained));
  SF_HIP(hipMemset(p, 0, bytes));
  return p;
}

static DemEngine::DirectSync direct_sync(const HaloComm& hc, int seq, int par)
{
  const DirectHalo& D = *hc.direct;
  DemEngine::DirectSync s;
  s.world = hc.world;
  s.rank = hc.rank;
  s.seq = seq;
  s.par = par;
  s.my_sync = D.my_sync;
  for (int p = 0; p < 32; p++) s.peer_sync[p] = p < hc.world ? D.peer_sync[p] : nullptr;
  static const double secs = getenv("SF_HALO_DIRECT_TIMEOUT") ? atof(getenv("SF_HALO_DIRECT_TIMEOUT")) : 20.0;
  s.max_ticks = (long long)(secs * 1.0e8);   // wall_clock64: 100 MHz
  return s;
}

// every rank's message for every other rank (kMsg int64 each; cnt[p] = 0: nothing for / from rank p) over RCCL
static void direct_messages(HaloComm& hc, hipStream_t st, const std::vector<int>& with)
{
  DirectHalo& D = *hc.direct;
  const int W = hc.world, K = DirectHalo::kMsg;
  std::vector<long long> off(W), cnt(W);
  for (int p = 0; p < W; p++) {
    off[p] = (long long)p * K;
    cnt[p] = with[p] ? K : 0;
  }
  SF_HIP(hipMemcpyAsync(D.d_msg, D.h_msg, sizeof(long long) * W * K, hipMemcpyHostToDevice, st));
  brick_segments<long long>(hc, st, D.d_msg, off, cnt, D.d_msg + (size_t)W * K, off, cnt, ncclInt64);
  SF_HIP(hipMemcpyAsync(D.h_msg + (size_t)W * K, D.d_msg + (size_t)W * K, sizeof(long long) * W * K, hipMemcpyDeviceToHost, st));
  SF_HIP(hipStreamSynchronize(st));
}

// SF_HALO_DIRECT: unset / 0 = the RCCL exchange; "auto" = try, and fall back to the RCCL exchange when anything in the
// bring-up fails on any rank; 1 = must come up (an error otherwise).  Bring-up: the flag / vote areas of all ranks are
// mapped into each other and a first round of {vote, flag} goes through them before a single ghost depends on it.
// (bench.py --gpus N asks for "auto" and proves the decomposed result against a single-domain run before it times
// anything; a library default of "auto" would put an unproven transport under every host.)
static void direct_init(SfLammps& S, HaloComm& hc)
{
  const char* env = getenv("SF_HALO_DIRECT");
  // ("2" / "slots": ghost slots, must come up; "auto2": ghost slots, or the RCCL exchange when the bring-up fails)
  const bool slots = env && (!strcmp(env, "2") || !strcmp(env, "slots") || !strcmp(env, "auto2"));
  const int want = !env ? 0 : (!strcmp(env, "auto") || !strcmp(env, "auto2") ? -1 : (slots ? 1 : atoi(env)));
  bool slots_effective = slots;
  const bool self_only = hc.world == 1 && hc.brick && (hc.brick->ext[0] || hc.brick->ext[1] || hc.brick->ext[2]);
  if (want == 0 || (hc.world < 2 && !self_only)) return;
  if (hc.world > 32) {
    if (want == 1) fail("SF_HALO_DIRECT=1: %d ranks (at most 32)", hc.world);
    return;
  }
  DemEngine& e = S.eng;
  hipStream_t st = e.stream();
  const int W = hc.world, K = DirectHalo::kMsg;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle = 8 int64");
  hc.direct = new DirectHalo();
  DirectHalo& D = *hc.direct;
  double ok = 1.0;
  std::string why;
  D.peer_sync.assign(W, nullptr);
  D.peers.assign(W, DirectHalo::Peer());
  SF_HIP(hipMalloc(&D.d_msg, sizeof(long long) * 2 * W * K));
  SF_HIP(hipHostMalloc(&D.h_msg, sizeof(long long) * 2 * W * K));
  memset(D.h_msg, 0, sizeof(long long) * 2 * W * K);
  try {
    D.my_sync = static_cast<int*>(alloc_fine_grained(sizeof(int) * DemEngine::kSyncStride * 32));
    hipIpcMemHandle_t h;
    SF_HIP(hipIpcGetMemHandle(&h, D.my_sync));
    // which device this rank runs on (PCI bus id, hashed): ranks that SHARE a device cannot use the ghost slots -- the gates
    // of one rank's kernel spin in the wave slots the other rank's kernel needs to finish -- and every rank must know
    int dev = 0;
    char bus[64] = {0};
    long long devid = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetPCIBusId(bus, (int)sizeof(bus) - 1, dev) == hipSuccess)
      for (const char* c = bus; *c; c++) devid = devid * 131 + (unsigned char)*c;
    devid = (devid & 0x7fffffffffffLL) + 1;   // (> 0: a word a rank did not fill reads as "unknown")
    for (int p = 0; p < W; p++) {
      memcpy(D.h_msg + (size_t)p * K + 1, &h, 64);
      D.h_msg[(size_t)p * K + 10] = devid;
    }
    D.h_msg[(size_t)(W + hc.rank) * K + 10] = devid;
  } catch (const std::exception& ex) {
    ok = 0.0;
    why = ex.what();
  }
  // (collective from here on: every rank takes part whatever happened to it)
  std::vector<int> all(W, 1);
  all[hc.rank] = 0;
  direct_messages(hc, st, all);
  if (ok != 0.0) {
    try {
      for (int p = 0; p < W; p++) {
        if (p == hc.rank) {
          D.peer_sync[p] = D.my_sync;
          continue;
        }
        hipIpcMemHandle_t h;
        memcpy(&h, D.h_msg + (size_t)(W + p) * K + 1, 64);
        void* m = nullptr;
        SF_HIP(hipIpcOpenMemHandle(&m, h, hipIpcMemLazyEnablePeerAccess));
        D.peer_sync[p] = static_cast<int*>(m);
      }
    } catch (const std::exception& ex) {
      ok = 0.0;
      why = ex.what();
    }
  }
  ok = slab_allreduce(hc, st, ok, ncclMin);
  // two ranks on one device (every rank holds every rank's word: the same answer everywhere)
  bool shared_device = false;
  for (int p = 0; p < W && ok != 0.0; p++)
    for (int q = p + 1; q < W; q++) {
      const long long a = D.h_msg[(size_t)(W + p) * K + 10], b = D.h_msg[(size_t)(W + q) * K + 10];
      if (a > 0 && a == b) shared_device = true;
    }
  if (slots && shared_device && !self_only && !(getenv("SF_HALO_SHARED_DEVICE_OK") && atoi(getenv("SF_HALO_SHARED_DEVICE_OK")))) {
    if (strcmp(env, "auto2") != 0) {
      delete hc.direct;
      hc.direct = nullptr;
      fail("SF_HALO_DIRECT=%s: two ranks share one GPU -- the ghost slots' in-kernel hand-off cannot make progress there "
           "(use auto2, auto or 1; SF_HALO_SHARED_DEVICE_OK=1 overrides for small test beds)", env);
    }
    slots_effective = false;   // auto2: the receive areas (SF_HALO_DIRECT=auto), which ranks sharing a device can run
    if (getenv("SF_DEBUG_HALO")) fprintf(stderr, "[sedifoam_amd] rank %d: ranks share a GPU, ghost slots -> receive areas\n", hc.rank);
  }
  if (ok != 0.0) {
    // the first round: vote + flag to every rank, every rank's flag awaited (no records yet)
    DemEngine::BrickBlocks none{};
    DemEngine::DirectSync s = direct_sync(hc, 1, 0);
    // (the wait of SF_HALO_DIRECT_TIMEOUT -- direct_sync -- capped at 5 s unless the variable asks for more: ranks that share
    // one GPU time-slice it and need longer)
    if (!getenv("SF_HALO_DIRECT_TIMEOUT") && s.max_ticks > 500000000LL) s.max_ticks = 500000000LL;
    ok = e.brick_direct_probe(none, s) ? 1.0 : 0.0;
    if (ok == 0.0) why = "the first flag round timed out";
    ok = slab_allreduce(hc, st, ok, ncclMin);
    D.xseq = 1;
  }
  if (ok == 0.0) {
    delete hc.direct;
    hc.direct = nullptr;
    if (want == 1) fail("SF_HALO_DIRECT=1: direct ghost writes did not come up on every rank (%s)", why.empty() ? "another rank" : why.c_str());
    if (getenv("SF_DEBUG_HALO")) fprintf(stderr, "[sedifoam_amd] rank %d: direct ghost writes off, RCCL exchange (%s)\n", hc.rank, why.empty() ? "another rank failed" : why.c_str());
    return;
  }
  D.on = true;
  D.must = want == 1;
  D.mode = slots_effective ? 2 : 1;
  if (D.mode == 2) {
    GsSync y;
    memset(&y, 0, sizeof(y));
    y.world = W;
    y.rank = hc.rank;
    y.max_ticks = direct_sync(hc, 0, 0).max_ticks;
    y.my_sync = D.my_sync;
    for (int p = 0; p < W; p++) y.peer_sync[p] = D.peer_sync[p];
    // (ghost slots keep ONE 64-bit word per line, (flag << 32) | vote: wipe what the first round left there -- every rank its
    // own area, nobody writes between the two collectives -- and number the launches from 2)
    SF_HIP(hipStreamSynchronize(st));
    SF_HIP(hipMemset(D.my_sync, 0, sizeof(int) * DemEngine::kSyncStride * 32));
    (void)slab_allreduce(hc, st, 1.0, ncclMin);
    e.gs_configure(y, 2);
  }
  if (getenv("SF_DEBUG_HALO"))
    fprintf(stderr, "[sedifoam_amd] rank %d: direct ghost writes on (%d ranks, %s)\n", hc.rank, W,
            D.mode == 2 ? "ghost slots" : "receive areas");
}

// after a rebuild fixed the chunk layout: the receive areas (grown if need be), their handles and this rank's chunk
// offsets to every neighbour, the neighbours' to this rank; fills the [2][kMaxDirs] table of block starts.
// Collective, and it FAILS collectively: whatever goes wrong on one rank (an allocation, an IPC handle a neighbour's
// process cannot open) reaches every rank through an all-reduce -- the ranks then leave the direct transport together
// (returns false: the caller goes on over RCCL) or, when SF_HALO_DIRECT demanded it, fail together with the reason,
// instead of one rank throwing while the others spin on flags that will never come.
static bool direct_lost(SfLammps& S, HaloComm& hc, const std::string& why)
{
  DirectHalo& D = *hc.direct;
  D.on = false;
  S.eng.gs_off();
  if (D.must)
    fail("SF_HALO_DIRECT: the direct transport was lost at a rebuild (%s)", why.empty() ? "on another rank" : why.c_str());
  if (getenv("SF_DEBUG_HALO"))
    fprintf(stderr, "[sedifoam_amd] rank %d: direct ghost writes off from this rebuild on, RCCL exchange (%s)\n", hc.rank,
            why.empty() ? "another rank failed" : why.c_str());
  return false;
}

static bool direct_rebuild(SfLammps& S, HaloComm& hc, double** blk2)
{
  DirectHalo& D = *hc.direct;
  BrickState& B = *hc.brick;
  DemEngine& e = S.eng;
  hipStream_t st = e.stream();
  const int W = hc.world, K = DirectHalo::kMsg;
  double ok = 1.0;
  std::string why;
  // areas a neighbour may still have mapped are freed only after every rank has seen the new generation (below)
  double* retired[2] = {nullptr, nullptr};
  hipIpcMemHandle_t h0, h1;
  memset(&h0, 0, sizeof(h0));
  memset(&h1, 0, sizeof(h1));
  try {
    const size_t need = (size_t)(hc.recv_off[W - 1] + hc.recv_cnt[W - 1]) + 1;
    if (need > D.rx_cap) {
      SF_HIP(hipStreamSynchronize(st));
      for (int k = 0; k < 2; k++) {
        retired[k] = D.rx[k];
        D.rx[k] = nullptr;
      }
      D.rx_cap = need + need / 2 + 4096;
      D.rx_gen++;
      for (double*& b : D.rx) b = static_cast<double*>(alloc_fine_grained(sizeof(double) * D.rx_cap));
    }
    SF_HIP(hipIpcGetMemHandle(&h0, D.rx[0]));
    SF_HIP(hipIpcGetMemHandle(&h1, D.rx[1]));
  } catch (const std::exception& ex) {
    ok = 0.0;
    why = ex.what();
  }
  std::vector<int> nbr(W, 0);
  for (const auto& sd : B.sdirs) nbr[sd.peer] = 1;
  for (int p = 0; p < W; p++) {
    long long* m = D.h_msg + (size_t)p * K;
    m[0] = D.rx_gen;
    memcpy(m + 1, &h0, 64);
    memcpy(m + 9, &h1, 64);
    m[17] = hc.recv_off[p];   // where rank p's chunk starts in this rank's areas
  }
  {
    std::vector<int> with = nbr;
    with[hc.rank] = 0;
    direct_messages(hc, st, with);   // (every rank takes part, whatever happened to it above)
  }
  ok = slab_allreduce(hc, st, ok, ncclMin);   // (a handle of a failed rank must not be opened)
  if (ok != 0.0) {
    try {
      for (int p = 0; p < W; p++) {
        if (!nbr[p]) continue;
        const long long* m = D.h_msg + (size_t)(W + p) * K;
        DirectHalo::Peer& P = D.peers[p];
        if (p == hc.rank) {   // (SF_HALO_SELF_COMM: this rank's own areas, no handle)
          P.map[0] = D.rx[0];
          P.map[1] = D.rx[1];
          P.remote_off = hc.recv_off[p];
          continue;
        }
        if (m[0] != P.gen_seen) {
          for (void*& mm : P.map) {
            if (mm) SF_HIP(hipIpcCloseMemHandle(mm));
            mm = nullptr;
          }
          hipIpcMemHandle_t a, b;
          memcpy(&a, m + 1, 64);
          memcpy(&b, m + 9, 64);
          SF_HIP(hipIpcOpenMemHandle(&P.map[0], a, hipIpcMemLazyEnablePeerAccess));
          SF_HIP(hipIpcOpenMemHandle(&P.map[1], b, hipIpcMemLazyEnablePeerAccess));
          P.gen_seen = m[0];
        }
        P.remote_off = m[17];
      }
    } catch (const std::exception& ex) {
      ok = 0.0;
      why = ex.what();
    }
    ok = slab_allreduce(hc, st, ok, ncclMin);
  }
  // (every rank has closed its mappings of the old generation, or never will use them again: the old areas can go)
  for (double* b : retired)
    if (b) (void)hipFree(b);
  if (ok == 0.0) return direct_lost(S, hc, why);
  // block q of this rank's chunk for peer p sits where it sits in the local send buffer, relative to the chunk's start
  for (int par = 0; par < 2; par++)
    for (int q = 0; q < DemEngine::kMaxDirs; q++) {
      blk2[par * DemEngine::kMaxDirs + q] = nullptr;
      if (q >= (int)B.sdirs.size()) continue;
      const int p = B.sdirs[q].peer;
      blk2[par * DemEngine::kMaxDirs + q] =
          static_cast<double*>(D.peers[p].map[par]) + D.peers[p].remote_off + (B.snd.off[q] - hc.send_off[p]);
    }
  return true;
}

// Ghost slots, after a rebuild (or when the buffer parity moved against the launch numbers): every rank tells its
// neighbours the IPC handles of the record arrays its launches of even / odd number will read, and where their ghosts start
// in them; fills blk6[(par * 3 + a) * kMaxDirs + q] -- where block q's first x | v | omega record goes.  The arrays are
// the engine's own (a re-sort swaps them with its scratch arrays, a capacity growth replaces them): mappings are cached
// by handle, an allocation is opened once per neighbour.  Collective, fails collectively like direct_rebuild.
static bool gs_rebuild(SfLammps& S, HaloComm& hc, double4** blk6)
{
  DirectHalo& D = *hc.direct;
  BrickState& B = *hc.brick;
  DemEngine& e = S.eng;
  hipStream_t st = e.stream();
  const int W = hc.world, K = DirectHalo::kMsg;
  static_assert(DirectHalo::kMsg >= 50, "six handles + the first ghost index");
  double ok = 1.0;
  std::string why;
  void* mine[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipIpcMemHandle_t h[6];
  memset(h, 0, sizeof(h));
  try {
    if (!e.brick_fused_pack_possible())
      fail("ghost slots need the sub-step kernel to write the border records itself: not in a brick thinner than twice the "
           "ghost cutoff, not with SF_HALO_FUSED_PACK=0");
    for (int par = 0; par < 2; par++) e.gs_input_arrays(par, &mine[3 * par], &mine[3 * par + 1], &mine[3 * par + 2]);
    if (W > 1)
      for (int k = 0; k < 6; k++) SF_HIP(hipIpcGetMemHandle(&h[k], mine[k]));
  } catch (const std::exception& ex) {
    ok = 0.0;
    why = ex.what();
  }
  // where the ghosts of rank p start in this rank's arrays: behind the owned atoms, its blocks contiguous in its send order
  std::vector<long long> gfirst(W, 0);
  for (int q = (int)B.rdirs.size() - 1; q >= 0; q--) gfirst[B.rdirs[q].peer] = (long long)e.nlocal() + B.rcv.first[q];
  std::vector<int> nbr(W, 0);
  for (const auto& sd : B.sdirs) nbr[sd.peer] = 1;
  for (int p = 0; p < W; p++) {
    long long* m = D.h_msg + (size_t)p * K;
    memcpy(m, h, sizeof(h));   // 48 int64
    m[48] = gfirst[p];
  }
  {
    std::vector<int> with = nbr;
    with[hc.rank] = 0;
    direct_messages(hc, st, with);
  }
  ok = slab_allreduce(hc, st, ok, ncclMin);
  std::vector<std::array<void*, 6>> base(W);
  std::vector<long long> first(W, 0);
  if (ok != 0.0) {
    try {
      for (int p = 0; p < W; p++) {
        if (!nbr[p]) continue;
        if (p == hc.rank) {   // (SF_HALO_SELF_COMM: this rank's own arrays, no handle)
          for (int k = 0; k < 6; k++) base[p][k] = mine[k];
          first[p] = gfirst[p];
          continue;
        }
        const long long* m = D.h_msg + (size_t)(W + p) * K;
        for (int k = 0; k < 6; k++) {
          std::array<char, 64> key;
          memcpy(key.data(), m + 8 * k, 64);
          auto it = D.peers[p].opened.find(key);
          if (it == D.peers[p].opened.end()) {
            hipIpcMemHandle_t hh;
            memcpy(&hh, key.data(), 64);
            void* mm = nullptr;
            SF_HIP(hipIpcOpenMemHandle(&mm, hh, hipIpcMemLazyEnablePeerAccess));
            it = D.peers[p].opened.emplace(key, mm).first;
          }
          base[p][k] = it->second;
        }
        first[p] = m[48];
      }
    } catch (const std::exception& ex) {
      ok = 0.0;
      why = ex.what();
    }
    ok = slab_allreduce(hc, st, ok, ncclMin);
  }
  if (ok == 0.0) return direct_lost(S, hc, why);
  // a neighbour's arrays grow and are swapped with its scratch arrays: a mapping none of its six current handles refers
  // to is closed here (every rank is past the second all-reduce: nobody uses it any more) -- an open mapping keeps the
  // owner's hipFree from releasing the memory, and a long run with particle injection would collect them
  for (int p = 0; p < W; p++) {
    if (!nbr[p] || p == hc.rank) continue;
    const long long* m = D.h_msg + (size_t)(W + p) * K;
    auto& opened = D.peers[p].opened;
    for (auto it = opened.begin(); it != opened.end();) {
      bool used = false;
      for (int k = 0; k < 6 && !used; k++) used = memcmp(it->first.data(), m + 8 * k, 64) == 0;
      if (used) {
        ++it;
      } else {
        (void)hipIpcCloseMemHandle(it->second);
        it = opened.erase(it);
      }
    }
  }
  std::vector<long long> sent(W, 0);
  for (int q = 0; q < DemEngine::kMaxDirs; q++) {
    for (int k = 0; k < 6; k++) blk6[k * DemEngine::kMaxDirs + q] = nullptr;
    if (q >= (int)B.sdirs.size()) continue;
    const int p = B.sdirs[q].peer;
    for (int k = 0; k < 6; k++)
      blk6[k * DemEngine::kMaxDirs + q] = static_cast<double4*>(base[p][k]) + first[p] + sent[p];
    sent[p] += B.nsend[q];
  }
  return true;
}

static void brick_rebuild(SfLammps& S, HaloComm& hc)
{
  Range r("neighbor rebuild");
  const auto t_begin = std::chrono::steady_clock::now();
  BrickState& B = *hc.brick;
  DemEngine& e = S.eng;
  hipStream_t st = e.stream();
  const int W = hc.world;
  e.rebuild_begin();
  // ---- migration, staged over the dimensions ([3P] Comm::exchange) ----
  const long long crossed = e.migrate_count3();
  double slots = (double)e.max_neigh_used(), any_crossed = crossed ? 1.0 : 0.0;
  slab_allreduce_max2(hc, st, slots, any_crossed);
  e.migrate_set_slots((int)slots);
  if (any_crossed != 0.0) {
    const int rec = e.migrate_record_doubles();
    for (int k = 0; k < 3; k++) {
      if (!B.ext[k]) continue;
      // (atoms handed on by an earlier stage may leave through this one: the buffers follow the owned count)
      const size_t cap = (size_t)(e.nlocal() + 1) * rec;
      double* b0 = hc.mig[0].need(cap);
      double* b1 = hc.mig[1].need(cap);
      const long long n0 = e.migrate_pack_dim(k, 0, B.mshift[k][0], b0, (long long)cap);
      const long long n1 = e.migrate_pack_dim(k, 1, B.mshift[k][1], b1, (long long)cap);
      if ((B.nbr[k][0] < 0 && n0) || (B.nbr[k][1] < 0 && n1))
        fail("Lost atoms: an atom left the non-periodic box in dimension %d", k);
      long long m0 = 0, m1 = 0;
      pair_exchange(hc, st, B.nbr[k][0], B.nbr[k][1], b0, n0, b1, n1, m0, m1);
      e.migrate_unpack(hc.rx[0].p, m0);
      e.migrate_unpack(hc.rx[1].p, m1);
    }
  }
  e.rebuild_sort();
  // ---- borders: every direction's atoms straight to the brick behind it ----
  const int ns = (int)B.sdirs.size(), nr = (int)B.rdirs.size();
  e.brick_border_select(B.nsend.data());
  std::vector<long long> soff(W, 0), scnt(W, 0), roff(W, 0), rcnt(W, 0);
  for (int q = 0; q < ns; q++) {
    if (!scnt[B.sdirs[q].peer]) soff[B.sdirs[q].peer] = q;
    scnt[B.sdirs[q].peer]++;
    B.h_cnt[q] = B.nsend[q];
  }
  for (int q = 0; q < nr; q++) {
    if (!rcnt[B.rdirs[q].peer]) roff[B.rdirs[q].peer] = q;
    rcnt[B.rdirs[q].peer]++;
  }
  if (ns) SF_HIP(hipMemcpyAsync(B.d_cnt, B.h_cnt, sizeof(long long) * ns, hipMemcpyHostToDevice, st));
  brick_segments<long long>(hc, st, B.d_cnt, soff, scnt, B.d_cnt + DemEngine::kMaxDirs, roff, rcnt, ncclInt64);
  if (nr)
    SF_HIP(hipMemcpyAsync(B.h_cnt + DemEngine::kMaxDirs, B.d_cnt + DemEngine::kMaxDirs, sizeof(long long) * nr,
                          hipMemcpyDeviceToHost, st));
  SF_HIP(hipStreamSynchronize(st));
  for (int q = 0; q < nr; q++) B.nrecv[q] = B.h_cnt[DemEngine::kMaxDirs + q];
  // block q of the border buffers: kBorderDoubles per atom, blocks in direction order (contiguous per peer)
  std::vector<long long> sb(ns + 1, 0), rb(nr + 1, 0);
  for (int q = 0; q < ns; q++) sb[q + 1] = sb[q] + B.nsend[q] * kBorderDoublesC;
  for (int q = 0; q < nr; q++) rb[q + 1] = rb[q] + B.nrecv[q] * kBorderDoublesC;
  double* btx = B.btx.need((size_t)sb[ns] + 1);
  double* brx = B.brx.need((size_t)rb[nr] + 1);
  e.brick_border_pack(btx);
  std::fill(soff.begin(), soff.end(), 0);
  std::fill(scnt.begin(), scnt.end(), 0);
  std::fill(roff.begin(), roff.end(), 0);
  std::fill(rcnt.begin(), rcnt.end(), 0);
  for (int q = ns - 1; q >= 0; q--) {
    soff[B.sdirs[q].peer] = sb[q];
    scnt[B.sdirs[q].peer] += sb[q + 1] - sb[q];
  }
  for (int q = nr - 1; q >= 0; q--) {
    roff[B.rdirs[q].peer] = rb[q];
    rcnt[B.rdirs[q].peer] += rb[q + 1] - rb[q];
  }
  brick_segments<double>(hc, st, btx, soff, scnt, brx, roff, rcnt, ncclDouble);
  e.brick_ghost_unpack(brx, rb[nr] / kBorderDoublesC);
  e.rebuild_finish();
  hc.n_rebuilds++;
  // ---- layout of the forward halo (valid until the next rebuild): per rank one header double + the blocks ----
  hc.send_cnt.assign(W, 1);
  hc.recv_cnt.assign(W, 1);
  hc.send_cnt[hc.rank] = hc.recv_cnt[hc.rank] = 0;
  for (int q = 0; q < ns; q++) hc.send_cnt[B.sdirs[q].peer] += B.nsend[q] * kForwardDoublesC;
  for (int q = 0; q < nr; q++) hc.recv_cnt[B.rdirs[q].peer] += B.nrecv[q] * kForwardDoublesC;
  hc.send_off.assign(W, 0);
  hc.recv_off.assign(W, 0);
  for (int p = 1; p < W; p++) {
    hc.send_off[p] = hc.send_off[p - 1] + hc.send_cnt[p - 1];
    // (a sender's chunk starts on a 128-byte line of its own: with direct ghost writes the chunks of one receive area
    // are written by different processes, and a cache line must have ONE writer)
    hc.recv_off[p] = (hc.recv_off[p - 1] + hc.recv_cnt[p - 1] + 15) / 16 * 16;
  }
  B.snd = e.brick_send_blocks();
  // the shift of a received block: what its sender -- the brick at offset -d of the block's direction d, seen from
  // here the neighbour at offset s -- adds when it sends in direction -s; recomputed from the sender's coordinates
  for (int q = 0; q < nr; q++) {
    const int code = B.rdirs[q].sender_code;
    const int d[3] = {code % 3 - 1, (code / 3) % 3 - 1, code / 9 - 1};   // the sender's direction
    int cs[3];
    for (int k = 0; k < 3; k++) cs[k] = ((B.c[k] - d[k]) % B.P[k] + B.P[k]) % B.P[k];   // the sender's coordinates
    double sh[3];
    (void)brick_neighbour(B, cs, d, sh);
    for (int k = 0; k < 3; k++) B.rcv.shift[q][k] = sh[k];
  }
  B.rcv.n = nr;
  B.rcv.first[0] = 0;
  {
    std::vector<long long> fill(W, 1);   // doubles already placed in every chunk (the header)
    for (int q = 0; q < ns; q++) {
      const int p = B.sdirs[q].peer;
      B.snd.off[q] = hc.send_off[p] + fill[p];
      fill[p] += B.nsend[q] * kForwardDoublesC;
    }
    std::fill(fill.begin(), fill.end(), 1);
    for (int q = 0; q < nr; q++) {
      const int p = B.rdirs[q].peer;
      B.rcv.off[q] = hc.recv_off[p] + fill[p];
      fill[p] += B.nrecv[q] * kForwardDoublesC;
      B.rcv.first[q + 1] = B.rcv.first[q] + (int)B.nrecv[q];
    }
  }
  int nh = 0;
  for (int p = 0; p < W; p++) {
    if (p == hc.rank) continue;
    hc.h_hdr[nh] = (int)hc.send_off[p];
    hc.h_hdr[W + nh] = (int)hc.recv_off[p];
    nh++;
  }
  SF_HIP(hipMemcpyAsync(hc.d_hdr, hc.h_hdr, sizeof(int) * 2 * W, hipMemcpyHostToDevice, st));
  sf_halo_layout& L = hc.lay;
  L.world = W;
  L.send_off = hc.send_off.data();
  L.send_cnt = hc.send_cnt.data();
  L.recv_off = hc.recv_off.data();
  L.recv_cnt = hc.recv_cnt.data();
  L.dev_shdr = hc.d_hdr;
  L.dev_rhdr = hc.d_hdr + W;
  L.dev_tx = hc.a2a_tx.need((size_t)(hc.send_off[W - 1] + hc.send_cnt[W - 1]) + 1);
  L.dev_rx = hc.a2a_rx.need((size_t)(hc.recv_off[W - 1] + hc.recv_cnt[W - 1]) + 1);
  hc.lay_valid = true;
  double* blk2[2 * DemEngine::kMaxDirs];
  double4* blk6[6 * DemEngine::kMaxDirs];
  if (hc.direct && hc.direct->on && hc.direct->mode == 2 && gs_rebuild(S, hc, blk6)) {
    e.brick_set_forward_gs(B.snd, blk6);
  } else if (hc.direct && hc.direct->on && hc.direct->mode == 1 && direct_rebuild(S, hc, blk2)) {
    e.brick_set_forward_tx(B.snd, L.dev_tx, L.dev_shdr, W - 1, blk2);
    e.set_tx_parity((int)(hc.direct->xseq & 1));
  } else {
    e.brick_set_forward_tx(B.snd, L.dev_tx, L.dev_shdr, W - 1);   // (the RCCL exchange; also after a lost direct transport)
  }
  hc.rebuild_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
}

// sub-steps first_k .. end_k - 1 of a run of n: per sub-step one pack kernel, ONE grouped ncclSend/ncclRecv with the
// neighbour bricks, one unpack kernel, the sub-step kernel; one synchronisation; returns the voted rebuild trigger
static int brick_halo_run(SfLammps& S, HaloComm& hc, int first_k, int end_k, int n)
{
  if (!hc.lay_valid) fail("sf_brick_step: no halo layout (rebuild first)");
  BrickState& B = *hc.brick;
  DemEngine& e = S.eng;
  hipStream_t main = e.stream();
  const int nh = hc.world - 1;
  if (hc.direct && hc.direct->on && hc.direct->mode == 2) {
    // ghost slots: the sub-step kernels hand the border records and the votes to each other; what is left for the host
    // is the stand-alone pack in front of a launch no sub-step kernel has written the records for (start of a run, first
    // launch after a rebuild) and, at the end of a piece that stops early, the wait for the last votes
    static FILE* tr = nullptr;   // (SF_DEBUG_HALO_TRACE=<prefix>: one line per piece and rank, flushed: what a stalled rank did last)
    if (!tr && getenv("SF_DEBUG_HALO_TRACE")) {
      char name[512];
      snprintf(name, sizeof(name), "%s.%d", getenv("SF_DEBUG_HALO_TRACE"), hc.rank);
      tr = fopen(name, "w");
    }
    if (tr) {
      fprintf(tr, "piece [%d, %d) of %d: first launch %lld, records %s\n", first_k, end_k, n, e.gs_seq(),
              e.forward_tx_written() ? "written by the last kernel" : "to be packed");
      fflush(tr);
    }
    for (int s = first_k; s < end_k; s++) {
      if (!e.forward_tx_written()) e.gs_pack();
      e.substep_k(s == n - 1, s);
    }
    if (end_k < n) e.gs_close(end_k);
    const int trigger = e.batch_end(first_k, end_k - first_k);
    if (tr) {
      fprintf(tr, "  done: trigger %d, next launch %lld, timeout word %d (peer %d at %d)\n", trigger, e.gs_seq(), e.halo_timeout(),
              e.halo_timeout_peer(), e.halo_timeout_seen());
      fflush(tr);
    }
    if (e.halo_timeout() && getenv("SF_DEBUG_HALO")) {
      // what this rank sees of everybody: the (flag << 32) | vote word of every sending rank
      int lines[32 * DemEngine::kSyncStride];
      (void)hipMemcpy(lines, hc.direct->my_sync, sizeof(int) * DemEngine::kSyncStride * hc.world, hipMemcpyDeviceToHost);
      fprintf(stderr, "[sedifoam_amd] rank %d: ghost-slot wait ran out in launch %d (peer %d stood at %d); next launch %lld, "
              "piece [%d, %d) of %d, trigger word %d; lines:", hc.rank, e.halo_timeout(), e.halo_timeout_peer(),
              e.halo_timeout_seen(), e.gs_seq(), first_k, end_k, n, trigger);
      for (int r = 0; r < hc.world; r++)
        fprintf(stderr, "  r%d {flag %d, vote %d}", r, lines[DemEngine::kSyncStride * r + 1], lines[DemEngine::kSyncStride * r]);
      fprintf(stderr, "\n");
    }
    if (e.halo_timeout())
      fail("ghost slots: rank %d waited for the flag of launch %d, rank %d stood at %d when the wait ran out (a peer died, or "
           "the ranks disagree about the launches they queue); SF_HALO_DIRECT=0 selects the RCCL exchange",
           hc.rank, e.halo_timeout(), e.halo_timeout_peer(), e.halo_timeout_seen());
    return trigger;
  }
  auto exchange = [&]() {
    // (the sub-step kernel that integrated the border atoms has written their records and the vote headers itself;
    // the pack kernel runs only in front of the first sub-step after a rebuild / at the start of a run)
    if (!e.forward_tx_written()) e.brick_forward_pack(B.snd, hc.lay.dev_tx, hc.lay.dev_shdr, nh);
    if (hc.direct && hc.direct->on) {
      // the records are already in the neighbours' receive areas: one kernel publishes, waits and unpacks
      DirectHalo& D = *hc.direct;
      const int par = (int)(D.xseq & 1);
      e.brick_direct_unpack(B.rcv, D.rx[par], direct_sync(hc, (int)(D.xseq + 1), par));
      D.xseq++;
      e.set_tx_parity((int)(D.xseq & 1));
      return;
    }
    hc.all_to_all(hc.lay, main);
    e.brick_forward_unpack(B.rcv, hc.lay.dev_rx, hc.lay.dev_rhdr, nh);
  };
  for (int s = first_k; s < end_k; s++) {
    if (s == first_k && hc.pre_exchanged) {   // (ghosts and votes in front of this sub-step are in place)
      hc.pre_exchanged = false;
      e.substep_k(s == n - 1, s);
      continue;
    }
    const bool timed = e.profiling() && s % 8 == 4 && s > first_k;
    if (timed) {
      if (hc.xev_used + 2 > hc.xev.size())
        for (int k = 0; k < 2; k++) {
          hipEvent_t ev;
          SF_HIP(hipEventCreate(&ev));
          hc.xev.push_back(ev);
        }
      SF_HIP(hipEventRecord(hc.xev[hc.xev_used], main));
    }
    exchange();
    if (timed) {
      SF_HIP(hipEventRecord(hc.xev[hc.xev_used + 1], main));
      hc.xev_used += 2;
    }
    e.substep_k(s == n - 1, s);
  }
  if (end_k < n) {   // (a piece that stops early closes with the vote exchange: see halo_run_layout)
    exchange();
    hc.pre_exchanged = true;
  }
  const int trigger = e.batch_end(first_k, end_k - first_k);
  if (hc.direct && hc.direct->on && e.halo_timeout())
    fail("direct ghost writes: rank %d waited for exchange %d, rank %d had confirmed %d when the wait ran out (a peer "
         "died, or the ranks disagree about the exchanges they run); SF_HALO_DIRECT=0 selects the RCCL exchange",
         hc.rank, e.halo_timeout(), e.halo_timeout_peer(), e.halo_timeout_seen());
  hc.harvest_exchange_profile();
  return trigger;
}

static void brick_step(SfLammps& S, HaloComm& hc, int n)
{
  Range r("lammps");
  DemEngine& e = S.eng;
  e.run_begin();
  hc.pre_exchanged = false;
  if (hc.direct && hc.direct->on && hc.direct->mode == 2 && e.gs_on() && !e.gs_mapping_valid()) {
    // (the buffer parity moved against the launch numbers without a rebuild -- the setup evaluation flips the buffers: the
    // neighbours must learn which arrays this rank's next launch reads.  The same event on every rank: collective)
    double4* blk6[6 * DemEngine::kMaxDirs];
    if (gs_rebuild(S, hc, blk6)) e.brick_set_forward_gs(hc.brick->snd, blk6);
    else e.brick_set_forward_tx(hc.brick->snd, hc.lay.dev_tx, hc.lay.dev_shdr, hc.world - 1);
  }
  int k = 0;
  while (k < n) {
    const int end = k + hc.predict.chunk(e.nsteps(), n - k);
    const int trig = brick_halo_run(S, hc, k, end, n);
    if (trig >= end) {
      k = end;
      continue;
    }
    k = trig + 1;
    hc.pre_exchanged = false;
    {
      DemEngine::InRunGuard guard(e);
      brick_rebuild(S, hc);
    }
    hc.predict.rebuilt(e.nsteps());
  }
}

// ------------------------------------------------------------------------------------------------
// Mesh partitioned by the slab planes: the exchanges of the coupled step over the engine's communicator.
// Fields are [nz][ny][nxs][ncomp], nxs = owned layers + one ghost layer on each side.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_plane_get(const double* f, int nlines, int nxs, int ncomp, int ix, double* out)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nlines * ncomp) return;
  const int line = k / ncomp, c = k - line * ncomp;
  out[k] = f[((size_t)line * nxs + ix) * ncomp + c];
}
// mode 0: f[ix] += in ; 1: f[ix] = in ; 2: f[ix] = f[ix_src] (no neighbour: zero gradient)
__global__ __launch_bounds__(256) void k_plane_put(double* f, int nlines, int nxs, int ncomp, int ix, const double* in,
                                                   int mode, int ix_src)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nlines * ncomp) return;
  const int line = k / ncomp, c = k - line * ncomp;
  double* dst = &f[((size_t)line * nxs + ix) * ncomp + c];
  if (mode == 0) *dst += in[k];
  else if (mode == 1) *dst = in[k];
  else *dst = f[((size_t)line * nxs + ix_src) * ncomp + c];
}
// owned columns of the work array [NL][nxs] -> [W * nlq][nxl] (rows beyond NL: zero)
__global__ __launch_bounds__(256) void k_lines_interior(const double* work, long long NL, int nxs, long long rows, double* out)
{
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nxl = nxs - 2;
  if (k >= rows * nxl) return;
  const long long l = k / nxl;
  const int x = (int)(k - l * nxl);
  out[k] = l < NL ? work[l * nxs + 1 + x] : 0.0;
}
// recv [W][nlq][nxl] (chunk p = rank p's columns of my lines) -> lines [nlq][W * nxl]
__global__ __launch_bounds__(256) void k_lines_assemble(const double* recv, int W, long long nlq, int nxl, double* lines)
{
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long nxg = (long long)W * nxl;
  if (k >= nlq * nxg) return;
  const long long l = k / nxg;
  const int xg = (int)(k - l * nxg);
  const int p = xg / nxl, x = xg - p * nxl;
  lines[k] = recv[((long long)p * nlq + l) * nxl + x];
}
// lines [nlq][nxg] -> send [W][nlq][nxs]: for rank p its owned columns and the two next to its slab (wrapped / clamped)
__global__ __launch_bounds__(256) void k_lines_columns(const double* lines, int W, long long nlq, int nxl, int periodic,
                                                       double* send)
{
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nxs = nxl + 2;
  const long long nxg = (long long)W * nxl;
  if (k >= (long long)W * nlq * nxs) return;
  const int c = (int)(k % nxs);
  const long long l = (k / nxs) % nlq;
  const int p = (int)(k / ((long long)nxs * nlq));
  long long xg = (long long)p * nxl - 1 + c;
  if (periodic) xg = (xg + nxg) % nxg;
  else xg = xg < 0 ? 0 : (xg >= nxg ? nxg - 1 : xg);
  send[k] = lines[l * nxg + xg];
}

struct CloudSlab {
  SfLammps* L = nullptr;
  HaloComm* hc = nullptr;
  int n[3] = {0, 0, 0}, nxg = 0, per_x = 0, smoothing = 0;
  int left = -1, right = -1;
};

static CloudSlab cloud_slab(void* cloud)
{
  CloudSlab C;
  void* lmp = nullptr;
  if (sf_cloud_slab_info(cloud, &lmp, C.n, &C.nxg, &C.per_x, &C.smoothing) != 0) fail("%s", last_error().c_str());
  C.L = static_cast<SfLammps*>(lmp);
  C.hc = static_cast<HaloComm*>(C.L->halo);
  if (!C.hc || !C.hc->comm || C.hc->brick) fail("sf_cloud_slab_*: the engine has no slab communicator (sf_slab_init first)");
  if (C.nxg <= 0) fail("sf_cloud_slab_*: not a slab mesh (sf_cloud_mesh.slab_nx_global)");
  const int W = C.hc->world, r = C.hc->rank;
  if ((long long)(C.n[0] - 2) * W != C.nxg) fail("sf_cloud_slab_*: %d owned layers x %d ranks != %d layers", C.n[0] - 2, W, C.nxg);
  C.left = r > 0 ? r - 1 : (C.per_x ? W - 1 : -1);
  C.right = r < W - 1 ? r + 1 : (C.per_x ? 0 : -1);
  return C;
}

// two face messages: s_left -> left neighbour, s_right -> right neighbour; r_left / r_right <- what they sent
static void face_exchange(CloudSlab& C, hipStream_t st, const double* s_left, const double* s_right, double* r_left,
                          double* r_right, size_t n)
{
  RcclApi& a = rccl();
  HaloComm& hc = *C.hc;
  if (hc.world == 1) {   // (a single periodic slab is its own neighbour on both sides)
    if (C.left >= 0) SF_HIP(hipMemcpyAsync(r_left, s_right, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
    if (C.right >= 0) SF_HIP(hipMemcpyAsync(r_right, s_left, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
    return;
  }
  SF_NCCL(a.GroupStart());
  if (C.left >= 0) SF_NCCL(a.Send(s_left, n, ncclDouble, C.left, hc.comm, st));
  if (C.right >= 0) SF_NCCL(a.Send(s_right, n, ncclDouble, C.right, hc.comm, st));
  if (C.right >= 0) SF_NCCL(a.Recv(r_right, n, ncclDouble, C.right, hc.comm, st));   // (leftward traffic comes from my right)
  if (C.left >= 0) SF_NCCL(a.Recv(r_left, n, ncclDouble, C.left, hc.comm, st));
  SF_NCCL(a.GroupEnd());
}

static void cloud_halo_add_field(CloudSlab& C, hipStream_t st, double* f, int ncomp)
{
  HaloComm& hc = *C.hc;
  const int nxs = C.n[0], nlines = C.n[1] * C.n[2];
  const size_t np = (size_t)nlines * ncomp;
  double* buf = hc.mig[0].need(4 * np + 4);
  double *sl = buf, *sr = buf + np, *rl = buf + 2 * np, *rr = buf + 3 * np;
  const int nb = div_up((int)np, 256);
  // what this rank's particles deposited in its ghost layers belongs to the neighbours' edge layers
  k_plane_get<<<nb, 256, 0, st>>>(f, nlines, nxs, ncomp, 0, sl);
  k_plane_get<<<nb, 256, 0, st>>>(f, nlines, nxs, ncomp, nxs - 1, sr);
  face_exchange(C, st, sl, sr, rl, rr, np);
  if (C.left >= 0) k_plane_put<<<nb, 256, 0, st>>>(f, nlines, nxs, ncomp, 1, rl, 0, 0);
  if (C.right >= 0) k_plane_put<<<nb, 256, 0, st>>>(f, nlines, nxs, ncomp, nxs - 2, rr, 0, 0);
  // refresh the ghost layers with the neighbours' edge values (no neighbour: zero gradient)
  k_plane_get<<<nb, 256, 0, st>>>(f, nlines, nxs, ncomp, 1, sl);
  k_plane_get<<<nb, 256, 0, st>>>(f, nlines, nxs, ncomp, nxs - 2, sr);
  face_exchange(C, st, sl, sr, rl, rr, np);
  k_plane_put<<<nb, 256, 0, st>>>(f, nlines, nxs, ncomp, 0, rl, C.left >= 0 ? 1 : 2, 1);
  k_plane_put<<<nb, 256, 0, st>>>(f, nlines, nxs, ncomp, nxs - 1, rr, C.right >= 0 ? 1 : 2, nxs - 2);
}

// chunk p of `send` (equal sizes) to rank p, chunk p of `recv` from rank p
static void equal_all_to_all(HaloComm& hc, hipStream_t st, const double* send, double* recv, size_t chunk)
{
  RcclApi& a = rccl();
  SF_HIP(hipMemcpyAsync(recv + (size_t)hc.rank * chunk, send + (size_t)hc.rank * chunk, sizeof(double) * chunk,
                        hipMemcpyDeviceToDevice, st));
  if (hc.world == 1) return;
  SF_NCCL(a.GroupStart());
  for (int p = 0; p < hc.world; p++) {
    if (p == hc.rank) continue;
    SF_NCCL(a.Send(send + (size_t)p * chunk, chunk, ncclDouble, p, hc.comm, st));
    SF_NCCL(a.Recv(recv + (size_t)p * chunk, chunk, ncclDouble, p, hc.comm, st));
  }
  SF_NCCL(a.GroupEnd());
}

static void cloud_xsolve(CloudSlab& C, void* cloud, hipStream_t st)
{
  HaloComm& hc = *C.hc;
  const int W = hc.world, r = hc.rank, nxs = C.n[0], nxl = nxs - 2;
  double* work = nullptr;
  int nf = 0;
  if (sf_cloud_smooth_work(cloud, &work, &nf) != 0) fail("%s", last_error().c_str());
  const long long NL = (long long)nf * C.n[2] * C.n[1];
  const long long nlq = (NL + W - 1) / W;
  const size_t fwd = (size_t)W * nlq * nxl, bwd = (size_t)W * nlq * nxs;
  double* a = hc.mig[0].need(fwd + bwd + 8);        // a: send / lines ; b: recv / send back
  double* b = hc.mig[1].need(fwd + bwd + 8);
  k_lines_interior<<<div_up((long long)fwd, 256), 256, 0, st>>>(work, NL, nxs, (long long)W * nlq, a);
  equal_all_to_all(hc, st, a, b, (size_t)nlq * nxl);                  // b[p] = rank p's columns of my lines
  double* lines = a;
  k_lines_assemble<<<div_up((long long)fwd, 256), 256, 0, st>>>(b, W, nlq, nxl, lines);
  const long long nvalid = std::max(0LL, std::min(nlq, NL - (long long)r * nlq));
  if (sf_cloud_smooth_xsolve(cloud, lines, nvalid, (long long)r * nlq) != 0) fail("%s", last_error().c_str());
  double* cols = b;
  k_lines_columns<<<div_up((long long)bwd, 256), 256, 0, st>>>(lines, W, nlq, nxl, C.per_x, cols);
  double* back = a + fwd;
  equal_all_to_all(hc, st, cols, back, (size_t)nlq * nxs);            // back[q] = rank q's lines, my columns
  SF_HIP(hipMemcpyAsync(work, back, sizeof(double) * (size_t)NL * nxs, hipMemcpyDeviceToDevice, st));
}

}  // namespace sf

using sf::SfLammps;
static SfLammps* H(void* p)
{
  if (!p) sf::fail("null engine handle");
  return static_cast<SfLammps*>(p);
}

extern "C" {

int sf_dem_comm_unique_id(char* id128)
{
  SF_API_BEGIN
  ncclUniqueId id;
  SF_NCCL(sf::rccl().GetUniqueId(&id));
  static_assert(sizeof(id.internal) == 128, "ncclUniqueId is 128 bytes");
  memcpy(id128, id.internal, 128);
  SF_API_END(0)
}

int sf_dem_comm_init(void* ptr, const char* id128, int rank, int world)
{
  SF_API_BEGIN
  SfLammps* L = H(ptr);
  if (world < 1 || rank < 0 || rank >= world) sf::fail("sf_dem_comm_init: rank %d of %d", rank, world);
  if (L->halo && L->halo_delete) L->halo_delete(L->halo);   // (a second init replaces the communicator)
  auto* hc = new sf::HaloComm();
  L->halo = hc;
  L->halo_delete = sf::halo_deleter;
  hc->rank = rank;
  hc->world = world;
  ncclUniqueId id;
  memcpy(id.internal, id128, 128);
  SF_NCCL(sf::rccl().CommInitRank(&hc->comm, world, id, rank));
  SF_HIP(hipEventCreateWithFlags(&hc->ev_boundary, hipEventDisableTiming));
  SF_HIP(hipEventCreateWithFlags(&hc->ev_halo, hipEventDisableTiming));
  SF_API_END(0)
}

int sf_dem_halo_run(void* ptr, int first_k, int n, const sf_halo_layout* lay, int* trigger)
{
  SF_API_BEGIN
  SfLammps* L = H(ptr);
  auto* hc = static_cast<sf::HaloComm*>(L->halo);
  if (!hc || !hc->comm) sf::fail("sf_dem_halo_run: call sf_dem_comm_init first");
  if (!lay || !trigger) sf::fail("sf_dem_halo_run: null argument");
  if (lay->world != hc->world) sf::fail("sf_dem_halo_run: layout for %d ranks, communicator has %d", lay->world, hc->world);
  *trigger = sf::halo_run_layout(*L, *hc, first_k, n, n, *lay);
  SF_API_END(0)
}

int sf_slab_init(void* ptr, const char* id128, int rank, int world, double xlo, double xhi, int periodic_x)
{
  SF_API_BEGIN
  SfLammps* L = H(ptr);
  if (sf_dem_comm_init(ptr, id128, rank, world) != 0) sf::fail("%s", sf::last_error().c_str());
  auto* hc = static_cast<sf::HaloComm*>(L->halo);
  hc->slab = true;
  hc->periodic_x = periodic_x != 0;
  hc->box_lo = xlo;
  hc->box_len = xhi - xlo;
  hc->left = rank > 0 ? rank - 1 : (periodic_x ? world - 1 : -1);
  hc->right = rank < world - 1 ? rank + 1 : (periodic_x ? 0 : -1);
  // shift applied to what goes out through the global box faces
  hc->shift_left = (rank == 0 && periodic_x) ? hc->box_len : 0.0;
  hc->shift_right = (rank == world - 1 && periodic_x) ? -hc->box_len : 0.0;
  const double w = hc->box_len / world;
  const double sublo = xlo + rank * w, subhi = rank == world - 1 ? xhi : xlo + (rank + 1) * w;
  L->eng.set_subdomain(rank, world, sublo, subhi);
  sf::slab_scratch(*hc);
  if (const char* q = getenv("SF_QUEUE_PREDICT")) hc->predict.on = atoi(q) != 0;
  SF_API_END(0)
}

// 3-D processor grid px x py x pz (= world) over the engine's box (set_box / boundary before this call); rank r owns
// the brick (r % px, (r / px) % py, r / (px py)).  sf_slab_setup / _step / _rebuild then drive the brick decomposition.
int sf_brick_init(void* ptr, const char* id128, int rank, int world, int px, int py, int pz)
{
  SF_API_BEGIN
  SfLammps* L = H(ptr);
  if (px < 1 || py < 1 || pz < 1 || px * py * pz != world) sf::fail("sf_brick_init: %d x %d x %d bricks for %d ranks", px, py, pz, world);
  if (sf_dem_comm_init(ptr, id128, rank, world) != 0) sf::fail("%s", sf::last_error().c_str());
  auto* hc = static_cast<sf::HaloComm*>(L->halo);
  hc->slab = true;
  hc->brick = new sf::BrickState();
  sf::BrickState& B = *hc->brick;
  B.P[0] = px; B.P[1] = py; B.P[2] = pz;
  B.c[0] = rank % px; B.c[1] = (rank / px) % py; B.c[2] = rank / (px * py);
  L->eng.box(B.lo, B.hi, B.periodic);
  double lo[3], hi[3];
  int ext[3];
  // SF_HALO_SELF_COMM=1 (development, one rank): the periodic dimensions are external too -- the rank exchanges border
  // records and migrating atoms with ITSELF through the same code as with a neighbour (what the exchange costs next to
  // the sub-step kernel without another process' kernels on the GPU: tests/trace_selfcomm.sh)
  const bool self_comm = world == 1 && getenv("SF_HALO_SELF_COMM") && atoi(getenv("SF_HALO_SELF_COMM")) != 0;
  for (int k = 0; k < 3; k++) {
    B.ext[k] = B.P[k] > 1 || (self_comm && B.periodic[k]);
    ext[k] = B.ext[k] ? 1 : 0;
    const double w = (B.hi[k] - B.lo[k]) / B.P[k];
    lo[k] = B.lo[k] + B.c[k] * w;
    hi[k] = B.c[k] == B.P[k] - 1 ? B.hi[k] : B.lo[k] + (B.c[k] + 1) * w;
  }
  L->eng.set_subdomain3(rank, world, lo, hi, ext);
  sf::slab_scratch(*hc);
  sf::brick_topology(*hc, L->eng);
  if (const char* q = getenv("SF_QUEUE_PREDICT")) hc->predict.on = atoi(q) != 0;
  sf::direct_init(*L, *hc);
  SF_API_END(0)
}

// The exchange pattern of one rank of a px x py x pz grid, host logic only (no device, no communicator): the blocks it
// sends -- (peer, direction code (dx+1) + 3 (dy+1) + 9 (dz+1)) in send order -- and the blocks it receives -- (peer,
// the SENDER's direction code) in receive order.  For every pair of ranks the sender's list restricted to the receiver
// must equal the receiver's list restricted to the sender: what tests/test_abi_and_host.py checks on the CPU.
int sf_brick_pattern(int rank, int px, int py, int pz, const int* periodic, int* nsend, int* send_peer, int* send_code,
                     int* nrecv, int* recv_peer, int* recv_code, int* face_nbr)
{
  SF_API_BEGIN
  if (px < 1 || py < 1 || pz < 1 || rank < 0 || rank >= px * py * pz) sf::fail("sf_brick_pattern: rank %d of %d x %d x %d", rank, px, py, pz);
  sf::BrickState B;
  B.P[0] = px; B.P[1] = py; B.P[2] = pz;
  B.c[0] = rank % px; B.c[1] = (rank / px) % py; B.c[2] = rank / (px * py);
  for (int k = 0; k < 3; k++) {
    B.lo[k] = 0.0;
    B.hi[k] = 1.0;
    B.periodic[k] = periodic[k];
    B.ext[k] = B.P[k] > 1;
  }
  sf::brick_directions(B);
  *nsend = (int)B.sdirs.size();
  *nrecv = (int)B.rdirs.size();
  for (size_t q = 0; q < B.sdirs.size(); q++) {
    send_peer[q] = B.sdirs[q].peer;
    send_code[q] = B.sdirs[q].code;
  }
  for (size_t q = 0; q < B.rdirs.size(); q++) {
    recv_peer[q] = B.rdirs[q].peer;
    recv_code[q] = B.rdirs[q].sender_code;
  }
  if (face_nbr)
    for (int k = 0; k < 3; k++)
      for (int side = 0; side < 2; side++) face_nbr[2 * k + side] = B.nbr[k][side];
  SF_API_END(0)
}

int sf_cloud_slab_halo_add(void* cloud, int fields)
{
  SF_API_BEGIN
  sf::CloudSlab C = sf::cloud_slab(cloud);
  hipStream_t st = C.L->eng.stream();
  double *g = nullptr, *u = nullptr, *a = nullptr;
  int nc = 0;
  if (sf_cloud_device_fields(cloud, &g, &u, &a, &nc) != 0) sf::fail("%s", sf::last_error().c_str());
  if (fields & 1) sf::cloud_halo_add_field(C, st, g, 1);
  if (fields & 2) sf::cloud_halo_add_field(C, st, u, 3);
  if (fields & 4) sf::cloud_halo_add_field(C, st, a, 3);
  SF_API_END(0)
}

int sf_cloud_slab_phase(void* cloud, int phase)
{
  SF_API_BEGIN
  int rc = sf_cloud_phase(cloud, phase);
  if (rc < 0) sf::fail("%s", sf::last_error().c_str());
  if (rc == 1) {
    sf::CloudSlab C = sf::cloud_slab(cloud);
    sf::cloud_xsolve(C, cloud, C.L->eng.stream());
    rc = sf_cloud_phase(cloud, phase);
    if (rc != 0) sf::fail("sf_cloud_slab_phase %d did not finish after its x solve (%d)", phase, rc);
  }
  SF_API_END(0)
}

static sf::HaloComm* slab_of(SfLammps* L)
{
  auto* hc = static_cast<sf::HaloComm*>(L->halo);
  if (!hc || !hc->slab) sf::fail("sf_slab_*: call sf_slab_init first");
  return hc;
}

int sf_slab_setup(void* ptr)
{
  SF_API_BEGIN
  SfLammps* L = H(ptr);
  sf::HaloComm* hc = slab_of(L);
  sf::DemEngine& e = L->eng;
  hipStream_t st = e.stream();
  // list / ghost cutoff 2 r_max + skin: r_max over ALL ranks ([3P] MPI_Allreduce of maxrad_dynamic)
  e.set_global_max_radius(sf::slab_allreduce(*hc, st, e.local_max_radius(), ncclMax));
  if (hc->brick) sf::brick_rebuild(*L, *hc);
  else sf::slab_rebuild(*L, *hc);
  // pair lubricate/poly: volume fraction of ALL particles (MPI_Allreduce, pair_lubricate_poly.cpp:540-543)
  e.set_global_particle_volume(sf::slab_allreduce(*hc, st, e.local_particle_volume(), ncclSum));
  e.setup();
  hc->is_setup = true;
  hc->rebuild_ms = 0.0;   // (the first rebuild allocates)
  hc->rebuilds_at_setup = hc->n_rebuilds;
  SF_API_END(0)
}

int sf_slab_rebuild(void* ptr)
{
  SF_API_BEGIN
  SfLammps* L = H(ptr);
  if (slab_of(L)->brick) sf::brick_rebuild(*L, *slab_of(L));
  else sf::slab_rebuild(*L, *slab_of(L));
  slab_of(L)->predict.external(L->eng.nsteps());   // (not a trigger of the stepping loop: the interval estimate stands)
  SF_API_END(0)
}

int sf_slab_step(void* ptr, int n)
{
  SF_API_BEGIN
  SfLammps* L = H(ptr);
  sf::HaloComm* hc = slab_of(L);
  if (!hc->is_setup) {
    if (sf_slab_setup(ptr) != 0) sf::fail("%s", sf::last_error().c_str());
  }
  if (n > 0) {
    if (hc->brick) sf::brick_step(*L, *hc, n);
    else sf::slab_step(*L, *hc, n);
  }
  SF_API_END(0)
}

int sf_slab_exchange_profile(void* ptr, long long* exchanges, double* ms)
{
  SF_API_BEGIN
  sf::HaloComm* hc = slab_of(H(ptr));
  *exchanges = hc->x_count;
  *ms = hc->x_ms;
  hc->x_count = 0;
  hc->x_ms = 0.0;
  SF_API_END(0)
}

int sf_slab_rebuild_profile(void* ptr, long long* rebuilds, double* ms)
{
  SF_API_BEGIN
  sf::HaloComm* hc = slab_of(H(ptr));
  *rebuilds = hc->n_rebuilds - hc->rebuilds_at_setup;
  *ms = hc->rebuild_ms;
  SF_API_END(0)
}

// what the communicator of this engine really is: ranks RCCL itself counts on it (ncclCommCount), the library's
// version and the file its symbols were loaded from -- so that a benchmark line can prove which transport it ran on
int sf_slab_comm_info(void* ptr, int* comm_ranks, int* rccl_version, char* lib_path, int lib_path_len)
{
  SF_API_BEGIN
  sf::HaloComm* hc = slab_of(H(ptr));
  sf::RcclApi& a = sf::rccl();
  int n = 0, v = 0;
  SF_NCCL(a.CommCount(hc->comm, &n));
  SF_NCCL(a.GetVersion(&v));
  if (comm_ranks) *comm_ranks = n;
  if (rccl_version) *rccl_version = v;
  if (lib_path && lib_path_len > 0) snprintf(lib_path, (size_t)lib_path_len, "%s", a.path);
  SF_API_END(0)
}

int sf_slab_active(void* ptr)
{
  SF_API_BEGIN
  auto* hc = static_cast<sf::HaloComm*>(H(ptr)->halo);
  const int on = hc && hc->slab ? 1 : 0;
  SF_API_END(on)
}

int sf_slab_direct_halo(void* ptr)
{
  SF_API_BEGIN
  auto* hc = static_cast<sf::HaloComm*>(H(ptr)->halo);
  const int on = hc && hc->direct && hc->direct->on ? hc->direct->mode : 0;   // (1: receive areas, 2: ghost slots)
  SF_API_END(on)
}

int sf_slab_allreduce_sum(void* ptr, double* values, int n)
{
  SF_API_BEGIN
  SfLammps* L = H(ptr);
  auto* hc = static_cast<sf::HaloComm*>(L->halo);
  if (!hc || !hc->comm) sf::fail("sf_slab_allreduce_sum: no communicator (sf_slab_init / sf_brick_init first)");
  if (n < 0 || (n && !values)) sf::fail("sf_slab_allreduce_sum: bad arguments");
  if (hc->world > 1 && n > 0) {
    hipStream_t st = L->eng.stream();
    sf::GrowBuf buf;
    double* d = buf.need(2 * (size_t)n);
    SF_HIP(hipMemcpyAsync(d, values, sizeof(double) * n, hipMemcpyHostToDevice, st));
    SF_NCCL(sf::rccl().AllReduce(d, d + n, (size_t)n, ncclDouble, ncclSum, hc->comm, st));
    SF_HIP(hipMemcpyAsync(values, d + n, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    SF_HIP(hipStreamSynchronize(st));
  }
  SF_API_END(0)
}

long long sf_slab_rebuild_count(void* ptr)
{
  SF_API_BEGIN
  const long long n = slab_of(H(ptr))->n_rebuilds;
  SF_API_END(n)
}

int sf_slab_layout_get(void* ptr, sf_halo_layout* out)
{
  SF_API_BEGIN
  sf::HaloComm* hc = slab_of(H(ptr));
  if (!hc->lay_valid) sf::fail("sf_slab_layout_get: no rebuild yet");
  *out = hc->lay;
  SF_API_END(0)
}

}  // extern "C"
