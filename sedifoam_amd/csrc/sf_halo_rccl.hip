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

#include <cstdlib>
#include <cstring>

#include "../../include/sedifoam_amd.h"
#include "sf_handles.h"

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
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi& rccl()
{
  static RcclApi api;
  if (api.lib) return api;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (api.lib) break;
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
  api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
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

struct HaloComm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  hipEvent_t ev_boundary = nullptr, ev_halo = nullptr;
  ~HaloComm()
  {
    if (comm) (void)rccl().CommDestroy(comm);
    if (ev_boundary) (void)hipEventDestroy(ev_boundary);
    if (ev_halo) (void)hipEventDestroy(ev_halo);
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
    static const int fake_us = getenv("SF_HALO_FAKE_DELAY_US") ? atoi(getenv("SF_HALO_FAKE_DELAY_US")) : 0;
    if (fake_us > 0) k_fake_link_delay<<<1, 64, 0, st>>>((long long)fake_us * 100);   // wall_clock64: 100 MHz
  }
};

static void halo_deleter(void* p) { delete static_cast<HaloComm*>(p); }

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
  sf::DemEngine& e = L->eng;
  hipStream_t main = e.stream();
  auto exchange = [&](int kstep, hipStream_t st) {
    e.forward_pack_fused(lay->shift_left, lay->soff_l, lay->shift_right, lay->soff_r, lay->dev_shdr, lay->world,
                         lay->dev_tx);
    hc->all_to_all(*lay, st);
    e.forward_unpack_fused(lay->dev_rx, lay->roff_l, lay->n_from_left, lay->roff_r, lay->n_from_right, lay->dev_rhdr,
                           lay->world, kstep);
  };
  const int launched = n - first_k;
  if (!e.overlap()) {
    for (int s = first_k; s < n; s++) {
      exchange(-1, main);
      e.substep_k(s == n - 1, s);
    }
    *trigger = e.batch_end(first_k, launched);
  } else {
    hipStream_t cs = e.comm_stream();
    SF_HIP(hipEventRecord(hc->ev_boundary, main));
    SF_HIP(hipStreamWaitEvent(cs, hc->ev_boundary, 0));
    exchange(first_k - 1, cs);                       // ghosts + vote before sub-step first_k
    SF_HIP(hipEventRecord(hc->ev_halo, cs));
    for (int s = first_k; s < n; s++) {
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
    *trigger = e.overlap_batch_end(first_k, launched, n - 1);
  }
  SF_API_END(0)
}

}  // extern "C"
