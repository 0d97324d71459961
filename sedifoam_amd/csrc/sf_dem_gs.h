// sf_dem_gs.h -- ghost slots: the device side of the forward halo that needs no kernel between two sub-step kernels
// (sub-step kernel: sf_dem_kernels.h; stand-alone pack / flag / piece-end kernels: sf_dem_halo.hip; driver: sf_halo_rccl.hip)
#pragma once
#include <climits>

#include "sf_dem.h"

namespace sf {

// ------------------------------------------------------------------------------------------------
// ghost slots (decomposed domain, template parameter GS of the sub-step kernel): the records of the ghosts of other GPUs
// live in a fine-grained area the NEIGHBOURS' sub-step kernels write straight into (IPC mapping), so nothing stands
// between two sub-step kernels.  Both sides use system-coherent accesses (sc0 sc1: write-through stores, loads that take
// nothing from L1 / L2), the hand-off is {records, s_waitcnt vmcnt(0), completion count, vote, flag} on the sending side
// and {flag poll, vote, records} on the receiving side: no fence instruction anywhere (MI355X_MICROARCH.md, visibility).
// ------------------------------------------------------------------------------------------------
typedef unsigned int sf_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double4 gs_load(__amdgpu_buffer_rsrc_t rs, unsigned byte_off)
{
  constexpr int kSystem = 17;   // aux bits of the buffer instructions: sc0 | sc1
  const sf_u4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, kSystem);
  const sf_u4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off + 16, 0, kSystem);
  return {__hiloint2double((int)a.y, (int)a.x), __hiloint2double((int)a.w, (int)a.z),
          __hiloint2double((int)b.y, (int)b.x), __hiloint2double((int)b.w, (int)b.z)};
}
__device__ __forceinline__ void gs_store(double4* p, double a, double b, double c, double d)
{
  double* q = reinterpret_cast<double*>(p);
  __hip_atomic_store(q, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(q + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(q + 2, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(q + 3, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// (flag words are 32-bit launch numbers compared by their difference, so the count may wrap)
__device__ __forceinline__ bool gs_behind(int flag, int seq) { return (int)((unsigned)flag - (unsigned)seq) < 0; }
// The gate of a wave: every rank's flag must have reached the number of this launch (its records and its vote are then in
// place), bounded by GsSync::max_ticks (F_HALO_TIMEOUT: an error of the step, never a hang).  Returns false when the wave
// must not run: a rank voted for a rebuild at an earlier sub-step (the vote is folded into this rank's trigger word, so
// later launches exit at their first test), or the wait ran out.  Wave-level: the k-th ACTIVE lane asks rank k.
__device__ __forceinline__ bool gs_gate(const GsSync* Y, int* flags, const int seq, const int kstep)
{
  const int W = Y->world, me = Y->rank;
  const int lane = threadIdx.x & 63;
  const unsigned long long act = __ballot(1);
  const int nact = __popcll(act), pos = __popcll(act & ((1ull << lane) - 1ull));
  int vote = INT_MAX;
  for (int base = 0; base < W; base += nact) {
    const int r = base + pos;
    const bool mine = r < W && r != me;
    int* line = Y->my_sync + kGsStride * (mine ? r : me);
    int f = seq;
    if (mine) f = __hip_atomic_load(line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (__ballot(gs_behind(f, seq))) {
      const long long t0 = wall_clock64();
      for (;;) {
        __builtin_amdgcn_s_sleep(2);
        if (mine) f = __hip_atomic_load(line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (!__ballot(gs_behind(f, seq))) break;
        if (wall_clock64() - t0 > Y->max_ticks) {
          if (mine && gs_behind(f, seq)) {
            flags[F_HALO_TIMEOUT_PEER] = r;
            flags[F_HALO_TIMEOUT_SEEN] = f;
            flags[F_HALO_TIMEOUT] = seq;
          }
          return false;
        }
      }
    }
    if (mine) vote = min(vote, __hip_atomic_load(line + 1 + (seq & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
  }
  // (no shuffle reduction: the lanes beyond the last atom are inactive and their registers hold anything)
  const bool stale = vote < kstep;
  if (__ballot(stale)) {
    if (stale) atomicMin(&flags[F_TRIGGER], vote);
    return false;
  }
  return true;
}
// this rank's vote and then its flag `seq` into every other rank's line (one wave; the caller has drained its stores)
__device__ __forceinline__ void gs_publish(const GsSync* Y, const int vote, const int seq)
{
  const int W = Y->world, me = Y->rank;
  const int lane = threadIdx.x & 63;
  const unsigned long long act = __ballot(1);
  const int nact = __popcll(act), pos = __popcll(act & ((1ull << lane) - 1ull));
  for (int base = 0; base < W; base += nact) {
    const int r = base + pos;
    if (r < W && r != me)
      __hip_atomic_store(Y->peer_sync[r] + kGsStride * me + 1 + (seq & 1), vote, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int base = 0; base < W; base += nact) {
    const int r = base + pos;
    if (r < W && r != me)
      __hip_atomic_store(Y->peer_sync[r] + kGsStride * me, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// The end of a workgroup (one wave): its border records and its trigger have left (vmcnt), it counts itself done on its
// XCD's line; the last one of an XCD counts the XCD, the last XCD tells every rank: vote first, then the flag.
// `expected`: workgroups of this XCD that run the kernel, `nxcd`: XCDs that have any.
__device__ __forceinline__ void gs_done(const DemPtrs& P, const StepParams& S, int expected, int nxcd)
{
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const int lane = threadIdx.x & 63;
  const unsigned long long act = __ballot(1);
  const int first = __ffsll((long long)act) - 1;
  int last = 0;
  if (lane == first) {
    int* c = P.gs_count + 32 * (int)(blockIdx.x & 7);
    if (__hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == expected) {
      __hip_atomic_store(c, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int* t = P.gs_count + 32 * 8;
      if (__hip_atomic_fetch_add(t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == nxcd) {
        __hip_atomic_store(t, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = 1;
      }
    }
  }
  if (!__ballot(last)) return;
  // (the trigger word as the memory side holds it: a read-modify-write, not a load some cache could answer)
  int vote = 0;
  if (lane == first) vote = atomicMin(&P.flags[F_TRIGGER], INT_MAX);
  vote = __shfl(vote, first, 64);
  gs_publish(P.gs_sync, vote, S.gs_seq + 1);
}


// a rank that owns no atom launches no sub-step kernel: its part of the hand-off alone (the gate, then vote and flag)
__global__ __launch_bounds__(64) static void k_gs_idle(const GsSync* Y, int* flags, int seq, int kstep, int publish)
{
  if (__atomic_load_n(&flags[F_TRIGGER], __ATOMIC_RELAXED) < kstep) return;
  if (__atomic_load_n(&flags[F_HALO_TIMEOUT], __ATOMIC_RELAXED) != 0) return;
  if (!gs_gate(Y, flags, seq, kstep)) return;
  if (publish) gs_publish(Y, INT_MAX, seq + 1);
}

}  // namespace sf
