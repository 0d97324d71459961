// sf_dem_gs.h -- ghost slots: the device side of the forward halo that needs no kernel between two sub-step kernels
// (sub-step kernel: sf_dem_kernels.h; stand-alone pack / flag / piece-end kernels: sf_dem_halo.hip; driver: sf_halo_rccl.hip)
#pragma once
#include <climits>

#include "sf_dem.h"

namespace sf {

// ------------------------------------------------------------------------------------------------
// ghost slots (decomposed domain, template parameter GS of the sub-step kernel): the neighbours' sub-step kernels write the
// records of this rank's ghosts straight into the ghost range of its record arrays xr / vm / om (IPC mappings of the
// arrays themselves), so nothing stands between two sub-step kernels and the gathers of the sub-step kernel are the same
// instructions for owned atoms and ghosts.  The writer uses write-through system-scope stores; the hand-off is
// {records, s_waitcnt vmcnt(0), completion count, the flag | vote word} on the sending side and {the word, gathers} on the
// receiving side.  The reader needs no fence: a kernel starts with its caches invalidated, and no wave touches a cache
// line that holds ghost records before its gate has seen every flag (the waves whose OWN records share a line with the
// first ghosts pass the gate before their first load).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void gs_store(double4* p, double a, double b, double c, double d)
{
#ifdef SF_GS_EXP_PLAIN_STORE   // (pricing arm of tests/ab_gs_arms.sh: not a correct hand-off)
  *p = double4{a, b, c, d};
  return;
#endif
  // two 16-byte write-through system-scope stores per record (`global_store_dwordx4 ... sc0 sc1`; HIP has no builtin for a
  // 16-byte store with scope bits: four 8-byte atomic stores cost 1 us more per sub-step on the 126 k brick).  The
  // compiler does not count them in vmcnt: the hand-off drains them with its own `s_waitcnt vmcnt(0)` (gs_done / the end
  // of the pack kernel), nothing else depends on them.  `s_nop 1`: the store has read its data registers before the next
  // instruction may overwrite them (cdna_hip_programming.md, asm stores).
  typedef double sf_d2 __attribute__((ext_vector_type(2)));
  double* q = reinterpret_cast<double*>(p);
  const sf_d2 lo = {a, b}, hi = {c, d};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(q), "v"(lo) : "memory");
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(q + 2), "v"(hi) : "memory");
}
// A rank's line in another rank's sync area holds ONE 64-bit word: (flag << 32) | vote -- the number of the launch this
// rank's records are in place for, and its vote for that launch (the sub-step index of its trigger, INT_MAX: none) --
// written by ONE store and read by ONE load: nothing to order.  A flag AHEAD of the launch a reader is in means "no vote":
// the sender has run the reader's launch number itself, which a trigger of its own would have stopped at the first test.
// (32-bit launch numbers compared by their difference, so the count may wrap)
__device__ __forceinline__ bool gs_behind(int flag, int seq) { return (int)((unsigned)flag - (unsigned)seq) < 0; }
__device__ __forceinline__ unsigned long long gs_word(int flag, int vote)
{
  return ((unsigned long long)(unsigned)flag << 32) | (unsigned long long)(unsigned)vote;
}
__device__ __forceinline__ int gs_flag_of(unsigned long long w) { return (int)(unsigned)(w >> 32); }
__device__ __forceinline__ int gs_vote_of(unsigned long long w, int seq)
{
  return gs_flag_of(w) == seq ? (int)(unsigned)(w & 0xffffffffull) : INT_MAX;
}
__device__ __forceinline__ unsigned long long gs_read_line(int* line)
{
  return __hip_atomic_load(reinterpret_cast<unsigned long long*>(line), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
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
    unsigned long long w = gs_word(seq, INT_MAX);
    if (mine) w = gs_read_line(line);
    if (__ballot(gs_behind(gs_flag_of(w), seq))) {
      const long long t0 = wall_clock64();
      for (;;) {
        __builtin_amdgcn_s_sleep(2);
        if (mine) w = gs_read_line(line);
        if (!__ballot(gs_behind(gs_flag_of(w), seq))) break;
        if (wall_clock64() - t0 > Y->max_ticks) {
          if (mine && gs_behind(gs_flag_of(w), seq)) {
            flags[F_HALO_TIMEOUT_PEER] = r;
            flags[F_HALO_TIMEOUT_SEEN] = gs_flag_of(w);
            flags[F_HALO_TIMEOUT] = seq;
          }
          return false;
        }
      }
    }
    if (mine) vote = min(vote, gs_vote_of(w, seq));
  }
  // (no shuffle reduction: the lanes beyond the last atom are inactive and their registers hold anything)
  const bool stale = vote < kstep;
  if (__ballot(stale)) {
    if (stale) atomicMin(&flags[F_TRIGGER], vote);
    return false;
  }
  return true;
}
// this rank's flag `seq` and its vote for that launch into every other rank's line (one wave; the caller has drained the
// record stores the flag stands for)
__device__ __forceinline__ void gs_publish(const GsSync* Y, const int vote, const int seq)
{
  const int W = Y->world, me = Y->rank;
  const int lane = threadIdx.x & 63;
  const unsigned long long act = __ballot(1);
  const int nact = __popcll(act), pos = __popcll(act & ((1ull << lane) - 1ull));
  for (int base = 0; base < W; base += nact) {
    const int r = base + pos;
    if (r < W && r != me)
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(Y->peer_sync[r] + kGsStride * me), gs_word(seq, vote),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// The end of a workgroup (one wave).  A wave that wrote border records waits until those have left (vmcnt), then every
// wave counts itself done on its XCD's line with ONE fire-and-forget 64-bit atomic: +1, and +2^32 when one of its atoms
// moved beyond skin / 2 (this launch's vote travels with the count: no second word to order, nothing to wait for).  ONE
// wave of the launch (`poller`: the first workgroup of the first XCD that has any) stays behind, watches the eight counters
// until every XCD has counted all of its workgroups, clears them and tells every rank (one flag | vote word each).
// `expected_lane`: workgroups of XCD x that run the kernel (the poller's lane x < 8 holds it).
__device__ __forceinline__ void gs_done(const DemPtrs& P, const StepParams& S, const bool wrote, const bool triggered,
                                        const bool poller, const int expected_lane)
{
#ifdef SF_GS_EXP_NODONE
  return;
#endif
#ifndef SF_GS_EXP_NOWAIT
  if (__ballot(wrote)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  const int lane = threadIdx.x & 63;
  const unsigned long long trig = __ballot(triggered) ? (1ull << 32) : 0ull;
  unsigned long long* cnt = reinterpret_cast<unsigned long long*>(P.gs_count);
  if (lane == 0)
    (void)__hip_atomic_fetch_add(cnt + 16 * (int)(blockIdx.x & 7), 1ull + trig, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (!poller) return;
  // (a full wave: in the ghost-slot kernel the lanes beyond the last atom do not return before the hand-off)
  const GsSync* Y = P.gs_sync;
  const long long t0 = wall_clock64();
  unsigned long long c = 0;
  for (;;) {
    c = (unsigned long long)(unsigned)expected_lane;
    if (lane < 8) c = __hip_atomic_load(cnt + 16 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!__ballot(lane < 8 && (unsigned)c != (unsigned)expected_lane)) break;
    if (wall_clock64() - t0 > Y->max_ticks) {
      if (lane == 0) {
        P.flags[F_HALO_TIMEOUT_PEER] = Y->rank;   // (this rank's own launch did not complete)
        P.flags[F_HALO_TIMEOUT_SEEN] = -1;
        P.flags[F_HALO_TIMEOUT] = S.gs_seq;
      }
      return;
    }
    __builtin_amdgcn_s_sleep(2);
  }
  if (lane < 8) __hip_atomic_store(cnt + 16 * lane, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // this rank's vote: a trigger of an EARLIER sub-step would have stopped this launch at its first test, so the vote is
  // this sub-step's index when any wave counted a trigger, "none" otherwise
  const int vote = __ballot(lane < 8 && (c >> 32) != 0) ? S.kstep + S.trig_add : INT_MAX;
  gs_publish(Y, vote, S.gs_seq + 1);
}

// a rank that owns no atom launches no sub-step kernel: its part of the hand-off alone (the gate, then its word)
__global__ __launch_bounds__(64) static void k_gs_idle(const GsSync* Y, int* flags, int seq, int kstep, int publish)
{
  if (__atomic_load_n(&flags[F_TRIGGER], __ATOMIC_RELAXED) < kstep) return;
  if (__atomic_load_n(&flags[F_HALO_TIMEOUT], __ATOMIC_RELAXED) != 0) return;
  if (!gs_gate(Y, flags, seq, kstep)) return;
  if (publish) gs_publish(Y, INT_MAX, seq + 1);   // (no atom: no trigger of its own)
}

}  // namespace sf
