// sf_dem_variants.h -- instrumentation of VARIANT builds of the sub-step kernel (tests/build_variant.sh defines
// SF_VARIANT_BUILD; the shipped library never carries any of this: every macro below expands to nothing there).
//   -DSF_EXP_STAMP=1   every workgroup of one chosen launch records {XCC_ID, HW_ID, start, end} (100 MHz clock): the
//                      fill / drain timeline per XCD (tests/exp_stamp.sh, tests/micro/stamp_timeline.py; valid results)
//   -DSF_EXP_PHASE=1   with it: lane 0 of every wave also keeps the clock at the phases of its life (entry, every slot of
//                      the neighbour loop, fixes, stores): [32 * workgroups] after the stamps
#pragma once
#if (defined(SF_EXP_STAMP) || defined(SF_EXP_PHASE)) && !defined(SF_VARIANT_BUILD)
#error "SF_EXP_* instrumentation: variant builds only (tests/build_variant.sh)"
#endif
#ifndef SF_EXP_STAMP
#define SF_EXP_STAMP 0
#endif
#ifndef SF_EXP_PHASE
#define SF_EXP_PHASE 0
#endif

namespace sf {

#if SF_EXP_STAMP
__device__ unsigned long long* g_stamp = nullptr;   // [4 * workgroups] of the launch being recorded, else null
#endif
#if SF_EXP_PHASE
// (kept in LDS while the wave runs -- a global store per mark would sit in the in-order vmcnt queue of the loads it is
// meant to observe: measured +25 % -- and copied out by the StampEnd destructor)
__device__ __forceinline__ unsigned long long* sf_phase_slots()
{
  __shared__ unsigned long long ph[32];
  return ph;
}
#define SF_PH(k)                                                                                              \
  do {                                                                                                        \
    if ((threadIdx.x & 63) == 0) sf_phase_slots()[(k)] = (unsigned long long)wall_clock64();                  \
  } while (0)
#else
#define SF_PH(k) do { } while (0)
#endif

#if SF_EXP_STAMP
struct StampEnd {
  unsigned long long* s;
  unsigned long long t0;
  __device__ ~StampEnd()
  {
    if (!s) return;
    __builtin_amdgcn_s_waitcnt(0);   // (the stores of this wave have been issued and acknowledged)
#if SF_EXP_PHASE
    if (threadIdx.x < 32) s[4 * (size_t)gridDim.x + 32 * (size_t)blockIdx.x + threadIdx.x] = sf_phase_slots()[threadIdx.x];
#endif
    if (threadIdx.x == 0) {
      unsigned long long* q = s + 4 * (size_t)blockIdx.x;
      q[0] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);     // HW_REG_XCC_ID[3:0]
      q[1] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);     // HW_REG_HW_ID
      q[2] = t0;
      q[3] = wall_clock64();
    }
  }
};
#if SF_EXP_PHASE
#define SF_STAMP_CLEAR_PHASES() do { if (threadIdx.x < 32) sf_phase_slots()[threadIdx.x] = 0ull; } while (0)
#else
#define SF_STAMP_CLEAR_PHASES() do { } while (0)
#endif
#define SF_STAMP_WORKGROUP()                                            \
  unsigned long long* const stamp_ = g_stamp;                           \
  const unsigned long long stamp_t0_ = stamp_ ? wall_clock64() : 0ull;  \
  SF_STAMP_CLEAR_PHASES();                                              \
  StampEnd stamp_end_{stamp_, stamp_t0_}
#else
#define SF_STAMP_WORKGROUP() do { } while (0)
#endif

}  // namespace sf
