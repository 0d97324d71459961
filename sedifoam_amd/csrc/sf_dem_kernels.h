// sf_dem_kernels.h -- the sub-step kernel family of the DEM engine (gfx950, wave64, FP64).  Included by sf_dem.hip; the kernels
// of the neighbour rebuild live in sf_dem_rebuild.h, the marshalling kernels of the lammps_* surface in sf_dem_io.h.
//
// Hot kernel: k_substep = PairGranHertzFixHistory::compute (pair_gran_hertzFix_history.cpp:45-287)
//   + FixCohe::post_force (fix_cohesive.cpp:138-263) + PairLubricatePoly::compute
//   (pair_lubricate_poly.cpp:194-407) + FixFluidDrag::post_force (fix_fluid_drag.cpp:114-164)
//   + FixWallGranFix::post_force (fix_wall_granFix.cpp:247-345) + [3P] gravity and the two
//   nve/sphere half-kicks, for one owned atom per lane.
#pragma once
#include <climits>

#include "sf_dem.h"

// Compile-time variants of the sub-step kernel; the defaults are the measured best, the
// others are built next to the shipped library by tests/build_variant.sh for A/B runs.
#ifndef SF_UNROLL2
#define SF_UNROLL2 1          // neighbour loop unrolled by two: the prefetch ping-pongs between two register sets
#endif
#ifndef SF_NT
#define SF_NT 1               // read-once rows as non-temporal loads
#endif
#ifndef SF_NT_ST
#define SF_NT_ST SF_NT        // write-once rows as non-temporal stores
#endif
#ifndef SF_NT_OUT
#define SF_NT_OUT 0           // output records non-temporal too (slower: the next sub-step gathers them)
#endif
#ifndef SF_ST_SHUFFLE
#define SF_ST_SHUFFLE 1       // output records exchanged between lanes so that every store covers whole cache lines
#endif
// (whether v, omega of a neighbour are prefetched always or only when the pair touched one sub-step ago is the template
// parameter TP of k_substep, chosen per list from the fraction of listed neighbours that touch)
#ifndef SF_HIST_PREFETCH
#define SF_HIST_PREFETCH 1    // the history of slot s+1 is requested with the records of slot s+1, one contact evaluation ahead
#endif
#ifndef SF_PERS_PREFETCH
#define SF_PERS_PREFETCH 1    // k_substep_persist: the next tile's records are requested under the current tile's epilogue
#endif
#ifndef SF_PERS_DYNAMIC
#define SF_PERS_DYNAMIC 1     // k_substep_persist: tiles beyond a wave's first two come from the XCD's head word (0: by position)
#endif
#ifndef SF_COOP_GATHER
#define SF_COOP_GATHER 1      // a neighbour's 32-byte record is read by the two lanes l, l + 32 together (see coop_merge)
#endif
// Instrumentation of variant builds (tests/build_variant.sh): SF_EXP_STAMP / SF_EXP_PHASE, the workgroup timeline of one
// launch -- sf_dem_variants.h.  The pricing arms of rounds 1-4 (history traffic off, agent-coherent access forms, LDS-DMA
// row streams, the no-wait persistent kernel, neighbour records by lane shuffle, the two-lane launch tail) were measured,
// written up (docs/history_r01_r03.md, profiles/r04_README.md, profiles/r05_README.md) and removed from this file.
#include "sf_dem_variants.h"
#include "sf_dem_gs.h"

namespace sf {

__device__ __forceinline__ Vec3 v3(const double4& a) { return {a.x, a.y, a.z}; }

// one lane per atom with exactly one of the cohesive / lubrication arms: the lean loop at three waves per SIMD (see
// substep_particle); the kernels of small systems (several lanes per atom) and the one with both arms stay as they were
#ifndef SF_LEAN_VARIANTS
#define SF_LEAN_VARIANTS 1
#endif
constexpr bool sf_lean_variant(bool cohe, bool lub, int lpa) { return SF_LEAN_VARIANTS && lpa == 1 && cohe != lub; }

// Streamed (read-once / write-once per sub-step) rows can be marked non-temporal so that they do not evict the
// neighbour records the gathers want to find again in the 32 KB vector L1 and the 4 MB L2 of the XCD.  Whether that
// pays depends on what the 256 MB memory-side cache can keep from one sub-step to the next -- the template parameter
// NTP of the sub-step kernels, chosen by DemEngine::choose_kernel from the bytes a sub-step touches:
//   0  nothing non-temporal: the whole state fits the memory-side cache and is found there by the next sub-step
//      (measured against policy 2: 100 k grains 33 -> 28 us, 200 k 56 -> 44, 300 k 70 -> 55, 400 k 91 -> 81);
//   3  loads non-temporal except the history, stores plain: the state almost fits (500 k: 103 -> 96 us);
//   1  rows and stores non-temporal, history loads plain: one history copy per contact, read twice per sub-step (by
//      the owner and by the partner side) -- the second reader finds the line in L2 (-10 % L2 misses; 1 M grains
//      201 -> 196 us, 2 M 393 -> 382, the settled disordered bed 215 -> 203);
//   2  everything non-temporal: two copies per contact, every history row is read once (the loose bed: 232 us, 242
//      with plain history loads).
template <bool NT, class T>
__device__ __forceinline__ T ld_stream(const T* p)
{
  if (NT && SF_NT) return __builtin_nontemporal_load(p);
  return *p;
}
template <bool NT, class T>
__device__ __forceinline__ void st_stream(T* p, T v)
{
  if (NT && SF_NT_ST) __builtin_nontemporal_store(v, p);
  else *p = v;
}
template <bool NT>
__device__ __forceinline__ void st_stream4(double4* p, double4 v)
{
  if (NT && SF_NT) {
    typedef double d4v __attribute__((ext_vector_type(4)));
    d4v t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<d4v*>(p));
  } else {
    *p = v;
  }
}

// ------------------------------------------------------------------------------------------------
// fused DEM sub-step
// ------------------------------------------------------------------------------------------------
// One owned atom: neighbour loop + post_force fixes + integration.  LDS = false: the neighbour's records are
// gathered from HBM/L2 by global index; LDS = true: from the tile's staged copy in LDS (k_substep_lds).
// LPA lanes per atom (1, 2 or 4): lane q of an atom's group handles the slots q, q + LPA, ...; the partial force and
// torque sums are combined with a fixed shuffle tree and lane 0 integrates.  Small systems (< ~3 waves per SIMD at
// one lane per atom) are bound by the latency of one lane's 12 dependent neighbour iterations, not by bandwidth.
// GS: ghost slots (sf_dem_gs.h).  Returns 0 when the wave stopped at the gate (nothing was stored), else 1 | 2 when the atom
// may have written border records (what the hand-off must wait for) | 4 when it moved beyond skin / 2 (its vote).
// ------------------------------------------------------------------------------------------------
// Half-wave gather.  The texture addresser prices a vector load by the 128-byte lines it touches (tests/micro/gather_bench:
// 2.2 cycles per line + ~5), and a record gather as "16 bytes per lane, twice" touches every neighbour's line twice: 37.8
// cycles per instruction on consecutive records, 141 on scattered ones.  Here instruction A reads the records of the
// neighbours of lanes 0..31 -- lane l their first 16 bytes, lane l + 32 their second 16 bytes -- and instruction B those of
// lanes 32..63: every line is touched once (22.7 cycles on consecutive records, half the lines on scattered ones).  One
// v_permlane32_swap per dword puts the halves where the arithmetic expects them: A' = first halves, B' = second halves of
// every lane's OWN neighbour.  Same bytes into the same operations in the same order (against the plain instantiation:
// the same bits after one sub-step, last bits in one pair in two thousand later -- tests/test_dem_gpu.py).
// ------------------------------------------------------------------------------------------------
// Which kernels gather that way -- measured (profiles/r05_README.md section 6): it pays where the launch is bound by the
// throughput of the memory pipe -- beds that do not fit the memory-side cache (non-temporal policies 1, 2: beyond ~650 k
// grains; the 1 M headline bed -2 to -3 %, 2 M -2 %), and most in the kernel with both the cohesive and the lubrication
// arm on a polydisperse bed, whose gathers scatter (500 k: -5 to -8 %).  Beds of a few rounds of resident waves are bound
// by the length of ONE wave, which the exchanges lengthen (60 k - 250 k grains: 0 to +2 %, loose 126 k: +5 %); the kernels
// that request v, omega only of touching pairs (loose beds) gain nothing at any size, the lean kernels with one arm lose
// 1 - 1.5 %: all of those keep the plain gather.
__host__ __device__ constexpr bool sf_coop_variant(bool cohe, bool lub, int lpa, bool tp, int ntp)
{
  return SF_COOP_GATHER && lpa == 1 && (ntp == 1 || ntp == 2) && ((cohe && lub) || (!cohe && !lub && !tp));
}
__device__ __forceinline__ void swap32(unsigned& a, unsigned& b)
{
  typedef unsigned sf_u2 __attribute__((ext_vector_type(2)));
  const sf_u2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);   // a[32..63] <-> b[0..31]
  a = r.x;
  b = r.y;
}
__device__ __forceinline__ void swap32(double& a, double& b)
{
  unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
  unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
  swap32(alo, blo);
  swap32(ahi, bhi);
  a = __hiloint2double((int)ahi, (int)alo);
  b = __hiloint2double((int)bhi, (int)blo);
}
// what the two cooperative loads of a record returned (.x .y: instruction A, .z .w: instruction B) -> the lane's record
__device__ __forceinline__ double4 coop_merge(double4 r)
{
  swap32(r.x, r.z);
  swap32(r.y, r.w);
  return r;
}
// lane l < 32: (j(l), j(l + 32)); lane l + 32: the same pair -- and whether v, omega of the two are wanted (bit 31 travels
// with the index)
__device__ __forceinline__ void coop_indices(const int j, const bool vw, int& ja, int& jb, bool& vwa, bool& vwb)
{
  unsigned a = (unsigned)j | (vw ? 0x80000000u : 0u), b = a;
  swap32(a, b);
  ja = (int)(a & (unsigned)kIdxMask);   // (tells the compiler what it knew about j: record offsets fit 32 bits)
  jb = (int)(b & (unsigned)kIdxMask);
  vwa = (int)a < 0;
  vwb = (int)b < 0;
}
// (half: byte offset of the lane's half of a record -- 0 in lanes 0..31, 16 in lanes 32..63)
__device__ __forceinline__ double4 coop_load(const double4* arr, const int ja, const int jb, const int half,
                                             const bool wa = true, const bool wb = true)
{
  const char* p = reinterpret_cast<const char*>(arr);
  double2 a = {0.0, 0.0}, b = {0.0, 0.0};
  if (wa) a = *reinterpret_cast<const double2*>(p + (((unsigned)ja << 5) | (unsigned)half));
  if (wb) b = *reinterpret_cast<const double2*>(p + (((unsigned)jb << 5) | (unsigned)half));
  return double4{a.x, a.y, b.x, b.y};
}

// Persistent tiles (k_substep_persist, PERS): what a wave reads about its atom before it can start the neighbour loop -- the
// atom's three records, its row count and its first two list words.  The persistent kernel requests these for the wave's NEXT
// tile when the neighbour loop of the current tile is over (the loop's register sets are dead by then), so that they arrive
// while the current tile's fixes, integration and stores run.
struct TilePre {
  double4 x, v, w;
  int nn_all, w_first, w_second;
  int q;        // lane 0: what the XCD's head word returned (the wave's tile after the next one is 2 x resident + q)
  int* head;
  const DemPtrs* Pe;       // the kernel arguments as the epilogue reads them (see substep_particle)
  const StepParams* Se;
};
template <bool NT_LD>
__device__ __forceinline__ void tile_prefetch(const DemPtrs& P, const StepParams& S, const int i, TilePre& t)
{
  const size_t cap = (size_t)S.cap;
  t.x = P.xr_in[i];
  t.v = P.vm_in[i];
  t.w = P.om_in[i];
  t.nn_all = ld_stream<NT_LD>(&P.numneigh[i]);
  t.w_first = ld_stream<NT_LD>(&P.neigh[i]);
  t.w_second = ld_stream<NT_LD>(&(P.neigh + (S.nslots > 1 ? cap : 0))[i]);
}

template <int STYLE, bool COHE, bool LUB, bool LDS, int LPA, bool TP, int NTP, bool GS = false, bool PERS = false>
__device__ __forceinline__ int substep_particle(const DemPtrs& P, const StepParams& S, const int i, const int q,
                                                 const double4* lx, const double4* lv, const double* lw,
                                                 const unsigned long long gs_w = 0ull, const bool live = true,
                                                 TilePre* pre = nullptr, const int i_next = -1)
{
  // COOP: the records of a neighbour are read by lane pairs (l, l + 32) -- every lane of the wave walks the neighbour loop
  // to the wave's largest count, `live` = this lane holds an atom (the caller clamps i of the others to a valid one: they
  // load, take part in the exchanges, and store nothing)
  constexpr bool COOP = !LDS && sf_coop_variant(COHE, LUB, LPA, TP, NTP);
  const int coop_half = (int)(threadIdx.x & 32) >> 1;   // (byte offset of this lane's half of a record: 0 | 16)
  const size_t cap = (size_t)S.cap;
  const bool shearupdate = (S.mode != 2);
  // the gathering kernel always runs on (root, image code) words, the LDS-staged one on plain indices: a compile-time
  // fact lets the compiler see that a gathered index has 25 bits, i.e. that record offsets fit 32 bits (scalar base +
  // 32-bit offset addressing instead of 64-bit address arithmetic in vector registers)
  constexpr int ROOTS = LDS ? 0 : 1;
  constexpr bool NT_LD = NTP != 0, NT_ST = NTP == 1 || NTP == 2, NT_HIST = NTP == 2;
  __builtin_assume(i >= 0 && i < (1 << kIdxBits));

  SF_PH(0);
  // (ghost slots: the gate normally waits behind the wave's own rows -- one round trip for both.  The wave whose own
  // records share a 128-byte line with the first ghost records must not pull that line in before the ghosts are there)
  bool gs_gated = false;
#ifdef SF_GS_EXP_NOGATE   // (pricing arm of tests/ab_gs_arms.sh: not a correct hand-off)
  gs_gated = true;
#endif
  if (GS && S.gs_wait && !gs_gated && __ballot(i >= (S.nlocal & ~3))) {
    if (!gs_gate(P.gs_sync, P.flags, S.gs_seq, S.kstep)) return 0;
    gs_gated = true;
  }
  // (PERS: requested one tile ago, see TilePre)
  const double4 xi4 = PERS ? pre->x : P.xr_in[i];   // also a gather target of the neighbours: keep it cached
  const double4 vi4 = PERS ? pre->v : P.vm_in[i];
  const double4 wi4 = PERS ? pre->w : P.om_in[i];
  const Vec3 xi = v3(xi4), vi = v3(vi4), wi = v3(wi4);
  const double radi = xi4.w, mi = vi4.w;

  Vec3 F = {0.0, 0.0, 0.0}, T = {0.0, 0.0, 0.0};
  const int mk = S.use_groups ? P.mask[i] : 1;   // group bits of this atom (bit 0 = all)
  const int nn_all = PERS ? pre->nn_all : ld_stream<NT_LD>(&P.numneigh[i]);
  const int nn = (COOP || PERS) ? (live ? nn_all : 0)   // (PERS: the lanes past the last atom stay for the wave's other tiles)
                      : LPA == 1 ? nn_all : (nn_all > q ? (nn_all - q + LPA - 1) / LPA : 0);   // slots of this lane
  const double lub_cutsq = S.lub.cut_global * S.lub.cut_global;

  // Latency structure of one slot: index -> gather of the neighbour's three records -> contact law.
  // Software pipeline: the index of slot s+2 and the x, v, omega records of slot s+1 are requested before the
  // contact of slot s is evaluated, so the next neighbour's gather latency hides behind ~230 FP64 instructions
  // (4 waves per SIMD cannot hide it alone).  The shear loads of slot s are issued BEFORE that prefetch: vmcnt
  // retires in order, so the contact law only waits for them and the prefetch stays in flight.  Measured at 1 M
  // atoms: 249 -> 224 us.  Prefetching the shear history as well costs 10 VGPRs (3 waves/SIMD) and loses 10 %.
  constexpr bool NEED_VW = (STYLE != 0) || LUB;
  // The kernels that carry ONE of the cohesive / lubrication arms run leaner -- one register set (no unroll by two), the
  // history requested where it is consumed -- and fit three waves per SIMD that way: measured on the 500 k polydisperse
  // bed against the two-wave form, cohesive 134.0 -> 120.6 us, lubricate/poly 176.8 -> 164.7 us; the kernel with both arms
  // spills at three waves and gains nothing (192.3 / 198.7 -> 191.9 us), it keeps two (profiles/r05_c5_README.md)
  constexpr bool LEAN = sf_lean_variant(COHE, LUB, LPA);
  constexpr bool HIST_PF = SF_HIST_PREFETCH && !LEAN;
  struct Rec {
    double4 x, v, w;
    Vec3 sh; // the pair's history as THIS side sees it (SF_HIST_PREFETCH)
    int l;   // LDS: position of the neighbour in the staged tile
    bool vw; // v and w were requested
  };
  // history of the pair in list slot `slotrow` whose word is jraw: the owner reads its own row (coalesced), the
  // partner the owner's row, the pair seen from the other side
  auto load_history = [&](const int jraw, const int slotrow, Vec3& sh) {
    sh = {0.0, 0.0, 0.0};
    if (STYLE == 0 || !(jraw & kTouchBit)) return;
    const bool own = (jraw & kOwnBit) != 0;
    auto ldh = [&](const double* p) { return ld_stream<NT_HIST>(p); };
    if (own) {
      const double* const hin = P.shear_in + (size_t)(3 * slotrow) * cap;
      sh.x = ldh(&hin[i]);
      sh.y = ldh(&(hin + cap)[i]);
      sh.z = ldh(&(hin + 2 * cap)[i]);
    } else {
      // (the owner's value, NOT yet negated: flipping the sign here would make the wave wait for these loads on the
      // spot -- the prefetch of the next slot's history would be no prefetch at all; the consumer flips it)
      const double* src = P.shear_in + (size_t)(3 * ((jraw >> kIdxBits) & 31)) * cap + (size_t)(jraw & kIdxMask);
      sh.x = ldh(src);
      sh.y = ldh(src + cap);
      sh.z = ldh(src + 2 * cap);
      if (LPA > 1) sh = {-sh.x, -sh.y, -sh.z};   // (several lanes per atom: registers are tighter than time, flip at once)
    }
  };
  // The contact law needs the neighbour's v and omega only if the pair touches, and a pair that touches now almost
  // always touched one sub-step ago (its touch bit): for the others only x is gathered, and the rare new contact loads
  // v and omega on demand.  A settled bed lists about twice as many neighbours as it has contacts.
  auto wants_vw = [&](const int jraw) {
    return NEED_VW && (LUB || !TP || (jraw & kTouchBit) != 0);
  };
  // request the records of the neighbour in `slot` (global gather), or its LDS position
  auto fetch = [&](int jraw, int slotrow, Rec& R) {
    if (LDS) {
      R.l = (P.nloc + (size_t)slotrow * cap)[i];
    } else {
      const int j = neigh_index(jraw, ROOTS);
      R.l = j;   // (gather mode: the root index, used by the register reuse below)
      R.vw = wants_vw(jraw);
      if (COOP) {   // (every lane of the wave is here; merged by the consumer, coop_merge)
        int ja, jb;
        bool vwa, vwb;
        coop_indices(j, R.vw, ja, jb, vwa, vwb);
        R.x = coop_load(P.xr_in, ja, jb, coop_half);
        if (NEED_VW) {
          if (TP) {   // (v, omega only of the neighbours whose pair touched one sub-step ago: both lanes of a pair know)
            R.v = coop_load(P.vm_in, ja, jb, coop_half, vwa, vwb);
            R.w = coop_load(P.om_in, ja, jb, coop_half, vwa, vwb);
          } else {
            R.v = coop_load(P.vm_in, ja, jb, coop_half);
            R.w = coop_load(P.om_in, ja, jb, coop_half);
          }
        }
      } else {
        R.x = P.xr_in[j];
        if (R.vw) {
          R.v = P.vm_in[j];
          R.w = P.om_in[j];
        }
      }
    }
    if (HIST_PF) load_history(jraw, slotrow, R.sh);
  };
  // COOP: the halves of a prefetched record are exchanged at the END of the slot that requested it (the contact evaluation
  // in between hides the gather), so the register set that crosses into the next slot holds plain records
  auto coop_finish = [&](Rec& R) {
    if (!COOP) return;
    R.x = coop_merge(R.x);
    if (NEED_VW) {
      R.v = coop_merge(R.v);
      R.w = coop_merge(R.w);
    }
  };
  // rows of the slot-major arrays are addressed as (row pointer)[i]: with one lane per atom the slot -- hence the row
  // pointer -- is wave-uniform (scalar registers), the element offset 32 bits
  // (the first two words are requested whatever the count says -- rows q and q + LPA exist -- so that they travel
  // together with numneigh and the atom's own records instead of one memory round trip behind them)
  const int row1 = q + LPA < S.nslots ? q + LPA : S.nslots - 1;
  // (one lane per atom only: with several lanes per atom the two extra live registers spill)
  const bool ld0 = LPA == 1 || nn > 0, ld1 = LPA == 1 || nn > 1;
  const int w_first = PERS ? pre->w_first : ld0 ? ld_stream<NT_LD>(&(P.neigh + (size_t)q * cap)[i]) : 0;
  const int w_second = PERS ? pre->w_second : ld1 ? ld_stream<NT_LD>(&(P.neigh + (size_t)row1 * cap)[i]) : 0;
  int jraw_n1 = nn > 0 ? w_first : 0;
  int jraw_n2 = nn > 1 ? w_second : 0;
  // One history copy per contact: a partner-side slot (kOwnBit clear) reads the owner's previous value from the
  // owner's slot; the five image-code bits of a partner-side word hold that slot (a partner-side neighbour is never
  // a periodic image, see k_back_slots).
  Rec RA, RB;
  RA.x = RA.v = RA.w = RB.x = RB.v = RB.w = double4{0, 0, 0, 0};
  RA.sh = RB.sh = Vec3{0.0, 0.0, 0.0};
  RA.l = RB.l = 0;
  RA.vw = RB.vw = false;
  // (ghost slots: the flags are asked for behind the wave's own rows -- one round trip for both -- and before anything is
  // stored or any ghost record is read)
  if (GS && S.gs_wait && !gs_gated) {
    // gs_w: the flag | vote word of rank `lane`, requested by the kernel before anything else (a full wave: lane r holds
    // rank r's).  All there and nobody voted: on; a flag missing: the bounded wait; a vote: fold it and stop.
    if (__ballot(1) == ~0ull && !__ballot(gs_behind(gs_flag_of(gs_w), S.gs_seq))) {
      const int gs_v = gs_vote_of(gs_w, S.gs_seq);
      const bool stale = gs_v < S.kstep;
      if (__ballot(stale)) {
        if (stale) atomicMin(&P.flags[F_TRIGGER], gs_v);
        return 0;
      }
    } else if (!gs_gate(P.gs_sync, P.flags, S.gs_seq, S.kstep)) {
      return 0;
    }
  }
  // COOP: every lane walks the neighbour loop to the largest count of the wave (all 64 lanes are here)
  int nn_wave = nn;
  if (COOP) {
    for (int off = 32; off > 0; off >>= 1) nn_wave = max(nn_wave, __shfl_xor(nn_wave, off, 64));
    nn_wave = __builtin_amdgcn_readfirstlane(nn_wave);
  }
  if (COOP ? nn_wave > 0 : nn > 0) {
    fetch(jraw_n1, q, RA);
    coop_finish(RA);
  }

  // one slot: `cur` holds the neighbour's records, `nxt` receives the prefetch of slot s+1
  auto slot_body = [&](const int s, const Rec& cur, Rec& nxt, const bool more, const bool uniform_s) {
    SF_PH(2 + (s < 24 ? s : 24));
    // the list slot this iteration works on; with one lane per atom and inside the (convergent) loop it is the same in
    // every active lane of the wave -- NOT in the tail call after the loop, where lanes with different neighbour
    // counts arrive with different s
    const int sl = (LPA == 1 && uniform_s) ? __builtin_amdgcn_readfirstlane(s) : q + LPA * s;
    int* const nrow = P.neigh + (size_t)sl * cap;                       // this slot's row of the list
    double* const hout = P.shear_out + (size_t)(3 * sl) * cap;
    const int jraw = jraw_n1;
    const bool own = (jraw & kOwnBit) != 0;
    Vec3 sh = cur.sh;
    if (!HIST_PF) load_history(jraw, sl, sh);
    // the pair seen from the partner's side (a pair that did not touch starts from +0.0 on both sides, as before)
    if (STYLE != 0 && LPA == 1 && !own && (jraw & kTouchBit)) sh = {-sh.x, -sh.y, -sh.z};
    jraw_n1 = jraw_n2;
    if (s + 2 < nn) jraw_n2 = ld_stream<NT_LD>(&(nrow + (size_t)(2 * LPA) * cap)[i]);
    else if (COOP) jraw_n2 = 0;   // (a lane beyond its count keeps loading for its partner: word 0 = atom 0, no bits)
    if (more) {
      bool reuse = false;
      if (!reuse) fetch(jraw_n1, sl + LPA, nxt);
    }
    double4 xj4 = cur.x, vj4 = cur.v, wj4 = cur.w;
    if (LDS) {
      xj4 = lx[cur.l];
      if (NEED_VW) {
        vj4 = lv[cur.l];
        wj4 = {lw[3 * cur.l], lw[3 * cur.l + 1], lw[3 * cur.l + 2], 0.0};
      }
    }
    if (ROOTS && own) {
      // periodic image of the root: the same x_root + shift the reference's forward_comm would have stored
      // (a partner-side word never refers to an image: its code bits hold the owner's slot)
      const int code = (jraw >> kIdxBits) & 31;
      if (code != kNoShift) {
        const int cz = code / 9, cy = (code - 9 * cz) / 3, cx = code - 9 * cz - 3 * cy;
        xj4.x += (double)(cx - 1) * S.prd[0];
        xj4.y += (double)(cy - 1) * S.prd[1];
        xj4.z += (double)(cz - 1) * S.prd[2];
      }
    }
    const Vec3 del = xi - v3(xj4);
    // (COOP: a lane beyond its own count is here for its partner's loads: no pair, nothing touches, nothing is stored)
    const double rsq = (COOP && s >= nn) ? 1.0e300 : dot(del, del);
    const double radj = xj4.w;
    const double radsum = radi + radj;

    // |del| and its reciprocal once per slot for the kernels that carry the cohesive / lubrication arms (each arm took its
    // own: three 19-instruction sequences per pair); the plain contact kernel takes them only for a touching pair
    double r_pair = 0.0, rinv_pair = 0.0;
    if (COHE || LUB) sf_sqrt_rsqrt(rsq, r_pair, rinv_pair);
    if (STYLE != 0) {
      if (rsq >= radsum * radsum) {
        // unset non-touching neighbours (:131-139); the stale shear is ignored once the bit is clear
        if (jraw & kTouchBit) nrow[i] = jraw & ~kTouchBit;
      } else {
        if (!LDS && !cur.vw) {   // a contact that did not exist one sub-step ago
          vj4 = P.vm_in[cur.l];
          wj4 = P.om_in[cur.l];
        }
        ContactIn c;
        c.del = del;
        c.rsq = rsq;
        if (COHE || LUB) {
          c.r = r_pair;
          c.rinv = rinv_pair;
        } else {
          sf_sqrt_rsqrt(rsq, c.r, c.rinv);
        }
        c.vr = vi - v3(vj4);
        c.wsum = {radi * wi.x + radj * wj4.x, radi * wi.y + radj * wj4.y, radi * wi.z + radj * wj4.z};
        const double mj = vj4.w;
        c.overlap = radsum - c.r;
#if SF_FAST_MATH
        // meff = mi mj/(mi+mj) and reff = overlap radi radj/radsum share one reciprocal
        const double msum = mi + mj;
        const double inv = sf_rcp(msum * radsum);
        c.meff = (mi * mj) * (radsum * inv);
        c.reff = c.overlap * ((radi * radj) * (msum * inv));
#else
        c.meff = mi * mj / (mi + mj);
        c.reff = (radsum - c.r) * radi * radj / radsum;
#endif
        if (S.freeze_bit) {   // pair_gran_hertzFix_history.cpp:188-189: a frozen partner is infinitely heavy
          if (wi4.w != 0.0) c.meff = mj;
          if (wj4.w != 0.0) c.meff = mi;
        }
        ContactOut o;
        gran_history_law<STYLE>(S.gran, S.dt, shearupdate, c, sh, o);
        if (own) {
          st_stream<NT_ST>(&hout[i], sh.x);
          st_stream<NT_ST>(&(hout + cap)[i], sh.y);
          st_stream<NT_ST>(&(hout + 2 * cap)[i], sh.z);
        }
        if (!(jraw & kTouchBit)) nrow[i] = jraw | kTouchBit;
        F = F + o.F;
        T = T - radi * o.tor;
      }
    }
    if (COHE) {
      // fix cohesive is skipped during setup (FixCohe::setup() never runs, fix_cohesive.cpp:117)
      const double rc = radsum + S.cohe.smax;
      if (S.mode != 2 && (mk & S.cohe_bit) && rsq < rc * rc) {   // fix_cohesive.cpp:167 group of i
        const double cc = cohesive_ccel(S.cohe, r_pair, radsum) * rinv_pair;
        F = F + Vec3{del.x * cc, del.y * cc, del.z * cc};
      }
    }
    if (LUB) {
      if (S.lub.flagHI && rsq < lub_cutsq) {
        lubricate_poly_pair(S.lub, del, r_pair, rinv_pair, radi, radj, vi, v3(vj4), wi, v3(wj4), F, T);
      }
    }
    if (more) coop_finish(nxt);
  };
  // unrolled by two so that the prefetch ping-pongs between RA and RB without register copies
  SF_PH(1);
  int s = 0;
  if constexpr (COOP && SF_UNROLL2 && !LEAN) {
    // (a counted loop in scalar registers: the largest count of the wave)
    for (; s + 1 < nn_wave; s += 2) {
      slot_body(s, RA, RB, true, true);
      slot_body(s + 1, RB, RA, s + 2 < nn_wave, true);
    }
    if (s < nn_wave) slot_body(s, RA, RB, false, true);
  } else if constexpr (COOP) {
    for (; s < nn_wave; s++) {
      slot_body(s, RA, RB, s + 1 < nn_wave, true);
      RA = RB;
    }
  } else if constexpr (SF_UNROLL2 && !LEAN) {
    for (; s + 1 < nn; s += 2) {
      slot_body(s, RA, RB, true, true);
      slot_body(s + 1, RB, RA, s + 2 < nn, true);
    }
    if (s < nn) slot_body(s, RA, RB, false, false);
  } else {
    for (; s < nn; s++) {
      slot_body(s, RA, RB, s + 1 < nn, true);
      RA = RB;
    }
  }
  SF_PH(27);
  if (COOP && !PERS && !live) return 1;   // (PERS: behind the prefetch of the next tile, below)
  if (LPA > 1) {
    // fixed tree: (q0 + q1) [+ (q2 + q3)] -- the same bits on every run
    for (int off = 1; off < LPA; off <<= 1) {
      F.x += __shfl_xor(F.x, off, 64); F.y += __shfl_xor(F.y, off, 64); F.z += __shfl_xor(F.z, off, 64);
      T.x += __shfl_xor(T.x, off, 64); T.y += __shfl_xor(T.y, off, 64); T.z += __shfl_xor(T.z, off, 64);
    }
    if (q != 0) return 1;
  }
  if (LUB) {
    if (S.lub.flagfld) {  // isotropic FLD terms, pair_lubricate_poly.cpp:213-220
      const double a = S.lub.vxmu2f * S.lub.R0 * radi;
      const double b = S.lub.vxmu2f * S.lub.RT0 * (radi * radi * radi);
      F = F - a * vi;
      T = T - b * wi;
    }
  }

  // (PERS: what follows reads the kernel arguments through a pointer the compiler cannot see through -- pre->ka, renewed per
  // tile -- so that the ~100 scalar words only the fixes and the integration need are loaded where they are used, as in
  // k_substep, instead of being kept live across the tile loop; the neighbour loop above keeps its scalars in registers)
  const DemPtrs& PE = PERS ? *pre->Pe : P;
  const StepParams& SE = PERS ? *pre->Se : S;
  // the rows the fixes and the integration read, requested TOGETHER here (one memory round trip instead of one per fix)
  const bool use_fd = SE.have_fdrag && (mk & SE.fdrag_bit);   // fix_fluid_drag.cpp:145
  Vec3 fd_in = {0.0, 0.0, 0.0}, xh_in = {0.0, 0.0, 0.0};
  unsigned wt_in = 0;
  if (SE.have_fdrag) {
    const size_t k = use_fd ? (size_t)i : 0;   // (a lane outside the group reads a valid element and drops it)
    fd_in = {PE.fdrag[k], PE.fdrag[cap + k], PE.fdrag[2 * cap + k]};
  }
  if (SE.mode == 0 && SE.have_nve)
    xh_in = {ld_stream<NT_LD>(&PE.xhold[i]), ld_stream<NT_LD>(&PE.xhold[cap + i]), ld_stream<NT_LD>(&PE.xhold[2 * cap + i])};
  if (SE.nwalls) wt_in = PE.wtouch[i];
  // (PERS: the next tile's records and first words, requested BEHIND this tile's fix rows -- memory returns in order, the
  // epilogue waits for its rows only -- and ahead of everything the epilogue computes and stores)
  // (i_next: 64 x the next tile -- wave-uniform, so that nothing per lane crosses the neighbour loop for it; -1: no next tile)
  if (PERS && SF_PERS_PREFETCH && i_next >= 0) {
    if (SF_PERS_DYNAMIC && (threadIdx.x & 63) == 0) pre->q = atomicAdd(pre->head, 1);
    const int in = i_next + (int)(threadIdx.x & 63);
    tile_prefetch<NT_LD>(P, S, in < S.nlocal ? in : 0, *pre);
  }
  if (PERS && !live) return 1;
  // (fused forward pack: the send slots of a border atom, requested here so that they have arrived by the end)
  int txk0 = -1, txk1 = -1;
  if (SE.tx_fused == 1 && (xi.x < SE.tx_xlo || xi.x >= SE.tx_xhi)) {
    txk0 = PE.sendslot[0][i];
    txk1 = PE.sendslot[1][i];
  }
  // (brick driver: near an external face in any dimension)
  const bool txb = SE.tx_fused == 2 && (xi.x < SE.tx_lo3[0] || xi.x >= SE.tx_hi3[0] || xi.y < SE.tx_lo3[1] ||
                                       xi.y >= SE.tx_hi3[1] || xi.z < SE.tx_lo3[2] || xi.z >= SE.tx_hi3[2]);

  // ---- post_force fixes: gravity -> fdrag -> walls; fix freeze zeroes what the fixes BEFORE it in the script (and
  // the pair styles) gave a frozen atom, the fixes after it still act ([3P] Modify::post_force runs them in script
  // order; the reference's bed cases have `fix 4 bottom freeze` followed by `fix ywall all wall/gran`).  Fa, Ta: what
  // the fixes after fix freeze add, in their order; a free atom sums everything in F, T as before ----
  Vec3 Fa = {0.0, 0.0, 0.0}, Ta = {0.0, 0.0, 0.0};
  const bool any_post = SE.freeze_bit != 0;   // (wave-uniform: no fix freeze, nothing to keep apart)
  if (SE.have_gravity && (mk & SE.grav_bit)) {
    const Vec3 g = {mi * SE.gacc[0], mi * SE.gacc[1], mi * SE.gacc[2]};
    F = F + g;
    if (any_post && (SE.post_freeze & 1)) Fa = Fa + g;
  }
  if (use_fd) {
    Vec3 fd = fd_in;
    if (SE.carrier_rho != 0.0) {
      const double rho = 3.0 * mi / (4.0 * kPiTypo * radi * radi * radi);
      const Vec3 vo = {PE.vOld[i], PE.vOld[cap + i], PE.vOld[2 * cap + i]};
      const Vec3 du = {PE.DuDt[i], PE.DuDt[cap + i], PE.DuDt[2 * cap + i]};
      const double k = SE.carrier_rho / rho * 0.5 * mi;
      fd.x += k * (du.x - (vi.x - vo.x) / SE.dt);
      fd.y += k * (du.y - (vi.y - vo.y) / SE.dt);
      fd.z += k * (du.z - (vi.z - vo.z) / SE.dt);
      PE.vOld[i] = vi.x;
      PE.vOld[cap + i] = vi.y;
      PE.vOld[2 * cap + i] = vi.z;
    }
    F = F + fd;
    if (any_post && (SE.post_freeze & 2)) Fa = Fa + fd;
  }
  if (SE.nwalls) {
    unsigned wt = wt_in, wt_new = 0;
    for (int w = 0; w < SE.nwalls; w++) {
      const WallParams& W = SE.wall[w];
      if (!(mk & W.bit)) continue;   // fix_wall_granFix.cpp:290
      Vec3 dw = {0.0, 0.0, 0.0};
      Vec3 vw = {W.vwall[0], W.vwall[1], W.vwall[2]};   // 0 unless the wall wiggles or shears (:255-264)
      if (W.dim < 3) {               // :294-308
        const double xc = (W.dim == 0) ? xi.x : (W.dim == 1) ? xi.y : xi.z;
        const double del1 = xc - W.lo, del2 = W.hi - xc;
        const double d = (del1 < del2) ? del1 : -del2;
        dw = {W.dim == 0 ? d : 0.0, W.dim == 1 ? d : 0.0, W.dim == 2 ? d : 0.0};
      } else {                       // z cylinder about the origin, :309-322
        const double delxy = sqrt(xi.x * xi.x + xi.y * xi.y);
        const double delr = W.cylradius - delxy;
        if (delr > radi) continue;   // (the reference sets dz = cylradius: no contact, shear reset)
        dw = {-delr / delxy * xi.x, -delr / delxy * xi.y, 0.0};
        if (W.vrot != 0.0) vw = {W.vrot * xi.y / delxy, -W.vrot * xi.x / delxy, 0.0};
      }
      const double rsq = dot(dw, dw);
      if (rsq > radi * radi) continue;   // shear reset = touch bit cleared
      ContactIn c;
      c.del = dw;
      c.rsq = rsq;
      c.r = sqrt(rsq);
      c.rinv = 1.0 / c.r;
      c.vr = vi - vw;
      c.wsum = radi * wi;
      c.meff = mi;
      c.overlap = radi - c.r;
      c.reff = (radi - c.r) * radi;
      const size_t wb = ((size_t)(3 * w)) * cap + i;
      Vec3 sh = {0.0, 0.0, 0.0};
      if (wt & (1u << w)) {
        sh.x = PE.wshear[wb];
        sh.y = PE.wshear[wb + cap];
        sh.z = PE.wshear[wb + 2 * cap];
      }
      ContactOut o;
      if (W.gp.style == 2) hertz_history_law(W.gp, SE.dt, shearupdate, c, sh, o);
      else hooke_history_law(W.gp, SE.dt, shearupdate, c, sh, o);
      PE.wshear[wb] = sh.x;
      PE.wshear[wb + cap] = sh.y;
      PE.wshear[wb + 2 * cap] = sh.z;
      wt_new |= (1u << w);
      F = F + o.F;
      T = T - radi * o.tor;
      if (any_post && W.post_freeze) {
        Fa = Fa + o.F;
        Ta = Ta - radi * o.tor;
      }
    }
    if (wt_new != wt) PE.wtouch[i] = (unsigned char)wt_new;
  }

  SF_PH(28);
  // ---- integrate: final(k) [+ initial(k+1)]  ([3P] FixNVESphere, dtf = dt/2, INERTIA = 0.4) ----
  // [3P] fix freeze: force and torque of the group's atoms are zeroed where the fix stands in the script
  if (mk & SE.freeze_bit) {
    F = Fa;
    T = Ta;
  }
  Vec3 vn = vi, wn = wi, xn = xi;
  bool gs_trig = false;
  if (SE.mode != 2 && SE.have_nve && (mk & SE.nve_bit)) {
    const double dtf = 0.5 * SE.dt;
    const double dtfm = dtf / mi;
    const double dtirot = (dtf / 0.4) / (radi * radi * mi);
    vn = vn + dtfm * F;
    wn = wn + dtirot * T;
    if (SE.mode == 0) {
      vn = vn + dtfm * F;
      xn = xn + SE.dt * vn;
      wn = wn + dtirot * T;
      const double dx = xn.x - xh_in.x, dy = xn.y - xh_in.y, dz = xn.z - xh_in.z;
      if (dx * dx + dy * dy + dz * dz > SE.trigger_sq) {
        gs_trig = true;
        atomicMin(&PE.flags[SE.trig_set], SE.kstep + SE.trig_add);
        // (fused forward pack: no kernel will copy the trigger word into the vote headers before the exchange)
        for (int p = 0; p < SE.tx_nhdr; p++) atomicMin(header_vote_ptr(PE.tx_sendbuf + PE.tx_hdr_off[p]), SE.kstep + SE.trig_add);
      }
      if (SE.margin_sq > 0.0) {
        const double sx = xn.x - xi.x, sy = xn.y - xi.y, sz = xn.z - xi.z;
        if (sx * sx + sy * sy + sz * sz > SE.margin_sq) PE.flags[F_MARGIN_FAIL] = 1;
      }
    }
  }
  // the forward halo record of an atom another GPU (or the periodic image of this slab) sees as a ghost: what
  // k_forward_pack_fused would gather after this kernel
  if (txb) {
    for (int k = 0; k < kBrickSlots; k++) {
      const int off = PE.bslot[(size_t)k * cap + i];
      if (off < 0) break;
      const int bq = off >> kBlkShift;
      if (GS) {   // whole records into the neighbour's ghost slots (its x | v | omega arrays), in the neighbour's frame
        const size_t r = (size_t)(off & kBlkMask);
        const double* sh = PE.tx_blkshift + 3 * bq;
        gs_store(reinterpret_cast<double4*>(PE.tx_blkptr[bq]) + r, xn.x + sh[0], xn.y + sh[1], xn.z + sh[2], radi);
        gs_store(reinterpret_cast<double4*>(PE.tx_blkptr[DemEngine::kMaxDirs + bq]) + r, vn.x, vn.y, vn.z, mi);
        gs_store(reinterpret_cast<double4*>(PE.tx_blkptr[2 * DemEngine::kMaxDirs + bq]) + r, wn.x, wn.y, wn.z, wi4.w);
        continue;
      }
      double* b = PE.tx_blkptr[bq] + (off & kBlkMask);
      const size_t n = PE.tx_blkcnt[bq];   // (component-major block: [kForwardDoubles][n])
      b[0] = xn.x; b[n] = xn.y; b[2 * n] = xn.z;
      b[3 * n] = vn.x; b[4 * n] = vn.y; b[5 * n] = vn.z;
      b[6 * n] = wn.x; b[7 * n] = wn.y; b[8 * n] = wn.z;
    }
  }
  if (SE.tx_fused == 1) {
    auto put = [&](double* b, size_t n, double shift) {
      b[0] = xn.x + shift; b[n] = xn.y; b[2 * n] = xn.z;
      b[3 * n] = vn.x; b[4 * n] = vn.y; b[5 * n] = vn.z;
      b[6 * n] = wn.x; b[7 * n] = wn.y; b[8 * n] = wn.z;
    };
    if (txk0 >= 0) put(PE.tx[0] + txk0, (size_t)SE.tx_n[0], SE.tx_shift[0]);
    if (txk1 >= 0) put(PE.tx[1] + txk1, (size_t)SE.tx_n[1], SE.tx_shift[1]);
  }
  SF_PH(29);
#if SF_ST_SHUFFLE
  // A 32-byte record per lane is two 16-byte stores at a 32-byte stride: each store instruction covers only half
  // of every cache line it touches.  When the whole wave holds consecutive atoms the halves are exchanged between
  // lanes so that each instruction writes 1 KiB of contiguous memory (lane l stores chunk l, then chunk 64 + l).
  if (LPA == 1 && SE.part == 0 && __ballot(1) == ~0ull && (i & 63) == (int)(threadIdx.x & 63)) {
    const int lane = threadIdx.x & 63;
    const int base = i - lane;
    // (the half-wave exchange of the gathers, backwards: the first store writes records 0..31 -- lane l their first 16
    // bytes, lane l + 32 their second 16 bytes --, the second one records 32..63; four v_permlane32_swap per record where
    // rounds 1-4 sent eight values through the LDS crossbar)
    auto store_shuffled = [&](double4* arr, double a0, double a1, double a2, double a3) {
      swap32(a0, a2);   // a0 a1: lanes < 32 keep their first half, lanes >= 32 receive the second half of lane - 32
      swap32(a1, a3);   // a2 a3: lanes < 32 receive the first half of lane + 32, lanes >= 32 keep their second half
      char* dst = reinterpret_cast<char*>(arr + base) + ((lane & 31) << 5) + ((lane & 32) >> 1);
      *reinterpret_cast<double2*>(dst) = double2{a0, a1};
      *reinterpret_cast<double2*>(dst + 1024) = double2{a2, a3};
    };
    store_shuffled(PE.xr_out, xn.x, xn.y, xn.z, radi);
    store_shuffled(PE.vm_out, vn.x, vn.y, vn.z, mi);
    store_shuffled(PE.om_out, wn.x, wn.y, wn.z, wi4.w);
  } else
#endif
  {
#if SF_NT_OUT
    st_stream4<NT_ST>(&PE.xr_out[i], double4{xn.x, xn.y, xn.z, radi});
    st_stream4<NT_ST>(&PE.vm_out[i], double4{vn.x, vn.y, vn.z, mi});
    st_stream4<NT_ST>(&PE.om_out[i], double4{wn.x, wn.y, wn.z, wi4.w});
#else
    PE.xr_out[i] = {xn.x, xn.y, xn.z, radi};
    PE.vm_out[i] = {vn.x, vn.y, vn.z, mi};
    PE.om_out[i] = {wn.x, wn.y, wn.z, wi4.w};   // .w: frozen mark travels with the record
#endif
  }
  if (SE.mode != 0) {
    PE.force[i] = {F.x, F.y, F.z, 0.0};
    PE.torque[i] = {T.x, T.y, T.z, 0.0};
  }
  SF_PH(30);
  return 1 | (txb ? 2 : 0) | (gs_trig ? 4 : 0);
}

// Registers: the plain contact kernel needs 169 VGPRs when left alone -- one more than three waves per SIMD allow
// (512 / 3, granule 8 = 168), and at two waves per SIMD it is 30 % slower (latency bound).  Asked for three waves the
// compiler finds 167 without spilling.  With ONE of the cohesive / lubrication arms the lean loop (sf_lean_variant) fits
// three waves (154 VGPRs / 168 with 12 bytes of scratch); with both arms the kernel needs 223 and stays at two.
#ifdef SF_WAVES_PER_EU
#define SF_SUBSTEP_ATTR __attribute__((amdgpu_waves_per_eu(SF_WAVES_PER_EU, SF_WAVES_PER_EU)))
#else
#define SF_SUBSTEP_ATTR __attribute__((amdgpu_waves_per_eu((COHE || LUB) && !sf_lean_variant(COHE, LUB, LPA) ? 1 : 3)))
#endif
// The dispatcher places block b on XCD b % 8 (each XCD has its own 4 MiB L2).  Atoms are sorted by bin, so giving
// every XCD one contiguous range of blocks keeps an atom's neighbours in the L2 of the XCD that gathers them
// (bijective remap, speed only: any placement gives the same result).
__device__ __forceinline__ int xcd_contiguous_block()
{
  const int bid = blockIdx.x, nb = gridDim.x, xcd = bid & 7, q = nb >> 3, r = nb & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// TP: v and omega of a neighbour are requested with its x only when the pair touched one sub-step ago (a bed that
// lists many more neighbours than it touches: -11 % in the loose disordered bed), or always (a bed whose listed
// neighbours nearly all touch: the bookkeeping of the former costs 4 % there)
template <int STYLE, bool COHE, bool LUB, int LPA, bool TP, int NTP, bool GS = false>
__global__ __launch_bounds__(256) SF_SUBSTEP_ATTR void k_substep(DemPtrs P, StepParams S)
{
  // a previous sub-step of this batch moved an atom beyond skin/2: the list is stale, do nothing
  // (the host rebuilds and relaunches from that sub-step)
  if (__atomic_load_n(&P.flags[S.trig_test], __ATOMIC_RELAXED) < S.kstep) return;
  if (GS && __atomic_load_n(&P.flags[F_HALO_TIMEOUT], __ATOMIC_RELAXED) != 0) return;   // (a wait ran out: an error already)
  SF_STAMP_WORKGROUP();   // (variant builds: this workgroup's start / end and phase marks, sf_dem_variants.h)
  // The dispatcher places block b on XCD b % 8 (each XCD has its own 4 MiB L2).  Atoms are sorted by
  // bin, so giving every XCD one contiguous range of blocks keeps an atom's neighbours in the L2 of
  // the XCD that gathers them (bijective remap, speed only: any placement gives the same result).
  int bid = blockIdx.x;
  // (ghost slots: lane x < 8 holds the number of workgroups of XCD x that run this launch; the poller of sf_dem_gs.h is the
  // first workgroup of the first XCD that has any)
  int gs_expected = 0;
  bool gs_poller = false;
  // S.sweep_rev: every other sub-step walks each XCD's range from its END.  A sub-step touches ~3 x the 256 MB of the
  // memory-side cache; sweeping always in the same direction it finds nothing of the previous sub-step there (cyclic
  // access, LRU), sweeping back and forth the first third of what it needs is what the previous sub-step touched last.
  if (S.xcd_remap == 2) {
    const int xcd = bid & 7, loc = bid >> 3;
    if (loc >= S.xcd_count[xcd]) return;
    bid = S.xcd_first[xcd] + (S.sweep_rev ? S.xcd_count[xcd] - 1 - loc : loc);
    if (GS) {
      const int lx = (int)(threadIdx.x & 7);
      gs_expected = S.xcd_count[lx];
      int firstx = 0;
      while (firstx < 7 && S.xcd_count[firstx] == 0) firstx++;
      gs_poller = loc == 0 && xcd == firstx;
    }
  } else {
    const int nb = gridDim.x, xcd = bid & 7, q = nb >> 3, r = nb & 7;
    const int cnt = xcd < r ? q + 1 : q, loc = bid >> 3;
    if (S.xcd_remap) bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (S.sweep_rev ? cnt - 1 - loc : loc);
    if (GS) {
      const int lx = (int)(threadIdx.x & 7);
      gs_expected = lx < r ? q + 1 : q;
      gs_poller = blockIdx.x == 0;
    }
  }
  // (ghost slots: rank `lane`'s flag | vote word, asked for before the wave's own rows so that
  // the round trip to the uncached lines runs under them; consumed at the gate in substep_particle)
  unsigned long long gs_w = gs_word(S.gs_seq, INT_MAX);
  if (GS && S.gs_wait) {
    const int r = (int)(threadIdx.x & 63);
    if (r < S.gs_world && r != S.gs_rank) gs_w = gs_read_line(P.gs_my_sync + kGsStride * r);
  }
  const int tid = bid * blockDim.x + threadIdx.x;
  int i = tid / LPA;          // LPA consecutive lanes share an atom
  const int q = tid % LPA;
  // (half-wave gather, COOP in substep_particle: the lanes of a wave that hold no atom stay -- on atom 0, storing nothing --
  // because their partner lanes l +- 32 read a neighbour's record together with them)
  constexpr bool COOPK = sf_coop_variant(COHE, LUB, LPA, TP, NTP);
  bool live = true;
  if (S.part == 2) {          // atoms next to the slab's x faces: a prefix and a suffix of the x-slowest order
    live = i < S.nb;
    if (live && i >= S.n_lo) i += S.n_hi - S.n_lo;
  } else if (S.part == 1) {   // everything in between
    i += S.n_lo;
    live = i < S.n_hi;
  } else {
    live = i < S.nlocal;
  }
  if (!live) i = 0;
  if (!GS && !(COOPK ? __ballot(live) != 0 : live)) return;
  // a timed launch (one in a few hundred): when did this XCD start, when did it finish?  (the engine evens the shares out)
  // (each XCD's two words on a cache line of their own, the end stamped by one workgroup in eight: atomics on one line
  // are resolved one after the other at the memory side, ~11 ns each -- 31 k of them doubled the launch)
  const int xq = (int)(blockIdx.x & 7) * 64;
  if (S.xcd_time && threadIdx.x == 0 && (blockIdx.x >> 3) == 0) atomicMin(&P.xcd_time[xq], (int)(wall_clock64() & 0x3fffffff));
  if (GS) {
    // (every workgroup that gets here counts itself done, whatever part of it holds atoms: the lanes beyond the last atom
    // stay for the wave-level hand-off instead of returning)
    int ran = 1;
    if (COOPK ? __ballot(live) != 0 : live)
      ran = substep_particle<STYLE, COHE, LUB, false, LPA, TP, NTP, true>(P, S, i, q, nullptr, nullptr, nullptr, gs_w, live);
    if (__ballot(ran == 0)) return;   // (stopped at the gate -- the whole wave did: the gate is a wave-level decision)
    if (S.mode == 0) gs_done(P, S, (ran & 2) != 0, (ran & 4) != 0, gs_poller, gs_expected);
  } else {
    substep_particle<STYLE, COHE, LUB, false, LPA, TP, NTP>(P, S, i, q, nullptr, nullptr, nullptr, 0ull, live);
  }
  if (S.xcd_time && threadIdx.x == 0 && ((blockIdx.x >> 3) & 7) == 0)
    atomicMax(&P.xcd_time[xq + 32], (int)(wall_clock64() & 0x3fffffff));
}


#ifdef SF_EXP_PERSIST   // (pricing arm of round 6, tests/build_variant.sh pers -DSF_EXP_PERSIST=1: measured slower, not shipped --
                        //  profiles/r06_README.md section 2)
// ------------------------------------------------------------------------------------------------
// Persistent tiles (round 6).  k_substep starts one wave per 64 atoms: every wave pays its own start -- kernel arguments,
// the round trip for its atom's records, row count and first list words (3.7 us of a 33 us life) -- with nothing else of its
// own to hide it behind.  Here the grid is the RESIDENT waves (three per SIMD), each walking tiles of 64 atoms of its XCD's
// range: the first two by position, the others pulled from the XCD's head word (one returning atomic per tile, requested
// two tiles ahead so that nothing ever waits for it).  When the neighbour loop of tile t is over -- its two prefetch
// register sets are dead -- the wave requests tile t + 1's records, row count and first words (TilePre) and only then runs
// tile t's fixes, integration and stores: the next tile's start-up round trip runs under the current tile's epilogue.
// Same atoms, same operations in the same order as k_substep (tiles are the one-wave workgroups of the plain launch): the
// results are bit-identical; which wave works on which tile never enters them.
// One lane per atom, no ghost slots, no boundary / interior split.  Head words: two sets, the launch pulls from set
// S.pq_par and clears the other one for the next launch (every launch does, before the trigger test: a launch that
// returns at once still leaves the invariant in place).
// ------------------------------------------------------------------------------------------------
template <int STYLE, bool COHE, bool LUB, bool TP, int NTP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu((COHE && LUB) ? 1 : 3)))
void k_substep_persist(DemPtrs P, StepParams S)
{
  const int lane = (int)threadIdx.x;
  if (blockIdx.x < 8 && lane == 0) P.pq_head[((S.pq_par ^ 1) * 8 + (int)blockIdx.x) * 32] = 0;
  if (__atomic_load_n(&P.flags[S.trig_test], __ATOMIC_RELAXED) < S.kstep) return;
  const int xcd = (int)(blockIdx.x & 7), loc = (int)(blockIdx.x >> 3), nres = (int)(gridDim.x >> 3);
  const int cnt = S.xcd_count[xcd], first = S.xcd_first[xcd];
  if (loc >= cnt) return;
  int* const head = P.pq_head + (S.pq_par * 8 + xcd) * 32;
  const int xq = xcd * 64;
  if (S.xcd_time && lane == 0 && loc == 0) atomicMin(&P.xcd_time[xq], (int)(wall_clock64() & 0x3fffffff));
  constexpr bool NT_LD = NTP != 0;
  // tile t of this XCD's range -> its first atom (wave-uniform); a lane past the last atom works on atom 0 and stores nothing
  auto base_of = [&](const int t) { return (first + (S.sweep_rev ? cnt - 1 - t : t)) * 64; };
  // tiles of this wave: loc, loc + nres (by position), then 2 nres + what the head word returns.  The pull for the tile
  // after the next one is issued with the next tile's prefetch (substep_particle, behind the neighbour loop) and read here,
  // one epilogue later: nothing waits for it
  int t_next = loc + nres;                   // (wave-uniform: scalar registers)
  int base = base_of(loc);
  TilePre pre;
  pre.head = head;
  pre.q = 0;
  {
    const int i0 = base + lane;
    tile_prefetch<NT_LD>(P, S, i0 < S.nlocal ? i0 : 0, pre);
  }
  // The epilogue of a tile reads the kernel arguments through a pointer the compiler cannot see through, renewed per tile:
  // left alone it hoists every scalar load of the ~150 words of StepParams out of the tile loop, keeps them all live across
  // it and spills (169 scalar + 37 vector registers).  The neighbour loop reads P and S directly: its scalars stay in
  // registers across the tiles, as they do across the loop in k_substep (re-reading them inside the loop: +12 %).
  typedef const char __attribute__((address_space(4))) * KernArg;
  const KernArg ka0 = (KernArg)__builtin_amdgcn_kernarg_segment_ptr();
  constexpr size_t kSOff = (sizeof(DemPtrs) + alignof(StepParams) - 1) / alignof(StepParams) * alignof(StepParams);
  for (;;) {
    KernArg ka = ka0;
    asm volatile("" : "+s"(ka));
    pre.Pe = (const DemPtrs*)(ka);
    pre.Se = (const StepParams*)(ka + kSOff);
    const int base_next = t_next < cnt ? base_of(t_next) : -1;
    const int i = base + lane;
    const bool live = i < S.nlocal;
    substep_particle<STYLE, COHE, LUB, false, 1, TP, NTP, false, true>(P, S, live ? i : 0, 0, nullptr, nullptr, nullptr, 0ull,
                                                                        live, &pre, base_next);
    if (base_next < 0) break;
    if (!SF_PERS_PREFETCH) {   // (pricing arm: the next tile's records requested when the current tile is done -- no overlap)
      if (SF_PERS_DYNAMIC && lane == 0) pre.q = atomicAdd(head, 1);
      const int in = base_next + lane;
      tile_prefetch<NT_LD>(P, S, in < S.nlocal ? in : 0, pre);
    }
    base = base_next;
    t_next = SF_PERS_DYNAMIC ? 2 * nres + __builtin_amdgcn_readfirstlane(pre.q) : t_next + nres;
  }
  if (S.xcd_time && lane == 0 && (loc & 7) == 0) atomicMax(&P.xcd_time[xq + 32], (int)(wall_clock64() & 0x3fffffff));
}

#endif

// (the LDS-staged cell-bin kernel, k_substep_lds -- the same substep_particle on a tile's staged copy -- is in
// sf_dem_lds_kernel.h)
// first half-kick of a run with the forces stored by the previous run's last sub-step
__global__ __launch_bounds__(256) void k_initial_integrate(double4* xr, double4* vm, double4* om,
                                                           const double4* force, const double4* torque,
                                                           const double* xhold, int* flags, int nlocal,
                                                           size_t cap, double dt, double trigger_sq,
                                                           const int* mask, int nve_bit)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nlocal) return;
  if (mask && !(mask[i] & nve_bit)) return;   // not in the group of fix nve/sphere
  double4 x = xr[i], v = vm[i], w = om[i];
  const double4 f = force[i], t = torque[i];
  const double dtf = 0.5 * dt;
  const double dtfm = dtf / v.w;
  const double dtirot = (dtf / 0.4) / (x.w * x.w * v.w);
  v.x += dtfm * f.x;
  v.y += dtfm * f.y;
  v.z += dtfm * f.z;
  x.x += dt * v.x;
  x.y += dt * v.y;
  x.z += dt * v.z;
  w.x += dtirot * t.x;
  w.y += dtirot * t.y;
  w.z += dtirot * t.z;
  xr[i] = x;
  vm[i] = v;
  om[i] = w;
  const double dx = x.x - xhold[i], dy = x.y - xhold[cap + i], dz = x.z - xhold[2 * cap + i];
  if (dx * dx + dy * dy + dz * dz > trigger_sq) atomicMin(flags, -1);   // flags: the trigger word to use
}

// [3P] Comm::forward_comm for images owned by this GPU: ghost = root atom + accumulated shift
__global__ __launch_bounds__(256) void k_ghost_forward(double4* xr, double4* vm, double4* om,
                                                       const int* gsrc, const double* gshift,
                                                       int nlocal, int nghost, size_t cap,
                                                       const int* flags, int kstep, int trig_word, int phase)
{
  if (__atomic_load_n(&flags[trig_word], __ATOMIC_RELAXED) < kstep) return;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nghost) return;
  const int g = nlocal + k;
  const int src = gsrc[g];
  if (src < 0) return;  // ghost owned by another GPU: filled by sf_dem_forward_unpack
  // phase 1: images of owned atoms only ; phase 2: images of ghosts owned by another GPU only ; 0: both
  if (phase == 1 && src >= nlocal) return;
  if (phase == 2 && src < nlocal) return;
  double4 x = xr[src];
  x.x += gshift[g];
  x.y += gshift[cap + g];
  x.z += gshift[2 * cap + g];
  xr[g] = x;
  vm[g] = vm[src];
  om[g] = om[src];
}

// overlapped halo: an owned atom is a BOUNDARY atom when the forward halo sends it or when its list holds a ghost
// whose root is owned by another GPU (gsrc < 0, or an image of such a ghost); everything else is interior and
// never reads what the halo exchange writes.
__global__ __launch_bounds__(256) void k_mark_boundary(const int* neigh, const int* numneigh, const int* gsrc,
                                                       const int* send0, int n0, const int* send1, int n1,
                                                       int nlocal, size_t cap, unsigned char* isb, int phase,
                                                       int roots)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (phase == 0) {
    if (t >= nlocal) return;
    const int nn = numneigh[t];
    unsigned char b = 0;
    for (int s = 0; s < nn; s++) {
      const int j = neigh_index(neigh[(size_t)s * cap + t], roots);
      if (j >= nlocal) {
        const int r = roots ? -1 : gsrc[j];   // root mode: an index >= nlocal IS a ghost owned by another GPU
        if (r < 0 || r >= nlocal) {
          b = 1;
          break;
        }
      }
    }
    isb[t] = b;
  } else {
    if (t < n0) isb[send0[t]] = 1;
    else if (t < n0 + n1) isb[send1[t - n0]] = 1;
  }
}

}  // namespace sf
