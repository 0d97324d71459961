// sf_dem_rebuild.h -- kernels of the neighbour rebuild (gfx950, wave64): [3P] Domain::pbc + binning, the counting sort by
// (cell, tag), the permutation of the per-atom arrays, periodic images as ghost atoms, [3P] Neighbor::build as a full list
// with FixShearHistory's re-injection (k_build_neigh), partner slots, list statistics, the LDS staging tables.
// Included by sf_dem.hip after sf_dem_kernels.h (the sub-step kernel family, whose sources alone carry the hash a committed
// counter pass is matched against: sedifoam_amd/build.py kernel_source_hash).
#pragma once
#include <climits>

#include "sf_dem_kernels.h"

namespace sf {

// ------------------------------------------------------------------------------------------------
// neighbour rebuild
// ------------------------------------------------------------------------------------------------
// FixShearHistory::pre_exchange [3P]: remember each touching partner by tag, and give BOTH sides of every contact a
// copy of its shear history (the partner side: the owner's value, negated -- what FixShearHistory's "sign-flipped
// copy for j" is).  hist_out is the ping-pong buffer the sub-steps are not using.
__global__ __launch_bounds__(256) void k_partner_tags(const int* neigh, const int* numneigh, const int* tag,
                                                      int* ptag, const double* shear, double* hist_out, int nlocal,
                                                      size_t cap, int M, int roots)
{
  const int i = xcd_contiguous_block() * blockDim.x + threadIdx.x;   // (gathers from the neighbours' rows)
  if (i >= nlocal) return;
  const int nn = numneigh[i];
  // (rows beyond an atom's count are never read -- the list build and the migration pack stop at numneigh -- and the walk
  // ends at the longest row of the WAVE, not of the bed: a loose bed's longest row is twice its mean)
  int nmax = nn;
  {
    const unsigned long long act = __ballot(1);
    const int lane = threadIdx.x & 63;
    for (int off = 32; off > 0; off >>= 1) {
      const int o = __shfl_xor(nmax, off, 64);
      if ((act >> (lane ^ off)) & 1ull) nmax = max(nmax, o);
    }
    nmax = min(nmax, M);
  }
  // four slots at a time, their loads issued together: the words, then the partner tags and the histories of those that
  // touch, then the stores (one slot after the other the walk is a chain of ~14 dependent round trips: 20 us at 100 k grains)
  constexpr int kU = 4;
  for (int s0 = 0; s0 < nmax; s0 += kU) {
    int jraw[kU];
#pragma unroll
    for (int u = 0; u < kU; u++) jraw[u] = s0 + u < nn ? neigh[(size_t)(s0 + u) * cap + i] : 0;
    int t[kU];
    double hx[kU], hy[kU], hz[kU];
#pragma unroll
    for (int u = 0; u < kU; u++) {
      t[u] = -1;
      hx[u] = hy[u] = hz[u] = 0.0;
      if (jraw[u] & kTouchBit) {   // (a word beyond the row is 0: no bit)
        const int j = neigh_index(jraw[u], roots);
        t[u] = tag[j];
        size_t src = (size_t)(3 * (s0 + u)) * cap + i;
        if (!(jraw[u] & kOwnBit)) src = (size_t)(3 * ((jraw[u] >> kIdxBits) & 31)) * cap + j;   // (partner sides: root mode only)
        hx[u] = shear[src];
        hy[u] = shear[src + cap];
        hz[u] = shear[src + 2 * cap];
      }
    }
#pragma unroll
    for (int u = 0; u < kU; u++) {
      if (s0 + u >= nn) continue;
      if (jraw[u] & kTouchBit) {
        const double sign = (jraw[u] & kOwnBit) ? 1.0 : -1.0;
        const size_t dst = (size_t)(3 * (s0 + u)) * cap + i;
        hist_out[dst] = sign * hx[u];
        hist_out[dst + cap] = sign * hy[u];
        hist_out[dst + 2 * cap] = sign * hz[u];
      }
      ptag[(size_t)(s0 + u) * cap + i] = t[u];
    }
  }
}

// After the list build: a partner-side slot (kOwnBit clear; root mode, the neighbour is an atom of this GPU itself,
// image code 13) needs the slot of this atom in the owner's list.  It goes into the word's five image-code bits --
// the kernel knows a partner-side neighbour is not an image.  The list criterion is symmetric for two atoms of this
// GPU, so the owner lists the partner back; should it not (a list cut short), or should the slot not fit five bits
// (> 32 neighbours), this side keeps a copy of its own (kOwnBit): a copy on each side is always valid, both evolve
// to bitwise opposite values.  Index mode (LDS-staged kernel) has no spare bits: every slot owns its copy.
// (flags: the rebuild trigger is re-armed here -- the last kernel of a rebuild -- instead of by a launch of its own; nullptr:
// the caller does that)
__global__ __launch_bounds__(128) void k_back_slots(int* neigh, const int* numneigh, int nlocal, size_t cap, int roots,
                                                    int* flags)
{
  if (flags && blockIdx.x == 0 && threadIdx.x == 0) flags[F_TRIGGER] = INT_MAX;
  const int i = xcd_contiguous_block() * blockDim.x + threadIdx.x;   // (walks the rows of this atom's neighbours)
  if (i >= nlocal) return;
  const int nn = numneigh[i];
  const int codemask = 31 << kIdxBits;
  const int want = (kNoShift << kIdxBits) | i;   // "atom i itself, not an image", as an owner-side word reads
  for (int s = 0; s < nn; s++) {
    const int w = neigh[(size_t)s * cap + i];
    if (w & kOwnBit) continue;
    int t = -1;
    if (roots && s < 32) {   // (the per-atom masks of k_partner_coalesced cover 32 slots)
      const int r = w & kIdxMask;
      const int nr = numneigh[r];
      // owner-side words of r are never rewritten by this kernel (only partner-side ones are, and those point below r)
      for (int u = 0; u < nr && u < 32; u++) {
        const int wu = neigh[(size_t)u * cap + r];
        if ((wu & kOwnBit) && (wu & (codemask | kIdxMask)) == want) {
          t = u;
          break;
        }
      }
    }
    neigh[(size_t)s * cap + i] = t < 0 ? (w | kOwnBit) : ((w & ~codemask) | (t << kIdxBits));
  }
}

// One history copy per contact pays off when the partner side's gather of the owner's row is coalesced: when the
// lane next to it (atom i - 1 or i + 1) reads, in the same slot, the row of the NEXT owner (r - 1 / r + 1) -- the rule
// in an ordered bed, the exception in a disordered one, where three scattered 8-byte gathers per contact cost more
// requests than the second copy saves (measured, 1 M grains, us per sub-step, two copies / one copy: lattice 214 / 204,
// jitter 0.15 bed 246 / 219, jitter 0.3 loose bed 249 / 272).  This kernel measures that on the finished list: of the
// slots that point at a LOWER-indexed atom of this GPU itself (the would-be partner sides, whichever mode the list
// was built in), how many have such a lane neighbour.  The engine picks the mode of the NEXT list build from the
// ratio (with hysteresis): the answer depends on the particles only, so runs stay reproducible.
// (statistics: every `stride`-th block of 1024 atoms is looked at -- the same atoms on every run)
__global__ __launch_bounds__(1024) void k_partner_coalescing(const int* neigh, const int* numneigh, int nlocal,
                                                            size_t cap, int* counters, int stride)
{
  const int i = blockIdx.x * stride * blockDim.x + threadIdx.x;
  int total = 0, coal = 0, listed = 0, touching = 0;
  if (i < nlocal) {
    const int nn = numneigh[i];
    const int codemask = 31 << kIdxBits;
    listed = nn;
    for (int s = 0; s < nn; s++) {
      const int w = neigh[(size_t)s * cap + i];
      touching += (w & kTouchBit) ? 1 : 0;
      const int r = w & kIdxMask;
      const bool partner_side = !(w & kOwnBit) || (r < i && (w & codemask) == (kNoShift << kIdxBits));
      if (!partner_side || r >= nlocal) continue;
      total++;
      bool ok = false;
      if (i + 1 < nlocal && s < numneigh[i + 1]) ok = (neigh[(size_t)s * cap + i + 1] & kIdxMask) == r + 1;
      if (!ok && i > 0 && s < numneigh[i - 1]) ok = (neigh[(size_t)s * cap + i - 1] & kIdxMask) == r - 1;
      coal += ok ? 1 : 0;
    }
  }
  const int t = block_sum_int_1024(total);
  const int c = block_sum_int_1024(coal);
  const int l = block_sum_int_1024(listed);
  const int u = block_sum_int_1024(touching);
  if (threadIdx.x == 0 && l) {
    atomicAdd(&counters[0], t);
    atomicAdd(&counters[1], c);
    atomicAdd(&counters[2], l);
    atomicAdd(&counters[3], u);
  }
}

// Debug statistic (SF_DEBUG_LINES=1): how many 128-byte lines one gather instruction of the sub-step kernel touches.
// One wave = 64 consecutive atoms, slot s: lane l reads 16 bytes of the 32-byte record of its neighbour j(l), twice
// (the two halves) -> 2 x [distinct lines among all 64 lanes]; if lanes 2k, 2k + 1 read the 32 bytes of ONE record
// together (first j(2k), then j(2k + 1)) -> [distinct among even lanes] + [distinct among odd lanes].
// out: {instructions (slots with an active lane), active lanes, distinct lines all, distinct even, distinct odd,
//       distinct 64-byte half lines all}
__global__ __launch_bounds__(64) void k_gather_lines(const int* neigh, const int* numneigh, int nlocal, size_t cap,
                                                     int stride, unsigned long long* out)
{
  const int lane = threadIdx.x;
  const int i = blockIdx.x * stride * 64 + lane;
  const int nn = i < nlocal ? numneigh[i] : 0;
  int nmax = nn;
  for (int off = 32; off > 0; off >>= 1) nmax = max(nmax, __shfl_xor(nmax, off, 64));
  unsigned long long instr = 0, act = 0, all = 0, ev = 0, od = 0, half = 0;
  for (int s = 0; s < nmax; s++) {
    const int j = s < nn ? (neigh[(size_t)s * cap + i] & kIdxMask) : -1;
    const int line = j < 0 ? -1 : j >> 2, hl = j < 0 ? -1 : j >> 1;
    bool first_all = line >= 0, first_par = line >= 0, first_half = hl >= 0;
    for (int k = 1; k < 64; k++) {
      const int src = (lane + 64 - k) & 63;
      const int ol = __shfl(line, src, 64), oh = __shfl(hl, src, 64);
      if (src < lane) {
        if (ol == line) {
          first_all = false;
          if (((src ^ lane) & 1) == 0) first_par = false;
        }
        if (oh == hl) first_half = false;
      }
    }
    const unsigned long long a = __ballot(line >= 0), fa = __ballot(first_all), fp = __ballot(first_par),
                             fh = __ballot(first_half);
    const unsigned long long evens = 0x5555555555555555ull;
    instr += 1;
    act += __popcll(a);
    all += __popcll(fa);
    ev += __popcll(fp & evens);
    od += __popcll(fp & ~evens);
    half += __popcll(fh);
  }
  if (lane == 0 && nmax) {
    atomicAdd(&out[0], instr); atomicAdd(&out[1], act); atomicAdd(&out[2], all);
    atomicAdd(&out[3], ev); atomicAdd(&out[4], od); atomicAdd(&out[5], half);
  }
}

struct PbcParams {
  double lo[3], hi[3];
  int wrap[3];
};

__device__ __forceinline__ int bin_coord(double x, double lo, double inv, int n, int& lost)
{
  int c = (int)floor((x - lo) * inv);
  if (c < -1 || c > n) lost = 1;
  c = c < 0 ? 0 : c;
  c = c >= n ? n - 1 : c;
  return c;
}

__device__ __forceinline__ int bin_of(const double4& x, const BinGrid& g, int& lost)
{
  const int cx = bin_coord(x.x, g.lo[0], g.inv[0], g.n[0], lost);
  const int cy = bin_coord(x.y, g.lo[1], g.inv[1], g.n[1], lost);
  const int cz = bin_coord(x.z, g.lo[2], g.inv[2], g.n[2], lost);
  return bin_key(g, cx, cy, cz);
}

__global__ __launch_bounds__(1024) void k_max_int(const int* v, int n, int* out)
{
  __shared__ int ws[16];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int m = i < n ? v[i] : 0;
  for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_down(m, off, 64));
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) ws[w] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < (int)(blockDim.x >> 6); k++) m = max(m, ws[k]);
    atomicMax(out, m);
  }
}

// x-slowest order: number of owned atoms in cell layers cx < cx_lo and cx < cx_hi (prefix lengths)
__global__ __launch_bounds__(1024) void k_count_layers(const double4* xr, int nlocal, BinGrid g, int cx_lo, int cx_hi,
                                                       int* counters)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int lost = 0;
  int a = 0, b = 0;
  if (i < nlocal) {
    const int cx = bin_coord(xr[i].x, g.lo[0], g.inv[0], g.n[0], lost);
    a = cx < cx_lo;
    b = cx < cx_hi;
  }
  const int ta = block_sum_int_1024(a);
  const int tb = block_sum_int_1024(b);
  if (threadIdx.x == 0) {
    if (ta) atomicAdd(&counters[0], ta);
    if (tb) atomicAdd(&counters[1], tb);
  }
}

// Runs of equal keys among the lanes of a wave (the atoms arrive nearly sorted: ~2.7 neighbours in memory share a cell): the
// first lane of a run speaks for it -- one atomic per run instead of one per atom (the atomics of the cell histograms are
// served at the memory side, ~30 us per million).  `head`: the lane that starts this lane's run, `len`: the run's length
// (valid in its head), `rank`: this lane's place in it.
__device__ __forceinline__ void key_runs(const unsigned key, int& head, int& len, int& rank)
{
  const int lane = threadIdx.x & 63;
#ifdef SF_EXP_NO_KEY_RUNS
  head = lane;   // pricing / bisecting arm: every lane a run of its own
  len = 1;
  rank = 0;
  return;
#endif
  const unsigned long long act = __ballot(1);
  const unsigned prev = (unsigned)__shfl_up((int)key, 1, 64);
  const bool starts = lane == 0 || !((act >> (lane - 1)) & 1ull) || prev != key;
  const unsigned long long heads = __ballot(starts);
  const unsigned long long upto = heads & (~0ull >> (63 - lane));          // heads at or below this lane
  head = 63 - __clzll((long long)upto);
  rank = lane - head;
  const unsigned long long above = (heads & act) >> head >> 1;            // heads above this run's head ...
  const unsigned long long act_above = act >> head >> 1;                  // ... and how far the active lanes reach
  const int to_next = above ? __ffsll((long long)above) : 65;
  const int to_end = (~act_above) ? __ffsll((long long)~act_above) : 65;   // first inactive lane above the head
  len = (to_next < to_end ? to_next : to_end);
}

// [3P] Domain::pbc for owned atoms + bin key
// (count: the counting sort's histogram, filled in the same pass; nullptr on the radix-sort path)
__global__ __launch_bounds__(256) void k_pbc_keys(double4* xr, int nlocal, PbcParams pb, BinGrid g,
                                                  unsigned* keys, int* perm, int* flags, int* count)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nlocal) return;
  double4 x = xr[i];
  double c[3] = {x.x, x.y, x.z};
  bool moved = false;
  for (int k = 0; k < 3; k++) {
    if (!pb.wrap[k]) continue;
    const double prd = pb.hi[k] - pb.lo[k];
    if (c[k] < pb.lo[k]) {
      c[k] += prd;
      moved = true;
    }
    if (c[k] >= pb.hi[k]) {
      c[k] -= prd;
      if (c[k] < pb.lo[k]) c[k] = pb.lo[k];
      moved = true;
    }
  }
  if (moved) {
    x.x = c[0];
    x.y = c[1];
    x.z = c[2];
    xr[i] = x;
  }
  int lost = 0;
  const unsigned key = (unsigned)bin_of(x, g, lost);
  keys[i] = key;
  perm[i] = i;
  if (count) {
    int head, len, rank;
    key_runs(key, head, len, rank);
    if (rank == 0) atomicAdd(&count[key], len);
  }
  if (lost) flags[F_LOST] = 1;
  // (the counters of the list build that follows start from zero: no launch of their own -- DemEngine::build_flags_clean_)
  if (i == 0) {
    flags[F_NEIGH_OVER] = 0;
    flags[F_MAXNEIGH] = 0;
    flags[F_PARK_OVER] = 0;
  }
}

__global__ __launch_bounds__(256) void k_gather4(double4* dst, const double4* src, const int* perm, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[perm[i]];
}

// every per-atom array of the owned atoms in one launch: dst[i] = src[perm[i]] (DemEngine::permute_locals)
struct PermuteJobs {
  static constexpr int kRowArrays = 5;
  const double4* s4[3];
  double4* d4[3];
  const int* si[4];
  int* di[4];
  int nd;                       // component-major double arrays, rd[a] rows each
  const double* sd[kRowArrays];
  double* dd[kRowArrays];
  int rd[kRowArrays];
  const unsigned char* sb;      // (nullptr: none)
  unsigned char* db;
};
__global__ __launch_bounds__(256) void k_permute_all(PermuteJobs J, const int* perm, int n, size_t cap)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int p = perm[i];
  const double4 a = J.s4[0][p], b = J.s4[1][p], c = J.s4[2][p];
  const int t0 = J.si[0][p], t1 = J.si[1][p], t2 = J.si[2][p], t3 = J.si[3][p];
  J.d4[0][i] = a;
  J.d4[1][i] = b;
  J.d4[2][i] = c;
  J.di[0][i] = t0;
  J.di[1][i] = t1;
  J.di[2][i] = t2;
  J.di[3][i] = t3;
  for (int k = 0; k < J.nd; k++) {
    const double* s = J.sd[k];
    double* d = J.dd[k];
    for (int r = 0; r < J.rd[k]; r++) d[(size_t)r * cap + i] = s[(size_t)r * cap + p];
  }
  if (J.sb) J.db[i] = J.sb[p];
}

// k_key_rank and k_permute_all in one launch (beds whose rebuild chain is bound by the launches it takes): thread i ranks atom i
// inside its cell by tag -- its place p in the new order -- and SCATTERS the atom's records and rows there (the atoms arrive
// nearly sorted: p is close to i, the scattered stores of a wave cover nearly whole lines); perm[p] = i as k_key_rank writes it.
__global__ __launch_bounds__(256) void k_rank_permute(PermuteJobs J, const unsigned* keys, int n, const int* first,
                                                      const int* arrival, int* perm, size_t cap)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned b = keys[i];
  const int s = first[b], e = first[b + 1];
  const int* const tag = J.si[0];   // (the first int array of the job list is the tag array: DemEngine::permute_locals)
  const int ti = tag[i];
  int r = 0;
  for (int k = s; k < e; k++) {
    const int a = arrival[k], ta = tag[a];
    r += (ta < ti || (ta == ti && a < i)) ? 1 : 0;
  }
  const int p = s + r;
  perm[p] = i;
  const double4 x = J.s4[0][i], v = J.s4[1][i], w = J.s4[2][i];
  const int t1 = J.si[1][i], t2 = J.si[2][i], t3 = J.si[3][i];
  J.d4[0][p] = x;
  J.d4[1][p] = v;
  J.d4[2][p] = w;
  J.di[0][p] = ti;
  J.di[1][p] = t1;
  J.di[2][p] = t2;
  J.di[3][p] = t3;
  for (int k = 0; k < J.nd; k++) {
    const double* src = J.sd[k];
    double* dst = J.dd[k];
    for (int q = 0; q < J.rd[k]; q++) dst[(size_t)q * cap + p] = src[(size_t)q * cap + i];
  }
  if (J.sb) J.db[p] = J.sb[i];
}

template <class T>
__global__ __launch_bounds__(256) void k_gather_rows(T* dst, const T* src, const int* perm, int n, int rows,
                                                     size_t cap)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int p = perm[i];
  for (int r = 0; r < rows; r++) dst[(size_t)r * cap + i] = src[(size_t)r * cap + p];
}

// [3P] Comm::borders on one processor, one periodic dimension: atoms (owned or already-ghost)
// within cutghost of a face get an image on the other side
struct GhostPtrs {
  double4 *xr, *vm, *om;
  int *tag, *type, *mask, *gsrc;
  double* gshift;
};

// [3P] Comm::borders for the periodic images this GPU makes itself, one dimension per pass (so that images of
// images give the edge/corner ghosts).  Two kernels: the atoms within `cut` of a periodic face are first listed
// (4 bytes each; along the fastest sort dimension they are scattered one per row of atoms, and letting each of them
// write its ~150 bytes of ghost record next to another XCD's took 360 us at 1 M atoms), then one thread per listed
// atom writes the complete ghost, coalesced.
// The number of atoms to look at -- owned + ghosts made so far, flags[before_idx] -- and the running ghost count live
// on the device: the images of two or three periodic dimensions are made back to back without a host round trip each
// (the host reads the total once, after the last dimension).
__global__ __launch_bounds__(1024) void k_ghost_select(const double4* xr, const int* flags, int before_idx, int nlocal,
                                                       int dim, double lo, double hi, double cut, int* list,
                                                       int* counter, size_t cap)
{
  // ONE global atomic per 1024-thread block: same-address atomics from different XCDs cost ~11 ns each, and along
  // the fastest sort dimension nearly every wave holds a taker (one atomic per wave was 360 us at 1 M atoms)
  __shared__ int wcount[16];
  __shared__ int wbase[16];
  const size_t nall0 = min((size_t)nlocal + (size_t)flags[before_idx], cap);   // (a word nothing changes meanwhile)
  if ((size_t)blockIdx.x * blockDim.x >= nall0) return;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = (size_t)p < nall0;
  double xp = 0.0;
  if (valid) {
    const double4 x = xr[p];
    xp = (dim == 0) ? x.x : (dim == 1) ? x.y : x.z;
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const bool t0 = valid && xp >= lo && xp <= lo + cut;
  const bool t1 = valid && xp >= hi - cut && xp <= hi;
  const unsigned long long m0 = __ballot(t0), m1 = __ballot(t1);
  const int n0 = __popcll(m0), n1 = __popcll(m1);
  if (lane == 0) wcount[w] = n0 + n1;
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    const int nw = blockDim.x >> 6;
    for (int k = 0; k < nw; k++) {
      wbase[k] = tot;
      tot += wcount[k];
    }
    const int base = tot ? atomicAdd(counter, tot) : 0;
    for (int k = 0; k < nw; k++) wbase[k] += base;
  }
  __syncthreads();
  const unsigned long long below = (1ull << lane) - 1ull;
  // (a list entry beyond the capacity belongs to a ghost that will not be created: the host grows and repeats)
  const size_t k0 = (size_t)wbase[w] + __popcll(m0 & below), k1 = (size_t)wbase[w] + n0 + __popcll(m1 & below);
  if (t0 && k0 < cap) list[k0] = p;
  if (t1 && k1 < cap) list[k1] = p | 0x40000000;
}

// ghosts [flags[before_idx], flags[F_GHOST_COUNT]) of this dimension; flags[next_idx] <- the count the next dimension
// starts from
__global__ __launch_bounds__(256) void k_ghost_create(GhostPtrs G, const int* list, int nlocal, int dim, double prd,
                                                      size_t cap, int* flags, int before_idx, int next_idx)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int first = flags[before_idx], total = flags[F_GHOST_COUNT];
  const bool over = (size_t)nlocal + (size_t)total > cap;
  if (k == 0) {
    if (next_idx >= 0) flags[next_idx] = total;
    if (over) flags[F_GHOST_OVER] = 1;
  }
  if (over || k >= total - first) return;
  const int e = list[first + k];
  const int p = e & 0x3FFFFFFF;
  const int dir = (e >> 30) & 1;
  const size_t g = (size_t)nlocal + first + k;
  const double sh = (dir == 0) ? prd : -prd;
  double4 xg = G.xr[p];
  if (dim == 0) xg.x += sh;
  else if (dim == 1) xg.y += sh;
  else xg.z += sh;
  G.xr[g] = xg;
  G.vm[g] = G.vm[p];
  G.om[g] = G.om[p];
  G.tag[g] = G.tag[p];
  G.type[g] = G.type[p];
  G.mask[g] = G.mask[p];
  // root = an owned atom or a ghost owned by another GPU (gsrc < 0)
  int root = p;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  if (p >= nlocal && G.gsrc[p] >= 0) {
    root = G.gsrc[p];
    s0 = G.gshift[p];
    s1 = G.gshift[cap + p];
    s2 = G.gshift[2 * cap + p];
  }
  if (dim == 0) s0 += sh;
  else if (dim == 1) s1 += sh;
  else s2 += sh;
  G.gsrc[g] = root;
  G.gshift[g] = s0;
  G.gshift[cap + g] = s1;
  G.gshift[2 * cap + g] = s2;
}

// ghosts are not moved: they are index-sorted by (bin, tag) so the list order is deterministic
__global__ __launch_bounds__(256) void k_ghost_keys(const double4* xr, const int* tag, int nlocal, int nghost,
                                                    BinGrid g, unsigned long long* keys, int* idx, int* flags)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nghost) return;
  int lost = 0;
  const int b = bin_of(xr[nlocal + k], g, lost);
  keys[k] = ((unsigned long long)(unsigned)b << 32) | (unsigned)tag[nlocal + k];
  idx[k] = nlocal + k;
  if (lost) flags[F_LOST] = 1;
}

// cell_start/cell_end from sorted keys (key >> shift = bin)
// (stride 4: the cell table interleaves {owned start, owned end, ghost start, ghost end} per cell, one 16-byte load
// in the list build instead of four loads from four 32 MB arrays)
template <class K>
__global__ __launch_bounds__(256) void k_cell_bounds(const K* keys, int n, int shift, int* cstart, int* cend,
                                                     int stride)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int b = (int)(keys[i] >> shift);
  if (i == 0 || (int)(keys[i - 1] >> shift) != b) cstart[(size_t)b * stride] = i;
  if (i == n - 1 || (int)(keys[i + 1] >> shift) != b) cend[(size_t)b * stride] = i + 1;
}

struct BuildParams {
  int nlocal, M, Mold;
  size_t cap;
  double skin_gran;   // skin added to ri + rj (granular list) ; < 0: no granular criterion
  double cut_lub;     // absolute cutoff: lubrication cutoff + skin, the regular list of fix cohesive ; 0: off
  BinGrid g;
  const int* eoff;    // LDS staging: [tile][(T+2)^3] offsets (nullptr: no staging tables)
  unsigned short* nloc;
  int roots;          // store (root, image code) instead of the ghost's own index
  const int* gsrc;    // root of a periodic image (-1: ghost owned by another GPU)
  const double* gshift;
  double inv_prd[3];
  double prd[3];      // box lengths (BinGrid::wrap: the shift of a candidate found around the box)
  // first sorted position of EVERY cell (entry nbins = one past the last atom), or nullptr (tile-major keys, LDS
  // staging): see the row walk in k_build_neigh
  const int* lb_own;
  const int* lb_ghost;
  const int* old_index;   // new index -> index before the re-sort (history rows not permuted), or nullptr
  int two_copies;         // every side of every contact keeps its own history copy (see k_partner_coalescing)
  int touch_first;        // row path: touching neighbours take the first slots of a row (loose beds)
  // old list read in place (single domain, row path): the words of the list being replaced and the tags in the order
  // they index (the arrays the re-sort swapped out).  The partner tags and the partner side's history copies that
  // k_partner_tags would have staged are then looked up here, per touching pair; nullptr: staged rows (ptag_old, shear_old)
  const int* old_words;
  const int* old_tag;
  int P;                  // k_build_neigh<true>: parking rows in LDS (<= M; an atom with more candidates reports F_PARK_OVER)
};

// ---- counting sort of the owned atoms by cell key (plain keys): a by-product is first[b], the first sorted position
// of EVERY cell b (with or without atoms), which the list build reads instead of per-cell ranges ----
// (delta -1 after the scan has used the counts: the histogram is zero again for the next rebuild -- no 16 MB memset)
template <class K>
__global__ __launch_bounds__(256) void k_key_count(const K* keys, int n, int shift, int* count, int delta)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  atomicAdd(&count[(int)(keys[i] >> shift)], delta);
}
// slots inside a cell are handed out in arrival order, counting the cell's histogram entry back down to zero (the
// array needs no clearing before the next rebuild and no copy as a cursor) ...
__global__ __launch_bounds__(256) void k_key_place(const unsigned* keys, int n, int* count, const int* first, int* arrival)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned b = keys[i];
  // (one atomic per run of equal keys among the lanes: key_runs; the run's lanes take consecutive places)
  int head, len, rank;
  key_runs(b, head, len, rank);
  int base = 0;
  if (rank == 0) base = atomicSub(&count[b], len);
  base = __shfl(base, head, 64);
  arrival[first[b] + base - 1 - rank] = i;
}
// ... and then put into ascending TAG inside every cell (tags are unique): perm[new] = old.  The order of the owned
// atoms -- like that of the ghosts, (cell, tag) -- then depends on nothing but the particles themselves: the same
// system fed in another order, or arriving through another history of rebuilds, gives the same lists and the same bits.
// (base: 0 for the owned atoms; nlocal for the ghosts, whose keys / arrival entries count from the first ghost.  Two
// ghosts of one cell never carry the same tag -- images of an atom lie a box length apart -- the index breaks the tie
// all the same.)
__global__ __launch_bounds__(256) void k_key_rank(const unsigned* keys, int n, const int* first, const int* arrival,
                                                  const int* tag, int* perm, int base)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned b = keys[i];
  const int s = first[b], e = first[b + 1];
  const int ti = tag[base + i];
  int r = 0;
  for (int k = s; k < e; k++) {
    const int a = arrival[k], ta = tag[base + a];
    r += (ta < ti || (ta == ti && a < i)) ? 1 : 0;
  }
  perm[s + r] = base + i;
}

// counting sort of the ghosts by cell (row path): cell of every ghost + the histogram the scan turns into the first
// ghost-order position of every cell, which the list build needs anyway -- k_key_place / k_key_rank then order the
// ghosts by (cell, tag) like the owned atoms, instead of a 64-bit radix sort (7 launches) of (cell, tag) keys
__global__ __launch_bounds__(256) void k_ghost_cells(const double4* xr, int nlocal, int nghost, BinGrid g,
                                                     unsigned* keys, int* count, int* flags)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nghost) return;
  int lost = 0;
  const unsigned b = (unsigned)bin_of(xr[nlocal + k], g, lost);
  keys[k] = b;
  atomicAdd(&count[b], 1);
  if (lost) flags[F_LOST] = 1;
}

// The ghost count stays on the device between the ghost creation and the list build (no host round trip for it: the
// host learns it with the flags it reads behind the list build anyway).  These are k_ghost_cells / k_key_place /
// k_key_rank above for a count the kernel reads itself; launched for the most ghosts the capacity could
// hold.  A count that overflowed the capacity (F_GHOST_OVER) makes them do nothing: the host grows and repeats.
__device__ __forceinline__ int ghosts_on_device(const int* flags, int nlocal, size_t cap)
{
  const int n = flags[F_GHOST_COUNT];
  return (flags[F_GHOST_OVER] || (size_t)nlocal + (size_t)n > cap) ? 0 : n;
}
__global__ __launch_bounds__(256) void k_ghost_cells_dev(const double4* xr, int nlocal, size_t cap, BinGrid g,
                                                                unsigned* keys, int* count, int* flags)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= ghosts_on_device(flags, nlocal, cap)) return;
  int lost = 0;
  const unsigned b = (unsigned)bin_of(xr[nlocal + k], g, lost);
  keys[k] = b;
  atomicAdd(&count[b], 1);
  if (lost) flags[F_LOST] = 1;
}
__global__ __launch_bounds__(256) void k_key_place_dev(const unsigned* keys, const int* flags, int nlocal, size_t cap,
                                                              int* count, const int* first, int* arrival)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ghosts_on_device(flags, nlocal, cap)) return;
  const unsigned b = keys[i];
  arrival[first[b] + atomicSub(&count[b], 1) - 1] = i;
}
__global__ __launch_bounds__(256) void k_key_rank_dev(const unsigned* keys, const int* flags, int nlocal, size_t cap,
                                                             const int* first, const int* arrival, const int* tag, int* perm)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ghosts_on_device(flags, nlocal, cap)) return;
  const unsigned b = keys[i];
  const int s = first[b], e = first[b + 1];
  const int ti = tag[nlocal + i];
  int r = 0;
  for (int k = s; k < e; k++) {
    const int a = arrival[k], ta = tag[nlocal + a];
    r += (ta < ti || (ta == ti && a < i)) ? 1 : 0;
  }
  perm[s + r] = nlocal + i;
}

// [3P] Neighbor::build (granular criterion rsq <= (ri+rj+skin)^2) as a FULL list, with the shear
// history re-injected by partner tag (FixShearHistory)
// variant builds (tests/build_variant.sh bph -DSF_EXP_BUILD_PHASE=1): cycles the waves of k_build_neigh spend in each phase,
// summed over all launches, printed when the engine goes (SF_EXP_BUILD_PHASE in sf_dem.hip)
#ifdef SF_EXP_BUILD_PHASE
__device__ unsigned long long g_build_phase[8];
#define SF_BP(k)                                                                          \
  do {                                                                                    \
    const unsigned long long t_ = __builtin_readcyclecounter();                           \
    if ((threadIdx.x & 63) == 0) atomicAdd(&g_build_phase[(k)], t_ - bp_t_);              \
    bp_t_ = __builtin_readcyclecounter();                                                 \
  } while (0)
#else
#define SF_BP(k)
#endif
// LC: the accepted candidates of the first sweep are parked in LDS ([B.M][128] words + [B.M][128] image-code bytes of
// dynamic shared memory) instead of the scratch rows `cand`: a store to memory inside the candidate walk is waited for by
// the wait of the NEXT record loads (one counter for loads and stores), so every step of the walk paid a write round trip
#ifdef SF_EXP_BUILD_WAVES
#define SF_BUILD_ATTR __attribute__((amdgpu_waves_per_eu(SF_EXP_BUILD_WAVES, SF_EXP_BUILD_WAVES)))
#else
#define SF_BUILD_ATTR
#endif
// ROWS: the row path (plain keys: B.lb_own) -- the cell-by-cell path of the tiled / LDS-staged orderings is an instantiation of
// its own, so that neither carries the other's registers
// (Measured and removed, profiles/r06_README.md section 3: two / four lanes per atom -- the rows of the stencil split into
// contiguous blocks, the same list word for word in a chain of 1 / NL the length -- and a walk on single-precision shadow
// records, one 16-byte load per candidate, eight candidates per step: neither shortens the kernel, at 100 k grains or at 1 M.)
template <bool LC, bool ROWS = true>
__global__ __launch_bounds__(128) SF_BUILD_ATTR void k_build_neigh(BuildParams B, const double4* xr, const int* tag,
                                                     const int* cellLS, const int* cellLE,
                                                     const int* cellGS, const int* cellGE,
                                                     const int* ghost_order, const int* numneigh_old,
                                                     const int* ptag_old, const double* shear_old,
                                                     int* neigh, int* numneigh, double* shear, int* flags, int* cand,
                                                     double* xhold)
{
  // (every candidate record is read by the ~35 atoms around it: neighbouring blocks on the same XCD share them in L2)
  const int i = xcd_contiguous_block() * blockDim.x + threadIdx.x;
  if (i >= B.nlocal) return;
#ifdef SF_EXP_BUILD_PHASE
  unsigned long long bp_t_ = __builtin_readcyclecounter();
#endif
  const double4 xi = xr[i];
  // [3P] Neighbor::build: the positions the skin/2 displacement check (Neighbor::check_distance) refers to
  xhold[i] = xi.x;
  xhold[B.cap + i] = xi.y;
  xhold[2 * B.cap + i] = xi.z;
  int lost = 0;
  const int cx = bin_coord(xi.x, B.g.lo[0], B.g.inv[0], B.g.n[0], lost);
  const int cy = bin_coord(xi.y, B.g.lo[1], B.g.inv[1], B.g.n[1], lost);
  const int cz = bin_coord(xi.z, B.g.lo[2], B.g.inv[2], B.g.n[2], lost);
  // rows of the OLD list: at this atom's own index, or -- when the re-sort left the history rows where they were
  // (B.old_index) -- at the index the atom had before the sort
  const int io = B.old_index ? B.old_index[i] : i;
  const int nold = numneigh_old ? numneigh_old[io] : 0;
  // the old partner tags of this atom in registers (the re-injection below compares every accepted neighbour with
  // them; reading the rows again per neighbour was 45 % of this kernel)
  constexpr int kPT = 16;
  int pt[kPT];
  auto load_old_tags = [&]() {
    if (B.old_words) {
      int ow[kPT];
#pragma unroll
      for (int s = 0; s < kPT; s++) ow[s] = s < nold ? B.old_words[(size_t)s * B.cap + io] : 0;   // (0: no touch bit)
#pragma unroll
      for (int s = 0; s < kPT; s++) pt[s] = (ow[s] & kTouchBit) ? B.old_tag[neigh_index(ow[s], B.roots)] : -1;
    } else {
#pragma unroll
      for (int s = 0; s < kPT; s++) pt[s] = s < nold ? ptag_old[(size_t)s * B.cap + io] : -1;
    }
  };
  // (row path: loaded BEHIND the candidate walk, which does not look at them -- sixteen registers less while the walk's
  // record loads are in flight)
  if (!ROWS) load_old_tags();
  constexpr int BD = 128;   // (= the launch's block size)
  extern __shared__ int sf_build_lds[];
  int* const lc_w = sf_build_lds + threadIdx.x;                                                      // parked word s: lc_w[s * BD]
  unsigned char* const lc_c = reinterpret_cast<unsigned char*>(sf_build_lds + B.P * BD) + threadIdx.x;   // its image code
  const int park_rows = LC ? B.P : B.M;
  int n = 0;
  const int T = B.g.tile, E = T + 2;
  const int tx = cx / T, ty = cy / T, tz = cz / T;
  const int* eo = B.eoff ? B.eoff + (size_t)(tx + B.g.nt[0] * (ty + B.g.nt[1] * tz)) * (E * E * E) : nullptr;
  const int R = B.g.stencil;
  // cells are visited with the fastest key dimension innermost (x, or z in the x-slowest order), so consecutive
  // slots of an atom -- and the same slot of adjacent lanes -- point at consecutive atoms in memory
  const int co = B.g.xslow ? cx : cz, no = B.g.xslow ? B.g.n[0] : B.g.n[2];
  const int ci = B.g.xslow ? cz : cx, ni = B.g.xslow ? B.g.n[2] : B.g.n[0];
  // distance test of one candidate
  // (no granular criterion: ri + rj + -inf never exceeds the absolute cutoff -- one max instead of a select per candidate)
  const double skinv = B.skin_gran >= 0.0 ? B.skin_gran : -INFINITY;
  // (sx, sy, sz: the box lengths a candidate found AROUND the box is shifted by -- BinGrid::wrap; xj + shift is the position
  // LAMMPS gives the ghost copy)
  double sx = 0.0, sy = 0.0, sz = 0.0;
  auto in_range = [&](const int j, const double4 xj) {
    const double dx = xi.x - (xj.x + sx), dy = xi.y - (xj.y + sy), dz = xi.z - (xj.z + sz);
    const double rsq = dx * dx + dy * dy + dz * dz;
    const double cut = fmax(xi.w + xj.w + skinv, B.cut_lub);
    return j != i && rsq <= cut * cut;
  };
  // neighbour j enters the list; pos = its position in the tile's staged copy (LDS kernel only)
  // partner tag of the old list that equals tj (-1: the pair did not touch)
  auto find_old = [&](const int tj) {
    int found = -1;
#pragma unroll
    for (int s = 0; s < kPT; s++)
      if (pt[s] == tj) found = s;             // tags are unique: at most one match
    if (found < 0)
      for (int s = kPT; s < nold; s++) {
        int ts;
        if (B.old_words) {
          const int w = B.old_words[(size_t)s * B.cap + io];
          ts = (w & kTouchBit) ? B.old_tag[neigh_index(w, B.roots)] : -1;
        } else
          ts = ptag_old[(size_t)s * B.cap + io];
        if (ts == tj) {
          found = s;
          break;
        }
      }
    return found;
  };
  // row path: slot of the next touching / next non-touching neighbour (touching ones first, see the second sweep)
  int slot_touch = -1, slot_free = -1;
  int found_known = -2;   // >= -1: the old slot of the pair was looked up before (touch-first placement)
  int code_known = kNoShift;   // wrapped stencil: which periodic image of j the candidate is (kNoShift: j itself)
  auto accept = [&](const int j, const int tj, const int pos) {
    const int found = n >= B.M ? -1 : found_known >= -1 ? found_known : find_old(tj);
    const int dst = slot_touch < 0 ? n : (found >= 0 ? slot_touch++ : slot_free++);
    if (n < B.M && dst < B.M) {
      int entry = j;
      // history owner of the pair: the lower index of two atoms of this GPU; pairs with a periodic image or with a
      // ghost of another GPU keep a copy on each side (the reference's newton-off treatment of owned-ghost pairs)
      bool own = j > i || B.two_copies;
      if (B.roots) {
        int code = code_known;
        if (code != kNoShift) own = true;   // (an image found around the box: a copy on each side, like every owned-ghost pair)
        if (j >= B.nlocal) {
          const int r = B.gsrc[j];
          if (r >= 0) {   // periodic image made on this GPU: refer to its root + which image it is
            entry = r;
            const int ix = (int)rint(B.gshift[j] * B.inv_prd[0]);
            const int iy = (int)rint(B.gshift[B.cap + j] * B.inv_prd[1]);
            const int iz = (int)rint(B.gshift[2 * B.cap + j] * B.inv_prd[2]);
            code = (ix + 1) + 3 * (iy + 1) + 9 * (iz + 1);
            own = true;
          }
        }
        entry |= code << kIdxBits;
      }
      if (own) entry |= kOwnBit;
      if (found >= 0) {
        // (the history of a slot is read only while its touch bit is set, and a contact that forms later starts from
        // zero in registers: the slots of neighbours that do not touch are left as they are -- two thirds of the
        // history stores of a loose bed)
        entry |= kTouchBit;
        size_t ob = (size_t)(3 * found) * B.cap + io;
        double sgn = 1.0;
        if (B.old_words) {
          // (the old list in place: a partner side -- root mode only -- reads the owner's copy, negated: FixShearHistory's
          // sign-flipped copy for j, as k_partner_tags stages it)
          const int wf = B.old_words[(size_t)found * B.cap + io];
          if (!(wf & kOwnBit)) {
            ob = (size_t)(3 * ((wf >> kIdxBits) & 31)) * B.cap + neigh_index(wf, B.roots);
            sgn = -1.0;
          }
        }
        const double sx = shear_old[ob], sy = shear_old[ob + B.cap], sz = shear_old[ob + 2 * B.cap];
        const size_t nb = (size_t)(3 * dst) * B.cap + i;
        shear[nb] = sgn * sx;
        shear[nb + B.cap] = sgn * sy;
        shear[nb + 2 * B.cap] = sgn * sz;
      }
      neigh[(size_t)dst * B.cap + i] = entry;
      if (eo) B.nloc[(size_t)dst * B.cap + i] = (unsigned short)pos;
    }
    n++;
  };
  auto candidate = [&](const int j, const int pos) {
    if (in_range(j, xr[j])) accept(j, tag[j], pos);
  };
  SF_BP(0);   // prologue: own record, cell, the old partner tags
  int n_total = 0;   // row path: accepted candidates of the first sweep (may exceed the slots: overflow report)
  // Row path, first sweep: an accepted candidate is parked in the scratch rows `cand`, in candidate order; the old-list
  // look-up (tag gather + comparison with the partner tags held in registers) waits for the second sweep, where every
  // lane of a wave is at the same slot.
  constexpr int kFoundUnknown = 127;
  const bool tf = B.touch_first && nold > 0;
  int* cand_next = cand + i;   // (a running pointer: the row stride is added per accepted candidate, not multiplied)
  int row_code = kNoShift;      // wrapped stencil: image code of the row being walked (lanes near a periodic face)
  bool park_codes = false;      // ... which park it next to every accepted candidate (B.nloc: unused without LDS staging)
  auto note = [&](const int j) {
    if (n_total < park_rows) {
      if (LC) {
        lc_w[n_total * BD] = j;
        if (park_codes) lc_c[n_total * BD] = (unsigned char)row_code;
      } else {
        *cand_next = j;
        cand_next += B.cap;
        if (park_codes) B.nloc[(size_t)n_total * B.cap + i] = (unsigned short)row_code;
      }
    }
    n_total++;
  };
  auto parked = [&](const int s) { return LC ? lc_w[s * BD] : cand[(size_t)s * B.cap + i]; };
  if constexpr (ROWS) {
    // Plain (non-tiled) keys: the 2R+1 cells of one stencil row have consecutive keys, so their atoms are ONE
    // contiguous range of the sorted array, [lb(first cell), lb(last cell + 1)) -- two loads per row instead of one
    // dependent load + loop per cell, requested one row ahead.  The records of a row are consecutive in memory: four
    // are loaded at once, then tested and entered in order.
    const int W = 2 * R + 1;
    const int bi0 = ci - R < 0 ? 0 : ci - R, bi1 = ci + R >= ni ? ni - 1 : ci + R;
    const int n1 = B.g.n[1], nin1 = ni * n1;
    // Wrapped stencil (BinGrid::wrap, ghost-free build; x fastest: inner = x, outer = z).  A row whose y or z cell lies
    // beyond a periodic face is the row of the cell on the other side of the box, its atoms shifted by the box length;
    // the x range of a row that crosses a periodic x face is TWO ranges -- the cells inside the box (segment 0) and the
    // cells around the box (segment 1, of the lanes within R cells of that face only).
    const bool wrap_i = B.g.wrap[0] != 0, wrap_y = B.g.wrap[1] != 0, wrap_o = B.g.wrap[2] != 0;
    int seg1_lo = 0, seg1_hi = -1;      // cells of segment 1 (empty unless this lane's x range crosses a periodic face)
    double seg1_shift = 0.0;
    if (wrap_i && ci - R < 0) {
      seg1_lo = ci - R + ni;
      seg1_hi = ni - 1;
      seg1_shift = -B.prd[0];
    } else if (wrap_i && ci + R >= ni) {
      seg1_lo = 0;
      seg1_hi = ci + R - ni;
      seg1_shift = B.prd[0];
    }
    // (lanes whose stencil reaches around the box park the image code of every accepted candidate next to its index)
    const bool near_face = (wrap_i && (ci < R || ci + R >= ni)) || (wrap_y && (cy < R || cy + R >= n1)) ||
                           (wrap_o && (co < R || co + R >= no));
    const int nseg = (wrap_i && __ballot(seg1_hi >= seg1_lo)) ? 2 : 1;   // (wave-uniform)
    park_codes = near_face;
    for (int pass = 0; pass < 2; pass++) {
      const int* lb = pass ? B.lb_ghost : B.lb_own;   // owned atoms first, then ghosts in their (cell, tag) order
      if (!lb) break;
      // row r = (segment, ro, ry): its range of the sorted array and the shift of its atoms
      auto row_range = [&](const int seg, const int ro, const int ry, int& lo, int& hi, double& rsx, double& rsy,
                           double& rsz) {
        lo = hi = 0;
        rsx = rsy = rsz = 0.0;
        if (seg >= nseg || ro >= W) return;
        int bo = co - R + ro, by = cy - R + ry;
        if (wrap_o) {
          if (bo < 0) { bo += no; rsz = -B.prd[2]; }
          else if (bo >= no) { bo -= no; rsz = B.prd[2]; }
        } else if ((unsigned)bo >= (unsigned)no) return;
        if (wrap_y) {
          if (by < 0) { by += n1; rsy = -B.prd[1]; }
          else if (by >= n1) { by -= n1; rsy = B.prd[1]; }
        } else if ((unsigned)by >= (unsigned)n1) return;
        const int c0 = seg ? seg1_lo : bi0, c1 = seg ? seg1_hi : bi1;
        if (c1 < c0) return;
        if (seg) rsx = seg1_shift;
        const int key = c0 + ni * by + nin1 * bo;
        lo = lb[key];
        hi = lb[key + (c1 - c0 + 1)];
      };
      // rows in key order, one loop over r = (seg * W + ro) * W + ry (three nested loops kept 18 more registers alive: four
      // instead of five waves per SIMD)
      const int nrow = nseg * W * W;
      int seg_n = 0, ro_n = 0, ry_n = 0;   // coordinates of the row whose range is requested next
      int nlo = 0, nhi = 0;
      double nsx = 0.0, nsy = 0.0, nsz = 0.0;
      row_range(seg_n, ro_n, ry_n, nlo, nhi, nsx, nsy, nsz);
      for (int r = 0; r < nrow; r++) {
        const int lo = nlo, hi = nhi;
        sx = nsx;
        sy = nsy;
        sz = nsz;
        // (the range of the next row, requested one row ahead)
        if (++ry_n == W) {
          ry_n = 0;
          if (++ro_n == W) {
            ro_n = 0;
            seg_n++;
          }
        }
        if (r + 1 < nrow) row_range(seg_n, ro_n, ry_n, nlo, nhi, nsx, nsy, nsz);
        // image code of this row's atoms ((ix + 1) + 3 (iy + 1) + 9 (iz + 1), ix = shift / box length)
        if (near_face)
          row_code = ((sx < 0.0 ? 0 : sx > 0.0 ? 2 : 1)) + 3 * (sy < 0.0 ? 0 : sy > 0.0 ? 2 : 1) +
                     9 * (sz < 0.0 ? 0 : sz > 0.0 ? 2 : 1);
        if (pass) {
          for (int k = lo; k < hi; k++) {
            const int j = ghost_order[k];
            if (in_range(j, xr[j])) note(j);
          }
          continue;
        }
        // four records per step, loaded unconditionally (a lane whose row is shorter reads its last record again:
        // plain 16-byte loads instead of a branch around every 8 bytes), tested and entered in order
        for (int k = lo; k < hi; k += 4) {
          const int last = hi - 1;
          const int k1 = min(k + 1, last), k2 = min(k + 2, last), k3 = min(k + 3, last);
          const double4 x0 = xr[k], x1 = xr[k1], x2 = xr[k2], x3 = xr[k3];
          if (in_range(k, x0)) note(k);
          if (k + 1 < hi && in_range(k + 1, x1)) note(k + 1);
          if (k + 2 < hi && in_range(k + 2, x2)) note(k + 2);
          if (k + 3 < hi && in_range(k + 3, x3)) note(k + 3);
        }
      }
    }
    sx = sy = sz = 0.0;
    SF_BP(1);   // candidate walk
    load_old_tags();
    SF_BP(5);   // old partner tags
    // second sweep, slot by slot: every lane of the wave is at the same row of the slot-major arrays, so the history
    // re-injection reads and the neigh/shear stores are coalesced even when the lanes found their neighbours at
    // different moments of the candidate walk (disordered beds)
    // B.touch_first (loose beds: far fewer touching than listed neighbours): the neighbours that touched in the old
    // list (their history is re-injected) take the FIRST slots of the row, in candidate order, the others follow.  The
    // sub-step kernel then evaluates the contact law in the first slots, where most lanes of a wave touch, and skips
    // it wave-wide in the rest, instead of running it in every slot for the few lanes that touch there (loose bed:
    // 236 -> 189 us per sub-step at 1 M grains).  In an ordered bed the same shuffle costs the lane-to-lane
    // regularity of the slots -- slot s of adjacent lanes = adjacent atoms -- that the gathers coalesce on (+35 %
    // there), hence the switch (DemEngine::bin_and_build).
    const int nacc = n_total < park_rows ? n_total : park_rows;
    if (LC && n_total > park_rows) atomicMax(&flags[F_PARK_OVER], n_total);   // (the host builds again with more rows)
    // (overflowing rows are rebuilt with more slots: what was dropped does not matter)
    // Touch-first needs to know how many of the accepted candidates touched before it can place any of them: one more
    // pass over the parked candidates looks each of them up in the old list (kept in the word: j | (old slot + 1) << 25)
    // and counts.  Doing the look-up inside the candidate walk instead -- where a wave runs it once per candidate
    // POSITION of any lane, ~100 times, not once per accepted candidate, ~13 times -- cost a third of this kernel.
    int nt = 0;
    if (tf) {
      int wn = nacc > 0 ? parked(0) : 0;
      int wn2 = nacc > 1 ? parked(1) : 0;
      int tn = nacc > 0 ? tag[wn] : 0;
      for (int s = 0; s < nacc; s++) {
        const int j = wn, tj = tn;
        wn = wn2;
        if (s + 1 < nacc) tn = tag[wn];
        if (s + 2 < nacc) wn2 = parked(s + 2);
        const int f = find_old(tj);
        if (f >= 0) nt++;
        const int wf = j | ((f + 1 >= kFoundUnknown ? kFoundUnknown : f + 1) << kIdxBits);
        if (LC) lc_w[s * BD] = wf;
        else cand[(size_t)s * B.cap + i] = wf;
      }
      slot_touch = 0;
      slot_free = nt;
    }
    SF_BP(2);   // touch-first look-up pass
    n = 0;
    // the candidate word two slots ahead, its tag one slot ahead: neither load waits for the other inside an iteration
    int wn = nacc > 0 ? parked(0) : 0;
    int wn2 = nacc > 1 ? parked(1) : 0;
    int tn = (nacc > 0 && !tf) ? tag[wn & kIdxMask] : 0;
    for (int s = 0; s < nacc; s++) {
      const int w = wn, tj = tn;
      wn = wn2;
      if (s + 1 < nacc && !tf) tn = tag[wn & kIdxMask];
      if (s + 2 < nacc) wn2 = parked(s + 2);
      const int j = w & kIdxMask, fcode = (w >> kIdxBits) & 127;
      found_known = !tf || fcode == kFoundUnknown ? -2 : fcode - 1;
      code_known = !park_codes ? kNoShift : LC ? (int)(lc_c[s * BD] & 31) : (int)B.nloc[(size_t)s * B.cap + i];
      accept(j, tf && found_known == -2 ? tag[j] : tj, 0);
    }
    n = n_total;
    SF_BP(3);   // second sweep: history re-injection, list and history stores
  } else {
    const int4* cells = reinterpret_cast<const int4*>(cellLS);   // {owned start, end, ghost start, end} per cell
    for (int bo = co - R; bo <= co + R; bo++) {
      if (bo < 0 || bo >= no) continue;
      for (int by = cy - R; by <= cy + R; by++) {
        if (by < 0 || by >= B.g.n[1]) continue;
        for (int bi = ci - R; bi <= ci + R; bi++) {
          if (bi < 0 || bi >= ni) continue;
          const int bx = B.g.xslow ? bo : bi, bz = B.g.xslow ? bi : bo;
          const int b = bin_key(B.g, bx, by, bz);
          const int ebase = eo ? eo[((bz - (tz * T - 1)) * E + (by - (ty * T - 1))) * E + (bx - (tx * T - 1))] : 0;   // stencil 1 only
          const int4 cb = cells[b];
          for (int k = cb.x; k < cb.y; k++) candidate(k, ebase + (k - cb.x));
          for (int k = cb.z; k < cb.w; k++) candidate(ghost_order[k], ebase + (cb.y - cb.x) + (k - cb.z));
        }
      }
    }
  }
  if (n > B.M) {
    atomicMax(&flags[F_NEIGH_OVER], n);
    n = B.M;
  }
  numneigh[i] = n;
  // F_MAXNEIGH: one atomic per wave (a same-address atomic costs ~11 ns at the memory side: per atom that would be
  // 11 ms, per wave it is 0.17 ms spread over the kernel's 0.3-0.45 ms and behind other work)
  int m = n;
  const unsigned long long act = __ballot(1);   // (the last wave: lanes past the last atom have left)
  const int lane = threadIdx.x & 63;
  for (int off = 32; off > 0; off >>= 1) {
    const int o = __shfl_xor(m, off, 64);
    if ((act >> (lane ^ off)) & 1ull) m = max(m, o);
  }
  // (... and only by a wave that would raise it: the atomics of one address are served one after the other, ~11 ns each, and
  // the waves of a bed that fits the GPU in one round all end together -- 1 570 waves of a 100 k bed were a 17 us tail on a
  // 67 us kernel; after the first few the plain load already shows a value no wave exceeds)
  if (lane == __ffsll((long long)act) - 1 && m > __atomic_load_n(&flags[F_MAXNEIGH], __ATOMIC_RELAXED))
    atomicMax(&flags[F_MAXNEIGH], m);
  SF_BP(4);   // counts
}

// k_build_neigh with FOUR lanes per atom, the four of them testing four CONSECUTIVE records of a row per step (single domain,
// row path, no ghost pass, candidates parked in LDS, the old list in place or staged).  What the walk pays for is the lines
// its load instructions touch (profiles/r06_README.md section 3): with one lane per atom the 64 lanes of an instruction stand
// in ~24 cells and read ~18 lines; here 16 atoms stand in ~6 cells, the four lanes of an atom read one line, and 64 tests take
// two instructions that touch ~7 lines each.  The accepted candidates of a step enter the atom's LDS column in record order
// (their count below the lane inside the quad), the slots of the second sweep are handed out the same way: the list is the
// one k_build_neigh<true> builds, word for word.
// LQ: lanes per atom (2, 4 or 8: a loose bed's rows of full-cutoff cells hold ~8 records, a packed bed's rows of half-cutoff
// cells 2-3).
template <int LQ>
__global__ __launch_bounds__(128) void k_build_neigh_quad(BuildParams B, const double4* xr, const int* tag,
                                                          const int* numneigh_old, const int* ptag_old,
                                                          const double* shear_old, int* neigh, int* numneigh,
                                                          double* shear, int* flags, double* xhold)
{
  constexpr int AB = 128 / LQ;   // atoms per block
  const int a = threadIdx.x / LQ, u = threadIdx.x % LQ;
  const int i = xcd_contiguous_block() * AB + a;
  if (i >= B.nlocal) return;   // (whole quads leave)
  const int wl = threadIdx.x & 63;
  const double4 xi = xr[i];
  if (u == 0) {
    xhold[i] = xi.x;
    xhold[B.cap + i] = xi.y;
    xhold[2 * B.cap + i] = xi.z;
  }
  int lost = 0;
  const int cx = bin_coord(xi.x, B.g.lo[0], B.g.inv[0], B.g.n[0], lost);
  const int cy = bin_coord(xi.y, B.g.lo[1], B.g.inv[1], B.g.n[1], lost);
  const int cz = bin_coord(xi.z, B.g.lo[2], B.g.inv[2], B.g.n[2], lost);
  const int io = B.old_index ? B.old_index[i] : i;
  const int nold = numneigh_old ? numneigh_old[io] : 0;
  extern __shared__ int sf_build_lds[];
  int* const lc_w = sf_build_lds + a;                                                          // parked word s: lc_w[s * AB]
  unsigned char* const lc_c = reinterpret_cast<unsigned char*>(sf_build_lds + B.P * AB) + a;   // its image code
  const int park_rows = B.P;
  const int R = B.g.stencil, W = 2 * R + 1;
  const int co = cz, no = B.g.n[2], ci = cx, ni = B.g.n[0], n1 = B.g.n[1], nin1 = ni * n1;   // (x fastest: never the x-slowest order)
  const double skinv = B.skin_gran >= 0.0 ? B.skin_gran : -INFINITY;
  // the accept bits of this lane's quad in a wave-wide ballot
  auto quad_bits = [&](const bool b) { return (unsigned)(__ballot(b) >> (wl & ~(LQ - 1))) & ((1u << LQ) - 1u); };
  const unsigned below = (1u << u) - 1u;
  // ---- the walk: rows in key order, four records per step, one per lane ----
  const int bi0 = ci - R < 0 ? 0 : ci - R, bi1 = ci + R >= ni ? ni - 1 : ci + R;
  const bool wrap_i = B.g.wrap[0] != 0, wrap_y = B.g.wrap[1] != 0, wrap_o = B.g.wrap[2] != 0;
  int seg1_lo = 0, seg1_hi = -1;
  double seg1_shift = 0.0;
  if (wrap_i && ci - R < 0) {
    seg1_lo = ci - R + ni;
    seg1_hi = ni - 1;
    seg1_shift = -B.prd[0];
  } else if (wrap_i && ci + R >= ni) {
    seg1_lo = 0;
    seg1_hi = ci + R - ni;
    seg1_shift = B.prd[0];
  }
  const bool park_codes = (wrap_i && (ci < R || ci + R >= ni)) || (wrap_y && (cy < R || cy + R >= n1)) ||
                          (wrap_o && (co < R || co + R >= no));
  const int nseg = (wrap_i && __ballot(seg1_hi >= seg1_lo)) ? 2 : 1;   // (wave-uniform)
  const int* const lb = B.lb_own;
  auto row_range = [&](const int seg, const int ro, const int ry, int& lo, int& hi, double& rsx, double& rsy, double& rsz) {
    lo = hi = 0;
    rsx = rsy = rsz = 0.0;
    if (seg >= nseg || ro >= W) return;
    int bo = co - R + ro, by = cy - R + ry;
    if (wrap_o) {
      if (bo < 0) { bo += no; rsz = -B.prd[2]; }
      else if (bo >= no) { bo -= no; rsz = B.prd[2]; }
    } else if ((unsigned)bo >= (unsigned)no) return;
    if (wrap_y) {
      if (by < 0) { by += n1; rsy = -B.prd[1]; }
      else if (by >= n1) { by -= n1; rsy = B.prd[1]; }
    } else if ((unsigned)by >= (unsigned)n1) return;
    const int c0 = seg ? seg1_lo : bi0, c1 = seg ? seg1_hi : bi1;
    if (c1 < c0) return;
    if (seg) rsx = seg1_shift;
    const int key = c0 + ni * by + nin1 * bo;
    lo = lb[key];
    hi = lb[key + (c1 - c0 + 1)];
  };
  int n_total = 0;   // accepted candidates of the atom (the same number in its four lanes)
  {
    const int nrow = nseg * W * W;
    int seg_n = 0, ro_n = 0, ry_n = 0;
    int nlo = 0, nhi = 0;
    double nsx = 0.0, nsy = 0.0, nsz = 0.0;
    row_range(seg_n, ro_n, ry_n, nlo, nhi, nsx, nsy, nsz);
    for (int r = 0; r < nrow; r++) {
      const int lo = nlo, hi = nhi;
      const double sx = nsx, sy = nsy, sz = nsz;
      if (++ry_n == W) {
        ry_n = 0;
        if (++ro_n == W) {
          ro_n = 0;
          seg_n++;
        }
      }
      if (r + 1 < nrow) row_range(seg_n, ro_n, ry_n, nlo, nhi, nsx, nsy, nsz);
      const int row_code = ((sx < 0.0 ? 0 : sx > 0.0 ? 2 : 1)) + 3 * (sy < 0.0 ? 0 : sy > 0.0 ? 2 : 1) +
                           9 * (sz < 0.0 ? 0 : sz > 0.0 ? 2 : 1);
      for (int k = lo; k < hi; k += LQ) {
        const int j = k + u < hi ? k + u : hi - 1;   // (a lane past the row's end reads its last record again: no new line)
        const double4 xj = xr[j];
        const double dx = xi.x - (xj.x + sx), dy = xi.y - (xj.y + sy), dz = xi.z - (xj.z + sz);
        const double rsq = dx * dx + dy * dy + dz * dz;
        const double cut = fmax(xi.w + xj.w + skinv, B.cut_lub);
        const bool acc = k + u < hi && j != i && rsq <= cut * cut;
        const unsigned q4 = quad_bits(acc);
        if (acc) {
          const int pos = n_total + __popc(q4 & below);
          if (pos < park_rows) {
            lc_w[pos * AB] = j;
            if (park_codes) lc_c[pos * AB] = (unsigned char)row_code;
          }
        }
        n_total += __popc(q4);
      }
    }
  }
  const int nacc = n_total < park_rows ? n_total : park_rows;
  if (u == 0 && n_total > park_rows) atomicMax(&flags[F_PARK_OVER], n_total);
  // ---- the old partner tags (every lane of the quad holds all of them) ----
  constexpr int kPT = 16;
  int pt[kPT];
  auto old_tag_at = [&](const int s) {
    if (B.old_words) {
      const int w = B.old_words[(size_t)s * B.cap + io];
      return (w & kTouchBit) ? B.old_tag[neigh_index(w, B.roots)] : -1;
    }
    return ptag_old[(size_t)s * B.cap + io];
  };
  if (LQ == 4) {
    // each lane of the quad fetches four of the sixteen (slots 4 u .. 4 u + 3) and the quad exchanges them lane to lane
    // (quad_perm broadcasts: register moves) -- a quarter of the loads and tag gathers
    int mine[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int sl = 4 * u + k;
      mine[k] = sl < nold ? old_tag_at(sl) : -1;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      pt[k] = __builtin_amdgcn_update_dpp(0, mine[k], 0x00, 0xF, 0xF, false);        // quad_perm:[0,0,0,0]
      pt[4 + k] = __builtin_amdgcn_update_dpp(0, mine[k], 0x55, 0xF, 0xF, false);    // quad_perm:[1,1,1,1]
      pt[8 + k] = __builtin_amdgcn_update_dpp(0, mine[k], 0xAA, 0xF, 0xF, false);    // quad_perm:[2,2,2,2]
      pt[12 + k] = __builtin_amdgcn_update_dpp(0, mine[k], 0xFF, 0xF, 0xF, false);   // quad_perm:[3,3,3,3]
    }
  } else if (LQ == 2) {
    // (two lanes per atom: eight each, exchanged between the lanes of the pair)
    int mine[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int sl = 8 * u + k;
      mine[k] = sl < nold ? old_tag_at(sl) : -1;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
      pt[k] = __builtin_amdgcn_update_dpp(0, mine[k], 0xA0, 0xF, 0xF, false);       // quad_perm:[0,0,2,2]
      pt[8 + k] = __builtin_amdgcn_update_dpp(0, mine[k], 0xF5, 0xF, 0xF, false);   // quad_perm:[1,1,3,3]
    }
  } else if (B.old_words) {
    int ow[kPT];
#pragma unroll
    for (int s = 0; s < kPT; s++) ow[s] = s < nold ? B.old_words[(size_t)s * B.cap + io] : 0;
#pragma unroll
    for (int s = 0; s < kPT; s++) pt[s] = (ow[s] & kTouchBit) ? B.old_tag[neigh_index(ow[s], B.roots)] : -1;
  } else {
#pragma unroll
    for (int s = 0; s < kPT; s++) pt[s] = s < nold ? ptag_old[(size_t)s * B.cap + io] : -1;
  }
  auto find_old = [&](const int tj) {
    int found = -1;
#pragma unroll
    for (int s = 0; s < kPT; s++)
      if (pt[s] == tj) found = s;
    if (found < 0)
      for (int s = kPT; s < nold; s++)
        if (old_tag_at(s) == tj) {
          found = s;
          break;
        }
    return found;
  };
  // ---- second sweep: lane u takes the parked candidates u, u + 4, ...; their slots are their positions (candidate order) or,
  // touching neighbours first, the count of touching / other candidates before them ----
  const bool tf = B.touch_first && nold > 0;
  const int nstep = (nacc + LQ - 1) / LQ;
  int nt = 0;   // touching among the accepted (touch-first)
  constexpr int kFoundUnknown = 127;
  if (tf) {
    // (the look-up is kept in the parked word, j | (old slot + 1) << 25, for the sweep below)
    for (int t = 0; t < nstep; t++) {
      const int s = LQ * t + u;
      bool touch = false;
      if (s < nacc) {
        const int j = lc_w[s * AB];
        const int f = find_old(tag[j]);
        touch = f >= 0;
        lc_w[s * AB] = j | ((f + 1 >= kFoundUnknown ? kFoundUnknown : f + 1) << kIdxBits);
      }
      nt += __popc(quad_bits(touch));
    }
  }
  int tbase = 0, fbase = nt;
  for (int t = 0; t < nstep; t++) {
    const int s = LQ * t + u;
    const bool valid = s < nacc;
    int j = 0, found = -1;
    if (valid) {
      const int w = lc_w[s * AB];
      j = w & kIdxMask;
      const int fcode = (w >> kIdxBits) & 127;
      found = (!tf || fcode == kFoundUnknown) ? find_old(tag[j]) : fcode - 1;
    }
    int dst = s;
    if (tf) {
      const unsigned qt = quad_bits(valid && found >= 0), qv = quad_bits(valid);
      const unsigned qf = qv & ~qt;
      dst = found >= 0 ? tbase + __popc(qt & below) : fbase + __popc(qf & below);
      tbase += __popc(qt);
      fbase += __popc(qf);
    }
    if (valid && dst < B.M) {
      int entry = j;
      bool own = j > i || B.two_copies;
      if (B.roots) {
        const int code = park_codes ? (int)lc_c[s * AB] : kNoShift;
        if (code != kNoShift) own = true;   // (an image found around the box: a copy on each side, like every owned-ghost pair)
        entry |= code << kIdxBits;
      }
      if (own) entry |= kOwnBit;
      if (found >= 0) {
        entry |= kTouchBit;
        size_t ob = (size_t)(3 * found) * B.cap + io;
        double sgn = 1.0;
        if (B.old_words) {
          const int wf = B.old_words[(size_t)found * B.cap + io];
          if (!(wf & kOwnBit)) {
            ob = (size_t)(3 * ((wf >> kIdxBits) & 31)) * B.cap + neigh_index(wf, B.roots);
            sgn = -1.0;
          }
        }
        const double hx = shear_old[ob], hy = shear_old[ob + B.cap], hz = shear_old[ob + 2 * B.cap];
        const size_t nb = (size_t)(3 * dst) * B.cap + i;
        shear[nb] = sgn * hx;
        shear[nb + B.cap] = sgn * hy;
        shear[nb + 2 * B.cap] = sgn * hz;
      }
      neigh[(size_t)dst * B.cap + i] = entry;
    }
  }
  int n = n_total;
  if (n > B.M) {
    if (u == 0) atomicMax(&flags[F_NEIGH_OVER], n);
    n = B.M;
  }
  if (u == 0) numneigh[i] = n;
  int m = n;
  const unsigned long long act = __ballot(1);
  for (int off = 32; off > 0; off >>= 1) {
    const int o = __shfl_xor(m, off, 64);
    if ((act >> (wl ^ off)) & 1ull) m = max(m, o);
  }
  if (wl == __ffsll((long long)act) - 1 && m > __atomic_load_n(&flags[F_MAXNEIGH], __ATOMIC_RELAXED))
    atomicMax(&flags[F_MAXNEIGH], m);
}

// ---- LDS staging tables: which atoms a tile's workgroup copies into LDS, bin by bin ----
__global__ __launch_bounds__(128) void k_tile_stage_count(BinGrid g, const int* cellLS, const int* cellLE,
                                                          const int* cellGS, const int* cellGE, int ntiles,
                                                          int* count, int* flags)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntiles) return;
  const int T = g.tile, E = T + 2;
  const int tx = t % g.nt[0], ty = (t / g.nt[0]) % g.nt[1], tz = t / (g.nt[0] * g.nt[1]);
  int total = 0;
  for (int ez = 0; ez < E; ez++) {
    const int bz = tz * T - 1 + ez;
    if (bz < 0 || bz >= g.n[2]) continue;
    for (int ey = 0; ey < E; ey++) {
      const int by = ty * T - 1 + ey;
      if (by < 0 || by >= g.n[1]) continue;
      for (int ex = 0; ex < E; ex++) {
        const int bx = tx * T - 1 + ex;
        if (bx < 0 || bx >= g.n[0]) continue;
        const int b = bin_key(g, bx, by, bz);
        total += (cellLE[4 * b] - cellLS[4 * b]) + (cellGE[4 * b] - cellGS[4 * b]);
      }
    }
  }
  count[t] = total;
  atomicMax(&flags[F_STAGE_MAX], total);
}

__global__ __launch_bounds__(128) void k_tile_stage_fill(BinGrid g, const int* cellLS, const int* cellLE,
                                                         const int* cellGS, const int* cellGE,
                                                         const int* ghost_order, int ntiles, const int* start,
                                                         int* eoff, int* stage_idx)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntiles) return;
  const int T = g.tile, E = T + 2;
  const int tx = t % g.nt[0], ty = (t / g.nt[0]) % g.nt[1], tz = t / (g.nt[0] * g.nt[1]);
  int run = 0;
  int* out = stage_idx + start[t];
  for (int ez = 0; ez < E; ez++) {
    const int bz = tz * T - 1 + ez;
    for (int ey = 0; ey < E; ey++) {
      const int by = ty * T - 1 + ey;
      for (int ex = 0; ex < E; ex++) {
        const int bx = tx * T - 1 + ex;
        eoff[(size_t)t * (E * E * E) + (ez * E + ey) * E + ex] = run;
        if (bz < 0 || bz >= g.n[2] || by < 0 || by >= g.n[1] || bx < 0 || bx >= g.n[0]) continue;
        const int b = bin_key(g, bx, by, bz);
        for (int k = cellLS[4 * b]; k < cellLE[4 * b]; k++) out[run++] = k;
        for (int k = cellGS[4 * b]; k < cellGE[4 * b]; k++) out[run++] = ghost_order[k];
      }
    }
  }
}

}  // namespace sf
