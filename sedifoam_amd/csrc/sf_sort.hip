// sf_sort.hip -- key/value radix sorts used only on neighbour rebuilds (rocPRIM device primitives;
// kept in their own translation unit because the header is heavy to compile).
#include <cstring>

#include <cstdint>
#include <cstdlib>
#include <rocprim/rocprim.hpp>

#include "sf_dem.h"

namespace sf {

template <class K>
static void sort_pairs(void*& tmp, size_t& tmp_bytes, K* keys_in, K* keys_out, int* vals_in, int* vals_out, int n,
                       int end_bit, hipStream_t s)
{
  if (n <= 0) return;
  size_t need = 0;
  SF_HIP(rocprim::radix_sort_pairs(nullptr, need, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u,
                                   (unsigned)end_bit, s));
  if (need > tmp_bytes) {
    if (tmp) {
      SF_HIP(hipStreamSynchronize(s));
      SF_HIP(hipFree(tmp));
    }
    tmp_bytes = need + need / 4 + 4096;
    SF_HIP(hipMalloc(&tmp, tmp_bytes));
  }
  size_t avail = tmp_bytes;
  SF_HIP(rocprim::radix_sort_pairs(tmp, avail, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u,
                                   (unsigned)end_bit, s));
}

// Exclusive scan of a large int table (the cell histograms of a rebuild: ~4 M entries, mostly zeros, twice per rebuild):
// partial sums per 4096-entry tile, one workgroup over the tile sums, then every tile rescanned with its offset --
// 48 MB of traffic in three plain launches (~17 us) against rocPRIM's decoupled look-back scan + its state-initialising
// launch (~33 us for this size on MI355X).  in != out.
namespace {
constexpr int kScanThreads = 256, kScanItems = 16, kScanTile = kScanThreads * kScanItems;

__device__ __forceinline__ int scan_block_sum(int v, int* ws)
{
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) ws[w] = v;
  __syncthreads();
  return ws[0] + ws[1] + ws[2] + ws[3];
}

__device__ __forceinline__ void scan_load_tile(const int* in, int n, long long base, int (&x)[kScanItems])
{
  const long long first = base + (long long)threadIdx.x * kScanItems;
  if (first + kScanItems <= n) {
    const int4* p = reinterpret_cast<const int4*>(in + first);   // (tiles start on 16 KB boundaries of a hipMalloc)
    for (int q = 0; q < kScanItems / 4; q++) {
      const int4 v = p[q];
      x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
    }
  } else {
    for (int q = 0; q < kScanItems; q++) x[q] = first + q < n ? in[first + q] : 0;
  }
}

__global__ __launch_bounds__(kScanThreads) void k_scan_tile_sums(const int* in, int n, int* sums)
{
  __shared__ int ws[4];
  int x[kScanItems];
  scan_load_tile(in, n, (long long)blockIdx.x * kScanTile, x);
  int t = 0;
  for (int q = 0; q < kScanItems; q++) t += x[q];
  const int total = scan_block_sum(t, ws);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// exclusive scan of the tile sums in place, one workgroup (carry over chunks of 1024)
__global__ __launch_bounds__(1024) void k_scan_top(int* sums, int nb)
{
  __shared__ int buf[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < nb ? sums[i] : 0;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan
      const int a = threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
      __syncthreads();
      buf[threadIdx.x] += a;
      __syncthreads();
    }
    if (i < nb) sums[i] = carry + buf[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += buf[1023];
    __syncthreads();
  }
}

__global__ __launch_bounds__(kScanThreads) void k_scan_apply(const int* in, int n, const int* sums, int* out)
{
  __shared__ int ws[4];
  int x[kScanItems];
  const long long base = (long long)blockIdx.x * kScanTile;
  scan_load_tile(in, n, base, x);
  int t = 0;
  for (int q = 0; q < kScanItems; q++) t += x[q];
  // exclusive prefix of the thread totals inside the tile: wave scan + wave offsets
  int incl = t;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int off = 1; off < 64; off <<= 1) {
    const int a = __shfl_up(incl, off, 64);
    if (lane >= off) incl += a;
  }
  if (lane == 63) ws[w] = incl;
  __syncthreads();
  int woff = 0;
  for (int k = 0; k < w; k++) woff += ws[k];
  int run = sums[blockIdx.x] + woff + incl - t;
  const long long first = base + (long long)threadIdx.x * kScanItems;
  if (first + kScanItems <= n) {
    int4* p = reinterpret_cast<int4*>(out + first);
    for (int q = 0; q < kScanItems / 4; q++) {
      int4 v;
      v.x = run; run += x[4 * q];
      v.y = run; run += x[4 * q + 1];
      v.z = run; run += x[4 * q + 2];
      v.w = run; run += x[4 * q + 3];
      p[q] = v;
    }
  } else {
    for (int q = 0; q < kScanItems; q++) {
      if (first + q < n) out[first + q] = run;
      run += x[q];
    }
  }
}

// a small table (the cell histogram of a bed of ~100 k grains): ONE workgroup, one launch -- the rebuild chain of a small bed
// is bound by the launches it takes (rocPRIM's scan is two).  Tiles of 16 k entries, four coalesced int4 loads per thread in
// flight at once (a tile is one round trip + one block scan: ~1 us).
constexpr int kOneBlockMax = 1 << 14;   // (one tile: ~4 us; four tiles measured 16.6 us against the 9.3 us of rocPRIM's two launches)
__global__ __launch_bounds__(1024) void k_scan_one_block(const int* in, int n, int* out)
{
  __shared__ int wsum[2][16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  constexpr int kQ = 4, kTile = 1024 * 4 * kQ;
  int carry = 0;
  for (int base = 0, it = 0; base < n; base += kTile, it++) {
    int4 v[kQ];
    int t = 0;
#pragma unroll
    for (int q = 0; q < kQ; q++) {
      // (thread t holds entries [16 t, 16 t + 16) of the tile: four int4, consecutive)
      const int k = base + 4 * (kQ * (int)threadIdx.x + q);
      v[q] = make_int4(0, 0, 0, 0);
      if (k + 3 < n) v[q] = *reinterpret_cast<const int4*>(in + k);   // (the tables are 16-byte aligned: exclusive_scan_i32)
      else {
        if (k < n) v[q].x = in[k];
        if (k + 1 < n) v[q].y = in[k + 1];
        if (k + 2 < n) v[q].z = in[k + 2];
      }
      t += v[q].x + v[q].y + v[q].z + v[q].w;
    }
    int incl = t;
    for (int off = 1; off < 64; off <<= 1) {
      const int a = __shfl_up(incl, off, 64);
      if (lane >= off) incl += a;
    }
    int* ws = wsum[it & 1];   // (two sets: a fast wave may be a tile ahead of a slow one's reads)
    if (lane == 63) ws[w] = incl;
    __syncthreads();
    int run = carry + incl - t, total = 0;
    for (int k = 0; k < 16; k++) {
      const int q = ws[k];
      if (k < w) run += q;
      total += q;
    }
    carry += total;
#pragma unroll
    for (int q = 0; q < kQ; q++) {
      const int k = base + 4 * (kQ * (int)threadIdx.x + q);
      int4 o;
      o.x = run;
      o.y = o.x + v[q].x;
      o.z = o.y + v[q].y;
      o.w = o.z + v[q].z;
      run = o.w + v[q].w;
      if (k + 3 < n) *reinterpret_cast<int4*>(out + k) = o;
      else {
        if (k < n) out[k] = o.x;
        if (k + 1 < n) out[k + 1] = o.y;
        if (k + 2 < n) out[k + 2] = o.z;
      }
    }
  }
}
}  // namespace

void exclusive_scan_i32(void*& tmp, size_t& tmp_bytes, const int* in, int* out, int n, hipStream_t s)
{
  if (n <= 0) return;
  // (read per call -- a rebuild-time function: the tests switch them inside one process)
  const bool own_scan = !(getenv("SF_ROCPRIM_SCAN") && atoi(getenv("SF_ROCPRIM_SCAN")));
  const int own_min = getenv("SF_SCAN_MIN") ? atoi(getenv("SF_SCAN_MIN")) : (1 << 16);   // (tests: 1)
  if (own_scan && n <= kOneBlockMax && in != out && !getenv("SF_SCAN_MIN") && (reinterpret_cast<uintptr_t>(in) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(out) & 15) == 0) {   // (SF_SCAN_MIN: the tests force the tiled scan)
    k_scan_one_block<<<1, 1024, 0, s>>>(in, n, out);
    SF_HIP(hipGetLastError());
    return;
  }
  if (own_scan && n >= own_min && in != out && (reinterpret_cast<uintptr_t>(in) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    const int nb = (n + kScanTile - 1) / kScanTile;
    const size_t need = sizeof(int) * (size_t)nb;
    if (need > tmp_bytes) {
      if (tmp) {
        SF_HIP(hipStreamSynchronize(s));
        SF_HIP(hipFree(tmp));
      }
      tmp_bytes = need + need / 4 + 4096;
      SF_HIP(hipMalloc(&tmp, tmp_bytes));
    }
    int* sums = static_cast<int*>(tmp);
    k_scan_tile_sums<<<nb, kScanThreads, 0, s>>>(in, n, sums);
    k_scan_top<<<1, 1024, 0, s>>>(sums, nb);
    k_scan_apply<<<nb, kScanThreads, 0, s>>>(in, n, sums, out);
    SF_HIP(hipGetLastError());
    return;
  }
  size_t need = 0;
  SF_HIP(rocprim::exclusive_scan(nullptr, need, in, out, 0, (size_t)n, rocprim::plus<int>(), s));
  if (need > tmp_bytes) {
    if (tmp) {
      SF_HIP(hipStreamSynchronize(s));
      SF_HIP(hipFree(tmp));
    }
    tmp_bytes = need + need / 4 + 4096;
    SF_HIP(hipMalloc(&tmp, tmp_bytes));
  }
  size_t avail = tmp_bytes;
  SF_HIP(rocprim::exclusive_scan(tmp, avail, in, out, 0, (size_t)n, rocprim::plus<int>(), s));
}

// indices i of [0, n) with keys[i] == 0, ascending (a stable compaction), and their number
namespace {
struct IsZero {
  __host__ __device__ bool operator()(unsigned k) const { return k == 0u; }
};
}  // namespace
void select_zero_keys(void*& tmp, size_t& tmp_bytes, const unsigned* keys, int* out, int* d_count, int n,
                      hipStream_t s)
{
  if (n <= 0) return;
  auto flags = rocprim::make_transform_iterator(keys, IsZero());
  rocprim::counting_iterator<int> idx(0);
  size_t need = 0;
  SF_HIP(rocprim::select(nullptr, need, idx, flags, out, d_count, (size_t)n, s));
  if (need > tmp_bytes) {
    if (tmp) {
      SF_HIP(hipStreamSynchronize(s));
      SF_HIP(hipFree(tmp));
    }
    tmp_bytes = need + need / 4 + 4096;
    SF_HIP(hipMalloc(&tmp, tmp_bytes));
  }
  size_t avail = tmp_bytes;
  SF_HIP(rocprim::select(tmp, avail, idx, flags, out, d_count, (size_t)n, s));
}

void sort_pairs_u32(void*& tmp, size_t& tmp_bytes, unsigned* keys_in, unsigned* keys_out, int* vals_in,
                    int* vals_out, int n, int end_bit, hipStream_t s)
{
  sort_pairs<unsigned>(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, end_bit, s);
}

void sort_pairs_u64(void*& tmp, size_t& tmp_bytes, unsigned long long* keys_in, unsigned long long* keys_out,
                    int* vals_in, int* vals_out, int n, int end_bit, hipStream_t s)
{
  sort_pairs<unsigned long long>(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, end_bit, s);
}

}  // namespace sf
