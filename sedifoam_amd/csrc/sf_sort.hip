// sf_sort.hip -- key/value radix sorts used only on neighbour rebuilds (rocPRIM device primitives;
// kept in their own translation unit because the header is heavy to compile).
#include <cstring>

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

void exclusive_scan_i32(void*& tmp, size_t& tmp_bytes, const int* in, int* out, int n, hipStream_t s)
{
  if (n <= 0) return;
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
