// Development micro-benchmark (not part of the product): how fast can gfx950 stream FP64 rows from HBM through the
// vector L1 as a function of the per-lane load width, the number of concurrent row streams and a write mix.
// hipcc --offload-arch=gfx950 -O3 -o stream_bench stream_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// each lane handles element i of ROWS rows (row stride = n elements), W doubles per lane per row
template <int W, int ROWS, bool WRITE, bool NT>
__global__ __launch_bounds__(256) void k_rows(const double* __restrict__ in, double* __restrict__ out, size_t n)
{
  typedef double vec __attribute__((ext_vector_type(W)));
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i * W >= n) return;
  double acc = 0.0;
#pragma unroll 4
  for (int r = 0; r < ROWS; r++) {
    const vec* p = reinterpret_cast<const vec*>(in + (size_t)r * n) + i;
    vec v = NT ? __builtin_nontemporal_load(p) : *p;
    if (W == 1) acc += v[0]; else for (int k = 0; k < W; k++) acc += v[k];
    if (WRITE) {
      vec* q = reinterpret_cast<vec*>(out + (size_t)r * n) + i;
      vec w = v * 1.0000001;
      if (NT) __builtin_nontemporal_store(w, q); else *q = w;
    }
  }
  if (acc == 1.2345e300) out[i] = acc;
}

template <int W, int ROWS, bool WRITE, bool NT>
void run(const char* name, const double* in, double* out, size_t n)
{
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const size_t threads = n / W;
  dim3 grid((threads + 255) / 256);
  k_rows<W, ROWS, WRITE, NT><<<grid, 256>>>(in, out, n);
  CK(hipDeviceSynchronize());
  const int reps = 10;
  CK(hipEventRecord(a));
  for (int k = 0; k < reps; k++) k_rows<W, ROWS, WRITE, NT><<<grid, 256>>>(in, out, n);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double bytes = (double)ROWS * n * 8 * (WRITE ? 2 : 1) * reps;
  printf("%-44s %8.1f us/launch  %7.2f TB/s\n", name, ms * 1e3 / reps, bytes / (ms * 1e-3) / 1e12);
}

int main()
{
  const size_t n = 1 << 20;          // elements per row (like cap = 1M particles)
  const int rows = 48;
  double *in, *out;
  CK(hipMalloc(&in, n * rows * 8)); CK(hipMalloc(&out, n * rows * 8));
  CK(hipMemset(in, 0, n * rows * 8)); CK(hipMemset(out, 0, n * rows * 8));
  run<1, 48, false, false>("read  8B/lane 48 rows", in, out, n);
  run<2, 48, false, false>("read 16B/lane 48 rows", in, out, n);
  run<4, 48, false, false>("read 32B/lane 48 rows", in, out, n);
  run<1, 48, false, true>("read  8B/lane 48 rows nt", in, out, n);
  run<2, 48, false, true>("read 16B/lane 48 rows nt", in, out, n);
  run<1, 48, true, false>("read+write  8B/lane 48 rows", in, out, n);
  run<2, 48, true, false>("read+write 16B/lane 48 rows", in, out, n);
  run<1, 48, true, true>("read+write  8B/lane 48 rows nt", in, out, n);
  run<2, 48, true, true>("read+write 16B/lane 48 rows nt", in, out, n);
  // same bytes in fewer, wider rows (n elements per row scaled so that every case reads 403 MB)
  run<4, 12, false, true>("read 32B/lane 12 rows x4 wide nt", in, out, 4 * n);
  run<4, 12, true, true>("read+write 32B/lane 12 rows x4 wide nt", in, out, 4 * n);
  run<2, 24, false, true>("read 16B/lane 24 rows x2 wide nt", in, out, 2 * n);
  run<2, 24, true, true>("read+write 16B/lane 24 rows x2 wide nt", in, out, 2 * n);
  run<1, 12, false, false>("read  8B/lane 12 rows", in, out, n);
  run<2, 12, false, false>("read 16B/lane 12 rows", in, out, n);
  run<1, 12, true, false>("read+write  8B/lane 12 rows", in, out, n);
  return 0;
}
