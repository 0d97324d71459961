// Development micro-benchmark (not part of the product): what does ONE vector-memory load instruction cost in the
// texture addresser as a function of its address pattern, when every line it touches is already in the vector L1?
// Each wave re-reads the same small window (2-4 KB per array) ITER times: lane-linear 16-byte loads (coalesced),
// 16-byte loads at a 32-byte stride (the product's record gather from consecutive neighbours), the same with the lanes
// permuted inside the window, 8- and 4-byte loads at a 32-byte stride.  Prints cycles per load instruction per CU
// (4 SIMDs x WAVES waves issuing).
// hipcc --offload-arch=gfx950 -O3 -o gather_bench gather_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

template <int BYTES, int ALT = 16>
__global__ __launch_bounds__(256) void k_gather(const char* base, const int* idx, int stride, int iters, double* out)
{
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const char* p = base + (size_t)wave * 8192 + (size_t)idx[lane] * stride;   // a private 8 KB window per wave
  double acc = 0.0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const char* q = p + ((it + u) & 1) * ALT;   // (alternate the two halves of a 32-byte record: defeats hoisting)
      if (BYTES == 16) { d2 v = *reinterpret_cast<const volatile d2*>(q); acc += v.x + v.y; }
      else if (BYTES == 8) { double v = *reinterpret_cast<const volatile double*>(q); acc += v; }
      else { float v = *reinterpret_cast<const volatile float*>(q); acc += v; }
    }
  }
  if (acc == 1.2345e300) out[0] = acc;
}

template <int BYTES, int ALT = 16>
void run(const char* name, const char* base, const int* d_idx, int stride, int blocks)
{
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  double* out; CK(hipMalloc(&out, 8));
  const int iters = 2000;
  k_gather<BYTES, ALT><<<blocks, 256>>>(base, d_idx, stride, 10, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  k_gather<BYTES, ALT><<<blocks, 256>>>(base, d_idx, stride, iters, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  int clk_khz = 0; CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
  const double instr_per_cu = (double)blocks / 256.0 * 4.0 * iters * 8.0;   // wave-level load instructions per CU
  const double cycles = ms * 1e-3 * clk_khz * 1e3;
  printf("%-58s %8.3f ms  %6.1f cycles per load instruction per CU  (%.0f B/clk/CU)\n", name, ms, cycles / instr_per_cu,
         64.0 * BYTES * instr_per_cu / cycles);
  CK(hipFree(out));
}

int main()
{
  const int blocks = 256 * 3;   // 3 workgroups of 4 waves per CU: 12 waves per CU like the product kernel
  char* base; CK(hipMalloc(&base, (size_t)blocks * 4 * 8192 + 65536)); CK(hipMemset(base, 0, (size_t)blocks * 4 * 8192 + 65536));
  std::vector<int> lin(64), perm(64), pairs(64);
  for (int l = 0; l < 64; l++) { lin[l] = l; perm[l] = (l * 37 + 11) & 63; pairs[l] = l ^ 1; }
  int *d_lin, *d_perm, *d_pairs;
  CK(hipMalloc(&d_lin, 256)); CK(hipMalloc(&d_perm, 256)); CK(hipMalloc(&d_pairs, 256));
  CK(hipMemcpy(d_lin, lin.data(), 256, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_perm, perm.data(), 256, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_pairs, pairs.data(), 256, hipMemcpyHostToDevice));
  run<16>("16 B/lane, lane-linear (stride 16: 1 KB contiguous)", base, d_lin, 16, blocks);
  run<16>("16 B/lane, stride 32 (records of consecutive neighbours)", base, d_lin, 32, blocks);
  run<16>("16 B/lane, stride 32, lanes permuted in the window", base, d_perm, 32, blocks);
  run<16>("16 B/lane, stride 16, lanes permuted in the window", base, d_perm, 16, blocks);
  run<16>("16 B/lane, stride 16, adjacent lanes swapped", base, d_pairs, 16, blocks);
  run<16>("16 B/lane, stride 96 (AoS 96-byte records)", base, d_lin, 96, blocks);
  // scattered gathers: every lane in a 128-byte line of its own (64 lines per instruction) against lane pairs reading the 32
  // bytes of ONE record together (32 lines per instruction, the two instructions cover 64 records)
  run<16>("16 B/lane, stride 128: every lane its own line (64 lines)", base, d_lin, 128, blocks);
  {
    std::vector<int> off(64);
    int* d_off; CK(hipMalloc(&d_off, 256));
    for (int l = 0; l < 64; l++) off[l] = (l >> 1) * 128 + (l & 1) * 16;
    CK(hipMemcpy(d_off, off.data(), 256, hipMemcpyHostToDevice));
    run<16, 32>("16 B/lane, lane PAIRS share a 32-B record, 128 B apart (32 lines)", base, d_off, 1, blocks);
    for (int l = 0; l < 64; l++) off[l] = (l >> 1) * 64 + (l & 1) * 16;
    CK(hipMemcpy(d_off, off.data(), 256, hipMemcpyHostToDevice));
    run<16, 32>("16 B/lane, lane pairs share a record, 64 B apart (16 lines)", base, d_off, 1, blocks);
    for (int l = 0; l < 64; l++) off[l] = (l >> 1) * 32 + (l & 1) * 16;
    CK(hipMemcpy(d_off, off.data(), 256, hipMemcpyHostToDevice));
    run<16, 1024>("16 B/lane, lane pairs share a record, consecutive records (8 lines)", base, d_off, 1, blocks);
    // the half-wave split: lanes 0..31 read the first 16 bytes of 32 consecutive records, lanes 32..63 the second 16 bytes
    // of the SAME records (1 KB contiguous, 8 lines; v_permlane32_swap afterwards puts the halves together)
    for (int l = 0; l < 64; l++) off[l] = (l & 31) * 32 + (l >> 5) * 16;
    CK(hipMemcpy(d_off, off.data(), 256, hipMemcpyHostToDevice));
    run<16, 1024>("16 B/lane, HALF-WAVES share consecutive records (8 lines)", base, d_off, 1, blocks);
    // the same by quarters: lanes 16 q .. 16 q + 15 read bytes 16 (q & 1) of records 16 (q >> 1) + (l & 15) ...
    for (int l = 0; l < 64; l++) off[l] = ((l >> 5) * 16 + (l & 15)) * 32 + ((l >> 4) & 1) * 16;
    CK(hipMemcpy(d_off, off.data(), 256, hipMemcpyHostToDevice));
    run<16, 1024>("16 B/lane, 16-lane groups share consecutive records (8 lines)", base, d_off, 1, blocks);
    // half-wave split over records two apart (the neighbours of every second atom): 16 lines
    for (int l = 0; l < 64; l++) off[l] = (l & 31) * 64 + (l >> 5) * 16;
    CK(hipMemcpy(d_off, off.data(), 256, hipMemcpyHostToDevice));
    run<16, 32>("16 B/lane, half-waves share records 64 B apart (16 lines)", base, d_off, 1, blocks);
    for (int l = 0; l < 64; l++) off[l] = (l >> 2) * 128 + (l & 3) * 32;
    CK(hipMemcpy(d_off, off.data(), 256, hipMemcpyHostToDevice));
    run<16>("16 B/lane, four lanes per line, own records (16 lines)", base, d_off, 1, blocks);
    for (int l = 0; l < 64; l++) off[l] = (l >> 1) * 128 + (l & 1) * 32;
    CK(hipMemcpy(d_off, off.data(), 256, hipMemcpyHostToDevice));
    run<16>("16 B/lane, two lanes per line, own records (32 lines)", base, d_off, 1, blocks);
  }
  run<8>(" 8 B/lane, lane-linear (stride 8: 512 B contiguous)", base, d_lin, 8, blocks);
  run<8>(" 8 B/lane, stride 32", base, d_lin, 32, blocks);
  run<8>(" 8 B/lane, stride 8, lanes permuted", base, d_perm, 8, blocks);
  run<4>(" 4 B/lane, lane-linear (stride 4)", base, d_lin, 4, blocks);
  run<4>(" 4 B/lane, stride 32", base, d_lin, 32, blocks);
  return 0;
}
