// Development probe (not part of the product; round-5 review, ask 8): what does a barrier between the resident waves of ONE
// XCD cost when nothing crosses an XCD -- arrival by an L2 atomic, release = wait for the own stores (the vector L1 is
// write-through: a finished store is in the XCD's L2), acquire = `buffer_inv sc0` (vector L1 only) -- against the barrier
// of round 4 between all eight XCDs (agent-scope release / acquire: `buffer_wbl2 sc1` / `buffer_inv sc1`)?  Every wave
// writes a line per round and checks the line another wave wrote in the same round after the barrier (visibility is
// verified, not assumed).  One XCD: a CU-masked stream does not do it (a mask of 32 CUs -- bits 0-31 or bits 0, 8, 16, ... --
// is spread over all eight XCDs by the driver, first run of this probe); workgroups are dealt to the XCDs round-robin, so the
// launch holds eight times the workgroups and those that do not find themselves on XCD `xcc0` (XCC_ID) leave at once.
// hipcc --offload-arch=gfx950 -O2 -o xcd_barrier xcd_barrier.hip ; ./xcd_barrier [rounds]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)

__device__ __forceinline__ int xcc_id()
{
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

__global__ void k_probe(int* seen)
{
  if (threadIdx.x == 0) atomicAdd(&seen[xcc_id()], 1);
}

// MODE 0: XCD-local (no L2 write-back / invalidate; acquire = buffer_inv sc0)   MODE 1: agent scope (round 4's barrier)
// MODE 2: no barrier (the loop body alone)   MODE 3: XCD-local release (own stores done), acquire = buffer_inv sc1 (what
// MODE 0 turned out to need: sc0 leaves the vector L1's lines in place outside threadgroup-split mode).  One counter per round parity would do; a monotone counter is simpler: round r waits for nb * (r + 1).
template <int MODE>
__global__ __launch_bounds__(64) void k_rounds(int* counter, int* lines, int rounds, int* errors, long long* ticks, int one_xcd)
{
  // one_xcd: 8 x the workgroups, only those on XCD 0 take part (round-robin dispatch: they are workgroups 0, 8, 16, ...)
  if (one_xcd && xcc_id() != 0) return;
  const int b = one_xcd ? blockIdx.x >> 3 : blockIdx.x, nb = one_xcd ? gridDim.x >> 3 : gridDim.x, lane = threadIdx.x;
  if (one_xcd && lane == 0) atomicAdd(counter + 2, 1);   // (how many took part: must be nb)
  const long long t0 = wall_clock64();
  int bad = 0;
  for (int r = 0; r < rounds; r++) {
    // the round's work: one 128-byte line per wave (32 ints), written by all lanes < 32
    if (lane < 32) lines[(size_t)b * 32 + lane] = r * 1000 + lane;
    if (MODE != 2) {
      if (MODE == 0 || MODE == 3) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // own stores done (s_waitcnt vmcnt(0)): in the XCD's L2
      } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");       // + buffer_wbl2 sc1
      }
      if (lane == 0) {
        __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int want = nb * (r + 1);
        // (bounded: a launch whose waves are not all resident must end in an error, not hang the box)
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
          __builtin_amdgcn_s_sleep(1);
          if (wall_clock64() - t0 > 200000000ll || __hip_atomic_load(counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            __hip_atomic_store(counter + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // everybody leaves
            bad = 1 << 20;
            break;
          }
        }
        if (bad >= (1 << 20)) r = rounds;
      }
      __builtin_amdgcn_wave_barrier();
      if (MODE == 0) {
        asm volatile("buffer_inv sc0" ::: "memory");             // vector L1 only
      } else if (MODE == 3) {
        asm volatile("buffer_inv sc1" ::: "memory");
      } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");       // buffer_inv sc1
      }
    }
    // after the barrier: the line of the next wave, written in THIS round
    const int o = (b + 1) % nb;
    if (lane < 32 && MODE != 2) {
      const int v = lines[(size_t)o * 32 + lane];
      if (v != r * 1000 + lane) bad++;
    }
    if (MODE != 2) {
      // a second barrier keeps a fast wave from overwriting its line before its reader has looked (as a sub-step's output
      // buffer would be protected by the ping-pong): costed separately, not part of the figure -- use two line sets instead
    }
    lines += (size_t)nb * 32 * ((r & 1) ? -1 : 1);   // ping-pong between two line sets
  }
  if (bad) atomicAdd(errors, bad);
  if (lane == 0 && b == 0) *ticks = wall_clock64() - t0;
}

template <int MODE>
static double run(hipStream_t s, int nb, int rounds, int* d_counter, int* d_lines, int* d_err, long long* d_ticks, int* err_out,
                  int one_xcd = 0)
{
  CK(hipMemsetAsync(d_counter, 0, 12, s));
  CK(hipMemsetAsync(d_err, 0, 4, s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, s));
  k_rounds<MODE><<<one_xcd ? 8 * nb : nb, 64, 0, s>>>(d_counter, d_lines, rounds, d_err, d_ticks, one_xcd);
  CK(hipEventRecord(e1, s));
  CK(hipStreamSynchronize(s));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipMemcpy(err_out, d_err, 4, hipMemcpyDeviceToHost));
  if (one_xcd) {
    int took = 0;
    CK(hipMemcpy(&took, d_counter + 2, 4, hipMemcpyDeviceToHost));
    if (took != nb) {
      printf("(one XCD: %d workgroups took part instead of %d -- dispatch is not round-robin here)\n", took, nb);
      *err_out += 1 << 24;
    }
  }
  return 1e3 * ms / rounds;
}

int main(int argc, char** argv)
{
  const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
  int *d_seen, *d_counter, *d_lines, *d_err;
  long long* d_ticks;
  CK(hipMalloc(&d_seen, 64));
  CK(hipMalloc(&d_counter, 256));
  CK(hipMalloc(&d_lines, 2 * 4096 * 128));
  CK(hipMalloc(&d_err, 4));
  CK(hipMalloc(&d_ticks, 8));
  hipStream_t all;
  CK(hipStreamCreate(&all));
  int err = 0;
  printf("%-44s %8s %12s %12s %12s %12s\n", "launch", "waves", "no barrier", "inv sc0", "inv sc1", "agent scope");
  const int sizes_one[] = {32, 128, 157, 315, 384};
  const int sizes_all[] = {157, 630, 1024, 2048, 3072};
  for (int pass = 0; pass < 2; pass++) {
    hipStream_t s = all;
    for (int k = 0; k < 5; k++) {
      const int nb = pass == 0 ? sizes_one[k] : sizes_all[k];
      int e0 = 0, e1 = 0, e2 = 0, e3 = 0;
      double t3 = -1.0;
      const int ox = pass == 0 ? 1 : 0;
      const double t2 = run<2>(s, nb, rounds, d_counter, d_lines, d_err, d_ticks, &e2, ox);
      double t0 = -1.0;
      if (pass == 0) t0 = run<0>(s, nb, rounds, d_counter, d_lines, d_err, d_ticks, &e0, ox);   // (XCD-local is only valid on one XCD)
      if (pass == 0) t3 = run<3>(s, nb, rounds, d_counter, d_lines, d_err, d_ticks, &e3, ox);
      const double t1 = run<1>(s, nb, rounds, d_counter, d_lines, d_err, d_ticks, &e1, ox);
      printf("%-44s %8d %9.2f us %9.2f us %9.2f us %9.2f us   stale reads: %d / %d / %d\n",
             pass == 0 ? "one XCD (workgroups 0, 8, 16, ... of 8 x)" : "all eight XCDs", nb, t2, t0, t3, t1, e0, e3, e1);
      err += e3 + e1;
    }
  }
  // what the XCD-local barrier would see across XCDs (expected: stale reads -- the reason it is only valid on one XCD)
  {
    int e0 = 0;
    const double t0 = run<0>(all, 630, rounds, d_counter, d_lines, d_err, d_ticks, &e0);
    printf("%-44s %8d %12s %9.2f us %12s   stale reads: %d (expected > 0: L2s of different XCDs)\n",
           "all eight XCDs, XCD-local barrier (INVALID)", 630, "", t0, "", e0);
  }
  return err ? 1 : 0;
}
