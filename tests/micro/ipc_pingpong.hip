// Development probe (not part of the product): can one process write into another process's device memory through an
// IPC mapping, publish a flag with a system-scope release store, and be seen by a kernel of the owner polling with
// acquire loads -- both processes on ONE GPU (what the one-GPU box can test) or on two (GPUs a b given)?
// Measures the one-way latency of {payload + flag} and checks the payload for every round.
// hipcc --offload-arch=gfx950 -O2 -o ipc_pingpong ipc_pingpong.hip ; ./ipc_pingpong [fine|coarse] [rounds] [devA devB]
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("[%d] HIP error %s at line %d\n", getpid(), hipGetErrorString(e_), __LINE__); exit(2);} } while (0)

constexpr int kPayload = 4096;   // doubles per message (32 KB, a face of ~450 ghost records)

// writer: payload[i] = round * 1000 + i, then flag = round (release, system scope)
__global__ void k_write(double* remote, int* remote_flag, int round)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < kPayload) remote[i] = round * 1000.0 + i;
}
__global__ void k_publish(int* remote_flag, int round)
{
  __hip_atomic_store(remote_flag, round, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// owner: wait for flag >= round (bounded), then check the payload; result[0] = mismatches, result[1] = polls
__global__ void k_wait_check(const double* mine, const int* flag, int round, int* result, long long max_ticks)
{
  __shared__ int ok;
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();
    int polls = 0;
    ok = 1;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < round) {
      polls++;
      __builtin_amdgcn_s_sleep(8);
      if (wall_clock64() - t0 > max_ticks) { ok = 0; break; }
    }
    if (blockIdx.x == 0) result[1] = polls;
    if (!ok) atomicAdd(&result[2], 1);
  }
  __syncthreads();
  if (!ok) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (mine && i < kPayload && mine[i] != round * 1000.0 + i) atomicAdd(&result[0], 1);
}

int main(int argc, char** argv)
{
  const bool fine = argc < 2 || strcmp(argv[1], "coarse") != 0;
  const int rounds = argc > 2 ? atoi(argv[2]) : 200;
  const int devA = argc > 4 ? atoi(argv[3]) : 0, devB = argc > 4 ? atoi(argv[4]) : 0;
  int p2c[2], c2p[2];
  if (pipe(p2c) || pipe(c2p)) return 1;
  const pid_t pid = fork();
  if (pid == 0) {
    // ---- child B: maps A's buffer, writes rounds into it; owns a buffer A writes its acknowledgements into
    CK(hipSetDevice(devB));
    hipIpcMemHandle_t hA;
    if (read(p2c[0], &hA, sizeof hA) != (ssize_t)sizeof hA) return 3;
    void* a_mem = nullptr;
    CK(hipIpcOpenMemHandle(&a_mem, hA, hipIpcMemLazyEnablePeerAccess));
    double* a_payload = static_cast<double*>(a_mem);
    int* a_flag = reinterpret_cast<int*>(a_payload + kPayload);
    // B's own area for the acknowledgement flag
    void* b_mem = nullptr;
    if (fine) CK(hipExtMallocWithFlags(&b_mem, 4096, hipDeviceMallocFinegrained)); else CK(hipMalloc(&b_mem, 4096));
    CK(hipMemset(b_mem, 0, 4096));
    hipIpcMemHandle_t hB;
    CK(hipIpcGetMemHandle(&hB, b_mem));
    if (write(c2p[1], &hB, sizeof hB) != (ssize_t)sizeof hB) return 3;
    int* result;
    CK(hipMalloc(&result, 16)); CK(hipMemset(result, 0, 16));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int r = 1; r <= rounds; r++) {
      k_write<<<(kPayload + 255) / 256, 256, 0, st>>>(a_payload, a_flag, r);
      k_publish<<<1, 1, 0, st>>>(a_flag, r);
      // wait for A's acknowledgement of round r before overwriting the payload
      k_wait_check<<<1, 64, 0, st>>>(nullptr, static_cast<int*>(b_mem), r, result, 500000000LL);
      CK(hipStreamSynchronize(st));
    }
    int h[4]; CK(hipMemcpy(h, result, 16, hipMemcpyDeviceToHost));
    printf("[B] done, timeouts %d\n", h[2]);
    CK(hipIpcCloseMemHandle(a_mem));
    return h[2] ? 4 : 0;
  }
  // ---- parent A: owns the payload + flag area
  CK(hipSetDevice(devA));
  void* a_mem = nullptr;
  const size_t bytes = sizeof(double) * kPayload + 4096;
  if (fine) CK(hipExtMallocWithFlags(&a_mem, bytes, hipDeviceMallocFinegrained)); else CK(hipMalloc(&a_mem, bytes));
  CK(hipMemset(a_mem, 0, bytes));
  hipIpcMemHandle_t hA;
  CK(hipIpcGetMemHandle(&hA, a_mem));
  if (write(p2c[1], &hA, sizeof hA) != (ssize_t)sizeof hA) return 3;
  hipIpcMemHandle_t hB;
  if (read(c2p[0], &hB, sizeof hB) != (ssize_t)sizeof hB) return 3;
  void* b_mem = nullptr;
  CK(hipIpcOpenMemHandle(&b_mem, hB, hipIpcMemLazyEnablePeerAccess));
  double* payload = static_cast<double*>(a_mem);
  int* flag = reinterpret_cast<int*>(payload + kPayload);
  int* result;
  CK(hipMalloc(&result, 16)); CK(hipMemset(result, 0, 16));
  hipStream_t st; CK(hipStreamCreate(&st));
  const auto t0 = std::chrono::steady_clock::now();
  for (int r = 1; r <= rounds; r++) {
    k_wait_check<<<(kPayload + 255) / 256, 256, 0, st>>>(payload, flag, r, result, 500000000LL);
    k_publish<<<1, 1, 0, st>>>(static_cast<int*>(b_mem), r);
    CK(hipStreamSynchronize(st));
  }
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
  int h[4]; CK(hipMemcpy(h, result, 16, hipMemcpyDeviceToHost));
  int status = 0; waitpid(pid, &status, 0);
  printf("[A] %s-grained, %d rounds of {%d-byte payload + flag} there and {flag} back: %.1f us per round trip (host-synchronised "
         "each round); payload mismatches %d, timeouts %d, child rc %d\n", fine ? "fine" : "coarse", rounds, kPayload * 8, us, h[0],
         h[2], WEXITSTATUS(status));
  return (h[0] || h[2] || WEXITSTATUS(status)) ? 1 : 0;
}
