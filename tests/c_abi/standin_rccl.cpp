// standin_rccl.cpp -- TEST INFRASTRUCTURE, not part of the product.
//
// A stand-in for librccl that lets SEVERAL ranks share ONE GPU: RCCL itself refuses two ranks on one device
// ("Duplicate GPU detected"), so on a one-GPU box the C++ slab driver (csrc/sf_halo_rccl.hip: migration, border
// exchange, fused forward halo + rebuild vote, the setup all-reduces) could only ever talk to itself.  This library
// exports the nine entry points that driver loads (SF_RCCL_LIB points the loader here) with NCCL's semantics --
// grouped point-to-point operations, the k-th send to a peer matches that peer's k-th receive from the sender,
// operations ordered after the work already queued on the stream -- and moves the bytes through a file in /tmp mapped
// by every rank: device -> host copy on the sending process, host -> device copy on the receiving one.  Slow, synchronous,
// and exactly what is needed to run world_size 2 to 8 of the driver on a one-GPU box.
//
// build: hipcc -shared -fPIC -O2 tests/c_abi/standin_rccl.cpp -o libstandin_rccl.so   (tests/test_halo_gpu.py does it)
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr int kMaxRanks = 8;
constexpr int kSlots = 2;                    // messages in flight per directed pair (a group of the driver has <= 2)
constexpr size_t kSlotBytes = 8u << 20;      // (untouched pages of the segment cost nothing)
constexpr double kTimeoutS = 120.0;

struct Slot {
  std::atomic<uint32_t> full;
  uint64_t bytes;
  alignas(64) unsigned char data[kSlotBytes];
};
struct Ring {                                // sender -> receiver
  uint64_t head;                             // written by the sender only
  uint64_t tail;                             // written by the receiver only
  Slot slot[kSlots];
};
struct Shared {
  std::atomic<uint32_t> arrive;
  std::atomic<uint32_t> generation;
  alignas(64) unsigned char red[kMaxRanks][256];
  Ring ring[kMaxRanks][kMaxRanks];           // [from][to]
};

struct Op {
  bool send;
  void* ptr;
  size_t bytes;
  int peer;
  hipStream_t stream;
  bool done;
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

double now()
{
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

size_t type_bytes(ncclDataType_t t)
{
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
  }
}

}  // namespace

struct ncclComm {
  Shared* sh = nullptr;
  int rank = 0, world = 1;
  char name[64];
  uint32_t gen = 0;                          // barrier generation this rank waits for next
};

namespace {

bool barrier(ncclComm* c)
{
  Shared* s = c->sh;
  const uint32_t g = c->gen++;
  if (s->arrive.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->world) {
    s->arrive.store(0, std::memory_order_relaxed);
    s->generation.store(g + 1, std::memory_order_release);
    return true;
  }
  const double t0 = now();
  while (s->generation.load(std::memory_order_acquire) != g + 1) {
    if (now() - t0 > kTimeoutS) return false;
    usleep(20);
  }
  return true;
}

// one pass over the queued operations; per peer and direction only the oldest pending one may go
ncclResult_t run_ops(ncclComm* c, std::vector<Op>& ops)
{
  for (Op& o : ops)
    if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
  size_t left = ops.size();
  const double t0 = now();
  while (left) {
    bool blocked_s[kMaxRanks] = {}, blocked_r[kMaxRanks] = {};
    bool progress = false;
    for (Op& o : ops) {
      if (o.done) continue;
      if (o.send) {
        if (blocked_s[o.peer]) continue;
        Ring& r = c->sh->ring[c->rank][o.peer];
        Slot& sl = r.slot[r.head % kSlots];
        if (sl.full.load(std::memory_order_acquire)) {
          blocked_s[o.peer] = true;
          continue;
        }
        if (o.bytes > kSlotBytes) {
          fprintf(stderr, "standin_rccl: a %zu-byte message does not fit a %zu-byte slot\n", o.bytes, kSlotBytes);
          return ncclInvalidArgument;
        }
        if (o.bytes && hipMemcpy(sl.data, o.ptr, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        sl.bytes = o.bytes;
        sl.full.store(1, std::memory_order_release);
        r.head++;
      } else {
        if (blocked_r[o.peer]) continue;
        Ring& r = c->sh->ring[o.peer][c->rank];
        Slot& sl = r.slot[r.tail % kSlots];
        if (!sl.full.load(std::memory_order_acquire)) {
          blocked_r[o.peer] = true;
          continue;
        }
        if (sl.bytes != o.bytes) {
          fprintf(stderr, "standin_rccl: rank %d expects %zu bytes from rank %d, %llu were sent\n", c->rank, o.bytes,
                  o.peer, (unsigned long long)sl.bytes);
          return ncclInvalidArgument;
        }
        if (o.bytes && hipMemcpy(o.ptr, sl.data, o.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
        sl.full.store(0, std::memory_order_release);
        r.tail++;
      }
      o.done = true;
      left--;
      progress = true;
    }
    if (!progress) {
      if (now() - t0 > kTimeoutS) {
        fprintf(stderr, "standin_rccl: rank %d timed out with %zu operations pending\n", c->rank, left);
        return ncclInternalError;
      }
      usleep(10);
    }
  }
  return ncclSuccess;
}

ncclResult_t submit(ncclComm* c, Op o)
{
  if (o.peer < 0 || o.peer >= c->world) return ncclInvalidArgument;
  g_ops.push_back(o);
  if (g_depth > 0) return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(g_ops);
  return run_ops(c, ops);
}

ncclComm* g_group_comm = nullptr;   // (one communicator per group is all the driver uses)

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "/tmp/sf_standin_rccl_%d_%llx", (int)getpid(),
           (unsigned long long)(now() * 1e6));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
  if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  ncclComm* c = new ncclComm;
  c->rank = rank;
  c->world = nranks;
  strncpy(c->name, id.internal, sizeof(c->name) - 1);
  c->name[sizeof(c->name) - 1] = 0;
  const int fd = open(c->name, O_CREAT | O_RDWR, 0600);   // (sparse: only the pages messages touch exist)
  if (fd < 0 || ftruncate(fd, sizeof(Shared)) != 0) {
    perror("standin_rccl: mapping file");
    return ncclSystemError;
  }
  void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);   // a new segment reads as zeros
  close(fd);
  if (p == MAP_FAILED) return ncclSystemError;
  c->sh = static_cast<Shared*>(p);
  if (!barrier(c)) return ncclInternalError;
  *comm = c;
  return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t c, int* count)
{
  *count = c->world;
  return ncclSuccess;
}

ncclResult_t ncclGetVersion(int* version)
{
  *version = 0;   // (0: not a real RCCL)
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c)
{
  if (!c) return ncclSuccess;
  unlink(c->name);
  munmap(c->sh, sizeof(Shared));
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclGroupStart()
{
  g_depth++;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
  if (--g_depth > 0) return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(g_ops);
  if (ops.empty()) return ncclSuccess;
  return run_ops(g_group_comm, ops);
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s)
{
  g_group_comm = c;
  return submit(c, Op{true, const_cast<void*>(buf), count * type_bytes(t), peer, s, false});
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s)
{
  g_group_comm = c;
  return submit(c, Op{false, buf, count * type_bytes(t), peer, s, false});
}

// doubles and 64-bit integers, sum / max / min, combined in rank order on every rank
ncclResult_t ncclAllReduce(const void* sendbuf, void* recvbuf, size_t count, ncclDataType_t t, ncclRedOp_t op,
                           ncclComm_t c, hipStream_t s)
{
  const size_t bytes = count * type_bytes(t);
  if (bytes > sizeof(c->sh->red[0]) || (t != ncclDouble && t != ncclInt64)) return ncclInvalidArgument;
  if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpy(c->sh->red[c->rank], sendbuf, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  if (!barrier(c)) return ncclInternalError;
  unsigned char out[256];
  memcpy(out, c->sh->red[0], bytes);
  for (int r = 1; r < c->world; r++)
    for (size_t k = 0; k < count; k++) {
      if (t == ncclDouble) {
        double a, b;
        memcpy(&a, out + 8 * k, 8);
        memcpy(&b, c->sh->red[r] + 8 * k, 8);
        a = op == ncclSum ? a + b : (op == ncclMax ? (a > b ? a : b) : (a < b ? a : b));
        memcpy(out + 8 * k, &a, 8);
      } else {
        long long a, b;
        memcpy(&a, out + 8 * k, 8);
        memcpy(&b, c->sh->red[r] + 8 * k, 8);
        a = op == ncclSum ? a + b : (op == ncclMax ? (a > b ? a : b) : (a < b ? a : b));
        memcpy(out + 8 * k, &a, 8);
      }
    }
  if (!barrier(c)) return ncclInternalError;   // (everybody has read before the next all-reduce overwrites)
  if (hipMemcpy(recvbuf, out, bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r)
{
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "unhandled HIP error (stand-in)";
    case ncclSystemError: return "system error (stand-in)";
    case ncclInternalError: return "internal error or time-out (stand-in)";
    case ncclInvalidArgument: return "invalid argument (stand-in)";
    default: return "error (stand-in)";
  }
}

}  // extern "C"
