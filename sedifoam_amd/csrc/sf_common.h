// sf_common.h -- error plumbing and small device-memory helpers shared by all translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>

namespace sf {

// last error text, per host thread (returned by sf_last_error())
std::string& last_error();
void set_error(const char* fmt, ...);

struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};

[[noreturn]] void fail(const char* fmt, ...);

#define SF_HIP(call)                                                                            \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      ::sf::fail("HIP error %s at %s:%d: %s", hipGetErrorName(e_), __FILE__, __LINE__,          \
                 hipGetErrorString(e_));                                                        \
  } while (0)

// C-ABI wrapper: run body, translate exceptions into -1 + error text.
#define SF_API_BEGIN try {
#define SF_API_END(okval)                                  \
  return okval;                                            \
  }                                                        \
  catch (const std::exception& ex) {                       \
    ::sf::set_error("%s", ex.what());                      \
    return -1;                                             \
  }

// A [rows][cap] device array of fixed-size elements, component(row)-major, that can be re-strided
// when the particle capacity grows.
struct DevArray {
  void* ptr = nullptr;
  size_t elem = 0;
  int rows = 0;
  size_t cap = 0;

  void alloc(size_t elem_, int rows_, size_t cap_, hipStream_t s)
  {
    release();
    elem = elem_;
    rows = rows_;
    cap = cap_;
    SF_HIP(hipMalloc(&ptr, elem * (size_t)rows * cap));
    SF_HIP(hipMemsetAsync(ptr, 0, elem * (size_t)rows * cap, s));
  }
  void grow(size_t newcap, hipStream_t s)
  {
    if (newcap <= cap) return;
    void* np = nullptr;
    SF_HIP(hipMalloc(&np, elem * (size_t)rows * newcap));
    SF_HIP(hipMemsetAsync(np, 0, elem * (size_t)rows * newcap, s));
    if (ptr)
      SF_HIP(hipMemcpy2DAsync(np, newcap * elem, ptr, cap * elem, cap * elem, rows,
                              hipMemcpyDeviceToDevice, s));
    SF_HIP(hipStreamSynchronize(s));
    if (ptr) SF_HIP(hipFree(ptr));
    ptr = np;
    cap = newcap;
  }
  void release()
  {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
  }
  template <class T>
  T* as() const
  {
    return reinterpret_cast<T*>(ptr);
  }
};

inline int div_up(long long a, int b) { return (int)((a + b - 1) / b); }

}  // namespace sf
