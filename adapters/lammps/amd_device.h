// amd_device.h -- device buffers for the LAMMPS style adapters, on the C ABI only (sf_dev_*): the adapters are
// compiled by LAMMPS' host compiler, without HIP headers.
#ifndef SEDIFOAM_AMD_DEVICE_H
#define SEDIFOAM_AMD_DEVICE_H

#include <cstddef>

#include "sedifoam_amd.h"

namespace sedifoam_amd {

// grows, never shrinks; contents are NOT preserved on growth (every user refills it)
class DevBuf {
 public:
  DevBuf() : p_(NULL), bytes_(0) {}
  ~DevBuf() { if (p_) sf_dev_free(p_); }
  void* reserve(std::size_t bytes)
  {
    if (bytes > bytes_) {
      if (p_) sf_dev_free(p_);
      bytes_ = bytes + bytes / 4 + 256;
      p_ = sf_dev_alloc(bytes_);
    }
    return p_;
  }
  template <class T>
  T* upload(const T* host, std::size_t n)
  {
    reserve(n * sizeof(T));
    if (p_ && n) sf_dev_upload(p_, host, n * sizeof(T), NULL);
    return static_cast<T*>(p_);
  }
  template <class T>
  T* zeros(std::size_t n)
  {
    reserve(n * sizeof(T));
    if (p_ && n) sf_dev_zero(p_, n * sizeof(T), NULL);
    return static_cast<T*>(p_);
  }
  template <class T>
  T* as() const { return static_cast<T*>(p_); }

 private:
  DevBuf(const DevBuf&);
  DevBuf& operator=(const DevBuf&);
  void* p_;
  std::size_t bytes_;
};

}  // namespace sedifoam_amd

#endif
