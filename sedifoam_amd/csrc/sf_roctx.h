// sf_roctx.h -- roctx ranges named after the reference's timer buckets (lammpsFoam/writeCPUTime.H:1-19:
// "OpenFOAM/evolve/calcTcField/diffusion/particle move" and "assemble/transpose/flatten/foam->lammps/lammps/
// lammps->foam"), so that a rocprofv3 --marker-trace / roctx-enabled run of the product lines up with the split the
// reference prints.  The roctx library is dlopen'ed on first use (rocprofiler-sdk-roctx, else roctx64); without it
// the ranges are no-ops.  Assemble / transpose / flatten have no counterpart here (the cloud reads the engine's
// records in place).
#pragma once
#include <dlfcn.h>

namespace sf {

struct RoctxApi {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  RoctxApi()
  {
    for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4",
                             "libroctx64.so"}) {
      void* lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (!lib) continue;
      push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
      pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
      if (push && pop) return;
      push = nullptr;
      pop = nullptr;
    }
  }
};

inline RoctxApi& roctx()
{
  static RoctxApi api;
  return api;
}

// RAII range: Range r("lammps");
struct Range {
  explicit Range(const char* name)
  {
    if (roctx().push) roctx().push(name);
  }
  ~Range()
  {
    if (roctx().pop) roctx().pop();
  }
  Range(const Range&) = delete;
  Range& operator=(const Range&) = delete;
};

}  // namespace sf
