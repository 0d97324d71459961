// sf_handles.h -- what the opaque `void* ptr` of the sf_lammps_* / sf_dem_* C-ABI points to.
#pragma once
#include <cstdint>
#include <string>

#include "sf_dem.h"

namespace sf {
struct SfLammps {
  DemEngine eng;
  bool pair_hybrid = false;
  intptr_t comm = 0;
  // RCCL communicator + events of the C++ halo loop (sf_halo_rccl.hip); opaque here so that only that file
  // sees the RCCL headers
  void* halo = nullptr;
  void (*halo_delete)(void*) = nullptr;
  ~SfLammps()
  {
    if (halo && halo_delete) halo_delete(halo);
  }
};
}  // namespace sf
