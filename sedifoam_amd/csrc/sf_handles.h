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
};
}  // namespace sf
