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
  // The world the LAMMPS object was opened on (sf_lammps_open_world <- `new LAMMPS(0, NULL, commLammps)` on a
  // duplicated world communicator, lammpsFoam/softParticleCloud.C:60-62).  With more than one rank the engine
  // decomposes ITSELF the way LAMMPS does: `processors px py pz` -> a grid of bricks when the box is created
  // (read_data), every later lammps_* call collective (interfaceToLammps/library.cpp:94-131,372-386).
  int world_rank = 0, world_size = 1;
  char comm_id[128] = {0};       // RCCL unique id of rank 0, broadcast by the caller's MPI
  int procgrid[3] = {0, 0, 0};   // `processors px py pz`, 0 = `*` ([3P] LAMMPS chooses by surface area)
  bool decomposed = false;       // the bricks were set up by the script path (sf_brick_init behind read_data)
  bool pending_rebuild = false;  // lammps_create_particle / lammps_delete_particle: next_reneighbor (library.cpp:482-486)
  long long natoms = -1;         // atom->natoms (library.cpp:94-98): set by read_data, create / delete particle
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
