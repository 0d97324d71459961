// Shared by the MPI / C++ host programs of tests/c_abi (mpi_slab_host.cpp, mpi_cloud_host.cpp): a small FCC bed every
// rank builds identically, one engine per rank fed through the C-ABI, and the owned atoms back on the host.
#pragma once
#include <mpi.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sedifoam_amd.h"

namespace {

struct Bed {
  std::vector<double> x, v, d, rho;
  std::vector<int> tag;
  double lo[3], hi[3];
};

// FCC bed, periodic in x and z, resting on a wall at y = 0; jitter and velocities from a fixed linear congruential
// sequence so that every rank builds the same bed
Bed make_bed(int ncx, int ncy, int ncz, double vmax)
{
  Bed b;
  const double d = 1.0e-3, a = 0.98 * d, edge = a * std::sqrt(2.0);
  const double basis[4][3] = {{0, 0, 0}, {0.5, 0.5, 0}, {0.5, 0, 0.5}, {0, 0.5, 0.5}};
  uint64_t s = 88172645463325252ull;
  auto uni = [&]() {   // (-1, 1)
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0;
  };
  int t = 1;
  for (int i = 0; i < ncx; i++)
    for (int j = 0; j < ncy; j++)
      for (int k = 0; k < ncz; k++)
        for (int q = 0; q < 4; q++) {
          double p[3] = {(i + basis[q][0]) * edge + 0.25 * edge, (j + basis[q][1]) * edge + 0.5 * a,
                         (k + basis[q][2]) * edge + 0.25 * edge};
          for (int c = 0; c < 3; c++) {
            b.x.push_back(p[c] + 0.005 * d * uni());
            b.v.push_back(vmax * uni());
          }
          b.d.push_back(d);
          b.rho.push_back(2650.0);
          b.tag.push_back(t++);
        }
  b.lo[0] = b.lo[1] = b.lo[2] = 0.0;
  b.hi[0] = ncx * edge;
  b.hi[1] = ncy * edge * 1.25 + 4 * d;
  b.hi[2] = ncz * edge;
  return b;
}

#define CHECK(call)                                                                  \
  do {                                                                               \
    if ((call) != 0) {                                                               \
      std::printf("FAIL %s: %s\n", #call, sf_last_error());                          \
      MPI_Abort(MPI_COMM_WORLD, 1);                                                  \
    }                                                                                \
  } while (0)

void* make_engine(const Bed& b, const std::vector<int>& pick, MPI_Comm comm)
{
  void* ptr = nullptr;
  intptr_t h = 0;
  static_assert(sizeof(MPI_Comm) <= sizeof(intptr_t), "MPI_Comm fits the opaque handle");
  std::memcpy(&h, &comm, sizeof(MPI_Comm));
  CHECK(sf_lammps_open(0, nullptr, h, &ptr));
  CHECK(sf_dem_set_box(ptr, b.lo, b.hi));
  std::vector<double> x, v, d, rho;
  std::vector<int> tag;
  for (int i : pick) {
    for (int c = 0; c < 3; c++) {
      x.push_back(b.x[3 * i + c]);
      v.push_back(b.v[3 * i + c]);
    }
    d.push_back(b.d[i]);
    rho.push_back(b.rho[i]);
    tag.push_back(b.tag[i]);
  }
  CHECK(sf_dem_create_atoms(ptr, (int)pick.size(), x.data(), v.data(), nullptr, d.data(), rho.data(), tag.data(), nullptr));
  char wall[256];
  std::snprintf(wall, sizeof wall, "fix ywall all wall/granFix 1.0e7 NULL 0.5 NULL 0.4 1 yplane %.17g %.17g", b.lo[1], b.hi[1]);
  const char* script[] = {"atom_style sphere", "boundary p f p", "newton off", "communicate single vel yes",
                          "neighbor 0.05e-3 bin", "neigh_modify delay 0",
                          "pair_style gran/hertzFix/history 1.0e7 NULL 0.5 NULL 0.4 1", "pair_coeff * *",
                          "timestep 1.0e-6", "fix 1 all nve/sphere", "fix 2 all gravity 9.81 vector 0 -1 0",
                          "fix 3 all fdrag", wall};
  for (const char* line : script) {
    const char* err = sf_lammps_command(ptr, line);
    if (err) {
      std::printf("FAIL script line `%s`: %s\n", line, err);
      MPI_Abort(MPI_COMM_WORLD, 1);
    }
  }
  return ptr;
}

// owned atoms of an engine: tag, x, v (host)
void fetch(void* ptr, std::vector<int>& tag, std::vector<double>& x, std::vector<double>& v)
{
  const int n = sf_lammps_get_local_n(ptr);
  tag.resize(n);
  x.resize(3 * (size_t)n);
  v.resize(3 * (size_t)n);
  std::vector<int> foam(n), lmp(n);
  CHECK(sf_lammps_get_local_info(ptr, x.data(), v.data(), foam.data(), lmp.data(), tag.data()));
}

}  // namespace
