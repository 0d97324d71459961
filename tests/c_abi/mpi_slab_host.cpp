// An MPI / C++ host that drives N x-slabs of one bed through the C-ABI alone -- the calls a lammpsFoam built against
// libsedifoam_amd.so makes on a decomposed case: every MPI rank opens one engine (`lammps_open(0, NULL, comm, &ptr)`
// shape), feeds it the input-script lines, owns the atoms of its slab, then
//     sf_dem_comm_unique_id (rank 0) -> MPI_Bcast -> sf_slab_init -> sf_slab_setup -> sf_slab_step(n) ...
// No Python, no torch: the sub-step loop, the forward halo, the rebuild vote, migration and the border exchange all
// run inside the library (csrc/sf_halo_rccl.hip).  Rank 0 also runs the WHOLE bed on a second, single-domain engine;
// the slabs' atoms are gathered by tag with MPI_Gatherv and compared with it.
//
// Built and run by tests/test_halo_gpu.py (g++ against the image's MPICH, `mpirun -np 2|3`).  On a one-GPU box the
// ranks share the GPU and SF_RCCL_LIB points the library at tests/c_abi/standin_rccl.cpp instead of librccl (which
// refuses two ranks on one device); on a multi-GPU node the same binary runs on librccl with one GPU per rank
// (HIP_VISIBLE_DEVICES per rank).  Prints "OK ranks <N> atoms <n> rebuilds <r> max|dx| <e> rel(v) <e>" or "FAIL ...".
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

int main(int argc, char** argv)
{
  MPI_Init(&argc, &argv);
  int rank = 0, world = 1;
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  MPI_Comm_size(MPI_COMM_WORLD, &world);
  const int ncx = argc > 1 ? std::atoi(argv[1]) : 8;
  const int nsteps = argc > 2 ? std::atoi(argv[2]) : 50;
  const int nruns = 2;
  const Bed bed = make_bed(ncx, 5, 5, 0.5);
  const int n = (int)bed.tag.size();
  const double L = bed.hi[0] - bed.lo[0], w = L / world;

  // this rank's slab: x in [lo + rank w, lo + (rank + 1) w)
  std::vector<int> mine;
  for (int i = 0; i < n; i++) {
    int r = (int)std::floor((bed.x[3 * i] - bed.lo[0]) / w);
    r = r < 0 ? 0 : (r >= world ? world - 1 : r);
    if (r == rank) mine.push_back(i);
  }
  MPI_Comm comm;
  MPI_Comm_dup(MPI_COMM_WORLD, &comm);
  void* slab = make_engine(bed, mine, comm);
  char id[128];
  if (rank == 0) CHECK(sf_dem_comm_unique_id(id));
  MPI_Bcast(id, 128, MPI_CHAR, 0, MPI_COMM_WORLD);
  CHECK(sf_slab_init(slab, id, rank, world, bed.lo[0], bed.hi[0], 1));
  CHECK(sf_slab_setup(slab));
  for (int r = 0; r < nruns; r++) CHECK(sf_slab_step(slab, nsteps));
  const long long rebuilds = sf_slab_rebuild_count(slab);

  std::vector<int> tag;
  std::vector<double> x, v;
  fetch(slab, tag, x, v);
  int nl = (int)tag.size();
  std::vector<int> counts(world), displs(world), counts3(world), displs3(world);
  MPI_Gather(&nl, 1, MPI_INT, counts.data(), 1, MPI_INT, 0, MPI_COMM_WORLD);
  int total = 0;
  for (int r = 0; r < world; r++) {
    displs[r] = total;
    displs3[r] = 3 * total;
    counts3[r] = 3 * counts[r];
    total += counts[r];
  }
  std::vector<int> gtag(rank == 0 ? total : 1);
  std::vector<double> gx(rank == 0 ? 3 * (size_t)total : 1), gv(rank == 0 ? 3 * (size_t)total : 1);
  MPI_Gatherv(tag.data(), nl, MPI_INT, gtag.data(), counts.data(), displs.data(), MPI_INT, 0, MPI_COMM_WORLD);
  MPI_Gatherv(x.data(), 3 * nl, MPI_DOUBLE, gx.data(), counts3.data(), displs3.data(), MPI_DOUBLE, 0, MPI_COMM_WORLD);
  MPI_Gatherv(v.data(), 3 * nl, MPI_DOUBLE, gv.data(), counts3.data(), displs3.data(), MPI_DOUBLE, 0, MPI_COMM_WORLD);

  int fail = 0;
  if (rank == 0) {
    // the same bed on one engine
    std::vector<int> all(n);
    for (int i = 0; i < n; i++) all[i] = i;
    void* one = make_engine(bed, all, comm);
    CHECK(sf_dem_setup(one));
    for (int r = 0; r < nruns; r++) CHECK(sf_lammps_step(one, nsteps));
    std::vector<int> rt;
    std::vector<double> rx, rv;
    fetch(one, rt, rx, rv);
    std::vector<int> where(n + 1, -1);
    for (int i = 0; i < (int)rt.size(); i++) where[rt[i]] = i;
    double ex = 0.0, ev = 0.0, sv = 0.0;
    std::vector<char> seen(n + 1, 0);
    if (total != n) fail = 1;
    for (int k = 0; k < total && !fail; k++) {
      const int t = gtag[k];
      if (t < 1 || t > n || seen[t] || where[t] < 0) {
        fail = 1;
        break;
      }
      seen[t] = 1;
      const int i = where[t];
      for (int c = 0; c < 3; c++) {
        double dx = gx[3 * k + c] - rx[3 * i + c];
        if (c == 0) dx -= L * std::round(dx / L);   // (an atom that crossed the periodic face is wrapped at a rebuild)
        ex = std::fmax(ex, std::fabs(dx));
        ev = std::fmax(ev, std::fabs(gv[3 * k + c] - rv[3 * i + c]));
        sv = std::fmax(sv, std::fabs(rv[3 * i + c]));
      }
    }
    if (fail) std::printf("FAIL the slabs hold %d atoms, the bed %d (or a tag is missing / doubled)\n", total, n);
    else if (!(ex <= 1e-12) || !(ev <= 1e-9 * sv) || rebuilds < 3) {
      std::printf("FAIL ranks %d atoms %d rebuilds %lld max|dx| %.3e rel(v) %.3e\n", world, n, rebuilds, ex, ev / sv);
      fail = 1;
    } else {
      std::printf("OK ranks %d atoms %d rebuilds %lld max|dx| %.3e rel(v) %.3e\n", world, n, rebuilds, ex, ev / sv);
    }
    CHECK(sf_lammps_close(one));
  }
  CHECK(sf_lammps_close(slab));
  MPI_Bcast(&fail, 1, MPI_INT, 0, MPI_COMM_WORLD);
  MPI_Finalize();
  return fail;
}
