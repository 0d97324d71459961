// A C++ caller that OWNS LAMMPS the way lammpsFoam does (softParticleCloud.C:57-62,80,104-106,119,131,153,189,192,
// 212,227,838,893,900,914,1198,1231,357): MPI_Comm_dup, `new LAMMPS(0, NULL, comm)`, every script line through
// `lmp->input->one(line)`, the patched-library calls with the LAMMPS object as their void*, `delete lmp`.
// It includes the same five headers as lammpsFoam/include/LammpsCollection.H, found on the include path
// include/lammps_shim (this repo) + a stand-in mpi.h (tests/c_abi/fake_mpi_{int,ptr}: both MPI_Comm ABI families).
// Own code written against those names -- not a copy of the reference file.  Built by tests/test_c_abi.py with
// -std=c++98 and -std=c++17, run on the GPU.  It is written for N ranks the way initLammps is (`npArray = new
// int[nprocs]`, `nLocal = npArray[myrank]`, :122-132): under `mpirun -np N` every rank runs it, LAMMPS -- here the
// engine -- decomposes itself from the script's `processors` line, and NOT ONE sf_* call appears below.
// Rank 0 prints "OK <nGlobal> <ymean_before> <ymean_after> <n_after_create_delete> <checksum_x> <checksum_v> <ranks>
// <atoms that changed rank>".
#include "mpi.h"
#include "lammps.h"
#include "input.h"
#include "atom.h"
#include "library.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace LAMMPS_NS;

int main(int argc, char** argv)
{
  const char* script = argc > 1 ? argv[1] : "in.lammps";
#ifdef SHIM_DRIVER_REAL_MPI   // built against a real MPI library (the image's MPICH) instead of the stand-in headers
  MPI_Init(&argc, &argv);
#endif
  int nprocs = 1, myrank = 0;   // (Pstream::nProcs() / myProcNo(), softParticleCloud.C:122-123)
  MPI_Comm_size(MPI_COMM_WORLD, &nprocs);
  MPI_Comm_rank(MPI_COMM_WORLD, &myrank);
  MPI_Comm commLammps;
  MPI_Comm_dup(MPI_COMM_WORLD, &commLammps);
  LAMMPS* lmp = new LAMMPS(0, NULL, commLammps);

  // every rank runs every line (the reference reads on the master and MPI_Bcasts each line, :84-106)
  std::FILE* fp = std::fopen(script, "r");
  if (!fp) {
    std::printf("FAIL cannot open %s\n", script);
    return 1;
  }
  lammps_sync(lmp);
  char line[1024];
  int nlines = 0, ntimestep_lines = 0;
  while (std::fgets(line, sizeof line, fp)) {
    lmp->input->one(line);
    nlines++;
    if (std::strstr(line, "timestep")) ntimestep_lines++;
  }
  std::fclose(fp);

  const int nGlobal = lammps_get_global_n(lmp);
  std::vector<int> npArray(nprocs, 0);          // `new int[nprocs]`, :125
  lammps_get_initial_np(lmp, &npArray[0]);
  const int n = npArray[myrank];                // :132
  int nsum = 0;
  for (int r = 0; r < nprocs; r++) nsum += npArray[r];
  if (nGlobal <= 0 || nsum != nGlobal || n != lammps_get_local_n(lmp)) {
    std::printf("FAIL counts: rank %d holds %d, sum %d, global %d\n", myrank, n, nsum, nGlobal);
    return 1;
  }
  std::vector<double> x(3 * n + 3), v(3 * n + 3), d(n + 1), rho(n + 1);
  std::vector<int> tag(n + 1), lmpCpuId(n + 1), type(n + 1);
  lammps_get_initial_info(lmp, &x[0], &v[0], &d[0], &rho[0], &tag[0], &lmpCpuId[0], &type[0]);
  double part[2] = {0.0, 0.0}, y0 = 0.0;
  for (int i = 0; i < n; i++) part[0] += x[3 * i + 1];
  MPI_Allreduce(&part[0], &y0, 1, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
  y0 /= nGlobal;
  for (int i = 0; i < n; i++)
    if (lmpCpuId[i] != myrank) {
      std::printf("FAIL lmpCpuId %d on rank %d\n", lmpCpuId[i], myrank);
      return 1;
    }

  const std::vector<int> tag0(tag.begin(), tag.begin() + n);
  lammps_step(lmp, 0);
  double box[6];
  lammps_get_local_domain(lmp, box);
  const double dtIn = lammps_get_timestep(lmp);
  lammps_set_timestep(lmp, dtIn);   // (adjustLampTimestep writes back the reconciled value)
  if (!(box[1] > box[0]) || !(dtIn > 0.0) || ntimestep_lines != 1) {
    std::printf("FAIL domain/timestep\n");
    return 1;
  }
  // every atom this rank was handed lies in the sub-domain it reports (x and z are periodic: inside [lo, hi))
  for (int i = 0; i < n; i++)
    for (int c = 0; c < 3; c += 2)
      if (nprocs > 1 && (x[3 * i + c] < box[2 * c] || x[3 * i + c] >= box[2 * c + 1])) {
        std::printf("FAIL rank %d: atom %d at %g outside [%g, %g) in dimension %d\n", myrank, tag[i], x[3 * i + c],
                    box[2 * c], box[2 * c + 1], c);
        return 1;
      }
  double whole[6];   // the whole box = the union of the sub-domains
  for (int k = 0; k < 3; k++) {
    double lo = -box[2 * k], hi = box[2 * k + 1], glo = 0.0, ghi = 0.0;
    MPI_Allreduce(&lo, &glo, 1, MPI_DOUBLE, MPI_MAX, MPI_COMM_WORLD);
    MPI_Allreduce(&hi, &ghi, 1, MPI_DOUBLE, MPI_MAX, MPI_COMM_WORLD);
    whole[2 * k] = -glo;
    whole[2 * k + 1] = ghi;
  }

  // coupled loop: an upward fluid force of twice the weight, rows in reverse order (matched by tag)
  int nLocal = n;
  const double pi = 3.14159265358979323846;
  std::vector<double> mass(nGlobal + 2, 0.0);   // by tag (monodisperse here; a migrating atom keeps its force)
  {
    std::vector<double> mine(nGlobal + 2, 0.0);
    for (int i = 0; i < n; i++) mine[tag[i]] = rho[i] * pi * d[i] * d[i] * d[i] / 6.0;
    MPI_Allreduce(&mine[0], &mass[0], nGlobal + 2, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
  }
  for (int cfd = 0; cfd < 3; cfd++) {
    std::vector<double> fdrag(3 * nLocal + 3, 0.0), DuDt(3 * nLocal + 3, 0.0);
    std::vector<int> foamCpuId(nLocal + 1, 0), tagIn(nLocal + 1);
    for (int k = 0; k < nLocal; k++) {
      const int i = nLocal - 1 - k;
      fdrag[3 * k + 1] = 2.0 * mass[tag[i]] * 9.8;
      tagIn[k] = tag[i];
    }
    lammps_put_local_info(lmp, nLocal, &fdrag[0], &DuDt[0], &foamCpuId[0], &tagIn[0]);
    lammps_step(lmp, 20);
    nLocal = lammps_get_local_n(lmp);   // (atoms migrate between the ranks: :893-914 re-reads the count)
    x.resize(3 * nLocal + 3); v.resize(3 * nLocal + 3);
    tag.resize(nLocal + 1); lmpCpuId.resize(nLocal + 1); foamCpuId.resize(nLocal + 1);
    lammps_get_local_info(lmp, &x[0], &v[0], &foamCpuId[0], &lmpCpuId[0], &tag[0]);
  }
  // atoms that changed their rank since lammps_get_initial_info (Comm::exchange at the rebuilds)
  int movedLocal = 0, moved = 0;
  {
    std::vector<char> wasMine(nGlobal + 2, 0);
    for (int i = 0; i < n; i++) wasMine[tag0[i]] = 1;
    for (int i = 0; i < nLocal; i++) movedLocal += wasMine[tag[i]] ? 0 : 1;
    MPI_Allreduce(&movedLocal, &moved, 1, MPI_INT, MPI_SUM, MPI_COMM_WORLD);
  }
  // mean height and a tag-weighted checksum of positions and velocities over ALL ranks
  double loc[3] = {0.0, 0.0, 0.0}, glob[3] = {0.0, 0.0, 0.0};
  for (int i = 0; i < nLocal; i++) {
    // (x, z periodic: LAMMPS wraps a coordinate only when it reneighbours; compare wrapped positions)
    const double Lx = whole[1] - whole[0], Lz = whole[5] - whole[4];
    const double xw = x[3 * i] - whole[0] - Lx * std::floor((x[3 * i] - whole[0]) / Lx);
    const double zw = x[3 * i + 2] - whole[4] - Lz * std::floor((x[3 * i + 2] - whole[4]) / Lz);
    loc[0] += x[3 * i + 1];
    loc[1] += (tag[i] % 17 + 1) * (xw + 2.0 * x[3 * i + 1] + 3.0 * zw);
    loc[2] += (tag[i] % 13 + 1) * (v[3 * i] + 2.0 * v[3 * i + 1] + 3.0 * v[3 * i + 2]);
  }
  MPI_Allreduce(loc, glob, 3, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
  const double y1 = glob[0] / nGlobal;
  // SHIM_DUMP=<prefix>: every rank writes its atoms -- tag, wrapped position, velocity -- to <prefix>.<rank>, so that the
  // test can compare the N-rank run with the 1-rank run atom by atom instead of by checksum
  if (const char* prefix = std::getenv("SHIM_DUMP")) {
    char name[1024];
    std::snprintf(name, sizeof(name), "%s.%d", prefix, myrank);
    if (std::FILE* f = std::fopen(name, "w")) {
      const double Lx = whole[1] - whole[0], Lz = whole[5] - whole[4];
      std::fprintf(f, "# %.17g %.17g\n", Lx, Lz);
      for (int i = 0; i < nLocal; i++) {
        const double xw = x[3 * i] - whole[0] - Lx * std::floor((x[3 * i] - whole[0]) / Lx);
        const double zw = x[3 * i + 2] - whole[4] - Lz * std::floor((x[3 * i + 2] - whole[4]) / Lz);
        std::fprintf(f, "%d %.17g %.17g %.17g %.17g %.17g %.17g\n", tag[i], xw, x[3 * i + 1], zw, v[3 * i], v[3 * i + 1],
                     v[3 * i + 2]);
      }
      std::fclose(f);
    }
  }

  // particle injection / removal (softParticleCloud.C:1198, :1231): collective calls; a rank creates what falls into
  // its sub-domain (npAdd = 0 elsewhere) and deletes the listed atoms it owns
  double pos[3] = {0.5 * (whole[0] + whole[1]), 0.9 * whole[3], 0.5 * (whole[4] + whole[5])};
  lammps_get_local_domain(lmp, box);
  const bool mine = pos[0] >= box[0] && pos[0] < box[1] && pos[1] >= box[2] && pos[1] <= box[3] && pos[2] >= box[4] &&
                    pos[2] < box[5];
  double newtag[1] = {(double)(nGlobal + 1)};
  double vel[3] = {0.0, 0.0, 0.0};
  lammps_create_particle(lmp, mine ? 1 : 0, pos, newtag, d[0] > 0.0 ? d[0] : 5.0e-4, 2650.0, 1, vel);
  int dead[2] = {1, 2};
  lammps_delete_particle(lmp, dead, 2);
  int nAfterLocal = lammps_get_local_n(lmp), nAfter = 0;
  MPI_Allreduce(&nAfterLocal, &nAfter, 1, MPI_INT, MPI_SUM, MPI_COMM_WORLD);
  if (lammps_get_global_n(lmp) != nAfter) {
    std::printf("FAIL natoms %d after create / delete, the ranks hold %d\n", lammps_get_global_n(lmp), nAfter);
    return 1;
  }
  lammps_step(lmp, 5);
  int nEndLocal = lammps_get_local_n(lmp), nEnd = 0;
  MPI_Allreduce(&nEndLocal, &nEnd, 1, MPI_INT, MPI_SUM, MPI_COMM_WORLD);
  if (nEnd != nAfter) {
    std::printf("FAIL %d atoms after the last run, %d before\n", nEnd, nAfter);
    return 1;
  }

  delete lmp;   // finishLammps, :357
#ifdef SHIM_DRIVER_REAL_MPI
  MPI_Finalize();
#endif
  if (myrank == 0)
    std::printf("OK %d %.12g %.12g %d %.15g %.15g %d %d\n", nGlobal, y0, y1, nAfter, glob[1], glob[2], nprocs, moved);
  return 0;
}
