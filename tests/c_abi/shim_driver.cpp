// A C++ caller that OWNS LAMMPS the way lammpsFoam does (softParticleCloud.C:57-62,80,104-106,119,131,153,189,192,
// 212,227,838,893,900,914,1198,1231,357): MPI_Comm_dup, `new LAMMPS(0, NULL, comm)`, every script line through
// `lmp->input->one(line)`, the patched-library calls with the LAMMPS object as their void*, `delete lmp`.
// It includes the same five headers as lammpsFoam/include/LammpsCollection.H, found on the include path
// include/lammps_shim (this repo) + a stand-in mpi.h (tests/c_abi/fake_mpi_{int,ptr}: both MPI_Comm ABI families).
// Own code written against those names -- not a copy of the reference file.  Built by tests/test_c_abi.py with
// -std=c++98 and -std=c++17, run on the GPU.  Prints "OK <n> <ymean_before> <ymean_after> <n_after_create_delete>".
#include "mpi.h"
#include "lammps.h"
#include "input.h"
#include "atom.h"
#include "library.h"

#include <cstdio>
#include <cstring>
#include <vector>

using namespace LAMMPS_NS;

int main(int argc, char** argv)
{
  const char* script = argc > 1 ? argv[1] : "in.lammps";
#ifdef SHIM_DRIVER_REAL_MPI   // built against a real MPI library (the image's MPICH) instead of the stand-in headers
  MPI_Init(&argc, &argv);
#endif
  MPI_Comm commLammps;
  MPI_Comm_dup(MPI_COMM_WORLD, &commLammps);
  LAMMPS* lmp = new LAMMPS(0, NULL, commLammps);

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
  int npArray[1] = {0};
  lammps_get_initial_np(lmp, npArray);
  const int n = npArray[0];
  if (n <= 0 || n != nGlobal) {
    std::printf("FAIL counts %d %d\n", n, nGlobal);
    return 1;
  }
  std::vector<double> x(3 * n), v(3 * n), d(n), rho(n);
  std::vector<int> tag(n), lmpCpuId(n), type(n);
  lammps_get_initial_info(lmp, &x[0], &v[0], &d[0], &rho[0], &tag[0], &lmpCpuId[0], &type[0]);
  double y0 = 0.0;
  for (int i = 0; i < n; i++) y0 += x[3 * i + 1] / n;

  lammps_step(lmp, 0);
  double box[6];
  lammps_get_local_domain(lmp, box);
  const double dtIn = lammps_get_timestep(lmp);
  lammps_set_timestep(lmp, dtIn);   // (adjustLampTimestep writes back the reconciled value)
  if (!(box[1] > box[0]) || !(dtIn > 0.0) || ntimestep_lines != 1) {
    std::printf("FAIL domain/timestep\n");
    return 1;
  }

  // coupled loop: an upward fluid force of twice the weight, rows in reverse order (matched by tag)
  std::vector<double> fdrag(3 * n, 0.0), DuDt(3 * n, 0.0);
  std::vector<int> foamCpuId(n, 0), tagIn(n);
  const double pi = 3.14159265358979323846;
  for (int k = 0; k < n; k++) {
    const int i = n - 1 - k;
    const double m = rho[i] * pi * d[i] * d[i] * d[i] / 6.0;
    fdrag[3 * k + 1] = 2.0 * m * 9.8;
    tagIn[k] = tag[i];
  }
  int nLocal = n;
  for (int cfd = 0; cfd < 3; cfd++) {
    lammps_put_local_info(lmp, nLocal, &fdrag[0], &DuDt[0], &foamCpuId[0], &tagIn[0]);
    lammps_step(lmp, 20);
    nLocal = lammps_get_local_n(lmp);
    lammps_get_local_info(lmp, &x[0], &v[0], &foamCpuId[0], &lmpCpuId[0], &tag[0]);
  }
  double y1 = 0.0;
  for (int i = 0; i < nLocal; i++) y1 += x[3 * i + 1] / nLocal;

  // particle injection / removal (softParticleCloud.C:1198, :1231)
  double pos[3] = {0.5 * (box[0] + box[1]), 0.9 * box[3], 0.5 * (box[4] + box[5])};
  double newtag[1] = {(double)(n + 1)};
  double vel[3] = {0.0, 0.0, 0.0};
  lammps_create_particle(lmp, 1, pos, newtag, d[0], rho[0], 1, vel);
  int dead[2] = {1, 2};
  lammps_delete_particle(lmp, dead, 2);
  const int nAfter = lammps_get_local_n(lmp);
  lammps_step(lmp, 5);

  delete lmp;   // finishLammps, :357
#ifdef SHIM_DRIVER_REAL_MPI
  MPI_Finalize();
#endif
  std::printf("OK %d %.12g %.12g %d\n", n, y0, y1, nAfter);
  return 0;
}
