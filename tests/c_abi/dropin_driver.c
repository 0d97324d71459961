/* A C caller written against the reference's spelling of the LAMMPS library interface
 * (interfaceToLammps/library.h:29-63) -- what lammpsFoam/softParticleCloud.C does per CFD step: script lines through
 * lammps_command, lammps_get_initial_np/info, then put fluid drag by tag -> lammps_step(n) -> get positions.
 * Built by tests/test_c_abi.py with gcc (as C) and g++ (as C++) against include/sedifoam_amd.h and run on the GPU.
 * Prints "OK <n> <ymean_before> <ymean_after>" or "FAIL <what>". */
#define SEDIFOAM_AMD_LAMMPS_NAMES
#include "sedifoam_amd.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

static int cmd(void *lmp, const char *line)
{
  char buf[512];
  snprintf(buf, sizeof buf, "%s", line);
  if (lammps_command(lmp, buf)) {
    printf("FAIL command: %s (%s)\n", line, sf_last_error());
    return 1;
  }
  return 0;
}

int main(int argc, char **argv)
{
  void *lmp = NULL;
  const char *data = argc > 1 ? argv[1] : "bed.in";
  char line[600];
  int np[1], n, i, k;
  lammps_open(0, NULL, 0, &lmp);
  if (!lmp) {
    printf("FAIL open: %s\n", sf_last_error());
    return 1;
  }
  snprintf(line, sizeof line, "read_data %s", data);
  if (cmd(lmp, "atom_style sphere") || cmd(lmp, "boundary pp ff pp") || cmd(lmp, "newton off") ||
      cmd(lmp, "communicate single vel yes") || cmd(lmp, line) || cmd(lmp, "neighbor 1.0e-4 bin") ||
      cmd(lmp, "neigh_modify delay 0") ||
      cmd(lmp, "pair_style gran/hertzFix/history 1.0e7 NULL 0.5 NULL 0.4 1") || cmd(lmp, "pair_coeff * *") ||
      cmd(lmp, "timestep 1e-6") || cmd(lmp, "fix 1 all nve/sphere") ||
      cmd(lmp, "fix 2 all gravity 9.8 vector 0 -1 0") || cmd(lmp, "fix 3 all fdrag") ||
      cmd(lmp, "fix ywall all wall/granFix 1.0e7 NULL 0.5 NULL 0.4 1 yplane 0.0 0.004"))
    return 1;
  lammps_get_initial_np(lmp, np);
  n = np[0];
  if (n <= 0 || n != lammps_get_global_n(lmp) || n != lammps_get_local_n(lmp)) {
    printf("FAIL counts\n");
    return 1;
  }
  {
    double *x = (double *)malloc(sizeof(double) * 3 * n), *v = (double *)malloc(sizeof(double) * 3 * n);
    double *d = (double *)malloc(sizeof(double) * n), *rho = (double *)malloc(sizeof(double) * n);
    double *fd = (double *)calloc(3 * (size_t)n, sizeof(double)), *du = (double *)calloc(3 * (size_t)n, sizeof(double));
    int *tag = (int *)malloc(sizeof(int) * n), *cpu = (int *)malloc(sizeof(int) * n);
    int *type = (int *)malloc(sizeof(int) * n), *foam = (int *)calloc(n, sizeof(int));
    double y0 = 0.0, y1 = 0.0, lift;
    lammps_get_initial_info(lmp, x, v, d, rho, tag, cpu, type);
    for (i = 0; i < n; i++) y0 += x[3 * i + 1] / n;
    lammps_step(lmp, 0);
    /* an upward fluid force of twice the weight on every particle, rows handed over in reverse order (matched by tag) */
    for (i = 0; i < n; i++) {
      const double m = rho[i] * 3.14159265358979323846 * d[i] * d[i] * d[i] / 6.0;
      fd[3 * (n - 1 - i) + 1] = 2.0 * 9.8 * m;
      foam[n - 1 - i] = 0;
    }
    {
      int *rtag = (int *)malloc(sizeof(int) * n);
      for (i = 0; i < n; i++) rtag[n - 1 - i] = tag[i];
      for (k = 0; k < 5; k++) {
        lammps_put_local_info(lmp, n, fd, du, foam, rtag);
        lammps_step(lmp, 100);
        lammps_get_local_info(lmp, x, v, foam, cpu, rtag);
        for (i = 0; i < n; i++) {   /* get returns the engine's order: rebuild the by-row force for the next put */
          const double m = 2650.0 * 3.14159265358979323846 * d[0] * d[0] * d[0] / 6.0;
          fd[3 * i] = 0.0; fd[3 * i + 1] = 2.0 * 9.8 * m; fd[3 * i + 2] = 0.0;
        }
      }
      free(rtag);
    }
    for (i = 0; i < n; i++) y1 += x[3 * i + 1] / n;
    lift = y1 - y0;
    if (!(lift > 0.0) || !isfinite(lift)) {
      printf("FAIL bed did not rise: %g\n", lift);
      return 1;
    }
    if (fabs(lammps_get_timestep(lmp) - 1e-6) > 1e-18) {
      printf("FAIL timestep\n");
      return 1;
    }
    printf("OK %d %.9g %.9g\n", n, y0, y1);
    free(x); free(v); free(d); free(rho); free(fd); free(du); free(tag); free(cpu); free(type); free(foam);
  }
  lammps_close(lmp);
  return 0;
}
