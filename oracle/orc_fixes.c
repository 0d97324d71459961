/* orc_fixes.c -- oracle restatement of the non-contact particle forces (TEST INFRASTRUCTURE ONLY).
 *
 *   orc_fix_cohesive        : interfaceToLammps/fix_cohesive.cpp:138-263
 *   orc_lubricate_init      : interfaceToLammps/pair_lubricate_poly.cpp:450-577
 *   orc_pair_lubricate_poly : interfaceToLammps/pair_lubricate_poly.cpp:65-444
 *   orc_fix_fluid_drag      : interfaceToLammps/fix_fluid_drag.cpp:114-164
 *   orc_nve_sphere_*        : [3P] LAMMPS 1Feb14 fix_nve_sphere.cpp (called by every in.lammps:
 *                             "fix 1 all nve/sphere", e.g. cases/auto-testing/test-cases/xiaocase3/in.lammps:21)
 *   orc_fix_gravity         : [3P] LAMMPS 1Feb14 fix_gravity.cpp ("fix 2 all gravity g vector 0 -1 0")
 */
#include <math.h>
#include <stddef.h>
#include "sedifoam_oracle.h"

#define ORC_PI 3.14159265358979323846

int orc_fix_cohesive(double ah, double lam, double smin, double smax, int opt, int nlocal,
                     int newton_pair, const double *x, const double *radius, const int *mask,
                     int groupbit, const orc_neighlist *list, double *f)
{
  const double PInv = 0.25 / atan(1.0);                             /* :154 */
  int ii, jj, k;
  if (opt != 0 && opt != 1) return -1;                              /* :262 */
  for (ii = 0; ii < nlocal; ii++) {                                 /* :165 / :216 */
    int i = list->ilist[ii];
    if (!(mask[i] & groupbit)) continue;
    double radi = radius[i];
    for (jj = list->first[ii]; jj < list->first[ii + 1]; jj++) {
      int j = list->jlist[jj];                                      /* no NEIGHMASK: :176 */
      double del3[3];
      for (k = 0; k < 3; k++) del3[k] = x[3 * i + k] - x[3 * j + k];
      double rsq = del3[0] * del3[0] + del3[1] * del3[1] + del3[2] * del3[2];
      double radsum = radi + radius[j];
      if (!(rsq < (radsum + smax) * (radsum + smax))) continue;     /* :184 / :236 */
      double r = sqrt(rsq);
      double del = r - radsum;
      double ccel;
      if (opt == 0) {                                               /* :187-195 */
        if (del > lam * PInv)
          ccel = -ah * radsum * lam *
                 (6.4988e-3 - 4.5316e-4 * lam / del + 1.1326e-5 * lam * lam / del / del) / del /
                 del / del;
        else if (del > smin)
          ccel = -ah * (lam + 22.242 * del) * radsum * lam / 24.0 / (lam + 11.121 * del) /
                 (lam + 11.121 * del) / del / del;
        else
          ccel = -ah * (lam + 22.242 * smin) * radsum * lam / 24.0 / (lam + 11.121 * smin) /
                 (lam + 11.121 * smin) / smin / smin;
      } else {                                                      /* :239-244 */
        if (del > smin)
          ccel = -ah * pow(radsum, 6) / 6.0 / del / del / (r + radsum) / (r + radsum) / r / r / r;
        else
          ccel = -ah * pow(radsum, 6) / 6.0 / smin / smin / (smin + 2.0 * radsum) /
                 (smin + 2.0 * radsum) / (smin + radsum) / (smin + radsum) / (smin + radsum);
      }
      double rinv = 1 / r;
      for (k = 0; k < 3; k++) {
        double c = del3[k] * ccel * rinv;                           /* :198-203 */
        f[3 * i + k] += c;
        if (newton_pair || j < nlocal) f[3 * j + k] -= c;           /* :205-209 */
      }
    }
  }
  return 0;
}

double orc_particle_volume(int nlocal, const double *radius)
{
  int i;
  double volP = 0.0;
  for (i = 0; i < nlocal; i++) volP += (4.0 / 3.0) * ORC_PI * pow(radius[i], 3.0); /* :540-542 */
  return volP;
}

/* volP = particle volume of ALL ranks (MPI_Allreduce, pair_lubricate_poly.cpp:543) */
void orc_lubricate_init_vol(orc_lub_params *p, double volP, double vol_T)
{
  double vol_f = volP / vol_T;
  double mu = p->mu;
  if (!p->flagVF) vol_f = 0;                                        /* :547 */
  if (p->flaglog == 0) {                                            /* :551-559 */
    p->R0 = 6 * ORC_PI * mu * (1.0 + 2.16 * vol_f);
    p->RT0 = 8 * ORC_PI * mu;
    p->RS0 = 20.0 / 3.0 * ORC_PI * mu * (1.0 + 3.33 * vol_f + 2.80 * vol_f * vol_f);
  } else {
    p->R0 = 6 * ORC_PI * mu * (1.0 + 2.725 * vol_f - 6.583 * vol_f * vol_f);
    p->RT0 = 8 * ORC_PI * mu * (1.0 + 0.749 * vol_f - 2.469 * vol_f * vol_f);
    p->RS0 = 20.0 / 3.0 * ORC_PI * mu * (1.0 + 3.64 * vol_f - 6.95 * vol_f * vol_f);
  }
}

void orc_lubricate_init(orc_lub_params *p, int nlocal_all, const double *radius, double vol_T)
{
  orc_lubricate_init_vol(p, orc_particle_volume(nlocal_all, radius), vol_T);
}

void orc_pair_lubricate_poly(const orc_lub_params *p, int nlocal, const double *x,
                             const double *v, const double *omega, const double *radius,
                             const orc_neighlist *list, double *f, double *torque)
{
  const double vxmu2f = p->vxmu2f, mu = p->mu;
  const double cutsq = p->cut_global * p->cut_global;
  int ii, jj, k;
  (void)nlocal;
  for (ii = 0; ii < list->inum; ii++) {
    int i = list->ilist[ii];
    double radi = radius[i];
    double wi[3] = {omega[3 * i], omega[3 * i + 1], omega[3 * i + 2]};
    if (p->flagfld) {                                               /* :213-220 */
      const double radi3 = radi * radi * radi;
      for (k = 0; k < 3; k++) {
        f[3 * i + k] -= vxmu2f * p->R0 * radi * v[3 * i + k];
        torque[3 * i + k] -= vxmu2f * p->RT0 * radi3 * wi[k];
      }
    }
    if (!p->flagHI) continue;                                       /* :230 */
    for (jj = list->first[ii]; jj < list->first[ii + 1]; jj++) {
      int j = list->jlist[jj];
      double del[3];
      for (k = 0; k < 3; k++) del[k] = x[3 * i + k] - x[3 * j + k];
      double rsq = del[0] * del[0] + del[1] * del[1] + del[2] * del[2];
      double radj = radius[j];
      if (!(rsq < cutsq)) continue;                                 /* :241 */
      double r = sqrt(rsq);
      double wj[3] = {omega[3 * j], omega[3 * j + 1], omega[3 * j + 2]};
      double xl[3], jl[3], vi[3], vj[3];
      for (k = 0; k < 3; k++) {
        xl[k] = -del[k] / r * radi;                                 /* :252-257 */
        jl[k] = -del[k] / r * radj;
      }
      /* :264-282 with Ef = 0: the "- (Ef . xl)" terms subtract an exact 0.0 */
      vi[0] = v[3 * i + 0] + (wi[1] * xl[2] - wi[2] * xl[1]) - (0.0 * xl[0] + 0.0 * xl[1] + 0.0 * xl[2]);
      vi[1] = v[3 * i + 1] + (wi[2] * xl[0] - wi[0] * xl[2]) - (0.0 * xl[0] + 0.0 * xl[1] + 0.0 * xl[2]);
      vi[2] = v[3 * i + 2] + (wi[0] * xl[1] - wi[1] * xl[0]) - (0.0 * xl[0] + 0.0 * xl[1] + 0.0 * xl[2]);
      vj[0] = v[3 * j + 0] - (wj[1] * jl[2] - wj[2] * jl[1]) + (0.0 * jl[0] + 0.0 * jl[1] + 0.0 * jl[2]);
      vj[1] = v[3 * j + 1] - (wj[2] * jl[0] - wj[0] * jl[2]) + (0.0 * jl[0] + 0.0 * jl[1] + 0.0 * jl[2]);
      vj[2] = v[3 * j + 2] - (wj[0] * jl[1] - wj[1] * jl[0]) + (0.0 * jl[0] + 0.0 * jl[1] + 0.0 * jl[2]);

      double h_sep = r - radi - radj;                               /* :286 */
      if (r < p->cut_inner) h_sep = 100 * radi + 100 * radj;        /* :294-295 (reference's edit) */
      h_sep = h_sep / radi;                                         /* :301-303 */
      double beta0 = radj / radi;
      double beta1 = 1.0 + beta0;
      double a_sq, a_sh = 0.0, a_pu = 0.0;
      if (p->flaglog) {                                             /* :307-323 */
        a_sq = beta0 * beta0 / beta1 / beta1 / h_sep +
               (1.0 + 7.0 * beta0 + beta0 * beta0) / 5.0 / pow(beta1, 3.0) * log(1.0 / h_sep);
        a_sq += (1.0 + 18.0 * beta0 - 29.0 * beta0 * beta0 + 18.0 * pow(beta0, 3.0) +
                 pow(beta0, 4.0)) / 21.0 / pow(beta1, 4.0) * h_sep * log(1.0 / h_sep);
        a_sq *= 6.0 * ORC_PI * mu * radi;
        a_sh = 4.0 * beta0 * (2.0 + beta0 + 2.0 * beta0 * beta0) / 15.0 / pow(beta1, 3.0) *
               log(1.0 / h_sep);
        a_sh += 4.0 * (16.0 - 45.0 * beta0 + 58.0 * beta0 * beta0 - 45.0 * pow(beta0, 3.0) +
                       16.0 * pow(beta0, 4.0)) / 375.0 / pow(beta1, 4.0) * h_sep * log(1.0 / h_sep);
        a_sh *= 6.0 * ORC_PI * mu * radi;
        a_pu = beta0 * (4.0 + beta0) / 10.0 / beta1 / beta1 * log(1.0 / h_sep);
        a_pu += (32.0 - 33.0 * beta0 + 83.0 * beta0 * beta0 + 43.0 * pow(beta0, 3.0)) / 250.0 /
                pow(beta1, 3.0) * h_sep * log(1.0 / h_sep);
        a_pu *= 8.0 * ORC_PI * mu * pow(radi, 3.0);
      } else
        a_sq = 6.0 * ORC_PI * mu * radi * (beta0 * beta0 / beta1 / beta1 / h_sep); /* :324 */

      double vr[3], vn[3], vt[3], F[3];
      for (k = 0; k < 3; k++) vr[k] = vi[k] - vj[k];                /* :329-331 */
      double vnnr = (vr[0] * del[0] + vr[1] * del[1] + vr[2] * del[2]) / r; /* :335 */
      for (k = 0; k < 3; k++) {
        vn[k] = vnnr * del[k] / r;                                  /* :336-338 */
        vt[k] = vr[k] - vn[k];
        F[k] = a_sq * vn[k];                                        /* :348-350 */
        if (p->flaglog) F[k] = F[k] + a_sh * vt[k];                 /* :354-358 */
        F[k] *= vxmu2f;                                             /* :362-364 */
        f[3 * i + k] -= F[k];                                       /* :368-370 */
      }
      if (p->flaglog) {                                             /* :374-398 */
        double t[3];
        t[0] = xl[1] * F[2] - xl[2] * F[1];
        t[1] = xl[2] * F[0] - xl[0] * F[2];
        t[2] = xl[0] * F[1] - xl[1] * F[0];
        for (k = 0; k < 3; k++) torque[3 * i + k] -= vxmu2f * t[k];
        double wdotn = ((wi[0] - wj[0]) * del[0] + (wi[1] - wj[1]) * del[1] +
                        (wi[2] - wj[2]) * del[2]) / r;
        for (k = 0; k < 3; k++) {
          double wt = (wi[k] - wj[k]) - wdotn * del[k] / r;
          torque[3 * i + k] -= vxmu2f * (a_pu * wt);
        }
      }
    }
  }
}

void orc_fix_fluid_drag(int nlocal, double dt, double carrier_rho, const double *v,
                        const double *rmass, const double *radius, const int *mask,
                        int groupbit, const double *ffluiddrag, const double *DuDt,
                        double *vOld, double *f)
{
  int i, k;
  for (i = 0; i < nlocal; i++) {
    if (!(mask[i] & groupbit)) continue;
    /* :147 -- the reference's pi literal is mistyped (…58917… instead of …58979…); kept */
    double rho = 3.0 * rmass[i] / (4.0 * 3.14159265358917323846 * radius[i] * radius[i] * radius[i]);
    for (k = 0; k < 3; k++) {
      double acc = ((v[3 * i + k] - vOld[3 * i + k]) / dt);         /* :148-150 */
      f[3 * i + k] += ffluiddrag[3 * i + k] +
                      carrier_rho / rho * 0.5 * rmass[i] * (DuDt[3 * i + k] - acc); /* :152-157 */
    }
    for (k = 0; k < 3; k++) vOld[3 * i + k] = v[3 * i + k];         /* :159-161 */
  }
}

/* [3P] FixNVESphere::initial_integrate : dtf = 0.5*dt*ftm2v (ftm2v = 1 in lj units),
 * dtfrotate = dtf / INERTIA with INERTIA = 0.4 */
void orc_nve_sphere_initial(int nlocal, double dt, double *x, double *v, double *omega,
                            const double *f, const double *torque, const double *radius,
                            const double *rmass)
{
  const double dtv = dt, dtf = 0.5 * dt;
  const double dtfrotate = dtf / 0.4;
  int i, k;
  for (i = 0; i < nlocal; i++) {
    double dtfm = dtf / rmass[i];
    for (k = 0; k < 3; k++) v[3 * i + k] += dtfm * f[3 * i + k];
    for (k = 0; k < 3; k++) x[3 * i + k] += dtv * v[3 * i + k];
    double dtirotate = dtfrotate / (radius[i] * radius[i] * rmass[i]);
    for (k = 0; k < 3; k++) omega[3 * i + k] += dtirotate * torque[3 * i + k];
  }
}

void orc_nve_sphere_final(int nlocal, double dt, double *v, double *omega, const double *f,
                          const double *torque, const double *radius, const double *rmass)
{
  const double dtf = 0.5 * dt;
  const double dtfrotate = dtf / 0.4;
  int i, k;
  for (i = 0; i < nlocal; i++) {
    double dtfm = dtf / rmass[i];
    for (k = 0; k < 3; k++) v[3 * i + k] += dtfm * f[3 * i + k];
    double dtirotate = dtfrotate / (radius[i] * radius[i] * rmass[i]);
    for (k = 0; k < 3; k++) omega[3 * i + k] += dtirotate * torque[3 * i + k];
  }
}

/* [3P] FixGravity::post_force, "vector" style: gvec = magnitude * dir/|dir| */
void orc_fix_gravity(int nlocal, double magnitude, const double dir[3], const double *rmass,
                     double *f)
{
  double len = sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
  double acc[3];
  int i, k;
  for (k = 0; k < 3; k++) acc[k] = (len > 0.0) ? magnitude * (dir[k] / len) : 0.0;
  for (i = 0; i < nlocal; i++)
    for (k = 0; k < 3; k++) f[3 * i + k] += rmass[i] * acc[k];
}

/* ---- group-aware variants ([3P] LAMMPS 1Feb14: every fix acts on the atoms whose mask has the fix's group bit) ---- */
void orc_nve_sphere_initial_group(int nlocal, double dt, double *x, double *v, double *omega, const double *f,
                                  const double *torque, const double *radius, const double *rmass,
                                  const int *mask, int groupbit)
{
  const double dtv = dt, dtf = 0.5 * dt;
  const double dtfrotate = dtf / 0.4;
  int i, k;
  for (i = 0; i < nlocal; i++) {
    if (!(mask[i] & groupbit)) continue;                            /* fix_nve_sphere.cpp: if (mask[i] & groupbit) */
    double dtfm = dtf / rmass[i];
    for (k = 0; k < 3; k++) v[3 * i + k] += dtfm * f[3 * i + k];
    for (k = 0; k < 3; k++) x[3 * i + k] += dtv * v[3 * i + k];
    double dtirotate = dtfrotate / (radius[i] * radius[i] * rmass[i]);
    for (k = 0; k < 3; k++) omega[3 * i + k] += dtirotate * torque[3 * i + k];
  }
}

void orc_nve_sphere_final_group(int nlocal, double dt, double *v, double *omega, const double *f,
                                const double *torque, const double *radius, const double *rmass,
                                const int *mask, int groupbit)
{
  const double dtf = 0.5 * dt;
  const double dtfrotate = dtf / 0.4;
  int i, k;
  for (i = 0; i < nlocal; i++) {
    if (!(mask[i] & groupbit)) continue;
    double dtfm = dtf / rmass[i];
    for (k = 0; k < 3; k++) v[3 * i + k] += dtfm * f[3 * i + k];
    double dtirotate = dtfrotate / (radius[i] * radius[i] * rmass[i]);
    for (k = 0; k < 3; k++) omega[3 * i + k] += dtirotate * torque[3 * i + k];
  }
}

void orc_fix_gravity_group(int nlocal, double magnitude, const double dir[3], const double *rmass,
                           const int *mask, int groupbit, double *f)
{
  double len = sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
  double acc[3];
  int i, k;
  for (k = 0; k < 3; k++) acc[k] = (len > 0.0) ? magnitude * (dir[k] / len) : 0.0;
  for (i = 0; i < nlocal; i++)
    if (mask[i] & groupbit)
      for (k = 0; k < 3; k++) f[3 * i + k] += rmass[i] * acc[k];
}

/* [3P] FixFreeze::post_force (fix_freeze.cpp): force and torque of the group's atoms are zeroed; the granular pair
 * styles additionally treat a frozen partner as infinitely heavy (pair_gran_hertzFix_history.cpp:188-189) */
void orc_fix_freeze(int nlocal, const int *mask, int groupbit, double *f, double *torque)
{
  int i, k;
  for (i = 0; i < nlocal; i++)
    if (mask[i] & groupbit)
      for (k = 0; k < 3; k++) f[3 * i + k] = torque[3 * i + k] = 0.0;
}
