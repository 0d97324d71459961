/* orc_contact.c -- oracle restatement of the granular contact laws (TEST INFRASTRUCTURE ONLY).
 *
 *   orc_pair_gran_hertzfix_history : interfaceToLammps/pair_gran_hertzFix_history.cpp:45-287
 *   orc_pair_gran_hooke_history    : [3P] LAMMPS 1Feb14 pair_gran_hooke_history.cpp compute();
 *                                    the Hookean law itself is the one the reference carries in
 *                                    interfaceToLammps/fix_wall_granFix.cpp:441-554
 *   orc_pair_gran_hooke            : [3P] LAMMPS 1Feb14 pair_gran_hooke.cpp compute() (absent from the reference
 *                                    tree; restated from the published algorithm).  The law itself is the one the
 *                                    reference carries as FixWallGranFix::hooke, fix_wall_granFix.cpp:347-437
 *   orc_fix_wall_gran              : interfaceToLammps/fix_wall_granFix.cpp:247-345 (plane walls),
 *                                    hooke :347-437, hooke_history :441-554, hertz_history :558-679
 *   orc_gran_settings              : pair_gran_hertzFix_history.cpp:293-317
 *
 * The floating-point expression order of the reference is kept (the repeated sub-expressions are
 * evaluated once: they are pure and give the same double every time).
 */
#include <math.h>
#include <stddef.h>
#include "sedifoam_oracle.h"

#define ORC_PI 3.14159265358979323846 /* MathConst::MY_PI [3P] */

int orc_gran_settings(orc_gran_params *p, double kn, int kt_null, double kt, double gamman,
                      int gammat_null, double gammat, double xmu, int dampflag, double nktv2p)
{
  p->kn = kn;
  p->kt = kt_null ? kn * 2.0 / 7.0 : kt;              /* :298 */
  p->gamman = gamman;
  p->gammat = gammat_null ? 0.5 * gamman : gammat;    /* :302 */
  p->xmu = xmu;
  p->dampflag = dampflag;
  if (dampflag == 0) p->gammat = 0.0;                 /* :307 */
  if (p->kn < 0.0 || p->kt < 0.0 || p->gamman < 0.0 || p->gammat < 0.0 || p->xmu < 0.0 ||
      p->xmu > 10000.0 || dampflag < 0 || dampflag > 1)
    return -1;                                        /* :309-311 */
  p->kn /= nktv2p;                                    /* :315-316 */
  p->kt /= nktv2p;
  return 0;
}

/* beta of :195-196 -- gamman is used as a restitution coefficient there */
static double hertz_beta(double gamman)
{
  double lg = log(gamman) / log(exp(1.0));
  return -(lg) / sqrt(lg * lg + ORC_PI * ORC_PI);
}

/* What one touching contact needs; `del` points from the partner (or wall) to the particle. */
typedef struct {
  double del[3], rsq;
  double vr[3];        /* relative translational velocity */
  double wr[3];        /* (radi*omega_i + radj*omega_j) * rinv, filled by caller */
  double meff;
  double radi, radj;   /* radj unused for walls */
  double overlap;      /* radsum - r (pair) or radius - r (wall) */
  double reff_term;    /* (overlap)*radi*radj/radsum (pair) or (overlap)*radius (wall) */
} contact_in;

typedef struct {
  double F[3];         /* force on the particle */
  double tor[3];       /* rinv * (del x fs): caller scales by -radius */
} contact_out;

/* Hertzian history law: pair :191-261, wall :598-678.  wall_form selects the two spots where
 * the wall twin writes `/ rsq` instead of `* rsqinv` (fix_wall_granFix.cpp:582-584). */
static void hertz_history_law(const orc_gran_params *p, double dt, int shearupdate, int wall_form,
                              const contact_in *c, double *shear, contact_out *o)
{
  const double kn = p->kn, kt = p->kt, xmu = p->xmu;
  double r = sqrt(c->rsq);
  double rinv = 1.0 / r;
  double rsqinv = 1.0 / c->rsq;
  double vnnr = c->vr[0] * c->del[0] + c->vr[1] * c->del[1] + c->vr[2] * c->del[2];
  double vn[3], vt[3], vtr[3], fs[3];
  int k;
  for (k = 0; k < 3; k++) {
    vn[k] = wall_form ? c->del[k] * vnnr / c->rsq : c->del[k] * vnnr * rsqinv;
    vt[k] = c->vr[k] - vn[k];
  }
  double polyhertz = sqrt(c->reff_term);
  double sn = 2.0 * 1.0 / 1.82 * kn * polyhertz;
  double st = 8.0 * 1.0 / 8.84 * kn * polyhertz;
  double beta = hertz_beta(p->gamman);
  double damp = 2.0 * sqrt(5.0 / 6.0) * beta * vnnr * rsqinv;
  double ccel = polyhertz * 4.0 / 5.46 * kn * c->overlap * rinv - sqrt(sn * c->meff) * damp;

  vtr[0] = vt[0] - (c->del[2] * c->wr[1] - c->del[1] * c->wr[2]);
  vtr[1] = vt[1] - (c->del[0] * c->wr[2] - c->del[2] * c->wr[0]);
  vtr[2] = vt[2] - (c->del[1] * c->wr[0] - c->del[0] * c->wr[1]);

  if (shearupdate)
    for (k = 0; k < 3; k++) shear[k] += vtr[k] * dt;
  double shrmag = sqrt(shear[0] * shear[0] + shear[1] * shear[1] + shear[2] * shear[2]);
  double rsht = shear[0] * c->del[0] + shear[1] * c->del[1] + shear[2] * c->del[2];
  rsht *= rsqinv;
  if (shearupdate)
    for (k = 0; k < 3; k++) shear[k] -= rsht * c->del[k];

  double sdamp = sqrt(st * c->meff) * 2.0 * sqrt(5.0 / 6.0) * beta;
  for (k = 0; k < 3; k++) fs[k] = -polyhertz * 8.0 / 8.84 * kt * shear[k] - sdamp * vtr[k];

  double fsmag = sqrt(fs[0] * fs[0] + fs[1] * fs[1] + fs[2] * fs[2]);
  double fn = xmu * fabs(ccel * r);
  if (fsmag > fn) {
    if (shrmag != 0.0) {
      for (k = 0; k < 3; k++) {
        double q = sdamp * vtr[k] / 8.84 * 8.0 / kt;
        shear[k] = (fn / fsmag) * (shear[k] + q) - q;
        fs[k] *= fn / fsmag;
      }
    } else
      fs[0] = fs[1] = fs[2] = 0.0;
  }
  for (k = 0; k < 3; k++) o->F[k] = c->del[k] * ccel + fs[k];
  o->tor[0] = rinv * (c->del[1] * fs[2] - c->del[2] * fs[1]);
  o->tor[1] = rinv * (c->del[2] * fs[0] - c->del[0] * fs[2]);
  o->tor[2] = rinv * (c->del[0] * fs[1] - c->del[1] * fs[0]);
}

/* Hookean history law: fix_wall_granFix.cpp:480-553 (wall, meff = mass) and its pair twin [3P] */
static void hooke_history_law(const orc_gran_params *p, double dt, int shearupdate,
                              const contact_in *c, double *shear, contact_out *o)
{
  const double kn = p->kn, kt = p->kt, xmu = p->xmu;
  const double gamman = p->gamman, gammat = p->gammat;
  double r = sqrt(c->rsq);
  double rinv = 1.0 / r;
  double rsqinv = 1.0 / c->rsq;
  double vnnr = c->vr[0] * c->del[0] + c->vr[1] * c->del[1] + c->vr[2] * c->del[2];
  double vn[3], vt[3], vtr[3], fs[3];
  int k;
  for (k = 0; k < 3; k++) {
    vn[k] = c->del[k] * vnnr * rsqinv;
    vt[k] = c->vr[k] - vn[k];
  }
  double damp = c->meff * gamman * vnnr * rsqinv;
  double ccel = kn * c->overlap * rinv - damp;

  vtr[0] = vt[0] - (c->del[2] * c->wr[1] - c->del[1] * c->wr[2]);
  vtr[1] = vt[1] - (c->del[0] * c->wr[2] - c->del[2] * c->wr[0]);
  vtr[2] = vt[2] - (c->del[1] * c->wr[0] - c->del[0] * c->wr[1]);

  if (shearupdate)
    for (k = 0; k < 3; k++) shear[k] += vtr[k] * dt;
  double shrmag = sqrt(shear[0] * shear[0] + shear[1] * shear[1] + shear[2] * shear[2]);
  double rsht = shear[0] * c->del[0] + shear[1] * c->del[1] + shear[2] * c->del[2];
  rsht = rsht * rsqinv;
  if (shearupdate)
    for (k = 0; k < 3; k++) shear[k] -= rsht * c->del[k];

  for (k = 0; k < 3; k++) fs[k] = -(kt * shear[k] + c->meff * gammat * vtr[k]);

  double fsmag = sqrt(fs[0] * fs[0] + fs[1] * fs[1] + fs[2] * fs[2]);
  double fn = xmu * fabs(ccel * r);
  if (fsmag > fn) {
    if (shrmag != 0.0) {
      for (k = 0; k < 3; k++) {
        double q = c->meff * gammat * vtr[k] / kt;
        shear[k] = (fn / fsmag) * (shear[k] + q) - q;
        fs[k] *= fn / fsmag;
      }
    } else
      fs[0] = fs[1] = fs[2] = 0.0;
  }
  for (k = 0; k < 3; k++) o->F[k] = c->del[k] * ccel + fs[k];
  o->tor[0] = rinv * (c->del[1] * fs[2] - c->del[2] * fs[1]);
  o->tor[1] = rinv * (c->del[2] * fs[0] - c->del[0] * fs[2]);
  o->tor[2] = rinv * (c->del[0] * fs[1] - c->del[1] * fs[0]);
}

/* Plain Hookean law, no shear history: FixWallGranFix::hooke, fix_wall_granFix.cpp:347-437 (wall), and its pair twin
 * [3P] PairGranHooke::compute with radsum / the summed rotation in place of radius / radius*omega. */
static void hooke_law(const orc_gran_params *p, const contact_in *c, contact_out *o)
{
  const double kn = p->kn, xmu = p->xmu, gamman = p->gamman, gammat = p->gammat;
  double r = sqrt(c->rsq);                                            /* :358-360 */
  double rinv = 1.0 / r;
  double rsqinv = 1.0 / c->rsq;
  double vnnr = c->vr[0] * c->del[0] + c->vr[1] * c->del[1] + c->vr[2] * c->del[2];   /* :370 */
  double vn[3], vt[3], vtr[3], fs[3];
  int k;
  for (k = 0; k < 3; k++) {
    vn[k] = c->del[k] * vnnr * rsqinv;                                /* :371-373 */
    vt[k] = c->vr[k] - vn[k];                                         /* :377-379 */
  }
  double damp = c->meff * gamman * vnnr * rsqinv;                     /* :389-391 */
  double ccel = kn * c->overlap * rinv - damp;
  vtr[0] = vt[0] - (c->del[2] * c->wr[1] - c->del[1] * c->wr[2]);     /* :395-399 */
  vtr[1] = vt[1] - (c->del[0] * c->wr[2] - c->del[2] * c->wr[0]);
  vtr[2] = vt[2] - (c->del[1] * c->wr[0] - c->del[0] * c->wr[1]);
  double vrel = vtr[0] * vtr[0] + vtr[1] * vtr[1] + vtr[2] * vtr[2];
  vrel = sqrt(vrel);
  double fn = xmu * fabs(ccel * r);                                   /* :403-406 */
  double fsd = c->meff * gammat * vrel;
  double ft;
  if (vrel != 0.0) ft = (fn < fsd ? fn : fsd) / vrel;
  else ft = 0.0;
  for (k = 0; k < 3; k++) fs[k] = -ft * vtr[k];                       /* :410-412 */
  for (k = 0; k < 3; k++) o->F[k] = c->del[k] * ccel + fs[k];         /* :416-418 */
  o->tor[0] = rinv * (c->del[1] * fs[2] - c->del[2] * fs[1]);         /* :424-426 */
  o->tor[1] = rinv * (c->del[2] * fs[0] - c->del[0] * fs[2]);
  o->tor[2] = rinv * (c->del[0] * fs[1] - c->del[1] * fs[0]);
}

/* the ii/jj double loop shared by the pair styles (pair_gran_hertzFix_history.cpp:109-286); law 0 hooke/history,
 * 1 hertzFix/history, 3 plain hooke (no FixShearHistory there: the list's touch flag is kept for the callers that
 * count contacts, the shear slots stay zero) */
static void pair_gran_loop(int law, const orc_gran_params *p, double dt, int shearupdate,
                           int nlocal, const double *x, const double *v, const double *omega,
                           const double *radius, const double *rmass, const int *mask,
                           int freeze_group_bit, const orc_neighlist *list, double *f,
                           double *torque)
{
  int ii, jj, k;
  for (ii = 0; ii < list->inum; ii++) {
    int i = list->ilist[ii];
    double radi = radius[i];
    for (jj = list->first[ii]; jj < list->first[ii + 1]; jj++) {
      int j = list->jlist[jj] & ORC_NEIGHMASK;                      /* :121-122 */
      contact_in c;
      contact_out o;
      double *shear = &list->shear[3 * (size_t)jj];
      for (k = 0; k < 3; k++) c.del[k] = x[3 * i + k] - x[3 * j + k];
      c.rsq = c.del[0] * c.del[0] + c.del[1] * c.del[1] + c.del[2] * c.del[2];
      double radj = radius[j];
      double radsum = radi + radj;
      if (c.rsq >= radsum * radsum) {                               /* :131-139 */
        list->touch[jj] = 0;
        shear[0] = shear[1] = shear[2] = 0.0;
        continue;
      }
      double r = sqrt(c.rsq);
      double rinv = 1.0 / r;
      for (k = 0; k < 3; k++) {
        c.vr[k] = v[3 * i + k] - v[3 * j + k];                      /* :148-150 */
        c.wr[k] = (radi * omega[3 * i + k] + radj * omega[3 * j + k]) * rinv; /* :167-169 */
      }
      double mi = rmass[i], mj = rmass[j];                          /* :175-177 */
      c.meff = mi * mj / (mi + mj);                                 /* :187 */
      if (mask[i] & freeze_group_bit) c.meff = mj;                  /* :188-189 */
      if (mask[j] & freeze_group_bit) c.meff = mi;
      c.radi = radi;
      c.radj = radj;
      c.overlap = radsum - r;
      c.reff_term = (radsum - r) * radi * radj / radsum;            /* :192-199 */
      list->touch[jj] = 1;                                          /* :212 */
      if (law == 1)
        hertz_history_law(p, dt, shearupdate, 0, &c, shear, &o);
      else if (law == 3)
        hooke_law(p, &c, &o);
      else
        hooke_history_law(p, dt, shearupdate, &c, shear, &o);
      for (k = 0; k < 3; k++) {
        f[3 * i + k] += o.F[k];                                     /* :262-264 */
        torque[3 * i + k] -= radi * o.tor[k];                       /* :269-271 */
      }
      if (j < nlocal) {                                             /* :273-280 */
        for (k = 0; k < 3; k++) {
          f[3 * j + k] -= o.F[k];
          torque[3 * j + k] -= radj * o.tor[k];
        }
      }
    }
  }
}

void orc_pair_gran_hertzfix_history(const orc_gran_params *p, double dt, int shearupdate,
                                    int nlocal, const double *x, const double *v,
                                    const double *omega, const double *radius,
                                    const double *rmass, const int *mask, int freeze_group_bit,
                                    const orc_neighlist *list, double *f, double *torque)
{
  pair_gran_loop(1, p, dt, shearupdate, nlocal, x, v, omega, radius, rmass, mask,
                 freeze_group_bit, list, f, torque);
}

void orc_pair_gran_hooke_history(const orc_gran_params *p, double dt, int shearupdate,
                                 int nlocal, const double *x, const double *v,
                                 const double *omega, const double *radius,
                                 const double *rmass, const int *mask, int freeze_group_bit,
                                 const orc_neighlist *list, double *f, double *torque)
{
  pair_gran_loop(0, p, dt, shearupdate, nlocal, x, v, omega, radius, rmass, mask,
                 freeze_group_bit, list, f, torque);
}

void orc_pair_gran_hooke(const orc_gran_params *p, int nlocal, const double *x, const double *v,
                         const double *omega, const double *radius, const double *rmass, const int *mask,
                         int freeze_group_bit, const orc_neighlist *list, double *f, double *torque)
{
  pair_gran_loop(3, p, 0.0, 0, nlocal, x, v, omega, radius, rmass, mask, freeze_group_bit, list, f, torque);
}

/* FixWallGranFix::post_force, fix_wall_granFix.cpp:247-345, with the moving walls of :255-264 (wiggle: the wall
 * oscillates along `axis`, lo/hi follow when the axis is the wall's normal; shear: the wall slides along `axis`) and the
 * z cylinder of :309-322 (wallstyle 3; with shear about x or y the cylinder ROTATES: vwall = vshear (y, -x, 0)/|xy|).
 * steps = update->ntimestep - time_origin. */
void orc_fix_wall_gran_moving(const orc_gran_params *p, int pairstyle, int wallstyle, double lo, double hi,
                              double cylradius, int wiggle, int wshear, int axis, double amplitude,
                              double period, double vshear, long steps, double dt, int shearupdate,
                              int nlocal, const double *x, const double *v, const double *omega,
                              const double *radius, const double *rmass, const int *mask, int groupbit,
                              double *shear, double *f, double *torque)
{
  int i, k;
  double wlo = lo, whi = hi;                                        /* :254-264 */
  double vwall[3] = {0.0, 0.0, 0.0};
  if (wiggle) {
    double om = 2.0 * 3.14159265358979323846 / period;              /* :165 (MY_PI) */
    double arg = om * (double)steps * dt;
    if (wallstyle == axis) {
      wlo = lo + amplitude - amplitude * cos(arg);
      whi = hi + amplitude - amplitude * cos(arg);
    }
    vwall[axis] = amplitude * om * sin(arg);
  } else if (wshear)
    vwall[axis] = vshear;
  for (i = 0; i < nlocal; i++) {
    if (!(mask[i] & groupbit)) continue;
    contact_in c;
    contact_out o;
    double rad = radius[i];
    c.del[0] = c.del[1] = c.del[2] = 0.0;
    if (wallstyle < 3) {
      double del1 = x[3 * i + wallstyle] - wlo;                     /* :294-308 */
      double del2 = whi - x[3 * i + wallstyle];
      if (del1 < del2) c.del[wallstyle] = del1;
      else c.del[wallstyle] = -del2;
    } else {                                                        /* :309-322 */
      double delxy = sqrt(x[3 * i] * x[3 * i] + x[3 * i + 1] * x[3 * i + 1]);
      double delr = cylradius - delxy;
      if (delr > rad) c.del[2] = cylradius;
      else {
        c.del[0] = -delr / delxy * x[3 * i];
        c.del[1] = -delr / delxy * x[3 * i + 1];
        if (wshear && axis != 2) {
          vwall[0] = vshear * x[3 * i + 1] / delxy;
          vwall[1] = -vshear * x[3 * i] / delxy;
          vwall[2] = 0.0;
        }
      }
    }
    c.rsq = c.del[0] * c.del[0] + c.del[1] * c.del[1] + c.del[2] * c.del[2];
    if (c.rsq > rad * rad) {                                        /* :326-331 (HOOKE keeps no shear array) */
      if (pairstyle != 3) shear[3 * i] = shear[3 * i + 1] = shear[3 * i + 2] = 0.0;
      continue;
    }
    double r = sqrt(c.rsq);
    double rinv = 1.0 / r;
    for (k = 0; k < 3; k++) {
      c.vr[k] = v[3 * i + k] - vwall[k];                            /* :457-459, :575-577 */
      c.wr[k] = rad * omega[3 * i + k] * rinv;                      /* :476-478, :594-596 */
    }
    c.meff = rmass[i];                                              /* :482, :600 */
    c.radi = rad;
    c.radj = 0.0;
    c.overlap = rad - r;
    c.reff_term = (rad - r) * rad;                                  /* :602-608 */
    if (pairstyle == 2)
      hertz_history_law(p, dt, shearupdate, 1, &c, &shear[3 * i], &o);
    else if (pairstyle == 3)
      hooke_law(p, &c, &o);                                         /* :333-335 */
    else
      hooke_history_law(p, dt, shearupdate, &c, &shear[3 * i], &o);
    for (k = 0; k < 3; k++) {
      f[3 * i + k] += o.F[k];
      torque[3 * i + k] -= rad * o.tor[k];
    }
  }
}

void orc_fix_wall_gran(const orc_gran_params *p, int pairstyle, int wallstyle, double lo,
                       double hi, double dt, int shearupdate, int nlocal, const double *x,
                       const double *v, const double *omega, const double *radius,
                       const double *rmass, const int *mask, int groupbit, double *shear,
                       double *f, double *torque)
{
  orc_fix_wall_gran_moving(p, pairstyle, wallstyle, lo, hi, 0.0, 0, 0, 0, 0.0, 1.0, 0.0, 0, dt, shearupdate,
                           nlocal, x, v, omega, radius, rmass, mask, groupbit, shear, f, torque);
}
