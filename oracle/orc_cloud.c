/* orc_cloud.c -- oracle restatement of the OpenFOAM-side particle path (TEST INFRASTRUCTURE ONLY).
 *
 *   orc_ergun_wenyu_jd       : lammpsFoam/dragModels/ErgunWenYu/ErgunWenYu.C:86-145
 *   orc_syamlal_obrien_jd    : lammpsFoam/dragModels/SyamlalOBrien/SyamlalOBrien.C:85-144
 *   orc_drag_on_particles    : lammpsFoam/enhancedCloud.C:56-76 (alpha), :83-109 (Ur), :112-257 (forces)
 *   orc_particle_to_eulerian : lammpsFoam/enhancedCloud.C:911-980
 *   orc_calc_tc_fields       : lammpsFoam/enhancedCloud.C:316-441
 *   orc_adjust_timestep      : lammpsFoam/softParticleCloud.C:209-261
 *   orc_cell_owner           : the cell that OpenFOAM's tracking (softParticle.C:102-151, [3P]
 *                              particle::trackToFace) ends in, for ONE uniform blockMesh hex block
 *                              whose cells are numbered ix + nx*(iy + ny*iz) [3P blockMesh ordering]
 *
 * OpenFOAM constants [3P]: ROOTVSMALL = 1e-150 (double precision), pi = constant::mathematical::pi.
 * Fields are plain arrays: scalars [n], vectors AoS [3*n].
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include "sedifoam_oracle.h"

#define ROOTVSMALL 1.0e-150
#define FOAM_PI 3.14159265358979323846

static double dmax(double a, double b) { return a > b ? a : b; }

void orc_ergun_wenyu_jd(int n, const double *Ur, const double *alpha, const double *pd,
                        double nuf, double rhof, double *Jd)
{
  int i;
  for (i = 0; i < n; i++) {
    double beta = dmax(1.0 - alpha[i], ROOTVSMALL);                 /* :104 */
    double bp = pow(beta, -2.65);                                   /* :105 */
    double Re = dmax(beta * Ur[i] * pd[i] / nuf, ROOTVSMALL);       /* :106 */
    double Cds = 24.0 * (1.0 + 0.15 * pow(Re, 0.687)) / Re;         /* :107 */
    if (Re > 1000.0) Cds = 0.44;                                    /* :111-114 */
    double K = 0.75 * Cds * rhof * Ur[i] * bp / pd[i];              /* :118 Wen-Yu */
    if (beta <= 0.8) {                                              /* :124-131 Ergun */
      double bd = beta * pd[i];
      K = 150.0 * alpha[i] * nuf * rhof / (bd * bd) + 1.75 * rhof * Ur[i] / (beta * pd[i]);
    }
    Jd[i] = K;
  }
}

void orc_syamlal_obrien_jd(int n, const double *Ur, const double *alpha, const double *pd,
                           double nuf, double rhof, double *Jd)
{
  int i;
  for (i = 0; i < n; i++) {
    double beta = dmax(1.0 - alpha[i], ROOTVSMALL);                 /* :105 */
    double Ai = pow(beta, 4.14);                                    /* :106 */
    double Bi = 0.8 * pow(beta, 1.28);                              /* :107 */
    if (beta > 0.85) Bi = pow(beta, 2.65);                          /* :111-114 */
    double Re = dmax(Ur[i] * pd[i] / nuf, ROOTVSMALL);              /* :117 */
    double a = 0.06 * Re;
    double Vr = 0.5 * (Ai - 0.06 * Re + sqrt(a * a + 0.12 * Re * (2.0 * Bi - Ai) + Ai * Ai)); /* :119-124 */
    double s = 0.63 + 4.8 * sqrt(Vr / Re);
    double Cds = s * s;                                             /* :126 */
    Jd[i] = 0.75 * Cds * rhof * Ur[i] / (pd[i] * (Vr * Vr));        /* :143 */
  }
}

/* NoCorrection::Jd  lammpsFoam/dragModels/NoCorrection/NoCorrection.C:85-146: the Syamlal-O'Brien velocity-voidage
 * correlation with beta >= 1e-6, Re >= 1e-3 and the drag coefficient 24/Re + 4/sqrt(Re) + 0.4 */
void orc_no_correction_jd(int n, const double *Ur, const double *alpha, const double *pd,
                          double nuf, double rhof, double *Jd)
{
  int i;
  for (i = 0; i < n; i++) {
    double beta = dmax(1.0 - alpha[i], 1.0e-6);                     /* :104 */
    double Ai = pow(beta, 4.14);                                    /* :105 */
    double Bi = 0.8 * pow(beta, 1.28);                              /* :106 */
    if (beta > 0.85) Bi = pow(beta, 2.65);                          /* :108-114 */
    double Re = dmax(Ur[i] * pd[i] / nuf, 1.0e-3);                  /* :116 */
    double a = 0.06 * Re;
    double Vr = 0.5 * (Ai - 0.06 * Re + sqrt(a * a + 0.12 * Re * (2.0 * Bi - Ai) + Ai * Ai)); /* :118-124 */
    double Cds = 24 * 1.0 / Re + 4.0 * pow(Re, -0.5) + 0.4;         /* :126 */
    Jd[i] = 0.75 * Cds * rhof * Ur[i] / (pd[i] * (Vr * Vr));        /* :144 */
  }
}

void orc_cell_owner(int n, const double *x, const double origin[3], const double dx[3],
                    const int ncell[3], int *cell)
{
  int i, k;
  for (i = 0; i < n; i++) {
    int c[3], inside = 1;
    for (k = 0; k < 3; k++) {
      double s = (x[3 * i + k] - origin[k]) / dx[k];
      double fl = floor(s);
      if (fl < 0.0 || fl >= (double)ncell[k]) inside = 0;
      c[k] = (int)fl;
    }
    cell[i] = inside ? c[0] + ncell[0] * (c[1] + ncell[1] * c[2]) : -1;
  }
}

/* the same on a graded (blockMesh simpleGrading) block: faces[k] = ncell[k]+1 ascending face coordinates, or NULL =
 * uniform along k.  What the tracking returns on such a mesh is the cell whose face interval holds the point. */
void orc_cell_owner_graded(int n, const double *x, const double origin[3], const double dx[3],
                           const int ncell[3], const double *const faces[3], int *cell)
{
  int i, k;
  for (i = 0; i < n; i++) {
    int c[3], inside = 1;
    for (k = 0; k < 3; k++) {
      double xx = x[3 * i + k];
      const double *f = faces[k];
      if (!f) {
        double fl = floor((xx - origin[k]) / dx[k]);
        if (fl < 0.0 || fl >= (double)ncell[k]) inside = 0;
        c[k] = (int)fl;
      } else if (!(xx >= f[0]) || !(xx < f[ncell[k]])) {
        inside = 0;
        c[k] = 0;
      } else {
        int lo = 0, hi = ncell[k];
        while (hi - lo > 1) {
          int mid = (lo + hi) >> 1;
          if (xx >= f[mid]) lo = mid;
          else hi = mid;
        }
        c[k] = lo;
      }
    }
    cell[i] = inside ? c[0] + ncell[0] * (c[1] + ncell[1] * c[2]) : -1;
  }
}

/* enhancedCloud::g1n  enhancedCloud.C:1372-1384 */
static double g1n(double n)
{
  if (n < 1) return 0.9279;
  return 0.9279 * (2 * n - 1) / n * pow(n, -n / (2 * n - 1)) + 0.001531;
}

void orc_drag_on_particles(const orc_cloud_flags *fl, int dragModel, int n, const int *cell,
                           const double *pos, const double *d, const double *U,
                           const double *UOld, const double *gamma, const double *UfSmoothed,
                           const double *gradp, const double *DDtUf, const double *curlU,
                           double *Uri, double *magUri, double *Jd, double *pDrag,
                           double *pDuDt)
{
  orc_drag_on_particles_hist(fl, dragModel, n, cell, pos, d, U, UOld, gamma, UfSmoothed, gradp, DDtUf, curlU,
                             -1, NULL, NULL, NULL, Uri, magUri, Jd, pDrag, pDuDt);
}

/* the same with the reduced-order history (Basset) force of :197-233 (particleHistoryForce): timeIndex =
 * runTime().timeIndex(), UfSmoothedOld = UfSmoothed_.oldTime(), sumDeltaFb [3n] and n0 [n] = the per-particle state
 * of softParticle.H:104-107 (both start at zero); timeIndex < 0 switches the term off */
void orc_drag_on_particles_hist(const orc_cloud_flags *fl, int dragModel, int n, const int *cell,
                                const double *pos, const double *d, const double *U,
                                const double *UOld, const double *gamma, const double *UfSmoothed,
                                const double *gradp, const double *DDtUf, const double *curlU,
                                int timeIndex, const double *UfSmoothedOld, double *sumDeltaFb, double *n0,
                                double *Uri, double *magUri, double *Jd, double *pDrag,
                                double *pDuDt)
{
  int i, k;
  /* updateParticleUr :83-109 ; updateParticleAlpha :56-76 (alpha buffered in pDuDt[0..n)) */
  for (i = 0; i < n; i++) {
    int c = cell[i];
    if (c < 0) {
      Uri[3 * i] = Uri[3 * i + 1] = Uri[3 * i + 2] = 0.0;
      magUri[i] = 0.0;
      continue;
    }
    for (k = 0; k < 3; k++) Uri[3 * i + k] = UfSmoothed[3 * c + k] - U[3 * i + k];
    magUri[i] = sqrt(Uri[3 * i] * Uri[3 * i] + Uri[3 * i + 1] * Uri[3 * i + 1] +
                     Uri[3 * i + 2] * Uri[3 * i + 2]);
  }
  /* Jd_ = drag_->Jd(magUri_) :129, with pAlpha_ = gamma[cell] and pDia_ = d */
  for (i = 0; i < n; i++) {
    double a = (cell[i] >= 0) ? gamma[cell[i]] : 0.0;
    if (dragModel == 0) orc_ergun_wenyu_jd(1, &magUri[i], &a, &d[i], fl->nub, fl->rhob, &Jd[i]);
    else if (dragModel == 2) orc_no_correction_jd(1, &magUri[i], &a, &d[i], fl->nub, fl->rhob, &Jd[i]);
    else orc_syamlal_obrien_jd(1, &magUri[i], &a, &d[i], fl->nub, fl->rhob, &Jd[i]);
  }
  for (i = 0; i < n; i++) {
    int c = cell[i];
    double F[3] = {0.0, 0.0, 0.0};
    for (k = 0; k < 3; k++) pDrag[3 * i + k] = pDuDt[3 * i + k] = 0.0; /* :125-126 */
    if (c < 0) continue;                                            /* :141 */
    double Vol = FOAM_PI * d[i] * d[i] * d[i] / 6.0;                /* softParticle.H:270-273 */
    double alpha = gamma[c];
    for (k = 0; k < 3; k++) pDuDt[3 * i + k] = DDtUf[3 * c + k];    /* :155 */
    if (fl->particleDrag)                                           /* :157-162 */
      for (k = 0; k < 3; k++) F[k] += Jd[i] * (1.0 - alpha) * Vol * Uri[3 * i + k];
    if (fl->particlePressureGrad)                                   /* :163-168 */
      for (k = 0; k < 3; k++) F[k] += -gradp[3 * c + k] * Vol;
    if (fl->particleBuoyancy)                                       /* :169-173 */
      for (k = 0; k < 3; k++) F[k] += -fl->gravity[k] * fl->rhob * Vol;
    if (fl->particleAddedMass) {                                    /* :175-188 */
      double acc[3], m = 0.0;
      for (k = 0; k < 3; k++) {
        double dupdt = (U[3 * i + k] - UOld[3 * i + k]) / fl->deltaT;
        acc[k] = DDtUf[3 * c + k] - dupdt;
        m += acc[k] * acc[k];
      }
      m = sqrt(m);
      if (m > 10)
        for (k = 0; k < 3; k++) acc[k] = acc[k] / (m + ROOTVSMALL) * 10;
      for (k = 0; k < 3; k++) F[k] += 0.5 * fl->rhob * Vol * acc[k];
    }
    if (fl->particleLift) {                                         /* :189-196 */
      const double *w = &curlU[3 * c];
      const double *u = &Uri[3 * i];
      double cr[3] = {u[1] * w[2] - u[2] * w[1], u[2] * w[0] - u[0] * w[2],
                      u[0] * w[1] - u[1] * w[0]};
      double magw = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
      for (k = 0; k < 3; k++)
        F[k] += 1.6 * fl->rhob * sqrt(fl->nub) * (d[i] * d[i]) * cr[k] / sqrt(magw + ROOTVSMALL);
    }
    if (timeIndex >= 0) {                                           /* :197-234 Elghannay & Tafti 2016 */
      double tau_d = d[i] * d[i] / fl->nub;
      double uri[3], uriOld[3], m1 = 0.0, m2 = 0.0;
      for (k = 0; k < 3; k++) {
        uri[k] = UfSmoothed[3 * c + k] - U[3 * i + k];
        uriOld[k] = UfSmoothedOld[3 * c + k] - UOld[3 * i + k];
        m1 += uri[k] * uri[k];
        m2 += uriOld[k] * uriOld[k];
      }
      double ReP = sqrt(m1) * d[i] / fl->nub, RePOld = sqrt(m2) * d[i] / fl->nub;
      double a1 = 0.632 / (ReP + ROOTVSMALL) + 0.087, a2 = 0.632 / (RePOld + ROOTVSMALL) + 0.087;
      double tau_h = tau_d * (a1 * a1), tau_h_old = tau_d * (a2 * a2);
      double Cb = -1.5 * (d[i] * d[i]) * fl->rhob * pow(3.1416 * fl->nub, 0.5);
      double deltaT = fl->deltaT;
      double tau_t = deltaT * (timeIndex - n0[i]);
      double delta_fb[3], FH[3];
      for (k = 0; k < 3; k++) delta_fb[k] = Cb * ((U[3 * i + k] - UOld[3 * i + k]) / deltaT) / sqrt(deltaT);
      if (tau_t < tau_h) {
        double delta_n_h = timeIndex - n0[i];
        for (k = 0; k < 3; k++) sumDeltaFb[3 * i + k] = sumDeltaFb[3 * i + k] + delta_fb[k];
        for (k = 0; k < 3; k++) FH[k] = g1n(delta_n_h) * sumDeltaFb[3 * i + k];
      } else {
        double delta_n_h = tau_h / deltaT;
        for (k = 0; k < 3; k++) sumDeltaFb[3 * i + k] = tau_h / tau_h_old * sumDeltaFb[3 * i + k];
        for (k = 0; k < 3; k++) sumDeltaFb[3 * i + k] = (delta_n_h - 1) / delta_n_h * sumDeltaFb[3 * i + k];
        n0[i] = timeIndex - delta_n_h;
        for (k = 0; k < 3; k++) sumDeltaFb[3 * i + k] = sumDeltaFb[3 * i + k] + delta_fb[k];
        for (k = 0; k < 3; k++) FH[k] = g1n(delta_n_h) * sumDeltaFb[3 * i + k];
      }
      for (k = 0; k < 3; k++) F[k] += FH[k] * deltaT;
    }
    if (fl->lubricationForce) {                                     /* :235-248 (y wall at 0) */
      double distMin = 0.0001 * d[i], distMax = 0.1 * d[i];
      double distWall = pos[3 * i + 1] - 0.5 * d[i];
      double pVel = U[3 * i + 1];
      if (distWall < distMax && distWall > distMin)
        F[1] += 6 * 3.1416 * fl->nub * fl->rhob * (-pVel) / distWall * (d[i] * d[i]) / 4.0 * 1.0;
    }
    for (k = 0; k < 3; k++) pDrag[3 * i + k] = F[k];
  }
}

/* softParticleCloud::pointInRegion  softParticleCloud.C:1354-1417 (box = the tensor's components 0..7) */
static int point_in_region(int option, const double *b, const double *ecc, const double *pt)
{
  double x1 = b[0], x2 = b[1], y1 = b[2], y2 = b[3], z1 = b[4], z2 = b[5], r1 = b[6], r2 = b[7];
  if (option == 1) {
    if ((pt[0] - x1) * (pt[0] - x2) < ROOTVSMALL && (pt[1] - y1) * (pt[1] - y2) < ROOTVSMALL &&
        (pt[2] - z1) * (pt[2] - z2) < ROOTVSMALL)
      return 1;
    return 0;
  } else if (option == 2) {
    double p2p1[3] = {x2 - x1, y2 - y1, z2 - z1};
    double h = sqrt(p2p1[0] * p2p1[0] + p2p1[1] * p2p1[1] + p2p1[2] * p2p1[2]);
    double pxp1[3] = {pt[0] - x1, pt[1] - y1, pt[2] - z1};
    double dot = p2p1[0] * pxp1[0] + p2p1[1] * pxp1[1] + p2p1[2] * pxp1[2];
    double pxp1E[3] = {pxp1[0] - ecc[0], pxp1[1] - ecc[1], pxp1[2] - ecc[2]};
    if (dot < 0.0 || dot > pow(h, 2)) return 0;
    {
      double dsq = (pxp1[0] * pxp1[0] + pxp1[1] * pxp1[1] + pxp1[2] * pxp1[2]) - dot * dot / pow(h, 2);
      double dsqE = (pxp1E[0] * pxp1E[0] + pxp1E[1] * pxp1E[1] + pxp1E[2] * pxp1E[2]) - dot * dot / pow(h, 2);
      if (dsqE > r1 * r1 && dsq < r2 * r2) return 1;
      return 0;
    }
  }
  return 0;
}

/* the inlet override at the end of the particle loop of updateDragOnParticles, enhancedCloud.C:249-257:
 * inletForceRatio_ is zero unless addParticleOption_ > 0 (:600-608); mass = softParticle::m() */
void orc_inlet_force_override(int addParticleOption, const double inletForce[3], const double inletBox[9],
                              const double eccentricity[3], double deltaT, int n, const double *pos,
                              const double *mass, const double *U, double *pDrag)
{
  int i, k;
  double magF = sqrt(inletForce[0] * inletForce[0] + inletForce[1] * inletForce[1] + inletForce[2] * inletForce[2]);
  if (addParticleOption <= 0 || !(magF > 0)) return;
  for (i = 0; i < n; i++)
    if (point_in_region(addParticleOption, inletBox, eccentricity, pos + 3 * i))
      for (k = 0; k < 3; k++) pDrag[3 * i + k] = mass[i] * (inletForce[k] - U[3 * i + k]) / deltaT;
}

void orc_particle_to_eulerian(int n, const int *cell, const double *d, const double *U,
                              int ncells, const double *V, double *gamma, double *Ue)
{
  int i, c, k;
  for (c = 0; c < ncells; c++) {
    gamma[c] = 0.0;
    Ue[3 * c] = Ue[3 * c + 1] = Ue[3 * c + 2] = 0.0;                /* :914-915 */
  }
  for (i = 0; i < n; i++) {                                         /* :918-928 */
    double Vol = FOAM_PI * d[i] * d[i] * d[i] / 6.0;
    c = cell[i];
    if (c < 0) continue; /* such particles were already dropped from the cloud (softParticle.C:177-184) */
    gamma[c] += Vol;
    for (k = 0; k < 3; k++) Ue[3 * c + k] += Vol * U[3 * i + k];
  }
  for (c = 0; c < ncells; c++) {
    gamma[c] /= V[c];                                               /* :930 */
    for (k = 0; k < 3; k++) Ue[3 * c + k] /= V[c];                  /* :941 */
    if (gamma[c] > ROOTVSMALL)                                      /* :955-962 */
      for (k = 0; k < 3; k++) Ue[3 * c + k] /= gamma[c];
  }
}

void orc_calc_tc_fields(int n, const int *cell, const double *d, const double *U,
                        const double *Jd, int ncells, const double *V, const double *gamma,
                        const double *UfSmoothed, double *Asrc, double *Omega)
{
  int i, c, k;
  for (c = 0; c < ncells; c++) {
    Omega[c] = 0.0;
    Asrc[3 * c] = Asrc[3 * c + 1] = Asrc[3 * c + 2] = 0.0;          /* :318-320 */
  }
  for (i = 0; i < n; i++) {                                         /* :364-389 */
    c = cell[i];
    if (c < 0) continue;
    double Vol = FOAM_PI * d[i] * d[i] * d[i] / 6.0;
    double omg = Vol * Jd[i] / V[c];
    Omega[c] += omg;
    for (k = 0; k < 3; k++) Asrc[3 * c + k] += omg * (U[3 * i + k] - UfSmoothed[3 * c + k]);
  }
  for (c = 0; c < ncells; c++) {
    Omega[c] *= 0;                                                  /* :391 */
    for (k = 0; k < 3; k++) {
      Asrc[3 * c + k] = Asrc[3 * c + k] * (1 - gamma[c]);           /* :407-408 */
      Asrc[3 * c + k] /= (1 - gamma[c]);                            /* :415-416 */
    }
  }
}

int orc_adjust_timestep(double deltaT, double dtLampIn, int subCycles_in, double *dtLampAdj,
                        int *solidStepsPerDt, int *subCycles, int *subSteps)
{
  double dnSub = round(deltaT / dtLampIn);                          /* :216 */
  if (dnSub == 0) dnSub++;                                          /* :217 */
  int sc = subCycles_in;
  int steps = ((int)dnSub / sc) * sc;                               /* :219-221 */
  *dtLampAdj = deltaT / dnSub;                                      /* :224 */
  if (sc >= steps) {                                                /* :229-234 */
    sc = steps;
    *subSteps = 1;
  } else {
    *subSteps = steps / sc;                                         /* :237-238 */
    if (steps % sc != 0) return -1;
  }
  *solidStepsPerDt = steps;
  *subCycles = sc;
  return 0;
}

/* ---- N1: enhancedCloud::smoothField  lammpsFoam/enhancedCloud.C:790-907 (setup :564-583) --------------------
 * `diffusionSteps` implicit-Euler steps of d(phi)/dt = div(D grad phi) up to tau = b^2/4, zero-gradient
 * boundaries, on one uniform hex block: each step solves (I - dtau L) phi_new = phi_old with the 7-point
 * Laplacian ([3P]: what fvm::ddt - fvm::laplacian(DT, .) with "Gauss linear corrected" assembles on an orthogonal
 * uniform mesh).  The reference uses PCG/DIC to 1e-10; here plain CG to 1e-15 (a different solver than the
 * product's, same linear system). */
typedef struct {
  int n[3];
  double c[3];
  int per[3];   /* cyclic patch pair along this axis ([3P] cyclicFvPatch: the first and the last cell are neighbours) */
} orc_stencil;

static void apply_A(const orc_stencil *st, const double *v, double *out)
{
  int i, j, k;
  const int sx = 1, sy = st->n[0], sz = st->n[0] * st->n[1];
  for (k = 0; k < st->n[2]; k++)
    for (j = 0; j < st->n[1]; j++)
      for (i = 0; i < st->n[0]; i++) {
        int c = i + st->n[0] * (j + st->n[1] * k);
        double vc = v[c], acc = vc;
        if (i > 0) acc += st->c[0] * (vc - v[c - sx]);
        else if (st->per[0]) acc += st->c[0] * (vc - v[c + (st->n[0] - 1) * sx]);
        if (i < st->n[0] - 1) acc += st->c[0] * (vc - v[c + sx]);
        else if (st->per[0]) acc += st->c[0] * (vc - v[c - (st->n[0] - 1) * sx]);
        if (j > 0) acc += st->c[1] * (vc - v[c - sy]);
        else if (st->per[1]) acc += st->c[1] * (vc - v[c + (st->n[1] - 1) * sy]);
        if (j < st->n[1] - 1) acc += st->c[1] * (vc - v[c + sy]);
        else if (st->per[1]) acc += st->c[1] * (vc - v[c - (st->n[1] - 1) * sy]);
        if (k > 0) acc += st->c[2] * (vc - v[c - sz]);
        else if (st->per[2]) acc += st->c[2] * (vc - v[c + (st->n[2] - 1) * sz]);
        if (k < st->n[2] - 1) acc += st->c[2] * (vc - v[c + sz]);
        else if (st->per[2]) acc += st->c[2] * (vc - v[c - (st->n[2] - 1) * sz]);
        out[c] = acc;
      }
}

void orc_smooth_field(const int n[3], const double dx[3], const double D[3], double band, int steps, int ncomp,
                      double *field)
{
  orc_smooth_field_periodic(n, dx, D, band, steps, ncomp, field, NULL);
}

/* the same with cyclic patch pairs: the diffusion mesh of the reference's channel cases (blockMeshDict `cyclic`
 * patches, in.lammps `boundary pp ff pp`) couples the first and the last cell of a periodic axis instead of closing
 * them with zeroGradient */
void orc_smooth_field_periodic(const int n[3], const double dx[3], const double D[3], double band, int steps,
                               int ncomp, double *field, const int *periodic)
{
  int nc = n[0] * n[1] * n[2], s, comp, c, it;
  orc_stencil st;
  if (!(band > 0.0) || steps <= 0) return;
  double dtau = (band * band / 4.0) / (steps + 1.0e-150);         /* :564-565 */
  for (c = 0; c < 3; c++) {
    st.n[c] = n[c];
    st.c[c] = dtau * D[c] / (dx[c] * dx[c]);
    st.per[c] = (periodic && periodic[c] && n[c] > 1) ? 1 : 0;
  }
  double *x = malloc(sizeof(double) * nc), *r = malloc(sizeof(double) * nc), *p = malloc(sizeof(double) * nc),
         *ap = malloc(sizeof(double) * nc);
  for (s = 0; s < steps; s++)
    for (comp = 0; comp < ncomp; comp++) {
      double rr = 0.0, bb = 0.0;
      for (c = 0; c < nc; c++) x[c] = field[(size_t)c * ncomp + comp];
      apply_A(&st, x, ap);
      for (c = 0; c < nc; c++) {
        r[c] = x[c] - ap[c];
        p[c] = r[c];
        rr += r[c] * r[c];
        bb += x[c] * x[c];
      }
      for (it = 0; it < 5000 && rr > 1e-30 * bb; it++) {
        double pap = 0.0, rrn = 0.0;
        apply_A(&st, p, ap);
        for (c = 0; c < nc; c++) pap += p[c] * ap[c];
        double alpha = rr / pap;
        for (c = 0; c < nc; c++) {
          x[c] += alpha * p[c];
          r[c] -= alpha * ap[c];
          rrn += r[c] * r[c];
        }
        double beta = rrn / rr;
        for (c = 0; c < nc; c++) p[c] = r[c] + beta * p[c];
        rr = rrn;
      }
      for (c = 0; c < nc; c++) field[(size_t)c * ncomp + comp] = x[c];
    }
  free(x); free(r); free(p); free(ap);
}

/* The same smoothing on a graded (blockMesh simpleGrading) block, w[k] = cell widths along k or NULL = uniform dx[k].
 * Finite volumes on the orthogonal mesh ([3P] fvm::ddt - fvm::laplacian, Gauss linear corrected):
 *   V_c (phi_new - phi_old)_c = dtau * sum_faces D_k S_f (phi_nb - phi_c)_new / |d_f|
 * S_f the face area, |d_f| the distance of the two cell centres, zero flux through the boundary.  Written as
 * (V + dtau K) phi_new = V phi_old the matrix is symmetric positive definite: plain CG to 1e-15. */
typedef struct {
  int n[3];
  const double *w[3];
  double dx[3], D[3], dtau;
  int per[3];
} orc_gstencil;

static double gwidth(const orc_gstencil *st, int k, int i) { return st->w[k] ? st->w[k][i] : st->dx[k]; }

static void apply_G(const orc_gstencil *st, const double *v, double *out)
{
  int idx[3], k;
  const int stride[3] = {1, st->n[0], st->n[0] * st->n[1]};
  for (idx[2] = 0; idx[2] < st->n[2]; idx[2]++)
    for (idx[1] = 0; idx[1] < st->n[1]; idx[1]++)
      for (idx[0] = 0; idx[0] < st->n[0]; idx[0]++) {
        int c = idx[0] + st->n[0] * (idx[1] + st->n[1] * idx[2]);
        double h[3] = {gwidth(st, 0, idx[0]), gwidth(st, 1, idx[1]), gwidth(st, 2, idx[2])};
        double vc = v[c], acc = h[0] * h[1] * h[2] * vc;
        for (k = 0; k < 3; k++) {
          double area = h[(k + 1) % 3] * h[(k + 2) % 3];
          if (idx[k] > 0) {
            double d = 0.5 * (h[k] + gwidth(st, k, idx[k] - 1));
            acc += st->dtau * st->D[k] * area / d * (vc - v[c - stride[k]]);
          } else if (st->per[k]) {
            double d = 0.5 * (h[k] + gwidth(st, k, st->n[k] - 1));
            acc += st->dtau * st->D[k] * area / d * (vc - v[c + (st->n[k] - 1) * stride[k]]);
          }
          if (idx[k] < st->n[k] - 1) {
            double d = 0.5 * (h[k] + gwidth(st, k, idx[k] + 1));
            acc += st->dtau * st->D[k] * area / d * (vc - v[c + stride[k]]);
          } else if (st->per[k]) {
            double d = 0.5 * (h[k] + gwidth(st, k, 0));
            acc += st->dtau * st->D[k] * area / d * (vc - v[c - (st->n[k] - 1) * stride[k]]);
          }
        }
        out[c] = acc;
      }
}

void orc_smooth_field_graded(const int n[3], const double dx[3], const double *const w[3], const double D[3],
                             double band, int steps, int ncomp, double *field)
{
  orc_smooth_field_graded_periodic(n, dx, w, D, band, steps, ncomp, field, NULL);
}

void orc_smooth_field_graded_periodic(const int n[3], const double dx[3], const double *const w[3],
                                      const double D[3], double band, int steps, int ncomp, double *field,
                                      const int *periodic)
{
  int nc = n[0] * n[1] * n[2], s, comp, c, it, k;
  orc_gstencil st;
  if (!(band > 0.0) || steps <= 0) return;
  st.dtau = (band * band / 4.0) / (steps + 1.0e-150);             /* :564-565 */
  for (k = 0; k < 3; k++) {
    st.per[k] = (periodic && periodic[k] && n[k] > 1) ? 1 : 0;
    st.n[k] = n[k];
    st.w[k] = w ? w[k] : NULL;
    st.dx[k] = dx[k];
    st.D[k] = D[k];
  }
  double *x = malloc(sizeof(double) * nc), *b = malloc(sizeof(double) * nc), *r = malloc(sizeof(double) * nc),
         *p = malloc(sizeof(double) * nc), *ap = malloc(sizeof(double) * nc);
  for (s = 0; s < steps; s++)
    for (comp = 0; comp < ncomp; comp++) {
      double rr = 0.0, bb = 0.0;
      int i0, i1, i2;
      for (i2 = 0; i2 < n[2]; i2++)
        for (i1 = 0; i1 < n[1]; i1++)
          for (i0 = 0; i0 < n[0]; i0++) {
            c = i0 + n[0] * (i1 + n[1] * i2);
            x[c] = field[(size_t)c * ncomp + comp];
            b[c] = gwidth(&st, 0, i0) * gwidth(&st, 1, i1) * gwidth(&st, 2, i2) * x[c];
          }
      apply_G(&st, x, ap);
      for (c = 0; c < nc; c++) {
        r[c] = b[c] - ap[c];
        p[c] = r[c];
        rr += r[c] * r[c];
        bb += b[c] * b[c];
      }
      for (it = 0; it < 20000 && rr > 1e-30 * bb; it++) {
        double pap = 0.0, rrn = 0.0;
        apply_G(&st, p, ap);
        for (c = 0; c < nc; c++) pap += p[c] * ap[c];
        double alpha = rr / pap;
        for (c = 0; c < nc; c++) {
          x[c] += alpha * p[c];
          r[c] -= alpha * ap[c];
          rrn += r[c] * r[c];
        }
        double beta = rrn / rr;
        for (c = 0; c < nc; c++) p[c] = r[c] + beta * p[c];
        rr = rrn;
      }
      for (c = 0; c < nc; c++) field[(size_t)c * ncomp + comp] = x[c];
    }
  free(x); free(b); free(r); free(p); free(ap);
}

/* smoothField as the cloud functions below call it: uniform block or graded block (sm->w) */
static void smooth_any(const orc_smooth *sm, int ncomp, double *field)
{
  if (sm->w[0] || sm->w[1] || sm->w[2])
    orc_smooth_field_graded_periodic(sm->n, sm->dx, sm->w, sm->D, sm->band, sm->steps, ncomp, field, sm->periodic);
  else orc_smooth_field_periodic(sm->n, sm->dx, sm->D, sm->band, sm->steps, ncomp, field, sm->periodic);
}

/* particleToEulerianField with the smoothing branches (:944-962) */
void orc_particle_to_eulerian_smooth(int n, const int *cell, const double *d, const double *U, int ncells,
                                     const double *V, const orc_smooth *sm, double *gamma, double *Ue)
{
  int i, c, k;
  for (c = 0; c < ncells; c++) {
    gamma[c] = 0.0;
    Ue[3 * c] = Ue[3 * c + 1] = Ue[3 * c + 2] = 0.0;
  }
  for (i = 0; i < n; i++) {
    double Vol = FOAM_PI * d[i] * d[i] * d[i] / 6.0;
    c = cell[i];
    if (c < 0) continue;
    gamma[c] += Vol;
    for (k = 0; k < 3; k++) Ue[3 * c + k] += Vol * U[3 * i + k];
  }
  for (c = 0; c < ncells; c++) {
    gamma[c] /= V[c];
    for (k = 0; k < 3; k++) Ue[3 * c + k] /= V[c];
  }
  if (sm && sm->alphaSmooth) smooth_any(sm, 1, gamma);
  if (sm && sm->UpSmooth) smooth_any(sm, 3, Ue);
  for (c = 0; c < ncells; c++)
    if (gamma[c] > ROOTVSMALL)
      for (k = 0; k < 3; k++) Ue[3 * c + k] /= gamma[c];
}

/* UfSmoothed (:675-690) */
void orc_uf_smoothed(int ncells, const double *Uf, const double *gamma, const orc_smooth *sm, double *UfS)
{
  int c, k;
  memcpy(UfS, Uf, sizeof(double) * 3 * (size_t)ncells);
  if (!sm || !sm->UfSmooth || !(sm->band > 0.0) || sm->steps <= 0) return;
  for (c = 0; c < ncells; c++)
    for (k = 0; k < 3; k++) UfS[3 * c + k] *= (1 - gamma[c]);
  smooth_any(sm, 3, UfS);
  for (c = 0; c < ncells; c++)
    for (k = 0; k < 3; k++) UfS[3 * c + k] /= (1 - gamma[c]);
}

/* calcTcFields with the smoothing branch (:407-416) */
void orc_calc_tc_fields_smooth(int n, const int *cell, const double *d, const double *U, const double *Jd,
                               int ncells, const double *V, const double *gamma, const double *UfSmoothed,
                               const orc_smooth *sm, double *Asrc, double *Omega)
{
  int i, c, k;
  for (c = 0; c < ncells; c++) {
    Omega[c] = 0.0;
    Asrc[3 * c] = Asrc[3 * c + 1] = Asrc[3 * c + 2] = 0.0;
  }
  for (i = 0; i < n; i++) {
    c = cell[i];
    if (c < 0) continue;
    double Vol = FOAM_PI * d[i] * d[i] * d[i] / 6.0;
    double omg = Vol * Jd[i] / V[c];
    for (k = 0; k < 3; k++) Asrc[3 * c + k] += omg * (U[3 * i + k] - UfSmoothed[3 * c + k]);
  }
  for (c = 0; c < ncells; c++)
    for (k = 0; k < 3; k++) Asrc[3 * c + k] = Asrc[3 * c + k] * (1 - gamma[c]);
  if (sm && sm->dragSmooth) smooth_any(sm, 3, Asrc);
  for (c = 0; c < ncells; c++)
    for (k = 0; k < 3; k++) Asrc[3 * c + k] /= (1 - gamma[c]);
}
