/* orc_dem.c -- oracle DEM driver (TEST INFRASTRUCTURE ONLY).
 *
 * Restates what the reference's `lammps_step(ptr, n)` (interfaceToLammps/library.cpp:372-386,
 * "run n pre no post no") does to the particles when the input script is one of the reference's
 * own in.lammps files (e.g. cases/auto-testing/test-cases/xiaocase3/in.lammps): newton off,
 * atom_style sphere, communicate single vel yes, neighbor <skin> bin, neigh_modify delay 0.
 *
 * Everything here except the calls into orc_contact.c / orc_fixes.c is LAMMPS 1Feb14 machinery
 * that is NOT under /root/reference ([3P]): the velocity-Verlet loop (verlet.cpp), the
 * rebuild trigger (Neighbor::decide/check_distance: any atom moved > skin/2), periodic ghost
 * atoms (Comm::borders / forward_comm for one processor), the granular no-newton binned half
 * list (Neighbor::granular_bin_no_newton: criterion rsq <= (ri+rj+skin)^2, pair stored once with
 * j > i, owned-ghost pairs stored on the owned side) and the shear-history carry-over by partner
 * tag (FixShearHistory::pre_exchange + re-injection at list build).  It is restated from the
 * upstream algorithm and anchored by the reference's golden trajectories (see header).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "orc_dem_priv.h"

void *orc__xrealloc(void *p, size_t n)
{
  void *q = realloc(p, n ? n : 1);
  if (!q) {
    fprintf(stderr, "orc_dem: out of memory\n");
    abort();
  }
  return q;
}

void orc__grow_atoms(orc_dem *d, int nmax)
{
  int w;
  if (nmax <= d->nmax) return;
  nmax = nmax + nmax / 4 + 64;
  d->x = orc__xrealloc(d->x, sizeof(double) * 3 * nmax);
  d->v = orc__xrealloc(d->v, sizeof(double) * 3 * nmax);
  d->omega = orc__xrealloc(d->omega, sizeof(double) * 3 * nmax);
  d->f = orc__xrealloc(d->f, sizeof(double) * 3 * nmax);
  d->torque = orc__xrealloc(d->torque, sizeof(double) * 3 * nmax);
  d->radius = orc__xrealloc(d->radius, sizeof(double) * nmax);
  d->rmass = orc__xrealloc(d->rmass, sizeof(double) * nmax);
  d->tag = orc__xrealloc(d->tag, sizeof(int) * nmax);
  d->mask = orc__xrealloc(d->mask, sizeof(int) * nmax);
  d->gsrc = orc__xrealloc(d->gsrc, sizeof(int) * nmax);
  d->gshift = orc__xrealloc(d->gshift, sizeof(double) * 3 * nmax);
  d->binnext = orc__xrealloc(d->binnext, sizeof(int) * nmax);
  for (w = 0; w < d->nfix; w++)
    if (d->fix[w].kind == FIX_WALL) {
      int old = d->nmax, i;
      d->fix[w].wshear = orc__xrealloc(d->fix[w].wshear, sizeof(double) * 3 * nmax);
      for (i = 3 * old; i < 3 * nmax; i++) d->fix[w].wshear[i] = 0.0;
    }
  d->nmax = nmax;
}

orc_dem *orc_dem_create(int n, const double *x, const double *v, const double *omega,
                        const double *radius, const double *rmass, const int *tag,
                        const double boxlo[3], const double boxhi[3], const int periodic[3])
{
  orc_dem *d = calloc(1, sizeof(orc_dem));
  d->nve_bit = 1;
  int i, k;
  orc__grow_atoms(d, n);
  d->nlocal = n;
  for (i = 0; i < n; i++) {
    for (k = 0; k < 3; k++) {
      d->x[3 * i + k] = x[3 * i + k];
      d->v[3 * i + k] = v ? v[3 * i + k] : 0.0;
      d->omega[3 * i + k] = omega ? omega[3 * i + k] : 0.0;
      d->f[3 * i + k] = d->torque[3 * i + k] = 0.0;
    }
    d->radius[i] = radius[i];
    d->rmass[i] = rmass[i];
    d->tag[i] = tag ? tag[i] : i + 1;
    d->mask[i] = 1;
  }
  for (k = 0; k < 3; k++) {
    d->boxlo[k] = boxlo[k];
    d->boxhi[k] = boxhi[k];
    d->periodic[k] = periodic[k];
  }
  d->xhold = calloc(3 * (size_t)(n ? n : 1), sizeof(double));
  d->ffluiddrag = calloc(3 * (size_t)(n ? n : 1), sizeof(double));
  d->DuDt = calloc(3 * (size_t)(n ? n : 1), sizeof(double));
  d->vOld = calloc(3 * (size_t)(n ? n : 1), sizeof(double));
  d->localcap = n ? n : 1;
  d->ilist = malloc(sizeof(int) * (n ? n : 1));
  for (i = 0; i < n; i++) d->ilist[i] = i;
  d->skin = 0.0;
  d->dt = 0.0;
  d->nthreads = 1;
  return d;
}

void orc_dem_destroy(orc_dem *d)
{
  int w;
  if (!d) return;
  for (w = 0; w < d->nfix; w++) free(d->fix[w].wshear);
  free(d->x); free(d->v); free(d->omega); free(d->f); free(d->torque);
  free(d->radius); free(d->rmass); free(d->tag); free(d->mask); free(d->gsrc);
  free(d->gshift); free(d->xhold); free(d->ffluiddrag); free(d->DuDt); free(d->vOld);
  free(d->first); free(d->jlist); free(d->touch); free(d->shear);
  free(d->hfirst); free(d->hjlist); free(d->ffirst); free(d->fjlist);
  free(d->ilist); free(d->binhead); free(d->binnext);
  free(d);
}

int orc_dem_pair_gran(orc_dem *d, int style, double kn, int kt_null, double kt, double gamman,
                      int gammat_null, double gammat, double xmu, int dampflag)
{
  d->pair_style = style;
  return orc_gran_settings(&d->gp, kn, kt_null, kt, gamman, gammat_null, gammat, xmu, dampflag,
                           1.0);
}

void orc_dem_pair_lubricate(orc_dem *d, double mu, int flaglog, int flagfld, double cut_inner,
                            double cut_global, int flagHI, int flagVF)
{
  d->have_lub = 1;
  d->lub.mu = mu;
  d->lub.flaglog = flaglog;
  d->lub.flagfld = flagfld;
  d->lub.cut_inner = cut_inner;
  d->lub.cut_global = cut_global;
  d->lub.flagHI = flagHI;
  d->lub.flagVF = flagVF;
  d->lub.vxmu2f = 1.0;
}

static orc_fix *new_fix(orc_dem *d, int kind)
{
  orc_fix *fx = &d->fix[d->nfix++];
  memset(fx, 0, sizeof(*fx));
  fx->kind = kind;
  fx->groupbit = 1;
  return fx;
}

void orc_dem_fix_cohesive(orc_dem *d, double ah, double lam, double smin, double smax, int opt)
{
  orc_fix *fx = new_fix(d, FIX_COHESIVE);
  fx->ah = ah; fx->lam = lam; fx->smin = smin; fx->smax = smax; fx->opt = opt;
}

void orc_dem_fix_gravity(orc_dem *d, double magnitude, double gx, double gy, double gz)
{
  orc_fix *fx = new_fix(d, FIX_GRAVITY);
  fx->gmag = magnitude;
  fx->gdir[0] = gx; fx->gdir[1] = gy; fx->gdir[2] = gz;
}

/* [3P] fix freeze at ITS PLACE in the fix list: Modify::post_force runs the fixes in script order, so the fixes
 * registered after this call still act on the frozen atoms (cases/example-cases/transport-bedload/in.lammps:28-31:
 * `fix 4 bottom freeze`, then `fix ywall all wall/gran ...`) */
void orc_dem_fix_freeze(orc_dem *d, int groupbit)
{
  orc_fix *fx = new_fix(d, FIX_FREEZE);
  fx->groupbit = groupbit;
  d->freeze_bit = groupbit;
}

void orc_dem_fix_fdrag(orc_dem *d, double carrier_rho)
{
  orc_fix *fx = new_fix(d, FIX_FDRAG);
  fx->carrier_rho = carrier_rho;
}

void orc_dem_fix_wall(orc_dem *d, int wallstyle, int lo_null, double lo, int hi_null, double hi,
                      double kn, int kt_null, double kt, double gamman, int gammat_null,
                      double gammat, double xmu, int dampflag)
{
  orc_fix *fx = new_fix(d, FIX_WALL);
  fx->wallstyle = wallstyle;
  fx->lo = lo_null ? -1.0e20 : lo;   /* fix_wall_granFix.cpp:86-89, BIG = 1e20 */
  fx->hi = hi_null ? 1.0e20 : hi;
  orc_gran_settings(&fx->wp, kn, kt_null, kt, gamman, gammat_null, gammat, xmu, dampflag, 1.0);
  fx->wshear = calloc(3 * (size_t)d->nmax, sizeof(double));
}

/* the last registered wall becomes a z cylinder of the given radius (wallstyle zcylinder, :107-112) */
void orc_dem_wall_cylinder(orc_dem *d, double cylradius)
{
  orc_fix *fx = &d->fix[d->nfix - 1];
  fx->wallstyle = 3;
  fx->cylradius = cylradius;
  fx->lo = fx->hi = 0.0;
}

/* the last registered wall wiggles (kind 1: axis, amplitude, period) or shears (kind 2: axis, vshear), :117-141 */
void orc_dem_wall_motion(orc_dem *d, int kind, int axis, double a, double b)
{
  orc_fix *fx = &d->fix[d->nfix - 1];
  fx->wiggleflag = kind == 1;
  fx->shearflag = kind == 2;
  fx->axis = axis;
  if (kind == 1) {
    fx->amplitude = a;
    fx->period = b;
  } else
    fx->vshear = a;
}

void orc_dem_set_mask(orc_dem *d, const int *mask)
{
  int i;
  for (i = 0; i < d->nlocal; i++) d->mask[i] = mask[i] | 1;
}

void orc_dem_set_groups(orc_dem *d, int nve_bit, int gravity_bit, int fdrag_bit, int wall_bit, int cohesive_bit,
                        int freeze_bit)
{
  int w;
  d->nve_bit = nve_bit;
  if (freeze_bit || !d->freeze_bit) d->freeze_bit = freeze_bit;   /* (0 keeps what orc_dem_fix_freeze registered) */
  for (w = 0; w < d->nfix; w++) {
    orc_fix *fx = &d->fix[w];
    if (fx->kind == FIX_GRAVITY) fx->groupbit = gravity_bit;
    else if (fx->kind == FIX_FDRAG) fx->groupbit = fdrag_bit;
    else if (fx->kind == FIX_WALL) fx->groupbit = wall_bit;
    else if (fx->kind == FIX_COHESIVE) fx->groupbit = cohesive_bit;
  }
}

void orc_dem_neighbor(orc_dem *d, double skin) { d->skin = skin; }
void orc_dem_timestep(orc_dem *d, double dt) { d->dt = dt; }
void orc_dem_threads(orc_dem *d, int nthreads) { d->nthreads = nthreads > 0 ? nthreads : 1; }

/* ---- [3P] neighbor cutoffs: gran pair cutoff = 2*maxrad, lubricate = cut_global ---- */
static double max_radius(const orc_dem *d)
{
  double m = d->rmax_global; /* decomposed twin: the MAX over the ranks ([3P] MPI_Allreduce of maxrad_dynamic); else 0 */
  int i;
  for (i = 0; i < d->nlocal; i++)
    if (d->radius[i] > m) m = d->radius[i];
  return m;
}

double orc__cutneighmax(const orc_dem *d)
{
  double c = 0.0;
  if (d->pair_style) c = 2.0 * max_radius(d);
  if (d->have_lub && d->lub.cut_global > c) c = d->lub.cut_global;
  return c + d->skin;
}

/* ---- [3P] Domain::pbc for owned atoms ---- */
void orc__pbc(orc_dem *d)
{
  int i, k;
  for (k = 0; k < 3; k++) {
    if (!d->periodic[k]) continue;
    if (k == 0 && d->external_x) continue; /* wrapped by the migration shift */
    double lo = d->boxlo[k], hi = d->boxhi[k], prd = hi - lo;
    for (i = 0; i < d->nlocal; i++) {
      double *xi = &d->x[3 * i + k];
      if (*xi < lo) *xi += prd;
      if (*xi >= hi) {
        *xi -= prd;
        if (*xi < lo) *xi = lo;
      }
    }
  }
}

/* ---- [3P] Comm::borders on one processor: periodic images as ghost atoms ---- */
void orc__make_ghosts(orc_dem *d)
{
  double cutghost = orc__cutneighmax(d);
  int dim, dir, p, k;
  d->nghost = d->external_x ? d->next_ghost : 0;
  for (dim = 0; dim < 3; dim++) {
    if (!d->periodic[dim]) continue;
    if (dim == 0 && d->external_x) continue; /* x images come from the neighbour slabs */
    double lo = d->boxlo[dim], hi = d->boxhi[dim], prd = hi - lo;
    int nall0 = d->nlocal + d->nghost;
    for (dir = 0; dir < 2; dir++) {
      for (p = 0; p < nall0; p++) {
        double xp = d->x[3 * p + dim];
        int take = (dir == 0) ? (xp >= lo && xp <= lo + cutghost)
                              : (xp >= hi - cutghost && xp <= hi);
        if (!take) continue;
        int g = d->nlocal + d->nghost;
        orc__grow_atoms(d, g + 1);
        for (k = 0; k < 3; k++) {
          d->gshift[3 * g + k] = 0.0;
          d->x[3 * g + k] = d->x[3 * p + k];
          d->v[3 * g + k] = d->v[3 * p + k];
          d->omega[3 * g + k] = d->omega[3 * p + k];
        }
        d->gshift[3 * g + dim] = (dir == 0) ? prd : -prd;
        d->x[3 * g + dim] += d->gshift[3 * g + dim];
        d->radius[g] = d->radius[p];
        d->rmass[g] = d->rmass[p];
        d->tag[g] = d->tag[p];
        d->mask[g] = d->mask[p];
        d->gsrc[g] = p;
        d->nghost++;
      }
    }
  }
}

/* ---- [3P] Comm::forward_comm: ghosts follow their source (x with shift, v, omega) ---- */
void orc__forward_comm(orc_dem *d)
{
  int g, k;
  int nall = d->nlocal + d->nghost;
  for (g = d->nlocal; g < nall; g++) { /* sources always precede their ghosts */
    int p = d->gsrc[g];
    if (p < 0) continue; /* owned by another slab: refreshed by orc_dem_forward_unpack */
    for (k = 0; k < 3; k++) {
      d->x[3 * g + k] = d->x[3 * p + k] + d->gshift[3 * g + k];
      d->v[3 * g + k] = d->v[3 * p + k];
      d->omega[3 * g + k] = d->omega[3 * p + k];
    }
  }
}

/* ---- binning over the ghost-extended box ---- */
typedef struct {
  double lo[3], inv[3];
  int n[3];
} bingrid;

static void bin_atoms(orc_dem *d, bingrid *g)
{
  double cut = orc__cutneighmax(d);
  int k, i, nall = d->nlocal + d->nghost, nb;
  for (k = 0; k < 3; k++) {
    double lo = d->boxlo[k], hi = d->boxhi[k];
    if (k == 0 && d->external_x) { lo = d->sublo - cut; hi = d->subhi + cut; }
    else if (d->periodic[k]) { lo -= cut; hi += cut; }
    else {
      /* non-periodic: cover whatever the atoms span (walls keep them near the box) */
      for (i = 0; i < nall; i++) {
        if (d->x[3 * i + k] < lo) lo = d->x[3 * i + k];
        if (d->x[3 * i + k] > hi) hi = d->x[3 * i + k];
      }
    }
    double len = hi - lo;
    int n = (int)(len / cut);
    if (n < 1) n = 1;
    if (n > 1024) n = 1024;
    g->lo[k] = lo;
    g->n[k] = n;
    g->inv[k] = (len > 0.0) ? n / len : 0.0;
  }
  nb = g->n[0] * g->n[1] * g->n[2];
  if (nb > d->nbins_alloc) {
    d->binhead = orc__xrealloc(d->binhead, sizeof(int) * nb);
    d->nbins_alloc = nb;
  }
  for (i = 0; i < nb; i++) d->binhead[i] = -1;
  for (i = nall - 1; i >= 0; i--) { /* descending insert -> ascending traversal */
    int c[3];
    for (k = 0; k < 3; k++) {
      c[k] = (int)((d->x[3 * i + k] - g->lo[k]) * g->inv[k]);
      if (c[k] < 0) c[k] = 0;
      if (c[k] >= g->n[k]) c[k] = g->n[k] - 1;
    }
    int b = c[0] + g->n[0] * (c[1] + g->n[1] * c[2]);
    d->binnext[i] = d->binhead[b];
    d->binhead[b] = i;
  }
}

/* partner store built from the old list (FixShearHistory::pre_exchange [3P]) */

void orc__partners_from_list(const orc_dem *d, orc_partners *ps)
{
  int n = d->nlocal, i, jj, k;
  int *cnt = calloc((size_t)n + 1, sizeof(int));
  ps->pfirst = calloc((size_t)n + 1, sizeof(int));
  if (d->first) {
    for (i = 0; i < n; i++)
      for (jj = d->first[i]; jj < d->first[i + 1]; jj++)
        if (d->touch[jj]) {
          int j = d->jlist[jj] & ORC_NEIGHMASK;
          cnt[i]++;
          if (j < n) cnt[j]++;
        }
  }
  for (i = 0; i < n; i++) ps->pfirst[i + 1] = ps->pfirst[i] + cnt[i];
  ps->ptag = malloc(sizeof(int) * (ps->pfirst[n] ? ps->pfirst[n] : 1));
  ps->pshear = malloc(sizeof(double) * 3 * (ps->pfirst[n] ? ps->pfirst[n] : 1));
  for (i = 0; i < n; i++) cnt[i] = 0;
  if (d->first) {
    for (i = 0; i < n; i++)
      for (jj = d->first[i]; jj < d->first[i + 1]; jj++)
        if (d->touch[jj]) {
          int j = d->jlist[jj] & ORC_NEIGHMASK;
          int m = ps->pfirst[i] + cnt[i]++;
          ps->ptag[m] = d->tag[j];
          for (k = 0; k < 3; k++) ps->pshear[3 * m + k] = d->shear[3 * jj + k];
          if (j < n) {
            m = ps->pfirst[j] + cnt[j]++;
            ps->ptag[m] = d->tag[i];
            for (k = 0; k < 3; k++) ps->pshear[3 * m + k] = -d->shear[3 * jj + k];
          }
        }
  }
  free(cnt);
}

static void push_int(int **a, int *cap, int n, int val)
{
  if (n >= *cap) {
    *cap = *cap * 2 + 1024;
    *a = orc__xrealloc(*a, sizeof(int) * (size_t)*cap);
  }
  (*a)[n] = val;
}

/* ---- [3P] Neighbor::build: granular half list (+history), regular half list, full list ---- */
void orc__build_lists(orc_dem *d, orc_partners *pps)
{
  orc_partners ps = *pps;
  bingrid g;
  int n = d->nlocal, i, k, bx, by, bz;
  int need_half = 0, w;
  double cutmax = orc__cutneighmax(d);
  double cutmaxsq = cutmax * cutmax;
  double lubcut = d->have_lub ? d->lub.cut_global + d->skin : 0.0;
  for (w = 0; w < d->nfix; w++)
    if (d->fix[w].kind == FIX_COHESIVE) need_half = 1;

  bin_atoms(d, &g);

  d->first = orc__xrealloc(d->first, sizeof(int) * ((size_t)n + 1));
  d->hfirst = orc__xrealloc(d->hfirst, sizeof(int) * ((size_t)n + 1));
  d->ffirst = orc__xrealloc(d->ffirst, sizeof(int) * ((size_t)n + 1));
  int ng = 0, nh = 0, nf = 0;
  int shearcap = d->listcap;
  for (i = 0; i < n; i++) {
    d->first[i] = ng;
    d->hfirst[i] = nh;
    d->ffirst[i] = nf;
    int c[3];
    for (k = 0; k < 3; k++) {
      c[k] = (int)((d->x[3 * i + k] - g.lo[k]) * g.inv[k]);
      if (c[k] < 0) c[k] = 0;
      if (c[k] >= g.n[k]) c[k] = g.n[k] - 1;
    }
    double radi = d->radius[i];
    for (bz = c[2] - 1; bz <= c[2] + 1; bz++) {
      if (bz < 0 || bz >= g.n[2]) continue;
      for (by = c[1] - 1; by <= c[1] + 1; by++) {
        if (by < 0 || by >= g.n[1]) continue;
        for (bx = c[0] - 1; bx <= c[0] + 1; bx++) {
          if (bx < 0 || bx >= g.n[0]) continue;
          int j;
          for (j = d->binhead[bx + g.n[0] * (by + g.n[1] * bz)]; j >= 0; j = d->binnext[j]) {
            if (j == i) continue;
            double dx = d->x[3 * i] - d->x[3 * j];
            double dy = d->x[3 * i + 1] - d->x[3 * j + 1];
            double dz = d->x[3 * i + 2] - d->x[3 * j + 2];
            double rsq = dx * dx + dy * dy + dz * dz;
            if (d->have_lub && rsq <= lubcut * lubcut) {
              push_int(&d->fjlist, &d->fcap, nf, j);
              nf++;
            }
            if (j <= i) continue; /* half lists: each owned pair once, ghosts on owned side */
            if (need_half && rsq <= cutmaxsq) {
              push_int(&d->hjlist, &d->hcap, nh, j);
              nh++;
            }
            if (d->pair_style) {
              double radsum = radi + d->radius[j];
              double cut = radsum + d->skin;
              if (rsq <= cut * cut) {
                push_int(&d->jlist, &d->listcap, ng, j);
                if (d->listcap != shearcap) {
                  d->touch = orc__xrealloc(d->touch, sizeof(int) * (size_t)d->listcap);
                  d->shear = orc__xrealloc(d->shear, sizeof(double) * 3 * (size_t)d->listcap);
                  shearcap = d->listcap;
                }
                /* history re-injection by partner tag */
                int m, found = -1;
                for (m = ps.pfirst[i]; m < ps.pfirst[i + 1]; m++)
                  if (ps.ptag[m] == d->tag[j]) { found = m; break; }
                if (found >= 0) {
                  d->touch[ng] = 1;
                  for (k = 0; k < 3; k++) d->shear[3 * ng + k] = ps.pshear[3 * found + k];
                } else {
                  d->touch[ng] = 0;
                  for (k = 0; k < 3; k++) d->shear[3 * ng + k] = 0.0;
                }
                ng++;
              }
            }
          }
        }
      }
    }
  }
  d->first[n] = ng;
  d->hfirst[n] = nh;
  d->ffirst[n] = nf;
  for (i = 0; i < 3 * n; i++) d->xhold[i] = d->x[i];
  free(ps.pfirst); free(ps.ptag); free(ps.pshear);
  d->nbuilds++;
}

/* [3P] Neighbor::check_distance */
int orc__check_distance(const orc_dem *d)
{
  double delta = 0.5 * d->skin, deltasq = delta * delta;
  int i;
  for (i = 0; i < d->nlocal; i++) {
    double dx = d->x[3 * i] - d->xhold[3 * i];
    double dy = d->x[3 * i + 1] - d->xhold[3 * i + 1];
    double dz = d->x[3 * i + 2] - d->xhold[3 * i + 2];
    if (dx * dx + dy * dy + dz * dz > deltasq) return 1;
  }
  return 0;
}

void orc__compute_forces(orc_dem *d, int setupflag)
{
  /* update->ntimestep [3P]: incremented at the top of every step, so post_force of step n sees n; the setup
   * evaluation sees the value the run starts from.  Called exactly once per step and once per setup. */
  if (!setupflag) d->ntimestep++;
  else {
    int k;
    for (k = 0; k < d->nfix; k++) d->fix[k].time_origin = d->ntimestep;   /* FixWallGranFix::init, :181 */
  }
  int nall = d->nlocal + d->nghost, i, w, freeze_in_list = 0;
  int shearupdate = setupflag ? 0 : 1; /* pair_gran_hertzFix_history.cpp:65-66 */
  for (i = 0; i < 3 * nall; i++) d->f[i] = d->torque[i] = 0.0;

  orc_neighlist gl;
  gl.inum = d->nlocal; gl.ilist = d->ilist; gl.first = d->first; gl.jlist = d->jlist;
  gl.touch = d->touch; gl.shear = d->shear;
  if (d->pair_style == 2)
    orc_pair_gran_hertzfix_history(&d->gp, d->dt, shearupdate, d->nlocal, d->x, d->v, d->omega,
                                   d->radius, d->rmass, d->mask, d->freeze_bit, &gl, d->f, d->torque);
  else if (d->pair_style == 1)
    orc_pair_gran_hooke_history(&d->gp, d->dt, shearupdate, d->nlocal, d->x, d->v, d->omega,
                                d->radius, d->rmass, d->mask, d->freeze_bit, &gl, d->f, d->torque);
  else if (d->pair_style == 3)
    orc_pair_gran_hooke(&d->gp, d->nlocal, d->x, d->v, d->omega, d->radius, d->rmass, d->mask, d->freeze_bit, &gl,
                        d->f, d->torque);
  if (d->have_lub) {
    orc_neighlist fl;
    fl.inum = d->nlocal; fl.ilist = d->ilist; fl.first = d->ffirst; fl.jlist = d->fjlist;
    fl.touch = NULL; fl.shear = NULL;
    orc_pair_lubricate_poly(&d->lub, d->nlocal, d->x, d->v, d->omega, d->radius, &fl, d->f,
                            d->torque);
  }
  for (w = 0; w < d->nfix; w++) {
    orc_fix *fx = &d->fix[w];
    switch (fx->kind) {
      case FIX_GRAVITY:
        orc_fix_gravity_group(d->nlocal, fx->gmag, fx->gdir, d->rmass, d->mask, fx->groupbit, d->f);
        break;
      case FIX_FDRAG:
        orc_fix_fluid_drag(d->nlocal, d->dt, fx->carrier_rho, d->v, d->rmass, d->radius, d->mask,
                           fx->groupbit, d->ffluiddrag, d->DuDt, d->vOld, d->f);
        break;
      case FIX_WALL:
        /* wall/granFix follows the pair style (fix_wall_granFix.cpp:217-229) */
        orc_fix_wall_gran_moving(&fx->wp, d->pair_style == 2 ? 2 : (d->pair_style == 3 ? 3 : 1), fx->wallstyle, fx->lo, fx->hi,
                                 fx->cylradius, fx->wiggleflag, fx->shearflag, fx->axis, fx->amplitude,
                                 fx->period > 0.0 ? fx->period : 1.0, fx->vshear,
                                 d->ntimestep - fx->time_origin, d->dt, shearupdate, d->nlocal, d->x, d->v,
                                 d->omega, d->radius, d->rmass, d->mask, fx->groupbit, fx->wshear, d->f,
                                 d->torque);
        break;
      case FIX_COHESIVE:
        /* FixCohe::setup() has the wrong signature (fix_cohesive.cpp:117) so the fix is
         * never applied during setup */
        if (!setupflag) {
          orc_neighlist hl;
          hl.inum = d->nlocal; hl.ilist = d->ilist; hl.first = d->hfirst; hl.jlist = d->hjlist;
          hl.touch = NULL; hl.shear = NULL;
          orc_fix_cohesive(fx->ah, fx->lam, fx->smin, fx->smax, fx->opt, d->nlocal, 0, d->x,
                           d->radius, d->mask, fx->groupbit, &hl, d->f);
        }
        break;
      case FIX_FREEZE:
        orc_fix_freeze(d->nlocal, d->mask, fx->groupbit, d->f, d->torque);
        freeze_in_list = 1;
        break;
    }
  }
  /* a freeze group given through orc_dem_set_groups only (no place in the list): applied after every other fix */
  if (d->freeze_bit && !freeze_in_list) orc_fix_freeze(d->nlocal, d->mask, d->freeze_bit, d->f, d->torque);
}

void orc_dem_setup(orc_dem *d)
{
  if (d->have_lub) {
    double vol_T = (d->boxhi[0] - d->boxlo[0]) * (d->boxhi[1] - d->boxlo[1]) *
                   (d->boxhi[2] - d->boxlo[2]);
    orc_lubricate_init(&d->lub, d->nlocal, d->radius, vol_T);
  }
  orc_partners ps;
  orc__partners_from_list(d, &ps); /* empty: no list yet */
  orc__pbc(d);
  orc__make_ghosts(d);
  orc__build_lists(d, &ps);
  orc__compute_forces(d, 1);
  d->setup_done = 1;
}

void orc_dem_run(orc_dem *d, int nsteps)
{
  int s;
  if (!d->setup_done) orc_dem_setup(d);
  for (s = 0; s < nsteps; s++) {
    orc_nve_sphere_initial_group(d->nlocal, d->dt, d->x, d->v, d->omega, d->f, d->torque, d->radius,
                                 d->rmass, d->mask, d->nve_bit);
    if (orc__check_distance(d)) {
      /* Verlet::run order [3P]: pre_exchange (FixShearHistory copies the history out of the OLD
       * list, whose ghost indices are still valid) -> pbc -> borders -> build */
      orc_partners ps;
      orc__partners_from_list(d, &ps);
      orc__pbc(d);
      orc__make_ghosts(d);
      orc__build_lists(d, &ps);
    } else
      orc__forward_comm(d);
    orc__compute_forces(d, 0);
    orc_nve_sphere_final_group(d->nlocal, d->dt, d->v, d->omega, d->f, d->torque, d->radius, d->rmass, d->mask,
                               d->nve_bit);
  }
}

int orc_dem_nlocal(const orc_dem *d) { return d->nlocal; }
int orc_dem_nghost(const orc_dem *d) { return d->nghost; }
int orc_dem_nbuilds(const orc_dem *d) { return d->nbuilds; }
long orc_dem_npairs(const orc_dem *d) { return d->first ? d->first[d->nlocal] : 0; }

void orc_dem_get(const orc_dem *d, double *x, double *v, double *omega, double *f,
                 double *torque, int *tag)
{
  size_t nb = sizeof(double) * 3 * (size_t)d->nlocal;
  if (x) memcpy(x, d->x, nb);
  if (v) memcpy(v, d->v, nb);
  if (omega) memcpy(omega, d->omega, nb);
  if (f) memcpy(f, d->f, nb);
  if (torque) memcpy(torque, d->torque, nb);
  if (tag) memcpy(tag, d->tag, sizeof(int) * (size_t)d->nlocal);
}

static int cmp_pair(const void *a, const void *b)
{
  const int *p = a, *q = b;
  return (p[0] > q[0]) - (p[0] < q[0]);
}

/* library.cpp:344-366: sort both tag lists, pair them up rank by rank */
void orc_dem_put_fdrag(orc_dem *d, int n, const double *fdrag, const int *tag)
{
  int nl = d->nlocal, i, k;
  int *a = malloc(sizeof(int) * 2 * (size_t)(nl ? nl : 1));
  int *b = malloc(sizeof(int) * 2 * (size_t)(nl ? nl : 1));
  (void)n;
  for (i = 0; i < nl; i++) {
    a[2 * i] = d->tag[i]; a[2 * i + 1] = i;
    b[2 * i] = tag[i];    b[2 * i + 1] = i;
  }
  qsort(a, nl, 2 * sizeof(int), cmp_pair);
  qsort(b, nl, 2 * sizeof(int), cmp_pair);
  for (i = 0; i < nl; i++) {
    int to = a[2 * i + 1], from = b[2 * i + 1];
    for (k = 0; k < 3; k++) d->ffluiddrag[3 * to + k] = fdrag[3 * from + k];
  }
  free(a); free(b);
}

int orc_dem_get_history(const orc_dem *d, int max, int *tag_i, int *tag_j, double *shear)
{
  int i, jj, k, n = 0;
  if (!d->first) return 0;
  for (i = 0; i < d->nlocal; i++)
    for (jj = d->first[i]; jj < d->first[i + 1]; jj++)
      if (d->touch[jj]) {
        if (n < max) {
          tag_i[n] = d->tag[i];
          tag_j[n] = d->tag[d->jlist[jj] & ORC_NEIGHMASK];
          for (k = 0; k < 3; k++) shear[3 * n + k] = d->shear[3 * jj + k];
        }
        n++;
      }
  return n;
}

void orc_dem_get_wall_shear(const orc_dem *d, int w, double *shear)
{
  int idx = 0, k;
  for (k = 0; k < d->nfix; k++)
    if (d->fix[k].kind == FIX_WALL) {
      if (idx == w) {
        memcpy(shear, d->fix[k].wshear, sizeof(double) * 3 * (size_t)d->nlocal);
        return;
      }
      idx++;
    }
}
