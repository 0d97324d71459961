/* orc_halo.c -- oracle twin of the slab-decomposition halo (TEST INFRASTRUCTURE ONLY).
 *
 * Lets tests/test_halo_gloo.py run the product's multi-rank driver (sedifoam_amd/halo.py: migrate / border
 * / forward protocol over torch.distributed) on CPUs with the gloo backend: each rank holds an oracle DEM
 * driver for its x-slab and exposes the same pack/unpack calls, with the SAME record layouts, as
 * libsedifoam_amd.so (include/sedifoam_amd.h, csrc/sf_dem_halo.hip).  Restates LAMMPS 1Feb14 Comm
 * exchange/borders/forward_comm for a 1-D decomposition [3P], plus the per-atom data the reference's
 * fixes migrate: fix_fluid_drag.cpp:211-243, fix_wall_granFix.cpp:726-744, FixShearHistory [3P].
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "orc_dem_priv.h"

#define BORDER_DOUBLES 14   /* x r | v m | omega | tag type mask */
#define FORWARD_DOUBLES 9

static int nwalls_of(const orc_dem *d)
{
  int w, n = 0;
  for (w = 0; w < d->nfix; w++)
    if (d->fix[w].kind == FIX_WALL) n++;
  return n;
}

static orc_fix *wall_of(orc_dem *d, int idx)
{
  int w, n = 0;
  for (w = 0; w < d->nfix; w++)
    if (d->fix[w].kind == FIX_WALL) {
      if (n == idx) return &d->fix[w];
      n++;
    }
  return NULL;
}

static void ensure_local_cap(orc_dem *d, int n)
{
  int w, i;
  if (n <= d->localcap) return;
  int nc = n + n / 4 + 64;
  d->xhold = orc__xrealloc(d->xhold, sizeof(double) * 3 * nc);
  d->ffluiddrag = orc__xrealloc(d->ffluiddrag, sizeof(double) * 3 * nc);
  d->DuDt = orc__xrealloc(d->DuDt, sizeof(double) * 3 * nc);
  d->vOld = orc__xrealloc(d->vOld, sizeof(double) * 3 * nc);
  d->ilist = orc__xrealloc(d->ilist, sizeof(int) * nc);
  for (i = 3 * d->localcap; i < 3 * nc; i++) d->ffluiddrag[i] = d->DuDt[i] = d->vOld[i] = d->xhold[i] = 0.0;
  for (i = 0; i < nc; i++) d->ilist[i] = i;
  (void)w;
  if (d->leave) {
    d->leave = orc__xrealloc(d->leave, sizeof(int) * (size_t)nc);
    memset(d->leave, 0, sizeof(int) * (size_t)nc);
  }
  d->localcap = nc;
  orc__grow_atoms(d, nc);
}

void orc_dem_set_subdomain(orc_dem *d, double sublo, double subhi)
{
  d->external_x = 1;
  d->sublo = sublo;
  d->subhi = subhi;
}

int orc_dem_max_partners(const orc_dem *d)
{
  int i, m = 0;
  if (d->have_ptab) {
    for (i = 0; i < d->nlocal; i++)
      if (d->pcnt[i] > m) m = d->pcnt[i];
    return m;
  }
  return 0;
}

/* ---- step phases ---- */
void orc_dem_run_begin(orc_dem *d)
{
  orc_nve_sphere_initial_group(d->nlocal, d->dt, d->x, d->v, d->omega, d->f, d->torque, d->radius, d->rmass, d->mask,
                               d->nve_bit);
  d->flag = orc__check_distance(d);
  orc__forward_comm(d);
}

void orc_dem_substep(orc_dem *d, int last)
{
  orc__compute_forces(d, 0);
  orc_nve_sphere_final_group(d->nlocal, d->dt, d->v, d->omega, d->f, d->torque, d->radius, d->rmass, d->mask,
                             d->nve_bit);
  if (!last) {
    orc_nve_sphere_initial_group(d->nlocal, d->dt, d->x, d->v, d->omega, d->f, d->torque, d->radius, d->rmass, d->mask,
                               d->nve_bit);
    d->flag = orc__check_distance(d);
  }
  orc__forward_comm(d);
}

int orc_dem_need_rebuild(const orc_dem *d) { return d->flag; }

double orc_dem_local_particle_volume(const orc_dem *d) { return orc_particle_volume(d->nlocal, d->radius); }

void orc_dem_set_global_particle_volume(orc_dem *d, double volP)
{
  if (d->have_lub) {
    double vol_T = (d->boxhi[0] - d->boxlo[0]) * (d->boxhi[1] - d->boxlo[1]) * (d->boxhi[2] - d->boxlo[2]);
    orc_lubricate_init_vol(&d->lub, volP, vol_T);
  }
}

double orc_dem_local_max_radius(const orc_dem *d)
{
  double m = 0.0;
  int i;
  for (i = 0; i < d->nlocal; i++)
    if (d->radius[i] > m) m = d->radius[i];
  return m;
}

void orc_dem_set_global_max_radius(orc_dem *d, double rmax) { d->rmax_global = rmax; }

void orc_dem_ext_setup(orc_dem *d)
{
  orc__compute_forces(d, 1);
  d->setup_done = 1;
}

/* ---- rebuild protocol ---- */
static void free_ptab(orc_dem *d)
{
  free(d->pcnt); free(d->ptab); free(d->pshtab);
  d->pcnt = NULL; d->ptab = NULL; d->pshtab = NULL;
  d->have_ptab = 0;
}

/* pre_exchange: history of the OLD list by partner tag, as a fixed-width table so atoms can come and go */
void orc_dem_rebuild_begin(orc_dem *d)
{
  orc_partners ps;
  int n = d->nlocal, i, m, W = 0;
  orc__partners_from_list(d, &ps);
  for (i = 0; i < n; i++)
    if (ps.pfirst[i + 1] - ps.pfirst[i] > W) W = ps.pfirst[i + 1] - ps.pfirst[i];
  free_ptab(d);
  d->mrec = W; /* provisional; orc_dem_migrate_set_slots widens it to the global maximum */
  {
    const size_t rows = n > 0 ? (size_t)n : 1, cols = W > 0 ? (size_t)W : 1;
    d->pcnt = calloc(rows, sizeof(int));
    d->ptab = calloc(rows * cols, sizeof(int));
  }
  d->pshtab = calloc((size_t)(n ? n : 1) * (W ? W : 1) * 3, sizeof(double));
  for (i = 0; i < n; i++) {
    d->pcnt[i] = ps.pfirst[i + 1] - ps.pfirst[i];
    for (m = 0; m < d->pcnt[i]; m++) {
      d->ptab[(size_t)i * W + m] = ps.ptag[ps.pfirst[i] + m];
      memcpy(&d->pshtab[((size_t)i * W + m) * 3], &ps.pshear[3 * (ps.pfirst[i] + m)], 3 * sizeof(double));
    }
  }
  d->have_ptab = 1;
  free(ps.pfirst); free(ps.ptag); free(ps.pshear);
  d->nghost = 0;
  d->next_ghost = 0;
  d->nsend[0] = d->nsend[1] = 0;
  d->recv_count[0] = d->recv_count[1] = 0;
  if (d->leave) memset(d->leave, 0, sizeof(int) * (size_t)d->localcap);
}

static void widen_ptab(orc_dem *d, int W, int ncap)
{
  int n = d->nlocal, i, m, W0 = d->mrec;
  int *nt = calloc((size_t)(ncap ? ncap : 1) * (W ? W : 1), sizeof(int));
  double *ns = calloc((size_t)(ncap ? ncap : 1) * (W ? W : 1) * 3, sizeof(double));
  int *nc = calloc((size_t)(ncap ? ncap : 1), sizeof(int));
  for (i = 0; i < n; i++) {
    nc[i] = d->pcnt[i];
    for (m = 0; m < d->pcnt[i]; m++) {
      nt[(size_t)i * W + m] = d->ptab[(size_t)i * W0 + m];
      memcpy(&ns[((size_t)i * W + m) * 3], &d->pshtab[((size_t)i * W0 + m) * 3], 3 * sizeof(double));
    }
  }
  free(d->pcnt); free(d->ptab); free(d->pshtab);
  d->pcnt = nc; d->ptab = nt; d->pshtab = ns;
  d->mrec = W;
}

void orc_dem_migrate_set_slots(orc_dem *d, int mrec)
{
  if (mrec < d->mrec) mrec = d->mrec;
  widen_ptab(d, mrec, d->nlocal);
}

int orc_dem_migrate_record_doubles(const orc_dem *d) { return 26 + 3 * nwalls_of(d) + 4 * d->mrec; }

long orc_dem_migrate_pack(orc_dem *d, int side, double xshift, double *buf, long max_doubles)
{
  int n = d->nlocal, i, c, s, w, nw = nwalls_of(d), W = d->mrec;
  int rec = orc_dem_migrate_record_doubles(d);
  long k = 0;
  if (!d->leave) d->leave = calloc((size_t)d->localcap, sizeof(int));
  for (i = 0; i < n; i++) {
    double x = d->x[3 * i];
    int sel = side == 0 ? (x < d->sublo) : (x >= d->subhi);
    if (!sel) continue;
    if ((k + 1) * rec > max_doubles) return -1;
    double *b = buf + k * rec;
    b[0] = d->x[3 * i] + xshift; b[1] = d->x[3 * i + 1]; b[2] = d->x[3 * i + 2]; b[3] = d->radius[i];
    b[4] = d->v[3 * i]; b[5] = d->v[3 * i + 1]; b[6] = d->v[3 * i + 2]; b[7] = d->rmass[i];
    b[8] = d->omega[3 * i]; b[9] = d->omega[3 * i + 1]; b[10] = d->omega[3 * i + 2];
    b[11] = d->tag[i]; b[12] = 1; b[13] = d->mask[i]; b[14] = 0;
    for (c = 0; c < 3; c++) {
      b[15 + c] = d->ffluiddrag[3 * i + c];
      b[18 + c] = d->DuDt[3 * i + c];
      b[21 + c] = d->vOld[3 * i + c];
    }
    b[24] = 0;
    for (w = 0; w < nw; w++)
      for (c = 0; c < 3; c++) b[25 + 3 * w + c] = wall_of(d, w)->wshear[3 * i + c];
    double *h = b + 25 + 3 * nw;
    h[0] = d->pcnt[i];
    for (s = 0; s < W; s++) {
      int ok = s < d->pcnt[i];
      h[1 + 4 * s] = ok ? d->ptab[(size_t)i * W + s] : -1.0;
      for (c = 0; c < 3; c++) h[2 + 4 * s + c] = ok ? d->pshtab[((size_t)i * W + s) * 3 + c] : 0.0;
    }
    d->leave[i] = side + 1;
    k++;
  }
  return k * rec;
}

static void copy_atom(orc_dem *d, int to, int from)
{
  int c, w, s, W = d->mrec, nw = nwalls_of(d);
  for (c = 0; c < 3; c++) {
    d->x[3 * to + c] = d->x[3 * from + c];
    d->v[3 * to + c] = d->v[3 * from + c];
    d->omega[3 * to + c] = d->omega[3 * from + c];
    d->f[3 * to + c] = d->f[3 * from + c];
    d->torque[3 * to + c] = d->torque[3 * from + c];
    d->ffluiddrag[3 * to + c] = d->ffluiddrag[3 * from + c];
    d->DuDt[3 * to + c] = d->DuDt[3 * from + c];
    d->vOld[3 * to + c] = d->vOld[3 * from + c];
  }
  d->radius[to] = d->radius[from]; d->rmass[to] = d->rmass[from];
  d->tag[to] = d->tag[from]; d->mask[to] = d->mask[from];
  for (w = 0; w < nw; w++)
    for (c = 0; c < 3; c++) wall_of(d, w)->wshear[3 * to + c] = wall_of(d, w)->wshear[3 * from + c];
  d->pcnt[to] = d->pcnt[from];
  for (s = 0; s < W; s++) {
    d->ptab[(size_t)to * W + s] = d->ptab[(size_t)from * W + s];
    memcpy(&d->pshtab[((size_t)to * W + s) * 3], &d->pshtab[((size_t)from * W + s) * 3], 3 * sizeof(double));
  }
}

static void compact(orc_dem *d)
{
  int i, k = 0;
  if (!d->leave) return;
  for (i = 0; i < d->nlocal; i++) {
    if (d->leave[i]) continue;
    if (k != i) copy_atom(d, k, i);
    k++;
  }
  memset(d->leave, 0, sizeof(int) * (size_t)d->localcap);
  d->nlocal = k;
}

void orc_dem_migrate_unpack(orc_dem *d, const double *buf, long ndoubles)
{
  int rec = orc_dem_migrate_record_doubles(d), nw = nwalls_of(d), W = d->mrec;
  int n = (int)(ndoubles / rec), k, c, w, s;
  compact(d);
  if (!n) return;
  int n0 = d->nlocal;
  ensure_local_cap(d, n0 + n);
  /* room in the partner table */
  d->pcnt = orc__xrealloc(d->pcnt, sizeof(int) * (size_t)(n0 + n));
  d->ptab = orc__xrealloc(d->ptab, sizeof(int) * (size_t)(n0 + n) * (W ? W : 1));
  d->pshtab = orc__xrealloc(d->pshtab, sizeof(double) * (size_t)(n0 + n) * (W ? W : 1) * 3);
  for (k = 0; k < n; k++) {
    const double *b = buf + (long)k * rec;
    int i = n0 + k;
    for (c = 0; c < 3; c++) {
      d->x[3 * i + c] = b[c];
      d->v[3 * i + c] = b[4 + c];
      d->omega[3 * i + c] = b[8 + c];
      d->f[3 * i + c] = d->torque[3 * i + c] = 0.0;
      d->ffluiddrag[3 * i + c] = b[15 + c];
      d->DuDt[3 * i + c] = b[18 + c];
      d->vOld[3 * i + c] = b[21 + c];
    }
    d->radius[i] = b[3]; d->rmass[i] = b[7];
    d->tag[i] = (int)b[11]; d->mask[i] = (int)b[13];
    for (w = 0; w < nw; w++)
      for (c = 0; c < 3; c++) wall_of(d, w)->wshear[3 * i + c] = b[25 + 3 * w + c];
    const double *h = b + 25 + 3 * nw;
    d->pcnt[i] = (int)h[0];
    for (s = 0; s < W; s++) {
      d->ptab[(size_t)i * W + s] = (int)h[1 + 4 * s];
      for (c = 0; c < 3; c++) d->pshtab[((size_t)i * W + s) * 3 + c] = h[2 + 4 * s + c];
    }
  }
  d->nlocal = n0 + n;
}

void orc_dem_rebuild_sort(orc_dem *d)
{
  compact(d);
  orc__pbc(d); /* y, z only in external mode */
}

static void push_send(orc_dem *d, int side, int i)
{
  if (d->nsend[side] >= d->sendcap[side]) {
    d->sendcap[side] = d->sendcap[side] * 2 + 1024;
    d->sendlist[side] = orc__xrealloc(d->sendlist[side], sizeof(int) * (size_t)d->sendcap[side]);
  }
  d->sendlist[side][d->nsend[side]++] = i;
}

long orc_dem_border_pack(orc_dem *d, int side, double xshift, double *buf, long max_atoms)
{
  double cut = orc__cutneighmax(d);
  int i;
  d->nsend[side] = 0;
  for (i = 0; i < d->nlocal; i++) {
    double x = d->x[3 * i];
    int sel = side == 0 ? (x < d->sublo + cut) : (x >= d->subhi - cut);
    if (!sel) continue;
    if (d->nsend[side] >= max_atoms) return -1;
    double *b = buf + (long)d->nsend[side] * BORDER_DOUBLES;
    b[0] = x + xshift; b[1] = d->x[3 * i + 1]; b[2] = d->x[3 * i + 2]; b[3] = d->radius[i];
    b[4] = d->v[3 * i]; b[5] = d->v[3 * i + 1]; b[6] = d->v[3 * i + 2]; b[7] = d->rmass[i];
    b[8] = d->omega[3 * i]; b[9] = d->omega[3 * i + 1]; b[10] = d->omega[3 * i + 2];
    b[11] = d->tag[i]; b[12] = 1; b[13] = d->mask[i];
    push_send(d, side, i);
  }
  return d->nsend[side];
}

void orc_dem_border_unpack(orc_dem *d, int side, const double *buf, long natoms)
{
  int n = (int)natoms, k, c;
  int first = d->nlocal + d->next_ghost;
  orc__grow_atoms(d, first + n + 1);
  d->recv_first[side] = first;
  d->recv_count[side] = n;
  for (k = 0; k < n; k++) {
    const double *b = buf + (long)k * BORDER_DOUBLES;
    int g = first + k;
    for (c = 0; c < 3; c++) {
      d->x[3 * g + c] = b[c];
      d->v[3 * g + c] = b[4 + c];
      d->omega[3 * g + c] = b[8 + c];
      d->gshift[3 * g + c] = 0.0;
    }
    d->radius[g] = b[3]; d->rmass[g] = b[7];
    d->tag[g] = (int)b[11]; d->mask[g] = (int)b[13];   /* group bits: the pair style asks for the freeze group of j */
    d->gsrc[g] = -1;
  }
  d->next_ghost += n;
}

long orc_dem_forward_pack(orc_dem *d, int side, double xshift, double *buf)
{
  int k, c;
  for (k = 0; k < d->nsend[side]; k++) {
    int i = d->sendlist[side][k];
    double *b = buf + (long)k * FORWARD_DOUBLES;
    b[0] = d->x[3 * i] + xshift; b[1] = d->x[3 * i + 1]; b[2] = d->x[3 * i + 2];
    for (c = 0; c < 3; c++) {
      b[3 + c] = d->v[3 * i + c];
      b[6 + c] = d->omega[3 * i + c];
    }
  }
  return d->nsend[side];
}

int orc_dem_forward_unpack(orc_dem *d, int side, const double *buf, long natoms)
{
  int k, c;
  if (natoms != d->recv_count[side]) return -1;
  for (k = 0; k < (int)natoms; k++) {
    const double *b = buf + (long)k * FORWARD_DOUBLES;
    int g = d->recv_first[side] + k;
    for (c = 0; c < 3; c++) {
      d->x[3 * g + c] = b[c];
      d->v[3 * g + c] = b[3 + c];
      d->omega[3 * g + c] = b[6 + c];
    }
  }
  return 0;
}

void orc_dem_ghost_forward_local(orc_dem *d) { orc__forward_comm(d); }

void orc_dem_rebuild_finish(orc_dem *d)
{
  /* partner table -> CSR for the list build */
  orc_partners ps;
  int n = d->nlocal, i, m, W = d->mrec;
  ensure_local_cap(d, n);
  ps.pfirst = calloc((size_t)n + 1, sizeof(int));
  for (i = 0; i < n; i++) ps.pfirst[i + 1] = ps.pfirst[i] + (d->have_ptab ? d->pcnt[i] : 0);
  ps.ptag = malloc(sizeof(int) * (size_t)(ps.pfirst[n] ? ps.pfirst[n] : 1));
  ps.pshear = malloc(sizeof(double) * 3 * (size_t)(ps.pfirst[n] ? ps.pfirst[n] : 1));
  for (i = 0; i < n; i++)
    for (m = 0; m < (d->have_ptab ? d->pcnt[i] : 0); m++) {
      ps.ptag[ps.pfirst[i] + m] = d->ptab[(size_t)i * W + m];
      memcpy(&ps.pshear[3 * (ps.pfirst[i] + m)], &d->pshtab[((size_t)i * W + m) * 3], 3 * sizeof(double));
    }
  free_ptab(d);
  orc__make_ghosts(d);
  orc__build_lists(d, &ps);
  d->flag = 0;
}
