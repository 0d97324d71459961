/* orc_dem_priv.h -- internals of the oracle DEM driver shared by orc_dem.c and orc_halo.c
 * (TEST INFRASTRUCTURE ONLY). */
#ifndef ORC_DEM_PRIV_H
#define ORC_DEM_PRIV_H
#include "sedifoam_oracle.h"

enum { FIX_GRAVITY = 1, FIX_FDRAG, FIX_WALL, FIX_COHESIVE, FIX_FREEZE };

typedef struct {
  int kind;
  int groupbit;   /* fix ID <group> ...: bit of the group in mask[] (1 = all) */
  /* gravity */
  double gmag, gdir[3];
  /* fdrag */
  double carrier_rho;
  /* wall */
  int wallstyle;                       /* 0/1/2 plane normal, 3 z cylinder */
  double lo, hi, cylradius;
  int wiggleflag, shearflag, axis;     /* fix_wall_granFix.cpp:117-141 (wiggle, wshear) */
  double amplitude, period, vshear;
  long time_origin;                    /* :181 */
  orc_gran_params wp;
  double *wshear; /* 3*nmax */
  /* cohesive */
  double ah, lam, smin, smax;
  int opt;
} orc_fix;

#define MAXFIX 16

struct orc_dem {
  int nlocal, nghost, nmax;
  double *x, *v, *omega, *f, *torque, *radius, *rmass;
  int *tag, *mask;
  int *gsrc;           /* ghost -> index it copies (may itself be a ghost of an earlier dim) */
  double *gshift;      /* 3 per ghost */
  double *xhold;       /* 3*nlocal at last build */
  double boxlo[3], boxhi[3];
  int periodic[3];
  double skin, dt;
  int pair_style;      /* 0 none 1 hooke 2 hertz */
  orc_gran_params gp;
  int have_lub;
  orc_lub_params lub;
  int nfix;
  orc_fix fix[MAXFIX];
  /* fix fdrag per-atom arrays (fix_fluid_drag.cpp:181-187) */
  double *ffluiddrag, *DuDt, *vOld;
  /* granular half list + history */
  int *first, *jlist, *touch;
  double *shear;
  int listcap;
  /* regular half list (fix cohesive) and full list (lubricate/poly) */
  int *hfirst, *hjlist;
  int hcap;
  int *ffirst, *fjlist;
  int fcap;
  int *ilist;
  /* bins */
  int *binhead, *binnext;
  int nbins_alloc;
  int nbuilds;
  long ntimestep;        /* update->ntimestep [3P] */
  int setup_done;
  int nve_bit, freeze_bit;   /* group of fix nve/sphere (default all), of fix freeze (0: none) */
  int nthreads;
  /* ---- external x halo (orc_halo.c): this driver is one slab of an x-decomposed domain ---- */
  int external_x;
  double rmax_global;    /* largest radius over all ranks (0: single domain, local maximum) */
  double sublo, subhi;
  int localcap;          /* capacity of the per-owned-atom arrays (xhold, fix fdrag arrays, ilist, wall shear) */
  int next_ghost;        /* ghosts received from the neighbour slabs, appended before the local images */
  int mrec;              /* history slots per migrating atom */
  int flag;              /* an owned atom moved > skin/2 since the last build */
  int *pcnt, *ptab;      /* partner table [n][mrec] between rebuild_begin and rebuild_finish */
  double *pshtab;
  int have_ptab;
  int *sendlist[2];
  int nsend[2], sendcap[2];
  int recv_first[2], recv_count[2];
  int *leave;
};


typedef struct {
  int *pfirst;   /* nlocal+1 */
  int *ptag;
  double *pshear;
} orc_partners;

void *orc__xrealloc(void *p, size_t n);
void orc__grow_atoms(orc_dem *d, int nmax);
void orc__pbc(orc_dem *d);
void orc__make_ghosts(orc_dem *d);
void orc__forward_comm(orc_dem *d);
void orc__partners_from_list(const orc_dem *d, orc_partners *ps);
void orc__build_lists(orc_dem *d, orc_partners *ps); /* frees ps */
int orc__check_distance(const orc_dem *d);
void orc__compute_forces(orc_dem *d, int setupflag);
double orc__cutneighmax(const orc_dem *d);
#endif
