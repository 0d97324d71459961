/* sedifoam_oracle.h -- CPU restatement ("oracle") of sediFoam's CFD-DEM particle hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and there
 * only as the checker / the timed CPU baseline.  The product (sedifoam_amd/) never links,
 * imports or falls back to it.
 *
 * Every function restates, in plain C, the algorithm of the reference file:line it cites
 * (paths relative to the reference checkout).  Pieces that live in the un-vendored third-party
 * dependencies (LAMMPS 1Feb14: nve/sphere, gravity, gran/hooke/history, neighbor binning, shear
 * history carry-over; OpenFOAM: cell lookup of a uniform blockMesh) are marked [3P] and restate
 * the published upstream algorithm, anchored on the reference's call sites.
 *
 * PARITY PINNING STATUS
 *   pinned by the reference's own golden vectors (tests/test_oracle_golden.py):
 *     - cases/auto-testing/test-cases/xiaocase3/data/{lammps08,xiaoCase3}.dat
 *     - cases/auto-testing/test-cases/multiParticlesCollideRho/data/origin/p[1-4].dat
 *     which exercise fix fdrag, nve/sphere, gravity, wall/gran hooke_history, gran/hooke/history,
 *     SyamlalOBrien, the drag assembly of enhancedCloud and the sub-cycling rule.
 *   pinned by numbers computed by THE REFERENCE'S OWN SOURCE LINES (tests/golden/reference_pins.json, generated in the
 *   build container by tests/golden/make_reference_pins.py, which reads the line ranges below from the reference at
 *   generation time, transliterates them statement by statement and executes them on seeded LAMMPS-shaped inputs;
 *   tests/test_reference_pins.py: the oracle equals them BIT FOR BIT, tests/test_reference_pins_gpu.py: the HIP styles
 *   to 1e-12):
 *     - gran/hertzFix/history   pair_gran_hertzFix_history.cpp:109-286 (the ii / jj loop of compute)
 *     - fix cohesive            fix_cohesive.cpp:161-262 (opt 0 and 1)
 *     - fix fdrag               fix_fluid_drag.cpp:143-163
 *     - fix wall/granFix        fix_wall_granFix.cpp:286-344 + hooke :361-436, hooke_history :446-553,
 *                               hertz_history :563-678
 *     - pair lubricate/poly     pair_lubricate_poly.cpp:193-407 (compute loop) + :539-559 (init_style constants)
 *     - ErgunWenYu / SyamlalOBrien   ErgunWenYu.C:104-132, SyamlalOBrien.C:105-143
 *     The reference itself still cannot be BUILT here (LAMMPS / OpenFOAM headers are not in the image): what is pinned
 *     is the arithmetic of those lines, not their integration into LAMMPS' Verlet loop or OpenFOAM's cloud.
 *   "parity unpinned" (restated from the published upstream algorithm or the linear system the reference assembles, and
 *   checked by hand-derived known answers only): the [3P] LAMMPS 1Feb14 machinery (neighbour list, FixShearHistory,
 *   nve/sphere beyond what the golden curves exercise, gran/hooke) and smoothField's implicit diffusion solve.
 *
 * Layout convention: AoS like LAMMPS (double x[n][3] flattened to 3*n), int32 ids.
 */
#ifndef SEDIFOAM_ORACLE_H
#define SEDIFOAM_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NEIGHMASK 0x3FFFFFFF

/* ---- granular pair parameters (pair_gran_hertzFix_history.cpp:293-317) ---- */
typedef struct {
  double kn, kt, gamman, gammat, xmu;
  int dampflag;
} orc_gran_params;

/* settings(): "kn kt|NULL gamman gammat|NULL xmu dampflag"; kt_null/gammat_null flag "NULL".
 * nktv2p = 1 in the default lj units used by every reference case.  returns 0 ok, -1 illegal */
int orc_gran_settings(orc_gran_params *p, double kn, int kt_null, double kt, double gamman,
                      int gammat_null, double gammat, double xmu, int dampflag, double nktv2p);

/* half neighbour list in CSR form: neighbours of list row ii are jlist[first[ii]..first[ii+1]) */
typedef struct {
  int inum;
  const int *ilist;   /* inum */
  const int *first;   /* inum+1 */
  const int *jlist;   /* first[inum] ; may carry bits above ORC_NEIGHMASK */
  int *touch;         /* first[inum] */
  double *shear;      /* 3*first[inum] */
} orc_neighlist;

/* A1: PairGranHertzFixHistory::compute  pair_gran_hertzFix_history.cpp:45-287 */
void orc_pair_gran_hertzfix_history(const orc_gran_params *p, double dt, int shearupdate,
                                    int nlocal, const double *x, const double *v,
                                    const double *omega, const double *radius,
                                    const double *rmass, const int *mask, int freeze_group_bit,
                                    const orc_neighlist *list, double *f, double *torque);

/* [3P] PairGranHookeHistory::compute (LAMMPS 1Feb14); same skeleton, Hookean law as in
 * fix_wall_granFix.cpp:441-554 with meff of the pair */
/* [3P] PairGranHooke::compute (LAMMPS 1Feb14 pair_gran_hooke.cpp; not in the reference tree): the plain Hookean law
 * of fix_wall_granFix.cpp:347-437 between two grains, no shear history (list->shear stays zero) */
void orc_pair_gran_hooke(const orc_gran_params *p, int nlocal, const double *x, const double *v,
                         const double *omega, const double *radius, const double *rmass, const int *mask,
                         int freeze_group_bit, const orc_neighlist *list, double *f, double *torque);
void orc_pair_gran_hooke_history(const orc_gran_params *p, double dt, int shearupdate,
                                 int nlocal, const double *x, const double *v,
                                 const double *omega, const double *radius,
                                 const double *rmass, const int *mask, int freeze_group_bit,
                                 const orc_neighlist *list, double *f, double *torque);

/* A2: FixCohe::post_force  fix_cohesive.cpp:138-263.  list rows are ii<nlocal, i=ilist[ii]
 * (the reference loops ii<nlocal, not ii<inum).  returns 0, or -1 for an invalid opt */
int orc_fix_cohesive(double ah, double lam, double smin, double smax, int opt, int nlocal,
                     int newton_pair, const double *x, const double *radius, const int *mask,
                     int groupbit, const orc_neighlist *list, double *f);

/* A3: PairLubricatePoly::compute  pair_lubricate_poly.cpp:65-444 (no shearing: Ef = 0).
 * full list; one atom type: cutsq / cut_inner scalars */
typedef struct {
  double mu;
  int flaglog, flagfld, flagHI, flagVF;
  double cut_inner, cut_global;
  double R0, RT0, RS0;      /* set by orc_lubricate_init */
  double vxmu2f;            /* 1 in lj units */
} orc_lub_params;
/* init_style(): pair_lubricate_poly.cpp:450-577 (volume-fraction constants) */
void orc_lubricate_init(orc_lub_params *p, int nlocal_all, const double *radius, double vol_T);
double orc_particle_volume(int nlocal, const double *radius);
void orc_lubricate_init_vol(orc_lub_params *p, double volP, double vol_T);
void orc_pair_lubricate_poly(const orc_lub_params *p, int nlocal, const double *x,
                             const double *v, const double *omega, const double *radius,
                             const orc_neighlist *fulllist, double *f, double *torque);

/* A4: FixFluidDrag::post_force  fix_fluid_drag.cpp:114-164 */
void orc_fix_fluid_drag(int nlocal, double dt, double carrier_rho, const double *v,
                        const double *rmass, const double *radius, const int *mask,
                        int groupbit, const double *ffluiddrag, const double *DuDt,
                        double *vOld, double *f);

/* N2: FixWallGranFix::post_force for plane walls  fix_wall_granFix.cpp:247-345,
 * hooke_history :441-554, hertz_history :558-679.  wallstyle 0/1/2 = x/y/z plane;
 * lo/hi = +-1e20 when NULL.  pairstyle 1 = hooke_history, 2 = hertz_history, 3 = hooke (:347-437). */
void orc_fix_wall_gran_moving(const orc_gran_params *p, int pairstyle, int wallstyle, double lo, double hi,
                              double cylradius, int wiggle, int wshear, int axis, double amplitude,
                              double period, double vshear, long steps, double dt, int shearupdate,
                              int nlocal, const double *x, const double *v, const double *omega,
                              const double *radius, const double *rmass, const int *mask, int groupbit,
                              double *shear, double *f, double *torque);
void orc_fix_wall_gran(const orc_gran_params *p, int pairstyle, int wallstyle, double lo,
                       double hi, double dt, int shearupdate, int nlocal, const double *x,
                       const double *v, const double *omega, const double *radius,
                       const double *rmass, const int *mask, int groupbit, double *shear,
                       double *f, double *torque);

/* [3P] LAMMPS 1Feb14 FixNVESphere / FixGravity */
void orc_nve_sphere_initial(int nlocal, double dt, double *x, double *v, double *omega,
                            const double *f, const double *torque, const double *radius,
                            const double *rmass);
void orc_nve_sphere_final(int nlocal, double dt, double *v, double *omega, const double *f,
                          const double *torque, const double *radius, const double *rmass);
void orc_fix_gravity(int nlocal, double magnitude, const double dir[3], const double *rmass,
                     double *f);

/* ---- DEM driver: a restatement of what `lammps_step(n)` does to the particles ---- */
typedef struct orc_dem orc_dem;

void orc_nve_sphere_initial_group(int nlocal, double dt, double *x, double *v, double *omega, const double *f,
                                  const double *torque, const double *radius, const double *rmass,
                                  const int *mask, int groupbit);
void orc_nve_sphere_final_group(int nlocal, double dt, double *v, double *omega, const double *f,
                                const double *torque, const double *radius, const double *rmass,
                                const int *mask, int groupbit);
void orc_fix_gravity_group(int nlocal, double magnitude, const double dir[3], const double *rmass,
                           const int *mask, int groupbit, double *f);
void orc_fix_freeze(int nlocal, const int *mask, int groupbit, double *f, double *torque);

orc_dem *orc_dem_create(int n, const double *x, const double *v, const double *omega,
                        const double *radius, const double *rmass, const int *tag,
                        const double boxlo[3], const double boxhi[3], const int periodic[3]);
void orc_dem_destroy(orc_dem *d);
/* style: 1 gran/hooke/history, 2 gran/hertzFix/history, 3 gran/hooke [3P], 0 none */
int orc_dem_pair_gran(orc_dem *d, int style, double kn, int kt_null, double kt, double gamman,
                      int gammat_null, double gammat, double xmu, int dampflag);
void orc_dem_pair_lubricate(orc_dem *d, double mu, int flaglog, int flagfld, double cut_inner,
                            double cut_global, int flagHI, int flagVF);
void orc_dem_fix_cohesive(orc_dem *d, double ah, double lam, double smin, double smax, int opt);
void orc_dem_fix_gravity(orc_dem *d, double magnitude, double gx, double gy, double gz);
void orc_dem_fix_fdrag(orc_dem *d, double carrier_rho);
/* [3P] fix freeze at its place in the fix list: the fixes registered after it still act on the frozen atoms */
void orc_dem_fix_freeze(orc_dem *d, int groupbit);
/* wallstyle 0/1/2 ; lo_null/hi_null mark NULL bounds */
void orc_dem_fix_wall(orc_dem *d, int wallstyle, int lo_null, double lo, int hi_null, double hi,
                      double kn, int kt_null, double kt, double gamman, int gammat_null,
                      double gammat, double xmu, int dampflag);
/* groups: per-atom group bits (bit 0 = all) in creation order, then the group of every fix kind registered so far
 * (LAMMPS: `fix ID group style ...`); freeze_bit = 0: no fix freeze */
void orc_dem_set_mask(orc_dem *d, const int *mask);
/* the last registered wall becomes `zcylinder radius` (fix_wall_granFix.cpp:107-112) */
void orc_dem_wall_cylinder(orc_dem *d, double cylradius);
/* the last registered wall moves: kind 1 = wiggle axis amplitude period, kind 2 = shear axis vshear (:117-141) */
void orc_dem_wall_motion(orc_dem *d, int kind, int axis, double a, double b);
void orc_dem_set_groups(orc_dem *d, int nve_bit, int gravity_bit, int fdrag_bit, int wall_bit, int cohesive_bit,
                        int freeze_bit);
void orc_dem_neighbor(orc_dem *d, double skin);
void orc_dem_timestep(orc_dem *d, double dt);
/* threads > 1: split the i-loop of the pair kernel over pthreads (bench cpu_baseline only) */
void orc_dem_threads(orc_dem *d, int nthreads);

void orc_dem_setup(orc_dem *d);          /* first `run`: build list, forces with shearupdate=0 */
void orc_dem_run(orc_dem *d, int nsteps); /* run nsteps pre no post no */

int orc_dem_nlocal(const orc_dem *d);
int orc_dem_nghost(const orc_dem *d);
int orc_dem_nbuilds(const orc_dem *d);
long orc_dem_npairs(const orc_dem *d);    /* half-list pairs in the current list */
/* copy out local-atom arrays in the driver's current order (AoS) ; any pointer may be NULL */
void orc_dem_get(const orc_dem *d, double *x, double *v, double *omega, double *f,
                 double *torque, int *tag);
/* library.cpp:314-367 semantics: rows matched to atoms by tag */
void orc_dem_put_fdrag(orc_dem *d, int n, const double *fdrag, const int *tag);
/* touching pairs of the current list as (tag_i, tag_j, shear[3]) ; returns count (<= max) */
int orc_dem_get_history(const orc_dem *d, int max, int *tag_i, int *tag_j, double *shear);
/* per-atom wall shear of wall w (AoS 3*nlocal, driver order) */
void orc_dem_get_wall_shear(const orc_dem *d, int w, double *shear);

/* ---- one slab of an x-decomposed domain (orc_halo.c): same calls and record layouts as the product's
 * sf_dem_* halo entry points, so sedifoam_amd/halo.py can be driven on CPUs (gloo) ---- */
void orc_dem_set_subdomain(orc_dem *d, double sublo, double subhi);
int orc_dem_max_partners(const orc_dem *d);
void orc_dem_run_begin(orc_dem *d);
void orc_dem_substep(orc_dem *d, int last);
int orc_dem_need_rebuild(const orc_dem *d);
void orc_dem_ext_setup(orc_dem *d);
/* lubricate/poly on a decomposed domain: the twin of MPI_Allreduce(volP) pair_lubricate_poly.cpp:540-543 */
double orc_dem_local_particle_volume(const orc_dem *d);
void orc_dem_set_global_particle_volume(orc_dem *d, double volP);
double orc_dem_local_max_radius(const orc_dem *d);
void orc_dem_set_global_max_radius(orc_dem *d, double rmax);
void orc_dem_rebuild_begin(orc_dem *d);
void orc_dem_rebuild_sort(orc_dem *d);
void orc_dem_rebuild_finish(orc_dem *d);
void orc_dem_migrate_set_slots(orc_dem *d, int mrec);
int orc_dem_migrate_record_doubles(const orc_dem *d);
long orc_dem_migrate_pack(orc_dem *d, int side, double xshift, double *buf, long max_doubles);
void orc_dem_migrate_unpack(orc_dem *d, const double *buf, long ndoubles);
long orc_dem_border_pack(orc_dem *d, int side, double xshift, double *buf, long max_atoms);
void orc_dem_border_unpack(orc_dem *d, int side, const double *buf, long natoms);
long orc_dem_forward_pack(orc_dem *d, int side, double xshift, double *buf);
int orc_dem_forward_unpack(orc_dem *d, int side, const double *buf, long natoms);
void orc_dem_ghost_forward_local(orc_dem *d);

/* ---- OpenFOAM side (enhancedCloud / dragModels) ---- */
/* A5: ErgunWenYu::Jd  lammpsFoam/dragModels/ErgunWenYu/ErgunWenYu.C:86-145 */
void orc_ergun_wenyu_jd(int n, const double *Ur, const double *alpha, const double *pd,
                        double nuf, double rhof, double *Jd);
/* N4: SyamlalOBrien::Jd  lammpsFoam/dragModels/SyamlalOBrien/SyamlalOBrien.C:85-144 */
void orc_syamlal_obrien_jd(int n, const double *Ur, const double *alpha, const double *pd,
                           double nuf, double rhof, double *Jd);

/* A7: cell owner for one uniform blockMesh hex block; -1 outside (OpenFOAM drops the particle) */
void orc_no_correction_jd(int n, const double *Ur, const double *alpha, const double *pd,
                          double nuf, double rhof, double *Jd);
void orc_cell_owner_graded(int n, const double *x, const double origin[3], const double dx[3],
                           const int ncell[3], const double *const faces[3], int *cell);
void orc_cell_owner(int n, const double *x, const double origin[3], const double dx[3],
                    const int ncell[3], int *cell);

typedef struct {
  int particleDrag, particlePressureGrad, particleBuoyancy, particleAddedMass, particleLift,
      lubricationForce;
  double gravity[3];
  double rhob, nub, deltaT;
} orc_cloud_flags;

/* A6: updateParticleUr + updateDragOnParticles  enhancedCloud.C:83-109,112-257
 * dragModel 0 = ErgunWenYu, 1 = SyamlalOBrien.  History force and inlet override excluded. */
void orc_drag_on_particles(const orc_cloud_flags *fl, int dragModel, int n, const int *cell,
                           const double *pos, const double *d, const double *U,
                           const double *UOld, const double *gamma, const double *UfSmoothed,
                           const double *gradp, const double *DDtUf, const double *curlU,
                           double *Uri, double *magUri, double *Jd, double *pDrag,
                           double *pDuDt);
/* the inlet override that closes the particle loop of updateDragOnParticles (enhancedCloud.C:249-257) with
 * softParticleCloud::pointInRegion (softParticleCloud.C:1354-1417): applied to pDrag [n][3] after the assembly above */
void orc_inlet_force_override(int addParticleOption, const double inletForce[3], const double inletBox[9],
                              const double eccentricity[3], double deltaT, int n, const double *pos,
                              const double *mass, const double *U, double *pDrag);
void orc_drag_on_particles_hist(const orc_cloud_flags *fl, int dragModel, int n, const int *cell,
                                const double *pos, const double *d, const double *U,
                                const double *UOld, const double *gamma, const double *UfSmoothed,
                                const double *gradp, const double *DDtUf, const double *curlU,
                                int timeIndex, const double *UfSmoothedOld, double *sumDeltaFb, double *n0,
                                double *Uri, double *magUri, double *Jd, double *pDrag,
                                double *pDuDt);

/* A8: particleToEulerianField  enhancedCloud.C:911-980 (no diffusion smoothing) */
void orc_particle_to_eulerian(int n, const int *cell, const double *d, const double *U,
                              int ncells, const double *V, double *gamma, double *Ue);

/* A9: calcTcFields  enhancedCloud.C:316-441 (no smoothing): Asrc, Omega(=0 on exit) */
void orc_calc_tc_fields(int n, const int *cell, const double *d, const double *U,
                        const double *Jd, int ncells, const double *V, const double *gamma,
                        const double *UfSmoothed, double *Asrc, double *Omega);

/* N1: enhancedCloud::smoothField  enhancedCloud.C:790-907 on a uniform hex block, and the three places the
 * reference applies it (:675-690 UfSmoothed, :944-962 gamma / Ue, :407-416 Asrc) */
typedef struct {
  int n[3];
  double dx[3], D[3];
  double band;
  int steps;
  int UfSmooth, UpSmooth, dragSmooth, alphaSmooth;
  const double *w[3];   /* graded block: cell widths along each axis (NULL = uniform dx) */
  int periodic[3];      /* cyclic patch pair along this axis (the reference's channel cases) instead of zeroGradient */
} orc_smooth;
void orc_smooth_field_periodic(const int n[3], const double dx[3], const double D[3], double band, int steps,
                               int ncomp, double *field, const int *periodic);
void orc_smooth_field_graded_periodic(const int n[3], const double dx[3], const double *const w[3],
                                      const double D[3], double band, int steps, int ncomp, double *field,
                                      const int *periodic);
void orc_smooth_field_graded(const int n[3], const double dx[3], const double *const w[3], const double D[3],
                             double band, int steps, int ncomp, double *field);
void orc_smooth_field(const int n[3], const double dx[3], const double D[3], double band, int steps, int ncomp,
                      double *field);
void orc_particle_to_eulerian_smooth(int n, const int *cell, const double *d, const double *U, int ncells,
                                     const double *V, const orc_smooth *sm, double *gamma, double *Ue);
void orc_uf_smoothed(int ncells, const double *Uf, const double *gamma, const orc_smooth *sm, double *UfS);
void orc_calc_tc_fields_smooth(int n, const int *cell, const double *d, const double *U, const double *Jd,
                               int ncells, const double *V, const double *gamma, const double *UfSmoothed,
                               const orc_smooth *sm, double *Asrc, double *Omega);

/* A10: adjustLampTimestep  softParticleCloud.C:209-261.  returns 0, or -1 (FatalError case) */
int orc_adjust_timestep(double deltaT, double dtLampIn, int subCycles_in, double *dtLampAdj,
                        int *solidStepsPerDt, int *subCycles, int *subSteps);

#ifdef __cplusplus
}
#endif
#endif
