// PairStyle gran/hertzFix/history on MI355X.  The force law and its quirks live in libsedifoam_amd.so
// (csrc/sf_physics.h: hertz_history_law = pair_gran_hertzFix_history.cpp:142-261); this file is the LAMMPS plumbing:
// settings() exactly as :293-317, compute() = flatten the half list + its history pages, upload the per-atom
// arrays, one kernel, add forces / torques back, return touch / shear to FixShearHistory's pages.
#include "pair_gran_hertzFix_history_amd.h"

#include <cstring>

#include "atom.h"
#include "comm.h"
#include "error.h"
#include "fix.h"
#include "force.h"
#include "memory.h"
#include "neigh_list.h"
#include "neighbor.h"
#include "update.h"

using namespace LAMMPS_NS;

PairGranHertzFixHistoryAmd::PairGranHertzFixHistoryAmd(LAMMPS *lmp) : PairGranHookeHistory(lmp), nrows_(0), npairs_(0)
{
  std::memset(&gp_, 0, sizeof(gp_));
}

/* pair_style gran/hertzFix/history kn kt|NULL gamman gammat|NULL xmu dampflag   (:293-317) */
void PairGranHertzFixHistoryAmd::settings(int narg, char **arg)
{
  if (narg != 6) error->all(FLERR, "Illegal pair_style command");
  const int kt_null = std::strcmp(arg[1], "NULL") == 0, gammat_null = std::strcmp(arg[3], "NULL") == 0;
  kn = force->numeric(FLERR, arg[0]);
  kt = kt_null ? kn * 2.0 / 7.0 : force->numeric(FLERR, arg[1]);
  gamman = force->numeric(FLERR, arg[2]);
  gammat = gammat_null ? 0.5 * gamman : force->numeric(FLERR, arg[3]);
  xmu = force->numeric(FLERR, arg[4]);
  dampflag = force->inumeric(FLERR, arg[5]);
  if (dampflag == 0) gammat = 0.0;
  if (sfk_gran_settings(&gp_, kn, 0, kt, gamman, 0, gammat, xmu, dampflag, force->nktv2p) != 0)
    error->all(FLERR, "Illegal pair_style command");
  kn /= force->nktv2p;   // :315-316 (the base class members are read by fix wall/granFix)
  kt /= force->nktv2p;
}

void PairGranHertzFixHistoryAmd::flatten_list()
{
  const int inum = list->inum;
  int *il = list->ilist, *numneigh = list->numneigh, **firstneigh = list->firstneigh;
  int **firsttouch = listgranhistory->firstneigh;
  double **firstshear = listgranhistory->firstdouble;
  ilist_.assign(il, il + inum);
  first_.resize(inum + 1);
  first_[0] = 0;
  for (int ii = 0; ii < inum; ii++) first_[ii + 1] = first_[ii] + numneigh[il[ii]];
  npairs_ = first_[inum];
  nrows_ = inum;
  jlist_.resize(npairs_ ? npairs_ : 1);
  touch_.resize(npairs_ ? npairs_ : 1);
  shear_.resize(3 * (size_t)(npairs_ ? npairs_ : 1));
  for (int ii = 0; ii < inum; ii++) {
    const int i = il[ii], n = numneigh[i], o = first_[ii];
    // j &= NEIGHMASK as the reference does with every list entry (pair_gran_hertzFix_history.cpp:122): the two top
    // bits of a LAMMPS neighbour word carry special-bond flags
    for (int jj = 0; jj < n; jj++) jlist_[o + jj] = firstneigh[i][jj] & NEIGHMASK;
    std::memcpy(&touch_[o], firsttouch[i], sizeof(int) * n);
    std::memcpy(&shear_[3 * (size_t)o], firstshear[i], sizeof(double) * 3 * n);
  }
  d_ilist_.upload(&ilist_[0], inum ? inum : 1);
  d_first_.upload(&first_[0], inum + 1);
  d_jlist_.upload(&jlist_[0], jlist_.size());
  d_touch_.upload(&touch_[0], touch_.size());
  d_shear_.upload(&shear_[0], shear_.size());
}

void PairGranHertzFixHistoryAmd::compute(int eflag, int vflag)
{
  if (eflag || vflag) ev_setup(eflag, vflag);
  else evflag = vflag_fdotr = 0;
  computeflag = 1;
  const int shearupdate = update->setupflag ? 0 : 1;   // :65-66

  // rigid body masses for owned & ghost atoms when a fix rigid is present (:68-86): body[i] = the body atom i is in, -1 if
  // none; the base class found fix_rigid in init_style and forwards mass_rigid to the ghosts (pack / unpack_comm)
  if (fix_rigid && neighbor->ago == 0) {
    int tmp;
    int *body = (int *) fix_rigid->extract("body", tmp);
    double *mass_body = (double *) fix_rigid->extract("masstotal", tmp);
    if (atom->nmax > nmax) {
      memory->destroy(mass_rigid);
      nmax = atom->nmax;
      memory->create(mass_rigid, nmax, "pair:mass_rigid");
    }
    for (int i = 0; i < atom->nlocal; i++) mass_rigid[i] = body[i] >= 0 ? mass_body[body[i]] : 0.0;
    comm->forward_comm_pair(this);
  }

  if (neighbor->ago == 0 || nrows_ != list->inum) flatten_list();

  const int nlocal = atom->nlocal, nall = nlocal + atom->nghost;
  const double *x = &atom->x[0][0], *v = &atom->v[0][0], *omega = &atom->omega[0][0];
  d_x_.upload(x, 3 * (size_t)nall);
  d_v_.upload(v, 3 * (size_t)nall);
  d_omega_.upload(omega, 3 * (size_t)nall);
  d_radius_.upload(atom->radius, nall);
  d_rmass_.upload(atom->rmass, nall);
  d_mask_.upload(atom->mask, nall);
  if (fix_rigid) d_mass_rigid_.upload(mass_rigid, nall);
  double *df = d_f_.zeros<double>(3 * (size_t)nall), *dt_ = d_torque_.zeros<double>(3 * (size_t)nall);

  // (:182-185: an atom of a rigid body collides with the mass of its body)
  if (sfk_pair_gran_history_compute_rigid(1, &gp_, dt, shearupdate, nlocal, nrows_, d_ilist_.as<int>(),
                                          d_first_.as<int>(), d_jlist_.as<int>(), d_touch_.as<int>(),
                                          d_shear_.as<double>(), d_x_.as<double>(), d_v_.as<double>(),
                                          d_omega_.as<double>(), d_radius_.as<double>(), d_rmass_.as<double>(),
                                          d_mask_.as<int>(), freeze_group_bit, df, dt_,
                                          fix_rigid ? d_mass_rigid_.as<double>() : NULL, NULL) != 0)
    error->one(FLERR, sf_last_error());

  // f / torque += (owned atoms only: newton off, :273), history back into FixShearHistory's pages
  hf_.resize(3 * (size_t)nlocal + 1);
  ht_.resize(3 * (size_t)nlocal + 1);
  sf_dev_download(&hf_[0], df, sizeof(double) * 3 * nlocal, NULL);
  sf_dev_download(&ht_[0], dt_, sizeof(double) * 3 * nlocal, NULL);
  double *f = &atom->f[0][0], *torque = &atom->torque[0][0];
  for (int k = 0; k < 3 * nlocal; k++) {
    f[k] += hf_[k];
    torque[k] += ht_[k];
  }
  if (npairs_) {
    sf_dev_download(&touch_[0], d_touch_.as<int>(), sizeof(int) * npairs_, NULL);
    sf_dev_download(&shear_[0], d_shear_.as<double>(), sizeof(double) * 3 * npairs_, NULL);
    int *il = list->ilist, *numneigh = list->numneigh;
    for (int ii = 0; ii < nrows_; ii++) {
      const int i = il[ii], n = numneigh[i], o = first_[ii];
      std::memcpy(listgranhistory->firstneigh[i], &touch_[o], sizeof(int) * n);
      std::memcpy(listgranhistory->firstdouble[i], &shear_[3 * (size_t)o], sizeof(double) * 3 * n);
    }
  }
}
