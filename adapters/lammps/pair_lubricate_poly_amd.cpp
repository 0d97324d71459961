// PairStyle lubricate/poly on MI355X: settings / coeff / init_style (R0, RT0, RS0 from the global volume fraction,
// pair_lubricate_poly.cpp:450-577) stay the stock class's; compute() (:65-444, Ef = 0: no fix deform) is
// sfk_pair_lubricate_poly_compute on the full list.
#include "pair_lubricate_poly_amd.h"

#include <cstring>

#include "atom.h"
#include "error.h"
#include "force.h"
#include "neigh_list.h"
#include "neighbor.h"

using namespace LAMMPS_NS;

void PairLubricatePolyAmd::compute(int eflag, int vflag)
{
  if (eflag || vflag) ev_setup(eflag, vflag);
  else evflag = vflag_fdotr = 0;
  const int inum = list->inum, nlocal = atom->nlocal, nall = nlocal + atom->nghost;
  if (!inum) return;
  if (neighbor->ago == 0 || nrows_ != inum) {
    int *il = list->ilist, *numneigh = list->numneigh, **firstneigh = list->firstneigh;
    ilist_.assign(il, il + inum);
    first_.resize(inum + 1);
    first_[0] = 0;
    for (int ii = 0; ii < inum; ii++) first_[ii + 1] = first_[ii] + numneigh[il[ii]];
    jlist_.resize(first_[inum] ? first_[inum] : 1);
    for (int ii = 0; ii < inum; ii++)
      std::memcpy(&jlist_[first_[ii]], firstneigh[il[ii]], sizeof(int) * numneigh[il[ii]]);
    d_ilist_.upload(&ilist_[0], inum);
    d_first_.upload(&first_[0], inum + 1);
    d_jlist_.upload(&jlist_[0], jlist_.size());
    nrows_ = inum;
  }
  sfk_lub_params p;
  p.mu = mu;
  p.flaglog = flaglog;
  p.flagfld = flagfld;
  p.flagHI = flagHI;
  p.flagVF = flagVF;
  p.cut_inner = cut_inner_global;
  p.cut_global = cut_global;
  p.R0 = R0;
  p.RT0 = RT0;
  p.RS0 = RS0;
  p.vxmu2f = force->vxmu2f;
  d_x_.upload(&atom->x[0][0], 3 * (size_t)nall);
  d_v_.upload(&atom->v[0][0], 3 * (size_t)nall);
  d_omega_.upload(&atom->omega[0][0], 3 * (size_t)nall);
  d_radius_.upload(atom->radius, nall);
  double *df = d_f_.zeros<double>(3 * (size_t)nlocal), *dt = d_torque_.zeros<double>(3 * (size_t)nlocal);
  if (sfk_pair_lubricate_poly_compute(&p, inum, d_ilist_.as<int>(), d_first_.as<int>(), d_jlist_.as<int>(),
                                      d_x_.as<double>(), d_v_.as<double>(), d_omega_.as<double>(),
                                      d_radius_.as<double>(), df, dt, NULL) != 0)
    error->one(FLERR, sf_last_error());
  hf_.resize(3 * (size_t)nlocal);
  ht_.resize(3 * (size_t)nlocal);
  sf_dev_download(&hf_[0], df, sizeof(double) * 3 * nlocal, NULL);
  sf_dev_download(&ht_[0], dt, sizeof(double) * 3 * nlocal, NULL);
  double *f = &atom->f[0][0], *torque = &atom->torque[0][0];
  for (int k = 0; k < 3 * nlocal; k++) {   // only i is updated (full list, newton off)
    f[k] += hf_[k];
    torque[k] += ht_[k];
  }
}
