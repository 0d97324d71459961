// FixStyle cohesive on MI355X: `fix ID group cohesive ah lam smin smax opt` (fix_cohesive.cpp:38-47), its own half list
// (:75-77), post_force = sfk_fix_cohesive_post_force (:138-263).
#include "fix_cohesive_amd.h"

#include <cstdlib>
#include <cstring>

#include "atom.h"
#include "error.h"
#include "force.h"
#include "neigh_list.h"
#include "neigh_request.h"
#include "neighbor.h"
#include "update.h"

using namespace LAMMPS_NS;
using namespace FixConst;

FixCoheAmd::FixCoheAmd(LAMMPS *lmp, int narg, char **arg) : Fix(lmp, narg, arg), list(NULL), nrows_(-1)
{
  if (narg != 8) error->all(FLERR, "Illegal fix cohesive command");
  ah = std::atof(arg[3]);
  lam = std::atof(arg[4]);
  smin = std::atof(arg[5]);
  smax = std::atof(arg[6]);
  opt = std::atoi(arg[7]);
  if (opt != 0 && opt != 1) error->all(FLERR, "invalid option for cohesive force model");
}

int FixCoheAmd::setmask() { return POST_FORCE | POST_FORCE_RESPA | MIN_POST_FORCE; }

void FixCoheAmd::init()
{
  const int irequest = neighbor->request((void *)this);
  neighbor->requests[irequest]->pair = 0;
  neighbor->requests[irequest]->fix = 1;
}

void FixCoheAmd::init_list(int, NeighList *ptr) { list = ptr; }

void FixCoheAmd::setup() { post_force(1); }

void FixCoheAmd::post_force(int)
{
  const int nlocal = atom->nlocal, nall = nlocal + atom->nghost;
  if (!nlocal || !list) return;
  if (neighbor->ago == 0 || nrows_ != nlocal) {
    // rows = ilist[ii] for ii < nlocal, as the reference's loop (:156-160)
    int *il = list->ilist, *numneigh = list->numneigh, **firstneigh = list->firstneigh;
    ilist_.assign(il, il + nlocal);
    first_.resize(nlocal + 1);
    first_[0] = 0;
    for (int ii = 0; ii < nlocal; ii++) first_[ii + 1] = first_[ii] + numneigh[il[ii]];
    jlist_.resize(first_[nlocal] ? first_[nlocal] : 1);
    for (int ii = 0; ii < nlocal; ii++)
      std::memcpy(&jlist_[first_[ii]], firstneigh[il[ii]], sizeof(int) * numneigh[il[ii]]);
    d_ilist_.upload(&ilist_[0], nlocal);
    d_first_.upload(&first_[0], nlocal + 1);
    d_jlist_.upload(&jlist_[0], jlist_.size());
    nrows_ = nlocal;
  }
  d_x_.upload(&atom->x[0][0], 3 * (size_t)nall);
  d_radius_.upload(atom->radius, nall);
  d_mask_.upload(atom->mask, nall);
  double *df = d_f_.zeros<double>(3 * (size_t)nall);
  if (sfk_fix_cohesive_post_force(ah, lam, smin, smax, opt, nlocal, force->newton_pair, d_ilist_.as<int>(),
                                  d_first_.as<int>(), d_jlist_.as<int>(), d_x_.as<double>(), d_radius_.as<double>(),
                                  d_mask_.as<int>(), groupbit, df, NULL) != 0)
    error->one(FLERR, sf_last_error());
  const size_t nf = 3 * (size_t)(force->newton_pair ? nall : nlocal);
  hf_.resize(nf);
  sf_dev_download(&hf_[0], df, sizeof(double) * nf, NULL);
  double *f = &atom->f[0][0];
  for (size_t k = 0; k < nf; k++) f[k] += hf_[k];
}
