// FixStyle fdrag on MI355X: the per-atom arrays and their migration are LAMMPS plumbing kept on the host exactly as
// the reference has them (fix_fluid_drag.cpp:30-112, :166-243); post_force() (:114-164, incl. the mistyped pi of :147
// and the never-written DuDt) is sfk_fix_fluid_drag_post_force.
#include "fix_fluid_drag_amd.h"

#include <cstdlib>
#include <cstring>

#include "atom.h"
#include "error.h"
#include "memory.h"
#include "update.h"

using namespace LAMMPS_NS;
using namespace FixConst;

FixFluidDragAmd::FixFluidDragAmd(LAMMPS *lmp, int narg, char **arg) : Fix(lmp, narg, arg)
{
  if (narg < 3) error->all(FLERR, "Illegal fix fdrag command");
  ffluiddrag = DuDt = vOld = NULL;
  foamCpuId = NULL;
  grow_arrays(atom->nmax);
  atom->add_callback(0);
  force_reneighbor = 0;
  carrier_rho = narg == 4 ? std::atoi(arg[3]) : 0;   // integer parse, as the reference (:49-54)
}

FixFluidDragAmd::~FixFluidDragAmd()
{
  atom->delete_callback(id, 0);
  memory->destroy(ffluiddrag);
  memory->destroy(DuDt);
  memory->destroy(vOld);
  memory->destroy(foamCpuId);
}

int FixFluidDragAmd::setmask() { return POST_FORCE; }

void FixFluidDragAmd::init()
{
  const int nlocal = atom->nlocal;
  const int *mask = atom->mask;
  for (int i = 0; i < nlocal; i++)
    if (mask[i] & groupbit) {
      for (int k = 0; k < 3; k++) ffluiddrag[i][k] = DuDt[i][k] = vOld[i][k] = 0.0;
      foamCpuId[i] = 0;
    }
}

void FixFluidDragAmd::setup(int vflag)
{
  if (std::strcmp(update->integrate_style, "verlet") == 0) post_force(vflag);
}

void FixFluidDragAmd::post_force(int)
{
  const int nlocal = atom->nlocal;
  if (!nlocal) return;
  const size_t n3 = 3 * (size_t)nlocal;
  d_v_.upload(&atom->v[0][0], n3);
  d_rmass_.upload(atom->rmass, nlocal);
  d_radius_.upload(atom->radius, nlocal);
  d_mask_.upload(atom->mask, nlocal);
  d_fd_.upload(&ffluiddrag[0][0], n3);
  d_dudt_.upload(&DuDt[0][0], n3);
  d_vold_.upload(&vOld[0][0], n3);
  double *df = d_f_.zeros<double>(n3);
  if (sfk_fix_fluid_drag_post_force(nlocal, update->dt, carrier_rho, d_v_.as<double>(), d_rmass_.as<double>(),
                                    d_radius_.as<double>(), d_mask_.as<int>(), groupbit, d_fd_.as<double>(),
                                    d_dudt_.as<double>(), d_vold_.as<double>(), df, NULL) != 0)
    error->one(FLERR, sf_last_error());
  hf_.resize(n3);
  sf_dev_download(&hf_[0], df, sizeof(double) * n3, NULL);
  sf_dev_download(&vOld[0][0], d_vold_.as<double>(), sizeof(double) * n3, NULL);
  double *f = &atom->f[0][0];
  for (size_t k = 0; k < n3; k++) f[k] += hf_[k];
}

double FixFluidDragAmd::memory_usage() { return (double)atom->nmax * (9 * sizeof(double) + sizeof(int)); }

void FixFluidDragAmd::grow_arrays(int nmax)
{
  memory->grow(ffluiddrag, nmax, 3, "fdrag:ffluiddrag");
  memory->grow(DuDt, nmax, 3, "fdrag:DuDt");
  memory->grow(vOld, nmax, 3, "fdrag:vOld");
  memory->grow(foamCpuId, nmax, "fdrag:foamCpuId");
}

void FixFluidDragAmd::copy_arrays(int i, int j, int)
{
  for (int k = 0; k < 3; k++) {
    ffluiddrag[j][k] = ffluiddrag[i][k];
    DuDt[j][k] = DuDt[i][k];
    vOld[j][k] = vOld[i][k];
  }
  foamCpuId[j] = foamCpuId[i];
}

/* the migration payload of the reference (:211-243): ffluiddrag, foamCpuId, DuDt, vOld */
int FixFluidDragAmd::pack_exchange(int i, double *buf)
{
  int m = 0;
  for (int k = 0; k < 3; k++) buf[m++] = ffluiddrag[i][k];
  buf[m++] = foamCpuId[i];
  for (int k = 0; k < 3; k++) buf[m++] = DuDt[i][k];
  for (int k = 0; k < 3; k++) buf[m++] = vOld[i][k];
  return m;
}

int FixFluidDragAmd::unpack_exchange(int nlocal, double *buf)
{
  int m = 0;
  for (int k = 0; k < 3; k++) ffluiddrag[nlocal][k] = buf[m++];
  foamCpuId[nlocal] = (int)buf[m++];
  for (int k = 0; k < 3; k++) DuDt[nlocal][k] = buf[m++];
  for (int k = 0; k < 3; k++) vOld[nlocal][k] = buf[m++];
  return m;
}
