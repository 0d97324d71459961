/* Replaces interfaceToLammps/fix_fluid_drag.{h,cpp}: same style name and the same public per-atom arrays
 * (ffluiddrag, DuDt, vOld, foamCpuId are poked by library.cpp:259-367); post_force() runs on the GPU. */
#ifdef FIX_CLASS

FixStyle(fdrag,FixFluidDragAmd)

#else

#ifndef LMP_FIX_FLUID_DRAG_AMD_H
#define LMP_FIX_FLUID_DRAG_AMD_H

#include <vector>

#include "amd_device.h"
#include "fix.h"

namespace LAMMPS_NS {

class FixFluidDragAmd : public Fix {
 public:
  double **ffluiddrag;   // fix_fluid_drag.h:30-33: written by lammps_put_local_info
  double **DuDt;
  double **vOld;
  int *foamCpuId;

  FixFluidDragAmd(class LAMMPS *, int, char **);
  ~FixFluidDragAmd();
  int setmask();
  void init();
  void setup(int);
  virtual void post_force(int);
  double memory_usage();
  void grow_arrays(int);
  void copy_arrays(int, int, int);
  int pack_exchange(int, double *);
  int unpack_exchange(int, double *);

 private:
  double carrier_rho;
  std::vector<double> hf_;
  sedifoam_amd::DevBuf d_v_, d_rmass_, d_radius_, d_mask_, d_fd_, d_dudt_, d_vold_, d_f_;
};

}

#endif
#endif
