/* Replaces interfaceToLammps/fix_cohesive.{h,cpp} (post_force path; the `compute cohe/local` output of the reference
 * loops forever for opt 0, fix_cohesive.cpp:394, and is not provided). */
#ifdef FIX_CLASS

FixStyle(cohesive,FixCoheAmd)

#else

#ifndef LMP_FIX_COHESIVE_AMD_H
#define LMP_FIX_COHESIVE_AMD_H

#include <vector>

#include "amd_device.h"
#include "fix.h"

namespace LAMMPS_NS {

class FixCoheAmd : public Fix {
 public:
  FixCoheAmd(class LAMMPS *, int, char **);
  int setmask();
  void init();
  void init_list(int, class NeighList *);
  void setup();              // the reference's signature (fix_cohesive.cpp:117): never called by Modify::setup(int),
                             // so cohesion is not applied during setup -- kept
  void post_force(int);

 private:
  double ah, lam, smin, smax;
  int opt;
  class NeighList *list;
  int nrows_;
  std::vector<int> ilist_, first_, jlist_;
  std::vector<double> hf_;
  sedifoam_amd::DevBuf d_ilist_, d_first_, d_jlist_, d_x_, d_radius_, d_mask_, d_f_;
};

}

#endif
#endif
