/* Replaces interfaceToLammps/pair_lubricate_poly.cpp (the reference edits the stock PairLubricatePoly in place:
 * h_sep = 100 (ri + rj) below the inner cutoff, :294-297). */
#ifdef PAIR_CLASS

PairStyle(lubricate/poly,PairLubricatePolyAmd)

#else

#ifndef LMP_PAIR_LUBRICATE_POLY_AMD_H
#define LMP_PAIR_LUBRICATE_POLY_AMD_H

#include <vector>

#include "amd_device.h"
#include "pair_lubricate_poly.h"

namespace LAMMPS_NS {

class PairLubricatePolyAmd : public PairLubricatePoly {
 public:
  PairLubricatePolyAmd(class LAMMPS *lmp) : PairLubricatePoly(lmp), nrows_(-1) {}
  virtual ~PairLubricatePolyAmd() {}
  void compute(int, int);

 private:
  int nrows_;
  std::vector<int> ilist_, first_, jlist_;
  std::vector<double> hf_, ht_;
  sedifoam_amd::DevBuf d_ilist_, d_first_, d_jlist_, d_x_, d_v_, d_omega_, d_radius_, d_f_, d_torque_;
};

}

#endif
#endif
