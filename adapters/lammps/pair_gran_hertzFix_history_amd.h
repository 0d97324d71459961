/* Replaces interfaceToLammps/pair_gran_hertzFix_history.{h,cpp}: same style name, same base class, same script
 * syntax; compute() runs on the GPU through sfk_pair_gran_history_compute. */
#ifdef PAIR_CLASS

PairStyle(gran/hertzFix/history,PairGranHertzFixHistoryAmd)

#else

#ifndef LMP_PAIR_GRAN_HERTZFIX_HISTORY_AMD_H
#define LMP_PAIR_GRAN_HERTZFIX_HISTORY_AMD_H

#include <vector>

#include "amd_device.h"
#include "pair_gran_hooke_history.h"

namespace LAMMPS_NS {

class PairGranHertzFixHistoryAmd : public PairGranHookeHistory {
 public:
  PairGranHertzFixHistoryAmd(class LAMMPS *);
  virtual ~PairGranHertzFixHistoryAmd() {}
  virtual void compute(int, int);
  void settings(int, char **);

 private:
  void flatten_list();       // NeighList pages -> CSR on the device (after every neighbour build)
  sfk_gran_params gp_;
  int nrows_, npairs_;
  std::vector<int> ilist_, first_, jlist_, touch_;
  std::vector<double> shear_, hf_, ht_;
  sedifoam_amd::DevBuf d_ilist_, d_first_, d_jlist_, d_touch_, d_shear_, d_x_, d_v_, d_omega_, d_radius_, d_rmass_,
      d_mask_, d_f_, d_torque_, d_mass_rigid_;
};

}

#endif
#endif
