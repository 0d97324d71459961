#ifndef LMP_PAIR_GRAN_HOOKE_HISTORY_H
#define LMP_PAIR_GRAN_HOOKE_HISTORY_H
#include "pair.h"
namespace LAMMPS_NS {
class PairGranHookeHistory : public Pair {
 public:
  int computeflag;
  PairGranHookeHistory(class LAMMPS *l) : Pair(l) {}
  virtual ~PairGranHookeHistory() {}
  virtual void compute(int, int);
  virtual void settings(int, char **);
 protected:
  double kn, kt, gamman, gammat, xmu; int dampflag; double dt; int freeze_group_bit;
  class NeighList *listgranhistory;
  class Fix *fix_rigid;      // storage of rigid body masses for use in granular interactions ([3P] pair_gran_hooke_history.h)
  double *mass_rigid;
  int nmax;
};
}
#endif
