#ifndef LMP_PAIR_LUBRICATE_POLY_H
#define LMP_PAIR_LUBRICATE_POLY_H
#include "pair.h"
namespace LAMMPS_NS {
class PairLubricate : public Pair {
 public:
  PairLubricate(class LAMMPS *l) : Pair(l) {}
  virtual void compute(int, int);
  virtual void settings(int, char **);
 protected:
  double mu, cut_inner_global, cut_global, R0, RT0, RS0; int flaglog, flagfld, flagHI, flagVF;
};
class PairLubricatePoly : public PairLubricate {
 public:
  PairLubricatePoly(class LAMMPS *l) : PairLubricate(l) {}
  virtual void compute(int, int);
};
}
#endif
