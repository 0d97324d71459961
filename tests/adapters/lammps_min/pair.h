#ifndef LMP_PAIR_H
#define LMP_PAIR_H
#include "pointers.h"
namespace LAMMPS_NS {
class Pair : protected Pointers {
 public:
  Pair(LAMMPS *l) : Pointers(l) {}
  virtual ~Pair() {}
  virtual void compute(int, int) = 0;
  virtual void settings(int, char **) = 0;
 protected:
  int evflag, vflag_fdotr;
  class NeighList *list;
  void ev_setup(int, int);
};
}
#define PairStyle(key,Class)
#endif
