#ifndef LMP_COMM_H
#define LMP_COMM_H
#include "pointers.h"
namespace LAMMPS_NS {
class Comm : protected Pointers {
 public:
  Comm(LAMMPS *l) : Pointers(l) {}
  void forward_comm_pair(class Pair *);
};
}
#endif
