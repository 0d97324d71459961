#ifndef LMP_NEIGH_REQUEST_H
#define LMP_NEIGH_REQUEST_H
#include "pointers.h"
namespace LAMMPS_NS {
class NeighRequest : protected Pointers {
 public:
  int pair, fix, half, full, gran, granhistory;
  NeighRequest(LAMMPS *l) : Pointers(l) {}
};
}
#endif
