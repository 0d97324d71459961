#ifndef LMP_NEIGHBOR_H
#define LMP_NEIGHBOR_H
#include "pointers.h"
namespace LAMMPS_NS {
class Neighbor : protected Pointers {
 public:
  int ago; class NeighRequest **requests;
  Neighbor(LAMMPS *l) : Pointers(l) {}
  int request(void *);
};
}
#endif
