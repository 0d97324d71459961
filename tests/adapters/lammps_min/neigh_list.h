#ifndef LMP_NEIGH_LIST_H
#define LMP_NEIGH_LIST_H
#include "pointers.h"
namespace LAMMPS_NS {
class NeighList : protected Pointers {
 public:
  int inum; int *ilist, *numneigh; int **firstneigh; double **firstdouble;
  NeighList *listgranhistory;
  NeighList(LAMMPS *l) : Pointers(l) {}
};
}
#endif
