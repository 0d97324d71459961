#ifndef LMP_ATOM_H
#define LMP_ATOM_H
#include "pointers.h"
namespace LAMMPS_NS {
class Atom : protected Pointers {
 public:
  int nlocal, nghost, nmax;
  double **x, **v, **f, **omega, **torque;
  double *radius, *rmass;
  int *mask, *type, *tag;
  Atom(LAMMPS *l) : Pointers(l) {}
  void add_callback(int);
  void delete_callback(const char *, int);
};
}
#endif
