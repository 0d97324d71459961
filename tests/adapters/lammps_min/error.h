#ifndef LMP_ERROR_H
#define LMP_ERROR_H
#include "pointers.h"
namespace LAMMPS_NS {
class Error : protected Pointers {
 public:
  Error(LAMMPS *l) : Pointers(l) {}
  void all(const char *, int, const char *);
  void one(const char *, int, const char *);
};
}
#endif
