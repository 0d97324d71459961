#ifndef LMP_FORCE_H
#define LMP_FORCE_H
#include "pointers.h"
namespace LAMMPS_NS {
class Force : protected Pointers {
 public:
  double nktv2p, vxmu2f; int newton_pair;
  Force(LAMMPS *l) : Pointers(l) {}
  double numeric(const char *, int, char *);
  int inumeric(const char *, int, char *);
};
}
#endif
