#ifndef LMP_UPDATE_H
#define LMP_UPDATE_H
#include "pointers.h"
namespace LAMMPS_NS {
class Update : protected Pointers {
 public:
  double dt; bigint ntimestep; int setupflag; char *integrate_style;
  Update(LAMMPS *l) : Pointers(l) {}
};
}
#endif
