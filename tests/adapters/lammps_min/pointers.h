#ifndef LMP_POINTERS_H
#define LMP_POINTERS_H
#include <cstddef>
#include "lmptype.h"
namespace LAMMPS_NS {
class Atom; class Update; class Force; class Neighbor; class Memory; class Error; class Modify; class Comm;
class LAMMPS {
 public:
  Atom *atom; Update *update; Force *force; Neighbor *neighbor; Memory *memory; Error *error; Modify *modify; Comm *comm;
};
class Pointers {
 public:
  Pointers(LAMMPS *ptr) : lmp(ptr), memory(ptr->memory), error(ptr->error), atom(ptr->atom), update(ptr->update),
                          force(ptr->force), neighbor(ptr->neighbor), modify(ptr->modify), comm(ptr->comm) {}
  virtual ~Pointers() {}
 protected:
  LAMMPS *lmp; Memory *&memory; Error *&error; Atom *&atom; Update *&update; Force *&force; Neighbor *&neighbor;
  Modify *&modify; Comm *&comm;
};
}
#endif
