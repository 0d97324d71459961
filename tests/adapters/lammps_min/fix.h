#ifndef LMP_FIX_H
#define LMP_FIX_H
#include "pointers.h"
namespace LAMMPS_NS {
class Fix : protected Pointers {
 public:
  char *id, *style; int igroup, groupbit; int force_reneighbor; bigint next_reneighbor;
  Fix(class LAMMPS *l, int, char **) : Pointers(l) {}
  virtual ~Fix() {}
  virtual int setmask() = 0;
  virtual void init() {}
  virtual void init_list(int, class NeighList *) {}
  virtual void setup(int) {}
  virtual void post_force(int) {}
  virtual double memory_usage() { return 0.0; }
  virtual void *extract(const char *, int &) { return NULL; }
  virtual void grow_arrays(int) {}
  virtual void copy_arrays(int, int, int) {}
  virtual int pack_exchange(int, double *) { return 0; }
  virtual int unpack_exchange(int, double *) { return 0; }
};
namespace FixConst {
  static const int POST_FORCE = 1 << 5, POST_FORCE_RESPA = 1 << 12, MIN_POST_FORCE = 1 << 16;
}
}
#define FixStyle(key,Class)
#endif
