#ifndef LMP_MEMORY_H
#define LMP_MEMORY_H
#include "pointers.h"
namespace LAMMPS_NS {
class Memory : protected Pointers {
 public:
  Memory(LAMMPS *l) : Pointers(l) {}
  template <typename T> T *create(T *&array, int n, const char *name);
  template <typename T> T *grow(T *&array, int n, const char *name);
  template <typename T> T **grow(T **&array, int n1, int n2, const char *name);
  template <typename T> void destroy(T *&array);
  template <typename T> void destroy(T **&array);
};
}
#endif
