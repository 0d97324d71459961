#ifndef LMP_LMPTYPE_H
#define LMP_LMPTYPE_H
#include <stdint.h>
#define FLERR __FILE__,__LINE__
namespace LAMMPS_NS { typedef int64_t bigint; }
#endif
