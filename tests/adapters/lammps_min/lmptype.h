#ifndef LMP_LMPTYPE_H
#define LMP_LMPTYPE_H
#include <stdint.h>
#define FLERR __FILE__,__LINE__
#define NEIGHMASK 0x3FFFFFFF   /* [3P] lmptype.h: the two top bits of a neighbour word are special-bond flags */
namespace LAMMPS_NS { typedef int64_t bigint; }
#endif
