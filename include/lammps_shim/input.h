/* input.h -- stands where LAMMPS' own "input.h" stands on the include path of lammpsFoam
 * (lammpsFoam/include/LammpsCollection.H:7-11 includes "mpi.h", "lammps.h", "input.h", "atom.h", "library.h").
 * Everything the reference uses of it is in sedifoam_lammps_shim.h. */
#include "sedifoam_lammps_shim.h"
