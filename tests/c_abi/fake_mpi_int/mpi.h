/* Test stand-in for an MPICH-family <mpi.h>: MPI_Comm is an int handle.  Only what the LAMMPS-object call sequence of
 * lammpsFoam/softParticleCloud.C:57-62 needs (MPI_Comm_dup of MPI_COMM_WORLD, MPI_Abort); no MPI library is linked. */
#ifndef FAKE_MPI_INT_H
#define FAKE_MPI_INT_H
#include <stdlib.h>
#define MPI_VERSION 3
typedef int MPI_Comm;
#define MPI_COMM_WORLD ((MPI_Comm)0x44000000)
#define MPI_SUCCESS 0
static inline int MPI_Comm_dup(MPI_Comm in, MPI_Comm *out) { *out = in + 1; return MPI_SUCCESS; }
static inline int MPI_Abort(MPI_Comm c, int code) { (void)c; exit(code ? code : 1); return 0; }
#endif
