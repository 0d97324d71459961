/* Test stand-in for an MPICH-family <mpi.h>: MPI_Comm is an int handle.  Only what the LAMMPS-object call sequence of
 * lammpsFoam/softParticleCloud.C:57-62 needs (MPI_Comm_dup of MPI_COMM_WORLD, MPI_Abort) and the one-process forms of the calls the shim makes on N ranks; no MPI library is linked. */
#ifndef FAKE_MPI_INT_H
#define FAKE_MPI_INT_H
#include <stdlib.h>
#define MPI_VERSION 3
typedef int MPI_Comm;
#define MPI_COMM_WORLD ((MPI_Comm)0x44000000)
#define MPI_SUCCESS 0
static inline int MPI_Comm_dup(MPI_Comm in, MPI_Comm *out) { *out = in + 1; return MPI_SUCCESS; }
static inline int MPI_Abort(MPI_Comm c, int code) { (void)c; exit(code ? code : 1); return 0; }
/* a one-process world: what the LAMMPS-object shim and tests/c_abi/shim_driver.cpp call on N ranks, for N = 1 */
#include <string.h>
typedef int MPI_Datatype;   /* = the size of one element */
typedef int MPI_Op;
#define MPI_CHAR 1
#define MPI_INT ((MPI_Datatype)sizeof(int))
#define MPI_DOUBLE ((MPI_Datatype)sizeof(double))
#define MPI_SUM 0
#define MPI_MAX 1
static inline int MPI_Initialized(int *flag) { *flag = 1; return MPI_SUCCESS; }
static inline int MPI_Comm_rank(MPI_Comm c, int *r) { (void)c; *r = 0; return MPI_SUCCESS; }
static inline int MPI_Comm_size(MPI_Comm c, int *n) { (void)c; *n = 1; return MPI_SUCCESS; }
static inline int MPI_Barrier(MPI_Comm c) { (void)c; return MPI_SUCCESS; }
static inline int MPI_Bcast(void *b, int n, MPI_Datatype t, int root, MPI_Comm c)
{ (void)b; (void)n; (void)t; (void)root; (void)c; return MPI_SUCCESS; }
static inline int MPI_Allreduce(const void *in, void *out, int n, MPI_Datatype t, MPI_Op op, MPI_Comm c)
{ (void)op; (void)c; memcpy(out, in, (size_t)n * (size_t)t); return MPI_SUCCESS; }
#endif
