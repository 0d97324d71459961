/* Test stand-in for an Open MPI-family <mpi.h>: MPI_Comm is a pointer to an opaque struct.  See fake_mpi_int/mpi.h. */
#ifndef FAKE_MPI_PTR_H
#define FAKE_MPI_PTR_H
#include <stdlib.h>
#define MPI_VERSION 3
struct fake_ompi_communicator_t { int id; };
typedef struct fake_ompi_communicator_t *MPI_Comm;
static struct fake_ompi_communicator_t fake_ompi_mpi_comm_world = {0}, fake_ompi_dup = {1};
#define MPI_COMM_WORLD (&fake_ompi_mpi_comm_world)
#define MPI_SUCCESS 0
static inline int MPI_Comm_dup(MPI_Comm in, MPI_Comm *out) { (void)in; *out = &fake_ompi_dup; return MPI_SUCCESS; }
static inline int MPI_Abort(MPI_Comm c, int code) { (void)c; exit(code ? code : 1); return 0; }
#endif
