/* sedifoam_lammps_shim.h -- the way lammpsFoam OWNS LAMMPS, on top of libsedifoam_amd.so.
 *
 * lammpsFoam/softParticleCloud.C does not go through lammps_open / lammps_command: it holds a C++ object,
 *     LAMMPS* lmp_                               (softParticleCloud.H:71, `using namespace LAMMPS_NS` :57)
 *     lmp_ = new LAMMPS(0, NULL, commLammps);    (softParticleCloud.C:62, after MPI_Comm_dup :60)
 *     lmp_->input->one(line);                    (:106, every line of in.lammps)
 *     delete lmp_;                               (:357)
 * and passes that pointer as the `void*` of the patched library (interfaceToLammps/library.h:29-63) at
 * :80, :119, :131, :153, :189, :192, :212, :227, :838, :868, :893, :900, :914, :1198, :1231, :1264.
 * This header gives the same names with the same signatures -- class LAMMPS_NS::LAMMPS with a public `input`
 * whose one(const char*) executes a script line, and lammps_*(void*, ...) taking that object -- forwarding to the
 * C ABI of include/sedifoam_amd.h.  The four headers lammpsFoam/include/LammpsCollection.H pulls in after "mpi.h"
 * ("lammps.h", "input.h", "atom.h", "library.h") live next to this file and include it, so the reference's
 * sources compile unchanged with   -I$(SEDIFOAM_AMD)/include/lammps_shim -I$(SEDIFOAM_AMD)/include
 * and link with                    -L$(SEDIFOAM_AMD)/sedifoam_amd -lsedifoam_amd   (instead of -llammps).
 *
 * MPI_Comm is whatever the application's mpi.h says (an int in MPICH, a pointer in Open MPI); it is carried to
 * sf_lammps_open as an opaque intptr_t.  Errors: the reference aborts (error->all / MPI_Abort), so does the shim:
 * message to stderr, then MPI_Abort(MPI_COMM_WORLD, 1) (abort() when no mpi.h was included).
 * C++03-clean (OpenFOAM 2.3 builds without -std=c++11). */
#ifndef SEDIFOAM_LAMMPS_SHIM_H
#define SEDIFOAM_LAMMPS_SHIM_H

#ifndef __cplusplus
#error "the LAMMPS object shim is C++ (softParticleCloud.C); C callers use SEDIFOAM_AMD_LAMMPS_NAMES of sedifoam_amd.h"
#endif

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "sedifoam_amd.h"

namespace sedifoam_shim {

/* MPI_Comm -> opaque handle, for both ABI families */
template <class T>
inline intptr_t comm_handle(T* c) { return reinterpret_cast<intptr_t>(c); }
inline intptr_t comm_handle(long c) { return static_cast<intptr_t>(c); }

inline void die(const char* what)
{
  std::fprintf(stderr, "ERROR: %s: %s\n", what, sf_last_error());
#ifdef MPI_VERSION
  MPI_Abort(MPI_COMM_WORLD, 1);
#endif
  std::abort();
}

}  // namespace sedifoam_shim

namespace LAMMPS_NS {

/* [3P] LAMMPS 1Feb14 input.h: `char *one(const char *)` runs one command and returns its name */
class Input {
 public:
  explicit Input(void* handle) : handle_(handle) { name_[0] = '\0'; }
  char* one(const char* line)
  {
    if (sf_lammps_command(handle_, line)) sedifoam_shim::die(line);
    /* command name = first word of the line (NULL for blank / comment lines, like LAMMPS) */
    const char* p = line;
    while (*p == ' ' || *p == '\t') p++;
    std::size_t n = 0;
    while (p[n] && p[n] != ' ' && p[n] != '\t' && p[n] != '\n' && p[n] != '\r' && p[n] != '#' && n + 1 < sizeof(name_)) n++;
    std::memcpy(name_, p, n);
    name_[n] = '\0';
    return n ? name_ : NULL;
  }

 private:
  void* handle_;
  char name_[64];
};

class LAMMPS {
 public:
  Input* input;      /* lmp_->input->one(line), softParticleCloud.C:106 */
  void* sf_handle;   /* the engine behind it (sf_lammps_* handle) */
#ifdef MPI_VERSION
  MPI_Comm world;    /* [3P] LAMMPS::world, the communicator the object lives on (library.cpp:84 MPI_Barrier) */
#endif

  /* `mpirun -np N lammpsFoam -parallel`: the object is created on a duplicate of the world communicator
   * (softParticleCloud.C:60-62) and LAMMPS decomposes itself over its N ranks.  The engine library links no MPI;
   * the three things it needs from the application's MPI happen HERE, with the application's own mpi.h: this
   * rank, the number of ranks, and the broadcast of rank 0's RCCL communicator id.  Everything after that (the
   * bricks of `processors px py pz`, the ghost halo, the collective counts of library.cpp:94-131) runs inside the
   * library over RCCL. */
  template <class Comm>
  LAMMPS(int narg, char** arg, Comm communicator) : input(NULL), sf_handle(NULL)
  {
    int rank = 0, size = 1;
    char id[128];
    std::memset(id, 0, sizeof id);
#ifdef MPI_VERSION
    world = communicator;
    int mpi_up = 0;
    MPI_Initialized(&mpi_up);
    if (mpi_up) {
      MPI_Comm_rank(communicator, &rank);
      MPI_Comm_size(communicator, &size);
      if (size > 1) {
        if (rank == 0 && sf_dem_comm_unique_id(id) != 0) sedifoam_shim::die("LAMMPS::LAMMPS (RCCL id)");
        MPI_Bcast(id, 128, MPI_CHAR, 0, communicator);
      }
    }
#endif
    if (sf_lammps_open_world(narg, arg, sedifoam_shim::comm_handle(communicator), rank, size, id, &sf_handle) != 0 ||
        !sf_handle)
      sedifoam_shim::die("LAMMPS::LAMMPS");
    input = new Input(sf_handle);
  }
  ~LAMMPS()
  {
    delete input;
    if (sf_handle) sf_lammps_close(sf_handle);
  }

 private:
  LAMMPS(const LAMMPS&);
  LAMMPS& operator=(const LAMMPS&);
};

}  // namespace LAMMPS_NS

/* ---- interfaceToLammps/library.h:29-63 with the reference's spellings; ptr is the LAMMPS object ---- */
namespace sedifoam_shim {
inline void* h(void* ptr) { return static_cast<LAMMPS_NS::LAMMPS*>(ptr)->sf_handle; }
}

template <class Comm>
inline void lammps_open(int argc, char** argv, Comm communicator, void** ptr)   /* library.cpp:40-45 */
{
  *ptr = static_cast<void*>(new LAMMPS_NS::LAMMPS(argc, argv, communicator));
}
inline void lammps_close(void* ptr) { delete static_cast<LAMMPS_NS::LAMMPS*>(ptr); }   /* library.cpp:52-56 */
inline void lammps_file(void* ptr, char* str)
{
  if (sf_lammps_file(sedifoam_shim::h(ptr), str) != 0) sedifoam_shim::die("lammps_file");
}
inline char* lammps_command(void* ptr, char* str) { return static_cast<LAMMPS_NS::LAMMPS*>(ptr)->input->one(str); }
inline void lammps_sync(void* ptr)   /* library.cpp:80-85: MPI_Barrier(lammps->world) */
{
  sf_lammps_sync(sedifoam_shim::h(ptr));
#ifdef MPI_VERSION
  int mpi_up = 0;
  MPI_Initialized(&mpi_up);
  if (mpi_up) MPI_Barrier(static_cast<LAMMPS_NS::LAMMPS*>(ptr)->world);
#endif
}
inline int lammps_get_global_n(void* ptr) { return sf_lammps_get_global_n(sedifoam_shim::h(ptr)); }
inline void lammps_get_initial_np(void* ptr, int* np_) { sf_lammps_get_initial_np(sedifoam_shim::h(ptr), np_); }
inline void lammps_get_initial_info(void* ptr, double* coords, double* velos, double* diam, double* rho_, int* tag_,
                                    int* lmpCpuId_, int* type_)
{
  if (sf_lammps_get_initial_info(sedifoam_shim::h(ptr), coords, velos, diam, rho_, tag_, lmpCpuId_, type_) != 0)
    sedifoam_shim::die("lammps_get_initial_info");
}
inline int lammps_get_local_n(void* ptr) { return sf_lammps_get_local_n(sedifoam_shim::h(ptr)); }
inline void lammps_get_local_domain(void* ptr, double* domain_)
{
  sf_lammps_get_local_domain(sedifoam_shim::h(ptr), domain_);
}
inline void lammps_get_local_info(void* ptr, double* coords, double* velos_, int* foamCpuId_, int* lmpCpuId_,
                                  int* tag_)
{
  if (sf_lammps_get_local_info(sedifoam_shim::h(ptr), coords, velos_, foamCpuId_, lmpCpuId_, tag_) != 0)
    sedifoam_shim::die("lammps_get_local_info");
}
inline void lammps_put_local_info(void* ptr, int nLocalIn, double* fdrag, double* DuDt, int* foamCpuIdIn,
                                  int* tagIn)
{
  if (sf_lammps_put_local_info(sedifoam_shim::h(ptr), nLocalIn, fdrag, DuDt, foamCpuIdIn, tagIn) != 0)
    sedifoam_shim::die("lammps_put_local_info");
}
inline void lammps_step(void* ptr, int n)
{
  if (sf_lammps_step(sedifoam_shim::h(ptr), n) != 0) sedifoam_shim::die("lammps_step");
}
inline void lammps_set_timestep(void* ptr, double dt_i) { sf_lammps_set_timestep(sedifoam_shim::h(ptr), dt_i); }
inline double lammps_get_timestep(void* ptr) { return sf_lammps_get_timestep(sedifoam_shim::h(ptr)); }
inline void lammps_create_particle(void* ptr, int npAdd, double* position, double* tag, double diameter,
                                   double rho, int type, double* vel)
{
  if (sf_lammps_create_particle(sedifoam_shim::h(ptr), npAdd, position, tag, diameter, rho, type, vel) != 0)
    sedifoam_shim::die("lammps_create_particle");
}
inline void lammps_delete_particle(void* ptr, int* deleteList, int nDelete)
{
  if (sf_lammps_delete_particle(sedifoam_shim::h(ptr), deleteList, nDelete) != 0)
    sedifoam_shim::die("lammps_delete_particle");
}

#endif
