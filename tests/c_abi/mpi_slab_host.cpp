// An MPI / C++ host that drives N x-slabs of one bed through the C-ABI alone -- the calls a lammpsFoam built against
// libsedifoam_amd.so makes on a decomposed case: every MPI rank opens one engine (`lammps_open(0, NULL, comm, &ptr)`
// shape), feeds it the input-script lines, owns the atoms of its slab, then
//     sf_dem_comm_unique_id (rank 0) -> MPI_Bcast -> sf_slab_init -> sf_slab_setup -> sf_slab_step(n) ...
// No Python, no torch: the sub-step loop, the forward halo, the rebuild vote, migration and the border exchange all
// run inside the library (csrc/sf_halo_rccl.hip).  Rank 0 also runs the WHOLE bed on a second, single-domain engine;
// the slabs' atoms are gathered by tag with MPI_Gatherv and compared with it.
//
// Arguments 3..5 (px py pz, px py pz = ranks): a 3-D processor grid instead of slabs -- sf_brick_init, every rank the
// atoms of its brick (the bed then has ncz = 8 cells along z so that two bricks fit), everything else the same calls.
//
// Built and run by tests/test_halo_gpu.py (g++ against the image's MPICH, `mpirun -np 2|3`).  On a one-GPU box the
// ranks share the GPU and SF_RCCL_LIB points the library at tests/c_abi/standin_rccl.cpp instead of librccl (which
// refuses two ranks on one device); on a multi-GPU node the same binary runs on librccl with one GPU per rank
// (HIP_VISIBLE_DEVICES per rank).  Prints "OK ranks <N> atoms <n> rebuilds <r> max|dx| <e> rel(v) <e>" or "FAIL ...".
#include "mpi_host_common.h"

int main(int argc, char** argv)
{
  MPI_Init(&argc, &argv);
  int rank = 0, world = 1;
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  MPI_Comm_size(MPI_COMM_WORLD, &world);
  const int ncx = argc > 1 ? std::atoi(argv[1]) : 8;
  const int nsteps = argc > 2 ? std::atoi(argv[2]) : 50;
  const int nruns = 2;
  int P[3] = {world, 1, 1};
  const bool bricks = argc > 5;
  if (bricks)
    for (int k = 0; k < 3; k++) P[k] = std::atoi(argv[3 + k]);
  if (P[0] * P[1] * P[2] != world) {
    std::printf("FAIL %d x %d x %d bricks for %d ranks\n", P[0], P[1], P[2], world);
    MPI_Abort(MPI_COMM_WORLD, 1);
  }
  const Bed bed = make_bed(ncx, 5, bricks ? 8 : 5, 0.5);
  const int n = (int)bed.tag.size();
  const double L = bed.hi[0] - bed.lo[0];

  // this rank's slab / brick: coordinate k in [lo + c_k w_k, lo + (c_k + 1) w_k), rank = cx + px (cy + py cz)
  std::vector<int> mine;
  for (int i = 0; i < n; i++) {
    int c[3];
    for (int k = 0; k < 3; k++) {
      const double w = (bed.hi[k] - bed.lo[k]) / P[k];
      c[k] = (int)std::floor((bed.x[3 * i + k] - bed.lo[k]) / w);
      c[k] = c[k] < 0 ? 0 : (c[k] >= P[k] ? P[k] - 1 : c[k]);
    }
    if (c[0] + P[0] * (c[1] + P[1] * c[2]) == rank) mine.push_back(i);
  }
  MPI_Comm comm;
  MPI_Comm_dup(MPI_COMM_WORLD, &comm);
  void* slab = make_engine(bed, mine, comm);
  char id[128];
  if (rank == 0) CHECK(sf_dem_comm_unique_id(id));
  MPI_Bcast(id, 128, MPI_CHAR, 0, MPI_COMM_WORLD);
  if (bricks) CHECK(sf_brick_init(slab, id, rank, world, P[0], P[1], P[2]));
  else CHECK(sf_slab_init(slab, id, rank, world, bed.lo[0], bed.hi[0], 1));
  CHECK(sf_slab_setup(slab));
  for (int r = 0; r < nruns; r++) CHECK(sf_slab_step(slab, nsteps));
  const long long rebuilds = sf_slab_rebuild_count(slab);

  std::vector<int> tag;
  std::vector<double> x, v;
  fetch(slab, tag, x, v);
  int nl = (int)tag.size();
  std::vector<int> counts(world), displs(world), counts3(world), displs3(world);
  MPI_Gather(&nl, 1, MPI_INT, counts.data(), 1, MPI_INT, 0, MPI_COMM_WORLD);
  int total = 0;
  for (int r = 0; r < world; r++) {
    displs[r] = total;
    displs3[r] = 3 * total;
    counts3[r] = 3 * counts[r];
    total += counts[r];
  }
  std::vector<int> gtag(rank == 0 ? total : 1);
  std::vector<double> gx(rank == 0 ? 3 * (size_t)total : 1), gv(rank == 0 ? 3 * (size_t)total : 1);
  MPI_Gatherv(tag.data(), nl, MPI_INT, gtag.data(), counts.data(), displs.data(), MPI_INT, 0, MPI_COMM_WORLD);
  MPI_Gatherv(x.data(), 3 * nl, MPI_DOUBLE, gx.data(), counts3.data(), displs3.data(), MPI_DOUBLE, 0, MPI_COMM_WORLD);
  MPI_Gatherv(v.data(), 3 * nl, MPI_DOUBLE, gv.data(), counts3.data(), displs3.data(), MPI_DOUBLE, 0, MPI_COMM_WORLD);

  int fail = 0;
  if (rank == 0) {
    // the same bed on one engine
    std::vector<int> all(n);
    for (int i = 0; i < n; i++) all[i] = i;
    void* one = make_engine(bed, all, comm);
    CHECK(sf_dem_setup(one));
    for (int r = 0; r < nruns; r++) CHECK(sf_lammps_step(one, nsteps));
    std::vector<int> rt;
    std::vector<double> rx, rv;
    fetch(one, rt, rx, rv);
    std::vector<int> where(n + 1, -1);
    for (int i = 0; i < (int)rt.size(); i++) where[rt[i]] = i;
    double ex = 0.0, ev = 0.0, sv = 0.0;
    std::vector<char> seen(n + 1, 0);
    if (total != n) fail = 1;
    for (int k = 0; k < total && !fail; k++) {
      const int t = gtag[k];
      if (t < 1 || t > n || seen[t] || where[t] < 0) {
        fail = 1;
        break;
      }
      seen[t] = 1;
      const int i = where[t];
      for (int c = 0; c < 3; c++) {
        double dx = gx[3 * k + c] - rx[3 * i + c];
        if (c == 0) dx -= L * std::round(dx / L);   // (an atom that crossed the periodic face is wrapped at a rebuild)
        if (c == 2) dx -= (bed.hi[2] - bed.lo[2]) * std::round(dx / (bed.hi[2] - bed.lo[2]));
        ex = std::fmax(ex, std::fabs(dx));
        ev = std::fmax(ev, std::fabs(gv[3 * k + c] - rv[3 * i + c]));
        sv = std::fmax(sv, std::fabs(rv[3 * i + c]));
      }
    }
    if (fail) std::printf("FAIL the slabs hold %d atoms, the bed %d (or a tag is missing / doubled)\n", total, n);
    else if (!(ex <= 1e-12) || !(ev <= 1e-9 * sv) || rebuilds < 3) {
      std::printf("FAIL ranks %d atoms %d rebuilds %lld max|dx| %.3e rel(v) %.3e\n", world, n, rebuilds, ex, ev / sv);
      fail = 1;
    } else {
      std::printf("OK ranks %d atoms %d rebuilds %lld max|dx| %.3e rel(v) %.3e\n", world, n, rebuilds, ex, ev / sv);
    }
    CHECK(sf_lammps_close(one));
  }
  CHECK(sf_lammps_close(slab));
  MPI_Bcast(&fail, 1, MPI_INT, 0, MPI_COMM_WORLD);
  MPI_Finalize();
  return fail;
}
