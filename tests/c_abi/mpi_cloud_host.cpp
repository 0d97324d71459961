// The coupled step on a decomposed case from an MPI / C++ host -- what a `lammpsFoam -parallel` built against
// libsedifoam_amd.so does per CFD time step, with nothing but the C-ABI and MPI:
//     enhancedCloud::evolve()        sf_cloud_phase(0) ; per sub-cycle: sf_cloud_phase(1) [drag on this rank's particles]
//                                    -> sf_slab_step(subSteps) [DEM through the halo driver] -> first sub-cycle:
//                                    sf_cloud_phase(2) [per-cell sums of this rank's particles] -> MPI_Allreduce of
//                                    gamma and Ue -> sf_cloud_phase(3) [smoothing, Ue / gamma]
//     enhancedCloud::calcTcFields()  sf_cloud_phase(4) -> MPI_Allreduce of Asrc -> sf_cloud_phase(5)
// Every rank holds the whole (small) mesh; the per-cell sums are linear in the particles.  They travel through host
// memory here (sf_dev_download / MPI_Allreduce / sf_dev_upload); a host with GPU-aware MPI passes the device pointers
// of sf_cloud_device_fields straight to MPI_Allreduce.  Added mass and the Basset history force are on, so the
// cloud's per-particle state (previous velocity, history sums) has to follow the grains that change rank.
//
// Third argument "partition": the mesh is cut by the same planes as the particles -- every rank holds its
// nx / ranks cell layers plus a ghost layer on each side, and the exchanges are the library's own, over the engine's RCCL
// communicator: sf_cloud_slab_halo_add (ghost-layer sums to the face neighbours) and sf_cloud_slab_phase (the x-line
// all-to-all of the distributed smoothing solve); no MPI collective of the size of the mesh is left in the loop.
//
// Rank 0 also runs the whole bed on one engine with sf_cloud_evolve / sf_cloud_calc_tc_fields and compares fields and
// particles.  Built and run by tests/test_halo_gpu.py like mpi_slab_host.cpp (SF_RCCL_LIB stand-in on a one-GPU box).
// Prints "OK ranks <N> cells <n> rel(gamma) <e> rel(Ue) <e> rel(Asrc) <e> max|dx| <e>" or "FAIL ...".
#include "mpi_host_common.h"

namespace {

double rel_err(const std::vector<double>& a, const std::vector<double>& b)
{
  double e = 0.0, s = 0.0;
  for (size_t k = 0; k < a.size(); k++) {
    e = std::fmax(e, std::fabs(a[k] - b[k]));
    s = std::fmax(s, std::fabs(b[k]));
  }
  return s > 0.0 ? e / s : e;
}

struct Fields {
  std::vector<double> gamma, Ue, Asrc, Omega;
  explicit Fields(int nc) : gamma(nc), Ue(3 * (size_t)nc), Asrc(3 * (size_t)nc), Omega(nc) {}
};

// sum of a device array over the ranks, in place
void add_over_ranks(double* dev, size_t n, void* stream)
{
  std::vector<double> a(n), b(n);
  CHECK(sf_dev_download(a.data(), dev, sizeof(double) * n, stream));
  MPI_Allreduce(a.data(), b.data(), (int)n, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
  CHECK(sf_dev_upload(dev, b.data(), sizeof(double) * n, stream));
  CHECK(sf_dev_sync(stream));
}

}  // namespace

int main(int argc, char** argv)
{
  MPI_Init(&argc, &argv);
  int rank = 0, world = 1;
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  MPI_Comm_size(MPI_COMM_WORLD, &world);
  const int ncx = argc > 1 ? std::atoi(argv[1]) : 8;
  const int ncfd = argc > 2 ? std::atoi(argv[2]) : 3;
  const bool partition = argc > 3 && std::strcmp(argv[3], "partition") == 0;
  const Bed bed = make_bed(ncx, 5, 5, 0.3);
  const int n = (int)bed.tag.size();
  const double L = bed.hi[0] - bed.lo[0], w = L / world;

  sf_cloud_mesh mesh;
  std::memset(&mesh, 0, sizeof mesh);
  int nc = 1;
  for (int k = 0; k < 3; k++) {
    mesh.origin[k] = bed.lo[k];
    mesh.n[k] = (int)std::fmax(1.0, std::floor((bed.hi[k] - bed.lo[k]) / 3.0e-3));
    mesh.dx[k] = (bed.hi[k] - bed.lo[k]) / mesh.n[k];
    if (k == 0 && partition) {   // the slab planes must be cell faces
      mesh.n[0] = world * (mesh.n[0] / world > 0 ? mesh.n[0] / world : 1);
      mesh.dx[0] = (bed.hi[0] - bed.lo[0]) / mesh.n[0];
    }
    nc *= mesh.n[k];
  }
  if (partition) {
    mesh.periodic[0] = 1;   // the channel is cyclic in x and z, for the particles and for the smoothing
    mesh.periodic[2] = 1;
  }
  sf_cloud_props pr;
  std::memset(&pr, 0, sizeof pr);
  pr.dragModel = 0;   // ErgunWenYu
  pr.subCycles = 2;
  pr.particleDrag = pr.particlePressureGrad = pr.particleBuoyancy = pr.particleAddedMass = pr.particleHistoryForce = 1;
  pr.gravity[1] = -9.81;
  pr.rhob = 1000.0;
  pr.nub = 1.0e-6;
  pr.maxPossibleAlpha = 0.65;
  pr.diffusionBandWidth = 0.006;
  pr.diffusionSteps = 6;
  pr.UfSmooth = pr.UpSmooth = pr.dragSmooth = pr.alphaSmooth = 1;
  pr.smoothDirection[0] = pr.smoothDirection[1] = pr.smoothDirection[2] = 1.0;
  const double deltaT = 40.0e-6;
  std::vector<double> Uf(3 * (size_t)nc), DDtUf(3 * (size_t)nc), gradp(3 * (size_t)nc);
  for (int c = 0; c < nc; c++) {
    Uf[3 * c] = 0.01 * std::sin(0.7 * c);
    Uf[3 * c + 1] = 0.05;
    Uf[3 * c + 2] = 0.0;
    DDtUf[3 * c] = 0.0;
    DDtUf[3 * c + 1] = 0.2;
    DDtUf[3 * c + 2] = 0.0;
    gradp[3 * c] = 0.0;
    gradp[3 * c + 1] = -9810.0;
    gradp[3 * c + 2] = 0.0;
  }

  // ---- this rank's slab + its cloud ----
  std::vector<int> mine;
  for (int i = 0; i < n; i++) {
    int r = (int)std::floor((bed.x[3 * i] - bed.lo[0]) / w);
    r = r < 0 ? 0 : (r >= world ? world - 1 : r);
    if (r == rank) mine.push_back(i);
  }
  MPI_Comm comm;
  MPI_Comm_dup(MPI_COMM_WORLD, &comm);
  void* slab = make_engine(bed, mine, comm);
  char id[128];
  if (rank == 0) CHECK(sf_dem_comm_unique_id(id));
  MPI_Bcast(id, 128, MPI_CHAR, 0, MPI_COMM_WORLD);
  CHECK(sf_slab_init(slab, id, rank, world, bed.lo[0], bed.hi[0], 1));
  CHECK(sf_slab_setup(slab));
  sf_dem_device_view view;
  CHECK(sf_dem_device_view_get(slab, &view));
  void* cloud = nullptr;
  sf_cloud_mesh lmesh = mesh;
  const int nxl = mesh.n[0] / world, nxs = nxl + 2;
  if (partition) {
    lmesh.origin[0] = mesh.origin[0] + (rank * nxl - 1) * mesh.dx[0];
    lmesh.n[0] = nxs;
    lmesh.slab_nx_global = mesh.n[0];
  }
  CHECK(sf_cloud_create(slab, partition ? &lmesh : &mesh, &pr, deltaT, &cloud));
  int subCycles = 0, subSteps = 0, ncells = 0;
  CHECK(sf_cloud_sub_cycling(cloud, &subCycles, &subSteps));
  double *d_gamma = nullptr, *d_Ue = nullptr, *d_Asrc = nullptr;
  CHECK(sf_cloud_device_fields(cloud, &d_gamma, &d_Ue, &d_Asrc, &ncells));
  const int nc_local = partition ? nxs * mesh.n[1] * mesh.n[2] : nc;
  if (ncells != nc_local) {
    std::printf("FAIL cells %d %d\n", ncells, nc_local);
    MPI_Abort(MPI_COMM_WORLD, 1);
  }
  // this rank's cells (ghost layers included, wrapped at the cyclic ends) of a whole-mesh [nz][ny][nx][3] array
  auto take = [&](const std::vector<double>& g) {
    std::vector<double> l(3 * (size_t)nc_local);
    for (int iz = 0; iz < mesh.n[2]; iz++)
      for (int iy = 0; iy < mesh.n[1]; iy++)
        for (int c = 0; c < nxs; c++) {
          const int ix = ((rank * nxl - 1 + c) % mesh.n[0] + mesh.n[0]) % mesh.n[0];
          for (int k = 0; k < 3; k++)
            l[3 * ((size_t)c + nxs * (iy + (size_t)mesh.n[1] * iz)) + k] =
                g[3 * ((size_t)ix + mesh.n[0] * (iy + (size_t)mesh.n[1] * iz)) + k];
        }
    return l;
  };
  auto sphase = [&](int ph) {
    if (sf_cloud_slab_phase(cloud, ph) != 0) {
      std::printf("FAIL sf_cloud_slab_phase(%d): %s\n", ph, sf_last_error());
      MPI_Abort(MPI_COMM_WORLD, 1);
    }
  };
  auto phase = [&](int ph) {
    if (sf_cloud_phase(cloud, ph) < 0) {
      std::printf("FAIL sf_cloud_phase(%d): %s\n", ph, sf_last_error());
      MPI_Abort(MPI_COMM_WORLD, 1);
    }
  };
  Fields got(nc);
  if (partition) {
    // (the constructor scattered this rank's particles into its slab + ghost layers)
    CHECK(sf_cloud_slab_halo_add(cloud, 1 | 2));
    sphase(3);
    sphase(6);
    const std::vector<double> lU = take(Uf), lD = take(DDtUf), lG = take(gradp);
    CHECK(sf_cloud_set_fluid(cloud, lU.data(), lD.data(), lG.data(), nullptr));
    sphase(6);                                  // UfSmoothed of the initial condition (enhancedCloud.C:641-655)
    for (int step = 0; step < ncfd; step++) {
      sphase(0);                                // evolve()
      for (int k = 0; k < subCycles; k++) {
        phase(1);
        CHECK(sf_slab_step(slab, subSteps));
        if (k == 0) {
          phase(2);
          CHECK(sf_cloud_slab_halo_add(cloud, 1 | 2));
          sphase(3);
        }
      }
      phase(4);                                 // calcTcFields()
      CHECK(sf_cloud_slab_halo_add(cloud, 4));
      sphase(5);
    }
    // owned cells of every rank -> whole-mesh arrays on every rank
    Fields loc(nc_local);
    CHECK(sf_cloud_get_fields(cloud, loc.gamma.data(), loc.Ue.data(), loc.Asrc.data(), loc.Omega.data()));
    const int nown = nxl * mesh.n[1] * mesh.n[2];
    auto gather = [&](const std::vector<double>& l, int ncomp, std::vector<double>& g) {
      std::vector<double> own((size_t)nown * ncomp), all((size_t)nown * ncomp * world);
      for (int iz = 0; iz < mesh.n[2]; iz++)
        for (int iy = 0; iy < mesh.n[1]; iy++)
          for (int x = 0; x < nxl; x++)
            for (int k = 0; k < ncomp; k++)
              own[ncomp * ((size_t)x + nxl * (iy + (size_t)mesh.n[1] * iz)) + k] =
                  l[ncomp * ((size_t)(x + 1) + nxs * (iy + (size_t)mesh.n[1] * iz)) + k];
      MPI_Allgather(own.data(), nown * ncomp, MPI_DOUBLE, all.data(), nown * ncomp, MPI_DOUBLE, MPI_COMM_WORLD);
      for (int r = 0; r < world; r++)
        for (int iz = 0; iz < mesh.n[2]; iz++)
          for (int iy = 0; iy < mesh.n[1]; iy++)
            for (int x = 0; x < nxl; x++)
              for (int k = 0; k < ncomp; k++)
                g[ncomp * ((size_t)(r * nxl + x) + mesh.n[0] * (iy + (size_t)mesh.n[1] * iz)) + k] =
                    all[(size_t)r * nown * ncomp + ncomp * ((size_t)x + nxl * (iy + (size_t)mesh.n[1] * iz)) + k];
    };
    gather(loc.gamma, 1, got.gamma);
    gather(loc.Ue, 3, got.Ue);
    gather(loc.Asrc, 3, got.Asrc);
  } else {
    // (the constructor scattered this rank's particles only: redo the averaging over all ranks)
    phase(2);
    add_over_ranks(d_gamma, nc, view.stream);
    add_over_ranks(d_Ue, 3 * (size_t)nc, view.stream);
    phase(3);
    phase(6);
    CHECK(sf_cloud_set_fluid(cloud, Uf.data(), DDtUf.data(), gradp.data(), nullptr));
    for (int step = 0; step < ncfd; step++) {
      phase(0);                                   // evolve()
      for (int k = 0; k < subCycles; k++) {
        phase(1);
        CHECK(sf_slab_step(slab, subSteps));
        if (k == 0) {
          phase(2);
          add_over_ranks(d_gamma, nc, view.stream);
          add_over_ranks(d_Ue, 3 * (size_t)nc, view.stream);
          phase(3);
        }
      }
      phase(4);                                   // calcTcFields()
      add_over_ranks(d_Asrc, 3 * (size_t)nc, view.stream);
      phase(5);
    }
    CHECK(sf_cloud_get_fields(cloud, got.gamma.data(), got.Ue.data(), got.Asrc.data(), got.Omega.data()));
  }

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
    std::vector<int> all(n);
    for (int i = 0; i < n; i++) all[i] = i;
    void* one = make_engine(bed, all, comm);
    void* ref = nullptr;
    CHECK(sf_cloud_create(one, &mesh, &pr, deltaT, &ref));
    CHECK(sf_cloud_set_fluid(ref, Uf.data(), DDtUf.data(), gradp.data(), nullptr));
    for (int step = 0; step < ncfd; step++) {
      CHECK(sf_cloud_evolve(ref));
      CHECK(sf_cloud_calc_tc_fields(ref));
    }
    Fields want(nc);
    CHECK(sf_cloud_get_fields(ref, want.gamma.data(), want.Ue.data(), want.Asrc.data(), want.Omega.data()));
    std::vector<int> rt;
    std::vector<double> rx, rv;
    fetch(one, rt, rx, rv);
    std::vector<int> where(n + 1, -1);
    for (int i = 0; i < (int)rt.size(); i++) where[rt[i]] = i;
    double ex = 0.0;
    if (total != n) fail = 1;
    for (int k = 0; k < total && !fail; k++) {
      const int t = gtag[k];
      if (t < 1 || t > n || where[t] < 0) {
        fail = 1;
        break;
      }
      for (int c = 0; c < 3; c++) {
        double dx = gx[3 * k + c] - rx[3 * where[t] + c];
        if (c == 0) dx -= L * std::round(dx / L);
        ex = std::fmax(ex, std::fabs(dx));
      }
    }
    const double eg = rel_err(got.gamma, want.gamma), eu = rel_err(got.Ue, want.Ue), ea = rel_err(got.Asrc, want.Asrc);
    if (fail || !(eg <= 1e-9) || !(eu <= 1e-8) || !(ea <= 1e-8) || !(ex <= 1e-11)) {
      std::printf("FAIL ranks %d cells %d atoms %d/%d rel(gamma) %.3e rel(Ue) %.3e rel(Asrc) %.3e max|dx| %.3e\n", world,
                  nc, total, n, eg, eu, ea, ex);
      fail = 1;
    } else {
      std::printf("OK ranks %d cells %d%s rel(gamma) %.3e rel(Ue) %.3e rel(Asrc) %.3e max|dx| %.3e\n", world, nc,
                  partition ? " (mesh partitioned)" : "", eg, eu, ea, ex);
    }
    CHECK(sf_cloud_destroy(ref));
    CHECK(sf_lammps_close(one));
  }
  CHECK(sf_cloud_destroy(cloud));
  CHECK(sf_lammps_close(slab));
  MPI_Bcast(&fail, 1, MPI_INT, 0, MPI_COMM_WORLD);
  MPI_Finalize();
  return fail;
}
