// sf_smooth.h -- diffusion-based coarse graining of cell fields (SURVEY "next" row N1):
//   enhancedCloud::smoothField  lammpsFoam/enhancedCloud.C:790-907, set up at :564-583
// The reference integrates d(phi)/dt = div(D grad phi) from 0 to b^2/4 (b = diffusionBandWidth) in
// `diffusionSteps` implicit-Euler steps with zero-gradient boundaries, each step one
// `solve(fvm::ddt(phi) - fvm::laplacian(DT, phi))` by PCG (tolerance 1e-10, system/fvSolution: tempDiffScalar).
// On the uniform hex block that is (I - dtau*L) phi_new = phi_old with the 7-point Laplacian; here it is solved
// directly in the cosine-transform basis that diagonalises it (all steps at once, sf_smooth.hip; on a graded axis of a
// blockMesh simpleGrading block in the eigenbasis of that axis' finite-volume operator instead), or, on meshes
// too wide for dense transforms, matrix-free: with the Chebyshev semi-iteration (spectrum of A known in closed form, no
// inner products, fixed iteration count, nothing for the host to check), optionally by conjugate gradients
// (SF_SMOOTH_CG=1: deterministic two-stage reductions, the host looks at the residual every 8 iterations).
#pragma once
#include "sf_common.h"

namespace sf {

class DiffusionSmoother {
 public:
  DiffusionSmoother() = default;
  ~DiffusionSmoother();
  // n, dx: the hex block; D: diagonal of smoothDirection; band: diffusionBandWidth; steps: diffusionSteps
  // widths[k]: the n[k] cell widths of a graded axis (blockMesh simpleGrading), or nullptr = uniform dx[k]
  // periodic[k] != 0: cyclic patch pair along axis k (first and last cell are neighbours) instead of zero gradient
  void configure(const int n[3], const double dx[3], const double D[3], double band, int steps, hipStream_t s,
                 const double* const widths[3] = nullptr, const int* periodic = nullptr);
  bool enabled() const { return enabled_; }
  // field: ncells*ncomp doubles, component-interleaved (AoS); smoothed in place
  void smooth(double* field, int ncomp);
  // two fields through the same launches (e.g. gamma [n] and Ue [n][3]); nb = 0: one field
  void smooth2(double* fa, int na, double* fb, int nb);
  long long iterations() const { return iters_; }

  // ---- mesh partitioned into x-slabs (one slab per GPU): the local block holds this rank's nx_local cell layers plus
  // one ghost layer on each side; the implicit diffusion solve is global, so the direct solver runs as
  //   begin()  : forward transforms along y and z of the local block (no communication)
  //   caller   : transposes the planar work array to complete x-lines (an all-to-all of local size), calls
  //   xsolve() : forward x transform, filter (1 + lx + ly + lz)^-steps, inverse x transform on those lines,
  //              and transposes back (every rank also receives the two layers next to its slab: the ghost layers)
  //   end()    : inverse transforms along z and y into the fields
  // nx_global: cells of the whole mesh along x (uniform dx; periodic[0] = cyclic pair of the WHOLE mesh)
  void configure_slab(int nx_global);
  bool slab() const { return nx_global_ > 0; }
  void begin(double* fa, int na, double* fb, int nb);
  void end();
  double* work() const;            // planar [ntot][nz][ny][nx_local + 2]
  int work_fields() const { return slab_ntot_; }
  void xsolve(double* lines, long long nlines, long long first_line);   // lines: [nlines][nx_global]

 private:
  void solve_component(double* x, int stride);
  bool enabled_ = false;
  int n_[3] = {0, 0, 0};
  int per_[3] = {0, 0, 0};
  int ncells_ = 0, steps_ = 0, nblocks_ = 0;
  double c_[3] = {0, 0, 0};   // dtau * D_d / dx_d^2
  hipStream_t s_ = nullptr;
  double *r_ = nullptr, *p_ = nullptr, *ap_ = nullptr, *partial_ = nullptr, *scal_ = nullptr;
  double* h_scal_ = nullptr;  // pinned
  long long iters_ = 0;
  // Chebyshev iteration (default; SF_SMOOTH_CG=1 selects conjugate gradients)
  static constexpr int kMaxCheb = 4;
  bool use_cg_ = false;
  double lmin_ = 1.0, lmax_ = 1.0;
  int cheb_iters_ = 0;
  double* cheb_ = nullptr;    // r, d (two buffers): 3 x [kMaxCheb][ncells]
  // spectral direct solve (default up to kMaxSpectral cells per direction; SF_SMOOTH_SPECTRAL=0: Chebyshev)
  void smooth_spectral(double* fa, int na, double* fb, int nb);
  bool use_spectral_ = false;
  double* spec_ = nullptr;    // DCT matrices, eigenvalues, two planar work arrays
  size_t specC_off_[3] = {0, 0, 0}, specB_off_[3] = {0, 0, 0}, specL_off_[3] = {0, 0, 0}, spec_work_off_ = 0;
  static constexpr int kMaxSpectral = 256;   // cells per direction the dense transforms are used for
  int nx_global_ = 0;
  double* slab_x_ = nullptr;      // global x direction: forward matrix, eigenvalues, a line buffer
  size_t slab_tmp_cap_ = 0;
  double* slab_tmp_ = nullptr;
  double *slab_fa_ = nullptr, *slab_fb_ = nullptr;
  int slab_na_ = 0, slab_nb_ = 0, slab_ntot_ = 0;
  bool slab_open_ = false;
};

}  // namespace sf
