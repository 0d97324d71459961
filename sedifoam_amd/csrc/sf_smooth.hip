// sf_smooth.hip -- see sf_smooth.h (enhancedCloud::smoothField, lammpsFoam/enhancedCloud.C:790-907).
#include "sf_smooth.h"
#include "sf_roctx.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

namespace sf {

struct Stencil {
  int n[3];
  double c[3];
  int ncells;
  int per[3];   // cyclic patch pair along this axis: the first and the last cell are neighbours
};

// scalars on the device: [0] rr, [1] pAp, [2] alpha, [3] beta, [4] bb (rhs norm^2), [5] rr_new
enum { S_RR = 0, S_PAP, S_ALPHA, S_BETA, S_BB, S_RRNEW, S_N = 8 };

// (A v)_c with A = I - dtau*L, zero-gradient boundaries (a missing neighbour contributes nothing)
__device__ __forceinline__ double apply_A(const Stencil& st, const double* v, int stride, int c)
{
  const int i = c % st.n[0], j = (c / st.n[0]) % st.n[1], k = c / (st.n[0] * st.n[1]);
  const double vc = v[(size_t)c * stride];
  double acc = vc;
  const int sx = 1, sy = st.n[0], sz = st.n[0] * st.n[1];
  if (i > 0) acc += st.c[0] * (vc - v[(size_t)(c - sx) * stride]);
  else if (st.per[0]) acc += st.c[0] * (vc - v[(size_t)(c + (st.n[0] - 1) * sx) * stride]);
  if (i < st.n[0] - 1) acc += st.c[0] * (vc - v[(size_t)(c + sx) * stride]);
  else if (st.per[0]) acc += st.c[0] * (vc - v[(size_t)(c - (st.n[0] - 1) * sx) * stride]);
  if (j > 0) acc += st.c[1] * (vc - v[(size_t)(c - sy) * stride]);
  else if (st.per[1]) acc += st.c[1] * (vc - v[(size_t)(c + (st.n[1] - 1) * sy) * stride]);
  if (j < st.n[1] - 1) acc += st.c[1] * (vc - v[(size_t)(c + sy) * stride]);
  else if (st.per[1]) acc += st.c[1] * (vc - v[(size_t)(c - (st.n[1] - 1) * sy) * stride]);
  if (k > 0) acc += st.c[2] * (vc - v[(size_t)(c - sz) * stride]);
  else if (st.per[2]) acc += st.c[2] * (vc - v[(size_t)(c + (st.n[2] - 1) * sz) * stride]);
  if (k < st.n[2] - 1) acc += st.c[2] * (vc - v[(size_t)(c + sz) * stride]);
  else if (st.per[2]) acc += st.c[2] * (vc - v[(size_t)(c - (st.n[2] - 1) * sz) * stride]);
  return acc;
}

// deterministic block sum: wave shuffles, then the first wave adds the per-wave results
__device__ __forceinline__ double block_sum(double v)
{
  __shared__ double ws[4];
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) ws[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) t = ws[0] + ws[1] + ws[2] + ws[3];
  __syncthreads();
  return t;
}

// x (the field, stride) holds the right-hand side b and is the initial guess: r = b - A b, p = r
__global__ __launch_bounds__(256) void k_cg_init(Stencil st, const double* x, int stride, double* r, double* p,
                                                 double* partial)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  double rr = 0.0, bb = 0.0;
  if (c < st.ncells) {
    const double b = x[(size_t)c * stride];
    const double res = b - apply_A(st, x, stride, c);
    r[c] = res;
    p[c] = res;
    rr = res * res;
    bb = b * b;
  }
  const double s1 = block_sum(rr);
  const double s2 = block_sum(bb);
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = s1;
    partial[gridDim.x + blockIdx.x] = s2;
  }
}

__global__ __launch_bounds__(256) void k_cg_ap(Stencil st, const double* p, double* ap, double* partial)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  double v = 0.0;
  if (c < st.ncells) {
    const double a = apply_A(st, p, 1, c);
    ap[c] = a;
    v = p[c] * a;
  }
  const double s = block_sum(v);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_cg_update(int ncells, const double* scal, double* x, int stride, double* r,
                                                   const double* p, const double* ap, double* partial)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const double alpha = scal[S_ALPHA];
  double v = 0.0;
  if (c < ncells) {
    x[(size_t)c * stride] += alpha * p[c];
    const double res = r[c] - alpha * ap[c];
    r[c] = res;
    v = res * res;
  }
  const double s = block_sum(v);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_cg_p(int ncells, const double* scal, const double* r, double* p)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < ncells) p[c] = r[c] + scal[S_BETA] * p[c];
}

// one workgroup adds the per-block partial sums in a fixed order and derives the CG scalars
// mode 0: init (rr, bb) ; 1: pAp -> alpha ; 2: rr_new -> beta, rr = rr_new
__global__ __launch_bounds__(256) void k_cg_scalars(int mode, int nblocks, const double* partial, double* scal)
{
  double v = 0.0, v2 = 0.0;
  for (int k = threadIdx.x; k < nblocks; k += blockDim.x) {
    v += partial[k];
    if (mode == 0) v2 += partial[nblocks + k];
  }
  const double s = block_sum(v);
  const double s2 = (mode == 0) ? block_sum(v2) : 0.0;
  if (threadIdx.x == 0) {
    if (mode == 0) {
      scal[S_RR] = s;
      scal[S_BB] = s2;
      scal[S_RRNEW] = s;
    } else if (mode == 1) {
      scal[S_PAP] = s;
      scal[S_ALPHA] = (s != 0.0) ? scal[S_RR] / s : 0.0;
    } else {
      scal[S_RRNEW] = s;
      scal[S_BETA] = (scal[S_RR] != 0.0) ? s / scal[S_RR] : 0.0;
      scal[S_RR] = s;
    }
  }
}

// ---- Chebyshev iteration: the default solver ---------------------------------------------------------------
// A = I - dtau*L is symmetric with spectrum inside [1, 1 + 4(cx+cy+cz)] (each 1-D zero-gradient second difference
// has eigenvalues in [0, 4)), so the Chebyshev semi-iteration (Saad, Iterative Methods, Alg. 12.1) needs NO inner
// products: one stencil kernel per iteration, a fixed iteration count from the condition number (residual factor
// 2 rho^k <= 1e-15), nothing for the host to look at, bitwise reproducible.  At the 32^3 meshes of the coupled
// cases conjugate gradients spends its time in launches and reductions (five launches + a host look every few
// iterations: 22 ms per CFD step for the reference's default 6 steps x 10 components); this path takes ~4 ms.
// All components of up to two fields go through the same launches (thread = cell x component).
struct ChebField {
  double* f;     // component-interleaved field
  int ncomp;
};
struct ChebArgs {
  Stencil st;
  ChebField a, b;        // b.ncomp = 0: one field
  int ntot;              // a.ncomp + b.ncomp
  double* r;             // [ntot][ncells]
  double* d0;            // [ntot][ncells] search direction (read)
  double* d1;            // [ntot][ncells] search direction (written)
  double c_rho, c_r;     // d1 = c_rho * d0 + c_r * r_new
};

__device__ __forceinline__ double* cheb_x(const ChebArgs& A, int k, int& stride)
{
  if (k < A.a.ncomp) {
    stride = A.a.ncomp;
    return A.a.f + k;
  }
  stride = A.b.ncomp;
  return A.b.f + (k - A.a.ncomp);
}

// r = b - A b with the field itself as right-hand side and initial guess; d1 = r / theta
__global__ __launch_bounds__(256) void k_cheb_init(ChebArgs A, double inv_theta)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;
  if (c >= A.st.ncells) return;
  int stride;
  const double* x = cheb_x(A, k, stride);
  const double res = x[(size_t)c * stride] - apply_A(A.st, x, stride, c);
  const size_t o = (size_t)k * A.st.ncells + c;
  A.r[o] = res;
  A.d1[o] = res * inv_theta;
}

// x += d ; r -= A d ; d_new = c_rho d + c_r r
__global__ __launch_bounds__(256) void k_cheb_iter(ChebArgs A)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;
  if (c >= A.st.ncells) return;
  int stride;
  double* x = cheb_x(A, k, stride);
  const size_t o = (size_t)k * A.st.ncells;
  const double dc = A.d0[o + c];
  const double ad = apply_A(A.st, A.d0 + o, 1, c);
  x[(size_t)c * stride] += dc;
  const double rn = A.r[o + c] - ad;
  A.r[o + c] = rn;
  A.d1[o + c] = A.c_rho * dc + A.c_r * rn;
}

// ---- spectral (cosine transform) direct solve: the default on meshes with at most 128 cells per direction --------
// The zero-gradient 7-point operator is diagonal in the DCT-II basis of each direction: its 1-D factor has
// eigenvectors cos(pi m (i + 1/2) / n) and eigenvalues 2 - 2 cos(pi m / n).  All `steps` implicit-Euler solves
// share that basis, so the WHOLE smoothing is   phi <- C^T [ (1 + lx_mx + ly_my + lz_mz)^-steps . (C phi) ]  with
// C = Cx (x) Cy (x) Cz: three small dense transforms forward, one multiplication, three back -- 6 launches instead
// of ~170 (Chebyshev) or ~900 (CG) at the reference's defaults, no iteration, no tolerance, exact to rounding and
// bitwise reproducible (fixed-order sums).  2.7 ms -> 0.1 ms per CFD step on the 29x32x29 mesh of the coupled bench.
struct SpecArgs {
  ChebField a, b;     // up to two fields, components interleaved
  int ntot;
  int n[3];
  int dim;            // direction transformed by this pass
  int inverse;        // 0: out_m = sum_i C[m][i] in_i ; 1: out_i = sum_m C[m][i] in_m
  const double* C;    // forward: [mode][cell]; inverse: [mode][cell] of the back transform.  Uniform direction: the
                      // orthonormal DCT-II matrix for both; graded direction: Q^T W^1/2 and (W^-1/2 Q)^T (see configure)
  const double* in;   // [ntot][ncells] planar (nullptr: read the interleaved fields)
  double* out;        // [ntot][ncells] planar (nullptr: write the interleaved fields)
  const double* lam[3];   // dtau D_d / dx_d^2 * (2 - 2 cos(pi m / n_d)) ; filter applied when `filter` != 0
  int filter, steps;
};

__global__ __launch_bounds__(256) void k_spectral_pass(SpecArgs A)
{
  const int ncells = A.n[0] * A.n[1] * A.n[2];
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;
  if (c >= ncells) return;
  int idx[3] = {c % A.n[0], (c / A.n[0]) % A.n[1], c / (A.n[0] * A.n[1])};
  const int stride_dim = A.dim == 0 ? 1 : (A.dim == 1 ? A.n[0] : A.n[0] * A.n[1]);
  const int nd = A.n[A.dim];
  const int m = idx[A.dim];
  const int base = c - m * stride_dim;          // the line through this cell along `dim`, at coordinate 0
  double acc = 0.0;
  if (A.in) {
    const double* src = A.in + (size_t)k * ncells + base;
    if (!A.inverse)
      for (int i = 0; i < nd; i++) acc += A.C[m * nd + i] * src[(size_t)i * stride_dim];
    else
      for (int i = 0; i < nd; i++) acc += A.C[i * nd + m] * src[(size_t)i * stride_dim];
  } else {
    const ChebField& f = k < A.a.ncomp ? A.a : A.b;
    const int comp = k < A.a.ncomp ? k : k - A.a.ncomp;
    const double* src = f.f + comp;
    if (!A.inverse)
      for (int i = 0; i < nd; i++) acc += A.C[m * nd + i] * src[(size_t)(base + i * stride_dim) * f.ncomp];
    else
      for (int i = 0; i < nd; i++) acc += A.C[i * nd + m] * src[(size_t)(base + i * stride_dim) * f.ncomp];
  }
  if (A.filter) {
    const double g = 1.0 / (1.0 + A.lam[0][idx[0]] + A.lam[1][idx[1]] + A.lam[2][idx[2]]);
    double w = 1.0;
    for (int s = 0; s < A.steps; s++) w *= g;     // one factor per implicit step, like the solves it replaces
    acc *= w;
  }
  if (A.out) {
    A.out[(size_t)k * ncells + c] = acc;
  } else {
    const ChebField& f = k < A.a.ncomp ? A.a : A.b;
    const int comp = k < A.a.ncomp ? k : k - A.a.ncomp;
    f.f[(size_t)c * f.ncomp + comp] = acc;
  }
}

// ---- slab mode: the x direction of the direct solve on complete lines (see sf_smooth.h) ----
// lines [nlines][nx] (x fastest); line L = first_line + l belongs to field k = L / (ny nz), row iz = (L / ny) % nz,
// iy = L % ny of the planar work array.  pass 0: tmp[l][m] = filter * sum_i C[m][i] lines[l][i] ; pass 1: lines[l][i] =
// sum_m C[m][i] tmp[l][m]
__global__ __launch_bounds__(256) void k_slab_xsolve(int pass, double* lines, double* tmp, long long nlines,
                                                     long long first_line, int nx, int ny, int nz, const double* C,
                                                     const double* lamx, const double* lamy, const double* lamz,
                                                     int steps)
{
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nlines * nx) return;
  const long long l = t / nx;
  const int m = (int)(t - l * nx);
  if (pass == 0) {
    const double* src = lines + l * nx;
    double acc = 0.0;
    for (int i = 0; i < nx; i++) acc += C[m * nx + i] * src[i];
    const long long L = first_line + l;
    const int iy = (int)(L % ny), iz = (int)((L / ny) % nz);
    const double g = 1.0 / (1.0 + lamx[m] + lamy[iy] + lamz[iz]);
    double w = 1.0;
    for (int s = 0; s < steps; s++) w *= g;
    tmp[l * nx + m] = acc * w;
  } else {
    const double* src = tmp + l * nx;
    double acc = 0.0;
    for (int i = 0; i < nx; i++) acc += C[i * nx + m] * src[i];
    lines[l * nx + m] = acc;
  }
}

DiffusionSmoother::~DiffusionSmoother()
{
  if (slab_x_) (void)hipFree(slab_x_);
  if (slab_tmp_) (void)hipFree(slab_tmp_);
  for (double* q : {r_, p_, ap_, partial_, scal_, cheb_, spec_})
    if (q) (void)hipFree(q);
  if (h_scal_) (void)hipHostFree(h_scal_);
}

// eigen-decomposition of a symmetric matrix by cyclic Jacobi rotations (n <= 256, set-up time only):
// a (n x n, row major) is destroyed, its diagonal ends up holding the eigenvalues, q the eigenvectors (columns)
static void jacobi_eigen(int n, std::vector<double>& a, std::vector<double>& q)
{
  q.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; i++) q[(size_t)i * n + i] = 1.0;
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) (i == j ? diag : off) += a[(size_t)i * n + j] * a[(size_t)i * n + j];
    if (off <= 1.0e-32 * diag) break;
    for (int p = 0; p < n - 1; p++)
      for (int r = p + 1; r < n; r++) {
        const double apr = a[(size_t)p * n + r];
        if (apr == 0.0) continue;
        const double theta = (a[(size_t)r * n + r] - a[(size_t)p * n + p]) / (2.0 * apr);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < n; k++) {   // columns p and r
          const double akp = a[(size_t)k * n + p], akr = a[(size_t)k * n + r];
          a[(size_t)k * n + p] = c * akp - sn * akr;
          a[(size_t)k * n + r] = sn * akp + c * akr;
        }
        for (int k = 0; k < n; k++) {   // rows p and r
          const double apk = a[(size_t)p * n + k], ark = a[(size_t)r * n + k];
          a[(size_t)p * n + k] = c * apk - sn * ark;
          a[(size_t)r * n + k] = sn * apk + c * ark;
        }
        for (int k = 0; k < n; k++) {
          const double qkp = q[(size_t)k * n + p], qkr = q[(size_t)k * n + r];
          q[(size_t)k * n + p] = c * qkp - sn * qkr;
          q[(size_t)k * n + r] = sn * qkp + c * qkr;
        }
      }
  }
}

void DiffusionSmoother::configure(const int n[3], const double dx[3], const double D[3], double band, int steps,
                                  hipStream_t s, const double* const widths[3], const int* periodic)
{
  s_ = s;
  for (int k = 0; k < 3; k++) per_[k] = (periodic && periodic[k] && n[k] > 1) ? 1 : 0;
  enabled_ = band > 0.0 && steps > 0;
  if (!enabled_) return;
  const bool graded = widths && (widths[0] || widths[1] || widths[2]);
  ncells_ = n[0] * n[1] * n[2];
  steps_ = steps;
  const double tau = band * band / 4.0;            // enhancedCloud.C:564
  const double dtau = tau / (steps + 1.0e-150);    // :565 (diffusionSteps + ROOTVSMALL)
  for (int k = 0; k < 3; k++) {
    n_[k] = n[k];
    double h = dx[k];
    if (widths && widths[k]) {   // graded: the mean width, only for the spectrum bound of the iterative solvers
      h = 0.0;
      for (int i = 0; i < n[k]; i++) h += widths[k][i];
      h /= n[k];
    }
    c_[k] = dtau * D[k] / (h * h);
  }
  nblocks_ = div_up(ncells_, 256);
  SF_HIP(hipMalloc(&r_, sizeof(double) * ncells_));
  SF_HIP(hipMalloc(&p_, sizeof(double) * ncells_));
  SF_HIP(hipMalloc(&ap_, sizeof(double) * ncells_));
  SF_HIP(hipMalloc(&partial_, sizeof(double) * 2 * nblocks_));
  SF_HIP(hipMalloc(&scal_, sizeof(double) * S_N));
  SF_HIP(hipHostMalloc(&h_scal_, sizeof(double) * S_N));
  // Chebyshev: spectrum bounds and the iteration count for a residual factor of 1e-15
  use_cg_ = getenv("SF_SMOOTH_CG") && atoi(getenv("SF_SMOOTH_CG")) != 0;
  lmin_ = 1.0;
  lmax_ = 1.0 + 4.0 * (c_[0] + c_[1] + c_[2]);
  const double kappa = lmax_ / lmin_;
  const double rho = (std::sqrt(kappa) - 1.0) / (std::sqrt(kappa) + 1.0);
  cheb_iters_ = rho > 0.0 ? (int)std::ceil(std::log(2.0e15) / std::log(1.0 / rho)) : 1;
  cheb_iters_ = std::max(cheb_iters_, 2);
  SF_HIP(hipMalloc(&cheb_, sizeof(double) * 3 * kMaxCheb * ncells_));
  // spectral path: DCT-II matrices and scaled eigenvalues of the three directions
  use_spectral_ = !use_cg_ && std::max(n_[0], std::max(n_[1], n_[2])) <= kMaxSpectral &&
                  !(getenv("SF_SMOOTH_SPECTRAL") && atoi(getenv("SF_SMOOTH_SPECTRAL")) == 0);
  if (graded && !use_spectral_)
    fail("diffusion smoothing on a graded block needs the dense-transform solver (at most %d cells per direction, "
         "no SF_SMOOTH_CG / SF_SMOOTH_SPECTRAL=0)", kMaxSpectral);
  if (use_spectral_) {
    size_t off = 0;
    std::vector<double> h, lam[3];
    for (int d = 0; d < 3; d++) {
      const int nd = n_[d];
      specC_off_[d] = off;
      if (!(widths && widths[d]) && per_[d]) {
        // cyclic axis: the operator is circulant, diagonal in the real Fourier basis -- mode 0 constant, modes
        // 2k-1 / 2k = cos / sin of frequency k, and for even n the alternating mode; eigenvalue 2 - 2 cos(2 pi k / n)
        for (int m = 0; m < nd; m++) {
          const int kf = (m + 1) / 2;
          const bool alt = (nd % 2 == 0) && m == nd - 1;
          for (int i = 0; i < nd; i++) {
            double v;
            if (m == 0) v = std::sqrt(1.0 / nd);
            else if (alt) v = std::sqrt(1.0 / nd) * ((i & 1) ? -1.0 : 1.0);
            else if (m & 1) v = std::sqrt(2.0 / nd) * std::cos(2.0 * M_PI * kf * i / nd);
            else v = std::sqrt(2.0 / nd) * std::sin(2.0 * M_PI * kf * i / nd);
            h.push_back(v);
          }
          lam[d].push_back(c_[d] * (2.0 - 2.0 * std::cos(2.0 * M_PI * kf / nd)));
        }
        off += (size_t)nd * nd;
        specB_off_[d] = specC_off_[d];
        continue;
      }
      if (!(widths && widths[d])) {
        for (int m = 0; m < nd; m++)
          for (int i = 0; i < nd; i++)
            h.push_back(std::sqrt((m == 0 ? 1.0 : 2.0) / nd) * std::cos(M_PI * m * (i + 0.5) / nd));
        off += (size_t)nd * nd;
        specB_off_[d] = specC_off_[d];   // orthonormal: the back transform is the transpose of the same matrix
        for (int m = 0; m < nd; m++) lam[d].push_back(c_[d] * (2.0 - 2.0 * std::cos(M_PI * m / nd)));
        continue;
      }
      // Graded direction.  The finite-volume operator of fvm::laplacian on this orthogonal mesh is, along d,
      //   (L phi)_i = (1/h_i) [ (phi_{i+1} - phi_i)/d_{i+1/2} - (phi_i - phi_{i-1})/d_{i-1/2} ],  d = distance of centres,
      // zero flux at both ends.  S = W^1/2 (-L) W^-1/2 (W = diag h) is symmetric: S = Q Lambda Q^T, so
      //   phi_hat = Q^T W^1/2 phi   and   phi = W^-1/2 Q phi_hat
      // play the roles of the cosine transform and its inverse; the directions still commute (tensor product).
      const double* w = widths[d];
      std::vector<double> S((size_t)nd * nd, 0.0), Q;
      for (int i = 0; i + 1 < nd; i++) {
        const double dist = 0.5 * (w[i] + w[i + 1]);
        S[(size_t)i * nd + i] += 1.0 / (w[i] * dist);
        S[(size_t)(i + 1) * nd + i + 1] += 1.0 / (w[i + 1] * dist);
        S[(size_t)i * nd + i + 1] = S[(size_t)(i + 1) * nd + i] = -1.0 / (std::sqrt(w[i] * w[i + 1]) * dist);
      }
      if (per_[d]) {   // cyclic pair: the last and the first cell are neighbours (nd = 2: the same pair twice)
        const int a = nd - 1, b = 0;
        const double dist = 0.5 * (w[a] + w[b]);
        S[(size_t)a * nd + a] += 1.0 / (w[a] * dist);
        S[(size_t)b * nd + b] += 1.0 / (w[b] * dist);
        S[(size_t)a * nd + b] += -1.0 / (std::sqrt(w[a] * w[b]) * dist);
        S[(size_t)b * nd + a] = S[(size_t)a * nd + b];
      }
      jacobi_eigen(nd, S, Q);
      for (int m = 0; m < nd; m++)          // forward [mode m][cell i] = Q[i][m] sqrt(h_i)
        for (int i = 0; i < nd; i++) h.push_back(Q[(size_t)i * nd + m] * std::sqrt(w[i]));
      off += (size_t)nd * nd;
      specB_off_[d] = off;
      for (int m = 0; m < nd; m++)          // back, stored [mode m][cell i] = Q[i][m] / sqrt(h_i)
        for (int i = 0; i < nd; i++) h.push_back(Q[(size_t)i * nd + m] / std::sqrt(w[i]));
      off += (size_t)nd * nd;
      for (int m = 0; m < nd; m++) lam[d].push_back(dtau * D[d] * std::max(S[(size_t)m * nd + m], 0.0));
    }
    for (int d = 0; d < 3; d++) {
      specL_off_[d] = off;
      h.insert(h.end(), lam[d].begin(), lam[d].end());
      off += n_[d];
    }
    spec_work_off_ = off;
    SF_HIP(hipMalloc(&spec_, sizeof(double) * (off + 2 * (size_t)kMaxCheb * ncells_)));
    SF_HIP(hipMemcpyAsync(spec_, h.data(), sizeof(double) * off, hipMemcpyHostToDevice, s_));
    SF_HIP(hipStreamSynchronize(s_));
  }
}

void DiffusionSmoother::configure_slab(int nx_global)
{
  if (!enabled_) {
    nx_global_ = nx_global;
    return;
  }
  if (!use_spectral_) fail("a mesh partitioned into x-slabs needs the dense-transform smoother (no SF_SMOOTH_CG / SF_SMOOTH_SPECTRAL=0)");
  if (nx_global < 1 || nx_global > kMaxSpectral) fail("slab smoothing: %d cells along x (1..%d)", nx_global, kMaxSpectral);
  nx_global_ = nx_global;
  const int nd = nx_global;
  // orthonormal basis of the WHOLE x direction (the local block only knows its slab): cosine modes, or the real
  // Fourier modes of a cyclic pair; eigenvalues scaled like c_[0] = dtau D_x / dx^2
  std::vector<double> h;
  std::vector<double> lam;
  for (int m = 0; m < nd; m++) {
    const int kf = (m + 1) / 2;
    const bool alt = (nd % 2 == 0) && m == nd - 1;
    for (int i = 0; i < nd; i++) {
      double v;
      if (!per_[0]) v = std::sqrt((m == 0 ? 1.0 : 2.0) / nd) * std::cos(M_PI * m * (i + 0.5) / nd);
      else if (m == 0) v = std::sqrt(1.0 / nd);
      else if (alt) v = std::sqrt(1.0 / nd) * ((i & 1) ? -1.0 : 1.0);
      else if (m & 1) v = std::sqrt(2.0 / nd) * std::cos(2.0 * M_PI * kf * i / nd);
      else v = std::sqrt(2.0 / nd) * std::sin(2.0 * M_PI * kf * i / nd);
      h.push_back(v);
    }
    lam.push_back(per_[0] ? c_[0] * (2.0 - 2.0 * std::cos(2.0 * M_PI * kf / nd)) : c_[0] * (2.0 - 2.0 * std::cos(M_PI * m / nd)));
  }
  h.insert(h.end(), lam.begin(), lam.end());
  SF_HIP(hipMalloc(&slab_x_, sizeof(double) * h.size()));
  SF_HIP(hipMemcpyAsync(slab_x_, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, s_));
  SF_HIP(hipStreamSynchronize(s_));
}

double* DiffusionSmoother::work() const { return spec_ + spec_work_off_ + (size_t)kMaxCheb * ncells_; }

void DiffusionSmoother::begin(double* fa, int na, double* fb, int nb)
{
  if (!enabled_ || !slab()) fail("DiffusionSmoother::begin: not in slab mode");
  if (slab_open_) fail("DiffusionSmoother::begin: the previous solve was not finished");
  if (na + nb > kMaxCheb) fail("DiffusionSmoother::begin: at most %d field components per solve", kMaxCheb);
  slab_fa_ = fa;
  slab_fb_ = fb;
  slab_na_ = na;
  slab_nb_ = nb;
  slab_ntot_ = na + nb;
  SpecArgs A;
  A.a = {fa, na};
  A.b = {fb, nb};
  A.ntot = na + nb;
  for (int d = 0; d < 3; d++) {
    A.n[d] = n_[d];
    A.lam[d] = spec_ + specL_off_[d];
  }
  A.steps = steps_;
  A.filter = 0;
  double* w0 = spec_ + spec_work_off_;
  double* w1 = w0 + (size_t)kMaxCheb * ncells_;
  const dim3 grid(div_up(ncells_, 256), A.ntot);
  // forward y (fields -> w0), forward z (w0 -> w1): the local part of the solve
  A.inverse = 0;
  A.dim = 1;
  A.C = spec_ + specC_off_[1];
  A.in = nullptr;
  A.out = w0;
  k_spectral_pass<<<grid, 256, 0, s_>>>(A);
  A.dim = 2;
  A.C = spec_ + specC_off_[2];
  A.in = w0;
  A.out = w1;
  k_spectral_pass<<<grid, 256, 0, s_>>>(A);
  SF_HIP(hipGetLastError());
  slab_open_ = true;
}

void DiffusionSmoother::xsolve(double* lines, long long nlines, long long first_line)
{
  if (!slab_open_) fail("DiffusionSmoother::xsolve: no solve in progress");
  if (nlines <= 0) return;
  const size_t need = (size_t)nlines * nx_global_;
  if (need > slab_tmp_cap_) {
    if (slab_tmp_) SF_HIP(hipFree(slab_tmp_));
    slab_tmp_cap_ = need + need / 4 + 1024;
    SF_HIP(hipMalloc(&slab_tmp_, sizeof(double) * slab_tmp_cap_));
  }
  const double* C = slab_x_;
  const double* lamx = slab_x_ + (size_t)nx_global_ * nx_global_;
  const int nb = div_up((long long)need, 256);
  for (int pass = 0; pass < 2; pass++)
    k_slab_xsolve<<<nb, 256, 0, s_>>>(pass, lines, slab_tmp_, nlines, first_line, nx_global_, n_[1], n_[2], C, lamx,
                                      spec_ + specL_off_[1], spec_ + specL_off_[2], steps_);
  SF_HIP(hipGetLastError());
}

void DiffusionSmoother::end()
{
  if (!slab_open_) fail("DiffusionSmoother::end: no solve in progress");
  SpecArgs A;
  A.a = {slab_fa_, slab_na_};
  A.b = {slab_fb_, slab_nb_};
  A.ntot = slab_ntot_;
  for (int d = 0; d < 3; d++) {
    A.n[d] = n_[d];
    A.lam[d] = spec_ + specL_off_[d];
  }
  A.steps = steps_;
  A.filter = 0;
  double* w0 = spec_ + spec_work_off_;
  double* w1 = w0 + (size_t)kMaxCheb * ncells_;
  const dim3 grid(div_up(ncells_, 256), A.ntot);
  // inverse z (w1 -> w0), inverse y (w0 -> fields)
  A.inverse = 1;
  A.dim = 2;
  A.C = spec_ + specB_off_[2];
  A.in = w1;
  A.out = w0;
  k_spectral_pass<<<grid, 256, 0, s_>>>(A);
  A.dim = 1;
  A.C = spec_ + specB_off_[1];
  A.in = w0;
  A.out = nullptr;
  k_spectral_pass<<<grid, 256, 0, s_>>>(A);
  SF_HIP(hipGetLastError());
  slab_open_ = false;
}

void DiffusionSmoother::smooth_spectral(double* fa, int na, double* fb, int nb)
{
  SpecArgs A;
  A.a = {fa, na};
  A.b = {fb, nb};
  A.ntot = na + nb;
  for (int d = 0; d < 3; d++) {
    A.n[d] = n_[d];
    A.lam[d] = spec_ + specL_off_[d];
  }
  A.steps = steps_;
  double* w0 = spec_ + spec_work_off_;
  double* w1 = w0 + (size_t)kMaxCheb * ncells_;
  const dim3 grid(div_up(ncells_, 256), A.ntot);
  // forward x, y, z (filter on the last), inverse z, y, x (the last one writes the fields)
  const int order[6] = {0, 1, 2, 2, 1, 0};
  const double* in = nullptr;
  for (int pass = 0; pass < 6; pass++) {
    A.dim = order[pass];
    A.inverse = pass >= 3;
    A.C = spec_ + (A.inverse ? specB_off_[A.dim] : specC_off_[A.dim]);
    A.in = in;
    A.out = pass == 5 ? nullptr : ((pass & 1) ? w1 : w0);
    A.filter = pass == 2;
    k_spectral_pass<<<grid, 256, 0, s_>>>(A);
    in = A.out;
  }
  SF_HIP(hipGetLastError());
}

void DiffusionSmoother::smooth2(double* fa, int na, double* fb, int nb)
{
  Range r("diffusion");
  if (!enabled_) return;
  if (use_cg_ || na + nb > kMaxCheb) {
    smooth(fa, na);
    if (nb) smooth(fb, nb);
    return;
  }
  if (use_spectral_) {
    smooth_spectral(fa, na, fb, nb);
    return;
  }
  ChebArgs A;
  for (int k = 0; k < 3; k++) {
    A.st.n[k] = n_[k];
    A.st.c[k] = c_[k];
    A.st.per[k] = per_[k];
  }
  A.st.ncells = ncells_;
  A.a = {fa, na};
  A.b = {fb, nb};
  A.ntot = na + nb;
  A.r = cheb_;
  double* da = cheb_ + (size_t)kMaxCheb * ncells_;
  double* db = da + (size_t)kMaxCheb * ncells_;
  const double theta = 0.5 * (lmax_ + lmin_), delta = 0.5 * (lmax_ - lmin_), sigma1 = theta / delta;
  const dim3 grid(div_up(ncells_, 256), A.ntot);
  for (int step = 0; step < steps_; step++) {     // while (diffusionRunTime_.loop()) :825-838
    A.d1 = da;
    k_cheb_init<<<grid, 256, 0, s_>>>(A, 1.0 / theta);
    double rho = 1.0 / sigma1;
    double *cur = da, *nxt = db;
    for (int it = 0; it < cheb_iters_; it++) {
      const double rho_new = 1.0 / (2.0 * sigma1 - rho);
      A.d0 = cur;
      A.d1 = nxt;
      A.c_rho = rho_new * rho;
      A.c_r = 2.0 * rho_new / delta;
      k_cheb_iter<<<grid, 256, 0, s_>>>(A);
      rho = rho_new;
      std::swap(cur, nxt);
    }
    iters_ += cheb_iters_;
  }
  SF_HIP(hipGetLastError());
}

void DiffusionSmoother::solve_component(double* x, int stride)
{
  Stencil st;
  for (int k = 0; k < 3; k++) {
    st.n[k] = n_[k];
    st.c[k] = c_[k];
    st.per[k] = per_[k];
  }
  st.ncells = ncells_;
  const dim3 grid(nblocks_);
  k_cg_init<<<grid, 256, 0, s_>>>(st, x, stride, r_, p_, partial_);
  k_cg_scalars<<<1, 256, 0, s_>>>(0, nblocks_, partial_, scal_);
  const double tol2 = 1.0e-26;   // ||r||^2 <= 1e-26 ||b||^2 : far below the reference's PCG tolerance of 1e-10
  for (int it = 0; it < 2000; it++) {
    if ((it & 7) == 0) {
      SF_HIP(hipMemcpyAsync(h_scal_, scal_, sizeof(double) * S_N, hipMemcpyDeviceToHost, s_));
      SF_HIP(hipStreamSynchronize(s_));
      if (!(h_scal_[S_RRNEW] > tol2 * h_scal_[S_BB])) return;   // also leaves on an all-zero field
    }
    k_cg_ap<<<grid, 256, 0, s_>>>(st, p_, ap_, partial_);
    k_cg_scalars<<<1, 256, 0, s_>>>(1, nblocks_, partial_, scal_);
    k_cg_update<<<grid, 256, 0, s_>>>(ncells_, scal_, x, stride, r_, p_, ap_, partial_);
    k_cg_scalars<<<1, 256, 0, s_>>>(2, nblocks_, partial_, scal_);
    k_cg_p<<<grid, 256, 0, s_>>>(ncells_, scal_, r_, p_);
    iters_++;
  }
  fail("smoothField: conjugate gradients did not converge in 2000 iterations");
}

void DiffusionSmoother::smooth(double* field, int ncomp)
{
  Range r("diffusion");   // writeCPUTime.H bucket (enhancedCloud::smoothField)
  if (!enabled_) return;
  if (!use_cg_ && ncomp <= kMaxCheb) {
    smooth2(field, ncomp, nullptr, 0);
    return;
  }
  for (int step = 0; step < steps_; step++)       // while (diffusionRunTime_.loop()) :825-838
    for (int k = 0; k < ncomp; k++) solve_component(field + k, ncomp);
  SF_HIP(hipGetLastError());
}

}  // namespace sf
