// sf_kernels_api.hip -- stand-alone per-style entry points on LAMMPS-shaped data (AoS double[n][3]
// atom arrays + CSR neighbour lists on the device): what a thin LAMMPS PairStyle / FixStyle adapter
// calls from its compute() / post_force() after flattening NeighList.
//   sfk_pair_gran_history_compute   <- PairGranHertzFixHistory::compute  pair_gran_hertzFix_history.cpp:45-287
//   sfk_fix_cohesive_post_force     <- FixCohe::post_force               fix_cohesive.cpp:138-263
//   sfk_pair_lubricate_poly_compute <- PairLubricatePoly::compute        pair_lubricate_poly.cpp:65-444
//   sfk_fix_fluid_drag_post_force   <- FixFluidDrag::post_force          fix_fluid_drag.cpp:114-164
// One owned atom (list row) per lane; the j side of a half-list pair is updated with FP64 atomics.
#include <cmath>

#include "../../include/sedifoam_amd.h"
#include "sf_common.h"
#include "sf_physics.h"

namespace sf {

__device__ __forceinline__ Vec3 ld3(const double* a, int i) { return {a[3 * i], a[3 * i + 1], a[3 * i + 2]}; }
__device__ __forceinline__ void atomic_add3(double* a, int i, Vec3 v)
{
  atomicAdd(&a[3 * i], v.x);
  atomicAdd(&a[3 * i + 1], v.y);
  atomicAdd(&a[3 * i + 2], v.z);
}

template <int STYLE>
__global__ __launch_bounds__(256) void k_pair_gran_csr(GranParams p, double dt, int shearupdate, int nlocal,
                                                       int inum, const int* ilist, const int* first,
                                                       const int* jlist, int* touch, double* shear,
                                                       const double* x, const double* v, const double* omega,
                                                       const double* radius, const double* rmass,
                                                       const int* mask, int freeze_bit, double* f, double* torque,
                                                       const double* mass_rigid)
{
  const int ii = blockIdx.x * blockDim.x + threadIdx.x;
  if (ii >= inum) return;
  const int i = ilist[ii];
  const Vec3 xi = ld3(x, i), vi = ld3(v, i), wi = ld3(omega, i);
  const double radi = radius[i];
  // (an atom of a fix rigid body collides with the mass of its body: pair_gran_hertzFix_history.cpp:72-86, 182-185)
  const double mi = (mass_rigid && mass_rigid[i] > 0.0) ? mass_rigid[i] : rmass[i];
  const int maski = mask[i];
  Vec3 F = {0, 0, 0}, T = {0, 0, 0};
  for (int jj = first[ii]; jj < first[ii + 1]; jj++) {
    const int j = jlist[jj] & 0x3FFFFFFF;
    const Vec3 del = xi - ld3(x, j);
    const double rsq = dot(del, del);
    const double radj = radius[j];
    const double radsum = radi + radj;
    if (rsq >= radsum * radsum) {
      touch[jj] = 0;
      shear[3 * (size_t)jj] = shear[3 * (size_t)jj + 1] = shear[3 * (size_t)jj + 2] = 0.0;
      continue;
    }
    ContactIn c;
    c.del = del;
    c.rsq = rsq;
    c.r = sqrt(rsq);
    c.rinv = 1.0 / c.r;
    c.vr = vi - ld3(v, j);
    const Vec3 wj = ld3(omega, j);
    c.wsum = {radi * wi.x + radj * wj.x, radi * wi.y + radj * wj.y, radi * wi.z + radj * wj.z};
    const double mj = (mass_rigid && mass_rigid[j] > 0.0) ? mass_rigid[j] : rmass[j];
    c.meff = mi * mj / (mi + mj);
    if (maski & freeze_bit) c.meff = mj;
    if (mask[j] & freeze_bit) c.meff = mi;
    c.overlap = radsum - c.r;
    c.reff = (radsum - c.r) * radi * radj / radsum;
    touch[jj] = 1;
    Vec3 sh = {shear[3 * (size_t)jj], shear[3 * (size_t)jj + 1], shear[3 * (size_t)jj + 2]};
    ContactOut o;
    gran_history_law<STYLE>(p, dt, shearupdate != 0, c, sh, o);
    shear[3 * (size_t)jj] = sh.x;
    shear[3 * (size_t)jj + 1] = sh.y;
    shear[3 * (size_t)jj + 2] = sh.z;
    F = F + o.F;
    T = T - radi * o.tor;
    if (j < nlocal) {
      atomic_add3(f, j, Vec3{-o.F.x, -o.F.y, -o.F.z});
      atomic_add3(torque, j, Vec3{-radj * o.tor.x, -radj * o.tor.y, -radj * o.tor.z});
    }
  }
  atomic_add3(f, i, F);
  atomic_add3(torque, i, T);
}

__global__ __launch_bounds__(256) void k_cohesive_csr(CoheParams p, int nlocal, int newton_pair, const int* ilist,
                                                      const int* first, const int* jlist, const double* x,
                                                      const double* radius, const int* mask, int groupbit,
                                                      double* f)
{
  const int ii = blockIdx.x * blockDim.x + threadIdx.x;
  if (ii >= nlocal) return;  // the reference loops ii < nlocal (fix_cohesive.cpp:165)
  const int i = ilist[ii];
  if (!(mask[i] & groupbit)) return;
  const Vec3 xi = ld3(x, i);
  const double radi = radius[i];
  Vec3 F = {0, 0, 0};
  for (int jj = first[ii]; jj < first[ii + 1]; jj++) {
    const int j = jlist[jj];
    const Vec3 del = xi - ld3(x, j);
    const double rsq = dot(del, del);
    const double radsum = radi + radius[j];
    const double rc = radsum + p.smax;
    if (!(rsq < rc * rc)) continue;
    const double r = sqrt(rsq);
    const double cc = cohesive_ccel(p, r, radsum) * (1 / r);
    const Vec3 c = {del.x * cc, del.y * cc, del.z * cc};
    F = F + c;
    if (newton_pair || j < nlocal) atomic_add3(f, j, Vec3{-c.x, -c.y, -c.z});
  }
  atomic_add3(f, i, F);
}

__global__ __launch_bounds__(256) void k_lubricate_csr(LubParams p, int inum, const int* ilist, const int* first,
                                                       const int* jlist, const double* x, const double* v,
                                                       const double* omega, const double* radius, double* f,
                                                       double* torque)
{
  const int ii = blockIdx.x * blockDim.x + threadIdx.x;
  if (ii >= inum) return;
  const int i = ilist[ii];
  const Vec3 xi = ld3(x, i), vi = ld3(v, i), wi = ld3(omega, i);
  const double radi = radius[i];
  Vec3 F = {0, 0, 0}, T = {0, 0, 0};
  if (p.flagfld) {
    F = F - (p.vxmu2f * p.R0 * radi) * vi;
    T = T - (p.vxmu2f * p.RT0 * (radi * radi * radi)) * wi;
  }
  if (p.flagHI) {
    const double cutsq = p.cut_global * p.cut_global;
    for (int jj = first[ii]; jj < first[ii + 1]; jj++) {
      const int j = jlist[jj];
      const Vec3 del = xi - ld3(x, j);
      const double rsq = dot(del, del);
      if (!(rsq < cutsq)) continue;
      double r, rinv;
      sf_sqrt_rsqrt(rsq, r, rinv);
      lubricate_poly_pair(p, del, r, rinv, radi, radius[j], vi, ld3(v, j), wi, ld3(omega, j), F, T);
    }
  }
  atomic_add3(f, i, F);
  atomic_add3(torque, i, T);
}

__global__ __launch_bounds__(256) void k_fdrag_aos(int nlocal, double dt, double carrier_rho, const double* v,
                                                   const double* rmass, const double* radius, const int* mask,
                                                   int groupbit, const double* ffluiddrag, const double* DuDt,
                                                   double* vOld, double* f)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nlocal || !(mask[i] & groupbit)) return;
  const double r = radius[i], m = rmass[i];
  const double rho = 3.0 * m / (4.0 * kPiTypo * r * r * r);
  for (int k = 0; k < 3; k++) {
    const double acc = (v[3 * i + k] - vOld[3 * i + k]) / dt;
    f[3 * i + k] += ffluiddrag[3 * i + k] + carrier_rho / rho * 0.5 * m * (DuDt[3 * i + k] - acc);
    vOld[3 * i + k] = v[3 * i + k];
  }
}

// FixWallGranFix::post_force (fix_wall_granFix.cpp:286-344) on LAMMPS-shaped AoS arrays: plane pair (wallstyle 0 / 1 / 2) or
// z cylinder (3), the three laws of :361-678 through the contact-law functions the engine's sub-step kernel uses
// (sf_physics.h); geometry as in substep_particle (sf_dem_kernels.h).  shear[n][3] is the fix's per-atom history.
__global__ __launch_bounds__(256) void k_wall_granfix_aos(GranParams p, int history, int wallstyle, double lo, double hi,
                                                          double cylradius, double dt, int shearupdate, int nlocal,
                                                          const double* x, const double* v, const double* omega,
                                                          const double* radius, const double* rmass, const int* mask,
                                                          int groupbit, double* shear, double* f, double* torque)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nlocal || !(mask[i] & groupbit)) return;   // :290
  const Vec3 xi = {x[3 * i], x[3 * i + 1], x[3 * i + 2]};
  const double radi = radius[i];
  Vec3 dw = {0.0, 0.0, 0.0};
  bool contact = true;
  if (wallstyle < 3) {   // :294-308
    const double xc = wallstyle == 0 ? xi.x : (wallstyle == 1 ? xi.y : xi.z);
    const double del1 = xc - lo, del2 = hi - xc;
    const double d = del1 < del2 ? del1 : -del2;
    dw = {wallstyle == 0 ? d : 0.0, wallstyle == 1 ? d : 0.0, wallstyle == 2 ? d : 0.0};
  } else {               // :309-322
    const double delxy = sqrt(xi.x * xi.x + xi.y * xi.y);
    const double delr = cylradius - delxy;
    if (delr > radi) contact = false;
    else dw = {-delr / delxy * xi.x, -delr / delxy * xi.y, 0.0};
  }
  const double rsq = dot(dw, dw);
  if (!contact || rsq > radi * radi) {   // :324-329
    if (history) shear[3 * i] = shear[3 * i + 1] = shear[3 * i + 2] = 0.0;
    return;
  }
  ContactIn c;
  c.del = dw;
  c.rsq = rsq;
  c.r = sqrt(rsq);
  c.rinv = 1.0 / c.r;
  c.vr = {v[3 * i], v[3 * i + 1], v[3 * i + 2]};   // (a wall at rest: vwall = 0)
  c.wsum = {radi * omega[3 * i], radi * omega[3 * i + 1], radi * omega[3 * i + 2]};
  c.meff = rmass[i];
  c.overlap = radi - c.r;
  c.reff = (radi - c.r) * radi;
  Vec3 sh = {0.0, 0.0, 0.0};
  if (history) sh = {shear[3 * i], shear[3 * i + 1], shear[3 * i + 2]};
  ContactOut o;
  if (p.style == 2) hertz_history_law(p, dt, shearupdate != 0, c, sh, o);
  else hooke_history_law(p, dt, shearupdate != 0, c, sh, o);
  if (history) {
    shear[3 * i] = sh.x;
    shear[3 * i + 1] = sh.y;
    shear[3 * i + 2] = sh.z;
  }
  f[3 * i] += o.F.x;
  f[3 * i + 1] += o.F.y;
  f[3 * i + 2] += o.F.z;
  torque[3 * i] -= radi * o.tor.x;
  torque[3 * i + 1] -= radi * o.tor.y;
  torque[3 * i + 2] -= radi * o.tor.z;
}

static double beta_of(double gamman)
{
  const double lg = std::log(gamman) / std::log(std::exp(1.0));
  return -(lg) / std::sqrt(lg * lg + kPi * kPi);
}

}  // namespace sf

extern "C" {

void* sf_dev_alloc(size_t bytes)
{
  void* p = nullptr;
  const hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
  if (e != hipSuccess) {
    sf::set_error("sf_dev_alloc(%zu): %s", bytes, hipGetErrorString(e));
    return nullptr;
  }
  return p;
}

int sf_dev_free(void* dev)
{
  SF_API_BEGIN
  if (dev) SF_HIP(hipFree(dev));
  SF_API_END(0)
}

int sf_dev_upload(void* dev_dst, const void* host_src, size_t bytes, void* stream)
{
  SF_API_BEGIN
  if (bytes) SF_HIP(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  SF_API_END(0)
}

int sf_dev_download(void* host_dst, const void* dev_src, size_t bytes, void* stream)
{
  SF_API_BEGIN
  if (bytes) SF_HIP(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  SF_HIP(hipStreamSynchronize((hipStream_t)stream));
  SF_API_END(0)
}

int sf_dev_zero(void* dev, size_t bytes, void* stream)
{
  SF_API_BEGIN
  if (bytes) SF_HIP(hipMemsetAsync(dev, 0, bytes, (hipStream_t)stream));
  SF_API_END(0)
}

int sf_dev_sync(void* stream)
{
  SF_API_BEGIN
  SF_HIP(hipStreamSynchronize((hipStream_t)stream));
  SF_API_END(0)
}

int sfk_gran_settings(sfk_gran_params* p, double kn, int kt_null, double kt, double gamman, int gammat_null,
                      double gammat, double xmu, int dampflag, double nktv2p)
{
  p->kn = kn;
  p->kt = kt_null ? kn * 2.0 / 7.0 : kt;
  p->gamman = gamman;
  p->gammat = gammat_null ? 0.5 * gamman : gammat;
  p->xmu = xmu;
  p->dampflag = dampflag;
  if (dampflag == 0) p->gammat = 0.0;
  if (p->kn < 0.0 || p->kt < 0.0 || p->gamman < 0.0 || p->gammat < 0.0 || p->xmu < 0.0 || p->xmu > 10000.0 ||
      dampflag < 0 || dampflag > 1) {
    sf::set_error("Illegal pair_style command");
    return -1;
  }
  p->kn /= nktv2p;
  p->kt /= nktv2p;
  return 0;
}

int sfk_pair_gran_history_compute_rigid(int hertz, const sfk_gran_params* p, double dt, int shearupdate, int nlocal,
                                  int inum, const int* ilist, const int* first, const int* jlist, int* touch,
                                  double* shear, const double* x, const double* v, const double* omega,
                                  const double* radius, const double* rmass, const int* mask,
                                  int freeze_group_bit, double* f, double* torque, const double* mass_rigid,
                                        void* stream)
{
  SF_API_BEGIN
  sf::GranParams g;
  g.kn = p->kn;
  g.kt = p->kt;
  g.gamman = p->gamman;
  g.gammat = p->gammat;
  g.xmu = p->xmu;
  g.dampflag = p->dampflag;
  g.style = hertz ? 2 : 1;
  g.beta = (hertz && p->gamman > 0.0) ? sf::beta_of(p->gamman) : 0.0;
  sf::fold_hertz_constants(g);
  if (inum > 0) {
    const dim3 grid(sf::div_up(inum, 256));
    hipStream_t s = (hipStream_t)stream;
    if (hertz)
      sf::k_pair_gran_csr<2><<<grid, 256, 0, s>>>(g, dt, shearupdate, nlocal, inum, ilist, first, jlist, touch,
                                                  shear, x, v, omega, radius, rmass, mask, freeze_group_bit, f,
                                                  torque, mass_rigid);
    else
      sf::k_pair_gran_csr<1><<<grid, 256, 0, s>>>(g, dt, shearupdate, nlocal, inum, ilist, first, jlist, touch,
                                                  shear, x, v, omega, radius, rmass, mask, freeze_group_bit, f,
                                                  torque, mass_rigid);
    SF_HIP(hipGetLastError());
  }
  SF_API_END(0)
}

int sfk_pair_gran_history_compute(int hertz, const sfk_gran_params* p, double dt, int shearupdate, int nlocal,
                                  int inum, const int* ilist, const int* first, const int* jlist, int* touch,
                                  double* shear, const double* x, const double* v, const double* omega,
                                  const double* radius, const double* rmass, const int* mask,
                                  int freeze_group_bit, double* f, double* torque, void* stream)
{
  return sfk_pair_gran_history_compute_rigid(hertz, p, dt, shearupdate, nlocal, inum, ilist, first, jlist, touch, shear, x, v,
                                             omega, radius, rmass, mask, freeze_group_bit, f, torque, nullptr, stream);
}

int sfk_fix_cohesive_post_force(double ah, double lam, double smin, double smax, int opt, int nlocal,
                                int newton_pair, const int* ilist, const int* first, const int* jlist,
                                const double* x, const double* radius, const int* mask, int groupbit, double* f,
                                void* stream)
{
  SF_API_BEGIN
  if (opt != 0 && opt != 1) sf::fail("invalid option for cohesive force model");
  sf::CoheParams p{ah, lam, smin, smax, opt, 1};
  if (nlocal > 0) {
    sf::k_cohesive_csr<<<sf::div_up(nlocal, 256), 256, 0, (hipStream_t)stream>>>(p, nlocal, newton_pair, ilist,
                                                                                 first, jlist, x, radius, mask,
                                                                                 groupbit, f);
    SF_HIP(hipGetLastError());
  }
  SF_API_END(0)
}

int sfk_pair_lubricate_poly_compute(const sfk_lub_params* p, int inum, const int* ilist, const int* first,
                                    const int* jlist, const double* x, const double* v, const double* omega,
                                    const double* radius, double* f, double* torque, void* stream)
{
  SF_API_BEGIN
  sf::LubParams l;
  l.mu = p->mu;
  l.cut_inner = p->cut_inner;
  l.cut_global = p->cut_global;
  l.R0 = p->R0;
  l.RT0 = p->RT0;
  l.RS0 = p->RS0;
  l.vxmu2f = p->vxmu2f;
  l.flaglog = p->flaglog;
  l.flagfld = p->flagfld;
  l.flagHI = p->flagHI;
  l.flagVF = p->flagVF;
  l.enabled = 1;
  if (inum > 0) {
    sf::k_lubricate_csr<<<sf::div_up(inum, 256), 256, 0, (hipStream_t)stream>>>(l, inum, ilist, first, jlist, x, v,
                                                                                omega, radius, f, torque);
    SF_HIP(hipGetLastError());
  }
  SF_API_END(0)
}

int sfk_fix_wall_granfix_post_force(int pairstyle, const sfk_gran_params* p, int wallstyle, double lo, double hi,
                                    double cylradius, double dt, int shearupdate, int nlocal, const double* x,
                                    const double* v, const double* omega, const double* radius, const double* rmass,
                                    const int* mask, int groupbit, double* shear, double* f, double* torque, void* stream)
{
  SF_API_BEGIN
  if (pairstyle < 0 || pairstyle > 2 || wallstyle < 0 || wallstyle > 3) sf::fail("Illegal fix wall/gran command");
  sf::GranParams g;
  g.kn = p->kn;
  g.kt = p->kt;
  g.gamman = p->gamman;
  g.gammat = p->gammat;
  g.xmu = p->xmu;
  g.dampflag = p->dampflag;
  // the reference's enum {HOOKE, HOOKE_HISTORY, HERTZ_HISTORY} (fix_wall_granFix.cpp:38) -> the engine's law selector
  g.style = pairstyle == 2 ? 2 : (pairstyle == 1 ? 1 : 3);
  g.beta = (pairstyle == 2 && p->gamman > 0.0) ? sf::beta_of(p->gamman) : 0.0;
  sf::fold_hertz_constants(g);
  if (nlocal > 0) {
    sf::k_wall_granfix_aos<<<sf::div_up(nlocal, 256), 256, 0, (hipStream_t)stream>>>(
        g, pairstyle != 0 ? 1 : 0, wallstyle, lo, hi, cylradius, dt, shearupdate, nlocal, x, v, omega, radius, rmass, mask,
        groupbit, shear, f, torque);
    SF_HIP(hipGetLastError());
  }
  SF_API_END(0)
}

int sfk_fix_fluid_drag_post_force(int nlocal, double dt, double carrier_rho, const double* v, const double* rmass,
                                  const double* radius, const int* mask, int groupbit, const double* ffluiddrag,
                                  const double* DuDt, double* vOld, double* f, void* stream)
{
  SF_API_BEGIN
  if (nlocal > 0) {
    sf::k_fdrag_aos<<<sf::div_up(nlocal, 256), 256, 0, (hipStream_t)stream>>>(
        nlocal, dt, carrier_rho, v, rmass, radius, mask, groupbit, ffluiddrag, DuDt, vOld, f);
    SF_HIP(hipGetLastError());
  }
  SF_API_END(0)
}

}  // extern "C"
