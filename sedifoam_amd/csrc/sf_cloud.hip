// sf_cloud.hip -- the OpenFOAM-side particle path on device-resident state:
//   dragModel::Jd      lammpsFoam/dragModels/ErgunWenYu/ErgunWenYu.C:86-145,
//                      lammpsFoam/dragModels/SyamlalOBrien/SyamlalOBrien.C:85-144
//   enhancedCloud      lammpsFoam/enhancedCloud.C: updateParticleAlpha :56-76, updateParticleUr :83-109,
//                      updateDragOnParticles :112-257, calcTcFields :316-441, evolve :669-787,
//                      particleToEulerianField :911-980
//   adjustLampTimestep lammpsFoam/softParticleCloud.C:209-261
//   cell owner         result of softParticle::move tracking (softParticle.C:102-151) on a uniform hex block
//
// Particles are NOT copied into a cloud-side list: every kernel reads the DEM engine's double4 records
// (same order, same HBM), the drag force is written straight into fix fdrag's array, so the reference's
// assemble/transpose/flatten/tag-sort marshalling (softParticleCloud.C:716-846, 970-1089) has no
// counterpart here.
#include <chrono>
#include <cmath>
#include <vector>

#include "../../include/sedifoam_amd.h"
#include "sf_handles.h"
#include "sf_roctx.h"
#include "sf_smooth.h"

namespace sf {

constexpr double kRootVSmall = 1.0e-150;  // OpenFOAM ROOTVSMALL (double precision)

__device__ __forceinline__ double jd_ergun_wenyu(double Ur, double alpha, double pd, double nuf, double rhof)
{
  const double beta = fmax(1.0 - alpha, kRootVSmall);
  const double bp = pow(beta, -2.65);
  const double Re = fmax(beta * Ur * pd / nuf, kRootVSmall);
  double Cds = 24.0 * (1.0 + 0.15 * pow(Re, 0.687)) / Re;
  if (Re > 1000.0) Cds = 0.44;
  double K = 0.75 * Cds * rhof * Ur * bp / pd;  // Wen-Yu
  if (beta <= 0.8) {                            // Ergun
    const double bd = beta * pd;
    K = 150.0 * alpha * nuf * rhof / (bd * bd) + 1.75 * rhof * Ur / (beta * pd);
  }
  return K;
}

__device__ __forceinline__ double jd_syamlal_obrien(double Ur, double alpha, double pd, double nuf, double rhof)
{
  const double beta = fmax(1.0 - alpha, kRootVSmall);
  const double Ai = pow(beta, 4.14);
  double Bi = 0.8 * pow(beta, 1.28);
  if (beta > 0.85) Bi = pow(beta, 2.65);
  const double Re = fmax(Ur * pd / nuf, kRootVSmall);
  const double a = 0.06 * Re;
  const double Vr = 0.5 * (Ai - 0.06 * Re + sqrt(a * a + 0.12 * Re * (2.0 * Bi - Ai) + Ai * Ai));
  const double s = 0.63 + 4.8 * sqrt(Vr / Re);
  return 0.75 * (s * s) * rhof * Ur / (pd * (Vr * Vr));
}

// NoCorrection::Jd  dragModels/NoCorrection/NoCorrection.C:85-146
__device__ __forceinline__ double jd_no_correction(double Ur, double alpha, double pd, double nuf, double rhof)
{
  const double beta = fmax(1.0 - alpha, 1.0e-6);
  const double Ai = pow(beta, 4.14);
  double Bi = 0.8 * pow(beta, 1.28);
  if (beta > 0.85) Bi = pow(beta, 2.65);
  const double Re = fmax(Ur * pd / nuf, 1.0e-3);
  const double a = 0.06 * Re;
  const double Vr = 0.5 * (Ai - 0.06 * Re + sqrt(a * a + 0.12 * Re * (2.0 * Bi - Ai) + Ai * Ai));
  const double Cds = 24 * 1.0 / Re + 4.0 * pow(Re, -0.5) + 0.4;
  return 0.75 * Cds * rhof * Ur / (pd * (Vr * Vr));
}

// the run-time selection table of dragModel::New (newDragModel.C:31-64): 0 ErgunWenYu, 1 SyamlalOBrien, 2 NoCorrection
__device__ __forceinline__ double jd_model(int model, double Ur, double alpha, double pd, double nuf, double rhof)
{
  return model == 0 ? jd_ergun_wenyu(Ur, alpha, pd, nuf, rhof)
                    : (model == 2 ? jd_no_correction(Ur, alpha, pd, nuf, rhof)
                                  : jd_syamlal_obrien(Ur, alpha, pd, nuf, rhof));
}

__global__ __launch_bounds__(256) void k_jd(int model, int n, const double* Ur, const double* alpha,
                                            const double* pd, double nuf, double rhof, double* Jd)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) Jd[i] = jd_model(model, Ur[i], alpha[i], pd[i], nuf, rhof);
}

struct MeshDev {
  double origin[3], dx[3];
  int n[3];
  int ncells;
  const double* f[3];   // graded axis: n+1 ascending face coordinates (device); nullptr: uniform origin + i dx
  int per[3];           // cyclic patch pair along a (uniform) axis: a position one period outside is tracked through
                        // the cyclic face into the cell on the other side ([3P] cyclicPolyPatch; LAMMPS wraps its
                        // periodic coordinates only when it reneighbours, so positions sit slightly outside in between)
};

// cell index along one axis, -1 outside: floor((x - origin)/dx) on a uniform axis, the interval [f[i], f[i+1]) that
// holds x on a graded one (the same comparisons as the oracle's search: bit-exact owners)
__device__ __forceinline__ int axis_cell(const MeshDev& m, int k, double x)
{
  const int n = m.n[k];
  const double* f = m.f[k];
  if (!f) {
    double fl = floor((x - m.origin[k]) / m.dx[k]);
    if (m.per[k]) {
      if (fl < 0.0 && fl >= -(double)n) fl += (double)n;
      else if (fl >= (double)n && fl < 2.0 * (double)n) fl -= (double)n;
    }
    if (fl < 0.0 || fl >= (double)n) return -1;
    return (int)fl;
  }
  if (!(x >= f[0]) || !(x < f[n])) return -1;
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (x >= f[mid]) lo = mid;
    else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ int cell_of(const MeshDev& m, double x, double y, double z)
{
  const int ix = axis_cell(m, 0, x), iy = axis_cell(m, 1, y), iz = axis_cell(m, 2, z);
  if (ix < 0 || iy < 0 || iz < 0) return -1;
  return ix + m.n[0] * (iy + m.n[1] * iz);
}

__global__ __launch_bounds__(256) void k_cell_owner_aos(int n, const double* x, MeshDev m, int* cell)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) cell[i] = cell_of(m, x[3 * i], x[3 * i + 1], x[3 * i + 2]);
}

struct CloudFlagsDev {
  int dragModel, particleDrag, particlePressureGrad, particleBuoyancy, particleAddedMass, particleLift,
      lubricationForce;
  double g[3], rhob, nub, deltaT;
  int inletOption;   // 0: no inlet override ; 1 box ; 2 hollow cylinder (addParticleOption with a non-zero inletForce)
  double inletForce[3], inletBox[9], ecc[3];
};

// softParticleCloud::pointInRegion  softParticleCloud.C:1354-1417 (option 1: box, faces included; option 2: between the
// cylinders r1 -- shifted by the eccentricity -- and r2 around the axis (x1,y1,z1)-(x2,y2,z2); as there, the inner
// distance uses the projection of the UNshifted point)
__device__ __forceinline__ bool point_in_region(int option, const double* b, const double* ecc, double px, double py,
                                                double pz)
{
  const double x1 = b[0], x2 = b[1], y1 = b[2], y2 = b[3], z1 = b[4], z2 = b[5], r1 = b[6], r2 = b[7];
  if (option == 1)
    return (px - x1) * (px - x2) < kRootVSmall && (py - y1) * (py - y2) < kRootVSmall && (pz - z1) * (pz - z2) < kRootVSmall;
  if (option == 2) {
    const double a[3] = {x2 - x1, y2 - y1, z2 - z1}, q[3] = {px - x1, py - y1, pz - z1};
    const double h = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    const double dot = a[0] * q[0] + a[1] * q[1] + a[2] * q[2];
    if (dot < 0.0 || dot > pow(h, 2)) return false;
    const double qe[3] = {q[0] - ecc[0], q[1] - ecc[1], q[2] - ecc[2]};
    const double dsq = (q[0] * q[0] + q[1] * q[1] + q[2] * q[2]) - dot * dot / pow(h, 2);
    const double dsqE = (qe[0] * qe[0] + qe[1] * qe[1] + qe[2] * qe[2]) - dot * dot / pow(h, 2);
    return dsqE > r1 * r1 && dsq < r2 * r2;
  }
  return false;
}

// updateParticleUr + updateParticleAlpha + Jd + updateDragOnParticles, one owned atom per lane.
// pDrag goes straight into fix fdrag's [3][cap] array (what lammps_put_local_info would copy).
__global__ __launch_bounds__(256) void k_drag_on_particles(
    int n, size_t cap, const double4* xr, const double4* vm, const int* tag, MeshDev m, CloudFlagsDev fl,
    const double* gamma, const double* UfS, const double* gradp, const double* DDtUf, const double* curlU,
    double* pstate, int maxtag, int* cell_bytag, double* Jd_bytag, double* pDrag_bytag,
    double* fdrag, int timeIndex, const double* UfSold)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double4 x = xr[i], v = vm[i];
  const int c = cell_of(m, x.x, x.y, x.z);
  const int t = tag[i];
  double F[3] = {0.0, 0.0, 0.0};
  double jd = 0.0;
  const double U[3] = {v.x, v.y, v.z};
  // per-particle state that stays with its atom through re-sorts and migration (DemEngine::register_extra):
  // rows 0-2 UOld (softParticle.H:95), 3-5 sumDeltaFb, 6 n0 (:104-107).  A particle the cloud has not seen yet holds
  // NaN in UOld: its UOld is its U (softParticle.C:74, UOld_ = U_ at construction)
  double* const st = pstate + i;
  double uold[3] = {st[0], st[cap], st[2 * cap]};
  if (uold[0] != uold[0])
    for (int k = 0; k < 3; k++) uold[k] = U[k];
  for (int k = 0; k < 3; k++) st[(size_t)k * cap] = U[k];   // setPositionVeloCpuId: UOld = U
  // pDuDt = DDtUf[c] (enhancedCloud.C:155) is handed to lammps_put_local_info, which drops it (library.cpp:314-367):
  // fix fdrag's DuDt array stays at the 0 it was created with (fix_fluid_drag.cpp:91), so the in-LAMMPS added-mass
  // term of `fix fdrag <carrier_rho>` sees DuDt = 0 on this path exactly as through sf_lammps_put_local_info
  if (c >= 0) {
    const double d = 2.0 * x.w;
    const double Vol = kPi * d * d * d / 6.0;  // softParticle.H:270-273
    const double alpha = gamma[c];
    double Uri[3];
    for (int k = 0; k < 3; k++) Uri[k] = UfS[3 * c + k] - U[k];
    const double mag = sqrt(Uri[0] * Uri[0] + Uri[1] * Uri[1] + Uri[2] * Uri[2]);
    jd = jd_model(fl.dragModel, mag, alpha, d, fl.nub, fl.rhob);
    if (fl.particleDrag)
      for (int k = 0; k < 3; k++) F[k] += jd * (1.0 - alpha) * Vol * Uri[k];
    if (fl.particlePressureGrad)
      for (int k = 0; k < 3; k++) F[k] += -gradp[3 * c + k] * Vol;
    if (fl.particleBuoyancy)
      for (int k = 0; k < 3; k++) F[k] += -fl.g[k] * fl.rhob * Vol;
    if (fl.particleAddedMass) {
      double acc[3], mm = 0.0;
      for (int k = 0; k < 3; k++) {
        acc[k] = DDtUf[3 * c + k] - (U[k] - uold[k]) / fl.deltaT;
        mm += acc[k] * acc[k];
      }
      mm = sqrt(mm);
      if (mm > 10)
        for (int k = 0; k < 3; k++) acc[k] = acc[k] / (mm + kRootVSmall) * 10;
      for (int k = 0; k < 3; k++) F[k] += 0.5 * fl.rhob * Vol * acc[k];
    }
    if (fl.particleLift) {
      const double w0 = curlU[3 * c], w1 = curlU[3 * c + 1], w2 = curlU[3 * c + 2];
      const double cr[3] = {Uri[1] * w2 - Uri[2] * w1, Uri[2] * w0 - Uri[0] * w2, Uri[0] * w1 - Uri[1] * w0};
      const double magw = sqrt(w0 * w0 + w1 * w1 + w2 * w2);
      for (int k = 0; k < 3; k++)
        F[k] += 1.6 * fl.rhob * sqrt(fl.nub) * (d * d) * cr[k] / sqrt(magw + kRootVSmall);
    }
    if (timeIndex >= 0) {
      // reduced-order history (Basset) force, Elghannay & Tafti 2016 -- enhancedCloud.C:197-233; per-particle state
      // sumDeltaFb / n0 (softParticle.H:104-107)
      const double tau_d = d * d / fl.nub;
      double m1 = 0.0, m2 = 0.0;
      for (int k = 0; k < 3; k++) {
        const double o = UfSold[3 * c + k] - uold[k];
        m1 += Uri[k] * Uri[k];
        m2 += o * o;
      }
      const double ReP = sqrt(m1) * d / fl.nub, RePOld = sqrt(m2) * d / fl.nub;
      const double a1 = 0.632 / (ReP + kRootVSmall) + 0.087, a2 = 0.632 / (RePOld + kRootVSmall) + 0.087;
      const double tau_h = tau_d * (a1 * a1), tau_h_old = tau_d * (a2 * a2);
      const double Cb = -1.5 * (d * d) * fl.rhob * pow(3.1416 * fl.nub, 0.5);
      const double n0 = st[6 * cap];
      const double tau_t = fl.deltaT * (timeIndex - n0);
      double sfb[3], dfb[3];
      for (int k = 0; k < 3; k++) {
        sfb[k] = st[(size_t)(3 + k) * cap];
        dfb[k] = Cb * ((U[k] - uold[k]) / fl.deltaT) / sqrt(fl.deltaT);
      }
      double dnh;
      if (tau_t < tau_h) {
        dnh = timeIndex - n0;
        for (int k = 0; k < 3; k++) sfb[k] = sfb[k] + dfb[k];
      } else {
        dnh = tau_h / fl.deltaT;
        for (int k = 0; k < 3; k++) {
          sfb[k] = tau_h / tau_h_old * sfb[k];
          sfb[k] = (dnh - 1) / dnh * sfb[k];
          sfb[k] = sfb[k] + dfb[k];
        }
        st[6 * cap] = timeIndex - dnh;
      }
      const double g1 = dnh < 1 ? 0.9279 : 0.9279 * (2 * dnh - 1) / dnh * pow(dnh, -dnh / (2 * dnh - 1)) + 0.001531;
      for (int k = 0; k < 3; k++) {
        st[(size_t)(3 + k) * cap] = sfb[k];
        F[k] += (g1 * sfb[k]) * fl.deltaT;
      }
    }
    if (fl.lubricationForce) {
      const double distMin = 0.0001 * d, distMax = 0.1 * d;
      const double distWall = x.y - 0.5 * d;
      if (distWall < distMax && distWall > distMin)
        F[1] += 6 * 3.1416 * fl.nub * fl.rhob * (-U[1]) / distWall * (d * d) / 4.0 * 1.0;
    }
    // inlet: the assembled force is replaced by what brings the particle to inletForce within one time step (:249-257)
    if (fl.inletOption && point_in_region(fl.inletOption, fl.inletBox, fl.ecc, x.x, x.y, x.z))
      for (int k = 0; k < 3; k++) F[k] = v.w * (fl.inletForce[k] - U[k]) / fl.deltaT;
  }
  for (int k = 0; k < 3; k++) fdrag[(size_t)k * cap + i] = F[k];
  // diagnostics kept by tag (the DEM engine may re-sort its atoms at any neighbour rebuild)
  if (t >= 1 && t <= maxtag) {
    cell_bytag[t - 1] = c;
    Jd_bytag[t - 1] = jd;
    for (int k = 0; k < 3; k++) pDrag_bytag[(size_t)k * maxtag + (t - 1)] = F[k];
  }
}

// enhancedCloud::averageInfo  enhancedCloud.C:1341-1370: sum of Vol and Vol*U over the particles.  One partial sum
// per block (fixed shuffle tree), added on the host in block order: deterministic.
__global__ __launch_bounds__(256) void k_average_info(int n, const double4* xr, const double4* vm, double* partial)
{
  __shared__ double ws[4][4];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double a[4] = {0.0, 0.0, 0.0, 0.0};
  if (i < n) {
    const double d = 2.0 * xr[i].w;
    const double Vol = kPi * d * d * d / 6.0;
    const double4 v = vm[i];
    a[0] = Vol;
    a[1] = Vol * v.x;
    a[2] = Vol * v.y;
    a[3] = Vol * v.z;
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int k = 0; k < 4; k++) {
    double t = a[k];
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
    if (lane == 0) ws[w][k] = t;
  }
  __syncthreads();
  if (threadIdx.x < 4) partial[4 * blockIdx.x + threadIdx.x] = ws[0][threadIdx.x] + ws[1][threadIdx.x] + ws[2][threadIdx.x] + ws[3][threadIdx.x];
}

// sort keys for the particle -> cell scatter: cell id (outside = ncells, sorts last)
__global__ __launch_bounds__(256) void k_cell_keys(int n, const double4* xr, MeshDev m, unsigned* keys, int* idx)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double4 x = xr[i];
  const int c = cell_of(m, x.x, x.y, x.z);
  keys[i] = (unsigned)(c < 0 ? m.ncells : c);
  idx[i] = i;
}

// ---- counting sort of the particles by cell (replaces a 25-launch merge sort per CFD step) ----
// The atoms are already in spatial order, so the 64 lanes of a wave fall into a handful of cells: lanes of the same
// cell are grouped with ballots and ONE atomic per group is issued (a same-address atomic per particle would
// serialise across the XCDs).  grp_leader / grp_rank / grp_size describe the lane's group.
__device__ __forceinline__ void wave_groups(const int c, const bool valid, int& leader, int& rank, int& size)
{
  const int lane = threadIdx.x & 63;
  bool done = !valid;
  leader = -1;
  rank = size = 0;
  for (;;) {
    const unsigned long long rem = __ballot(!done);
    if (!rem) break;
    const int l = __ffsll((long long)rem) - 1;
    const int cl = __shfl(c, l, 64);
    const bool mine = !done && c == cl;
    const unsigned long long m = __ballot(mine);
    if (mine) {
      leader = l;
      rank = __popcll(m & ((1ull << lane) - 1ull));
      size = __popcll(m);
      done = true;
    }
  }
}

// pass 1: cell id of every particle (outside the mesh = ncells, the last bin) and the particle count per cell
__global__ __launch_bounds__(256) void k_cell_count(int n, const double4* xr, MeshDev m, unsigned* cellid, int* count)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < n;
  int c = 0;
  if (valid) {
    const double4 x = xr[i];
    c = cell_of(m, x.x, x.y, x.z);
    if (c < 0) c = m.ncells;
    cellid[i] = (unsigned)c;
  }
  int leader, rank, size;
  wave_groups(c, valid, leader, rank, size);
  if (valid && (int)(threadIdx.x & 63) == leader) atomicAdd(&count[c], size);
}

// pass 2: cursor[c] starts at cstart[c] and ends at cend[c]; the slots inside a cell are handed out in arrival order
__global__ __launch_bounds__(256) void k_cell_place(int n, const unsigned* cellid, int* cursor, int* order)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < n;
  const int c = valid ? (int)cellid[i] : 0;
  int leader, rank, size;
  wave_groups(c, valid, leader, rank, size);
  int base = 0;
  if (valid && (int)(threadIdx.x & 63) == leader) base = atomicAdd(&cursor[c], size);
  base = __shfl(base, leader < 0 ? 0 : leader, 64);
  if (valid) order[base + rank] = i;
}

// pass 3: arrival order -> ascending particle index inside every cell, so that the per-cell sums are the same bits on
// every run.  One wave per cell: rank sort through shuffles (<= 64 particles), through memory otherwise.
__global__ __launch_bounds__(256) void k_cell_sort_segments(int nbins, const int* cstart, const int* cend, int* order,
                                                            int* scratch)
{
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (c >= nbins) return;
  const int s = cstart[c], len = cend[c] - s;
  if (len <= 1) return;
  if (len <= 64) {
    const int v = lane < len ? order[s + lane] : 0x7fffffff;
    int r = 0;
    for (int k = 0; k < len; k++) r += __shfl(v, k, 64) < v ? 1 : 0;
    if (lane < len) order[s + r] = v;
    return;
  }
  for (int a = lane; a < len; a += 64) {
    const int v = order[s + a];
    int r = 0;
    for (int k = 0; k < len; k++) r += order[s + k] < v ? 1 : 0;
    scratch[s + r] = v;
  }
  // (the wave reads `order` of its own segment only; copy back once every lane has written its ranks)
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
  __builtin_amdgcn_wave_barrier();
  for (int a = lane; a < len; a += 64) order[s + a] = scratch[s + a];
}

__global__ __launch_bounds__(256) void k_cell_ranges(const unsigned* keys, int n, int* cstart, int* cend)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned b = keys[i];
  if (i == 0 || keys[i - 1] != b) cstart[b] = i;
  if (i == n - 1 || keys[i + 1] != b) cend[b] = i + 1;
}

// particleToEulerianField: one cell per lane walks its (cell-sorted) particles -- deterministic,
// no atomics.  gamma = sum Vol / V ; Ue = sum Vol U / V, then Ue /= gamma where gamma > ROOTVSMALL.
// L lanes (1, 8 or 64) share a cell: they fetch the particles' records in parallel -- the latency of the gathers is
// what a one-lane-per-cell walk spends its time on -- and then add the contributions one by one in the cell's
// particle order, every lane the same sum, so the result is bit-for-bit the sequential one.
template <int L>
__global__ __launch_bounds__(256) void k_particle_to_eulerian(int ncells, const int* cstart, const int* cend,
                                                              const int* order, const double4* xr,
                                                              const double4* vm, const double* V, double* gamma,
                                                              double* Ue)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = t / L, q = t % L;
  if (c >= ncells) return;
  double g = 0.0, u0 = 0.0, u1 = 0.0, u2 = 0.0;
  const int ks = cstart[c], ke = cend[c];
  for (int base = ks; base < ke; base += L) {
    const int k = base + q;
    double Vol = 0.0;
    double4 v = {0.0, 0.0, 0.0, 0.0};
    if (k < ke) {
      const int i = order[k];
      const double d = 2.0 * xr[i].w;
      Vol = kPi * d * d * d / 6.0;
      v = vm[i];
    }
    const int m = ke - base < L ? ke - base : L;
    for (int s = 0; s < m; s++) {
      const double Vs = L == 1 ? Vol : __shfl(Vol, s, L);
      g += Vs;
      u0 += Vs * (L == 1 ? v.x : __shfl(v.x, s, L));
      u1 += Vs * (L == 1 ? v.y : __shfl(v.y, s, L));
      u2 += Vs * (L == 1 ? v.z : __shfl(v.z, s, L));
    }
  }
  if (q != 0) return;
  const double Vc = V[c];
  g /= Vc;
  u0 /= Vc;
  u1 /= Vc;
  u2 /= Vc;
  gamma[c] = g;   // Ue /= gamma happens after the (optional) smoothing of both fields (:944-962)
  Ue[3 * c] = u0;
  Ue[3 * c + 1] = u1;
  Ue[3 * c + 2] = u2;
}

// calcTcFields: Asrc[c] = sum (Vol Jd / V)(U - UfSmoothed[c]) ; Omega accumulated then zeroed (:391)
template <int L>
__global__ __launch_bounds__(256) void k_calc_tc(int ncells, const int* cstart, const int* cend, const int* order,
                                                 const double4* xr, const double4* vm, const double* V,
                                                 const double* gamma, const double* UfS, int dragModel,
                                                 double nub, double rhob, double* Asrc, double* Omega,
                                                 const int* tag, double* Jd_bytag, int maxtag)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = t / L, q = t % L;
  if (c >= ncells) return;
  const double alpha = gamma[c];
  const double uf[3] = {UfS[3 * c], UfS[3 * c + 1], UfS[3 * c + 2]};
  const double Vc = V[c];
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  const int ks = cstart[c], ke = cend[c];
  for (int base = ks; base < ke; base += L) {
    const int k = base + q;
    double omg = 0.0;
    double4 v = {0.0, 0.0, 0.0, 0.0};
    if (k < ke) {
      const int i = order[k];
      const double d = 2.0 * xr[i].w;
      const double Vol = kPi * d * d * d / 6.0;
      v = vm[i];
      const double r0 = uf[0] - v.x, r1 = uf[1] - v.y, r2 = uf[2] - v.z;
      const double mag = sqrt(r0 * r0 + r1 * r1 + r2 * r2);
      const double jd = jd_model(dragModel, mag, alpha, d, nub, rhob);
      const int tg = tag[i];
      if (tg >= 1 && tg <= maxtag) Jd_bytag[tg - 1] = jd;
      omg = Vol * jd / Vc;
    }
    const int m = ke - base < L ? ke - base : L;
    for (int s = 0; s < m; s++) {
      const double os = L == 1 ? omg : __shfl(omg, s, L);
      a0 += os * ((L == 1 ? v.x : __shfl(v.x, s, L)) - uf[0]);
      a1 += os * ((L == 1 ? v.y : __shfl(v.y, s, L)) - uf[1]);
      a2 += os * ((L == 1 ? v.z : __shfl(v.z, s, L)) - uf[2]);
    }
  }
  if (q != 0) return;
  // weighted by (1 - gamma) for the smoothing step (:407-408); k_unweight divides again (:415-416)
  const double w = 1 - alpha;
  Asrc[3 * c] = a0 * w;
  Asrc[3 * c + 1] = a1 * w;
  Asrc[3 * c + 2] = a2 * w;
  Omega[c] = 0.0;
}

// Ue /= gamma where gamma > ROOTVSMALL (:955-962)
__global__ __launch_bounds__(256) void k_divide_ue(int ncells, const double* gamma, double* Ue)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncells) return;
  const double g = gamma[c];
  if (g > kRootVSmall)
    for (int k = 0; k < 3; k++) Ue[3 * c + k] /= g;
}

// f *= (1 - gamma) (mode 0) or f /= (1 - gamma) (mode 1) on a vector field; mode 2: dst = src
__global__ __launch_bounds__(256) void k_weight(int ncells, int mode, const double* gamma, double* f, const double* src)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncells) return;
  if (mode == 2) {
    for (int k = 0; k < 3; k++) f[3 * c + k] = src[3 * c + k];
    return;
  }
  const double w = 1 - gamma[c];
  for (int k = 0; k < 3; k++) f[3 * c + k] = mode == 0 ? f[3 * c + k] * w : f[3 * c + k] / w;
}

__global__ __launch_bounds__(256) void k_cap_alpha(int ncells, double* gamma, double maxAlpha)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < ncells && gamma[c] > maxAlpha) gamma[c] = maxAlpha;
}

int adjust_timestep(double deltaT, double dtLampIn, int subCycles_in, double* dtLampAdj, int* solidStepsPerDt,
                    int* subCycles, int* subSteps)
{
  double dnSub = std::round(deltaT / dtLampIn);  // softParticleCloud.C:216
  if (dnSub == 0) dnSub++;
  int sc = subCycles_in;
  const int steps = ((int)dnSub / sc) * sc;      // :219-221
  *dtLampAdj = deltaT / dnSub;                   // :224
  if (sc >= steps) {                             // :229-234
    sc = steps;
    *subSteps = 1;
  } else {
    *subSteps = steps / sc;                      // :237-238
    if (steps % sc != 0) return -1;              // FatalError :239-247
  }
  *solidStepsPerDt = steps;
  *subCycles = sc;
  return 0;
}

class Cloud {
 public:
  Cloud(SfLammps* lmp, const sf_cloud_mesh& mesh, const sf_cloud_props& props, double deltaT)
      : lmp_(lmp), props_(props), deltaT_(deltaT)
  {
    for (int k = 0; k < 3; k++) {
      mesh_.origin[k] = mesh.origin[k];
      mesh_.dx[k] = mesh.dx[k];
      mesh_.n[k] = mesh.n[k];
      mesh_.f[k] = nullptr;
      // (a slab block is not periodic along x itself: its ghost layers are the way through the cyclic face)
      mesh_.per[k] = (mesh.periodic[k] && !mesh.faces[k] && !(k == 0 && mesh.slab_nx_global > 0)) ? 1 : 0;
    }
    mesh_.ncells = mesh.n[0] * mesh.n[1] * mesh.n[2];
    if (mesh_.ncells <= 0) fail("cloud mesh has no cells");
    {
      // UOld (NaN = "not seen yet": UOld = U, softParticle.C:74), sumDeltaFb, n0: they live in the engine so that
      // they follow their atom through re-sorts and, on a decomposed domain, in the migrate record
      const double nan = std::nan("");
      const double init[7] = {nan, nan, nan, 0.0, 0.0, 0.0, 0.0};
      xrow_ = lmp->eng.register_extra(7, init);
    }
    if (mesh.cell_label) {
      label_.assign(mesh.cell_label, mesh.cell_label + mesh_.ncells);
      std::vector<char> seen(mesh_.ncells, 0);
      for (int c = 0; c < mesh_.ncells; c++) {
        const int l = label_[c];
        if (l < 0 || l >= mesh_.ncells || seen[l]) fail("cloud mesh: cell_label is not a permutation of 0..%d", mesh_.ncells - 1);
        seen[l] = 1;
      }
    }
    // cell widths per axis (uniform or from the face coordinates of a graded block)
    std::vector<double> width[3];
    for (int k = 0; k < 3; k++) {
      width[k].assign(mesh.n[k], mesh.dx[k]);
      if (mesh.faces[k]) {
        for (int i = 0; i < mesh.n[k]; i++) {
          width[k][i] = mesh.faces[k][i + 1] - mesh.faces[k][i];
          if (!(width[k][i] > 0.0)) fail("cloud mesh: face coordinates along axis %d are not ascending", k);
        }
        SF_HIP(hipMalloc(&faces_dev_[k], sizeof(double) * (mesh.n[k] + 1)));
        SF_HIP(hipMemcpy(faces_dev_[k], mesh.faces[k], sizeof(double) * (mesh.n[k] + 1), hipMemcpyHostToDevice));
        mesh_.f[k] = faces_dev_[k];
        mesh_.origin[k] = mesh.faces[k][0];
      } else if (!(mesh.dx[k] > 0.0))
        fail("cloud mesh: dx[%d] must be positive", k);
    }
    DemEngine& e = lmp_->eng;
    s_ = e.stream();
    // adjustLampTimestep
    double dtadj;
    int steps;
    if (adjust_timestep(deltaT, e.timestep(), props.subCycles < 1 ? 1 : props.subCycles, &dtadj, &steps,
                        &subCycles_, &subSteps_) != 0)
      fail("softParticleCloud::adjustLampTimestep() Time step adjustment error.");
    e.set_timestep(dtadj);
    const size_t nc = (size_t)mesh_.ncells;
    auto alloc = [&](double*& p, size_t n) {
      SF_HIP(hipMalloc(&p, sizeof(double) * n));
      SF_HIP(hipMemsetAsync(p, 0, sizeof(double) * n, s_));
    };
    alloc(V_, nc);
    alloc(gamma_, nc);
    alloc(Ue_, 3 * nc);
    alloc(Asrc_, 3 * nc);
    alloc(Omega_, nc);
    alloc(Uf_, 3 * nc);
    alloc(DDtUf_, 3 * nc);
    alloc(gradp_, 3 * nc);
    alloc(curlU_, 3 * nc);
    alloc(UfS_, 3 * nc);
    alloc(UfSold_, 3 * nc);
    const double* wptr[3] = {mesh.faces[0] ? width[0].data() : nullptr, mesh.faces[1] ? width[1].data() : nullptr,
                             mesh.faces[2] ? width[2].data() : nullptr};
    smoother_.configure(mesh.n, mesh.dx, props.smoothDirection, props.diffusionBandWidth, props.diffusionSteps, s_,
                        wptr, mesh.periodic);
    slab_nx_global_ = mesh.slab_nx_global;
    slab_per_x_ = mesh.periodic[0] ? 1 : 0;
    if (mesh.slab_nx_global > 0) {
      // this block is one x-slab of a larger mesh plus a ghost layer on each side (the mesh partitioned by
      // the particle slab planes)
      if (mesh.faces[0]) fail("a slab mesh must be uniform along x");
      if (mesh.n[0] < 3) fail("a slab mesh needs at least one owned cell layer between its two ghost layers");
      smoother_.configure_slab(mesh.slab_nx_global);
    }
    SF_HIP(hipMalloc(&cstart_, sizeof(int) * 2 * (nc + 1)));
    std::vector<double> hV(nc);
    for (int iz = 0; iz < mesh.n[2]; iz++)
      for (int iy = 0; iy < mesh.n[1]; iy++)
        for (int ix = 0; ix < mesh.n[0]; ix++)
          hV[(size_t)ix + (size_t)mesh.n[0] * (iy + (size_t)mesh.n[1] * iz)] = width[0][ix] * width[1][iy] * width[2][iz];
    SF_HIP(hipMemcpyAsync(V_, hV.data(), sizeof(double) * nc, hipMemcpyHostToDevice, s_));
    SF_HIP(hipStreamSynchronize(s_));
    if (!e.is_setup()) e.setup();  // lammps_step(0) at construction, softParticleCloud.C:189
    if (smoother_.slab()) {
      scatter_local();             // (slab mesh: the caller moves the ghost-layer sums and runs phases 3 and 6)
    } else {
      particle_to_eulerian();      // enhancedCloud.C:635
      update_uf_smoothed();        // :641-655
    }
    SF_HIP(hipStreamSynchronize(s_));
  }

  ~Cloud()
  {
    if (xrow_ >= 0) lmp_->eng.unregister_extra(xrow_, 7);
    for (double* p : {V_, gamma_, Ue_, Asrc_, Omega_, Uf_, DDtUf_, gradp_, curlU_, Jd_, pDragT_, UfS_, UfSold_})
      if (p) (void)hipFree(p);
    for (void* p : {(void*)cstart_, (void*)cell_, (void*)keys_, (void*)keys2_, (void*)idx_, (void*)idx2_, sort_tmp_,
                    (void*)faces_dev_[0], (void*)faces_dev_[1], (void*)faces_dev_[2]})
      if (p) (void)hipFree(p);
  }

  // host arrays of the caller are in OpenFOAM label order, the device fields in grid order (label_ empty: the same)
  void upload_field(double* dev, const double* host, int ncomp)
  {
    const size_t nc = (size_t)mesh_.ncells;
    if (label_.empty()) {
      SF_HIP(hipMemcpyAsync(dev, host, sizeof(double) * ncomp * nc, hipMemcpyHostToDevice, s_));
      return;
    }
    std::vector<double> tmp(ncomp * nc);
    for (size_t c = 0; c < nc; c++)
      for (int k = 0; k < ncomp; k++) tmp[c * ncomp + k] = host[(size_t)label_[c] * ncomp + k];
    SF_HIP(hipMemcpyAsync(dev, tmp.data(), sizeof(double) * ncomp * nc, hipMemcpyHostToDevice, s_));
    SF_HIP(hipStreamSynchronize(s_));
  }
  void download_field(double* host, const double* dev, int ncomp)
  {
    const size_t nc = (size_t)mesh_.ncells;
    if (label_.empty()) {
      SF_HIP(hipMemcpyAsync(host, dev, sizeof(double) * ncomp * nc, hipMemcpyDeviceToHost, s_));
      return;
    }
    std::vector<double> tmp(ncomp * nc);
    SF_HIP(hipMemcpyAsync(tmp.data(), dev, sizeof(double) * ncomp * nc, hipMemcpyDeviceToHost, s_));
    SF_HIP(hipStreamSynchronize(s_));
    for (size_t c = 0; c < nc; c++)
      for (int k = 0; k < ncomp; k++) host[(size_t)label_[c] * ncomp + k] = tmp[c * ncomp + k];
  }

  void set_fluid(const double* Uf, const double* DDtUf, const double* gradp, const double* curlU)
  {
    if (Uf) upload_field(Uf_, Uf, 3);
    if (DDtUf) upload_field(DDtUf_, DDtUf, 3);
    if (gradp) upload_field(gradp_, gradp, 3);
    if (curlU) upload_field(curlU_, curlU, 3);
    // before the first step the fields are the initial condition: UfSmoothed_ (whose oldTime() the history force
    // reads in the first step) is built from them, as the reference does at construction (:641-655)
    if (Uf && time_index_ == 0 && !smoother_.slab()) update_uf_smoothed();   // (slab mesh: the caller runs phase 6)
    SF_HIP(hipStreamSynchronize(s_));
  }

  void evolve()
  {
    if (smoother_.slab()) fail("sf_cloud_evolve on a slab mesh: drive it through sf_cloud_phase");
    Range r_evolve("evolve");   // roctx ranges carry the bucket names of writeCPUTime.H:1-19
    const double t0 = now();
    DemEngine& e = lmp_->eng;
    advance_time();
    update_uf_smoothed();   // :675-690
    for (int k = 0; k < subCycles_; k++) {
      double t1 = now();
      {
        Range r("foam->lammps");   // updateDragOnParticles + what lammps_put_local_info would copy
        drag_on_particles();
        t_.dragOnParticles += sync_now() - t1;
      }
      t1 = now();
      {
        Range r("lammps");
        e.run(subSteps_);  // lammpsEvolveForward without the host round trip
        t_.lammps += sync_now() - t1;
      }
      // Cloud::move: the new cell owner is recomputed from the DEM positions wherever it is used
      if (k == 0) {
        Range r("particle move");   // cell owner + particleToEulerianField (enhancedCloud.C:749-776)
        t1 = now();
        particle_to_eulerian();
        t_.scatter += sync_now() - t1;
      }
    }
    t_.evolve += sync_now() - t0;
  }

  void calc_tc_fields()
  {
    if (smoother_.slab()) fail("sf_cloud_calc_tc_fields on a slab mesh: drive it through sf_cloud_phase");
    Range r_tc("calcTcField");
    const double t0 = now();
    calc_tc_local();
    calc_tc_finish();
    t_.calcTc += sync_now() - t0;
  }

  // per-cell sums over THIS engine's particles (linear in the particles: a decomposed domain adds the ranks up)
  void calc_tc_local()
  {
    DemEngine& e = lmp_->eng;
    const int n = e.nlocal();
    ensure_particle_arrays();
    // liftDragCoeffs.H:6-14: alpha capped before calcTcFields
    if (props_.maxPossibleAlpha > 0.0)
      k_cap_alpha<<<div_up(mesh_.ncells, 256), 256, 0, s_>>>(mesh_.ncells, gamma_, props_.maxPossibleAlpha);
    sort_by_cell(n);
    const int L = lanes_per_cell(n);
    auto launch = [&](auto kern) {
      kern<<<div_up((size_t)mesh_.ncells * L, 256), 256, 0, s_>>>(mesh_.ncells, cstart_, cstart_ + mesh_.ncells + 1,
                                                                  idx2_, e.d_xr(), e.d_vm(), V_, gamma_, UfS_,
                                                                  props_.dragModel, props_.nub, props_.rhob, Asrc_,
                                                                  Omega_, e.d_tag(), Jd_, maxtag_);
    };
    if (L == 64) launch(k_calc_tc<64>);
    else if (L == 8) launch(k_calc_tc<8>);
    else launch(k_calc_tc<1>);
  }

  // lanes that share one cell in the per-cell sums: by the mean number of particles per cell
  int lanes_per_cell(int n) const
  {
    static const int env = getenv("SF_CELL_LANES") ? atoi(getenv("SF_CELL_LANES")) : 0;
    if (env) return env;
    const double per_cell = (double)n / (double)std::max(mesh_.ncells, 1);
    // measured at 37 particles per cell (1 M particles, 29x32x29 cells): k_calc_tc 130 / 46 / 96 us and
    // k_particle_to_eulerian 62 / 26 / 91 us with 1 / 8 / 64 lanes per cell
    return per_cell >= 256.0 ? 64 : (per_cell >= 2.0 ? 8 : 1);
  }

  // smoothing of up to two fields: the whole solve, or -- on a slab mesh -- its first half (returns true: paused)
  // when `resume` is false and its second half when it is true
  bool smooth_step(bool resume, double* fa, int na, double* fb, int nb)
  {
    if (!smoother_.enabled()) return false;
    if (!smoother_.slab()) {
      if (nb) smoother_.smooth2(fa, na, fb, nb);
      else smoother_.smooth(fa, na);
      return false;
    }
    if (!resume) {
      smoother_.begin(fa, na, fb, nb);
      return true;
    }
    smoother_.end();
    return false;
  }

  bool calc_tc_finish(bool resume = false)
  {
    if (props_.dragSmooth && smooth_step(resume, Asrc_, 3, nullptr, 0)) return true;           // :410-413
    k_weight<<<div_up(mesh_.ncells, 256), 256, 0, s_>>>(mesh_.ncells, 1, gamma_, Asrc_, nullptr);   // :415-416
    return false;
  }

  // The pieces of evolve() / calcTcFields() for a decomposed domain (one engine per GPU, the whole mesh replicated on
  // every rank): the caller runs the DEM sub-steps through the halo driver between phase 1 and 2 and sums the
  // per-cell fields over the ranks between the "local" and the "finish" phases.
  // Returns 0 when the phase is complete.  On a mesh partitioned into x-slabs (sf_cloud_mesh.slab_nx_global) the
  // phases that smooth a field (0 / 6, 3, 5) stop after the local half of the implicit diffusion solve and return 1:
  // the caller then transposes the smoother's work array to complete x-lines, calls sf_cloud_smooth_xsolve on them,
  // transposes back (DiffusionSmoother, sf_smooth.h) and calls the SAME phase again to finish it.
  int phase(int ph)
  {
    const bool resume = pending_ == ph;
    if (pending_ >= 0 && !resume) fail("sf_cloud_phase %d: phase %d is waiting for its x solve", ph, pending_);
    pending_ = -1;
    bool paused = false;
    switch (ph) {
      case 0:
        if (!resume) advance_time();
        paused = update_uf_smoothed(resume);
        break;
      case 6: paused = update_uf_smoothed(resume); break;   // (re-)initialisation, no time advance
      case 1: drag_on_particles(); break;
      case 2: scatter_local(); break;
      case 3: paused = scatter_finish(resume); break;
      case 4: calc_tc_local(); break;
      case 5: paused = calc_tc_finish(resume); break;
      default: fail("sf_cloud_phase: unknown phase %d", ph);
    }
    if (paused) pending_ = ph;
    return paused ? 1 : 0;
  }
  DiffusionSmoother& smoother() { return smoother_; }
  int sub_cycles() const { return subCycles_; }
  int sub_steps() const { return subSteps_; }
  void device_fields(double** gamma, double** Ue, double** Asrc, int* ncells)
  {
    *gamma = gamma_;
    *Ue = Ue_;
    *Asrc = Asrc_;
    *ncells = mesh_.ncells;
  }

  // out[0] total particle volume, out[1..3] sum of Vol*U, out[4..6] volume-averaged velocity (:1365 with ROOTVSMALL)
  void average_info(double out[7])
  {
    DemEngine& e = lmp_->eng;
    const int n = e.nlocal();
    for (int k = 0; k < 7; k++) out[k] = 0.0;
    if (n) {
      const int nb = div_up(n, 256);
      double* dpart = nullptr;
      SF_HIP(hipMalloc(&dpart, sizeof(double) * 4 * nb));
      k_average_info<<<nb, 256, 0, s_>>>(n, e.d_xr(), e.d_vm(), dpart);
      std::vector<double> h(4 * (size_t)nb);
      SF_HIP(hipMemcpyAsync(h.data(), dpart, sizeof(double) * 4 * nb, hipMemcpyDeviceToHost, s_));
      SF_HIP(hipStreamSynchronize(s_));
      (void)hipFree(dpart);
      for (int b = 0; b < nb; b++)
        for (int k = 0; k < 4; k++) out[k] += h[4 * (size_t)b + k];
    }
    for (int k = 0; k < 3; k++) out[4 + k] = out[1 + k] / (out[0] + kRootVSmall);
  }

  void get_fields(double* gamma, double* Ue, double* Asrc, double* Omega)
  {
    if (gamma) download_field(gamma, gamma_, 1);
    if (Ue) download_field(Ue, Ue_, 3);
    if (Asrc) download_field(Asrc, Asrc_, 3);
    if (Omega) download_field(Omega, Omega_, 1);
    SF_HIP(hipStreamSynchronize(s_));
  }

  // results of the last drag evaluation, sorted by tag
  void get_particles(int* tag, int* cell, double* pDrag, double* Jd)
  {
    DemEngine& e = lmp_->eng;
    const int n = e.nlocal();
    if (!n) return;
    ensure_particle_arrays();
    std::vector<int> ht(n), hc(maxtag_);
    std::vector<double> hj(maxtag_), hf(3 * (size_t)maxtag_);
    SF_HIP(hipMemcpyAsync(ht.data(), e.d_tag(), sizeof(int) * n, hipMemcpyDeviceToHost, s_));
    SF_HIP(hipMemcpyAsync(hc.data(), cell_, sizeof(int) * maxtag_, hipMemcpyDeviceToHost, s_));
    SF_HIP(hipMemcpyAsync(hj.data(), Jd_, sizeof(double) * maxtag_, hipMemcpyDeviceToHost, s_));
    SF_HIP(hipMemcpyAsync(hf.data(), pDragT_, sizeof(double) * 3 * (size_t)maxtag_, hipMemcpyDeviceToHost, s_));
    SF_HIP(hipStreamSynchronize(s_));
    std::sort(ht.begin(), ht.end());
    for (int r = 0; r < n; r++) {
      const int t = ht[r] - 1;
      if (tag) tag[r] = ht[r];
      if (t < 0 || t >= maxtag_) continue;
      if (cell) cell[r] = (hc[t] >= 0 && !label_.empty()) ? label_[hc[t]] : hc[t];
      if (Jd) Jd[r] = hj[t];
      if (pDrag)
        for (int k = 0; k < 3; k++) pDrag[3 * r + k] = hf[(size_t)k * maxtag_ + t];
    }
  }

  int particle_count() const { return lmp_->eng.nlocal(); }
  void slab_info(void** lammps, int* n3, int* nx_global, int* periodic_x, int* smoothing)
  {
    *lammps = lmp_;
    for (int k = 0; k < 3; k++) n3[k] = mesh_.n[k];
    *nx_global = slab_nx_global_;
    *periodic_x = slab_per_x_;
    *smoothing = (smoother_.slab() && smoother_.enabled()) ? 1 : 0;
  }
  const sf_cloud_timers& timers() const { return t_; }

 private:
  static double now()
  {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }
  double sync_now()
  {
    SF_HIP(hipStreamSynchronize(s_));
    return now();
  }

  void ensure_particle_arrays()
  {
    DemEngine& e = lmp_->eng;
    const size_t cap = e.capacity();
    if (cap > pcap_) {
      auto re = [&](void** p, size_t bytes) {
        if (*p) SF_HIP(hipFree(*p));
        SF_HIP(hipMalloc(p, bytes));
        SF_HIP(hipMemsetAsync(*p, 0, bytes, s_));
      };
      re((void**)&keys_, sizeof(unsigned) * cap);
      re((void**)&keys2_, sizeof(unsigned) * cap);
      re((void**)&idx_, sizeof(int) * cap);
      re((void**)&idx2_, sizeof(int) * cap);
      pcap_ = cap;
    }
    const int mt = e.max_tag();
    if (mt > maxtag_) {
      const int newmax = mt + mt / 4 + 16;
      auto re2 = [&](void** p, size_t bytes) {
        if (*p) SF_HIP(hipFree(*p));
        SF_HIP(hipMalloc(p, bytes));
        SF_HIP(hipMemsetAsync(*p, 0, bytes, s_));
      };
      re2((void**)&Jd_, sizeof(double) * newmax);
      re2((void**)&cell_, sizeof(int) * newmax);
      re2((void**)&pDragT_, sizeof(double) * 3 * (size_t)newmax);
      maxtag_ = newmax;
    }
  }

  CloudFlagsDev flags() const
  {
    CloudFlagsDev f;
    f.dragModel = props_.dragModel;
    f.particleDrag = props_.particleDrag;
    f.particlePressureGrad = props_.particlePressureGrad;
    f.particleBuoyancy = props_.particleBuoyancy;
    f.particleAddedMass = props_.particleAddedMass;
    f.particleLift = props_.particleLift;
    f.lubricationForce = props_.lubricationForce;
    for (int k = 0; k < 3; k++) f.g[k] = props_.gravity[k];
    f.rhob = props_.rhob;
    f.nub = props_.nub;
    f.deltaT = deltaT_;
    // enhancedCloud.C:600-608: inletForce is only read with addParticleOption > 0; :249 mag(inletForceRatio_) > 0
    const double* iF = props_.inletForce;
    const bool on = props_.addParticleOption > 0 && (iF[0] != 0.0 || iF[1] != 0.0 || iF[2] != 0.0);
    f.inletOption = on ? props_.addParticleOption : 0;
    for (int k = 0; k < 3; k++) {
      f.inletForce[k] = iF[k];
      f.ecc[k] = props_.eccentricity[k];
    }
    for (int k = 0; k < 9; k++) f.inletBox[k] = props_.inletBox[k];
    return f;
  }

  void drag_on_particles()
  {
    DemEngine& e = lmp_->eng;
    const int n = e.nlocal();
    if (!n) return;
    ensure_particle_arrays();
    k_drag_on_particles<<<div_up(n, 256), 256, 0, s_>>>(n, e.capacity(), e.d_xr(), e.d_vm(), e.d_tag(), mesh_,
                                                        flags(), gamma_, UfS_, gradp_, DDtUf_, curlU_,
                                                        e.d_extra() + (size_t)xrow_ * e.capacity(), maxtag_, cell_,
                                                        Jd_, pDragT_, e.d_fdrag(),
                                                        props_.particleHistoryForce ? time_index_ : -1, UfSold_);
  }

  void sort_by_cell(int n)
  {
    DemEngine& e = lmp_->eng;
    const int nc = mesh_.ncells;
    // cstart_ = [start | end] per cell (+ the "outside" bin); idx2_ = particle indices cell by cell, ascending inside
    SF_HIP(hipMemsetAsync(cstart_, 0, sizeof(int) * 2 * ((size_t)nc + 1), s_));
    if (!n) return;
    int* cend = cstart_ + nc + 1;
    k_cell_count<<<div_up(n, 256), 256, 0, s_>>>(n, e.d_xr(), mesh_, keys_, cend);              // counts in `end`
    exclusive_scan_i32(sort_tmp_, sort_tmp_bytes_, cend, cstart_, nc + 1, s_);
    SF_HIP(hipMemcpyAsync(cend, cstart_, sizeof(int) * ((size_t)nc + 1), hipMemcpyDeviceToDevice, s_));
    k_cell_place<<<div_up(n, 256), 256, 0, s_>>>(n, keys_, cend, idx2_);                       // end = start + count
    k_cell_sort_segments<<<div_up((nc + 1) * 64, 256), 256, 0, s_>>>(nc + 1, cstart_, cend, idx2_, idx_);
  }

  void particle_to_eulerian()
  {
    scatter_local();
    scatter_finish();
  }

  void scatter_local()
  {
    DemEngine& e = lmp_->eng;
    const int n = e.nlocal();
    ensure_particle_arrays();
    sort_by_cell(n);
    const int L = lanes_per_cell(n);
    auto launch = [&](auto kern) {
      kern<<<div_up((size_t)mesh_.ncells * L, 256), 256, 0, s_>>>(mesh_.ncells, cstart_, cstart_ + mesh_.ncells + 1,
                                                                  idx2_, e.d_xr(), e.d_vm(), V_, gamma_, Ue_);
    };
    if (L == 64) launch(k_particle_to_eulerian<64>);
    else if (L == 8) launch(k_particle_to_eulerian<8>);
    else launch(k_particle_to_eulerian<1>);
  }

  bool scatter_finish(bool resume = false)
  {
    // gamma (:944-948) and Ue (:950-953): independent solves, batched through the same launches
    bool paused = false;
    if (props_.alphaSmooth && props_.UpSmooth) paused = smooth_step(resume, gamma_, 1, Ue_, 3);
    else if (props_.alphaSmooth) paused = smooth_step(resume, gamma_, 1, nullptr, 0);
    else if (props_.UpSmooth) paused = smooth_step(resume, Ue_, 3, nullptr, 0);
    if (paused) return true;
    k_divide_ue<<<div_up(mesh_.ncells, 256), 256, 0, s_>>>(mesh_.ncells, gamma_, Ue_);
    return false;
  }

  // UfSmoothed_ = Uf_ [ * (1 - gamma), smoothed, / (1 - gamma) ]   enhancedCloud.C:675-690
  // ++runTime: UfSmoothed_ of the previous step becomes its oldTime(), the time index advances
  void advance_time()
  {
    SF_HIP(hipMemcpyAsync(UfSold_, UfS_, sizeof(double) * 3 * (size_t)mesh_.ncells, hipMemcpyDeviceToDevice, s_));
    time_index_++;
  }

  bool update_uf_smoothed(bool resume = false)
  {
    const int nb = div_up(mesh_.ncells, 256);
    if (!resume) k_weight<<<nb, 256, 0, s_>>>(mesh_.ncells, 2, gamma_, UfS_, Uf_);
    if (props_.UfSmooth && smoother_.enabled()) {
      if (!resume) k_weight<<<nb, 256, 0, s_>>>(mesh_.ncells, 0, gamma_, UfS_, nullptr);
      if (smooth_step(resume, UfS_, 3, nullptr, 0)) return true;
      k_weight<<<nb, 256, 0, s_>>>(mesh_.ncells, 1, gamma_, UfS_, nullptr);
    }
    return false;
  }

public:
  void smooth_host_field(double* field, int ncomp)
  {
    if (ncomp != 1 && ncomp != 3) fail("smoothField: ncomp must be 1 or 3");
    const size_t nb = sizeof(double) * (size_t)mesh_.ncells * ncomp;
    double* d = nullptr;
    SF_HIP(hipMalloc(&d, nb));
    upload_field(d, field, ncomp);
    smoother_.smooth(d, ncomp);
    download_field(field, d, ncomp);
    SF_HIP(hipStreamSynchronize(s_));
    (void)hipFree(d);
  }

 private:

  SfLammps* lmp_;
  int slab_nx_global_ = 0, slab_per_x_ = 0;
  sf_cloud_props props_;
  double deltaT_;
  MeshDev mesh_{};
  double* faces_dev_[3] = {nullptr, nullptr, nullptr};
  std::vector<int> label_;   // grid cell -> OpenFOAM cell label (empty: identity)
  hipStream_t s_ = nullptr;
  int subCycles_ = 1, subSteps_ = 1;
  double *V_ = nullptr, *gamma_ = nullptr, *Ue_ = nullptr, *Asrc_ = nullptr, *Omega_ = nullptr;
  double *Uf_ = nullptr, *DDtUf_ = nullptr, *gradp_ = nullptr, *curlU_ = nullptr, *UfS_ = nullptr;
  DiffusionSmoother smoother_;
  double *Jd_ = nullptr, *pDragT_ = nullptr;   // diagnostics by tag
  int xrow_ = -1;                              // first of the 7 per-atom rows this cloud keeps in the engine
  int pending_ = -1;                           // phase paused for the x solve of a slab mesh
  double* UfSold_ = nullptr;                                     // UfSmoothed_.oldTime()
  int time_index_ = 0;                                           // runTime().timeIndex()
  int *cstart_ = nullptr, *cell_ = nullptr, *idx_ = nullptr, *idx2_ = nullptr;
  unsigned *keys_ = nullptr, *keys2_ = nullptr;
  void* sort_tmp_ = nullptr;
  size_t sort_tmp_bytes_ = 0, pcap_ = 0;
  int maxtag_ = 0;
  sf_cloud_timers t_{};
};

}  // namespace sf

using sf::Cloud;

extern "C" {

int sf_cloud_create(void* lmp, const sf_cloud_mesh* mesh, const sf_cloud_props* props, double deltaT, void** cloud)
{
  SF_API_BEGIN
  if (!lmp) sf::fail("null engine handle");
  *cloud = new Cloud(static_cast<sf::SfLammps*>(lmp), *mesh, *props, deltaT);
  SF_API_END(0)
}

int sf_cloud_destroy(void* cloud)
{
  SF_API_BEGIN
  delete static_cast<Cloud*>(cloud);
  SF_API_END(0)
}

int sf_cloud_set_fluid(void* cloud, const double* Uf, const double* DDtUf, const double* gradp, const double* curlU)
{
  SF_API_BEGIN
  static_cast<Cloud*>(cloud)->set_fluid(Uf, DDtUf, gradp, curlU);
  SF_API_END(0)
}

int sf_cloud_evolve(void* cloud)
{
  SF_API_BEGIN
  static_cast<Cloud*>(cloud)->evolve();
  SF_API_END(0)
}

int sf_cloud_phase(void* cloud, int phase)
{
  SF_API_BEGIN
  const int rc = static_cast<Cloud*>(cloud)->phase(phase);
  SF_API_END(rc)
}

// what the slab exchanges (csrc/sf_halo_rccl.hip: sf_cloud_slab_halo_add / sf_cloud_slab_phase) need to know of a cloud
int sf_cloud_slab_info(void* cloud, void** lammps, int* n3, int* nx_global, int* periodic_x, int* smoothing)
{
  SF_API_BEGIN
  static_cast<Cloud*>(cloud)->slab_info(lammps, n3, nx_global, periodic_x, smoothing);
  SF_API_END(0)
}

int sf_cloud_smooth_work(void* cloud, double** dev_work, int* nfields)
{
  SF_API_BEGIN
  sf::DiffusionSmoother& sm = static_cast<Cloud*>(cloud)->smoother();
  if (!sm.slab() || !sm.enabled()) sf::fail("sf_cloud_smooth_work: not a smoothed slab mesh");
  *dev_work = sm.work();
  *nfields = sm.work_fields();
  SF_API_END(0)
}

int sf_cloud_smooth_xsolve(void* cloud, double* dev_lines, long long nlines, long long first_line)
{
  SF_API_BEGIN
  static_cast<Cloud*>(cloud)->smoother().xsolve(dev_lines, nlines, first_line);
  SF_API_END(0)
}

int sf_cloud_sub_cycling(void* cloud, int* subCycles, int* subSteps)
{
  SF_API_BEGIN
  *subCycles = static_cast<Cloud*>(cloud)->sub_cycles();
  *subSteps = static_cast<Cloud*>(cloud)->sub_steps();
  SF_API_END(0)
}

int sf_cloud_device_fields(void* cloud, double** gamma, double** Ue, double** Asrc, int* ncells)
{
  SF_API_BEGIN
  static_cast<Cloud*>(cloud)->device_fields(gamma, Ue, Asrc, ncells);
  SF_API_END(0)
}

int sf_cloud_calc_tc_fields(void* cloud)
{
  SF_API_BEGIN
  static_cast<Cloud*>(cloud)->calc_tc_fields();
  SF_API_END(0)
}

int sf_cloud_smooth_field(void* cloud, double* field, int ncomp)
{
  SF_API_BEGIN
  static_cast<Cloud*>(cloud)->smooth_host_field(field, ncomp);
  SF_API_END(0)
}

int sf_cloud_average_info(void* cloud, double* out7)
{
  SF_API_BEGIN
  static_cast<Cloud*>(cloud)->average_info(out7);
  SF_API_END(0)
}

int sf_cloud_get_fields(void* cloud, double* gamma, double* Ue, double* Asrc, double* Omega)
{
  SF_API_BEGIN
  static_cast<Cloud*>(cloud)->get_fields(gamma, Ue, Asrc, Omega);
  SF_API_END(0)
}

int sf_cloud_get_particles(void* cloud, int* tag, int* cell, double* pDrag, double* Jd)
{
  SF_API_BEGIN
  static_cast<Cloud*>(cloud)->get_particles(tag, cell, pDrag, Jd);
  SF_API_END(0)
}

int sf_cloud_particle_count(void* cloud)
{
  SF_API_BEGIN
  const int n = static_cast<Cloud*>(cloud)->particle_count();
  SF_API_END(n)
}

int sf_cloud_adjust_timestep(double deltaT, double dtLampIn, int subCycles_in, double* dtLampAdj,
                             int* solidStepsPerDt, int* subCycles, int* subSteps)
{
  return sf::adjust_timestep(deltaT, dtLampIn, subCycles_in, dtLampAdj, solidStepsPerDt, subCycles, subSteps);
}

int sf_cloud_get_timers(void* cloud, sf_cloud_timers* t)
{
  SF_API_BEGIN
  *t = static_cast<Cloud*>(cloud)->timers();
  SF_API_END(0)
}

int sfk_drag_model_jd(int model, int n, const double* Ur, const double* alpha, const double* pd, double nuf,
                      double rhof, double* Jd, void* stream)
{
  SF_API_BEGIN
  if (model < 0 || model > 2)
    sf::fail("Unknown dragModel type %d (valid: 0 ErgunWenYu, 1 SyamlalOBrien, 2 NoCorrection)", model);
  if (n > 0) {
    sf::k_jd<<<sf::div_up(n, 256), 256, 0, (hipStream_t)stream>>>(model, n, Ur, alpha, pd, nuf, rhof, Jd);
    SF_HIP(hipGetLastError());
  }
  SF_API_END(0)
}

int sfk_cell_owner(int n, const double* x, const double origin[3], const double dx[3], const int ncell[3],
                   int* cell, void* stream)
{
  const double* const none[3] = {nullptr, nullptr, nullptr};
  return sfk_cell_owner_graded(n, x, origin, dx, ncell, none, cell, stream);
}

int sfk_cell_owner_graded(int n, const double* x, const double origin[3], const double dx[3], const int ncell[3],
                          const double* const dev_faces[3], int* cell, void* stream)
{
  SF_API_BEGIN
  sf::MeshDev m;
  for (int k = 0; k < 3; k++) {
    m.origin[k] = origin[k];
    m.dx[k] = dx[k];
    m.n[k] = ncell[k];
    m.f[k] = dev_faces[k];
    m.per[k] = 0;
  }
  m.ncells = ncell[0] * ncell[1] * ncell[2];
  if (n > 0) {
    sf::k_cell_owner_aos<<<sf::div_up(n, 256), 256, 0, (hipStream_t)stream>>>(n, x, m, cell);
    SF_HIP(hipGetLastError());
  }
  SF_API_END(0)
}

}  // extern "C"
