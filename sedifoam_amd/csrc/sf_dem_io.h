// sf_dem_io.h -- host <-> device marshalling kernels of the lammps_* surface (library.cpp:203-420: lammps_get_local_info,
// lammps_put_local_info, forces / torques out), group and velocity commands, pair counts and history collection.
// Included by sf_dem.hip.
#pragma once
#include "sf_dem.h"

namespace sf {

// ------------------------------------------------------------------------------------------------
// host <-> device marshalling of the lammps_* surface
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_info(const double4* xr, const double4* vm, const double4* om,
                                                   const double4* force, const double4* torque, int n,
                                                   double* x, double* v, double* w, double* f, double* t,
                                                   double* diam, double* rho)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double4 a = xr[i], b = vm[i];
  if (x) { x[3 * i] = a.x; x[3 * i + 1] = a.y; x[3 * i + 2] = a.z; }
  if (v) { v[3 * i] = b.x; v[3 * i + 1] = b.y; v[3 * i + 2] = b.z; }
  if (w) { const double4 c = om[i]; w[3 * i] = c.x; w[3 * i + 1] = c.y; w[3 * i + 2] = c.z; }
  if (f) { const double4 c = force[i]; f[3 * i] = c.x; f[3 * i + 1] = c.y; f[3 * i + 2] = c.z; }
  if (t) { const double4 c = torque[i]; t[3 * i] = c.x; t[3 * i + 1] = c.y; t[3 * i + 2] = c.z; }
  if (diam) diam[i] = a.w * 2.0;                                               // library.cpp:196
  if (rho) rho[i] = 3.0 * b.w / (4.0 * kPiTypo * a.w * a.w * a.w);             // library.cpp:200
}

__global__ __launch_bounds__(256) void k_tag_map(const int* tag, int n, int* map, int maxtag)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int t = tag[i];
  if (t >= 1 && t <= maxtag) map[t - 1] = i;
}

// library.cpp:344-366: incoming rows are matched to atoms by tag
__global__ __launch_bounds__(256) void k_put_fdrag(const double* in, const int* tagIn, const int* cpuIn, int n,
                                                   const int* map, int maxtag, double* fdrag, int* foamCpuId,
                                                   size_t cap, int* flags)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int t = tagIn[k];
  const int i = (t >= 1 && t <= maxtag) ? map[t - 1] : -1;
  if (i < 0) {
    flags[F_LOST] = 2;
    return;
  }
  fdrag[i] = in[3 * k];
  fdrag[cap + i] = in[3 * k + 1];
  fdrag[2 * cap + i] = in[3 * k + 2];
  if (cpuIn) foamCpuId[i] = cpuIn[k];
}

// [3P] group ID type ... : op 0 = list of types, 1 <, 2 <=, 3 >, 4 >=, 5 ==, 6 !=, 7 <> (between, inclusive)
struct GroupTypeArgs {
  int op, v1, v2, nlist;
  int list[16];
};
__global__ __launch_bounds__(256) void k_group_type(int* mask, const int* type, int n, int bit, GroupTypeArgs A)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int t = type[i];
  bool in = false;
  switch (A.op) {
    case 0:
      for (int k = 0; k < A.nlist; k++) in = in || (t == A.list[k]);
      break;
    case 1: in = t < A.v1; break;
    case 2: in = t <= A.v1; break;
    case 3: in = t > A.v1; break;
    case 4: in = t >= A.v1; break;
    case 5: in = t == A.v1; break;
    case 6: in = t != A.v1; break;
    default: in = t >= A.v1 && t <= A.v2; break;
  }
  if (in) mask[i] |= bit;
}

// group ID subtract A B.. (in A, in none of the others) / union / intersect
struct GroupCombineArgs {
  int mode, n;
  int bits[16];
};
__global__ __launch_bounds__(256) void k_group_combine(int* mask, int n, int bit, GroupCombineArgs A)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int m = mask[i];
  bool in;
  if (A.mode == 0) {
    in = (m & A.bits[0]) != 0;
    for (int k = 1; k < A.n; k++) in = in && !(m & A.bits[k]);
  } else if (A.mode == 1) {
    in = false;
    for (int k = 0; k < A.n; k++) in = in || (m & A.bits[k]);
  } else {
    in = true;
    for (int k = 0; k < A.n; k++) in = in && (m & A.bits[k]);
  }
  if (in) mask[i] |= bit;
}

// omega.w = 1 for the atoms of the fix-freeze group, in both ping-pong buffers
__global__ __launch_bounds__(256) void k_mark_frozen(double4* om0, double4* om1, const int* mask, int n, int bit)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double f = (mask[i] & bit) ? 1.0 : 0.0;
  om0[i].w = f;
  om1[i].w = f;
}

__global__ __launch_bounds__(256) void k_set_velocity_group(double4* vm, const int* mask, int bit, int n, double vx,
                                                            double vy, double vz)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !(mask[i] & bit)) return;
  double4 v = vm[i];
  v.x = vx; v.y = vy; v.z = vz;
  vm[i] = v;
}

__global__ __launch_bounds__(256) void k_set_velocity(double4* vm, int n, double vx, double vy, double vz)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double4 v = vm[i];
  v.x = vx; v.y = vy; v.z = vz;
  vm[i] = v;
}

__global__ __launch_bounds__(1024) void k_count_pairs(const int* numneigh, int n, unsigned long long* out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = block_sum_int_1024(i < n ? numneigh[i] : 0);
  if (threadIdx.x == 0 && t) atomicAdd(out, (unsigned long long)t);
}

// touching pairs -> (tag_i < tag_j, shear as tag_i sees it) compacted with an atomic cursor.  A pair of two atoms of
// this GPU has one copy (the owner's slot); pairs with a periodic image or a ghost of another GPU have a copy on
// each side, of which the lower tag's is reported.
__global__ __launch_bounds__(256) void k_collect_history(const int* neigh, const int* numneigh, const double* shear,
                                                         const int* tag, int nlocal, size_t cap,
                                                         unsigned long long* cursor, long long max, int* ti,
                                                         int* tj, double* sh, int roots)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nlocal) return;
  const int nn = numneigh[i];
  const int tagi = tag[i];
  for (int s = 0; s < nn; s++) {
    const int jraw = neigh[(size_t)s * cap + i];
    if (!(jraw & kTouchBit) || !(jraw & kOwnBit)) continue;
    const int j = neigh_index(jraw, roots);
    const int tagj = tag[j];
    // is this the pair's only copy?  Only if the other side is an atom of this GPU itself (no image) whose word for
    // this atom is a partner-side one; otherwise both sides hold a copy (images, ghosts of another GPU, lists built
    // with two copies per contact) and the lower tag's is the one reported
    bool single = false;
    if (roots && j < nlocal && ((jraw >> kIdxBits) & 31) == kNoShift) {
      const int nj = numneigh[j];
      for (int u = 0; u < nj; u++) {
        const int wu = neigh[(size_t)u * cap + j];
        if (!(wu & kOwnBit) && (wu & kIdxMask) == i) {
          single = true;
          break;
        }
      }
    }
    if (!single && tagi >= tagj) continue;   // the other side's own copy is the one reported
    const bool flip = tagi > tagj;
    const long long k = (long long)atomicAdd(cursor, 1ull);
    if (k < max) {
      ti[k] = flip ? tagj : tagi;
      tj[k] = flip ? tagi : tagj;
      const size_t b = (size_t)(3 * s) * cap + i;
      const double sg = flip ? -1.0 : 1.0;
      sh[3 * k] = sg * shear[b];
      sh[3 * k + 1] = sg * shear[b + cap];
      sh[3 * k + 2] = sg * shear[b + 2 * cap];
    }
  }
}

}  // namespace sf
