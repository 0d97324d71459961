// sf_dem_lds_kernel.h -- the LDS-staged cell-bin form of the sub-step kernel (SF_LDS=1).  Included by sf_dem.hip after
// sf_dem_kernels.h: it runs the same substep_particle<..., LDS = true, ...> on a tile's staged copy.  Measured 1.5 x
// slower than the gathering kernel at 1 M grains (profiles/r01_*): an option kept for beds whose neighbour gathers do
// not coalesce, checked against the oracle like every other variant (tests/test_dem_gpu.py).
#pragma once
#include "sf_dem_kernels.h"

namespace sf {

// LDS-staged cell bins: one workgroup per tile of T x T x T bins.  The x/v/omega records of every atom in
// the tile and in the one-bin shell around it (owned and ghost) are copied ONCE into LDS with mostly
// sequential loads (atoms are sorted tile by tile, bin by bin); the 12-odd neighbour look-ups per atom then
// hit LDS (ds_read_b128) instead of issuing 6 scattered 16-byte global loads each, which is what saturates
// the vector-memory address pipe of a CU in k_substep.  nloc[slot][i] is the neighbour's position in that
// staged copy, written when the list is built.
template <int STYLE, bool COHE, bool LUB>
__global__ __launch_bounds__(1024) void k_substep_lds(DemPtrs P, StepParams S)
{
  extern __shared__ double4 lds4[];
  if (__atomic_load_n(&P.flags[S.trig_test], __ATOMIC_RELAXED) < S.kstep) return;
  int tile = blockIdx.x;
  if (S.xcd_remap) {
    const int nb = gridDim.x, xcd = tile & 7, q = nb >> 3, r = nb & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (tile >> 3);
  }
  const int first = P.tile_first[tile], last = P.tile_last[tile];
  if (first >= last) return;
  const int s0 = P.stage_start[tile], ns = P.stage_start[tile + 1] - s0;
  double4* lx = lds4;
  double4* lv = lds4 + S.stage_cap;
  double* lw = reinterpret_cast<double*>(lds4 + 2 * (size_t)S.stage_cap);
  for (int k = threadIdx.x; k < ns; k += blockDim.x) {
    const int g = P.stage_idx[s0 + k];
    lx[k] = P.xr_in[g];
    lv[k] = P.vm_in[g];
    const double4 w = P.om_in[g];
    lw[3 * k] = w.x;
    lw[3 * k + 1] = w.y;
    lw[3 * k + 2] = w.z;
  }
  __syncthreads();
  for (int i = first + threadIdx.x; i < last; i += blockDim.x)
    substep_particle<STYLE, COHE, LUB, true, 1, true, 2>(P, S, i, 0, lx, lv, lw);
}

}  // namespace sf
