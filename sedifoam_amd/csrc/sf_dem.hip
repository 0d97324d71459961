// sf_dem.hip -- host side of the device-resident DEM engine (see sf_dem.h for the HBM layout).
//
// Sub-step driver = LAMMPS 1Feb14 Verlet::run as the reference invokes it through
// lammps_step() (interfaceToLammps/library.cpp:372-386, "run n pre no post no") [3P].
#include "sf_dem.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "sf_dem_kernels.h"
#include "sf_dem_rebuild.h"
#include "sf_dem_io.h"
#include "sf_dem_lds_kernel.h"
#include "sf_roctx.h"

namespace sf {

std::string& last_error()
{
  static thread_local std::string e;
  return e;
}

void set_error(const char* fmt, ...)
{
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error() = buf;
}

void fail(const char* fmt, ...)
{
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  throw Error(buf);
}

static double hertz_beta(double gamman)
{
  // pair_gran_hertzFix_history.cpp:195-196 evaluated once (a pure function of gamman)
  const double lg = std::log(gamman) / std::log(std::exp(1.0));
  return -(lg) / std::sqrt(lg * lg + kPi * kPi);
}

void fold_hertz_constants(GranParams& p)
{
  const double c56 = 2.0 * std::sqrt(5.0 / 6.0);
  p.h_sn = 2.0 * 1.0 / 1.82 * p.kn;
  p.h_cn = 4.0 / 5.46 * p.kn;
  p.h_ct = 8.0 / 8.84 * p.kt;
  p.h_inv_ct = p.kt > 0.0 ? 8.0 / 8.84 / p.kt : 0.0;
  p.inv_kt = p.kt > 0.0 ? 1.0 / p.kt : 0.0;
  p.h_c56beta = c56 * p.beta;
  p.h_stsn_c56beta = std::sqrt((8.0 * 1.0 / 8.84) / (2.0 * 1.0 / 1.82)) * c56 * p.beta;
}

static void gran_settings(GranParams& p, int style, double kn, bool kt_null, double kt, double gamman,
                          bool gammat_null, double gammat, double xmu, int dampflag, double nktv2p)
{
  p.style = style;
  p.kn = kn;
  p.kt = kt_null ? kn * 2.0 / 7.0 : kt;
  p.gamman = gamman;
  p.gammat = gammat_null ? 0.5 * gamman : gammat;
  p.xmu = xmu;
  p.dampflag = dampflag;
  if (dampflag == 0) p.gammat = 0.0;
  if (p.kn < 0.0 || p.kt < 0.0 || p.gamman < 0.0 || p.gammat < 0.0 || p.xmu < 0.0 || p.xmu > 10000.0 ||
      dampflag < 0 || dampflag > 1)
    fail("Illegal pair_style command");
  p.kn /= nktv2p;
  p.kt /= nktv2p;
  p.beta = (style == 2 && gamman > 0.0) ? hertz_beta(gamman) : 0.0;
  fold_hertz_constants(p);
}

DemEngine::DemEngine()
{
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    fail("sedifoam_amd: no HIP device available (%s) -- this library has no CPU path",
         e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  SF_HIP(hipStreamCreateWithFlags(&own_stream_, hipStreamNonBlocking));
  stream_ = own_stream_;
  SF_HIP(hipMalloc(&own_flags_, sizeof(int) * F_NFLAGS));
  d_flags_ = own_flags_;
  SF_HIP(hipMalloc(&count64_, sizeof(unsigned long long)));
  SF_HIP(hipHostMalloc(&h_flags_, sizeof(int) * F_NFLAGS));
  memset(h_flags_, 0, sizeof(int) * F_NFLAGS);   // (the arrival word starts below every sequence number)
  SF_HIP(hipMemsetAsync(d_flags_, 0, sizeof(int) * F_NFLAGS, stream_));
  SF_HIP(hipEventCreate(&ev0_));
  SF_HIP(hipEventCreate(&ev1_));
  SF_HIP(hipEventCreateWithFlags(&ev_flags_, hipEventDisableTiming));
  if (const char* e = getenv("SF_TILE")) opt_tile_ = atoi(e);
  if (const char* e = getenv("SF_XCD_REMAP")) opt_xcd_remap_ = atoi(e);
  SF_HIP(hipMalloc(&d_xcd_time_, sizeof(int) * 1024));
  SF_HIP(hipHostMalloc(&h_xcd_time_, sizeof(int) * 1024));
  for (int k = 0; k < 512; k++) h_xcd_time_[512 + k] = (k & 63) == 0 ? INT_MAX : 0;   // start: atomicMin, end: atomicMax
  SF_HIP(hipMemcpyAsync(d_xcd_time_ + 512, h_xcd_time_ + 512, sizeof(int) * 512, hipMemcpyHostToDevice, stream_));
  if (const char* e = getenv("SF_XCD_BALANCE")) xcd_auto_ = atoi(e) != 0;
  if (const char* e = getenv("SF_XCD_WEIGHTS")) {   // eight relative shares, comma separated: pins them
    double w[8];
    if (sscanf(e, "%lf,%lf,%lf,%lf,%lf,%lf,%lf,%lf", &w[0], &w[1], &w[2], &w[3], &w[4], &w[5], &w[6], &w[7]) == 8) {
      for (int x = 0; x < 8; x++) xcd_weight_[x] = w[x] > 0.0 ? w[x] : 1.0;
      xcd_weighted_ = true;
      xcd_auto_ = false;
    }
  }
  if (const char* e = getenv("SF_LDS")) opt_lds_ = atoi(e);
  roots_ = !opt_lds_;
  if (const char* e = getenv("SF_SUB")) {
    opt_sub_ = std::max(1, atoi(e));
    opt_sub_env_ = true;
  }
  if (const char* e = getenv("SF_HIST_COPIES")) hist_mode_env_ = atoi(e);
  if (const char* e = getenv("SF_TOUCH_PREFETCH")) touch_prefetch_env_ = atoi(e);
  if (const char* e = getenv("SF_TOUCH_FIRST")) touch_first_env_ = atoi(e);
  if (const char* e = getenv("SF_NT_POLICY")) nt_policy_env_ = atoi(e);
  if (const char* e = getenv("SF_LPA")) opt_lpa_ = atoi(e);
  if (const char* e = getenv("SF_GHOST_FREE")) opt_ghost_free_ = atoi(e);
  if (const char* e = getenv("SF_PERSIST")) opt_persist_ = atoi(e);
  if (const char* e = getenv("SF_PERSIST_WAVES")) opt_persist_waves_ = atoi(e);
  SF_HIP(hipMalloc(&d_pq_head_, sizeof(int) * 2 * 8 * 32));
  SF_HIP(hipMemsetAsync(d_pq_head_, 0, sizeof(int) * 2 * 8 * 32, stream_));
  if (const char* e = getenv("SF_QUEUE_PREDICT")) predict_.on = atoi(e) != 0;
  memset(&gran_, 0, sizeof(gran_));
  memset(&cohe_, 0, sizeof(cohe_));
  memset(&lub_, 0, sizeof(lub_));
  per_atom_ = {&xr_[0], &xr_[1], &vm_[0], &vm_[1], &om_[0], &om_[1], &force_, &torque_, &tag_, &type_,
               &mask_, &foamCpuId_, &fdrag_, &DuDt_, &vOld_, &xhold_, &extra_, &wshear_, &wtouch_, &gsrc_, &gshift_,
               &neigh_, &numneigh_, &shear_[0], &shear_[1], &neigh_old_, &numneigh_old_, &ptag_, &tmp4_,
               &tmpd_, &tmpi_, &keys_, &keys_alt_, &perm_, &perm_alt_, &keys64_, &keys64_alt_,
               &sendlist_[0], &sendlist_[1], &leave_, &nloc_, &isb_, &hist_perm_, &bmask_, &tmp4b_, &tmp4c_,
               &tmpi_b_, &tmpi_c_, &tmpi_d_, &fdrag_alt_, &DuDt_alt_, &vOld_alt_, &extra_alt_, &wtouch_alt_};
}

DemEngine::~DemEngine()
{
  if (stream_) (void)hipStreamSynchronize(stream_);
  for (DevArray* a : per_atom_) a->release();
  if (d_blkptr_) (void)hipFree(d_blkptr_);
#ifdef SF_EXP_BUILD_PHASE
  {
    unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_build_phase), sizeof(h)) == hipSuccess && (h[0] | h[1] | h[3])) {
      const double tot = (double)(h[0] + h[1] + h[2] + h[3] + h[4] + h[5]);
      fprintf(stderr, "[sedifoam_amd] k_build_neigh wave cycles by phase (%d builds): prologue %.3f  walk %.3f  old tags %.3f  "
              "touch-first pass %.3f  second sweep %.3f  counts %.3f  (sum %.3e cycles)\n", (int)nbuilds_, h[0] / tot, h[1] / tot,
              h[5] / tot, h[2] / tot, h[3] / tot, h[4] / tot, tot);
    }
  }
#endif
  if (d_xcd_time_) (void)hipFree(d_xcd_time_);
  if (d_pq_head_) (void)hipFree(d_pq_head_);
  if (h_xcd_time_) (void)hipHostFree(h_xcd_time_);
  bslot_.release();   // (not in per_atom_: allocated by the first brick rebuild, re-allocated when the capacity moves)
  if (cell_start_) (void)hipFree(cell_start_);
  if (tile_tab_) (void)hipFree(tile_tab_);
  if (bsend_list_) (void)hipFree(bsend_list_);
  if (d_bcount_) (void)hipFree(d_bcount_);
  if (h_bcount_) (void)hipHostFree(h_bcount_);
  if (stage_idx_) (void)hipFree(stage_idx_);
  if (eoff_) (void)hipFree(eoff_);
  if (tagmap_) (void)hipFree(tagmap_);
  for (IoBuf& b : io_d_)
    if (b.p) (void)hipFree(b.p);
  for (IoBuf& b : io_i_)
    if (b.p) (void)hipFree(b.p);
  if (sort_tmp_) (void)hipFree(sort_tmp_);
  if (own_flags_) (void)hipFree(own_flags_);
  if (count64_) (void)hipFree(count64_);
  if (h_flags_) (void)hipHostFree(h_flags_);
  if (ev0_) (void)hipEventDestroy(ev0_);
  if (ev1_) (void)hipEventDestroy(ev1_);
  if (ev_flags_) (void)hipEventDestroy(ev_flags_);
  for (hipEvent_t e : prof_ev_) (void)hipEventDestroy(e);
  if (own_stream_) (void)hipStreamDestroy(own_stream_);
  if (masked_main_) (void)hipStreamDestroy(masked_main_);
  if (masked_comm_) (void)hipStreamDestroy(masked_comm_);
}

void DemEngine::set_stream(hipStream_t s)
{
  sync();
  if (s == (hipStream_t)-1) {  // sentinel: back to the engine's own stream
    stream_ = own_stream_;
    external_stream_ = false;
  } else {                     // includes the null (legacy default) stream, which is what torch uses by default
    stream_ = s;
    external_stream_ = true;
  }
}

__global__ __launch_bounds__(256) static void k_fill_row(double* row, int n, double v)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) row[i] = v;
}

void DemEngine::alloc_all(size_t cap)
{
  hipStream_t s = stream_;
  for (int b = 0; b < 2; b++) {
    xr_[b].alloc(sizeof(double4), 1, cap, s);
    vm_[b].alloc(sizeof(double4), 1, cap, s);
    om_[b].alloc(sizeof(double4), 1, cap, s);
  }
  force_.alloc(sizeof(double4), 1, cap, s);
  torque_.alloc(sizeof(double4), 1, cap, s);
  tag_.alloc(sizeof(int), 1, cap, s);
  type_.alloc(sizeof(int), 1, cap, s);
  mask_.alloc(sizeof(int), 1, cap, s);
  foamCpuId_.alloc(sizeof(int), 1, cap, s);
  fdrag_.alloc(sizeof(double), 3, cap, s);
  DuDt_.alloc(sizeof(double), 3, cap, s);
  vOld_.alloc(sizeof(double), 3, cap, s);
  xhold_.alloc(sizeof(double), 3, cap, s);
  extra_.alloc(sizeof(double), kMaxExtra, cap, s);
  wshear_.alloc(sizeof(double), 3 * kMaxWalls, cap, s);
  wtouch_.alloc(sizeof(unsigned char), 1, cap, s);
  gsrc_.alloc(sizeof(int), 1, cap, s);
  gshift_.alloc(sizeof(double), 3, cap, s);
  neigh_.alloc(sizeof(int), M_, cap, s);
  numneigh_.alloc(sizeof(int), 1, cap, s);
  shear_[0].alloc(sizeof(double), 3 * M_, cap, s);
  shear_[1].alloc(sizeof(double), 3 * M_, cap, s);
  neigh_old_.alloc(sizeof(int), M_, cap, s);
  numneigh_old_.alloc(sizeof(int), 1, cap, s);
  ptag_.alloc(sizeof(int), M_, cap, s);
  tmp4_.alloc(sizeof(double4), 1, cap, s);
  tmpd_.alloc(sizeof(double), 3 * kMaxWalls, cap, s);
  tmpi_.alloc(sizeof(int), 1, cap, s);
  tmp4b_.alloc(sizeof(double4), 1, cap, s);
  tmp4c_.alloc(sizeof(double4), 1, cap, s);
  tmpi_b_.alloc(sizeof(int), 1, cap, s);
  tmpi_c_.alloc(sizeof(int), 1, cap, s);
  tmpi_d_.alloc(sizeof(int), 1, cap, s);
  fdrag_alt_.alloc(sizeof(double), 3, cap, s);
  DuDt_alt_.alloc(sizeof(double), 3, cap, s);
  vOld_alt_.alloc(sizeof(double), 3, cap, s);
  extra_alt_.alloc(sizeof(double), kMaxExtra, cap, s);
  wtouch_alt_.alloc(sizeof(unsigned char), 1, cap, s);
  keys_.alloc(sizeof(unsigned), 1, cap, s);
  keys_alt_.alloc(sizeof(unsigned), 1, cap, s);
  perm_.alloc(sizeof(int), 1, cap, s);
  perm_alt_.alloc(sizeof(int), 1, cap, s);
  hist_perm_.alloc(sizeof(int), 1, cap, s);
  keys64_.alloc(sizeof(unsigned long long), 1, cap, s);
  keys64_alt_.alloc(sizeof(unsigned long long), 1, cap, s);
  sendlist_[0].alloc(sizeof(int), 1, cap, s);
  sendlist_[1].alloc(sizeof(int), 1, cap, s);
  leave_.alloc(sizeof(int), 1, cap, s);
  nloc_.alloc(sizeof(unsigned short), M_, cap, s);
  isb_.alloc(sizeof(unsigned char), 1, cap, s);
  bmask_.alloc(sizeof(unsigned char), 1, cap, s);
  cap_ = cap;
}

void DemEngine::ensure_capacity(size_t need)
{
  if (need <= cap_) return;
  size_t newcap = need + need / 4 + 1024;
  newcap = (newcap + 63) & ~(size_t)63;   // rows of the [rows][cap] arrays start on 512-byte boundaries
  if (cap_ == 0) {
    alloc_all(newcap);
    return;
  }
  for (DevArray* a : per_atom_) a->grow(newcap, stream_);
  cap_ = newcap;
}

void DemEngine::grow_neigh(int newM)
{
  // re-allocate the slot-major arrays with more rows (old rows keep their place)
  auto regrow = [&](DevArray& a, int rows_per_slot) {
    DevArray n;
    n.alloc(a.elem, rows_per_slot * newM, cap_, stream_);
    SF_HIP(hipMemcpyAsync(n.ptr, a.ptr, a.elem * (size_t)a.rows * cap_, hipMemcpyDeviceToDevice, stream_));
    sync();
    a.release();
    a = n;
  };
  regrow(neigh_, 1);
  regrow(shear_[0], 3);
  regrow(shear_[1], 3);
  regrow(neigh_old_, 1);
  regrow(ptag_, 1);
  regrow(nloc_, 1);
  M_ = newM;
}

int DemEngine::register_extra(int nrows, const double* init)
{
  if (nrows < 1 || nrows > kMaxExtra) fail("register_extra: %d rows requested (at most %d)", nrows, kMaxExtra);
  // first fit: a client that leaves out of order gives its rows back (ownership bitmap, not a stack)
  int first = -1;
  for (int f = 0; f + nrows <= kMaxExtra && first < 0; f++) {
    const unsigned want = ((1u << nrows) - 1u) << f;
    if (!(extra_used_ & want)) first = f;
  }
  if (first < 0) fail("register_extra: %d rows requested, rows in use 0x%x of %d", nrows, extra_used_, kMaxExtra);
  // an engine without atoms yet (an empty slab rank, a case that starts empty and injects particles): the rows live in
  // a small capacity that grows with the first atoms
  if (cap_ == 0) ensure_capacity(4096);
  for (int r = 0; r < nrows; r++) {
    extra_init_[first + r] = init ? init[r] : 0.0;
    k_fill_row<<<div_up((long long)cap_, 256), 256, 0, stream_>>>(extra_.as<double>() + (size_t)(first + r) * cap_, (int)cap_,
                                                                 extra_init_[first + r]);
  }
  extra_used_ |= ((1u << nrows) - 1u) << first;
  nextra_ = 32 - __builtin_clz(extra_used_);   // rows [0, nextra_) are permuted by re-sorts and travel with migrating atoms
  return first;
}

void DemEngine::unregister_extra(int first, int nrows)
{
  if (first < 0 || nrows < 1 || first + nrows > kMaxExtra) return;
  extra_used_ &= ~(((1u << nrows) - 1u) << first);
  nextra_ = extra_used_ ? 32 - __builtin_clz(extra_used_) : 0;
}

void DemEngine::set_max_neigh(int m)
{
  if (m < 4) m = 4;
  if (cap_ == 0) M_ = m;
  else if (m > M_) grow_neigh(m);
}

void DemEngine::set_box(const double lo[3], const double hi[3])
{
  for (int k = 0; k < 3; k++) {
    boxlo_[k] = lo[k];
    boxhi_[k] = hi[k];
  }
  for (int k = 0; k < 3; k++)
    if (!ext_[k]) {
      sublo_[k] = lo[k];
      subhi_[k] = hi[k];
    }
}

void DemEngine::set_periodic(int px, int py, int pz)
{
  periodic_[0] = px;
  periodic_[1] = py;
  periodic_[2] = pz;
}

void DemEngine::set_subdomain(int rank, int nranks, double sublo, double subhi)
{
  rank_ = rank;
  nranks_ = nranks;
  sublo_[0] = sublo;
  subhi_[0] = subhi;
  ext_[0] = true;
  have_subdomain_ = true;
}

void DemEngine::set_subdomain3(int rank, int nranks, const double lo[3], const double hi[3], const int ext[3])
{
  rank_ = rank;
  nranks_ = nranks;
  for (int k = 0; k < 3; k++) {
    ext_[k] = ext[k] != 0;
    sublo_[k] = ext_[k] ? lo[k] : boxlo_[k];
    subhi_[k] = ext_[k] ? hi[k] : boxhi_[k];
  }
  have_subdomain_ = true;
  brick_ = true;
}

void DemEngine::sublo_hi(double out[6]) const
{
  for (int k = 0; k < 3; k++) {
    out[2 * k] = sublo_[k];
    out[2 * k + 1] = subhi_[k];
  }
}

void DemEngine::create_atoms(int n, const double* x, const double* v, const double* omega,
                             const double* diameter, const double* density, const int* tag, const int* type)
{
  if (setup_done_) fail("create_atoms after setup: use lammps_create_particle");
  const int n0 = nlocal_;
  ensure_capacity((size_t)(n0 + n) + (size_t)(n0 + n) / 2 + 4096);
  std::vector<double4> hx(n), hv(n), hw(n);
  std::vector<int> ht(n), hty(n), hm(n, 1);
  for (int i = 0; i < n; i++) {
    const double r = 0.5 * diameter[i];
    // [3P] read_data, atom_style sphere: rmass = 4 pi/3 r^3 density
    const double m = 4.0 * kPi / 3.0 * r * r * r * density[i];
    hx[i] = {x[3 * i], x[3 * i + 1], x[3 * i + 2], r};
    hv[i] = {v ? v[3 * i] : 0.0, v ? v[3 * i + 1] : 0.0, v ? v[3 * i + 2] : 0.0, m};
    hw[i] = {omega ? omega[3 * i] : 0.0, omega ? omega[3 * i + 1] : 0.0, omega ? omega[3 * i + 2] : 0.0, 0.0};
    ht[i] = tag ? tag[i] : n0 + i + 1;
    hty[i] = type ? type[i] : 1;
    max_tag_ = std::max(max_tag_, ht[i]);
    rmax_ = std::max(rmax_, r);
  }
  auto up = [&](DevArray& a, const void* src, size_t bytes, size_t off) {
    SF_HIP(hipMemcpyAsync((char*)a.ptr + off, src, bytes, hipMemcpyHostToDevice, stream_));
  };
  up(xr_[cur_], hx.data(), sizeof(double4) * n, sizeof(double4) * n0);
  up(vm_[cur_], hv.data(), sizeof(double4) * n, sizeof(double4) * n0);
  up(om_[cur_], hw.data(), sizeof(double4) * n, sizeof(double4) * n0);
  up(tag_, ht.data(), sizeof(int) * n, sizeof(int) * n0);
  up(type_, hty.data(), sizeof(int) * n, sizeof(int) * n0);
  up(mask_, hm.data(), sizeof(int) * n, sizeof(int) * n0);
  for (int r = 0; r < nextra_ && n > 0; r++)   // client rows registered before the atoms existed: their initial value
    if (extra_used_ & (1u << r))
      k_fill_row<<<div_up(n, 256), 256, 0, stream_>>>(extra_.as<double>() + (size_t)r * cap_ + n0, n, extra_init_[r]);
  sync();
  nlocal_ = n0 + n;
  order_version_++;
}

void DemEngine::set_pair_gran(int style, double kn, bool kt_null, double kt, double gamman, bool gammat_null,
                              double gammat, double xmu, int dampflag)
{
  gran_settings(gran_, style, kn, kt_null, kt, gamman, gammat_null, gammat, xmu, dampflag, 1.0);
  // wall/granFix follows the pair style (fix_wall_granFix.cpp:217-229)
  for (int w = 0; w < nwalls_; w++) {
    walls_[w].gp.style = style;
    walls_[w].gp.beta = (style == 2 && walls_[w].gp.gamman > 0.0) ? hertz_beta(walls_[w].gp.gamman) : 0.0;
    fold_hertz_constants(walls_[w].gp);
  }
}

void DemEngine::set_pair_lubricate(double mu, int flaglog, int flagfld, double cut_inner, double cut_global,
                                   int flagHI, int flagVF)
{
  lub_.enabled = 1;
  lub_.mu = mu;
  lub_.flaglog = flaglog;
  lub_.flagfld = flagfld;
  lub_.cut_inner = cut_inner;
  lub_.cut_global = cut_global;
  lub_.flagHI = flagHI;
  lub_.flagVF = flagVF;
  lub_.vxmu2f = 1.0;
}

void DemEngine::set_cohesive(double ah, double lam, double smin, double smax, int opt, int groupbit)
{
  if (opt != 0 && opt != 1) fail("invalid option for cohesive force model");  // fix_cohesive.cpp:262
  if (freeze_bit_)
    fail("fix cohesive after fix freeze: the pair loop adds cohesion before every post_force fix; put the fix "
         "cohesive line before fix freeze (no input script of the reference has the two together)");
  cohe_ = {ah, lam, smin, smax, opt, 1};
  cohe_bit_ = groupbit;
  use_groups_ = use_groups_ || groupbit != 1;
}

void DemEngine::set_gravity(double mag, double gx, double gy, double gz, int groupbit)
{
  const double len = std::sqrt(gx * gx + gy * gy + gz * gz);
  have_gravity_ = true;
  grav_bit_ = groupbit;
  post_freeze_ = (post_freeze_ & ~1) | (freeze_bit_ ? 1 : 0);
  use_groups_ = use_groups_ || groupbit != 1;
  gacc_[0] = len > 0 ? mag * (gx / len) : 0.0;
  gacc_[1] = len > 0 ? mag * (gy / len) : 0.0;
  gacc_[2] = len > 0 ? mag * (gz / len) : 0.0;
}

void DemEngine::set_fdrag(double carrier_rho, int groupbit)
{
  have_fdrag_ = true;
  post_freeze_ = (post_freeze_ & ~2) | (freeze_bit_ ? 2 : 0);
  carrier_rho_ = carrier_rho;
  fdrag_bit_ = groupbit;
  use_groups_ = use_groups_ || groupbit != 1;
}

void DemEngine::set_freeze(int groupbit)
{
  if (freeze_bit_) fail("More than one fix freeze");   // [3P] fix_freeze.cpp
  if (!roots_) fail("fix freeze is not available with the LDS-staged kernel (SF_LDS)");
  freeze_bit_ = groupbit;
  use_groups_ = true;
  mark_frozen();
}

void DemEngine::mark_frozen()
{
  if (!freeze_bit_ || !nlocal_) return;
  k_mark_frozen<<<div_up(nlocal_, 256), 256, 0, stream_>>>(om_[0].as<double4>(), om_[1].as<double4>(), mask_.as<int>(),
                                                           nlocal_, freeze_bit_);
}

int DemEngine::group_bit(const std::string& name) const
{
  auto it = groups_.find(name);
  if (it == groups_.end()) fail("Could not find group ID %s", name.c_str());
  return it->second;
}

int DemEngine::new_group_bit(const std::string& name)
{
  auto it = groups_.find(name);
  if (it != groups_.end()) return it->second;          // LAMMPS adds atoms to an existing group
  if (groups_.size() >= 30) fail("Too many groups");
  const int bit = 1 << (int)groups_.size();
  groups_[name] = bit;
  return bit;
}

void DemEngine::group_type(const std::string& name, int op, int v1, int v2, const std::vector<int>& list)
{
  const int bit = new_group_bit(name);
  GroupTypeArgs A;
  A.op = op;
  A.v1 = v1;
  A.v2 = v2;
  A.nlist = (int)list.size();
  if (A.nlist > 16) fail("group %s type: more than 16 types", name.c_str());
  for (int k = 0; k < A.nlist; k++) A.list[k] = list[k];
  if (nlocal_)
    k_group_type<<<div_up(nlocal_, 256), 256, 0, stream_>>>(mask_.as<int>(), type_.as<int>(), nlocal_, bit, A);
}

void DemEngine::group_combine(const std::string& name, int mode, const std::vector<std::string>& args)
{
  if (args.empty() || args.size() > 16) fail("Illegal group command");
  GroupCombineArgs A;
  A.mode = mode;
  A.n = (int)args.size();
  for (int k = 0; k < A.n; k++) A.bits[k] = group_bit(args[k]);
  const int bit = new_group_bit(name);
  if (nlocal_) k_group_combine<<<div_up(nlocal_, 256), 256, 0, stream_>>>(mask_.as<int>(), nlocal_, bit, A);
}

void DemEngine::set_velocity_group(int groupbit, double vx, double vy, double vz)
{
  if (!nlocal_) return;
  k_set_velocity_group<<<div_up(nlocal_, 256), 256, 0, stream_>>>(vm_[cur_].as<double4>(), mask_.as<int>(), groupbit,
                                                                  nlocal_, vx, vy, vz);
}

void DemEngine::add_wall(int dim, bool lo_null, double lo, bool hi_null, double hi, double kn, bool kt_null,
                         double kt, double gamman, bool gammat_null, double gammat, double xmu, int dampflag,
                         bool granfix, int groupbit, bool cylinder)
{
  if (nwalls_ >= kMaxWalls) fail("too many wall fixes (max %d)", kMaxWalls);
  if (!cylinder && periodic_[dim]) fail("Cannot use wall in periodic dimension");  // fix_wall_granFix.cpp:143-148
  if (!granfix && gran_.style == 2)
    fail("Fix wall/gran is incompatible with Pair style");  // stock wall/gran does not know hertzFix
  WallParams& W = walls_[nwalls_];
  wall_motion_[nwalls_] = WallMotion();
  WallMotion& M = wall_motion_[nwalls_++];
  W.dim = dim;
  W.bit = groupbit;
  W.post_freeze = freeze_bit_ ? 1 : 0;
  use_groups_ = use_groups_ || groupbit != 1;
  W.lo = M.lo0 = lo_null ? -1.0e20 : lo;
  W.hi = M.hi0 = hi_null ? 1.0e20 : hi;
  W.cylradius = 0.0;
  W.vwall[0] = W.vwall[1] = W.vwall[2] = 0.0;
  W.vrot = 0.0;
  gran_settings(W.gp, gran_.style ? gran_.style : 1, kn, kt_null, kt, gamman, gammat_null, gammat, xmu,
                dampflag, 1.0);
}

void DemEngine::wall_cylinder(double radius)
{
  if (!nwalls_) fail("wall_cylinder: no wall registered");
  if (periodic_[0] || periodic_[1]) fail("Cannot use wall in periodic dimension");   // fix_wall_granFix.cpp:149-150
  WallParams& W = walls_[nwalls_ - 1];
  W.dim = 3;
  W.cylradius = radius;
  W.lo = W.hi = 0.0;
}

void DemEngine::wall_motion(int kind, int axis, double a, double b)
{
  if (!nwalls_) fail("wall_motion: no wall registered");
  const WallParams& W = walls_[nwalls_ - 1];
  WallMotion& M = wall_motion_[nwalls_ - 1];
  if (M.wiggle || M.shear) fail("Cannot wiggle and shear fix wall/granFix");          // :152-153
  if (axis < 0 || axis > 2) fail("Illegal fix wall/gran command");
  if (kind == 1) {
    if (W.dim == 3 && axis != 2) fail("Invalid wiggle direction for fix wall/granFix");   // :154-155
    if (!(b > 0.0)) fail("Illegal fix wall/gran command");
    M.wiggle = 1;
    M.axis = axis;
    M.amplitude = a;
    M.period = b;
  } else {
    if (W.dim < 3 && axis == W.dim) fail("Invalid shear direction for fix wall/granFix");   // :156-161
    M.shear = 1;
    M.axis = axis;
    M.vshear = a;
  }
}

void DemEngine::set_velocity_all(double vx, double vy, double vz)
{
  if (!nlocal_) return;
  k_set_velocity<<<div_up(nlocal_, 256), 256, 0, stream_>>>(vm_[cur_].as<double4>(), nlocal_, vx, vy, vz);
}

double DemEngine::max_radius() { return rmax_; }

// images LAMMPS' ghost creation would have made of the owned atoms (dimension by dimension, the later ones also of the
// ghosts of the earlier ones: k_ghost_select's criteria): an atom within the ghost cutoff of a periodic face counts once
// per face and dimension, the combinations multiply
// (xhold: the positions of the last list build, [3][cap] -- what LAMMPS made its ghosts from)
__global__ __launch_bounds__(1024) static void k_count_images(const double* xhold, size_t cap, int nlocal, double3 lo, double3 hi,
                                                              int3 per, double cut, int* out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int c = 0;
  if (i < nlocal) {
    const double xs[3] = {xhold[i], xhold[cap + i], xhold[2 * cap + i]};
    const double los[3] = {lo.x, lo.y, lo.z}, his[3] = {hi.x, hi.y, hi.z};
    const int ps[3] = {per.x, per.y, per.z};
    int prod = 1;
    for (int k = 0; k < 3; k++)
      if (ps[k]) prod *= 1 + (xs[k] >= los[k] && xs[k] <= los[k] + cut ? 1 : 0) + (xs[k] >= his[k] - cut && xs[k] <= his[k] ? 1 : 0);
    c = prod - 1;
  }
  c = block_sum_int_1024(c);
  if (threadIdx.x == 0 && c) atomicAdd(out, c);
}

int DemEngine::nghost()
{
  if (!ghost_free_ || !have_list_ || !nlocal_) return nghost_;
  if (nimages_ < 0) {
    int* d = nullptr;
    int h = 0;
    SF_HIP(hipMalloc(&d, sizeof(int)));
    SF_HIP(hipMemsetAsync(d, 0, sizeof(int), stream_));
    k_count_images<<<div_up(nlocal_, 1024), 1024, 0, stream_>>>(
        xhold_.as<double>(), cap_, nlocal_, double3{boxlo_[0], boxlo_[1], boxlo_[2]}, double3{boxhi_[0], boxhi_[1], boxhi_[2]},
        int3{periodic_[0] ? 1 : 0, periodic_[1] ? 1 : 0, periodic_[2] ? 1 : 0}, cutneighmax(), d);
    SF_HIP(hipMemcpyAsync(&h, d, sizeof(int), hipMemcpyDeviceToHost, stream_));
    SF_HIP(hipStreamSynchronize(stream_));
    SF_HIP(hipFree(d));
    nimages_ = h;
  }
  return nghost_ + nimages_;
}

double DemEngine::cutneighmax() const
{
  double c = 0.0;
  if (gran_.style) c = 2.0 * rmax_;   // [3P] PairGranHookeHistory::init_one: maxrad_dynamic[i] + maxrad_dynamic[j]
  if (lub_.enabled && lub_.cut_global > c) c = lub_.cut_global;
  return c + lskin();
}

// (a one-thread kernel: ~2 us on the stream; a 4-byte host-to-device copy is a 5 us blit each, and a rebuild resets
// sixteen of them)
__global__ static void k_set_flags3(int* flags, int i0, int v0, int i1, int v1, int i2, int v2)
{
  if (threadIdx.x == 0) {
    flags[i0] = v0;
    flags[i1] = v1;
    flags[i2] = v2;
  }
}
__global__ static void k_set_flags(int* flags, int idx, int count, int value)
{
  if ((int)threadIdx.x < count) flags[idx + threadIdx.x] = value;
}

void DemEngine::reset_flag(int idx, int value) { reset_flags(idx, 1, value); }

void DemEngine::reset_flags(int idx, int count, int value)
{
  for (int k = 0; k < count; k++) h_flags_[idx + k] = value;
  k_set_flags<<<1, 32, 0, stream_>>>(d_flags_, idx, count, value);
}

void DemEngine::set_flags3(int i0, int v0, int i1, int v1, int i2, int v2)
{
  h_flags_[i0] = v0;
  h_flags_[i1] = v1;
  h_flags_[i2] = v2;
  k_set_flags3<<<1, 32, 0, stream_>>>(d_flags_, i0, v0, i1, v1, i2, v2);
}

// The flag words travel to a pinned host block, and the host does not wait for the STREAM (a hipStreamSynchronize returns
// ~10 us after the words have landed: a fifth of a 100 k-grain rebuild, paid two or three times per rebuild) but for the
// words themselves.  A one-wave kernel writes them into the block (the pinned allocation is mapped into the device's address
// space), fences at system scope, and only then stores the call's sequence number into the block's last word with release
// semantics: a host that sees the number sees every word written before it.  (A plain asynchronous copy of the 128 bytes
// carries no such order between its two 64-byte halves: a first form that waited for a sentinel in the last word to be
// overwritten could, once in ~1e5 reads, look at a first half that had not landed yet.)  Bounded: after 20 ms without the
// number the stream is synchronised the ordinary way.  SF_FLAG_SPIN=0: always the ordinary way.
__global__ static void k_publish_flags(const int* flags, int* host_block, int seq)
{
  const int t = threadIdx.x;
  if (t < F_ARRIVAL) __hip_atomic_store(&host_block[t], flags[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __atomic_thread_fence(__ATOMIC_RELEASE);   // (system scope: every lane's store above is out before ...)
  __builtin_amdgcn_s_barrier();
  if (t == 0) __hip_atomic_store(&host_block[F_ARRIVAL], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // ... the number
}

void DemEngine::flags_copy_begin()
{
  static_assert(F_ARRIVAL == F_NFLAGS - 1, "the arrival word is the last of the block");
  flags_seq_ = flags_seq_ == INT_MAX ? 1 : flags_seq_ + 1;
  k_publish_flags<<<1, 64, 0, stream_>>>(d_flags_, h_flags_, flags_seq_);
}

void DemEngine::flags_copy_wait()
{
  static const bool spin = !(getenv("SF_FLAG_SPIN") && !atoi(getenv("SF_FLAG_SPIN")));
  const int* hf = h_flags_;
  bool seen = false;
  if (spin) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned it = 0;; it++) {
      if (__atomic_load_n(&hf[F_ARRIVAL], __ATOMIC_ACQUIRE) == flags_seq_) {
        seen = true;
        break;
      }
      if ((it & 1023u) == 1023u &&
          std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > 20.0)
        break;
    }
  }
  if (!seen) sync();   // (the kernel is complete: its stores are)
}

void DemEngine::read_flags()
{
  flags_copy_begin();
  flags_copy_wait();
  if (xcd_sample_pending_) apply_xcd_sample();
}

// The shares of the eight XCDs follow what the timed launch measured: XCD x took T_x for its share, so its rate is
// share / T_x; the new shares are proportional to the rates (damped, bounded).  An XCD is slower when its range is
// harder -- the ends of the sorted range gather across the periodic face, a bed that does not fill the box has rows of
// different lengths -- not because of anything in the results, which no placement can change.
void DemEngine::apply_xcd_sample()
{
  xcd_sample_pending_ = false;
  static const bool dbg = getenv("SF_DEBUG_XCD") != nullptr;
  int t0 = INT_MAX;
  for (int x = 0; x < 8; x++) t0 = std::min(t0, h_xcd_time_[64 * x]);
  double T[8], mean = 0.0;
  for (int x = 0; x < 8; x++) {
    T[x] = (double)(h_xcd_time_[64 * x + 32] - t0);   // 10 ns units
    if (h_xcd_time_[64 * x] == INT_MAX || T[x] <= 100.0 || T[x] > 1.0e7) return;   // (an early-exit launch, or the clock wrapped)
    mean += T[x] / 8.0;
  }
  double w[8], wsum = 0.0;
  for (int x = 0; x < 8; x++) {
    w[x] = xcd_weight_[x] * std::pow(mean / T[x], 0.7);
    wsum += w[x] / 8.0;
  }
  for (int x = 0; x < 8; x++) xcd_weight_[x] = std::min(1.2, std::max(0.8, w[x] / wsum));
  xcd_weighted_ = true;
  if (dbg) {
    fprintf(stderr, "[sedifoam_amd] XCD times [us]");
    for (int x = 0; x < 8; x++) fprintf(stderr, " %.1f", T[x] * 0.01);
    fprintf(stderr, "  -> shares");
    for (int x = 0; x < 8; x++) fprintf(stderr, " %.3f", xcd_weight_[x]);
    fprintf(stderr, "\n");
  }
}

DemPtrs DemEngine::ptrs(int in_buf) const
{
  DemPtrs P;
  const int ob = in_buf ^ 1;
  P.xr_in = xr_[in_buf].as<double4>();
  P.vm_in = vm_[in_buf].as<double4>();
  P.om_in = om_[in_buf].as<double4>();
  P.xr_out = xr_[ob].as<double4>();
  P.vm_out = vm_[ob].as<double4>();
  P.om_out = om_[ob].as<double4>();
  P.force = force_.as<double4>();
  P.torque = torque_.as<double4>();
  P.neigh = neigh_.as<int>();
  P.numneigh = numneigh_.as<int>();
  P.shear_in = shear_[in_buf].as<double>();
  P.shear_out = shear_[ob].as<double>();
  P.fdrag = fdrag_.as<double>();
  P.DuDt = DuDt_.as<double>();
  P.vOld = vOld_.as<double>();
  P.wshear = wshear_.as<double>();
  P.wtouch = wtouch_.as<unsigned char>();
  P.xhold = xhold_.as<double>();
  P.mask = mask_.as<int>();
  P.flags = d_flags_;
  P.nloc = nloc_.as<unsigned short>();
  P.tile_first = tile_tab_;
  for (int f = 0; f < 2; f++) {
    P.sendslot[f] = tx_ready_ ? sendslot_.as<int>() + (size_t)f * sendslot_.cap : nullptr;
    P.tx[f] = tx_ptr_[f];
  }
  P.tx_sendbuf = tx_sendbuf_;
  P.bslot = bslot_.as<int>();
  P.tx_blkptr = d_blkptr_ ? d_blkptr_ + (size_t)tx_par_ * kMaxDirs : nullptr;
  static_assert(sizeof(size_t) == sizeof(double*), "the count row of the block table");
  P.tx_blkcnt = d_blkptr_ ? reinterpret_cast<const size_t*>(d_blkptr_ + 2 * (size_t)kMaxDirs) : nullptr;
  P.tx_hdr_off = tx_hdr_off_;
  P.tx_blkshift = nullptr;
  P.gs_sync = nullptr;
  P.gs_my_sync = nullptr;
  P.gs_count = nullptr;
  P.xcd_time = d_xcd_time_;
  P.pq_head = d_pq_head_;
  P.tile_last = tile_tab_ ? tile_tab_ + tile_alloc_ : nullptr;
  P.stage_start = tile_tab_ ? tile_tab_ + 3 * tile_alloc_ : nullptr;
  P.stage_idx = stage_idx_;
  return P;
}

StepParams DemEngine::step_params(int mode, int kstep) const
{
  StepParams S;
  memset(&S, 0, sizeof(S));
  S.nlocal = nlocal_;
  S.cap = (int)cap_;
  S.mode = mode;
  S.kstep = kstep;
  S.nslots = M_;
  S.dt = dt_;
  S.trigger_sq = (0.5 * skin_) * (0.5 * skin_);
  S.roots = roots_ ? 1 : 0;
  for (int k = 0; k < 3; k++) S.prd[k] = boxhi_[k] - boxlo_[k];
  S.part = 0;
  S.nb = 0;
  S.trig_test = F_TRIGGER;
  S.trig_set = F_TRIGGER;
  S.trig_add = 0;
  S.margin_sq = 0.0;
  S.gran = gran_;
  S.cohe = cohe_;
  S.lub = lub_;
  S.nwalls = nwalls_;
  S.xcd_remap = opt_xcd_remap_;
  static const int sweep_env = getenv("SF_SWEEP_REVERSE") ? atoi(getenv("SF_SWEEP_REVERSE")) : 0;   // (measured neutral at 1 M grains: off)
  S.sweep_rev = (sweep_env && mode != 2) ? (int)((run_base_step_ + kstep) & 1) : 0;
  S.stage_cap = stage_cap_;
  // wall positions / velocities of the LAMMPS step this launch evaluates: post_force of step n sees ntimestep = n,
  // the setup evaluation sees the value the run starts from (fix_wall_granFix.cpp:255-264)
  const long long steps = (mode == 2 ? nsteps_ : run_base_step_ + kstep + 1) - wall_time_origin_;
  for (int w = 0; w < nwalls_; w++) {
    WallParams W = walls_[w];
    const WallMotion& M = wall_motion_[w];
    if (M.wiggle) {
      const double om = 2.0 * 3.14159265358979323846 / M.period;   // :165
      const double arg = om * (double)steps * dt_;
      if (W.dim == M.axis) {
        W.lo = M.lo0 + M.amplitude - M.amplitude * cos(arg);
        W.hi = M.hi0 + M.amplitude - M.amplitude * cos(arg);
      }
      W.vwall[M.axis] = M.amplitude * om * sin(arg);
    } else if (M.shear) {
      if (W.dim == 3 && M.axis != 2) W.vrot = M.vshear;
      else W.vwall[M.axis] = M.vshear;
    }
    S.wall[w] = W;
  }
  S.have_gravity = have_gravity_;
  for (int k = 0; k < 3; k++) S.gacc[k] = gacc_[k];
  S.have_fdrag = have_fdrag_;
  S.carrier_rho = carrier_rho_;
  S.have_nve = have_nve_;
  S.use_groups = use_groups_ ? 1 : 0;
  S.nve_bit = nve_bit_;
  S.grav_bit = grav_bit_;
  S.fdrag_bit = fdrag_bit_;
  S.cohe_bit = cohe_bit_;
  S.freeze_bit = freeze_bit_;
  S.post_freeze = post_freeze_;
  return S;
}

template <int STYLE, int LPA, bool TP, int NTP, bool GS = false>
static void launch_substep_lpa(bool cohe, bool lub, dim3 grid, int block, hipStream_t s, const DemPtrs& P,
                               const StepParams& S)
{
  if (cohe && lub) k_substep<STYLE, true, true, LPA, TP, NTP, GS><<<grid, block, 0, s>>>(P, S);
  else if (cohe) k_substep<STYLE, true, false, LPA, TP, NTP, GS><<<grid, block, 0, s>>>(P, S);
  else if (lub) k_substep<STYLE, false, true, LPA, true, NTP, GS><<<grid, block, 0, s>>>(P, S);   // (lubrication needs v, omega
  else k_substep<STYLE, false, false, LPA, TP, NTP, GS><<<grid, block, 0, s>>>(P, S);            //  of every neighbour anyway)
}

template <int STYLE, int LPA, int NTP, bool GS = false>
static void launch_substep_tp(bool cohe, bool lub, bool tp, dim3 grid, int block, hipStream_t s, const DemPtrs& P,
                              const StepParams& S)
{
  if (lub) tp = true;   // one instantiation
  tp ? launch_substep_lpa<STYLE, LPA, true, NTP, GS>(cohe, lub, grid, block, s, P, S)
     : launch_substep_lpa<STYLE, LPA, false, NTP, GS>(cohe, lub, grid, block, s, P, S);
}

// ntp: non-temporal policy of the row streams (sf_dem_kernels.h); systems small enough for several lanes per atom
// always fit the memory-side cache (policy 0)
template <int STYLE>
static void launch_substep_style(bool cohe, bool lub, int lpa, bool tp, int ntp, dim3 grid, int block, hipStream_t s,
                                 const DemPtrs& P, const StepParams& S)
{
  // ghost slots (sf_halo_rccl.hip): the per-GPU share of a decomposed bed, nothing non-temporal -- one variant per lane count
  if (S.gs_on) {
    if (lpa == 4) launch_substep_tp<STYLE, 4, 0, true>(cohe, lub, tp, grid, block, s, P, S);
    else if (lpa == 2) launch_substep_tp<STYLE, 2, 0, true>(cohe, lub, tp, grid, block, s, P, S);
    else launch_substep_tp<STYLE, 1, 0, true>(cohe, lub, tp, grid, block, s, P, S);
    return;
  }
  if (lpa == 4) launch_substep_tp<STYLE, 4, 0>(cohe, lub, tp, grid, block, s, P, S);
  else if (lpa == 2) launch_substep_tp<STYLE, 2, 0>(cohe, lub, tp, grid, block, s, P, S);
  else if (ntp == 0) launch_substep_tp<STYLE, 1, 0>(cohe, lub, tp, grid, block, s, P, S);
  else if (ntp == 1) launch_substep_tp<STYLE, 1, 1>(cohe, lub, tp, grid, block, s, P, S);
  else if (ntp == 3) launch_substep_tp<STYLE, 1, 3>(cohe, lub, tp, grid, block, s, P, S);
  else launch_substep_tp<STYLE, 1, 2>(cohe, lub, tp, grid, block, s, P, S);
}

#ifdef SF_EXP_PERSIST
// persistent tiles (k_substep_persist): the plain Hertz contact kernel, one lane per atom
static void launch_substep_persist(bool tp, int ntp, dim3 grid, hipStream_t s, const DemPtrs& P, const StepParams& S)
{
  if (tp) {
    if (ntp == 0) k_substep_persist<2, false, false, true, 0><<<grid, 64, 0, s>>>(P, S);
    else if (ntp == 1) k_substep_persist<2, false, false, true, 1><<<grid, 64, 0, s>>>(P, S);
    else if (ntp == 3) k_substep_persist<2, false, false, true, 3><<<grid, 64, 0, s>>>(P, S);
    else k_substep_persist<2, false, false, true, 2><<<grid, 64, 0, s>>>(P, S);
  } else {
    if (ntp == 0) k_substep_persist<2, false, false, false, 0><<<grid, 64, 0, s>>>(P, S);
    else if (ntp == 1) k_substep_persist<2, false, false, false, 1><<<grid, 64, 0, s>>>(P, S);
    else if (ntp == 3) k_substep_persist<2, false, false, false, 3><<<grid, 64, 0, s>>>(P, S);
    else k_substep_persist<2, false, false, false, 2><<<grid, 64, 0, s>>>(P, S);
  }
}

#endif

template <int STYLE, bool COHE, bool LUB>
static void launch_lds_one(dim3 grid, size_t lds, hipStream_t s, const DemPtrs& P, const StepParams& S)
{
  static size_t granted = 0;   // dynamic LDS above the 64 KiB default has to be requested per kernel
  if (lds > 65536 && lds > granted) {
    SF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_substep_lds<STYLE, COHE, LUB>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    granted = 160 * 1024;
  }
  static const int threads = getenv("SF_LDS_THREADS") ? atoi(getenv("SF_LDS_THREADS")) : 256;
  k_substep_lds<STYLE, COHE, LUB><<<grid, threads, lds, s>>>(P, S);
}

template <int STYLE>
static void launch_lds_style(bool cohe, bool lub, dim3 grid, size_t lds, hipStream_t s, const DemPtrs& P,
                             const StepParams& S)
{
  if (cohe && lub) launch_lds_one<STYLE, true, true>(grid, lds, s, P, S);
  else if (cohe) launch_lds_one<STYLE, true, false>(grid, lds, s, P, S);
  else if (lub) launch_lds_one<STYLE, false, true>(grid, lds, s, P, S);
  else launch_lds_one<STYLE, false, false>(grid, lds, s, P, S);
}

int DemEngine::lanes_per_atom(int nwork) const
{
  if (opt_lpa_ == 1 || opt_lpa_ == 2 || opt_lpa_ == 4) return opt_lpa_;
  if (nwork < 20 * 1024) return 4;
  if (nwork >= 300 * 1024) return 1;
  // In between the kernel is a few rounds of resident waves (3 per SIMD): one lane per atom is N/64 waves that each walk
  // all ~12 slots, two lanes per atom twice as many waves of ~0.55 the length.  Whichever needs the cheaper whole number
  // of rounds wins -- measured at 63 k / 126 k / 170 k / 200 k / 250 k / 300 k grains: 2 / either / 1 / 2 / 2 / 1 lanes,
  // up to 10 % apart (profiles/r03_README.md)
  static const int slots = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus * 4 * 3;
  }();
  const double r = (double)nwork / 64.0 / (double)slots;
  // (a loose bed -- touching neighbours first, the slot loop as long as the wave's longest row -- gains a little more from the
  // shorter waves: 100 k grains +1 %, 300 k +2 % with two lanes where the packed-bed rule says one; profiles/r06_README.md)
  if (touch_first_) return 0.5 * std::ceil(2.0 * r) <= std::ceil(r) ? 2 : 1;
  return 0.55 * std::ceil(2.0 * r) < std::ceil(r) ? 2 : 1;
}

void DemEngine::launch_substep(int in_buf, int mode, int kstep, int part)
{
  if (!nlocal_) {
    // (ghost slots: the other ranks wait for this one's flag whether it owns atoms or not)
    if (gs_ready_ && brick_ && part == 0 && !lds_active_ && mode != 2) {
      k_gs_idle<<<1, 64, 0, stream_>>>(d_gs_sync_, d_flags_, (int)gs_seq_, kstep, mode == 0 ? 1 : 0);
      gs_seq_++;
      tx_written_ = mode == 0;
    }
    return;
  }
  if (part == 2 && !nb_) return;   // (no event pair opened: the interior part then times itself)
  DemPtrs P = ptrs(in_buf);
  StepParams S = step_params(mode, kstep);
  // ghost slots: every stepping launch of a decomposed engine reads the ghosts of other GPUs from the area of its number's
  // parity and (mode 0) writes its border records into the neighbours' area of the next parity; the setup evaluation
  // (mode 2) runs on the ghosts the border exchange has just put into the record arrays
  const bool gs = gs_ready_ && brick_ && part == 0 && !lds_active_ && mode != 2;
  if (gs) {
    // (before the launch takes its number: a refused launch must not leave a gap in the sequence the neighbours count)
    static const int gs_block_env = getenv("SF_BLOCK") ? atoi(getenv("SF_BLOCK")) : 0;
    if (gs_block_env && gs_block_env != 64)
      fail("ghost slots: the hand-off at the end of the sub-step kernel is written for one-wave workgroups (SF_BLOCK)");
    S.gs_on = 1;
    S.gs_seq = (int)gs_seq_;
    S.gs_wait = h_gs_sync_.world > 1 ? 1 : 0;   // (one rank that exchanges with itself: stream order is the hand-off)
    S.gs_world = h_gs_sync_.world;
    S.gs_rank = h_gs_sync_.rank;
    P.gs_my_sync = h_gs_sync_.my_sync;
    P.gs_sync = d_gs_sync_;
    P.gs_count = d_gs_count_;
    // (the records of the NEXT launch: the neighbours read them from the buffers of that launch's parity)
    P.tx_blkptr = d_gsblk_ + (size_t)((gs_seq_ + 1) & 1) * 3 * kMaxDirs;
    P.tx_blkshift = reinterpret_cast<const double*>(d_gsblk_ + 6 * (size_t)kMaxDirs);
    gs_seq_++;
  }
  if (part) {
    // overlapped halo: both parts of sub-step k test the vote published by the exchange of sub-step k-1; an
    // interior trigger can only be voted one exchange later, hence "+1" (sub-step k+1 still runs everywhere)
    S.part = part;
    S.nb = nb_;
    S.n_lo = n_lo_;
    S.n_hi = n_hi_;
    S.trig_test = F_VOTE0 + ((kstep + 1) & 1);
    S.trig_set = F_TRIG_LOCAL;
    S.trig_add = part == 1 ? 1 : 0;
    const double half_margin = 0.5 * (lskin() - skin_);
    S.margin_sq = half_margin * half_margin;
  }
  // the forward halo of the exchange that follows: written by the kernel that integrates the border atoms
  if (part != 1) tx_written_ = false;   // (the interior part sends nothing: what the boundary part wrote stands)
  if (tx_ready_ && mode == 0 && !lds_active_) S.tx_nhdr = tx_direct_ ? 0 : tx_nhdr_;   // (direct: votes travel with the flags)
  if (tx_ready_ && brick_ && mode == 0 && part == 0 && !lds_active_) {
    S.tx_fused = 2;
    if (bslot_.cap != cap_) fail("launch_substep: the record-slot table has stride %zu, the engine %zu", bslot_.cap, cap_);
    const double cut = cutneighmax() + skin_;   // (an atom is within skin/2 of where the send lists were made)
    for (int k = 0; k < 3; k++) {
      S.tx_lo3[k] = ext_[k] ? sublo_[k] + cut : -1.0e300;
      S.tx_hi3[k] = ext_[k] ? subhi_[k] - cut : 1.0e300;
    }
    tx_written_ = true;
  } else if (tx_ready_ && mode == 0 && part != 1 && !lds_active_) {
    S.tx_fused = 1;
    const double cut = cutneighmax() + skin_;   // (an atom is within skin/2 of where the border lists were made)
    S.tx_xlo = sublo_[0] + cut;
    S.tx_xhi = subhi_[0] - cut;
    S.tx_shift[0] = tx_shift_[0];
    S.tx_shift[1] = tx_shift_[1];
    S.tx_n[0] = tx_n_[0];
    S.tx_n[1] = tx_n_[1];
    tx_written_ = true;
  }
  const bool cohe = cohe_.enabled, lub = lub_.enabled;
  // one event pair per sub-step: around the single kernel, or from the boundary part to the interior part
  hipEvent_t e0 = nullptr, e1 = nullptr;
  // Sampled: an event pair around EVERY launch costs ~10 us per sub-step (the record is a system-scope release, the
  // caches are written back before the next kernel starts) -- 5 % at 1 M atoms, 50 % at 10 k.
  static const int prof_stride = getenv("SF_PROF_STRIDE") ? std::max(1, atoi(getenv("SF_PROF_STRIDE"))) : 8;
  if (profiling_ && kstep % prof_stride == 0) {
    if (part != 1 || !prof_open_) {
      if (prof_used_ + 2 > prof_ev_.size()) {
        for (int k = 0; k < 2; k++) {
          hipEvent_t e;
          SF_HIP(hipEventCreate(&e));
          prof_ev_.push_back(e);
        }
      }
      if (prof_step_.size() < prof_ev_.size() / 2) prof_step_.resize(prof_ev_.size() / 2);
      prof_step_[prof_used_ / 2] = kstep;
      e0 = prof_ev_[prof_used_++];
      e1 = prof_ev_[prof_used_++];
      SF_HIP(hipEventRecord(e0, stream_));
      prof_open_ = (part == 2);
      if (part == 2) e1 = nullptr;
    } else {
      e1 = prof_ev_[prof_used_ - 1];
      prof_open_ = false;
    }
  }
  if (lds_active_ && part) fail("the LDS-staged kernel has no boundary/interior split (SF_LDS with the overlapped halo)");
  if (lds_active_) {
    // one workgroup per tile; LDS = staged x (32 B) + v (32 B) + omega (24 B) per atom of the extended tile
    const dim3 grid(ntiles_);
    const size_t lds = (size_t)stage_cap_ * 88;
    switch (gran_.style) {
      case 2: launch_lds_style<2>(cohe, lub, grid, lds, stream_, P, S); break;
      case 3:   // (plain gran/hooke: the Hookean kernel, the law itself branches on GranParams::style)
      case 1: launch_lds_style<1>(cohe, lub, grid, lds, stream_, P, S); break;
      default: launch_lds_style<0>(cohe, lub, grid, lds, stream_, P, S); break;
    }
  } else {
    // enough workgroups to cover the 256 CUs even for small beds (one atom per lane either way)
    static const int block_env = getenv("SF_BLOCK") ? atoi(getenv("SF_BLOCK")) : 0;
    const int nwork = part == 2 ? nb_ : (part == 1 ? n_hi_ - n_lo_ : nlocal_);
    if (nwork <= 0) return;
    // lanes per atom: small systems are latency bound (one lane walks all ~12 neighbours).  Measured: 10 k atoms
    // 17.3 -> 10.5 us per sub-step with 4 lanes, while at 100 k (1.5 waves per SIMD already) more lanes are slower
    const int lpa = lanes_per_atom(nwork);
    const long long lanes = (long long)nwork * lpa;
    // one wave per workgroup: the dispatcher then balances single waves (a 256-thread workgroup holds its CU slots
    // until its slowest wave is done); measured 207.0 -> 203.2 us per sub-step at 1 M atoms, never slower below
    const int block = block_env ? block_env : 64;
    if (gs && block != 64) fail("ghost slots: the hand-off at the end of the sub-step kernel is written for one-wave workgroups (SF_BLOCK)");
    dim3 grid((unsigned)((lanes + block - 1) / block));
    // The XCDs do not finish together when each gets the same number of workgroups: the two that hold the ends of the
    // sorted range gather across the periodic face from lines no neighbour of theirs has pulled into their L2, and run
    // ~5 % longer (workgroup timeline, profiles/r04_*).  xcd_weight_[x]: relative share of XCD x.
    const bool xcd_can = part == 0 && S.xcd_remap == 1 && grid.x >= 2048 && mode == 0;
    if (xcd_can && xcd_auto_ && xcd_countdown_ > 0 && --xcd_countdown_ == 0 && !xcd_sample_pending_) {
      SF_HIP(hipMemcpyAsync(d_xcd_time_, d_xcd_time_ + 512, sizeof(int) * 512, hipMemcpyDeviceToDevice, stream_));
      S.xcd_time = 1;
    }
    if (part == 0 && S.xcd_remap == 1 && xcd_weighted_ && grid.x >= 64) {
      const int nb = (int)grid.x;
      double wsum = 0.0;
      for (int x = 0; x < 8; x++) wsum += xcd_weight_[x];
      int first = 0, most = 0;
      double acc = 0.0;
      for (int x = 0; x < 8; x++) {
        acc += xcd_weight_[x];
        const int end = x == 7 ? nb : std::min(nb, (int)std::llround(nb * acc / wsum));
        S.xcd_first[x] = first;
        S.xcd_count[x] = std::max(0, end - first);
        most = std::max(most, S.xcd_count[x]);
        first = std::max(first, end);
      }
      S.xcd_remap = 2;
      grid = dim3((unsigned)(8 * most));
    }
    stamp_last_grid_ = grid.x;
#ifdef SF_EXP_PERSIST
    // Persistent tiles: the resident waves walk the tiles of their XCD's range and request the next tile's records under
    // the current tile's epilogue (k_substep_persist).  The plain Hertz kernel with one lane per atom and one-wave tiles,
    // when the launch is at least three rounds of resident waves.
    static const int resident_per_xcd = [] {
      int dev = 0, cus = 256;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      return std::max(1, cus * 4 * 3 / 8);
    }();
    const int pw = opt_persist_waves_ > 0 ? opt_persist_waves_ : resident_per_xcd;
    const int tiles = (int)((lanes + 63) / 64);
    const bool persist = opt_persist_ != 0 && !gs && part == 0 && lpa == 1 && block == 64 && gran_.style == 2 && !cohe &&
                         !lub && S.xcd_remap != 0 && mode != 2 && (opt_persist_ == 1 || tiles >= 3 * 8 * pw);
    if (persist) {
      if (S.xcd_remap == 1) {   // equal shares: the ranges xcd_contiguous_block() gives
        const int q = tiles >> 3, r = tiles & 7;
        for (int x = 0; x < 8; x++) {
          S.xcd_count[x] = x < r ? q + 1 : q;
          S.xcd_first[x] = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
        }
      }
      S.pq_par = pq_par_;
      pq_par_ ^= 1;
      launch_substep_persist(touch_prefetch_, nt_policy_, dim3((unsigned)(8 * pw)), stream_, P, S);
    } else
#endif
    switch (gran_.style) {
      case 2: launch_substep_style<2>(cohe, lub, lpa, touch_prefetch_, nt_policy_, grid, block, stream_, P, S); break;
      case 3:
      case 1: launch_substep_style<1>(cohe, lub, lpa, touch_prefetch_, nt_policy_, grid, block, stream_, P, S); break;
      default: launch_substep_style<0>(cohe, lub, lpa, touch_prefetch_, nt_policy_, grid, block, stream_, P, S); break;
    }
  }
  SF_HIP(hipGetLastError());
  if (e1) SF_HIP(hipEventRecord(e1, stream_));
  if (S.xcd_time) {   // read with the next synchronisation of the stepping loop (read_flags)
    SF_HIP(hipMemcpyAsync(h_xcd_time_, d_xcd_time_, sizeof(int) * 512, hipMemcpyDeviceToHost, stream_));
    xcd_sample_pending_ = true;
  }
#if SF_EXP_STAMP
  // SF_STAMP_FILE=<path> [SF_STAMP_AT=<n>]: the n-th full-size launch (default 60) is recorded workgroup by workgroup
  static const char* stamp_file = getenv("SF_STAMP_FILE");
  static const int stamp_at = getenv("SF_STAMP_AT") ? atoi(getenv("SF_STAMP_AT")) : 60;
  static int stamp_seen = 0;
  static unsigned long long* stamp_buf = nullptr;
  static size_t stamp_blocks = 0, stamp_grid = 0;
  if (stamp_buf && stamp_file && !lds_active_ && part == 0 && mode == 0) stamp_grid = stamp_last_grid_;
  if (stamp_file && !lds_active_ && part == 0 && mode == 0) {
    stamp_seen++;
    if (stamp_seen == stamp_at) {   // arm: the NEXT launch records
      const int lpa = lanes_per_atom(nlocal_);
      stamp_blocks = ((size_t)nlocal_ * lpa + 63) / 64 + 8192;   // (a weighted XCD split launches a few more)
      SF_HIP(hipMalloc(&stamp_buf, sizeof(unsigned long long) * 36 * stamp_blocks));   // (+ 32 phase marks each)
      SF_HIP(hipMemsetAsync(stamp_buf, 0, sizeof(unsigned long long) * 36 * stamp_blocks, stream_));
      SF_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_stamp), &stamp_buf, sizeof(stamp_buf), 0, hipMemcpyHostToDevice, stream_));
    } else if (stamp_seen == stamp_at + 1 && stamp_buf) {
      unsigned long long* none = nullptr;
      SF_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_stamp), &none, sizeof(none), 0, hipMemcpyHostToDevice, stream_));
      // [4 x grid] stamps, then [32 x grid] phase marks (SF_EXP_PHASE); the file starts with the grid size
      std::vector<unsigned long long> h(36 * stamp_blocks);
      SF_HIP(hipMemcpyAsync(h.data(), stamp_buf, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost, stream_));
      SF_HIP(hipStreamSynchronize(stream_));
      if (FILE* f = fopen(stamp_file, "wb")) {
        const unsigned long long g = stamp_grid;
        fwrite(&g, sizeof(g), 1, f);
        fwrite(h.data(), sizeof(unsigned long long), (size_t)36 * stamp_grid, f);
        fclose(f);
      }
      (void)hipFree(stamp_buf);
      stamp_buf = nullptr;
    }
  }
#endif
}

void DemEngine::set_profiling(bool on)
{
  sync();
  prof_used_ = 0;
  profiling_ = on;
  prof_launches_ = 0;
  prof_ms_ = 0.0;
  prof_rebuilds_ = 0;
  prof_rebuild_ms_ = 0.0;
}

void DemEngine::harvest_profile(int last_step)
{
  // event pairs of sub-steps <= last_step belong to launches that really executed; later pairs of the batch were
  // early exits on a stale list (a few microseconds each) and are not counted
  for (size_t q = 0; 2 * q + 1 < prof_used_; q++) {
    if (prof_step_[q] > last_step) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, prof_ev_[2 * q], prof_ev_[2 * q + 1]) == hipSuccess) {
      prof_ms_ += ms;
      prof_launches_++;
    }
  }
}

void DemEngine::get_profile(long long* launches, double* kernel_ms)
{
  sync();
  harvest_profile(INT_MAX);
  prof_used_ = 0;
  *launches = prof_launches_;
  *kernel_ms = prof_ms_;
}

void DemEngine::get_rebuild_profile(long long* rebuilds, double* ms)
{
  *rebuilds = prof_rebuilds_;
  *ms = prof_rebuild_ms_;
}

void DemEngine::launch_ghost_forward(int buf, int kstep, int phase, int trig_word, hipStream_t s)
{
  // root mode: the neighbour words point at the roots, periodic images are not refreshed between rebuilds
  if (!nghost_ || roots_) return;
  k_ghost_forward<<<div_up(nghost_, 256), 256, 0, s ? s : stream_>>>(
      xr_[buf].as<double4>(), vm_[buf].as<double4>(), om_[buf].as<double4>(), gsrc_.as<int>(),
      gshift_.as<double>(), nlocal_, nghost_, cap_, d_flags_, kstep, trig_word, phase);
}

void DemEngine::launch_initial_integrate()
{
  if (!nlocal_ || !have_nve_) return;
  k_initial_integrate<<<div_up(nlocal_, 256), 256, 0, stream_>>>(
      xr_[cur_].as<double4>(), vm_[cur_].as<double4>(), om_[cur_].as<double4>(), force_.as<double4>(),
      torque_.as<double4>(), xhold_.as<double>(), d_flags_ + (overlap_ ? F_TRIG_LOCAL : F_TRIGGER), nlocal_, cap_,
      dt_, (0.5 * skin_) * (0.5 * skin_), use_groups_ ? mask_.as<int>() : nullptr, nve_bit_);
}

// ------------------------------------------------------------------------------------------------
// neighbour rebuild
// ------------------------------------------------------------------------------------------------
void DemEngine::compute_grid()
{
  const double cut = cutneighmax();
  if (!(cut > 0.0)) fail("neighbor cutoff is zero: define a pair style and/or `neighbor <skin> bin`");
  const double* lo = sublo_;   // (= the box in every dimension whose halo is not external)
  const double* hi = subhi_;
  // (single domain, plain keys: the cell size follows the bed -- see sort_sub_)
  const int sub = (!opt_sub_env_ && !have_subdomain_ && opt_tile_ <= 1 && !opt_lds_ && sort_sub_ > 0) ? sort_sub_ : opt_sub_;
  grid_.nbins = 1;
  // ghost-free list build (ghost_free_): every periodic dimension holds at least three cells of the cutoff, so that the
  // wrapped stencil never meets the same atom twice
  ghost_free_ = opt_ghost_free_ != 0 && !have_subdomain_ && opt_tile_ <= 1 && !opt_lds_ && roots_ &&
                (periodic_[0] || periodic_[1] || periodic_[2]);
  for (int k = 0; k < 3 && ghost_free_; k++)
    if (periodic_[k] && (ext_[k] || (int)((hi[k] - lo[k]) / cut) < 3)) ghost_free_ = false;
  for (int k = 0; k < 3; k++) {
    const bool wrapk = ghost_free_ && periodic_[k];
    const bool ext = !wrapk && (periodic_[k] || ext_[k]);
    const double l = ext ? lo[k] - cut : lo[k];
    const double h = ext ? hi[k] + cut : hi[k];
    int n = (int)((h - l) / cut);
    n = std::max(1, std::min(n, 1 << 9)) * sub;   // cells of size >= cut / sub, searched +-sub cells
    grid_.lo[k] = l;
    grid_.n[k] = n;
    grid_.inv[k] = n / (h - l);
    grid_.wrap[k] = wrapk ? 1 : 0;
  }
  grid_.stencil = sub;
  grid_.tile = opt_tile_ > 1 ? opt_tile_ * sub : 1;   // tiles keep their physical size
  grid_.xslow = (have_subdomain_ && !brick_ && grid_.tile <= 1) ? 1 : 0;
  grid_.nbins = 1;
  for (int k = 0; k < 3; k++) {
    grid_.nt[k] = (grid_.n[k] + grid_.tile - 1) / grid_.tile;
    grid_.nbins *= grid_.nt[k] * grid_.tile;
  }
  if ((size_t)grid_.nbins > cell_alloc_) {
    if (cell_start_) SF_HIP(hipFree(cell_start_));
    cell_alloc_ = ((size_t)grid_.nbins + grid_.nbins / 8 + 16 + 3) & ~(size_t)3;   // (the four tables stay 16-byte aligned)
    SF_HIP(hipMalloc(&cell_start_, sizeof(int) * 4 * cell_alloc_));
    hist_clean_ = false;
  }
}

// k_build_neigh<true>: [M][128] parked words + [M][128] image-code bytes of dynamic LDS per block (160 KB per CU)
static constexpr size_t kBuildLdsPerSlot = 128 * 5;
static constexpr size_t kBuildLdsMax = 160 * 1024;

bool DemEngine::build_parks_in_lds() const
{
  static const bool env = !(getenv("SF_BUILD_LDS") && !atoi(getenv("SF_BUILD_LDS")));
  return env && (size_t)M_ * kBuildLdsPerSlot <= kBuildLdsMax / 2;
}

void DemEngine::compute_partner_tags()
{
  // the old list's history by partner tag, a copy per SIDE of every contact, in the ping-pong buffer the sub-steps
  // are not using: what migration packs and what the list build re-injects
  hist_buf_ = cur_ ^ 1;
  // Single domain, row path, parked candidates in LDS: the list build reads the old list IN PLACE (its words, the tags in
  // the order they index, the current history buffer) and writes the new words into the other word array -- no staging
  // pass (98 of a 1 M-grain loose bed's 774 us per rebuild, 15 of a 100 k bed's 210).  The staged rows remain what
  // migration packs (decomposed domains) and what the tiled / LDS-staged builds read.
  static const bool in_place_env = !(getenv("SF_HIST_IN_PLACE") && !atoi(getenv("SF_HIST_IN_PLACE")));
  hist_in_place_ = in_place_env && build_parks_in_lds() && !have_subdomain_ && roots_ && grid_.tile <= 1 && !opt_lds_ &&
                   have_list_ && nlocal_ > 0 && max_neigh_used_ > 0;
  if (hist_in_place_) {
    hist_buf_ = cur_;   // (k_build_neigh reads shear_[hist_buf_] and writes shear_[hist_buf_ ^ 1])
    return;
  }
  if (have_list_ && nlocal_ && max_neigh_used_ > 0)
    k_partner_tags<<<div_up(nlocal_, 256), 256, 0, stream_>>>(
        neigh_.as<int>(), numneigh_.as<int>(), tag_.as<int>(), ptag_.as<int>(), shear_[cur_].as<double>(),
        shear_[hist_buf_].as<double>(), nlocal_, cap_, max_neigh_used_, roots_ ? 1 : 0);
}

void DemEngine::rebuild_begin()
{
  if (!nlocal_ && !nghost_) return;
  compute_grid();
  compute_partner_tags();
  nghost_ = 0;
  next_ghost_ = 0;
  nsend_[0] = nsend_[1] = 0;
  recv_count_[0] = recv_count_[1] = 0;
  tx_ready_ = tx_written_ = false;
}

// Re-order (and possibly shrink to n_new) every per-atom array of the owned atoms: dst[i] = src[perm[i]].
// The old-list rows (partner tags, slot counts, shear) travel with their atom through the B-side buffers.
void DemEngine::permute_locals(const int* perm, int n_new, bool rows, const RankJob* rank)
{
  order_version_++;
  hist_indirect_ = false;
  if (n_new <= 0) return;
  const int nb = div_up(n_new, 256);
  // gather into the scratch array of the same shape, then swap the two allocations (no copy back): everything
  // beyond the owned atoms -- ghosts -- is re-created by the rebuild that follows
  auto g4 = [&](DevArray& a) {
    k_gather4<<<nb, 256, 0, stream_>>>(tmp4_.as<double4>(), a.as<double4>(), perm, n_new);
    std::swap(a.ptr, tmp4_.ptr);
  };
  // ONE gather kernel for the records and every per-atom fix array (a launch per array -- up to 15 of them at ~5 us of
  // launch floor each -- was a tenth of a rebuild): each array is gathered into a scratch array of its own shape and
  // the two allocations are swapped (device pointers handed out are valid until the next rebuild, sedifoam_amd.h)
  PermuteJobs J;
  memset(&J, 0, sizeof(J));
  DevArray* rec[3] = {&xr_[cur_], &vm_[cur_], &om_[cur_]};
  DevArray* rec_alt[3] = {&tmp4_, &tmp4b_, &tmp4c_};
  for (int k = 0; k < 3; k++) {
    J.s4[k] = rec[k]->as<double4>();
    J.d4[k] = rec_alt[k]->as<double4>();
  }
  DevArray* ints[4] = {&tag_, &type_, &mask_, &foamCpuId_};
  DevArray* ints_alt[4] = {&tmpi_, &tmpi_b_, &tmpi_c_, &tmpi_d_};
  for (int k = 0; k < 4; k++) {
    J.si[k] = ints[k]->as<int>();
    J.di[k] = ints_alt[k]->as<int>();
  }
  DevArray* rows_a[PermuteJobs::kRowArrays];
  DevArray* rows_alt[PermuteJobs::kRowArrays];
  auto add_rows = [&](DevArray& a, DevArray& alt, int nrows) {
    rows_a[J.nd] = &a;
    rows_alt[J.nd] = &alt;
    J.sd[J.nd] = a.as<double>();
    J.dd[J.nd] = alt.as<double>();
    J.rd[J.nd] = nrows;
    J.nd++;
  };
  add_rows(fdrag_, fdrag_alt_, 3);
  if (carrier_rho_ != 0.0) {
    add_rows(DuDt_, DuDt_alt_, 3);
    add_rows(vOld_, vOld_alt_, 3);
  }
  if (nextra_) add_rows(extra_, extra_alt_, nextra_);
  if (nwalls_) {
    add_rows(wshear_, tmpd_, 3 * nwalls_);
    J.sb = wtouch_.as<unsigned char>();
    J.db = wtouch_alt_.as<unsigned char>();
  }
  if (rank)   // (the permutation is not there yet: this launch ranks the atoms inside their cells and writes it)
    k_rank_permute<<<nb, 256, 0, stream_>>>(J, rank->keys, n_new, rank->first, rank->arrival, const_cast<int*>(perm), cap_);
  else
    k_permute_all<<<nb, 256, 0, stream_>>>(J, perm, n_new, cap_);
  for (int k = 0; k < 3; k++) std::swap(rec[k]->ptr, rec_alt[k]->ptr);
  for (int k = 0; k < 4; k++) std::swap(ints[k]->ptr, ints_alt[k]->ptr);
  for (int k = 0; k < J.nd; k++) std::swap(rows_a[k]->ptr, rows_alt[k]->ptr);
  if (nwalls_) std::swap(wtouch_.ptr, wtouch_alt_.ptr);
  // force / torque are stored by the LAST sub-step of a run only (and read by the first half-kick of the next run): a
  // rebuild inside a run -- always followed by at least one more sub-step, the last one storing them for every atom in
  // the new order -- need not carry them along
  if (!in_run_) {
    g4(force_);
    g4(torque_);
  }
  if (have_list_ && max_neigh_used_ > 0 && !rows) {
    // (the permutation itself becomes the index of the old rows: keep the array instead of copying it)
    if (perm == perm_alt_.as<int>()) std::swap(hist_perm_.ptr, perm_alt_.ptr);
    else SF_HIP(hipMemcpyAsync(hist_perm_.ptr, perm, sizeof(int) * n_new, hipMemcpyDeviceToDevice, stream_));
    hist_indirect_ = true;
  } else if (have_list_ && max_neigh_used_ > 0) {
    k_gather_rows<int><<<nb, 256, 0, stream_>>>(numneigh_old_.as<int>(), numneigh_.as<int>(), perm, n_new, 1, cap_);
    k_gather_rows<int><<<nb, 256, 0, stream_>>>(neigh_old_.as<int>(), ptag_.as<int>(), perm, n_new,
                                                max_neigh_used_, cap_);
    k_gather_rows<double><<<nb, 256, 0, stream_>>>(shear_[hist_buf_ ^ 1].as<double>(), shear_[hist_buf_].as<double>(),
                                                   perm, n_new, 3 * max_neigh_used_, cap_);
    std::swap(numneigh_, numneigh_old_);
    std::swap(ptag_, neigh_old_);
    hist_buf_ ^= 1;
  }
}

void DemEngine::rebuild_sort()
{
  if (migrate_pending_) migrate_compact();
  if (!nlocal_) return;
  PbcParams pb;
  for (int k = 0; k < 3; k++) {
    pb.lo[k] = boxlo_[k];
    pb.hi[k] = boxhi_[k];
    // x is wrapped here only when this GPU owns the whole periodic length; with several slabs the
    // wrap is applied by the migration shift
    pb.wrap[k] = periodic_[k] && !ext_[k];
  }
  const int nb = div_up(nlocal_, 256);
  // Plain keys (no tiles, no LDS staging): counting sort.  cell_start_ = [0] histogram of the owned atoms | [1] first
  // sorted position of every cell (owned) | [2] histogram of the ghosts | [3] first position in the ghost order.  The
  // histograms are counted back down to zero by the kernels that use them: cleared only when the table is new.
  row_tables_ = grid_.tile <= 1 && !opt_lds_;
  int* count = row_tables_ ? cell_start_ : nullptr;
  if (row_tables_ && !hist_clean_) {
    SF_HIP(hipMemsetAsync(cell_start_, 0, sizeof(int) * cell_alloc_, stream_));
    SF_HIP(hipMemsetAsync(cell_start_ + 2 * cell_alloc_, 0, sizeof(int) * cell_alloc_, stream_));
  }
  hist_clean_ = false;   // (until the kernels below have run: an error in between leaves it dirty)
  k_pbc_keys<<<nb, 256, 0, stream_>>>(xr_[cur_].as<double4>(), nlocal_, pb, grid_, keys_.as<unsigned>(),
                                      perm_.as<int>(), d_flags_, count);
  build_flags_clean_ = true;   // (F_NEIGH_OVER, F_MAXNEIGH zeroed by that kernel: the first list build needs no reset launch)
  if (row_tables_) {
    const int ne = grid_.nbins + 1;
    int* first = cell_start_ + cell_alloc_;
    exclusive_scan_i32(sort_tmp_, sort_tmp_bytes_, count, first, ne, stream_);
    k_key_place<<<nb, 256, 0, stream_>>>(keys_.as<unsigned>(), nlocal_, count, first, perm_.as<int>());
    hist_clean_ = true;
    // (rank and permutation in one launch: a launch less in a chain that small beds find bound by its launches -- 100 k loose
    // bed +0.9 to +2.4 % whole run; the scattered stores cost the 1 M bed nothing, +0.1 to +1 %: profiles/r06_rank_permute_ab.txt)
    const char* rp_env = getenv("SF_RANK_PERMUTE");
    const bool fused_rank = rp_env ? atoi(rp_env) != 0 : true;
    if (fused_rank) {
      const RankJob rj{keys_.as<unsigned>(), first, perm_.as<int>()};
      permute_locals(perm_alt_.as<int>(), nlocal_, /*rows=*/false, &rj);
    } else {
      k_key_rank<<<nb, 256, 0, stream_>>>(keys_.as<unsigned>(), nlocal_, first, perm_.as<int>(), tag_.as<int>(),
                                          perm_alt_.as<int>(), 0);
      permute_locals(perm_alt_.as<int>(), nlocal_, /*rows=*/false);
    }
    mark_frozen();   // migrated / created atoms arrive without the mark; cheap, rebuild-time only
  } else {
    int bits = 1;
    while ((1 << bits) < grid_.nbins) bits++;
    sort_pairs_u32(sort_tmp_, sort_tmp_bytes_, keys_.as<unsigned>(), keys_alt_.as<unsigned>(), perm_.as<int>(),
                   perm_alt_.as<int>(), nlocal_, bits, stream_);
    permute_locals(perm_alt_.as<int>(), nlocal_);
    mark_frozen();
    // sorted bin keys -> cell ranges of owned atoms
    hist_clean_ = false;
    SF_HIP(hipMemsetAsync(cell_start_, 0, sizeof(int) * 4 * cell_alloc_, stream_));
    k_cell_bounds<unsigned><<<nb, 256, 0, stream_>>>(keys_alt_.as<unsigned>(), nlocal_, 0, cell_start_,
                                                     cell_start_ + 1, 4);
  }
  // owned range of every tile (bin keys are tile-major: key >> log2(T^3) is the tile id)
  const int T = grid_.tile;
  ntiles_ = grid_.nt[0] * grid_.nt[1] * grid_.nt[2];
  if (T >= 2 && !(T & (T - 1))) {
    if ((size_t)ntiles_ + 1 > tile_alloc_) {
      if (tile_tab_) SF_HIP(hipFree(tile_tab_));
      tile_alloc_ = (size_t)ntiles_ + ntiles_ / 8 + 16;
      SF_HIP(hipMalloc(&tile_tab_, sizeof(int) * 4 * tile_alloc_));
    }
    int shift = 0;
    while ((1 << shift) < T * T * T) shift++;
    SF_HIP(hipMemsetAsync(tile_tab_, 0, sizeof(int) * 4 * tile_alloc_, stream_));
    k_cell_bounds<unsigned><<<nb, 256, 0, stream_>>>(keys_alt_.as<unsigned>(), nlocal_, shift, tile_tab_,
                                                     tile_tab_ + tile_alloc_, 1);
  }
}

void DemEngine::make_periodic_ghosts()
{
  // external ghosts (other GPUs) were appended by border_unpack: slots [nlocal, nlocal+next_ghost_)
  nghost_ = next_ghost_;
  nimages_ = -1;
  if (ghost_free_) return;   // (the list build walks around the box itself: compute_grid)
  const double cut = cutneighmax();
  int dims[3], nd = 0;
  for (int dim = 0; dim < 3; dim++)   // images across an external face come from the neighbour GPUs (or the driver's self loop)
    if (periodic_[dim] && !ext_[dim]) dims[nd++] = dim;
  if (!nd || cap_ == 0 || nlocal_ + next_ghost_ == 0) return;
  GhostPtrs G{xr_[cur_].as<double4>(), vm_[cur_].as<double4>(), om_[cur_].as<double4>(), tag_.as<int>(),
              type_.as<int>(), mask_.as<int>(), gsrc_.as<int>(), gshift_.as<double>()};
  for (int attempt = 0; attempt < 4; attempt++) {
    // F_GHOST_COUNT counts ghosts (external ones included); list slot = ghost slot.  Every dimension looks at the owned
    // atoms + the ghosts made before it; those counts stay on the device (k_ghost_select), the host reads the total
    set_flags3(F_GHOST_COUNT, next_ghost_, F_GHOST_OVER, 0, F_GHOST_BEFORE + dims[0], next_ghost_);
    for (int q = 0; q < nd; q++) {
      const int dim = dims[q];
      k_ghost_select<<<div_up((long long)cap_, 1024), 1024, 0, stream_>>>(
          G.xr, d_flags_, F_GHOST_BEFORE + dim, nlocal_, dim, boxlo_[dim], boxhi_[dim], cut, perm_.as<int>(),
          d_flags_ + F_GHOST_COUNT, cap_);
      k_ghost_create<<<std::max(1, div_up((long long)(cap_ - nlocal_), 256)), 256, 0, stream_>>>(
          G, perm_.as<int>(), nlocal_, dim, boxhi_[dim] - boxlo_[dim], cap_, d_flags_, F_GHOST_BEFORE + dim,
          q + 1 < nd ? F_GHOST_BEFORE + dims[q + 1] : -1);
    }
    // (the count can stay on the device until the flags are read behind the list build: bin_and_build; nghost_ < 0 =
    // "on the device".  SF_GHOST_DEFER=0: read it here, as before round 5)
    static const bool defer = !(getenv("SF_GHOST_DEFER") && !atoi(getenv("SF_GHOST_DEFER")));
    if (defer && attempt == 0 && row_tables_ && !ghost_sync_) {
      nghost_ = -1;
      return;
    }
    read_flags();
    if (!h_flags_[F_GHOST_OVER] && (size_t)nlocal_ + h_flags_[F_GHOST_COUNT] <= cap_) {
      nghost_ = h_flags_[F_GHOST_COUNT];
      return;
    }
    // (a count cut short by the capacity is a lower bound of what is needed)
    ensure_capacity((size_t)nlocal_ + (size_t)h_flags_[F_GHOST_COUNT] * 2 + 1024);
    G = GhostPtrs{xr_[cur_].as<double4>(), vm_[cur_].as<double4>(), om_[cur_].as<double4>(), tag_.as<int>(),
                  type_.as<int>(), mask_.as<int>(), gsrc_.as<int>(), gshift_.as<double>()};
  }
  fail("ghost creation: capacity could not be grown");
}

// Tables of the LDS-staged kernel: per tile the list of atoms (owned and ghost) in its (T+2)^3 extended bins and
// the offset of every extended bin in that list.  Falls back to the gathering kernel when tiles are off, T is
// not a power of two, or a tile would not fit the 160 KiB of LDS.
void DemEngine::build_stage_tables()
{
  lds_active_ = false;
  const int T = grid_.tile;
  if (!opt_lds_ || T < 2 || (T & (T - 1)) || !nlocal_ || grid_.stencil != 1) return;
  if (!tile_tab_) return;
  const int E = T + 2, EB = E * E * E;
  if ((size_t)ntiles_ * EB > eoff_alloc_) {
    if (eoff_) SF_HIP(hipFree(eoff_));
    eoff_alloc_ = (size_t)ntiles_ * EB + 1024;
    SF_HIP(hipMalloc(&eoff_, sizeof(int) * eoff_alloc_));
  }
  int* tcount = tile_tab_ + 2 * tile_alloc_;
  int* tstart = tile_tab_ + 3 * tile_alloc_;
  int* cellLS = cell_start_;        // interleaved per cell: {owned start, owned end, ghost start, ghost end}
  int* cellLE = cell_start_ + 1;
  int* cellGS = cell_start_ + 2;
  int* cellGE = cell_start_ + 3;
  reset_flag(F_STAGE_MAX, 0);
  k_tile_stage_count<<<div_up(ntiles_, 128), 128, 0, stream_>>>(grid_, cellLS, cellLE, cellGS, cellGE, ntiles_,
                                                               tcount, d_flags_);
  exclusive_scan_i32(sort_tmp_, sort_tmp_bytes_, tcount, tstart, ntiles_ + 1, stream_);
  int total = 0;
  SF_HIP(hipMemcpyAsync(&total, tstart + ntiles_, sizeof(int), hipMemcpyDeviceToHost, stream_));
  read_flags();
  stage_cap_ = h_flags_[F_STAGE_MAX];
  if ((size_t)stage_cap_ * 88 > 150 * 1024 || stage_cap_ > 65535) return;   // does not fit: gather kernel
  if ((size_t)total > stage_alloc_) {
    if (stage_idx_) SF_HIP(hipFree(stage_idx_));
    stage_alloc_ = (size_t)total + total / 8 + 1024;
    SF_HIP(hipMalloc(&stage_idx_, sizeof(int) * stage_alloc_));
  }
  k_tile_stage_fill<<<div_up(ntiles_, 128), 128, 0, stream_>>>(grid_, cellLS, cellLE, cellGS, cellGE,
                                                              perm_alt_.as<int>(), ntiles_, tstart, eoff_,
                                                              stage_idx_);
  lds_active_ = true;
}

void DemEngine::bin_and_build()
{
  if (!nlocal_) {
    have_list_ = true;
    return;
  }
  // history copies of this list: from the coalescing the previous list measured (h_flags_ was refreshed by the
  // synchronisation that led here), with hysteresis; the first list is built single-copy
  if (hist_mode_env_ == 1) hist_single_ = true;
  else if (hist_mode_env_ == 2 || !roots_) hist_single_ = false;
  else if (have_list_ && h_flags_[F_PART_SLOTS] > 0) {
    const double frac = (double)h_flags_[F_PART_COAL] / (double)h_flags_[F_PART_SLOTS];
    if (hist_single_ && frac < 0.45) hist_single_ = false;
    else if (!hist_single_ && frac > 0.60) hist_single_ = true;
    static const bool dbg = getenv("SF_DEBUG_HIST") != nullptr;
    if (dbg) fprintf(stderr, "[sedifoam_amd] partner-side coalescing %.3f -> %s history copy\n", frac, hist_single_ ? "one" : "two");
  }
  // slot order of this list: touching neighbours first when the previous list touched fewer than about half of what
  // it listed (k_build_neigh, touch_first; hysteresis 0.40 / 0.50; SF_TOUCH_FIRST=0 / 1 pins it)
  if (touch_first_env_ >= 0) touch_first_ = touch_first_env_ != 0;
  else if (have_list_ && h_flags_[F_LIST_SLOTS] > 0) {
    const double frac = (double)h_flags_[F_LIST_TOUCH] / (double)h_flags_[F_LIST_SLOTS];
    const bool before = touch_first_;
    if (!touch_first_ && frac < 0.40) touch_first_ = true;
    else if (touch_first_ && frac > 0.50) touch_first_ = false;
    static const bool dbg = getenv("SF_DEBUG_HIST") != nullptr;
    if (dbg && before != touch_first_)
      fprintf(stderr, "[sedifoam_amd] %.3f of the listed neighbours touch -> touching neighbours %s\n", frac,
              touch_first_ ? "first in their rows" : "in candidate order");
  }
  int* cellLS = cell_start_;        // interleaved per cell: {owned start, owned end, ghost start, ghost end}
  int* cellLE = cell_start_ + 1;
  int* cellGS = cell_start_ + 2;
  int* cellGE = cell_start_ + 3;
  // (nghost_ < 0: the ghost count is still on the device -- make_periodic_ghosts did not wait for it; the kernels below read
  // it themselves and are launched for the most ghosts the capacity could hold)
  const bool ghosts_pending = nghost_ < 0;
  if (ghosts_pending && !row_tables_) fail("bin_and_build: deferred ghost count without row tables");
  if (ghosts_pending) {
    const int ne = grid_.nbins + 1;
    const int ng = std::max(1, div_up((long long)(cap_ - (size_t)nlocal_), 256));
    int* count = cell_start_ + 2 * cell_alloc_;
    int* first = cell_start_ + 3 * cell_alloc_;
    hist_clean_ = false;
    k_ghost_cells_dev<<<ng, 256, 0, stream_>>>(xr_[cur_].as<double4>(), nlocal_, cap_, grid_, keys_.as<unsigned>(), count,
                                               d_flags_);
    exclusive_scan_i32(sort_tmp_, sort_tmp_bytes_, count, first, ne, stream_);
    k_key_place_dev<<<ng, 256, 0, stream_>>>(keys_.as<unsigned>(), d_flags_, nlocal_, cap_, count, first, perm_.as<int>());
    k_key_rank_dev<<<ng, 256, 0, stream_>>>(keys_.as<unsigned>(), d_flags_, nlocal_, cap_, first, perm_.as<int>(),
                                            tag_.as<int>(), perm_alt_.as<int>());
    hist_clean_ = true;
  } else if (nghost_ && row_tables_) {
    // ghosts in (cell, tag) order by the same counting sort as the owned atoms; its scan IS the table of first
    // ghost-order positions per cell that the list build reads
    const int ne = grid_.nbins + 1, ng = div_up(nghost_, 256);
    int* count = cell_start_ + 2 * cell_alloc_;
    int* first = cell_start_ + 3 * cell_alloc_;
    hist_clean_ = false;
    k_ghost_cells<<<ng, 256, 0, stream_>>>(xr_[cur_].as<double4>(), nlocal_, nghost_, grid_, keys_.as<unsigned>(), count,
                                           d_flags_);
    exclusive_scan_i32(sort_tmp_, sort_tmp_bytes_, count, first, ne, stream_);
    k_key_place<<<ng, 256, 0, stream_>>>(keys_.as<unsigned>(), nghost_, count, first, perm_.as<int>());
    k_key_rank<<<ng, 256, 0, stream_>>>(keys_.as<unsigned>(), nghost_, first, perm_.as<int>(), tag_.as<int>(),
                                        perm_alt_.as<int>(), nlocal_);
    hist_clean_ = true;
  } else if (nghost_) {
    k_ghost_keys<<<div_up(nghost_, 256), 256, 0, stream_>>>(xr_[cur_].as<double4>(), tag_.as<int>(), nlocal_,
                                                            nghost_, grid_, keys64_.as<unsigned long long>(),
                                                            perm_.as<int>(), d_flags_);
    sort_pairs_u64(sort_tmp_, sort_tmp_bytes_, keys64_.as<unsigned long long>(),
                   keys64_alt_.as<unsigned long long>(), perm_.as<int>(), perm_alt_.as<int>(), nghost_, 64,
                   stream_);
    k_cell_bounds<unsigned long long><<<div_up(nghost_, 256), 256, 0, stream_>>>(
        keys64_alt_.as<unsigned long long>(), nghost_, 32, cellGS, cellGE, 4);
  }
  build_stage_tables();
  for (int attempt = 0; attempt < 4; attempt++) {
    if (build_flags_clean_) h_flags_[F_NEIGH_OVER] = h_flags_[F_MAXNEIGH] = 0;
    else set_flags3(F_NEIGH_OVER, 0, F_MAXNEIGH, 0, F_PARK_OVER, 0);
    build_flags_clean_ = false;
    BuildParams B;
    B.nlocal = nlocal_;
    B.M = M_;
    B.Mold = max_neigh_used_;
    B.cap = cap_;
    B.skin_gran = gran_.style ? lskin() : -1.0;
    B.cut_lub = lub_.enabled ? lub_.cut_global + lskin() : 0.0;
    // fix cohesive walks a regular half list (fix_cohesive.cpp:75-77), whose criterion is the pair cutoff + skin for
    // every pair, not ri + rj + skin: pairs enter that list exactly when LAMMPS would list them
    if (cohe_.enabled) B.cut_lub = std::max(B.cut_lub, cutneighmax());
    B.g = grid_;
    B.eoff = lds_active_ ? eoff_ : nullptr;
    B.nloc = nloc_.as<unsigned short>();
    B.old_index = hist_indirect_ ? hist_perm_.as<int>() : nullptr;
    B.two_copies = hist_single_ ? 0 : 1;
    B.touch_first = touch_first_ ? 1 : 0;
    B.lb_own = row_tables_ ? cell_start_ + cell_alloc_ : nullptr;
    B.lb_ghost = (row_tables_ && nghost_) ? cell_start_ + 3 * cell_alloc_ : nullptr;   // (nghost_ < 0: on the device)
    // parked candidates in LDS (row path): as long as the rows fit; a list read in place needs them there (the scratch
    // rows of the other form ARE the word array the new list goes into)
    const bool lc = row_tables_ && (hist_in_place_ || build_parks_in_lds());
    // (parking rows: the longest row of the previous list + 8 -- a fifth of the LDS of a compute unit per block otherwise,
    // at four waves per SIMD; an atom with more candidates than that reports F_PARK_OVER and the list is built again)
    if (park_rows_ <= 0 || park_rows_ > M_ || attempt > 0) park_rows_ = M_;
    B.P = park_rows_;
    const size_t lds_bytes = lc ? (size_t)park_rows_ * kBuildLdsPerSlot : 0;
    if (lds_bytes > kBuildLdsMax)
      fail("neighbor list rows of more than %d slots do not fit the list build's LDS rows (SF_HIST_IN_PLACE=0 SF_BUILD_LDS=0 "
           "builds such lists through memory)", (int)(kBuildLdsMax / kBuildLdsPerSlot));
    static bool lds_attr_set = false;
    if (lc && !lds_attr_set) {
      SF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_build_neigh<true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBuildLdsMax));
      SF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_build_neigh_quad<4>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBuildLdsMax));
      SF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_build_neigh_quad<2>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBuildLdsMax));
      SF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_build_neigh_quad<8>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBuildLdsMax));
      lds_attr_set = true;
    }
    B.old_words = hist_in_place_ ? neigh_.as<int>() : nullptr;
    B.old_tag = hist_in_place_ ? tmpi_.as<int>() : nullptr;   // (permute_locals swapped the old tag array out into tmpi_)
    int* const new_words = hist_in_place_ ? neigh_old_.as<int>() : neigh_.as<int>();
    B.roots = roots_ ? 1 : 0;
    B.gsrc = gsrc_.as<int>();
    B.gshift = gshift_.as<double>();
    for (int k = 0; k < 3; k++) {
      B.inv_prd[k] = 1.0 / (boxhi_[k] - boxlo_[k]);
      B.prd[k] = boxhi_[k] - boxlo_[k];
    }
    if (roots_ && cap_ > (size_t)kIdxMask) fail("more than %d atom slots per GPU: not addressable by the neighbour word", kIdxMask);
    // four lanes per atom on four consecutive records (k_build_neigh_quad): single domain without a ghost pass
    const char* quad_env = getenv("SF_BUILD_QUAD");   // (read per build: the tests switch it inside a process)
    // (rows of full-cutoff cells -- loose beds, ~8 records -- keep four lanes busy: 405 -> 297 us on the loose 1 M bed, 60 -> 39 us
    // at 100 k grains)
    // (the 25 rows of 2-3 records of a packed bed's half-cutoff cells: two lanes, 399 -> 347-355 us at 1 M; four lanes 446)
    const int lq = quad_env ? atoi(quad_env) : (grid_.stencil == 1 ? 4 : 2);
    const bool quad = lc && !B.lb_ghost && !grid_.xslow && (lq == 2 || lq == 4 || lq == 8);
    if (quad && lq == 2)
      k_build_neigh_quad<2><<<div_up(nlocal_, 64), 128, lds_bytes / 2, stream_>>>(
          B, xr_[cur_].as<double4>(), tag_.as<int>(), have_list_ ? numneigh_.as<int>() : nullptr, ptag_.as<int>(),
          shear_[hist_buf_].as<double>(), new_words, numneigh_old_.as<int>(), shear_[hist_buf_ ^ 1].as<double>(), d_flags_,
          xhold_.as<double>());
    else if (quad && lq == 8)
      k_build_neigh_quad<8><<<div_up(nlocal_, 16), 128, lds_bytes / 8, stream_>>>(
          B, xr_[cur_].as<double4>(), tag_.as<int>(), have_list_ ? numneigh_.as<int>() : nullptr, ptag_.as<int>(),
          shear_[hist_buf_].as<double>(), new_words, numneigh_old_.as<int>(), shear_[hist_buf_ ^ 1].as<double>(), d_flags_,
          xhold_.as<double>());
    else if (quad)
      k_build_neigh_quad<4><<<div_up(nlocal_, 32), 128, lds_bytes / 4, stream_>>>(
          B, xr_[cur_].as<double4>(), tag_.as<int>(), have_list_ ? numneigh_.as<int>() : nullptr, ptag_.as<int>(),
          shear_[hist_buf_].as<double>(), new_words, numneigh_old_.as<int>(), shear_[hist_buf_ ^ 1].as<double>(), d_flags_,
          xhold_.as<double>());
    else if (lc)
      k_build_neigh<true><<<div_up(nlocal_, 128), 128, lds_bytes, stream_>>>(
          B, xr_[cur_].as<double4>(), tag_.as<int>(), cellLS, cellLE, cellGS, cellGE, perm_alt_.as<int>(),
          have_list_ ? numneigh_.as<int>() : nullptr, ptag_.as<int>(), shear_[hist_buf_].as<double>(), new_words,
          numneigh_old_.as<int>(), shear_[hist_buf_ ^ 1].as<double>(), d_flags_, nullptr, xhold_.as<double>());
    else if (!row_tables_)
      k_build_neigh<false, false><<<div_up(nlocal_, 128), 128, 0, stream_>>>(
          B, xr_[cur_].as<double4>(), tag_.as<int>(), cellLS, cellLE, cellGS, cellGE, perm_alt_.as<int>(),
          have_list_ ? numneigh_.as<int>() : nullptr, ptag_.as<int>(), shear_[hist_buf_].as<double>(), new_words,
          numneigh_old_.as<int>(), shear_[hist_buf_ ^ 1].as<double>(), d_flags_, neigh_old_.as<int>(), xhold_.as<double>());
    else
      k_build_neigh<false><<<div_up(nlocal_, 128), 128, 0, stream_>>>(
          B, xr_[cur_].as<double4>(), tag_.as<int>(), cellLS, cellLE, cellGS, cellGE, perm_alt_.as<int>(),
          have_list_ ? numneigh_.as<int>() : nullptr, ptag_.as<int>(), shear_[hist_buf_].as<double>(), new_words,
          numneigh_old_.as<int>(), shear_[hist_buf_ ^ 1].as<double>(), d_flags_, neigh_old_.as<int>(), xhold_.as<double>());
    // the host looks at the counts (overflow, widest row) while the partner-slot pass below is already running: it
    // needs nothing but the list, and a list that overflowed -- rare -- is built again and the pass repeated
    flags_copy_begin();
    // partner slots: where does the owner keep this pair?  (a partner whose owner does not list it back owns the pair)
    k_back_slots<<<div_up(nlocal_, 128), 128, 0, stream_>>>(new_words, numneigh_old_.as<int>(), nlocal_, cap_,
                                                            roots_ ? 1 : 0, d_flags_);
    trigger_rearmed_ = true;
    flags_copy_wait();
    if (ghosts_pending && nghost_ < 0) {
      // the ghost count arrives with these flags.  More ghosts than the capacity held: nothing above saw a ghost -- grow,
      // make the ghosts again (this time waiting for the count) and start over
      if (h_flags_[F_GHOST_OVER] || (size_t)nlocal_ + (size_t)h_flags_[F_GHOST_COUNT] > cap_) {
        ensure_capacity((size_t)nlocal_ + (size_t)h_flags_[F_GHOST_COUNT] * 2 + 1024);
        ghost_sync_ = true;
        make_periodic_ghosts();
        ghost_sync_ = false;
        bin_and_build();
        return;
      }
      nghost_ = h_flags_[F_GHOST_COUNT];
    }
    if (h_flags_[F_NEIGH_OVER] > M_) {
      // more neighbours than slots: widen the slot-major arrays and build again.  The old-history
      // arrays keep their first rows, so the re-injection still finds every partner.
      grow_neigh(h_flags_[F_NEIGH_OVER] + 4);
      continue;
    }
    if (lc && h_flags_[F_PARK_OVER] > park_rows_) continue;   // (again, with as many parking rows as list slots)
    break;
  }
  if (h_flags_[F_NEIGH_OVER] > M_) fail("neighbor list overflow (%d > %d slots)", h_flags_[F_NEIGH_OVER], M_);
  if (h_flags_[F_LOST] == 1) {
    reset_flag(F_LOST, 0);
    fail("Lost atoms: an atom left the (non-periodic) simulation box");  // thermo_modify lost error
  }
  std::swap(numneigh_, numneigh_old_);
  if (hist_in_place_) std::swap(neigh_.ptr, neigh_old_.ptr);   // (the new words went into the other array)
  hist_in_place_ = false;
  // the new list's history was built into shear_[hist_buf_ ^ 1]: that buffer is the one the next sub-step reads
  if ((hist_buf_ ^ 1) != cur_) std::swap(shear_[0].ptr, shear_[1].ptr);
  hist_indirect_ = false;   // the old rows are gone with the old list
  // (the statistics steer slow choices behind hysteresis -- history copies, slot order, v / omega prefetch, cache
  // policy.  A list is measured whenever the last measured fractions lie within 0.08 of a band a decision switches
  // on -- so the decisions are those of measuring EVERY list as long as a fraction does not jump a whole margin
  // between two lists, whatever the count or the chunking of the rebuilds -- and otherwise on the first lists and
  // every fourth one, which only refreshes numbers no decision is near; the counters keep their values in between)
  if (nbuilds_ < 4 || (nbuilds_ & 3) == 0 || list_stats_near_a_threshold()) measure_list();
  max_neigh_used_ = h_flags_[F_MAXNEIGH];
  // (SF_PARK_MARGIN: the tests make the rows too few on purpose -- the F_PARK_OVER path builds the list again)
  static const int park_margin = getenv("SF_PARK_MARGIN") ? atoi(getenv("SF_PARK_MARGIN")) : 8;
  park_rows_ = std::max(1, std::min(M_, max_neigh_used_ + park_margin));
  have_list_ = true;   // (xhold, the positions the skin/2 check refers to, was stored by k_build_neigh)
  nbuilds_++;
  if (xcd_auto_ && xcd_countdown_ == 0) xcd_countdown_ = 3;   // the third full launch on the new list is timed per XCD
}

// true when a decision taken from the list statistics could change with the next measurement: a fraction within
// 0.08 of the hysteresis band of the history copies (0.45 / 0.60), the slot order (0.40 / 0.50) or the v, omega
// prefetch (0.70 / 0.85)
bool DemEngine::list_stats_near_a_threshold() const
{
  const double m = 0.08;
  if (h_flags_[F_PART_SLOTS] > 0) {
    const double c = (double)h_flags_[F_PART_COAL] / (double)h_flags_[F_PART_SLOTS];
    if (c > 0.45 - m && c < 0.60 + m) return true;
  }
  if (h_flags_[F_LIST_SLOTS] > 0) {
    const double t = (double)h_flags_[F_LIST_TOUCH] / (double)h_flags_[F_LIST_SLOTS];
    if ((t > 0.40 - m && t < 0.50 + m) || (t > 0.70 - m && t < 0.85 + m)) return true;
  }
  return false;
}

// List statistics that pick the kernel variant and the history layout (read back with the flags at the next
// synchronisation): coalescing of the would-be partner-side gathers, listed and touching neighbours.
void DemEngine::measure_list()
{
  if (!nlocal_ || !roots_) return;
  static const bool dbg_lines = getenv("SF_DEBUG_LINES") != nullptr;
  if (dbg_lines) {
    unsigned long long* d = nullptr;
    unsigned long long h[6] = {0, 0, 0, 0, 0, 0};
    SF_HIP(hipMalloc(&d, sizeof h));
    SF_HIP(hipMemsetAsync(d, 0, sizeof h, stream_));
    const int nw = div_up(nlocal_, 64), st = nw > 4096 ? 16 : 1;
    k_gather_lines<<<div_up(nw, st), 64, 0, stream_>>>(neigh_.as<int>(), numneigh_.as<int>(), nlocal_, cap_, st, d);
    SF_HIP(hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, stream_));
    SF_HIP(hipStreamSynchronize(stream_));
    SF_HIP(hipFree(d));
    if (h[0])
      fprintf(stderr, "[sedifoam_amd] gather lines per wave-level slot (%llu sampled): %.1f active lanes; 128-B lines %.1f "
              "(x 2 halves = %.1f accesses per record gather); lane pairs sharing a record: %.1f + %.1f = %.1f; 64-B half "
              "lines %.1f\n", h[0], (double)h[1] / h[0], (double)h[2] / h[0], 2.0 * h[2] / h[0], (double)h[3] / h[0],
              (double)h[4] / h[0], (double)(h[3] + h[4]) / h[0], (double)h[5] / h[0]);
  }
  static_assert(F_PART_COAL == F_PART_SLOTS + 1 && F_LIST_SLOTS == F_PART_SLOTS + 2 && F_LIST_TOUCH == F_PART_SLOTS + 3,
                "adjacent counters");
  reset_flags(F_PART_SLOTS, 4, 0);
  // a sample of one block of 1024 atoms in eight above 64 k atoms (ratios and a mean are all that is used)
  const int nblk = div_up(nlocal_, 1024), stride = nlocal_ > 65536 ? 8 : 1;
  const int nsamp = div_up(nblk, stride);
  list_sampled_ = 0;
  for (int b = 0; b < nsamp; b++) list_sampled_ += std::min(1024, nlocal_ - b * stride * 1024);
  k_partner_coalescing<<<nsamp, 1024, 0, stream_>>>(neigh_.as<int>(), numneigh_.as<int>(), nlocal_, cap_,
                                                    d_flags_ + F_PART_SLOTS, stride);
}

void DemEngine::choose_kernel()
{
  static const bool dbg = getenv("SF_DEBUG_HIST") != nullptr;
  // Non-temporal policy of the row streams (sf_dem_kernels.h, NTP): what one sub-step touches against the 256 MB
  // memory-side cache.  Records in + out 192 B, history in + out 48 B per stored copy, list words, fix arrays.
  {
    const double khalf = (list_sampled_ > 0 && h_flags_[F_LIST_SLOTS] > 0)
                             ? 0.5 * (double)h_flags_[F_LIST_SLOTS] / (double)list_sampled_ : 6.0;
    const double copies = hist_single_ ? 1.0 : 2.0;
    const double touched = (double)nlocal_ * (252.0 + (8.0 + 48.0 * copies) * khalf);
    const double mall = 256.0 * 1024.0 * 1024.0;
    const int before = nt_policy_;
    if (nt_policy_env_ >= 0) nt_policy_ = nt_policy_env_;
    else if (touched < mall) nt_policy_ = 0;
    else if (touched < 1.4 * mall) nt_policy_ = 3;
    else nt_policy_ = hist_single_ ? 1 : 2;
    if (dbg && before != nt_policy_)
      fprintf(stderr, "[sedifoam_amd] a sub-step touches %.0f MB -> non-temporal policy %d\n", touched / 1048576.0,
              nt_policy_);
  }
  // Sort cells (compute_grid): half-cutoff cells give a packed bed the atom order its gathers coalesce on (cells of the
  // full cutoff: 297 against 185 us per sub-step at 1 M grains); a loose bed has no such order to lose, its sub-step kernel
  // is 1.3 % faster on full-cutoff cells and its many rebuilds 5 % (nine cell rows of ~8 candidates instead of 25 of 2-3):
  // +2.3 % on the loose 1 M bed.  Decided on the fraction of listed neighbours that touch, from the second list on (the
  // first list of a run carries no touch bits), same band as the v, omega prefetch; takes effect at the next rebuild.
  if (h_flags_[F_LIST_SLOTS] > 0 && nbuilds_ >= 2) {
    const double f = (double)h_flags_[F_LIST_TOUCH] / (double)h_flags_[F_LIST_SLOTS];
    if (f < 0.70) sort_sub_ = 1;
    else if (f > 0.85) sort_sub_ = 2;
  }
  if (touch_prefetch_env_ >= 0) {
    touch_prefetch_ = touch_prefetch_env_ != 0;
    return;
  }
  if (h_flags_[F_LIST_SLOTS] <= 0) return;
  const double frac = (double)h_flags_[F_LIST_TOUCH] / (double)h_flags_[F_LIST_SLOTS];
  // hysteresis: prefetch by touch bit below 0.70, always above 0.85
  const bool before = touch_prefetch_;
  if (touch_prefetch_ && frac > 0.85) touch_prefetch_ = false;
  else if (!touch_prefetch_ && frac < 0.70) touch_prefetch_ = true;
  if (dbg && before != touch_prefetch_)
    fprintf(stderr, "[sedifoam_amd] %.3f of the listed neighbours touch -> v, omega %s\n", frac,
            touch_prefetch_ ? "prefetched by touch bit" : "always prefetched");
}

void DemEngine::rebuild_finish()
{
  make_periodic_ghosts();
  trigger_rearmed_ = false;
  bin_and_build();
  if (overlap_) mark_boundary();
  if (trigger_rearmed_) h_flags_[F_TRIGGER] = INT_MAX;   // (by k_back_slots, the last kernel of the list build)
  else reset_flag(F_TRIGGER, INT_MAX);
}

void DemEngine::rebuild()
{
  Range r("neighbor rebuild");   // (inside the reference's "lammps" bucket)
  if (!in_run_) predict_.external(nsteps_);
  rebuild_begin();
  rebuild_sort();
  rebuild_finish();
}

// ------------------------------------------------------------------------------------------------
// stepping
// ------------------------------------------------------------------------------------------------
double DemEngine::local_particle_volume()
{
  // sum over the owned atoms of 4/3 pi r^3 (pair_lubricate_poly.cpp:540-542), in index order on the host: setup-time only
  std::vector<double4> hx(nlocal_);
  if (nlocal_)
    SF_HIP(hipMemcpyAsync(hx.data(), xr_[cur_].ptr, sizeof(double4) * nlocal_, hipMemcpyDeviceToHost, stream_));
  sync();
  double volP = 0.0;
  for (int i = 0; i < nlocal_; i++) volP += (4.0 / 3.0) * kPi * std::pow(hx[i].w, 3.0);
  return volP;
}

void DemEngine::setup()
{
  if (!have_nve_ && nlocal_) { /* allowed: static atoms */ }
  if (!have_subdomain_) {
    have_list_ = false;
    rebuild();
  } else if (!have_list_)
    fail("sf_dem_setup on a decomposed domain: run the rebuild protocol (sf_dem_rebuild_*) first");
  // (after the first sort: the sum then runs over the atoms in (cell, tag) order, the same bits for any input order)
  if (lub_.enabled) {
    // PairLubricatePoly::init_style pair_lubricate_poly.cpp:514-559: volume fraction constants.  volP is the volume of
    // ALL particles (MPI_Allreduce, :540-543): on a decomposed domain the driver sums local_particle_volume() over the
    // ranks and hands the total back through set_global_particle_volume() before setup
    if (have_subdomain_ && nranks_ > 1 && !(global_volP_ >= 0.0))
      fail("pair lubricate/poly on a decomposed domain: sum sf_dem_local_particle_volume over the ranks and pass it "
           "to sf_dem_set_global_particle_volume before sf_dem_setup (pair_lubricate_poly.cpp:540-543)");
    const double volP = global_volP_ >= 0.0 ? global_volP_ : local_particle_volume();
    const double vol_T = (boxhi_[0] - boxlo_[0]) * (boxhi_[1] - boxlo_[1]) * (boxhi_[2] - boxlo_[2]);
    double vol_f = volP / vol_T;
    if (!lub_.flagVF) vol_f = 0;
    const double mu = lub_.mu;
    if (lub_.flaglog == 0) {
      lub_.R0 = 6 * kPi * mu * (1.0 + 2.16 * vol_f);
      lub_.RT0 = 8 * kPi * mu;
      lub_.RS0 = 20.0 / 3.0 * kPi * mu * (1.0 + 3.33 * vol_f + 2.80 * vol_f * vol_f);
    } else {
      lub_.R0 = 6 * kPi * mu * (1.0 + 2.725 * vol_f - 6.583 * vol_f * vol_f);
      lub_.RT0 = 8 * kPi * mu * (1.0 + 0.749 * vol_f - 2.469 * vol_f * vol_f);
      lub_.RS0 = 20.0 / 3.0 * kPi * mu * (1.0 + 3.64 * vol_f - 6.95 * vol_f * vol_f);
    }
  }
  // (multi-rank: the driver has already run rebuild_begin / migrate / sort / borders / finish)
  reset_flag(F_TRIGGER, INT_MAX);
  wall_time_origin_ = nsteps_;   // FixWallGranFix::init, fix_wall_granFix.cpp:181
  launch_substep(cur_, 2, 0);
  launch_ghost_forward(cur_ ^ 1, 0);
  cur_ ^= 1;
  measure_list();   // (the setup evaluation has set the touch bits of the first list)
  read_flags();
  choose_kernel();
  setup_done_ = true;
}

void DemEngine::run_begin()
{
  run_base_step_ = nsteps_;
  tx_written_ = false;
  choose_kernel();
  if (overlap_) overlap_begin();
  reset_flag(F_TRIGGER, INT_MAX);
  launch_initial_integrate();
  launch_ghost_forward(cur_, 0);
}

void DemEngine::substep(bool last)
{
  run_base_step_ = nsteps_;
  launch_substep(cur_, last ? 1 : 0, 0);
  launch_ghost_forward(cur_ ^ 1, 0);
  cur_ ^= 1;
  nsteps_++;
}

void DemEngine::substep_k(bool last, int kstep)
{
  launch_substep(cur_, last ? 1 : 0, kstep);
  // decomposed domain: the driver refreshes every ghost (received ones, then their local images) right after
  if (!have_subdomain_) launch_ghost_forward(cur_ ^ 1, kstep);
  cur_ ^= 1;
}

int DemEngine::batch_end(int first_k, int launched)
{
  read_flags();
  const int trig = h_flags_[F_TRIGGER];
  if (profiling_) {
    harvest_profile(trig);   // (the launches after a trigger were early exits: not kernel time)
    prof_used_ = 0;
  }
  const int executed = trig == INT_MAX ? launched : std::max(0, trig + 1 - first_k);
  // every substep_k flipped the buffer parity; the early-exited ones must not count
  if ((launched - executed) & 1) cur_ ^= 1;
  nsteps_ += executed;
  return trig;
}

void DemEngine::set_overlap(bool on, hipStream_t comm_stream)
{
  sync();
  overlap_ = on;   // (set before the first rebuild: the list skin carries a 10 % margin in this mode)
  comm_stream_ = on ? comm_stream : nullptr;
}

void DemEngine::make_partitioned_streams(int comm_cus_per_xcd, hipStream_t* main_out, hipStream_t* comm_out)
{
  // Two streams on disjoint compute units: the small pack / RCCL / unpack kernels get `comm_cus_per_xcd` CUs of
  // every XCD to themselves, the sub-step kernels the rest.  Without the reservation the exchange kernels only get
  // compute units when the interior kernel's workgroups drain, i.e. nothing overlaps (measured, DESIGN.md section 7;
  // stream priority does not change that).  Bits 33k + 16j (k = XCD, j < comm_cus_per_xcd) address one CU per XCD
  // whether the mask numbers CUs XCD-major (bit / 32) or XCD-interleaved (bit % 8).
  sync();
  if (comm_cus_per_xcd < 1 || comm_cus_per_xcd > 2) fail("make_partitioned_streams: 1 or 2 CUs per XCD");
  uint32_t comm_mask[8] = {0, 0, 0, 0, 0, 0, 0, 0}, main_mask[8];
  for (int k = 0; k < 8; k++)
    for (int j = 0; j < comm_cus_per_xcd; j++) {
      const int bit = 33 * k + 16 * j;
      comm_mask[bit / 32] |= 1u << (bit % 32);
    }
  for (int w = 0; w < 8; w++) main_mask[w] = ~comm_mask[w];
  if (!masked_main_) {
    SF_HIP(hipExtStreamCreateWithCUMask(&masked_main_, 8, main_mask));
    SF_HIP(hipExtStreamCreateWithCUMask(&masked_comm_, 8, comm_mask));
  }
  stream_ = masked_main_;
  external_stream_ = true;   // the caller orders its own work against these streams (no per-call syncs)
  comm_stream_ = masked_comm_;
  *main_out = masked_main_;
  *comm_out = masked_comm_;
}

void DemEngine::mark_boundary()
{
  // Boundary atoms = owned atoms within the list cutoff of one of the slab's two x faces, rounded up to whole cell
  // layers: they contain every atom the forward halo sends and every atom whose list holds an atom of another GPU
  // (that atom lies beyond the face and closer than the list cutoff).  In the x-slowest order they are a prefix and
  // a suffix of the owned atoms.
  nb_ = n_lo_ = 0;
  n_hi_ = nlocal_;
  if (!nlocal_) return;
  if (!grid_.xslow) fail("overlapped halo: needs the x-slowest atom order (no SF_TILE)");
  const double cut = cutneighmax();
  const double cell = 1.0 / grid_.inv[0];
  int cx_lo = (int)std::ceil((sublo_[0] + cut - grid_.lo[0]) / cell - 1e-9);
  int cx_hi = (int)std::floor((subhi_[0] - cut - grid_.lo[0]) / cell + 1e-9);
  cx_lo = std::max(0, std::min(cx_lo, grid_.n[0]));
  cx_hi = std::max(cx_lo, std::min(cx_hi, grid_.n[0]));
  static_assert(F_SEND_COUNT2 == F_SEND_COUNT + 1, "adjacent counters");
  reset_flags(F_SEND_COUNT, 2, 0);
  k_count_layers<<<div_up(nlocal_, 1024), 1024, 0, stream_>>>(xr_[cur_].as<double4>(), nlocal_, grid_, cx_lo, cx_hi,
                                                            d_flags_ + F_SEND_COUNT);
  read_flags();
  n_lo_ = h_flags_[F_SEND_COUNT];
  n_hi_ = h_flags_[F_SEND_COUNT2];
  nb_ = n_lo_ + (nlocal_ - n_hi_);
  static const bool check = getenv("SF_CHECK_BOUNDARY") && atoi(getenv("SF_CHECK_BOUNDARY"));
  if (check) {
    // list-derived classification (sent, or has a neighbour rooted on another GPU) must be inside the two ranges
    k_mark_boundary<<<div_up(nlocal_, 256), 256, 0, stream_>>>(neigh_.as<int>(), numneigh_.as<int>(),
                                                               gsrc_.as<int>(), nullptr, 0, nullptr, 0, nlocal_, cap_,
                                                               isb_.as<unsigned char>(), 0, roots_ ? 1 : 0);
    const int ns = (int)(nsend_[0] + nsend_[1]);
    if (ns)
      k_mark_boundary<<<div_up(ns, 256), 256, 0, stream_>>>(nullptr, nullptr, nullptr, sendlist_[0].as<int>(),
                                                            (int)nsend_[0], sendlist_[1].as<int>(), (int)nsend_[1],
                                                            nlocal_, cap_, isb_.as<unsigned char>(), 1, roots_ ? 1 : 0);
    std::vector<unsigned char> h(nlocal_);
    SF_HIP(hipMemcpyAsync(h.data(), isb_.ptr, nlocal_, hipMemcpyDeviceToHost, stream_));
    sync();
    for (int i = n_lo_; i < n_hi_; i++)
      if (h[i]) fail("overlapped halo: atom %d talks to another GPU but lies in the interior range [%d, %d)", i, n_lo_, n_hi_);
  }
}

void DemEngine::overlap_begin()
{
  reset_flag(F_TRIG_LOCAL, INT_MAX);
  reset_flag(F_VOTE0, INT_MAX);
  reset_flag(F_VOTE1, INT_MAX);
  reset_flag(F_MARGIN_FAIL, 0);
}

void DemEngine::substep_part(int part, bool last, int kstep)
{
  if (part != 1 && part != 2) fail("substep_part: part must be 1 (interior) or 2 (boundary)");
  launch_substep(cur_, last ? 1 : 0, kstep, part);
}

void DemEngine::substep_flip(int kstep)
{
  // images of owned atoms (the images of received ghosts follow the unpack on the communication stream)
  launch_ghost_forward(cur_ ^ 1, kstep, 1, F_VOTE0 + ((kstep + 1) & 1));
  cur_ ^= 1;
}

int DemEngine::overlap_batch_end(int first_k, int launched, int last_kstep)
{
  read_flags();
  if (h_flags_[F_MARGIN_FAIL])
    fail("overlapped halo: an atom moved more than %.3g (5 %% of the skin) in one sub-step; the one-step-late "
         "rebuild vote of interior atoms is not safe at this speed -- run with SF_HALO_OVERLAP=0", 0.05 * skin_);
  const int trig = h_flags_[F_VOTE0 + (last_kstep & 1)];
  if (profiling_) {
    harvest_profile(trig);
    prof_used_ = 0;
  }
  const int executed = trig >= first_k + launched ? launched : std::max(0, trig + 1 - first_k);
  if ((launched - executed) & 1) cur_ ^= 1;
  nsteps_ += executed;
  return trig;
}

void DemEngine::set_flag_buffer(int* dev)
{
  sync();
  if (!dev) dev = own_flags_;
  if (dev != d_flags_) {
    SF_HIP(hipMemcpyAsync(dev, d_flags_, sizeof(int) * F_NFLAGS, hipMemcpyDeviceToDevice, stream_));
    sync();
    d_flags_ = dev;
  }
}

bool DemEngine::need_rebuild()
{
  read_flags();
  return h_flags_[F_TRIGGER] != INT_MAX;
}

void DemEngine::run(int nsteps)
{
  if (!setup_done_) setup();
  if (nsteps <= 0) return;
  // (an engine the SCRIPT decomposed never gets here: sf_lammps_step routes it to sf_slab_step, sf_lammps_api.hip)
  if (have_subdomain_) fail("DemEngine::run on a sub-domain set by sf_dem_set_subdomain / sf_slab_init / sf_brick_init: step it with sf_slab_step (or the sf_dem_* pieces)");
  run_base_step_ = nsteps_;
  choose_kernel();
  reset_flag(F_TRIGGER, INT_MAX);
  launch_initial_integrate();
  launch_ghost_forward(cur_, 0);
  int k = 0;
  SF_HIP(hipEventRecord(ev0_, stream_));
  while (k < nsteps) {
    const int base = cur_;
    prof_used_ = 0;
    // queue up to where the next rebuild is expected (RebuildPredictor), not blindly to the end of the run
    predict_.overshoot = nlocal_ >= 200000 && !(getenv("SF_QUEUE_OVERSHOOT") && !atoi(getenv("SF_QUEUE_OVERSHOOT")));
    const int end = k + predict_.chunk(run_base_step_ + k, nsteps - k);
    for (int s = k; s < end; s++) {
      const int in_buf = (base + (s - k)) & 1;
      launch_substep(in_buf, (s == nsteps - 1) ? 1 : 0, s);
      launch_ghost_forward(in_buf ^ 1, s);
    }
    read_flags();
    const int trig = h_flags_[F_TRIGGER];
    if (profiling_) {
      harvest_profile(trig);   // (INT_MAX: every launch of the batch ran)
      prof_used_ = 0;
    }
    if (trig == INT_MAX) {
      cur_ = (base + (end - k)) & 1;
      k = end;
    } else {
      // sub-steps k..trig ran; the list went stale for sub-step trig+1 (trig = -1: for sub-step 0)
      const int done = trig + 1 - k;
      cur_ = (base + done) & 1;
      k = trig + 1;
      InRunGuard guard(*this);   // (a rebuild that throws must not leave the engine marked "inside a run")
      const auto t_rb = std::chrono::steady_clock::now();   // (the stream is idle: read_flags above synchronised)
      rebuild();
      if (profiling_) {
        sync();
        prof_rebuild_ms_ += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_rb).count();
        prof_rebuilds_++;
      }
      guard.release();
      predict_.rebuilt(run_base_step_ + k);
    }
  }
  SF_HIP(hipEventRecord(ev1_, stream_));
  sync();
  float ms = 0.f;
  SF_HIP(hipEventElapsedTime(&ms, ev0_, ev1_));
  last_substep_ms_ = ms / nsteps;
  nsteps_ += nsteps;
}

// ------------------------------------------------------------------------------------------------
// data exchange (lammps_* surface)
// ------------------------------------------------------------------------------------------------
namespace {
// one-off scratch of the rarely used accessors (history, pair count)
struct ScratchD {
  double* p = nullptr;
  explicit ScratchD(size_t n) { SF_HIP(hipMalloc(&p, sizeof(double) * (n ? n : 1))); }
  ~ScratchD() { (void)hipFree(p); }
};
struct ScratchI {
  int* p = nullptr;
  explicit ScratchI(size_t n) { SF_HIP(hipMalloc(&p, sizeof(int) * (n ? n : 1))); }
  ~ScratchI() { (void)hipFree(p); }
};
}  // namespace

// Staging buffers of the per-CFD-step boundary calls (lammps_put_local_info / lammps_get_local_info / get_forces):
// persistent, grown geometrically -- a hipMalloc / hipFree pair per call is a device-wide synchronisation each.
double* DemEngine::io_doubles(int which, size_t n)
{
  IoBuf& b = io_d_[which];
  if (n > b.n) {
    if (b.p) SF_HIP(hipFree(b.p));
    b.n = n + n / 2 + 1024;
    SF_HIP(hipMalloc(&b.p, sizeof(double) * b.n));
  }
  return static_cast<double*>(b.p);
}

int* DemEngine::io_ints(int which, size_t n)
{
  IoBuf& b = io_i_[which];
  if (n > b.n) {
    if (b.p) SF_HIP(hipFree(b.p));
    b.n = n + n / 2 + 1024;
    SF_HIP(hipMalloc(&b.p, sizeof(int) * b.n));
  }
  return static_cast<int*>(b.p);
}

void DemEngine::get_local_info(double* x, double* v, int* foamCpuId, int* tag)
{
  const int n = nlocal_;
  if (!n) return;
  double *dx = io_doubles(0, 3 * (size_t)n), *dv = io_doubles(1, 3 * (size_t)n);
  k_pack_info<<<div_up(n, 256), 256, 0, stream_>>>(d_xr(), d_vm(), d_om(), d_force(), d_torque(), n, dx, dv,
                                                   nullptr, nullptr, nullptr, nullptr, nullptr);
  if (x) SF_HIP(hipMemcpyAsync(x, dx, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, stream_));
  if (v) SF_HIP(hipMemcpyAsync(v, dv, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, stream_));
  if (foamCpuId)
    SF_HIP(hipMemcpyAsync(foamCpuId, foamCpuId_.ptr, sizeof(int) * n, hipMemcpyDeviceToHost, stream_));
  if (tag) SF_HIP(hipMemcpyAsync(tag, tag_.ptr, sizeof(int) * n, hipMemcpyDeviceToHost, stream_));
  sync();
}

void DemEngine::get_initial_info(double* x, double* v, double* diam, double* rho, int* tag, int* type)
{
  const int n = nlocal_;
  if (!n) return;
  ScratchD dx(3 * (size_t)n), dv(3 * (size_t)n), dd(n), dr(n);
  k_pack_info<<<div_up(n, 256), 256, 0, stream_>>>(d_xr(), d_vm(), d_om(), d_force(), d_torque(), n, dx.p, dv.p,
                                                   nullptr, nullptr, nullptr, dd.p, dr.p);
  if (x) SF_HIP(hipMemcpyAsync(x, dx.p, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, stream_));
  if (v) SF_HIP(hipMemcpyAsync(v, dv.p, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, stream_));
  if (diam) SF_HIP(hipMemcpyAsync(diam, dd.p, sizeof(double) * n, hipMemcpyDeviceToHost, stream_));
  if (rho) SF_HIP(hipMemcpyAsync(rho, dr.p, sizeof(double) * n, hipMemcpyDeviceToHost, stream_));
  if (tag) SF_HIP(hipMemcpyAsync(tag, tag_.ptr, sizeof(int) * n, hipMemcpyDeviceToHost, stream_));
  if (type) SF_HIP(hipMemcpyAsync(type, type_.ptr, sizeof(int) * n, hipMemcpyDeviceToHost, stream_));
  sync();
}

void DemEngine::get_forces(double* f, double* torque, double* omega, int* tag)
{
  const int n = nlocal_;
  if (!n) return;
  double *df = io_doubles(0, 3 * (size_t)n), *dt = io_doubles(1, 3 * (size_t)n), *dw = io_doubles(2, 3 * (size_t)n);
  k_pack_info<<<div_up(n, 256), 256, 0, stream_>>>(d_xr(), d_vm(), d_om(), d_force(), d_torque(), n, nullptr,
                                                   nullptr, dw, df, dt, nullptr, nullptr);
  if (f) SF_HIP(hipMemcpyAsync(f, df, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, stream_));
  if (torque) SF_HIP(hipMemcpyAsync(torque, dt, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, stream_));
  if (omega) SF_HIP(hipMemcpyAsync(omega, dw, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, stream_));
  if (tag) SF_HIP(hipMemcpyAsync(tag, tag_.ptr, sizeof(int) * n, hipMemcpyDeviceToHost, stream_));
  sync();
}

void DemEngine::put_local_info(int n, const double* fdrag, const int* foamCpuId, const int* tagIn)
{
  if (!have_fdrag_) fail("lammps_put_local_info: no `fix ... fdrag` defined");  // library.cpp:324-333 derefs NULL
  if (n != nlocal_)
    fprintf(stderr, "Incoming drag not consistent with local particle number.\nIncoming drag is: %5d, local "
                    "particle number is: %5d.\n", n, nlocal_);  // library.cpp:335-341 (prints, continues)
  const int m = std::min(n, nlocal_);
  if (!m) return;
  // tag -> index table: indices only change when the atoms are re-sorted (a rebuild) or created / deleted / migrated
  if ((size_t)max_tag_ > tagmap_alloc_) {
    if (tagmap_) SF_HIP(hipFree(tagmap_));
    tagmap_alloc_ = (size_t)max_tag_ + max_tag_ / 4 + 16;
    SF_HIP(hipMalloc(&tagmap_, sizeof(int) * tagmap_alloc_));
    tagmap_builds_ = -1;
  }
  if (tagmap_builds_ != order_version_) {
    SF_HIP(hipMemsetAsync(tagmap_, 0xff, sizeof(int) * tagmap_alloc_, stream_));
    k_tag_map<<<div_up(nlocal_, 256), 256, 0, stream_>>>(tag_.as<int>(), nlocal_, tagmap_, max_tag_);
    tagmap_builds_ = order_version_;
  }
  double* din = io_doubles(0, 3 * (size_t)m);
  int *dtag = io_ints(0, m), *dcpu = io_ints(1, m);
  SF_HIP(hipMemcpyAsync(din, fdrag, sizeof(double) * 3 * m, hipMemcpyHostToDevice, stream_));
  SF_HIP(hipMemcpyAsync(dtag, tagIn, sizeof(int) * m, hipMemcpyHostToDevice, stream_));
  if (foamCpuId) SF_HIP(hipMemcpyAsync(dcpu, foamCpuId, sizeof(int) * m, hipMemcpyHostToDevice, stream_));
  k_put_fdrag<<<div_up(m, 256), 256, 0, stream_>>>(din, dtag, foamCpuId ? dcpu : nullptr, m, tagmap_,
                                                   max_tag_, fdrag_.as<double>(), foamCpuId_.as<int>(), cap_,
                                                   d_flags_);
  read_flags();
  if (h_flags_[F_LOST] == 2) {
    reset_flag(F_LOST, 0);
    fail("lammps_put_local_info: incoming tag not owned by this rank");
  }
}

long long DemEngine::npairs_full()
{
  if (!nlocal_ || !have_list_) return 0;
  unsigned long long* d = count64_;   // (persistent: a hipMalloc / hipFree pair per call synchronises the device)
  SF_HIP(hipMemsetAsync(d, 0, sizeof(unsigned long long), stream_));
  k_count_pairs<<<div_up(nlocal_, 1024), 1024, 0, stream_>>>(numneigh_.as<int>(), nlocal_, d);
  unsigned long long h = 0;
  SF_HIP(hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, stream_));
  sync();
  return (long long)h;
}

long long DemEngine::get_history(long long max, int* tag_i, int* tag_j, double* shear)
{
  if (!nlocal_ || !have_list_) return 0;
  unsigned long long* d = nullptr;
  SF_HIP(hipMalloc(&d, sizeof(unsigned long long)));
  SF_HIP(hipMemsetAsync(d, 0, sizeof(unsigned long long), stream_));
  ScratchI ti(max), tj(max);
  ScratchD sh(3 * (size_t)max);
  k_collect_history<<<div_up(nlocal_, 256), 256, 0, stream_>>>(neigh_.as<int>(), numneigh_.as<int>(),
                                                               shear_[cur_].as<double>(), tag_.as<int>(), nlocal_, cap_,
                                                               d, max, ti.p, tj.p, sh.p, roots_ ? 1 : 0);
  unsigned long long h = 0;
  SF_HIP(hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, stream_));
  sync();
  (void)hipFree(d);
  const long long n = std::min<long long>((long long)h, max);
  if (n > 0) {
    SF_HIP(hipMemcpy(tag_i, ti.p, sizeof(int) * n, hipMemcpyDeviceToHost));
    SF_HIP(hipMemcpy(tag_j, tj.p, sizeof(int) * n, hipMemcpyDeviceToHost));
    SF_HIP(hipMemcpy(shear, sh.p, sizeof(double) * 3 * n, hipMemcpyDeviceToHost));
  }
  return (long long)h;
}

void DemEngine::get_wall_shear(int w, double* shear)
{
  if (w < 0 || w >= nwalls_) fail("no such wall fix %d", w);
  const int n = nlocal_;
  std::vector<double> row((size_t)n);
  std::vector<unsigned char> wt((size_t)n);
  SF_HIP(hipMemcpyAsync(wt.data(), wtouch_.ptr, n, hipMemcpyDeviceToHost, stream_));
  sync();
  for (int c = 0; c < 3; c++) {
    SF_HIP(hipMemcpy(row.data(), wshear_.as<double>() + (size_t)(3 * w + c) * cap_, sizeof(double) * n,
                     hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) shear[3 * i + c] = (wt[i] & (1u << w)) ? row[i] : 0.0;
  }
}

}  // namespace sf
