// sf_physics.h -- per-pair / per-particle force laws as __device__ inline functions (gfx950, FP64).
//
// One definition of each law, shared by the fused sub-step kernel (sf_dem_kernels.hip) and the
// stand-alone PairStyle/FixStyle entry points (sf_kernels_api.hip).  Expression order follows the
// reference so that results agree with its CPU build to rounding (FMA contraction is the only
// difference):
//   hertz_history_law  : interfaceToLammps/pair_gran_hertzFix_history.cpp:142-261 (pair),
//                        interfaceToLammps/fix_wall_granFix.cpp:558-679 (wall twin)
//   hooke_history_law  : interfaceToLammps/fix_wall_granFix.cpp:441-554 and its pair twin
//                        (LAMMPS 1Feb14 gran/hooke/history); style 3 = the plain law without history,
//                        fix_wall_granFix.cpp:347-437 and its pair twin (LAMMPS 1Feb14 gran/hooke)
//   cohesive_ccel      : interfaceToLammps/fix_cohesive.cpp:184-196, :236-245
//   lubricate_poly_pair: interfaceToLammps/pair_lubricate_poly.cpp:241-399
#pragma once
#include <hip/hip_runtime.h>

namespace sf {

constexpr double kPi = 3.14159265358979323846;          // MathConst::MY_PI
constexpr double kPiTypo = 3.14159265358917323846;      // fix_fluid_drag.cpp:147, library.cpp:200,460

struct GranParams {
  double kn, kt, gamman, gammat, xmu;
  double beta;      // Hertz: -ln(e)/sqrt(ln(e)^2+pi^2), evaluated once on the host (pure function of gamman)
  // host-folded Hertz constants (see hertz_history_law)
  double h_sn, h_cn, h_ct, h_inv_ct, h_c56beta, h_stsn_c56beta;
  double inv_kt;    // 1 / kt (Hooke law)
  int dampflag;
  int style;        // 0 none, 1 hooke/history, 2 hertzFix/history
};

void fold_hertz_constants(GranParams& p);   // sf_dem.hip (host)

struct Vec3 {
  double x, y, z;
};
__device__ __forceinline__ Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ Vec3 operator*(double s, Vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ double dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// ---- FP64 reciprocal / square root without the IEEE corner-case scaffolding -------------------------------
// hipcc expands an f64 divide to ~11 and an f64 sqrt to ~19 instructions (v_div_scale/v_div_fixup, ldexp
// rescaling and class tests for denormals/inf/nan).  The contact laws only ever see finite, normal, positive
// arguments (squared distances, masses, radii, overlaps), so these use the hardware seed (v_rcp_f64 / v_rsq_f64,
// ~2^-24 relative) and Newton / Goldschmidt steps: results are within 1-2 ulp of the IEEE ones.
#ifndef SF_FAST_MATH
#define SF_FAST_MATH 1
#endif
__device__ __forceinline__ double sf_rcp(double x)
{
#if SF_FAST_MATH
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
#else
  return 1.0 / x;
#endif
}
// g = sqrt(x), rinv = 1/sqrt(x) for x > 0 (normal range)
__device__ __forceinline__ void sf_sqrt_rsqrt(double x, double& g, double& rinv)
{
#if SF_FAST_MATH
  const double y = __builtin_amdgcn_rsq(x);
  double gg = x * y, h = 0.5 * y;
  double r = fma(-h, gg, 0.5);
  gg = fma(gg, r, gg);
  h = fma(h, r, h);
  r = fma(-h, gg, 0.5);
  gg = fma(gg, r, gg);
  h = fma(h, r, h);
  gg = fma(fma(-gg, gg, x), h, gg);
  g = gg;
  rinv = h + h;
#else
  g = sqrt(x);
  rinv = 1.0 / g;
#endif
}
// sqrt(x) for x >= 0 (0 stays 0)
__device__ __forceinline__ double sf_sqrt(double x)
{
#if SF_FAST_MATH
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  g = fma(fma(-g, g, x), h, g);
  g = fma(fma(-g, g, x), h, g);
  return x > 0.0 ? g : 0.0;
#else
  return sqrt(x);
#endif
}

// natural logarithm of a finite, normal x (positive: the fast path; zero / negative: -inf / NaN like the library): frexp, (m - 1) / (m + 1) and the odd atanh series to s^21 -- ~30
// instructions against the ~100 of the library's log with its special cases; within 1-2 ulp of it (the lubrication
// series takes one per listed pair: pair_lubricate_poly.cpp:311-333)
__device__ __forceinline__ double sf_log(double x)
{
#if SF_FAST_MATH
  int e = __builtin_amdgcn_frexp_exp(x);
  double m = __builtin_amdgcn_frexp_mant(x);               // [0.5, 1)
  const bool low = m < 0.70710678118654752440;
  m = low ? m + m : m;                                     // [sqrt(1/2), sqrt(2))
  e = low ? e - 1 : e;
  // (x <= 0 takes the library's answers -- log(0) = -inf, log(negative) = NaN -- through the same arithmetic: the reference
  // leaves h_sep negative for an overlapping pair beyond the inner cutoff, pair_lubricate_poly.cpp:286-300, and its run
  // shows NaN forces; a finite value made up from the bits of a negative number would hide that)
  const double s = x > 0.0 ? (m - 1.0) * sf_rcp(m + 1.0)   // |s| <= 0.1716
                           : (x == 0.0 ? -__builtin_huge_val() : __builtin_nan(""));
  const double z = s * s;
  double p = 2.0 / 21.0;
  p = fma(p, z, 2.0 / 19.0);
  p = fma(p, z, 2.0 / 17.0);
  p = fma(p, z, 2.0 / 15.0);
  p = fma(p, z, 2.0 / 13.0);
  p = fma(p, z, 2.0 / 11.0);
  p = fma(p, z, 2.0 / 9.0);
  p = fma(p, z, 2.0 / 7.0);
  p = fma(p, z, 2.0 / 5.0);
  p = fma(p, z, 2.0 / 3.0);
  p = p * z;                                               // log(m) = 2 s + s p
  const double de = (double)e;
  // e ln2 in two pieces (the high one exact for |e| < 2^11), smallest terms first
  return fma(de, 0.693147180369123816490, fma(s, p, fma(de, 1.90821492927058770002e-10, s + s)));
#else
  return log(x);
#endif
}

struct ContactIn {
  Vec3 del;        // from partner (or wall) to the particle
  double rsq;
  Vec3 vr;         // relative translational velocity
  Vec3 wsum;       // radi*omega_i + radj*omega_j   (wall: radius*omega)
  double meff;
  double overlap;  // radsum - r  |  radius - r
  double reff;     // overlap*radi*radj/radsum | overlap*radius  (Hertz only)
  double r, rinv;
};

struct ContactOut {
  Vec3 F;          // force on the particle
  Vec3 tor;        // rinv * (del x fs); caller applies -radius
};

// Hertzian history contact.  shear is read/updated in place (registers).
// Same algebra as the reference, arranged for the FP64 pipe of gfx950 (an IEEE f64 divide or sqrt is a
// ~12-instruction dependent sequence): products of constants are folded on the host (GranParams::h_*),
// 1/rsq = rinv*rinv, sqrt(st*meff) = sqrt(sn*meff)*sqrt(st/sn), and the Coulomb test compares squares so the
// two norms are only taken when a contact actually slides.  Each re-association moves a result by <= 1-2 ulp
// (tests/test_dem_gpu.py holds the HIP path to 1e-12 of the oracle after one evaluation).
__device__ __forceinline__ void hertz_history_law(const GranParams& p, double dt, bool shearupdate,
                                                  const ContactIn& c, Vec3& sh, ContactOut& o)
{
  const double rsqinv = c.rinv * c.rinv;
  const double vnnr = dot(c.vr, c.del);
  const double s = vnnr * rsqinv;
  const Vec3 vt = {c.vr.x - c.del.x * s, c.vr.y - c.del.y * s, c.vr.z - c.del.z * s};
  const Vec3 wr = c.rinv * c.wsum;

  const double polyhertz = sf_sqrt(c.reff);
  const double sqsn = sf_sqrt(p.h_sn * polyhertz * c.meff);    // sqrt(sn*meff), sn = (2/1.82) kn polyhertz
  const double damp = p.h_c56beta * vnnr * rsqinv;             // 2 sqrt(5/6) beta vnnr / rsq
  const double ccel = polyhertz * p.h_cn * c.overlap * c.rinv - sqsn * damp;   // h_cn = 4/5.46 kn

  const Vec3 vtr = {vt.x - (c.del.z * wr.y - c.del.y * wr.z), vt.y - (c.del.x * wr.z - c.del.z * wr.x),
                    vt.z - (c.del.y * wr.x - c.del.x * wr.y)};
  if (shearupdate) {
    sh.x += vtr.x * dt;
    sh.y += vtr.y * dt;
    sh.z += vtr.z * dt;
  }
  const double shr2 = dot(sh, sh);
  const double rsht = dot(sh, c.del) * rsqinv;
  if (shearupdate) {
    sh.x -= rsht * c.del.x;
    sh.y -= rsht * c.del.y;
    sh.z -= rsht * c.del.z;
  }
  const double kts = polyhertz * p.h_ct;                       // h_ct = 8/8.84 kt
  const double sdamp = sqsn * p.h_stsn_c56beta;                // sqrt(st*meff) 2 sqrt(5/6) beta
  Vec3 fs = {-kts * sh.x - sdamp * vtr.x, -kts * sh.y - sdamp * vtr.y, -kts * sh.z - sdamp * vtr.z};
  const double fs2 = dot(fs, fs);
  const double fn = p.xmu * fabs(ccel * c.r);
  if (fs2 > fn * fn) {
    if (shr2 != 0.0) {
      double fsmag, fsinv;
      sf_sqrt_rsqrt(fs2, fsmag, fsinv);
      const double ratio = fn * fsinv;
      const double qs = sdamp * p.h_inv_ct;                    // / 8.84 * 8 / kt
      const Vec3 q = {qs * vtr.x, qs * vtr.y, qs * vtr.z};
      sh.x = ratio * (sh.x + q.x) - q.x;
      sh.y = ratio * (sh.y + q.y) - q.y;
      sh.z = ratio * (sh.z + q.z) - q.z;
      fs = ratio * fs;
    } else
      fs = {0.0, 0.0, 0.0};
  }
  o.F = {c.del.x * ccel + fs.x, c.del.y * ccel + fs.y, c.del.z * ccel + fs.z};
  o.tor = {c.rinv * (c.del.y * fs.z - c.del.z * fs.y), c.rinv * (c.del.z * fs.x - c.del.x * fs.z),
           c.rinv * (c.del.x * fs.y - c.del.y * fs.x)};
}

// Hookean history contact (the law all of the reference's example cases run: gran/hooke/history).  Same algebra;
// 1/rsq = rinv^2, the Coulomb test compares squares, norms only when a contact slides.
__device__ __forceinline__ void hooke_history_law(const GranParams& p, double dt, bool shearupdate,
                                                  const ContactIn& c, Vec3& sh, ContactOut& o)
{
  const double rsqinv = c.rinv * c.rinv;
  const double vnnr = dot(c.vr, c.del);
  const double s = vnnr * rsqinv;
  const Vec3 vt = {c.vr.x - c.del.x * s, c.vr.y - c.del.y * s, c.vr.z - c.del.z * s};
  const Vec3 wr = c.rinv * c.wsum;
  const double damp = c.meff * p.gamman * vnnr * rsqinv;
  const double ccel = p.kn * c.overlap * c.rinv - damp;
  const Vec3 vtr = {vt.x - (c.del.z * wr.y - c.del.y * wr.z), vt.y - (c.del.x * wr.z - c.del.z * wr.x),
                    vt.z - (c.del.y * wr.x - c.del.x * wr.y)};
  if (p.style == 3) {
    // plain Hookean contact, no shear history (`pair_style gran/hooke` [3P]; its wall twin is
    // FixWallGranFix::hooke, fix_wall_granFix.cpp:347-437): the tangential force is the velocity damping alone, capped
    // by Coulomb.  The history slot of the pair stays zero.
    const double vrel = sqrt(dot(vtr, vtr));
    const double fn = p.xmu * fabs(ccel * c.r);
    const double fsd = c.meff * p.gammat * vrel;
    const double ft = vrel != 0.0 ? (fn < fsd ? fn : fsd) / vrel : 0.0;
    const Vec3 fs = {-ft * vtr.x, -ft * vtr.y, -ft * vtr.z};
    sh = {0.0, 0.0, 0.0};
    o.F = {c.del.x * ccel + fs.x, c.del.y * ccel + fs.y, c.del.z * ccel + fs.z};
    o.tor = {c.rinv * (c.del.y * fs.z - c.del.z * fs.y), c.rinv * (c.del.z * fs.x - c.del.x * fs.z),
             c.rinv * (c.del.x * fs.y - c.del.y * fs.x)};
    return;
  }
  if (shearupdate) {
    sh.x += vtr.x * dt;
    sh.y += vtr.y * dt;
    sh.z += vtr.z * dt;
  }
  const double shr2 = dot(sh, sh);
  const double rsht = dot(sh, c.del) * rsqinv;
  if (shearupdate) {
    sh.x -= rsht * c.del.x;
    sh.y -= rsht * c.del.y;
    sh.z -= rsht * c.del.z;
  }
  const double mg = c.meff * p.gammat;
  Vec3 fs = {-(p.kt * sh.x + mg * vtr.x), -(p.kt * sh.y + mg * vtr.y), -(p.kt * sh.z + mg * vtr.z)};
  const double fs2 = dot(fs, fs);
  const double fn = p.xmu * fabs(ccel * c.r);
  if (fs2 > fn * fn) {
    if (shr2 != 0.0) {
      double fsmag, fsinv;
      sf_sqrt_rsqrt(fs2, fsmag, fsinv);
      const double ratio = fn * fsinv;
      const double qs = mg * p.inv_kt;
      const Vec3 q = {qs * vtr.x, qs * vtr.y, qs * vtr.z};
      sh.x = ratio * (sh.x + q.x) - q.x;
      sh.y = ratio * (sh.y + q.y) - q.y;
      sh.z = ratio * (sh.z + q.z) - q.z;
      fs = ratio * fs;
    } else
      fs = {0.0, 0.0, 0.0};
  }
  o.F = {c.del.x * ccel + fs.x, c.del.y * ccel + fs.y, c.del.z * ccel + fs.z};
  o.tor = {c.rinv * (c.del.y * fs.z - c.del.z * fs.y), c.rinv * (c.del.z * fs.x - c.del.x * fs.z),
           c.rinv * (c.del.x * fs.y - c.del.y * fs.x)};
}

template <int STYLE>
__device__ __forceinline__ void gran_history_law(const GranParams& p, double dt, bool shearupdate,
                                                 const ContactIn& c, Vec3& sh, ContactOut& o)
{
  if (STYLE == 2) hertz_history_law(p, dt, shearupdate, c, sh, o);
  else hooke_history_law(p, dt, shearupdate, c, sh, o);
}

struct CoheParams {
  double ah, lam, smin, smax;
  int opt;
  int enabled;
};

// scalar cohesive coefficient: force on i = del * ccel / r  (fix_cohesive.cpp:187-203, :239-252).
// The reference's chains of divisions (a / b / c / d ...) are evaluated as one product of the denominators and one
// reciprocal (an f64 divide is ~11 dependent instructions on gfx950): <= a few ulp from the reference's order.
__device__ __forceinline__ double cohesive_ccel(const CoheParams& p, double r, double radsum)
{
  const double del = r - radsum;
  double ccel;
  if (p.opt == 0) {
    const double PInv = 0.3183098861837907;   // 0.25 / atan(1.0)
    const double lam = p.lam;
    if (del > lam * PInv) {
      const double id = sf_rcp(del);
      const double ld = lam * id;
      ccel = -p.ah * radsum * lam * (6.4988e-3 - 4.5316e-4 * ld + 1.1326e-5 * ld * ld) * (id * id * id);
    } else {
      const double s = (del > p.smin) ? del : p.smin;
      const double q = lam + 11.121 * s;
      ccel = -p.ah * (lam + 22.242 * s) * radsum * lam * sf_rcp(24.0 * (q * q) * (s * s));
    }
  } else {
    const double r2 = radsum * radsum;
    const double r6 = r2 * r2 * r2;  // pow(radsum,6)
    if (del > p.smin) {
      const double rr = r + radsum;
      ccel = -p.ah * r6 * sf_rcp(6.0 * (del * del) * (rr * rr) * (r * r * r));
    } else {
      const double a = p.smin + 2.0 * radsum, b2 = p.smin + radsum;
      ccel = -p.ah * r6 * sf_rcp(6.0 * (p.smin * p.smin) * (a * a) * (b2 * b2 * b2));
    }
  }
  return ccel;
}

struct LubParams {
  double mu, cut_inner, cut_global, R0, RT0, RS0, vxmu2f;
  int flaglog, flagfld, flagHI, flagVF;
  int enabled;
};

// lubrication force/torque on i from neighbour j (full list: only i is updated), Ef = 0.
// Same algebra as pair_lubricate_poly.cpp:241-399; its ~25 divisions are four reciprocals (1/r, 1/radi, 1/beta1,
// 1/h_sep) and products, the constant divisors are folded.
// (r, rinv = sf_sqrt_rsqrt of the pair's squared distance: the fused kernel shares them between its arms)
__device__ __forceinline__ void lubricate_poly_pair(const LubParams& p, Vec3 del, double r, double rinv, double radi,
                                                    double radj, Vec3 vi0, Vec3 vj0, Vec3 wi, Vec3 wj,
                                                    Vec3& F, Vec3& T)
{
  const Vec3 n = {del.x * rinv, del.y * rinv, del.z * rinv};
  const Vec3 xl = {-n.x * radi, -n.y * radi, -n.z * radi};
  const Vec3 jl = {-n.x * radj, -n.y * radj, -n.z * radj};
  const Vec3 vi = {vi0.x + (wi.y * xl.z - wi.z * xl.y), vi0.y + (wi.z * xl.x - wi.x * xl.z),
                   vi0.z + (wi.x * xl.y - wi.y * xl.x)};
  const Vec3 vj = {vj0.x - (wj.y * jl.z - wj.z * jl.y), vj0.y - (wj.z * jl.x - wj.x * jl.z),
                   vj0.z - (wj.x * jl.y - wj.y * jl.x)};
  double h_sep = r - radi - radj;
  if (r < p.cut_inner) h_sep = 100 * radi + 100 * radj;  // the reference's edit, :294-295
  const double iradi = sf_rcp(radi);
  h_sep = h_sep * iradi;
  const double beta0 = radj * iradi;
  const double beta1 = 1.0 + beta0;
  const double ib1 = sf_rcp(beta1), ib12 = ib1 * ib1;
  const double ih = sf_rcp(h_sep);
  const double b02 = beta0 * beta0;
  const double mu_r = kPi * p.mu * radi;
  double a_sq, a_sh = 0.0, a_pu = 0.0;
  if (p.flaglog) {
    const double b03 = b02 * beta0, b04 = b02 * b02;
    const double ib13 = ib12 * ib1, ib14 = ib12 * ib12;
    const double lg = -sf_log(h_sep);      // log(1 / h_sep)
    const double hl = h_sep * lg;
    a_sq = b02 * ib12 * ih + (1.0 + 7.0 * beta0 + b02) * (0.2 * ib13) * lg;
    a_sq += (1.0 + 18.0 * beta0 - 29.0 * b02 + 18.0 * b03 + b04) * ((1.0 / 21.0) * ib14) * hl;
    a_sq *= 6.0 * mu_r;
    a_sh = 4.0 * beta0 * (2.0 + beta0 + 2.0 * b02) * ((1.0 / 15.0) * ib13) * lg;
    a_sh += 4.0 * (16.0 - 45.0 * beta0 + 58.0 * b02 - 45.0 * b03 + 16.0 * b04) * ((1.0 / 375.0) * ib14) * hl;
    a_sh *= 6.0 * mu_r;
    a_pu = beta0 * (4.0 + beta0) * (0.1 * ib12) * lg;
    a_pu += (32.0 - 33.0 * beta0 + 83.0 * b02 + 43.0 * b03) * ((1.0 / 250.0) * ib13) * hl;
    a_pu *= 8.0 * mu_r * (radi * radi);
  } else
    a_sq = 6.0 * mu_r * (b02 * ib12 * ih);

  const Vec3 vr = vi - vj;
  const double vnnr = dot(vr, n);
  const Vec3 vn = {vnnr * n.x, vnnr * n.y, vnnr * n.z};
  const Vec3 vt = vr - vn;
  Vec3 f = a_sq * vn;
  if (p.flaglog) f = f + a_sh * vt;
  f = p.vxmu2f * f;
  F = F - f;
  if (p.flaglog) {
    const Vec3 t = {xl.y * f.z - xl.z * f.y, xl.z * f.x - xl.x * f.z, xl.x * f.y - xl.y * f.x};
    T = T - p.vxmu2f * t;
    const Vec3 dw = wi - wj;
    const double wdotn = dot(dw, n);
    const Vec3 wt = {dw.x - wdotn * n.x, dw.y - wdotn * n.y, dw.z - wdotn * n.z};
    T = T - p.vxmu2f * (a_pu * wt);
  }
}

}  // namespace sf
