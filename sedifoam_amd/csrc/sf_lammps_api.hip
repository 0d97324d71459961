// sf_lammps_api.hip -- the LAMMPS-shaped plug-in surface of the DEM engine:
//   * input-script commands for the pair_style / fix lines of the hot path
//     (style names and argument order as registered in interfaceToLammps/style_user.h:43-50,65-74
//      and parsed in pair_gran_hertzFix_history.cpp:293-317, fix_fluid_drag.cpp:31-55,
//      fix_cohesive.cpp:38-57, fix_wall_granFix.cpp:44-141, [3P] pair_lubricate.cpp settings)
//   * the patched C library interface, interfaceToLammps/library.h:29-63 / library.cpp
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/sedifoam_amd.h"
#include "sf_handles.h"
#include "sf_roctx.h"

using sf::DemEngine;
using sf::SfLammps;

namespace {

std::vector<std::string> split(const std::string& line)
{
  std::string s = line.substr(0, line.find('#'));
  std::istringstream is(s);
  std::vector<std::string> w;
  std::string t;
  while (is >> t) w.push_back(t);
  return w;
}

double num(const std::string& s)
{
  char* end = nullptr;
  const double v = std::strtod(s.c_str(), &end);
  if (end == s.c_str() || *end != '\0') sf::fail("Expected floating point parameter in input script or data file: %s", s.c_str());
  return v;
}

int inum(const std::string& s)
{
  char* end = nullptr;
  const long v = std::strtol(s.c_str(), &end, 10);
  if (end == s.c_str() || *end != '\0') sf::fail("Expected integer parameter in input script or data file: %s", s.c_str());
  return (int)v;
}

// [3P] LAMMPS 1Feb14 ProcMap::onelevel_grid -> factor / cull_user / best_factors: of all px py pz with product N that
// agree with the non-`*` entries of the `processors` command, the one with the least sub-domain surface
//   xprd yprd / (px py) + xprd zprd / (px pz) + yprd zprd / (py pz);
// factorisations are visited with px slowest and the first strictly smaller surface wins.
bool procgrid_rule(int n, const double lo[3], const double hi[3], const int user[3], int P[3])
{
  const double xprd = hi[0] - lo[0], yprd = hi[1] - lo[1], zprd = hi[2] - lo[2];
  const double area[3] = {xprd * yprd, xprd * zprd, yprd * zprd};
  double best = 2.0 * (area[0] + area[1] + area[2]);
  P[0] = P[1] = P[2] = 0;
  for (int i = 1; i <= n; i++) {
    if (n % i) continue;
    const int nyz = n / i;
    for (int j = 1; j <= nyz; j++) {
      if (nyz % j) continue;
      const int k = nyz / j;
      if ((user[0] && user[0] != i) || (user[1] && user[1] != j) || (user[2] && user[2] != k)) continue;
      const double surf = area[0] / i / j + area[1] / i / k + area[2] / j / k;
      if (surf < best) {
        best = surf;
        P[0] = i; P[1] = j; P[2] = k;
      }
    }
  }
  return P[0] != 0;
}

void choose_procgrid(const SfLammps& L, int P[3])
{
  double lo[3], hi[3];
  int per[3];
  L.eng.box(lo, hi, per);
  if (!procgrid_rule(L.world_size, lo, hi, L.procgrid, P)) sf::fail("Bad grid of processors");   // [3P] Comm::set_proc_grid
}

// the processor grid exists once the box does ([3P] read_data / create_box call Comm::set_proc_grid): one brick per rank
void decompose(SfLammps& L)
{
  if (L.world_size <= 1 || L.decomposed) return;
  if (L.halo) sf::fail("the engine was decomposed by sf_slab_init / sf_brick_init before the script created its box");
  int P[3];
  choose_procgrid(L, P);
  if (sf_brick_init(&L, L.comm_id, L.world_rank, L.world_size, P[0], P[1], P[2]) != 0)
    sf::fail("%s", sf::last_error().c_str());
  L.procgrid[0] = P[0]; L.procgrid[1] = P[1]; L.procgrid[2] = P[2];
  L.decomposed = true;
}

// rank that owns a point: brick (cx, cy, cz) with the bounds sf_brick_init computes, lo + c w (the last one ends at the
// box face); a point outside the box in a wall dimension belongs to the end brick, a periodic coordinate is wrapped
int brick_owner(const SfLammps& L, const double* x)
{
  double lo[3], hi[3];
  int per[3], c[3];
  L.eng.box(lo, hi, per);
  for (int k = 0; k < 3; k++) {
    const int Pk = L.procgrid[k];
    const double len = hi[k] - lo[k], w = len / Pk;
    double xk = x[k];
    if (per[k]) {
      while (xk < lo[k]) xk += len;
      while (xk >= hi[k]) xk -= len;
    }
    int ck = (int)std::floor((xk - lo[k]) / w);
    ck = ck < 0 ? 0 : (ck >= Pk ? Pk - 1 : ck);
    while (ck > 0 && xk < lo[k] + ck * w) ck--;
    while (ck < Pk - 1 && xk >= lo[k] + (ck + 1) * w) ck++;
    c[k] = ck;
  }
  return c[0] + L.procgrid[0] * (c[1] + L.procgrid[1] * c[2]);
}

// "run n pre no post no" on whatever the engine is: one domain, or the bricks of a -parallel run (library.cpp:372-386
// is collective there: Verlet::run with its forward communication and the reneighbouring vote)
void run_steps(SfLammps& L, int n)
{
  if (!L.decomposed && sf_slab_active(&L) != 1) {   // (a host may also have set the domains up itself: sf_slab_init)
    L.eng.run(n);
    return;
  }
  if (L.pending_rebuild) {
    // atoms were created / deleted since the last list (next_reneighbor = ntimestep + 1, library.cpp:482-486)
    L.pending_rebuild = false;
    if (L.eng.is_setup() && sf_slab_rebuild(&L) != 0) sf::fail("%s", sf::last_error().c_str());
  }
  if (sf_slab_step(&L, n) != 0) sf::fail("%s", sf::last_error().c_str());
}

// atom->natoms after atoms came or went: the sum of nlocal over the ranks (library.cpp:470-473)
void recount_atoms(SfLammps& L)
{
  double n = (double)L.eng.nlocal();
  if (L.decomposed && sf_slab_allreduce_sum(&L, &n, 1) != 0) sf::fail("%s", sf::last_error().c_str());
  L.natoms = (long long)n;
}

void read_data(SfLammps& L, const std::string& path)
{
  // [3P] read_data for atom_style sphere: "N atoms", "lo hi xlo xhi" ..., section "Atoms":
  // id type diameter density x y z   (e.g. cases/auto-testing/test-cases/xiaocase3/IC_uniform.in)
  std::ifstream f(path);
  if (!f) sf::fail("Cannot open file %s", path.c_str());
  std::string line;
  std::getline(f, line);  // title
  long natoms = -1;
  double lo[3] = {0, 0, 0}, hi[3] = {1, 1, 1};
  bool in_atoms = false;
  std::vector<double> x, diam, dens;
  std::vector<int> tag, type;
  while (std::getline(f, line)) {
    std::vector<std::string> w = split(line);
    if (w.empty()) continue;
    if (!in_atoms) {
      if (w.size() >= 2 && w[1] == "atoms") natoms = std::atol(w[0].c_str());
      else if (w.size() >= 4 && w[2] == "xlo") { lo[0] = num(w[0]); hi[0] = num(w[1]); }
      else if (w.size() >= 4 && w[2] == "ylo") { lo[1] = num(w[0]); hi[1] = num(w[1]); }
      else if (w.size() >= 4 && w[2] == "zlo") { lo[2] = num(w[0]); hi[2] = num(w[1]); }
      else if (w[0] == "Atoms") in_atoms = true;
      continue;
    }
    if (w[0] == "Velocities") break;
    if (w.size() < 7) sf::fail("Incorrect atom format in data file");
    tag.push_back(inum(w[0]));
    type.push_back(inum(w[1]));
    diam.push_back(num(w[2]));
    dens.push_back(num(w[3]));
    x.push_back(num(w[4]));
    x.push_back(num(w[5]));
    x.push_back(num(w[6]));
  }
  if (natoms >= 0 && (long)tag.size() != natoms) sf::fail("Did not assign all atoms correctly");
  L.eng.set_box(lo, hi);
  L.natoms = (long long)tag.size();
  if (L.world_size > 1) {
    // [3P] read_data on N ranks: every rank reads the file, the box is cut by the processor grid, a rank keeps the
    // atoms of its sub-domain (Comm::set_proc_grid, then sublo <= x < subhi in Atom::data_atoms)
    decompose(L);
    double dom[6];
    L.eng.sublo_hi(dom);
    size_t keep = 0;
    for (size_t i = 0; i < tag.size(); i++) {
      if (brick_owner(L, &x[3 * i]) != L.world_rank) continue;
      tag[keep] = tag[i];
      type[keep] = type[i];
      diam[keep] = diam[i];
      dens[keep] = dens[i];
      for (int k = 0; k < 3; k++) x[3 * keep + k] = x[3 * i + k];
      keep++;
    }
    tag.resize(keep);
  }
  L.eng.create_atoms((int)tag.size(), x.data(), nullptr, nullptr, diam.data(), dens.data(), tag.data(),
                     type.data());
}

void cmd_pair_style(SfLammps& L, const std::vector<std::string>& w, size_t a)
{
  if (a >= w.size()) sf::fail("Illegal pair_style command");
  const std::string& st = w[a];
  if (st == "hybrid/overlay" || st == "hybrid") {
    // sub-styles follow, each with its own arguments
    size_t k = a + 1;
    while (k < w.size()) {
      size_t e = k + 1;
      while (e < w.size() && w[e] != "gran/hertzFix/history" && w[e] != "gran/hooke/history" &&
             w[e] != "gran/hooke" && w[e] != "lubricate/poly")
        e++;
      std::vector<std::string> sub(w.begin() + k, w.begin() + e);
      sub.insert(sub.begin(), "pair_style");
      cmd_pair_style(L, sub, 1);
      k = e;
    }
    L.pair_hybrid = true;
    return;
  }
  if (st == "gran/hertzFix/history" || st == "gran/hooke/history" || st == "gran/hooke") {
    if (w.size() - a - 1 != 6) sf::fail("Illegal pair_style command");  // pair_gran_hertzFix_history.cpp:295
    const bool ktn = w[a + 2] == "NULL", gtn = w[a + 4] == "NULL";
    // 3: plain gran/hooke [3P] -- the style FixWallGranFix's HOOKE branch belongs to (fix_wall_granFix.cpp:219-220)
    L.eng.set_pair_gran(st == "gran/hertzFix/history" ? 2 : (st == "gran/hooke" ? 3 : 1), num(w[a + 1]), ktn, ktn ? 0.0 : num(w[a + 2]),
                        num(w[a + 3]), gtn, gtn ? 0.0 : num(w[a + 4]), num(w[a + 5]), inum(w[a + 6]));
    return;
  }
  if (st == "lubricate/poly") {
    // [3P] PairLubricate::settings: mu flaglog flagfld cutinner cutoff [flagHI flagVF]
    const size_t n = w.size() - a - 1;
    if (n != 5 && n != 7) sf::fail("Illegal pair_style command");
    int flagHI = 1, flagVF = 1;
    if (n == 7) {
      flagHI = inum(w[a + 6]);
      flagVF = inum(w[a + 7]);
    }
    L.eng.set_pair_lubricate(num(w[a + 1]), inum(w[a + 2]), inum(w[a + 3]), num(w[a + 4]), num(w[a + 5]), flagHI,
                             flagVF);
    return;
  }
  if (st == "none") return;
  sf::fail("Unknown pair style %s", st.c_str());
}

void cmd_fix(SfLammps& L, const std::vector<std::string>& w)
{
  if (w.size() < 4) sf::fail("Illegal fix command");
  const int gb = L.eng.group_bit(w[2]);   // "Could not find fix group ID" in LAMMPS
  const std::string& st = w[3];
  const int narg = (int)w.size() - 1;  // LAMMPS narg counts ID group style ...
  // LAMMPS runs the post_force fixes in script order.  The fused kernel adds gravity, fdrag and the walls in that
  // fixed order (sums only: order-free up to rounding); what matters is on which side of `fix freeze` a fix stands --
  // the reference's bed cases put `fix ywall all wall/gran` AFTER `fix 4 bottom freeze`, so the wall still pushes
  // frozen grains -- and the engine keeps that per fix (DemEngine::set_freeze and the setters called after it).
  if (st == "cohesive" && gb != 1)
    sf::fail("fix cohesive on a group other than `all` is not supported (the full-list evaluation would differ from "
             "the reference's half-list ownership of a pair, fix_cohesive.cpp:167)");
  if (st == "nve/sphere") {
    L.eng.set_nve_sphere(gb);
  } else if (st == "gravity") {
    // fix ID group gravity magnitude vector x y z
    if (narg < 8 || w[5] != "vector") sf::fail("Illegal fix gravity command (only `vector` style)");
    L.eng.set_gravity(num(w[4]), num(w[6]), num(w[7]), num(w[8]), gb);
  } else if (st == "fdrag") {
    if (narg < 3) sf::fail("Illegal fix fdrag command");  // fix_fluid_drag.cpp:34
    double carrier = 0.0;
    if (narg == 4) carrier = (double)std::atoi(w[4].c_str());  // integer parse, fix_fluid_drag.cpp:53
    L.eng.set_fdrag(carrier, gb);
  } else if (st == "cohesive") {
    if (narg != 8) sf::fail("Illegal fix cohesive command");  // fix_cohesive.cpp:41
    L.eng.set_cohesive(std::atof(w[4].c_str()), std::atof(w[5].c_str()), std::atof(w[6].c_str()),
                       std::atof(w[7].c_str()), std::atoi(w[8].c_str()), gb);
  } else if (st == "wall/gran" || st == "wall/granFix") {
    if (narg < 10) sf::fail("Illegal fix %s command", st.c_str());  // fix_wall_granFix.cpp:47
    const bool ktn = w[5] == "NULL", gtn = w[7] == "NULL";
    // wallstyle args, fix_wall_granFix.cpp:83-113: {x,y,z}plane lo hi | zcylinder radius
    int dim;
    if (w[10] == "xplane") dim = 0;
    else if (w[10] == "yplane") dim = 1;
    else if (w[10] == "zplane") dim = 2;
    else if (w[10] == "zcylinder") dim = 3;
    else sf::fail("Illegal fix %s command", st.c_str());
    size_t iarg;   // index in w of the first optional keyword
    if (dim < 3) {
      if (narg < 12) sf::fail("Illegal fix %s command", st.c_str());
      const bool lon = w[11] == "NULL", hin = w[12] == "NULL";
      L.eng.add_wall(dim, lon, lon ? 0.0 : num(w[11]), hin, hin ? 0.0 : num(w[12]), num(w[4]), ktn,
                     ktn ? 0.0 : num(w[5]), num(w[6]), gtn, gtn ? 0.0 : num(w[7]), num(w[8]), inum(w[9]),
                     st == "wall/granFix", gb);
      iarg = 13;
    } else {
      if (narg < 11) sf::fail("Illegal fix %s command", st.c_str());
      L.eng.add_wall(2, true, 0.0, true, 0.0, num(w[4]), ktn, ktn ? 0.0 : num(w[5]), num(w[6]), gtn,
                     gtn ? 0.0 : num(w[7]), num(w[8]), inum(w[9]), st == "wall/granFix", gb, /*z_periodic_ok=*/true);
      L.eng.wall_cylinder(num(w[11]));
      iarg = 12;
    }
    // optional keywords, :115-141: wiggle dim amplitude period | shear dim vshear
    auto axis_of = [&](const std::string& a) {
      if (a == "x") return 0;
      if (a == "y") return 1;
      if (a == "z") return 2;
      sf::fail("Illegal fix %s command", st.c_str());
      return 0;
    };
    while (iarg < w.size()) {
      if (w[iarg] == "wiggle") {
        if (iarg + 4 > w.size()) sf::fail("Illegal fix %s command", st.c_str());
        L.eng.wall_motion(1, axis_of(w[iarg + 1]), num(w[iarg + 2]), num(w[iarg + 3]));
        iarg += 4;
      } else if (w[iarg] == "shear") {
        if (iarg + 3 > w.size()) sf::fail("Illegal fix %s command", st.c_str());
        L.eng.wall_motion(2, axis_of(w[iarg + 1]), num(w[iarg + 2]), 0.0);
        iarg += 3;
      } else
        sf::fail("Illegal fix %s command", st.c_str());
    }
  } else if (st == "freeze") {
    if (narg != 3) sf::fail("Illegal fix freeze command");   // [3P] fix_freeze.cpp
    L.eng.set_freeze(gb);
  } else
    sf::fail("Unknown fix style %s", st.c_str());
}

// [3P] group ID style args: the styles the reference's input scripts use (type, subtract, union, intersect)
void cmd_group(SfLammps& L, const std::vector<std::string>& w)
{
  if (w.size() < 4) sf::fail("Illegal group command");
  const std::string& name = w[1];
  const std::string& style = w[2];
  if (style == "type") {
    static const char* ops[] = {"<", "<=", ">", ">=", "==", "!=", "<>"};
    int op = 0;
    for (int k = 0; k < 7; k++)
      if (w[3] == ops[k]) op = k + 1;
    if (op) {
      if (w.size() < (op == 7 ? 6u : 5u)) sf::fail("Illegal group command");
      L.eng.group_type(name, op, inum(w[4]), op == 7 ? inum(w[5]) : 0, {});
    } else {
      std::vector<int> list;
      for (size_t k = 3; k < w.size(); k++) list.push_back(inum(w[k]));
      L.eng.group_type(name, 0, 0, 0, list);
    }
  } else if (style == "subtract" || style == "union" || style == "intersect") {
    std::vector<std::string> args(w.begin() + 3, w.end());
    L.eng.group_combine(name, style == "subtract" ? 0 : (style == "union" ? 1 : 2), args);
  } else
    sf::fail("group style %s is not supported by this engine (type, subtract, union, intersect are)", style.c_str());
}

void command(SfLammps& L, const std::string& line)
{
  std::vector<std::string> w = split(line);
  if (w.empty()) return;
  const std::string& c = w[0];
  if (c == "units") {
    if (w.size() != 2 || (w[1] != "lj" && w[1] != "si")) sf::fail("units %s not supported (lj | si: nktv2p = 1)", w.size() > 1 ? w[1].c_str() : "");
  } else if (c == "atom_style") {
    if (w.size() < 2 || w[1] != "sphere") sf::fail("atom_style must be sphere");
  } else if (c == "newton") {
    if (w.size() < 2 || w[1] != "off") sf::fail("newton must be off (every reference case; pair lubricate/poly requires it)");
  } else if (c == "communicate") {
    // "communicate single vel yes": ghost velocities are always carried here
  } else if (c == "boundary") {
    if (w.size() != 4) sf::fail("Illegal boundary command");
    int p[3];
    for (int k = 0; k < 3; k++) p[k] = (w[k + 1] == "p" || w[k + 1] == "pp");
    if (L.decomposed) sf::fail("Boundary command after simulation box is defined");   // [3P] the text of Domain::set_boundary
    L.eng.set_periodic(p[0], p[1], p[2]);
  } else if (c == "read_data") {
    if (w.size() < 2) sf::fail("Illegal read_data command");
    read_data(L, w[1]);
  } else if (c == "neighbor") {
    if (w.size() < 2) sf::fail("Illegal neighbor command");
    L.eng.set_skin(num(w[1]));
  } else if (c == "neigh_modify") {
    // the engine decides like `delay 0 every 1 check yes` (every case of the reference): refuse anything else
    // instead of silently rebuilding at other times than LAMMPS would
    for (size_t k = 1; k + 1 < w.size(); k += 2) {
      if (w[k] == "one") L.eng.set_max_neigh(inum(w[k + 1]));
      else if (w[k] == "delay" && inum(w[k + 1]) != 0) sf::fail("neigh_modify delay %s: only delay 0 is supported", w[k + 1].c_str());
      else if (w[k] == "every" && inum(w[k + 1]) != 1) sf::fail("neigh_modify every %s: only every 1 is supported", w[k + 1].c_str());
      else if (w[k] == "check" && w[k + 1] != "yes") sf::fail("neigh_modify check %s: only check yes is supported", w[k + 1].c_str());
    }
  } else if (c == "pair_style") {
    cmd_pair_style(L, w, 1);
  } else if (c == "timestep") {
    if (w.size() != 2) sf::fail("Illegal timestep command");
    L.eng.set_timestep(num(w[1]));
  } else if (c == "velocity") {
    if (w.size() >= 6 && w[2] == "set") {
      for (int k = 3; k < 6; k++)
        if (w[k] == "NULL") sf::fail("velocity set: NULL components are not supported");
      L.eng.set_velocity_group(L.eng.group_bit(w[1]), num(w[3]), num(w[4]), num(w[5]));
    } else
      sf::fail("velocity: only `velocity GROUP set vx vy vz` is supported");
  } else if (c == "group") {
    cmd_group(L, w);
  } else if (c == "fix") {
    cmd_fix(L, w);
  } else if (c == "run") {
    if (w.size() < 2) sf::fail("Illegal run command");
    run_steps(L, inum(w[1]));
  } else if (c == "processors") {
    // [3P] processors px py pz (`*` = chosen by LAMMPS); must come before the box is created, like in LAMMPS
    if (w.size() < 4) sf::fail("Illegal processors command");
    if (L.decomposed) sf::fail("Processors command after simulation box is defined");
    for (int k = 0; k < 3; k++) {
      L.procgrid[k] = w[k + 1] == "*" ? 0 : inum(w[k + 1]);
      if (L.procgrid[k] < 0) sf::fail("Illegal processors command");
    }
    if (L.procgrid[0] && L.procgrid[1] && L.procgrid[2] &&
        L.procgrid[0] * L.procgrid[1] * L.procgrid[2] != L.world_size && L.world_size > 1)
      sf::fail("Specified processors != physical processors");   // [3P] Comm::set_proc_grid
  } else if (c == "pair_coeff" || c == "atom_modify" || c == "thermo" ||
             c == "thermo_style" || c == "thermo_modify" || c == "dump" || c == "dump_modify" ||
             c == "restart" || c == "echo" || c == "log" || c == "dimension") {
    // accepted, nothing to do on this path
  } else
    sf::fail("Unknown command: %s", c.c_str());
}

SfLammps* H(void* p)
{
  if (!p) sf::fail("null engine handle");
  return static_cast<SfLammps*>(p);
}

}  // namespace

extern "C" {

const char* sf_last_error(void) { return sf::last_error().c_str(); }
const char* sf_version(void) { return "sedifoam_amd 0.1 (gfx950)"; }

int sf_device_check(void)
{
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) {
    sf::set_error("no HIP device: %s", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return -1;
  }
  return 0;
}

int sf_lammps_open(int, char**, intptr_t comm, void** ptr)
{
  SF_API_BEGIN
  SfLammps* L = new SfLammps();
  L->comm = comm;
  *ptr = L;
  SF_API_END(0)
}

int sf_lammps_open_world(int, char**, intptr_t comm, int rank, int world, const char* id128, void** ptr)
{
  SF_API_BEGIN
  if (world < 1 || rank < 0 || rank >= world) sf::fail("sf_lammps_open_world: rank %d of %d", rank, world);
  if (world > 1 && !id128) sf::fail("sf_lammps_open_world: %d ranks need the communicator id of rank 0", world);
  // one process per GPU: choose the device BEFORE the engine creates its stream and buffers
  // (SF_DEVICE pins it; otherwise the NODE-LOCAL rank the launcher exports -- mpirun / srun / torchrun -- and only then the
  // global rank modulo the device count, which is right when ranks are placed node by node in blocks of the device count)
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 1 && world > 1) {
    int local = rank;
    for (const char* name : {"SF_DEVICE", "OMPI_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID", "MV2_COMM_WORLD_LOCAL_RANK",
                             "SLURM_LOCALID", "LOCAL_RANK"})
      if (const char* v = getenv(name)) {
        local = atoi(v);
        break;
      }
    SF_HIP(hipSetDevice(((local % ndev) + ndev) % ndev));
  }
  SfLammps* L = new SfLammps();
  L->comm = comm;
  L->world_rank = rank;
  L->world_size = world;
  if (id128) memcpy(L->comm_id, id128, 128);
  *ptr = L;
  SF_API_END(0)
}

// the grid `processors px py pz` resolves to on `world` ranks (0 = `*`), host logic only (no device): 0, or -1 when no
// factorisation fits ("Bad grid of processors")
int sf_procgrid_choose(int world, const double lo[3], const double hi[3], const int user[3], int out[3])
{
  SF_API_BEGIN
  if (world < 1 || !lo || !hi || !user || !out) sf::fail("sf_procgrid_choose: bad arguments");
  if (!procgrid_rule(world, lo, hi, user, out)) sf::fail("Bad grid of processors");
  SF_API_END(0)
}

int sf_lammps_close(void* ptr)
{
  SF_API_BEGIN
  delete H(ptr);
  SF_API_END(0)
}

const char* sf_lammps_command(void* ptr, const char* line)
{
  try {
    command(*H(ptr), line);
    return nullptr;
  } catch (const std::exception& ex) {
    sf::set_error("%s", ex.what());
    return sf::last_error().c_str();
  }
}

int sf_lammps_file(void* ptr, const char* path)
{
  SF_API_BEGIN
  std::ifstream f(path);
  if (!f) sf::fail("Cannot open input script %s", path);
  std::string line;
  while (std::getline(f, line)) command(*H(ptr), line);
  SF_API_END(0)
}

int sf_lammps_sync(void* ptr)
{
  SF_API_BEGIN
  SF_HIP(hipStreamSynchronize(H(ptr)->eng.stream()));
  SF_API_END(0)
}

int sf_lammps_get_global_n(void* ptr)
{
  SF_API_BEGIN
  // library.cpp:94-98: atom->natoms, the GLOBAL count (cached like LAMMPS' own; not a collective)
  SfLammps* L = H(ptr);
  if (L->decomposed && L->natoms < 0) recount_atoms(*L);   // (atoms handed over by sf_dem_create_atoms: counted once)
  const int n = L->decomposed ? (int)L->natoms : L->eng.nlocal();
  SF_API_END(n)
}

int sf_lammps_get_initial_np(void* ptr, int* np_)
{
  SF_API_BEGIN
  // library.cpp:112-131: every rank's nlocal in its slot, MPI_Allreduce(MPI_SUM) over the LAMMPS world
  SfLammps* L = H(ptr);
  DemEngine& e = L->eng;
  const int W = std::max(e.nranks(), L->world_size);
  std::vector<double> cnt(W, 0.0);
  cnt[L->decomposed ? L->world_rank : e.rank()] = (double)e.nlocal();
  if (L->decomposed && sf_slab_allreduce_sum(ptr, cnt.data(), W) != 0) sf::fail("%s", sf::last_error().c_str());
  for (int r = 0; r < W; r++) np_[r] = (int)cnt[r];
  SF_API_END(0)
}

int sf_lammps_get_initial_info(void* ptr, double* coords, double* velos, double* diam, double* rho_, int* tag_,
                               int* lmpCpuId_, int* type_)
{
  SF_API_BEGIN
  DemEngine& e = H(ptr)->eng;
  e.get_initial_info(coords, velos, diam, rho_, tag_, type_);
  if (lmpCpuId_)
    for (int i = 0; i < e.nlocal(); i++) lmpCpuId_[i] = e.rank();
  SF_API_END(0)
}

int sf_lammps_get_local_n(void* ptr)
{
  SF_API_BEGIN
  const int n = H(ptr)->eng.nlocal();
  SF_API_END(n)
}

int sf_lammps_get_local_domain(void* ptr, double* domain_)
{
  SF_API_BEGIN
  H(ptr)->eng.sublo_hi(domain_);
  SF_API_END(0)
}

int sf_lammps_get_local_info(void* ptr, double* coords, double* velos_, int* foamCpuId_, int* lmpCpuId_, int* tag_)
{
  SF_API_BEGIN
  sf::Range r("lammps->foam");   // writeCPUTime.H bucket
  DemEngine& e = H(ptr)->eng;
  e.get_local_info(coords, velos_, foamCpuId_, tag_);
  if (lmpCpuId_)
    for (int i = 0; i < e.nlocal(); i++) lmpCpuId_[i] = e.rank();
  SF_API_END(0)
}

int sf_lammps_put_local_info(void* ptr, int nLocalIn, const double* fdrag, const double* /*DuDt ignored*/,
                             const int* foamCpuIdIn, const int* tagIn)
{
  SF_API_BEGIN
  sf::Range r("foam->lammps");
  H(ptr)->eng.put_local_info(nLocalIn, fdrag, foamCpuIdIn, tagIn);
  SF_API_END(0)
}

int sf_lammps_step(void* ptr, int n)
{
  SF_API_BEGIN
  sf::Range r("lammps");
  run_steps(*H(ptr), n);
  SF_API_END(0)
}

int sf_lammps_set_timestep(void* ptr, double dt_i)
{
  SF_API_BEGIN
  H(ptr)->eng.set_timestep(dt_i);
  SF_API_END(0)
}

double sf_lammps_get_timestep(void* ptr)
{
  try {
    return H(ptr)->eng.timestep();
  } catch (const std::exception& ex) {
    sf::set_error("%s", ex.what());
    return -1.0;
  }
}

int sf_lammps_create_particle(void* ptr, int npAdd, const double* position, const double* tag, double diameter,
                              double rho, int type, const double* vel)
{
  SF_API_BEGIN
  SfLammps* L = H(ptr);
  L->eng.create_particles(npAdd, position, tag, diameter, rho, type, vel);
  if (L->decomposed) {   // library.cpp:470-473 (collective: every rank calls, possibly with npAdd = 0)
    recount_atoms(*L);
    L->pending_rebuild = true;
  }
  SF_API_END(0)
}

int sf_lammps_delete_particle(void* ptr, const int* deleteList, int nDelete)
{
  SF_API_BEGIN
  SfLammps* L = H(ptr);
  L->eng.delete_particles(deleteList, nDelete);
  if (L->decomposed) {   // library.cpp:527-537: collective counts; every rank deletes the listed atoms it owns
    recount_atoms(*L);
    L->pending_rebuild = true;
  }
  SF_API_END(0)
}

int sf_dem_create_atoms(void* ptr, int n, const double* x, const double* v, const double* omega,
                        const double* diameter, const double* density, const int* tag, const int* type)
{
  SF_API_BEGIN
  H(ptr)->eng.create_atoms(n, x, v, omega, diameter, density, tag, type);
  SF_API_END(0)
}

int sf_dem_set_box(void* ptr, const double lo[3], const double hi[3])
{
  SF_API_BEGIN
  H(ptr)->eng.set_box(lo, hi);
  SF_API_END(0)
}

int sf_dem_get_info(void* ptr, sf_dem_info* out)
{
  SF_API_BEGIN
  DemEngine& e = H(ptr)->eng;
  out->nlocal = e.nlocal();
  out->nghost = e.nghost();
  out->capacity = (int)e.capacity();
  out->max_neigh_used = e.max_neigh_used();
  out->max_neigh_cap = e.max_neigh_cap();
  out->nbuilds = e.nbuilds();
  out->nsteps = e.nsteps();
  out->npairs_full = e.npairs_full();
  SF_API_END(0)
}

int sf_dem_device_view_get(void* ptr, sf_dem_device_view* out)
{
  SF_API_BEGIN
  DemEngine& e = H(ptr)->eng;
  out->xr = (void*)e.d_xr();
  out->vm = (void*)e.d_vm();
  out->om = (void*)e.d_om();
  out->force = e.d_force();
  out->torque = e.d_torque();
  out->fdrag = e.d_fdrag();
  out->DuDt = e.d_DuDt();
  out->vOld = e.d_vOld();
  out->tag = e.d_tag();
  out->type = e.d_type();
  out->foamCpuId = e.d_foamCpuId();
  out->nlocal = e.nlocal();
  out->nghost = e.nghost();
  out->capacity = (int)e.capacity();
  out->stream = e.stream();
  SF_API_END(0)
}

int sf_dem_set_profiling(void* ptr, int on)
{
  SF_API_BEGIN
  H(ptr)->eng.set_profiling(on != 0);
  SF_API_END(0)
}

int sf_dem_get_profile(void* ptr, long long* launches, double* kernel_ms)
{
  SF_API_BEGIN
  H(ptr)->eng.get_profile(launches, kernel_ms);
  SF_API_END(0)
}

int sf_dem_get_rebuild_profile(void* ptr, long long* rebuilds, double* ms)
{
  SF_API_BEGIN
  H(ptr)->eng.get_rebuild_profile(rebuilds, ms);
  SF_API_END(0)
}

int sf_dem_get_forces(void* ptr, double* f, double* torque, double* omega, int* tag)
{
  SF_API_BEGIN
  H(ptr)->eng.get_forces(f, torque, omega, tag);
  SF_API_END(0)
}

long long sf_dem_get_history(void* ptr, long long max, int* tag_i, int* tag_j, double* shear)
{
  SF_API_BEGIN
  const long long n = H(ptr)->eng.get_history(max, tag_i, tag_j, shear);
  SF_API_END(n)
}

int sf_dem_get_wall_shear(void* ptr, int w, double* shear)
{
  SF_API_BEGIN
  H(ptr)->eng.get_wall_shear(w, shear);
  SF_API_END(0)
}

int sf_dem_set_subdomain(void* ptr, int rank, int nranks, double sublo, double subhi)
{
  SF_API_BEGIN
  H(ptr)->eng.set_subdomain(rank, nranks, sublo, subhi);
  SF_API_END(0)
}

int sf_dem_run_begin(void* ptr)
{
  SF_API_BEGIN
  H(ptr)->eng.run_begin();
  SF_API_END(0)
}

int sf_dem_substep(void* ptr, int last)
{
  SF_API_BEGIN
  H(ptr)->eng.substep(last != 0);
  SF_API_END(0)
}

int sf_dem_substep_k(void* ptr, int last, int kstep)
{
  SF_API_BEGIN
  H(ptr)->eng.substep_k(last != 0, kstep);
  SF_API_END(0)
}

int sf_dem_batch_end(void* ptr, int first_k, int launched, int* trigger)
{
  SF_API_BEGIN
  *trigger = H(ptr)->eng.batch_end(first_k, launched);
  SF_API_END(0)
}

int sf_dem_set_flag_buffer(void* ptr, void* dev_ints)
{
  SF_API_BEGIN
  H(ptr)->eng.set_flag_buffer((int*)dev_ints);
  SF_API_END(0)
}

int sf_dem_need_rebuild(void* ptr)
{
  SF_API_BEGIN
  const int r = H(ptr)->eng.need_rebuild() ? 1 : 0;
  SF_API_END(r)
}

int sf_dem_rebuild_begin(void* ptr)
{
  SF_API_BEGIN
  H(ptr)->eng.rebuild_begin();
  SF_API_END(0)
}

int sf_dem_rebuild_sort(void* ptr)
{
  SF_API_BEGIN
  H(ptr)->eng.rebuild_sort();
  SF_API_END(0)
}

int sf_dem_rebuild_finish(void* ptr)
{
  SF_API_BEGIN
  H(ptr)->eng.rebuild_finish();
  SF_API_END(0)
}

int sf_dem_setup(void* ptr)
{
  SF_API_BEGIN
  H(ptr)->eng.setup();
  SF_API_END(0)
}

long long sf_dem_border_pack(void* ptr, int side, double xshift, double* dev_buf, long long max_atoms)
{
  SF_API_BEGIN
  const long long n = H(ptr)->eng.border_pack(side, xshift, dev_buf, max_atoms);
  SF_API_END(n)
}

int sf_dem_border_unpack(void* ptr, int side, const double* dev_buf, long long natoms)
{
  SF_API_BEGIN
  H(ptr)->eng.border_unpack(side, dev_buf, natoms);
  SF_API_END(0)
}

long long sf_dem_forward_pack(void* ptr, int side, double xshift, double* dev_buf)
{
  SF_API_BEGIN
  const long long n = H(ptr)->eng.forward_pack(side, xshift, dev_buf);
  SF_API_END(n)
}

int sf_dem_forward_unpack(void* ptr, int side, const double* dev_buf, long long natoms)
{
  SF_API_BEGIN
  H(ptr)->eng.forward_unpack(side, dev_buf, natoms);
  SF_API_END(0)
}

int sf_dem_forward_pack2(void* ptr, double shift0, double* buf0, double shift1, double* buf1, long long* n0,
                         long long* n1)
{
  SF_API_BEGIN
  H(ptr)->eng.forward_pack2(shift0, buf0, shift1, buf1, n0, n1);
  SF_API_END(0)
}

int sf_dem_forward_unpack2(void* ptr, const double* buf0, long long n0, const double* buf1, long long n1)
{
  SF_API_BEGIN
  H(ptr)->eng.forward_unpack2(buf0, n0, buf1, n1);
  SF_API_END(0)
}

int sf_dem_forward_pack_fused(void* ptr, double shift0, long long off0, double shift1, long long off1,
                              const int* dev_hdr_off, int nhdr, double* dev_sendbuf)
{
  SF_API_BEGIN
  H(ptr)->eng.forward_pack_fused(shift0, off0, shift1, off1, dev_hdr_off, nhdr, dev_sendbuf);
  SF_API_END(0)
}

int sf_dem_forward_unpack_fused(void* ptr, const double* dev_recvbuf, long long off_from_left, long long n_from_left,
                                long long off_from_right, long long n_from_right, const int* dev_hdr_off, int nhdr,
                                int kstep)
{
  SF_API_BEGIN
  H(ptr)->eng.forward_unpack_fused(dev_recvbuf, off_from_left, n_from_left, off_from_right, n_from_right,
                                   dev_hdr_off, nhdr, kstep);
  SF_API_END(0)
}

int sf_dem_set_overlap(void* ptr, int on, void* comm_stream)
{
  SF_API_BEGIN
  H(ptr)->eng.set_overlap(on != 0, (hipStream_t)comm_stream);
  SF_API_END(0)
}

int sf_dem_partition_streams(void* ptr, int comm_cus_per_xcd, void** main_stream, void** comm_stream)
{
  SF_API_BEGIN
  hipStream_t m = nullptr, c = nullptr;
  H(ptr)->eng.make_partitioned_streams(comm_cus_per_xcd, &m, &c);
  *main_stream = m;
  *comm_stream = c;
  SF_API_END(0)
}

int sf_dem_overlap_begin(void* ptr)
{
  SF_API_BEGIN
  H(ptr)->eng.overlap_begin();
  SF_API_END(0)
}

int sf_dem_substep_part(void* ptr, int part, int last, int kstep)
{
  SF_API_BEGIN
  H(ptr)->eng.substep_part(part, last != 0, kstep);
  SF_API_END(0)
}

int sf_dem_substep_flip(void* ptr, int kstep)
{
  SF_API_BEGIN
  H(ptr)->eng.substep_flip(kstep);
  SF_API_END(0)
}

int sf_dem_overlap_batch_end(void* ptr, int first_k, int launched, int last_kstep, int* trigger)
{
  SF_API_BEGIN
  if (!trigger) sf::fail("sf_dem_overlap_batch_end: null output");
  *trigger = H(ptr)->eng.overlap_batch_end(first_k, launched, last_kstep);
  SF_API_END(0)
}

int sf_dem_boundary_count(void* ptr)
{
  SF_API_BEGIN
  return H(ptr)->eng.boundary_count();
  SF_API_END(-1)
}

long long sf_dem_migrate_pack(void* ptr, int side, double xshift, double* dev_buf, long long max_doubles)
{
  SF_API_BEGIN
  const long long n = H(ptr)->eng.migrate_pack(side, xshift, dev_buf, max_doubles);
  SF_API_END(n)
}

int sf_dem_local_particle_volume(void* ptr, double* volP)
{
  SF_API_BEGIN
  if (!volP) sf::fail("sf_dem_local_particle_volume: null argument");
  *volP = H(ptr)->eng.local_particle_volume();
  SF_API_END(0)
}

int sf_dem_set_global_particle_volume(void* ptr, double volP)
{
  SF_API_BEGIN
  H(ptr)->eng.set_global_particle_volume(volP);
  SF_API_END(0)
}

int sf_dem_local_max_radius(void* ptr, double* rmax)
{
  SF_API_BEGIN
  if (!rmax) sf::fail("sf_dem_local_max_radius: null argument");
  *rmax = H(ptr)->eng.local_max_radius();
  SF_API_END(0)
}

int sf_dem_set_global_max_radius(void* ptr, double rmax)
{
  SF_API_BEGIN
  H(ptr)->eng.set_global_max_radius(rmax);
  SF_API_END(0)
}

long long sf_dem_migrate_count(void* ptr)
{
  SF_API_BEGIN
  const long long n = H(ptr)->eng.migrate_count();
  SF_API_END(n)
}

int sf_dem_migrate_unpack(void* ptr, const double* dev_buf, long long ndoubles)
{
  SF_API_BEGIN
  H(ptr)->eng.migrate_unpack(dev_buf, ndoubles);
  SF_API_END(0)
}

int sf_dem_migrate_set_slots(void* ptr, int mrec)
{
  SF_API_BEGIN
  H(ptr)->eng.migrate_set_slots(mrec);
  SF_API_END(0)
}

int sf_dem_ghost_forward_local(void* ptr)
{
  SF_API_BEGIN
  H(ptr)->eng.ghost_forward_local();
  SF_API_END(0)
}

int sf_dem_set_stream(void* ptr, void* stream)
{
  SF_API_BEGIN
  H(ptr)->eng.set_stream((hipStream_t)stream);
  SF_API_END(0)
}

int sf_dem_migrate_record_doubles(void* ptr)
{
  SF_API_BEGIN
  const int n = H(ptr)->eng.migrate_record_doubles();
  SF_API_END(n)
}

}  // extern "C"
