// adapter_host.cpp -- runs the LAMMPS style adapters of adapters/lammps ONCE, the way LAMMPS would: it builds the objects the
// adapters touch (Atom, NeighList + its granular-history list, Update, Force, Neighbor, a fix rigid) from a case file,
// instantiates PairGranHertzFixHistoryAmd / FixCoheAmd / FixFluidDragAmd / PairLubricatePolyAmd through their public
// constructors, calls settings() / init() / init_list() / compute() / post_force() in LAMMPS' order and writes the per-atom
// results.  The declarations are tests/adapters/lammps_min (LAMMPS 1Feb14 is [3P] and not installed here); this file adds
// the few out-of-line members those declarations leave open.  Test infrastructure: tests/test_adapters_gpu.py compares the
// output with tests/golden/reference_pins.json (numbers computed by the reference's own lines).
//
//   adapter_host <kind> <case file> <output file>       kind = pair_gran | fix_cohesive | fix_fdrag | pair_lubricate
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "atom.h"
#include "comm.h"
#include "error.h"
#include "fix.h"
#include "force.h"
#include "memory.h"
#include "neigh_list.h"
#include "neigh_request.h"
#include "neighbor.h"
#include "update.h"
#include "pair_gran_hertzFix_history_amd.h"
#include "pair_lubricate_poly_amd.h"
#include "fix_cohesive_amd.h"
#include "fix_fluid_drag_amd.h"

using namespace LAMMPS_NS;

// ---------------------------------------------------------------- what the minimal declarations leave undefined
void Error::all(const char *file, int line, const char *msg)
{
  std::fprintf(stderr, "ERROR: %s (%s:%d)\n", msg, file, line);
  std::exit(3);
}
void Error::one(const char *file, int line, const char *msg)
{
  std::fprintf(stderr, "ERROR on proc 0: %s (%s:%d)\n", msg, file, line);
  std::exit(3);
}
double Force::numeric(const char *, int, char *s) { return std::atof(s); }
int Force::inumeric(const char *, int, char *s) { return std::atoi(s); }
void Atom::add_callback(int) {}
void Atom::delete_callback(const char *, int) {}
// [3P] Comm::forward_comm_pair: the pair style's per-atom values of the owned atoms travel to their ghost copies (the stock
// PairGranHookeHistory packs mass_rigid).  One rank here: the ghosts of the case carry their owners' values in the case file.
static int g_forward_calls = 0;
static void (*g_forward_fill)(Pair *) = NULL;
void Comm::forward_comm_pair(Pair *p)
{
  g_forward_calls++;
  if (g_forward_fill) g_forward_fill(p);
}

static LAMMPS g_lmp;   // (static storage: every pointer starts NULL; Pointers keeps REFERENCES to these fields)

int Neighbor::request(void *)
{
  static std::vector<NeighRequest *> all;
  NeighRequest *r = new NeighRequest(&g_lmp);
  r->pair = 1;
  r->fix = r->half = r->full = r->gran = r->granhistory = 0;
  all.push_back(r);
  requests = &all[0];
  return (int)all.size() - 1;
}
void Pair::ev_setup(int, int) { evflag = vflag_fdotr = 0; }
void PairGranHookeHistory::compute(int, int) {}
void PairGranHookeHistory::settings(int, char **) {}
void PairLubricate::compute(int, int) {}
void PairLubricate::settings(int, char **) {}
void PairLubricatePoly::compute(int, int) {}

// [3P] Memory: 2-d arrays are ONE block with row pointers (array[0] is the data), grow keeps the contents
template <typename T> T *Memory::create(T *&array, int n, const char *)
{
  array = (T *)std::malloc(sizeof(T) * (size_t)(n > 0 ? n : 1));
  return array;
}
template <typename T> T *Memory::grow(T *&array, int n, const char *)
{
  array = (T *)std::realloc(array, sizeof(T) * (size_t)(n > 0 ? n : 1));
  return array;
}
template <typename T> T **Memory::grow(T **&array, int n1, int n2, const char *)
{
  if (n1 < 1) n1 = 1;
  T *data = (T *)std::realloc(array ? array[0] : NULL, sizeof(T) * (size_t)n1 * n2);
  array = (T **)std::realloc(array, sizeof(T *) * (size_t)n1);
  for (int i = 0; i < n1; i++) array[i] = data + (size_t)i * n2;
  return array;
}
template <typename T> void Memory::destroy(T *&array)
{
  std::free(array);
  array = NULL;
}
template <typename T> void Memory::destroy(T **&array)
{
  if (array) std::free(array[0]);
  std::free(array);
  array = NULL;
}
template double *Memory::create<double>(double *&, int, const char *);
template int *Memory::grow<int>(int *&, int, const char *);
template double *Memory::grow<double>(double *&, int, const char *);
template double **Memory::grow<double>(double **&, int, int, const char *);
template void Memory::destroy<double>(double *&);
template void Memory::destroy<int>(int *&);
template void Memory::destroy<double>(double **&);

// ---------------------------------------------------------------- the case file:  name count v v v ...
typedef std::map<std::string, std::vector<double> > CaseMap;
static CaseMap read_case(const char *path)
{
  CaseMap c;
  std::FILE *f = std::fopen(path, "r");
  if (!f) { std::fprintf(stderr, "cannot open %s\n", path); std::exit(2); }
  char name[128];
  long n;
  while (std::fscanf(f, "%127s %ld", name, &n) == 2) {
    std::vector<double> &v = c[name];
    v.resize((size_t)n);
    char tok[64];
    for (long k = 0; k < n; k++) {
      if (std::fscanf(f, "%63s", tok) != 1) { std::fprintf(stderr, "short array %s\n", name); std::exit(2); }
      v[(size_t)k] = std::strtod(tok, NULL);
    }
  }
  std::fclose(f);
  return c;
}
static const std::vector<double> &arr(const CaseMap &c, const char *k)
{
  CaseMap::const_iterator it = c.find(k);
  if (it == c.end()) { std::fprintf(stderr, "case has no %s\n", k); std::exit(2); }
  return it->second;
}
static double num(const CaseMap &c, const char *k) { return arr(c, k)[0]; }
static bool has(const CaseMap &c, const char *k) { return c.find(k) != c.end(); }
static void put(std::FILE *f, const char *name, const double *v, size_t n)
{
  std::fprintf(f, "%s %lu", name, (unsigned long)n);
  for (size_t k = 0; k < n; k++) std::fprintf(f, " %a", v[k]);
  std::fprintf(f, "\n");
}
static void put_int(std::FILE *f, const char *name, const int *v, size_t n)
{
  std::fprintf(f, "%s %lu", name, (unsigned long)n);
  for (size_t k = 0; k < n; k++) std::fprintf(f, " %d", v[k]);
  std::fprintf(f, "\n");
}

// ---------------------------------------------------------------- the LAMMPS-side objects
static double **rows3(const CaseMap &c, const char *k, int n)   // n x 3, zero when the case has no such array
{
  double **a = NULL;
  g_lmp.memory->grow(a, n, 3, k);
  std::memset(a[0], 0, sizeof(double) * 3 * (size_t)n);
  if (has(c, k)) std::memcpy(a[0], &arr(c, k)[0], sizeof(double) * arr(c, k).size());   // (the case holds n <= nmax atoms)
  return a;
}
static double *col(const CaseMap &c, const char *k, int n)
{
  double *a = (double *)std::calloc((size_t)(n > 0 ? n : 1), sizeof(double));
  if (has(c, k)) std::memcpy(a, &arr(c, k)[0], sizeof(double) * arr(c, k).size());
  return a;
}
static int *icol(const CaseMap &c, const char *k, int n)
{
  int *a = (int *)std::calloc((size_t)(n > 0 ? n : 1), sizeof(int));
  if (has(c, k)) for (size_t i = 0; i < arr(c, k).size(); i++) a[i] = (int)arr(c, k)[i];
  return a;
}
static void make_world(const CaseMap &c)
{
  g_lmp.memory = new Memory(&g_lmp);
  g_lmp.error = new Error(&g_lmp);
  g_lmp.atom = new Atom(&g_lmp);
  g_lmp.update = new Update(&g_lmp);
  g_lmp.force = new Force(&g_lmp);
  g_lmp.neighbor = new Neighbor(&g_lmp);
  g_lmp.comm = new Comm(&g_lmp);
  const int n = (int)num(c, "n"), nlocal = has(c, "nlocal") ? (int)num(c, "nlocal") : n;
  Atom *a = g_lmp.atom;
  a->nlocal = nlocal;
  a->nghost = n - nlocal;
  a->nmax = n + 8;
  a->x = rows3(c, "x", a->nmax);
  a->v = rows3(c, "v", a->nmax);
  a->omega = rows3(c, "omega", a->nmax);
  a->f = rows3(c, "f", a->nmax);
  a->torque = rows3(c, "torque", a->nmax);
  a->radius = col(c, "radius", a->nmax);
  a->rmass = col(c, "rmass", a->nmax);
  a->mask = icol(c, "mask", a->nmax);
  a->type = icol(c, "type", a->nmax);
  a->tag = icol(c, "tag", a->nmax);
  g_lmp.update->dt = has(c, "dt") ? num(c, "dt") : 0.0;
  g_lmp.update->ntimestep = 1;
  g_lmp.update->setupflag = has(c, "shearupdate") ? (num(c, "shearupdate") != 0.0 ? 0 : 1) : 0;
  static char verlet[] = "verlet";
  g_lmp.update->integrate_style = verlet;
  g_lmp.force->nktv2p = 1.0;
  g_lmp.force->vxmu2f = 1.0;
  g_lmp.force->newton_pair = has(c, "newton_pair") ? (int)num(c, "newton_pair") : 0;
  g_lmp.neighbor->ago = 0;   // (a list built this step)
  g_lmp.neighbor->requests = NULL;
}
// NeighList pages from numneigh + the concatenated rows (ilist = 0 .. inum-1, as the pins' loops ran)
struct Pages {
  std::vector<int> ilist, numneigh, jl;
  std::vector<int *> first;
  std::vector<double> dbl;
  std::vector<double *> firstd;
};
static NeighList *make_list(const CaseMap &c, const char *rows_key, Pages &P, int inum, const char *dbl_key = NULL)
{
  NeighList *L = new NeighList(&g_lmp);
  const std::vector<double> &nn = arr(c, "numneigh"), &rows = arr(c, rows_key);
  P.ilist.resize((size_t)inum);
  P.numneigh.resize((size_t)inum);
  P.jl.resize(rows.size() + 1);
  for (size_t k = 0; k < rows.size(); k++) P.jl[k] = (int)rows[k];
  P.first.resize((size_t)inum);
  P.firstd.resize((size_t)inum);
  if (dbl_key) P.dbl = arr(c, dbl_key);
  P.dbl.resize(3 * rows.size() + 3);
  size_t o = 0;
  for (int i = 0; i < inum; i++) {
    P.ilist[(size_t)i] = i;
    P.numneigh[(size_t)i] = (int)nn[(size_t)i];
    P.first[(size_t)i] = &P.jl[o];
    P.firstd[(size_t)i] = &P.dbl[3 * o];
    o += (size_t)nn[(size_t)i];
  }
  L->inum = inum;
  L->ilist = &P.ilist[0];
  L->numneigh = &P.numneigh[0];
  L->firstneigh = &P.first[0];
  L->firstdouble = &P.firstd[0];
  L->listgranhistory = NULL;
  return L;
}

// a fix rigid as the pair style sees one ([3P] fix_rigid.cpp extract("body") / ("masstotal"))
class HostFixRigid : public Fix {
 public:
  std::vector<int> body;
  std::vector<double> masstotal;
  HostFixRigid(LAMMPS *l) : Fix(l, 0, NULL) {}
  int setmask() { return 0; }
  void *extract(const char *what, int &dim)
  {
    dim = 1;
    if (std::strcmp(what, "body") == 0) return &body[0];
    if (std::strcmp(what, "masstotal") == 0) return &masstotal[0];
    return NULL;
  }
};
// what Pair::init_style / init_list of the stock base classes hand to compute()
class HostPairGran : public PairGranHertzFixHistoryAmd {
 public:
  HostPairGran(LAMMPS *l) : PairGranHertzFixHistoryAmd(l) {}
  void wire(NeighList *l, NeighList *h, double dt_, int freeze_bit, Fix *rigid)
  {
    list = l;
    listgranhistory = h;
    dt = dt_;
    freeze_group_bit = freeze_bit;
    fix_rigid = rigid;
    mass_rigid = NULL;
    nmax = 0;
  }
  const std::vector<double> *ghost_src;   // mass_rigid of every atom as the case holds it (ghost entries used)
  static void fill_ghosts(Pair *p)
  {
    HostPairGran *me = static_cast<HostPairGran *>(p);
    for (int i = g_lmp.atom->nlocal; i < g_lmp.atom->nlocal + g_lmp.atom->nghost; i++)
      me->mass_rigid[i] = (*me->ghost_src)[(size_t)i];
  }
};
class HostPairLub : public PairLubricatePolyAmd {
 public:
  HostPairLub(LAMMPS *l) : PairLubricatePolyAmd(l) {}
  void wire(NeighList *l, const CaseMap &c)   // (the stock settings() / init_style() results, pair_lubricate_poly.cpp:450-577)
  {
    list = l;
    mu = num(c, "mu");
    flaglog = (int)num(c, "flaglog");
    flagfld = (int)num(c, "flagfld");
    flagHI = (int)num(c, "flagHI");
    flagVF = (int)num(c, "flagVF");
    cut_inner_global = num(c, "cut_inner");
    cut_global = num(c, "cut_global");
    R0 = num(c, "R0");
    RT0 = num(c, "RT0");
    RS0 = num(c, "RS0");
  }
};

static std::vector<std::string> g_words;
static char **argv_of(const std::vector<std::string> &w)
{
  static std::vector<char *> p;
  g_words = w;
  p.clear();
  for (size_t k = 0; k < g_words.size(); k++) p.push_back(&g_words[k][0]);
  return &p[0];
}
static std::string g17(double v)
{
  char b[64];
  std::snprintf(b, sizeof b, "%.17g", v);
  return b;
}

int main(int argc, char **argv)
{
  if (argc != 4) { std::fprintf(stderr, "usage: adapter_host <kind> <case> <out>\n"); return 2; }
  const std::string kind = argv[1];
  const CaseMap c = read_case(argv[2]);
  make_world(c);
  Atom *a = g_lmp.atom;
  const int nlocal = a->nlocal, nall = nlocal + a->nghost;
  std::FILE *out = std::fopen(argv[3], "w");
  if (!out) return 2;

  if (kind == "pair_gran") {
    Pages PL, PH;
    NeighList *list = make_list(c, "jlist", PL, nlocal);
    NeighList *hist = make_list(c, "touch", PH, nlocal, "shear");   // FixShearHistory's pages: touch words + 3 doubles
    HostFixRigid *rigid = NULL;
    if (has(c, "mass_rigid")) {   // every atom with a body mass is a body of its own
      rigid = new HostFixRigid(&g_lmp);
      const std::vector<double> &mr = arr(c, "mass_rigid");
      rigid->body.assign((size_t)nall, -1);
      rigid->masstotal.assign(1, 0.0);
      for (int i = 0; i < nall; i++)
        if (mr[(size_t)i] != 0.0) {
          rigid->body[(size_t)i] = (int)rigid->masstotal.size();
          rigid->masstotal.push_back(mr[(size_t)i]);
        }
    }
    HostPairGran *pair = new HostPairGran(&g_lmp);
    std::vector<std::string> w;   // pair_style gran/hertzFix/history kn kt gamman NULL xmu 1
    w.push_back(g17(num(c, "kn"))); w.push_back(g17(num(c, "kt"))); w.push_back(g17(num(c, "gamman")));
    w.push_back("NULL"); w.push_back(g17(num(c, "xmu"))); w.push_back("1");
    pair->settings(6, argv_of(w));
    pair->wire(list, hist, num(c, "dt"), (int)num(c, "freeze_group_bit"), rigid);
    if (rigid) {
      pair->ghost_src = &arr(c, "mass_rigid");
      g_forward_fill = HostPairGran::fill_ghosts;
    }
    pair->compute(0, 0);
    put(out, "f", a->f[0], 3 * (size_t)nall);
    put(out, "torque", a->torque[0], 3 * (size_t)nall);
    put_int(out, "touch", &PH.jl[0], PH.jl.size() - 1);
    put(out, "shear", &PH.dbl[0], 3 * (PH.jl.size() - 1));
    const int fw[1] = {g_forward_calls};
    put_int(out, "forward_comm_pair_calls", fw, 1);
    delete pair;
  } else if (kind == "fix_cohesive") {
    Pages PL;
    NeighList *list = make_list(c, "jlist", PL, nlocal);
    std::vector<std::string> w;   // fix ID group cohesive ah lam smin smax opt
    w.push_back("coh"); w.push_back("grp"); w.push_back("cohesive");
    w.push_back(g17(num(c, "ah"))); w.push_back(g17(num(c, "lam"))); w.push_back(g17(num(c, "smin")));
    w.push_back(g17(num(c, "smax")));
    char b[16];
    std::snprintf(b, sizeof b, "%d", (int)num(c, "opt"));
    w.push_back(b);
    FixCoheAmd *fix = new FixCoheAmd(&g_lmp, 8, argv_of(w));
    fix->groupbit = (int)num(c, "groupbit");
    if (fix->setmask() != (FixConst::POST_FORCE | FixConst::POST_FORCE_RESPA | FixConst::MIN_POST_FORCE)) return 4;
    fix->init();
    if (g_lmp.neighbor->requests[0]->pair != 0 || g_lmp.neighbor->requests[0]->fix != 1) return 4;   // fix_cohesive.cpp:75-77
    fix->init_list(0, list);
    fix->post_force(0);
    put(out, "f", a->f[0], 3 * (size_t)nall);
    delete fix;
  } else if (kind == "fix_fdrag") {
    std::vector<std::string> w;   // fix ID group fdrag [carrier_rho]
    w.push_back("fd"); w.push_back("grp"); w.push_back("fdrag");
    if (num(c, "carrier_rho") != 0.0) {
      char b[32];
      std::snprintf(b, sizeof b, "%d", (int)num(c, "carrier_rho"));
      w.push_back(b);
    }
    FixFluidDragAmd *fix = new FixFluidDragAmd(&g_lmp, (int)w.size(), argv_of(w));
    static char id[] = "fd";
    fix->id = id;
    fix->groupbit = (int)num(c, "groupbit");
    if (fix->setmask() != FixConst::POST_FORCE) return 4;
    fix->init();   // zeroes the arrays of the group's atoms (fix_fluid_drag.cpp:85-104) ...
    std::memcpy(fix->ffluiddrag[0], &arr(c, "ffluiddrag")[0], sizeof(double) * 3 * (size_t)nlocal);   // ... lammps_put_local_info
    std::memcpy(fix->DuDt[0], &arr(c, "DuDt")[0], sizeof(double) * 3 * (size_t)nlocal);
    std::memcpy(fix->vOld[0], &arr(c, "vOld")[0], sizeof(double) * 3 * (size_t)nlocal);
    fix->post_force(0);
    put(out, "f", a->f[0], 3 * (size_t)nlocal);
    put(out, "vOld", fix->vOld[0], 3 * (size_t)nlocal);
    // migration payload round trip (fix_fluid_drag.cpp:211-243): atom 0 -> slot nlocal
    double buf[16];
    const int m = fix->pack_exchange(0, buf);
    const int m2 = fix->unpack_exchange(nlocal, buf);
    const int ok[1] = {m == 10 && m2 == 10 && fix->vOld[nlocal][2] == fix->vOld[0][2] &&
                       fix->ffluiddrag[nlocal][0] == fix->ffluiddrag[0][0]};
    put_int(out, "exchange_ok", ok, 1);
    delete fix;
  } else if (kind == "pair_lubricate") {
    Pages PL;
    NeighList *list = make_list(c, "jlist", PL, nlocal);
    HostPairLub *pair = new HostPairLub(&g_lmp);
    pair->wire(list, c);
    pair->compute(0, 0);
    put(out, "f", a->f[0], 3 * (size_t)nall);
    put(out, "torque", a->torque[0], 3 * (size_t)nall);
    delete pair;
  } else {
    std::fprintf(stderr, "unknown kind %s\n", kind.c_str());
    return 2;
  }
  std::fclose(out);
  return 0;
}
