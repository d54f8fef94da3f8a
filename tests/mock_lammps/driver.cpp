// driver.cpp — drives the PairANI adapter through the mock LAMMPS objects.  C entry points for ctypes.
#include <chrono>
#include <cstdio>
#include <string>
#include <vector>

#include "pair_ani.h"

using namespace LAMMPS_NS;

extern "C" void lammpsplugin_init(void* lmp, void* handle, void* regfunc);

namespace {
struct Session {
  LAMMPS lmp;
  Pair* pair = nullptr;
  std::vector<double*> xrows, frows;
  std::vector<double> x, f;
  std::vector<int> type, ilist, numneigh, jflat;
  std::vector<int*> firstneigh;
  NeighList list;
  std::string err;
};
lammpsplugin_t g_plugin;
bool g_registered = false;
void reg(lammpsplugin_t* p, void*) { g_plugin = *p; g_registered = true; }
}  // namespace

extern "C" {

void* mock_create(const char* unit_style, int newton_pair) {
  Session* s = new Session;
  s->lmp.update->unit_style = strdup(unit_style);
  s->lmp.force->newton_pair = newton_pair;
  s->lmp.force->newton = newton_pair;
  return s;
}
const char* mock_error(void* h) { return ((Session*)h)->err.c_str(); }
const char* mock_plugin_name() { return g_registered ? g_plugin.name : ""; }

// pair_style / pair_coeff / init_style through the plugin factory
int mock_pair_style(void* h, int narg, const char** args, int ntypes) {
  Session* s = (Session*)h;
  try {
    if (!g_registered) lammpsplugin_init(&s->lmp, nullptr, (void*)&reg);
    s->lmp.atom->ntypes = ntypes;
    s->pair = (Pair*)g_plugin.creator.v1(&s->lmp);
    std::vector<char*> a;
    for (int i = 0; i < narg; i++) a.push_back(strdup(args[i]));
    s->pair->settings(narg, a.data());
    char star[] = "*";
    char* cargs[2] = {star, star};
    s->pair->coeff(2, cargs);
    s->pair->init_style();
    return 0;
  } catch (const std::exception& e) {
    s->err = e.what();
    return 1;
  }
}
int mock_last_request(void* h) { return ((Session*)h)->lmp.neighbor->last_request; }
double mock_init_one(void* h) { return ((Session*)h)->pair->init_one(1, 1); }

// one Verlet-style force call: positions, types, full or half list (per-atom lists, possibly with special bits set)
int mock_compute(void* h, int nlocal, int nghost, const double* x, const int* type, const int* numneigh, const int* jflat,
                 const int* owner, int ago, int eflag, int vflag, double* f_out, double* eng_vdwl, double* virial6, double* eatom_out) {
  Session* s = (Session*)h;
  try {
    const int nt = nlocal + nghost;
    s->x.assign(x, x + 3 * (size_t)nt);
    s->f.assign(3 * (size_t)nt, 0.0);
    s->type.assign(type, type + nt);
    s->xrows.resize(nt);
    s->frows.resize(nt);
    for (int i = 0; i < nt; i++) { s->xrows[i] = s->x.data() + 3 * (size_t)i; s->frows[i] = s->f.data() + 3 * (size_t)i; }
    Atom* a = s->lmp.atom;
    a->x = s->xrows.data(); a->f = s->frows.data(); a->type = s->type.data(); a->nlocal = nlocal; a->nghost = nghost;
    a->nmax = nt;
    if (ago == 0) {
      s->numneigh.assign(numneigh, numneigh + nlocal);
      size_t tot = 0;
      for (int i = 0; i < nlocal; i++) tot += numneigh[i];
      s->jflat.assign(jflat, jflat + tot);
      // exercise NEIGHMASK: LAMMPS stores special-bond bits in the top bits of neighbour indices
      for (size_t p = 0; p < tot; p += 7) s->jflat[p] |= (1 << 30);
      s->ilist.resize(nlocal);
      s->firstneigh.resize(nlocal);
      size_t off = 0;
      for (int i = 0; i < nlocal; i++) { s->ilist[i] = i; s->firstneigh[i] = s->jflat.data() + off; off += numneigh[i]; }
      s->list.inum = nlocal; s->list.ilist = s->ilist.data(); s->list.numneigh = s->numneigh.data(); s->list.firstneigh = s->firstneigh.data();
      s->pair->list = &s->list;
      s->lmp.comm->owner.assign(owner, owner + nghost);
      s->lmp.comm->nlocal = nlocal;
      // what Domain / Comm hold in a real run: the sub-box of the owned atoms and the ghost cutoff around it
      for (int k = 0; k < 3; k++) {
        double lo = nlocal > 0 ? x[k] : 0.0, hi = lo, glo = lo, ghi = lo;
        for (int i = 0; i < nt; i++) {
          const double v = x[3 * (size_t)i + k];
          if (i < nlocal) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
          glo = v < glo ? v : glo; ghi = v > ghi ? v : ghi;
        }
        s->lmp.domain->sublo[k] = lo; s->lmp.domain->subhi[k] = hi;
        const double ext = (lo - glo) > (ghi - hi) ? (lo - glo) : (ghi - hi);
        s->lmp.comm->cutghost[k] = ext > 0.0 ? ext : 0.0;
      }
    }
    s->lmp.neighbor->ago = ago;
    s->pair->compute(eflag, vflag);
    for (size_t i = 0; i < 3 * (size_t)nt; i++) f_out[i] = s->f[i];
    *eng_vdwl = s->pair->eng_vdwl;
    for (int k = 0; k < 6; k++) virial6[k] = s->pair->virial[k];
    if ((eflag & 2) && eatom_out) for (int i = 0; i < nlocal; i++) eatom_out[i] = s->pair->eatom[i];
    return 0;
  } catch (const std::exception& e) {
    s->err = e.what();
    return 1;
  }
}

// The adapter's own cost per timestep, as LAMMPS' Verlet loop would see it: nsteps calls of PairANI::compute on the session's
// PERSISTENT arrays (the state mock_compute left: call it once with ago = 0 first), positions nudged in place between calls
// (a deterministic jitter of 1e-4 A: the cached list stays valid), a re-neighbouring call (ago = 0, same list handed over again)
// every `every` steps, eflag = vflag = 0 as on a step without thermo output.  Only the compute() calls are timed.
// out_ms[0] = mean plain step, out_ms[1] = mean re-neighbouring step (0 if none), out_ms[2] = checksum of f (keeps the work alive)
int mock_md_loop(void* h, int nsteps, int every, double* out_ms) {
  Session* s = (Session*)h;
  try {
    Atom* a = s->lmp.atom;
    const int nt = a->nlocal + a->nghost;
    if (!s->pair || nt == 0 || (int)s->x.size() != 3 * nt) { s->err = "mock_md_loop: call mock_compute with ago = 0 first"; return 1; }
    double t_plain = 0.0, t_reb = 0.0, sum = 0.0;
    int n_plain = 0, n_reb = 0;
    for (int k = 1; k <= nsteps; k++) {
      const double d = 1e-4 * ((k & 1) ? 1.0 : -1.0);
      for (int i = 0; i < a->nlocal; i++) s->x[3 * (size_t)i + (k % 3)] += d;
      // ghosts follow their owners (the forward communication of a real run)
      for (int g = 0; g < a->nghost; g++) s->x[3 * (size_t)(a->nlocal + g) + (k % 3)] += d;
      for (size_t i = 0; i < s->f.size(); i++) s->f[i] = 0.0;   // force_clear
      const bool reb = every > 0 && k % every == 0;
      s->lmp.neighbor->ago = reb ? 0 : (every > 0 ? k % every : k);
      const auto t0 = std::chrono::steady_clock::now();
      s->pair->compute(0, 0);
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (reb) { t_reb += ms; n_reb++; } else { t_plain += ms; n_plain++; }
      sum += s->f[0] + s->f[s->f.size() / 2];
    }
    out_ms[0] = n_plain ? t_plain / n_plain : 0.0;
    out_ms[1] = n_reb ? t_reb / n_reb : 0.0;
    out_ms[2] = sum;
    return 0;
  } catch (const std::exception& e) {
    s->err = e.what();
    return 1;
  }
}

int mock_restart_roundtrip(void* h, const char* path) {
  Session* s = (Session*)h;
  try {
    FILE* fp = fopen(path, "wb");
    s->pair->write_restart(fp);
    fclose(fp);
    fp = fopen(path, "rb");
    s->pair->read_restart(fp);
    fclose(fp);
    s->pair->list = &s->list;
    return 0;
  } catch (const std::exception& e) {
    s->err = e.what();
    return 1;
  }
}

void mock_destroy(void* h) {
  Session* s = (Session*)h;
  delete s->pair;
  delete s;
}
}
