// lammps_mock.h — the smallest stand-in for the LAMMPS headers that lammps-ani_amd/csrc/pair_ani.cpp includes
// (atom.h comm.h error.h force.h memory.h neigh_list.h neighbor.h update.h pair.h lammpsplugin.h version.h mpi.h).
// TEST INFRASTRUCTURE: it lets the adapter be compiled and driven without LAMMPS (external/lammps is an empty
// submodule of the reference).  Member names and call signatures follow LAMMPS (stable release 2Aug2023).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

// ---- mpi.h -------------------------------------------------------------------------------------------
typedef int MPI_Comm;
typedef int MPI_Info;
#define MPI_COMM_WORLD 0
#define MPI_COMM_TYPE_SHARED 1
#define MPI_INFO_NULL 0
inline int MPI_Comm_rank(MPI_Comm, int* r) { *r = 0; return 0; }
inline int MPI_Comm_split_type(MPI_Comm, int, int, MPI_Info, MPI_Comm* c) { *c = 1; return 0; }
inline int MPI_Comm_free(MPI_Comm*) { return 0; }
// one rank: the collectives the adapter's rcclcomm bootstrap uses are copies
typedef int MPI_Datatype;
#define MPI_BYTE 1
#define MPI_INT 4
inline int MPI_Comm_size(MPI_Comm, int* n) { *n = 1; return 0; }
inline int MPI_Bcast(void*, int, MPI_Datatype, int, MPI_Comm) { return 0; }
inline int MPI_Alltoall(const void* s, int n, MPI_Datatype t, void* r, int, MPI_Datatype, MPI_Comm) { memcpy(r, s, (size_t)n * t); return 0; }
inline int MPI_Alltoallv(const void* s, const int* sn, const int* sd, MPI_Datatype t, void* r, const int*, const int* rd, MPI_Datatype, MPI_Comm) {
  memcpy((char*)r + (size_t)rd[0] * t, (const char*)s + (size_t)sd[0] * t, (size_t)sn[0] * t);
  return 0;
}

#define FLERR __FILE__, __LINE__
#define NEIGHMASK 0x1FFFFFFF
#define LAMMPS_VERSION "mock"

namespace LAMMPS_NS {

class LAMMPS;
class Pair;

class Error {
 public:
  [[noreturn]] void all(const char* f, int l, const std::string& m) { throw std::runtime_error(std::string(f) + ":" + std::to_string(l) + " " + m); }
  [[noreturn]] void one(const char* f, int l, const std::string& m) { all(f, l, m); }
};

class Memory {
 public:
  template <typename T>
  T** create(T**& a, int n1, int n2, const char*) {
    T* data = (T*)malloc(sizeof(T) * n1 * n2);
    a = (T**)malloc(sizeof(T*) * n1);
    for (int i = 0; i < n1; i++) a[i] = data + (size_t)i * n2;
    return a;
  }
  template <typename T>
  void destroy(T**& a) {
    if (!a) return;
    free(a[0]);
    free(a);
    a = nullptr;
  }
};

class Atom {
 public:
  double **x = nullptr, **f = nullptr;
  int* type = nullptr;
  int nlocal = 0, nghost = 0, ntypes = 0;
  int nmax = 0;   // rows allocated in x / f
};

class Force {
 public:
  int newton = 0, newton_pair = 0;
};

class Update {
 public:
  const char* unit_style = "real";
};

class NeighList {
 public:
  int inum = 0;
  int* ilist = nullptr;
  int* numneigh = nullptr;
  int** firstneigh = nullptr;
};

namespace NeighConst { enum { REQ_DEFAULT = 0, REQ_FULL = 1 << 0, REQ_OCCASIONAL = 1 << 4 }; }

class Neighbor {
 public:
  int ago = 0;
  double skin = 2.0;
  int last_request = -1;
  void add_request(Pair*, int flags = 0) { last_request = flags; }
};

class Comm {
 public:
  // single-rank periodic reverse communication: ghost g -> owner[g]
  std::vector<int> owner;
  int nlocal = 0;
  double cutghost[3] = {0, 0, 0};   // ghost cutoff per dimension (comm.h)
  void reverse_comm(Pair* p);
  void forward_comm(Pair* p);   // per-atom data of the pair style from every owner to its ghosts
};

class Domain {   // the members of domain.h the adapter reads
 public:
  int triclinic = 0;
  double sublo[3] = {0, 0, 0}, subhi[3] = {0, 0, 0};
};

class LAMMPS {
 public:
  Atom* atom = new Atom;
  Force* force = new Force;
  Update* update = new Update;
  Neighbor* neighbor = new Neighbor;
  Comm* comm = new Comm;
  Domain* domain = new Domain;
  Error* error = new Error;
  Memory* memory = new Memory;
  MPI_Comm world = 0;
};

class Pointers {
 public:
  explicit Pointers(LAMMPS* p)
      : lmp(p), atom(p->atom), force(p->force), update(p->update), neighbor(p->neighbor), comm(p->comm), domain(p->domain),
        error(p->error), memory(p->memory), world(p->world) {}
  virtual ~Pointers() = default;

 protected:
  LAMMPS* lmp;
  Atom*& atom;
  Force*& force;
  Update*& update;
  Neighbor*& neighbor;
  Comm*& comm;
  Domain*& domain;
  Error*& error;
  Memory*& memory;
  MPI_Comm& world;
};

class Pair : protected Pointers {
 public:
  explicit Pair(LAMMPS* p) : Pointers(p) {}
  ~Pair() override {
    free(eatom);
  }
  double eng_vdwl = 0, eng_coul = 0;
  double virial[6] = {0, 0, 0, 0, 0, 0};
  double* eatom = nullptr;
  int comm_forward = 0, comm_reverse = 0, comm_reverse_off = 0, single_enable = 1, writedata = 0, restartinfo = 1;
  NeighList* list = nullptr;

  virtual void compute(int, int) = 0;
  virtual void settings(int, char**) = 0;
  virtual void coeff(int, char**) = 0;
  virtual void init_style() {}
  virtual double init_one(int, int) { return 0; }
  virtual void* extract(const char*, int&) { return nullptr; }
  virtual void write_restart(FILE*) {}
  virtual void read_restart(FILE*) {}
  virtual int pack_reverse_comm(int, int, double*) { return 0; }
  virtual void unpack_reverse_comm(int, int*, double*) {}
  virtual int pack_forward_comm(int, int*, double*, int, int*) { return 0; }
  virtual void unpack_forward_comm(int, int, double*) {}

  void ev_init(int eflag, int vflag) {
    eflag_either = eflag;
    eflag_global = eflag & 1;
    eflag_atom = eflag & 2;
    vflag_either = vflag;
    vflag_global = vflag & 3;
    eng_vdwl = eng_coul = 0;
    for (double& v : virial) v = 0;
    if (eflag_atom) {
      const int n = atom->nlocal + atom->nghost;
      eatom = (double*)realloc(eatom, sizeof(double) * (n > 0 ? n : 1));
      for (int i = 0; i < n; i++) eatom[i] = 0;
    }
  }

 protected:
  int allocated = 0;
  int** setflag = nullptr;
  double** cutsq = nullptr;
  int eflag_either = 0, eflag_global = 0, eflag_atom = 0, vflag_either = 0, vflag_global = 0;
};

inline void Comm::reverse_comm(Pair* p) {
  const int ng = (int)owner.size();
  if (!ng) return;
  static thread_local std::vector<double> buf;   // LAMMPS keeps its communication buffers too
  if (buf.size() < 3 * (size_t)ng) buf.resize(3 * (size_t)ng);
  p->pack_reverse_comm(ng, nlocal, buf.data());
  p->unpack_reverse_comm(ng, owner.data(), buf.data());
}

inline void Comm::forward_comm(Pair* p) {
  const int ng = (int)owner.size();
  if (!ng) return;
  std::vector<double> buf((size_t)(p->comm_forward > 0 ? p->comm_forward : 1) * ng);
  p->pack_forward_comm(ng, owner.data(), buf.data(), 0, nullptr);
  p->unpack_forward_comm(ng, nlocal, buf.data());
}

namespace utils {
inline double numeric(const char*, int, const char* s, bool, LAMMPS*) { return atof(s); }
inline int inumeric(const char*, int, const char* s, bool, LAMMPS*) { return atoi(s); }
inline void bounds(const char* f, int l, const std::string& s, int nmin, int nmax, int& lo, int& hi, Error* e) {
  if (s == "*") { lo = nmin; hi = nmax; return; }
  lo = hi = atoi(s.c_str());
  if (lo < nmin || hi > nmax) e->all(f, l, "bounds");
}
inline void sfread(const char* f, int l, void* p, size_t sz, size_t n, FILE* fp, const char*, Error* e) {
  if (fread(p, sz, n, fp) != n) e->all(f, l, "unexpected end of restart file");
}
}  // namespace utils

// ---- lammpsplugin.h -------------------------------------------------------------------------------------
typedef void*(lammpsplugin_factory1)(LAMMPS*);
typedef struct {
  const char* version;
  const char* style;
  const char* name;
  const char* info;
  const char* author;
  union { lammpsplugin_factory1* v1; } creator;
  void* handle;
} lammpsplugin_t;
typedef void (*lammpsplugin_regfunc)(lammpsplugin_t*, void*);

}  // namespace LAMMPS_NS

using LAMMPS_NS::lammpsplugin_factory1;
using LAMMPS_NS::lammpsplugin_regfunc;
using LAMMPS_NS::lammpsplugin_t;
