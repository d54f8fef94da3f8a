// lmp_harness.cpp — a LAMMPS-free stand-in for what LAMMPS core hands to PairANI::compute().
//
// LAMMPS itself is not available (external/lammps is an empty submodule), so tests and
// bench.py need something that produces the same inputs the pair style sees (src/pair_ani.cpp:70-85):
//   * per-rank atom arrays x[ntotal][3], type[ntotal] with nlocal owned atoms followed by nghost ghost atoms
//     (periodic images and atoms owned by other ranks) within cutghost = cutoff + skin of the rank's brick,
//   * a neighbour list of the owned atoms over owned+ghost atoms: full (REQ_FULL, src/pair_ani.cpp:385) or
//     half with newton_pair off (src/pair_ani.cpp:388: every local-ghost pair is kept by the local side),
//   * the ghost -> owner map that LAMMPS' Comm uses for reverse_comm (src/pair_ani.cpp:197-201,461-484).
// Decomposition is LAMMPS' brick style: P = px*py*pz equal bricks of an orthogonal box
// (examples/benchmark/submit_scaling.py:13-21 uses 1, 2x1x1, 2x2x1, 2x2x2).
//
// Plain C ABI for ctypes; no GPU code here.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

struct Domain {
  int natoms = 0, nlocal = 0, nghost = 0;
  double lo[3], hi[3], sublo[3], subhi[3], cutghost = 0;
  std::vector<double> x;        // [ntotal*3]
  std::vector<int> type;        // [ntotal]
  std::vector<int> tag;         // [ntotal] global atom index (0-based)
  std::vector<int> owner_rank;  // [nghost] rank that owns the ghost's original atom
  std::vector<int> owner_lidx;  // [nghost] local index of that atom on its owner
  std::vector<int> shift;       // [nghost*3] periodic image shift (in box lengths) applied to the owner's position
  // neighbour list (last built)
  std::vector<int> numneigh;    // [nlocal]
  std::vector<int> jlist;       // flattened, in local-atom order
};

inline int brick_of(double v, double lo, double len, int p) {
  int b = (int)std::floor((v - lo) / len * p);
  return std::min(std::max(b, 0), p - 1);
}

}  // namespace

extern "C" {

// x_all: [natoms*3] positions (wrapped into the box for periodic dims on entry), type_all: [natoms] LAMMPS types.
void* hx_build(int natoms, const double* x_all, const int* type_all, const double* boxlo, const double* boxhi,
               const int* periodic, int px, int py, int pz, int rank, double cutghost) {
  Domain* d = new Domain;
  d->natoms = natoms;
  d->cutghost = cutghost;
  const int P[3] = {px, py, pz};
  const int r3[3] = {rank % px, (rank / px) % py, rank / (px * py)};
  double L[3];
  for (int c = 0; c < 3; c++) {
    d->lo[c] = boxlo[c];
    d->hi[c] = boxhi[c];
    L[c] = boxhi[c] - boxlo[c];
    d->sublo[c] = boxlo[c] + L[c] * r3[c] / P[c];
    d->subhi[c] = boxlo[c] + L[c] * (r3[c] + 1) / P[c];
  }
  // wrap, find owners and owner-local indices (original order is kept inside each rank)
  std::vector<double> xw(3 * (size_t)natoms);
  std::vector<int> orank(natoms), olidx(natoms);
  std::vector<int> count(px * py * pz, 0);
  for (int i = 0; i < natoms; i++) {
    int b[3];
    for (int c = 0; c < 3; c++) {
      double v = x_all[3 * i + c];
      if (periodic[c]) {
        v -= L[c] * std::floor((v - boxlo[c]) / L[c]);
        if (v >= boxhi[c]) v = boxlo[c];
      }
      xw[3 * i + c] = v;
      b[c] = brick_of(v, boxlo[c], L[c], P[c]);
    }
    int r = b[0] + px * (b[1] + py * b[2]);
    orank[i] = r;
    olidx[i] = count[r]++;
  }
  for (int i = 0; i < natoms; i++)
    if (orank[i] == rank) {
      for (int c = 0; c < 3; c++) d->x.push_back(xw[3 * i + c]);
      d->type.push_back(type_all[i]);
      d->tag.push_back(i);
    }
  d->nlocal = (int)d->tag.size();
  int nimg[3];
  for (int c = 0; c < 3; c++) nimg[c] = periodic[c] ? (int)std::ceil(cutghost / L[c]) : 0;
  for (int i = 0; i < natoms; i++) {
    for (int sz = -nimg[2]; sz <= nimg[2]; sz++)
      for (int sy = -nimg[1]; sy <= nimg[1]; sy++)
        for (int sx = -nimg[0]; sx <= nimg[0]; sx++) {
          if (sx == 0 && sy == 0 && sz == 0 && orank[i] == rank) continue;
          const int s[3] = {sx, sy, sz};
          double v[3];
          bool in = true;
          for (int c = 0; c < 3 && in; c++) {
            v[c] = xw[3 * i + c] + s[c] * L[c];
            in = (v[c] >= d->sublo[c] - cutghost) && (v[c] < d->subhi[c] + cutghost);
          }
          if (!in) continue;
          for (int c = 0; c < 3; c++) d->x.push_back(v[c]);
          d->type.push_back(type_all[i]);
          d->tag.push_back(i);
          d->owner_rank.push_back(orank[i]);
          d->owner_lidx.push_back(olidx[i]);
          for (int c = 0; c < 3; c++) d->shift.push_back(s[c]);
        }
  }
  d->nghost = (int)d->owner_rank.size();
  return d;
}

void hx_free(void* h) { delete (Domain*)h; }
int hx_nlocal(void* h) { return ((Domain*)h)->nlocal; }
int hx_nghost(void* h) { return ((Domain*)h)->nghost; }
const double* hx_x(void* h) { return ((Domain*)h)->x.data(); }
const int* hx_type(void* h) { return ((Domain*)h)->type.data(); }
const int* hx_tag(void* h) { return ((Domain*)h)->tag.data(); }
const int* hx_owner_rank(void* h) { return ((Domain*)h)->owner_rank.data(); }
const int* hx_owner_lidx(void* h) { return ((Domain*)h)->owner_lidx.data(); }
const int* hx_shift(void* h) { return ((Domain*)h)->shift.data(); }
void hx_sub_bounds(void* h, double* lo, double* hi) {
  Domain* d = (Domain*)h;
  for (int c = 0; c < 3; c++) { lo[c] = d->sublo[c]; hi[c] = d->subhi[c]; }
}

// Overwrite positions (e.g. after an MD step).  x_new: [ntotal*3].
void hx_set_x(void* h, const double* x_new) {
  Domain* d = (Domain*)h;
  std::memcpy(d->x.data(), x_new, sizeof(double) * d->x.size());
}

// Binned neighbour build of the owned atoms over owned+ghost atoms.  half != 0: each local-local pair once
// (under the lower local index), every local-ghost pair once under the local atom (newton_pair off).
// Returns the number of pairs; arrays are then available through hx_numneigh / hx_jlist.
int64_t hx_neigh_build(void* h, double cutneigh, int half) {
  Domain* d = (Domain*)h;
  const int ntotal = d->nlocal + d->nghost;
  const double* x = d->x.data();
  double blo[3], bhi[3];
  for (int c = 0; c < 3; c++) { blo[c] = 1e300; bhi[c] = -1e300; }
  for (int i = 0; i < ntotal; i++)
    for (int c = 0; c < 3; c++) { blo[c] = std::min(blo[c], x[3 * i + c]); bhi[c] = std::max(bhi[c], x[3 * i + c]); }
  int nb[3];
  double inv[3];
  for (int c = 0; c < 3; c++) {
    double len = std::max(bhi[c] - blo[c], 1e-9);
    nb[c] = std::max(1, (int)std::floor(len / cutneigh));
    nb[c] = std::min(nb[c], 256);
    inv[c] = nb[c] / (len * (1.0 + 1e-12));
  }
  const int ncell = nb[0] * nb[1] * nb[2];
  std::vector<int> cell(ntotal), head(ncell + 1, 0), order(ntotal);
  auto cidx = [&](int i, int* b) {
    for (int c = 0; c < 3; c++) { b[c] = (int)((x[3 * i + c] - blo[c]) * inv[c]); b[c] = std::min(std::max(b[c], 0), nb[c] - 1); }
    return b[0] + nb[0] * (b[1] + nb[1] * b[2]);
  };
  int b[3];
  for (int i = 0; i < ntotal; i++) { cell[i] = cidx(i, b); head[cell[i] + 1]++; }
  for (int c = 0; c < ncell; c++) head[c + 1] += head[c];
  { std::vector<int> fill(head.begin(), head.end() - 1); for (int i = 0; i < ntotal; i++) order[fill[cell[i]]++] = i; }
  const double c2 = cutneigh * cutneigh;
  d->numneigh.assign(d->nlocal, 0);
  d->jlist.clear();
  d->jlist.reserve((size_t)d->nlocal * 64);
  for (int i = 0; i < d->nlocal; i++) {
    cidx(i, b);
    int cnt = 0;
    for (int dz = -1; dz <= 1; dz++) {
      int z = b[2] + dz; if (z < 0 || z >= nb[2]) continue;
      for (int dy = -1; dy <= 1; dy++) {
        int y = b[1] + dy; if (y < 0 || y >= nb[1]) continue;
        for (int dx = -1; dx <= 1; dx++) {
          int xx = b[0] + dx; if (xx < 0 || xx >= nb[0]) continue;
          int c = xx + nb[0] * (y + nb[1] * z);
          for (int p = head[c]; p < head[c + 1]; p++) {
            int j = order[p];
            if (j == i) continue;
            if (half && j < d->nlocal && j < i) continue;
            double ddx = x[3 * j] - x[3 * i], ddy = x[3 * j + 1] - x[3 * i + 1], ddz = x[3 * j + 2] - x[3 * i + 2];
            if (ddx * ddx + ddy * ddy + ddz * ddz < c2) { d->jlist.push_back(j); cnt++; }
          }
        }
      }
    }
    // deterministic order independent of binning
    std::sort(d->jlist.end() - cnt, d->jlist.end());
    d->numneigh[i] = cnt;
  }
  return (int64_t)d->jlist.size();
}

const int* hx_numneigh(void* h) { return ((Domain*)h)->numneigh.data(); }
const int* hx_jlist(void* h) { return ((Domain*)h)->jlist.data(); }

}  // extern "C"
