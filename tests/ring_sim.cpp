// ring_sim.cpp — CPU replay of the weight slots of the fused MLP kernel (lammps-ani_amd/csrc/ani_fused_ring.h: the very
// functions the kernel runs) for every compiled shape, both arithmetics, every AEV width and member count of interest.
// Mirrors the order of the boundaries in ani_kernels_mlpf.hip:fused_tile (slab 0 issued at the start of a tile; at boundary j
// the wait, the barrier, then the loads of slab j + 1) and checks, at every boundary, that
//   (1) the slab about to be read was issued, with the size the consumer expects, into the slot the consumer reads,
//   (2) the loads started behind the barrier go to a slot whose slab is at least two behind the one about to be read
//       (a boundary may stand in front of the LAST block of the slab before, which is therefore still being read), and
//   (3) every slab fits a slot; at the end of the tile the stream has been issued and consumed to its last piece.
// Test infrastructure only.
// build: g++ -std=c++17 -O1 -I lammps-ani_amd/csrc tests/ring_sim.cpp -o ring_sim ; exit code 0 = all properties hold
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ani_fused_ring.h"

using namespace ani;

struct Sim {
  Ring r;
  int slot_slab[kSlots];                        // slab id whose pieces the slot holds (-1: none)
  std::vector<std::pair<int, int>> issued;      // slab id -> (slot, pieces)
  int consumed = 0;
  long long checks = 0, pieces = 0;
  bool ok = true;
  const char* what = "";

  void fail(const char* msg) { if (ok) { ok = false; what = msg; } }

  template <int NT1, int NT2, int NT3, int P>
  void issue_next() {   // ring_issue_next of the kernel
    int q0, n, slot;
    if (!ring_take<NT1, NT2, NT3, P>(r, q0, n, slot)) return;
    const int id = (int)issued.size();
    if (n <= 0 || n > kSlot) fail("a slab does not fit a slot");
    if (q0 != pieces) fail("the stream is not issued piece by piece in order");
    if (slot < 0 || slot >= kSlots || slot != id % kSlots) fail("slab j is not in slot j mod kSlots");
    // whoever starts these loads is past the barrier of boundary id - 1: slabs up to id - 3 are finished, slab id - 2 is at
    // most in its last block (early boundary) -- so the slot must not hold anything younger than id - 3
    if (slot_slab[slot] != -1 && slot_slab[slot] > id - kSlots) fail("a load writes a slot whose slab may still be read");
    if (slot_slab[slot] != -1 && slot_slab[slot] >= consumed - 1 && consumed > 0 && slot_slab[slot] > consumed - 2)
      fail("a load writes the slot of the slab being read or the one before");
    slot_slab[slot] = id;
    issued.push_back({slot, n});
    pieces += n;
  }
  template <int NT1, int NT2, int NT3, int P>
  void boundary(int n) {   // ring_boundary of the kernel: wait, barrier, consume, issue the next slab
    checks++;
    const int id = consumed;
    if (id >= (int)issued.size()) { fail("a boundary for a slab that was never issued"); return; }
    const int s = ring_consume(r);
    if (issued[id].first != s) fail("consumer and issuer disagree on a slab's slot");
    if (issued[id].second != n) fail("consumer and issuer disagree on a slab's size");
    if (slot_slab[s] != id) fail("the slab's slot was overwritten before it was read");
    consumed++;
    issue_next<NT1, NT2, NT3, P>();
  }
  template <int NT1, int NT2, int NT3, int P>
  void tile(int ks0, int nt0, int M) {
    constexpr int K1 = F1Slab<NT1, P>::k;
    for (int& s : slot_slab) s = -1;
    // pieces per member (ani_kernels_mlpf.hip:fused_pieces_per_member)
    const long long ppm = (long long)P * ((long long)ks0 * NT1 + 2LL * NT1 * NT2 + 2LL * NT2 * NT3 + 2LL * NT3 * NT2 + 2LL * NT2 * NT1 + 2LL * NT1 * nt0);
    ring_reset<NT1, NT2, NT3, P>(r, nullptr, (int)(ppm * M), ks0, nt0);
    issue_next<NT1, NT2, NT3, P>();
    for (int m = 0; m < M && ok; m++) {
      // F1
      if (K1 == 2) {
        const int nslab = (ks0 + 1) >> 1;
        for (int kp = 0; kp < nslab; kp++) {
          const int left = ks0 - 2 * kp;
          boundary<NT1, NT2, NT3, P>((left < 2 ? left : 2) * NT1 * P);
        }
      } else {
        for (int ks = 0; ks < ks0; ks++) boundary<NT1, NT2, NT3, P>(NT1 * P);
      }
      auto product = [&](int nslabs, int size) { for (int i = 0; i < nslabs; i++) boundary<NT1, NT2, NT3, P>(size); };
      product(NT1, 2 * NT2 * P);   // F2: 2 NT1 k-steps, two per slab
      product(NT2, 2 * NT3 * P);   // F3
      product(NT2, 2 * NT3 * P);   // B3: one output tile per slab
      product(NT1, 2 * NT2 * P);   // B2
      for (int c0 = 0; c0 < nt0; c0 += kChunk) {   // B1
        const int ntc = nt0 - c0 < kChunk ? nt0 - c0 : kChunk;
        product(NT1, 2 * ntc * P);
      }
    }
    if (ok && (pieces != r.total || r.qi != r.total || consumed != (int)issued.size())) fail("the stream was not issued and consumed to its end");
  }
};

template <int NT1, int NT2, int NT3, int P>
static long long sweep(int& bad) {
  long long checks = 0;
  for (int ks0 = 1; ks0 <= 64; ks0++)
    for (int M = 1; M <= 3; M++) {
      const int nt0 = (ks0 + 1) / 2;   // both follow from the AEV width
      Sim s;
      s.tile<NT1, NT2, NT3, P>(ks0, nt0, M);
      checks += s.checks;
      if (!s.ok) {
        std::printf("FAIL shape (%d,%d,%d) P=%d ks0=%d nt0=%d M=%d: %s\n", NT1, NT2, NT3, P, ks0, nt0, M, s.what);
        bad++;
      }
    }
  std::printf("shape (%d,%d,%d) P=%d: %lld boundaries\n", NT1, NT2, NT3, P, checks);
  return checks;
}

int main() {
  int bad = 0;
  long long checks = 0;
  checks += sweep<8, 6, 5, 3>(bad); checks += sweep<6, 5, 4, 3>(bad); checks += sweep<5, 4, 3, 3>(bad);
  checks += sweep<8, 6, 5, 2>(bad); checks += sweep<6, 5, 4, 2>(bad); checks += sweep<5, 4, 3, 2>(bad);
  std::printf("ring_sim: %lld boundaries replayed, %d failing configurations\n", checks, bad);
  return bad ? 1 : 0;
}
