// ring_sim.cpp — CPU replay of the weight ring of the fused MLP kernel (lammps-ani_amd/csrc/ani_fused_ring.h: the very
// functions the kernel runs) for every compiled shape, both arithmetics, every AEV width and member count of interest.
// Mirrors the order and kind (late / early) of the boundaries in ani_kernels_mlpf.hip:fused_tile and checks, at every
// boundary, that (1) the slab about to be read was issued, and is complete once the kernel's counted wait returns -- with
// the wait's rounding to the immediates {16, 8, 4, 0} and each of the four waves' own load queue --, and (2) no refill writes
// ring space of a slab that a wave may still be reading or that has not been consumed yet.  Test infrastructure only.
// build: g++ -std=c++17 -O1 -I lammps-ani_amd/csrc tests/ring_sim.cpp -o ring_sim ; exit code 0 = all properties hold
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <vector>

#include "ani_fused_ring.h"

using namespace ani;

static int wait_immediate(int c) { return c >= 16 ? 16 : (c >= 8 ? 8 : (c >= 4 ? 4 : 0)); }   // wait_vmcnt_le of the kernel

struct Sim {
  Ring r[4];                       // the four waves' copies of the ring state (identical but for the own_* counts)
  std::deque<int> queue[4];        // per wave: slab ids of its loads in flight, oldest first
  std::vector<int> owner;          // ring position -> slab id occupying it (-1 free)
  std::vector<std::pair<int, int>> where;   // slab id -> (pos, n)
  int next_issue_id = 0, next_grant_id = 0;
  std::vector<int> live;           // slabs a wave may be reading
  long long checks = 0;
  bool ok = true;
  const char* what = "";

  void fail(const char* msg) { if (ok) { ok = false; what = msg; } }

  template <int NT1, int NT2, int NT3, int P>
  void refill() {
    for (;;) {
      int q0[4], pos[4], n[4];
      bool took[4];
      for (int w = 0; w < 4; w++) took[w] = ring_take<NT1, NT2, NT3, P>(r[w], w, q0[w], pos[w], n[w]);
      for (int w = 1; w < 4; w++)
        if (took[w] != took[0] || (took[0] && (q0[w] != q0[0] || pos[w] != pos[0] || n[w] != n[0]))) fail("waves disagree on a refill");
      if (!took[0]) break;
      if (pos[0] + n[0] > kRing) fail("slab wraps");
      for (int k = 0; k < n[0]; k++) {
        if (owner[pos[0] + k] != -1) fail("refill writes over a slab that is live or not yet consumed");
        owner[pos[0] + k] = next_issue_id;
      }
      where.push_back({pos[0], n[0]});
      for (int w = 0; w < 4; w++)
        for (int k = w; k < n[0]; k += 4) queue[w].push_back(next_issue_id);
      next_issue_id++;
    }
  }
  void retire(int id) {
    for (int k = 0; k < where[id].second; k++) owner[where[id].first + k] = -1;
  }
  long long slow = 0;
  template <int NT1, int NT2, int NT3, int P, bool EARLY>
  void boundary(int n) {
    checks++;
    const int id = next_grant_id++;
    int vs[4], allowed[4];
    for (int w = 0; w < 4; w++) vs[w] = ring_place(r[w], n);
    if (!ring_issued(r[0], n)) {
      if (EARLY) { fail("an early boundary for a slab that was not issued"); return; }
      // slow path of a late boundary: barrier, every older slab is finished, refill, then the usual sequence
      slow++;
      for (int s : live) retire(s);
      live.clear();
      for (int w = 0; w < 4; w++) ring_before_refill<false>(r[w], vs[w]);
      refill<NT1, NT2, NT3, P>();
      if (!ring_issued(r[0], n)) { fail("a slab does not fit an empty ring"); return; }
    }
    for (int w = 0; w < 4; w++) allowed[w] = ring_grant(r[w], n, w);
    for (int w = 0; w < 4; w++) {
      if (allowed[w] < 0) { fail("negative count of loads allowed in flight"); return; }
      if (vs[w] != vs[0]) fail("waves disagree on a slab's position");
      // the counted wait: all but wait_immediate(allowed) youngest loads of the wave complete
      const int keep = wait_immediate(allowed[w]);
      while ((int)queue[w].size() > keep) queue[w].pop_front();
      for (int s : queue[w])
        if (s <= id) fail("a piece of the granted slab may still be in flight after the wait");
    }
    if (id >= (int)where.size() || where[id].first != ring_pos(vs[0]) || where[id].second != n)
      fail("consumer and issuer disagree on a slab's place or size");
    // barrier: every wave is here.  Late: nobody reads anything older any more.  Early: the slab before is still being read.
    if (!EARLY) { for (int s : live) retire(s); live.clear(); }
    for (int w = 0; w < 4; w++) ring_before_refill<EARLY>(r[w], vs[w]);
    refill<NT1, NT2, NT3, P>();
    for (int w = 0; w < 4; w++) ring_after_refill(r[w], vs[w], n);
    if (EARLY) {   // the last block of the slab before runs now; by the next boundary it is done
      for (int s : live) retire(s);
      live.clear();
    }
    live.push_back(id);
  }

  long long early = 0, late = 0;
  // what the kernel does in front of the last block of a slab that has a successor in its product
  template <int NT1, int NT2, int NT3, int P>
  void next_slab(int n) {
    bool e[4];
    for (int w = 0; w < 4; w++) e[w] = ring_can_go_early(r[w], n);
    if (e[1] != e[0] || e[2] != e[0] || e[3] != e[0]) fail("waves disagree on early / late");
    if (e[0]) { early++; boundary<NT1, NT2, NT3, P, true>(n); }
    else { late++; boundary<NT1, NT2, NT3, P, false>(n); }
  }
  template <int NT1, int NT2, int NT3, int P>
  void tile(int ks0, int nt0, int M) {
    constexpr int K1 = F1Slab<NT1, P>::k;
    owner.assign(kRing, -1);
    // pieces per member (ani_kernels_mlpf.hip:fused_pieces_per_member)
    const long long ppm = (long long)P * ((long long)ks0 * NT1 + 2LL * NT1 * NT2 + 2LL * NT2 * NT3 + 2LL * NT3 * NT2 + 2LL * NT2 * NT1 + 2LL * NT1 * nt0);
    for (int w = 0; w < 4; w++) ring_reset<NT1, NT2, NT3, P>(r[w], nullptr, (int)(ppm * M), ks0, nt0);
    refill<NT1, NT2, NT3, P>();
    for (int m = 0; m < M && ok; m++) {
      // F1
      if (K1 == 2) {
        const int nslab = (ks0 + 1) >> 1;
        boundary<NT1, NT2, NT3, P, false>((ks0 < 2 ? ks0 : 2) * NT1 * P);
        for (int kp = 0; kp < nslab; kp++) {
          const bool two = 2 * kp + 1 < ks0;
          if (two && kp + 1 < nslab) { const int left = ks0 - 2 * (kp + 1); next_slab<NT1, NT2, NT3, P>((left < 2 ? left : 2) * NT1 * P); }
        }
      } else {
        boundary<NT1, NT2, NT3, P, false>(NT1 * P);
        for (int ks = 0; ks + 1 < ks0; ks++) next_slab<NT1, NT2, NT3, P>(NT1 * P);
      }
      auto product = [&](int nslabs, int size) {
        boundary<NT1, NT2, NT3, P, false>(size);
        for (int i = 0; i + 1 < nslabs; i++) next_slab<NT1, NT2, NT3, P>(size);
      };
      product(NT1, 2 * NT2 * P);   // F2: 2 NT1 k-steps, two per slab
      product(NT2, 2 * NT3 * P);   // F3
      product(NT2, 2 * NT3 * P);   // B3: one output tile per slab
      product(NT1, 2 * NT2 * P);   // B2
      for (int c0 = 0; c0 < nt0; c0 += kChunk) {   // B1
        const int ntc = nt0 - c0 < kChunk ? nt0 - c0 : kChunk;
        if (ntc == kChunk) product(NT1, 2 * kChunk * P);
        else for (int kp = 0; kp < NT1; kp++) boundary<NT1, NT2, NT3, P, false>(2 * ntc * P);
      }
    }
    if (ok && (r[0].qg != r[0].total || r[0].qi != r[0].total)) fail("the stream was not consumed to its end");
  }
};

template <int NT1, int NT2, int NT3, int P>
static long long sweep(int& bad) {
  long long checks = 0, early = 0, late = 0, e8 = 0, l8 = 0, slow = 0;
  for (int ks0 = 1; ks0 <= 64; ks0++)
    for (int nt0 = (ks0 + 1) / 2; nt0 <= (ks0 + 1) / 2 + 0; nt0++)   // nt0 = ceil(ks0 / 2): both follow from the AEV width
      for (int M = 1; M <= 3; M++) {
        Sim s;
        s.tile<NT1, NT2, NT3, P>(ks0, nt0, M);
        checks += s.checks; early += s.early; late += s.late; slow += s.slow;
        if (ks0 == 8 && M == 1) { e8 = s.early; l8 = s.late; }
        if (!s.ok) {
          std::printf("FAIL shape (%d,%d,%d) P=%d ks0=%d nt0=%d M=%d: %s\n", NT1, NT2, NT3, P, ks0, nt0, M, s.what);
          bad++;
        }
      }
  std::printf("shape (%d,%d,%d) P=%d: boundaries inside products taken early %lld, late %lld (128-column AEV, one member: %lld / %lld); late boundaries that had to refill first: %lld\n",
              NT1, NT2, NT3, P, early, late, e8, l8, slow);
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
