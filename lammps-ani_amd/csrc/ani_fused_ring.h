// ani_fused_ring.h — bookkeeping of the LDS ring that carries the weight stream of the fused MLP kernel
// (ani_kernels_mlpf.hip).  Plain integer logic, compiled for the device AND for the host: tests/ring_sim.cpp replays it on
// the CPU for every compiled shape and checks the two properties the kernel's correctness rests on --
//   (1) a slab is never read before all of its pieces have been issued (and waited for), and
//   (2) a refill never writes ring space that a wave may still be reading.
// Internal to libani_hip.so.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ANI_RING_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define ANI_RING_HD inline
#endif

namespace ani {

#ifndef ANI_FUSED_RING
#define ANI_FUSED_RING 152
#endif
constexpr int kRing = ANI_FUSED_RING;   // pieces (KB) in the LDS ring: with the constants' 4 KB, all of a CU's 160 KB
ANI_RING_HD int ring_pos(int v) { return (int)((unsigned)v % (unsigned)kRing); }   // virtual position -> ring position
constexpr int kChunk = 4;               // dE/dAEV tiles walked together through all k-steps (accumulators: 16 registers each)

// A slab (the pieces a stretch of the kernel consumes between two boundaries) never wraps: a slab that would cross the ring's
// end starts at its beginning instead, issuer and consumer applying the same rule to the same sequence of slab sizes.
// Inside a slab every fragment is then at a compile-time offset from one base address (ds_read_b128 ... offset:imm).
struct Ring {
  const unsigned char* src;   // the tile's weight stream (global), members back to back
  int total;                  // pieces in it
  int qi;                     // next piece of the stream to issue
  int qg;                     // pieces the consumer has been granted so far (the slabs it has passed a boundary for)
  int iseg, nseg;             // issuer: segment (see segment()) of the next slab to issue; segments per member
  int ileft, isize;           // issuer: slabs left in that segment, pieces per slab of it
  int vw;                     // issuer: virtual write position (pieces, skipped space included; position = vw mod kRing)
  int vr;                     // consumer: virtual position of the oldest slab a wave may still be reading
  int ve;                     // consumer: virtual end of the newest slab it has been granted
  int own_issued, own_needed; // pieces this wave has issued / that the slabs granted so far needed from this wave
  int ks0, nt0;               // AEV k-steps / dE/dAEV tiles of the problem (the slab sequence depends on them)
};

// k-steps of the first product per slab: two where three such slabs fit the ring (what the early boundaries need), else one
template <int NT1, int P>
struct F1Slab { static constexpr int k = (2 * NT1 * P) * 3 <= kRing ? 2 : 1; };

// pieces k = wave, wave + 4, ... < n of a slab: this wave's share
ANI_RING_HD int own_share(int n, int wave) { return (n - wave + 3) >> 2; }

// A member's slabs come in SEGMENTS of equal slabs.  A slab of a forward product (and of dE/dAEV) is TWO k-steps of it (F1:
// F1Slab::k of them):
//   0: F1 ks0 / k x [k NT1]   1: F1's odd k-step (k = 2 only) (ks0 & 1) x [NT1]   2: F2 NT1 x [2 NT2]   3: F3 NT2 x [2 NT3]
//   4: B3 NT2 x [2 NT3] (one output tile through its 2 NT3 k-steps)   5: B2 NT1 x [2 NT2]
//   6 + c: chunk c of kChunk dE/dAEV tiles, NT1 x [2 tiles of the chunk]        (sizes times P)
// The issuer looks a segment up when it enters it, not per slab.
template <int NT1, int NT2, int NT3, int P>
ANI_RING_HD void segment(int seg, int ks0, int nt0, int& count, int& size) {
  constexpr int K1 = F1Slab<NT1, P>::k;
  if (seg == 0) { count = K1 == 2 ? ks0 >> 1 : ks0; size = K1 * NT1 * P; }
  else if (seg == 1) { count = K1 == 2 ? ks0 & 1 : 0; size = NT1 * P; }
  else if (seg == 2) { count = NT1; size = 2 * NT2 * P; }
  else if (seg == 3) { count = NT2; size = 2 * NT3 * P; }
  else if (seg == 4) { count = NT2; size = 2 * NT3 * P; }
  else if (seg == 5) { count = NT1; size = 2 * NT2 * P; }
  else {
    const int left = nt0 - kChunk * (seg - 6);
    count = NT1; size = 2 * (left < kChunk ? left : kChunk) * P;
  }
}
template <int NT1, int NT2, int NT3, int P>
ANI_RING_HD void next_segment(Ring& r) {
  do {
    r.iseg = r.iseg + 1 == r.nseg ? 0 : r.iseg + 1;
    segment<NT1, NT2, NT3, P>(r.iseg, r.ks0, r.nt0, r.ileft, r.isize);
  } while (r.ileft == 0);
}
template <int NT1, int NT2, int NT3, int P>
ANI_RING_HD void ring_reset(Ring& r, const unsigned char* src, int total, int ks0, int nt0) {
  r.src = src; r.total = total; r.qi = 0; r.qg = 0;
  r.ks0 = ks0; r.nt0 = nt0;
  r.nseg = 6 + (nt0 + kChunk - 1) / kChunk;
  r.iseg = r.nseg - 1;
  next_segment<NT1, NT2, NT3, P>(r);   // -> the first segment with slabs in it
  r.vw = 0; r.vr = 0; r.ve = 0; r.own_issued = 0; r.own_needed = 0;
}

// Issuer: if the next slab of the stream fits behind the consumer, take it: first stream piece q0, ring position pos (pieces),
// size n; the caller starts the loads.
template <int NT1, int NT2, int NT3, int P>
ANI_RING_HD bool ring_take(Ring& r, int wave, int& q0, int& pos, int& n) {
  if (r.qi >= r.total) return false;
  n = r.isize;
  const int pw = ring_pos(r.vw);
  const int vws = pw + n > kRing ? r.vw + (kRing - pw) : r.vw;
  if (vws + n - r.vr > kRing) return false;
  q0 = r.qi;
  pos = pw + n > kRing ? 0 : pw;
  r.own_issued += own_share(n, wave);
  r.qi += n;
  r.vw = vws + n;
  if (--r.ileft == 0) next_segment<NT1, NT2, NT3, P>(r);
  return true;
}
// Consumer, in front of a slab of n pieces.  ring_place: the slab's virtual start (no state change).  ring_issued: whether
// the refills so far have issued it; if not -- possible only behind early boundaries that protected their predecessor, and
// only at a LATE boundary, where every older slab is finished -- the caller frees the ring (barrier, ring_before_refill<false>),
// refills, and then proceeds as usual.  ring_grant, BEFORE the wait and the barrier: how many of this wave's own loads may
// still be in flight once the slab's pieces have landed.
ANI_RING_HD int ring_place(const Ring& r, int n) {
  const int pe = ring_pos(r.ve);
  return pe + n > kRing ? r.ve + (kRing - pe) : r.ve;
}
ANI_RING_HD bool ring_issued(const Ring& r, int n) { return r.qg + n <= r.qi; }
ANI_RING_HD int ring_grant(Ring& r, int n, int wave) {
  r.own_needed += own_share(n, wave);
  r.qg += n;
  return r.own_issued - r.own_needed;
}
// Whether the boundary of the next slab (n pieces) can be taken EARLY, in front of the last block of the slab being read:
// only if the refills so far have already issued it (an early boundary protects the slab being read, so its own refill
// might not find room for it).  Same answer on every wave.  Otherwise the boundary comes after that block.
ANI_RING_HD bool ring_can_go_early(const Ring& r, int n) { return ring_issued(r, n); }
// ... AFTER the barrier, around the refill.  EARLY = false: every wave has finished every slab before this one, whose space
// the refill may take.  EARLY = true: the boundary sits in front of the LAST block of the slab before, which stays protected
// from this refill; by the next one (behind the next barrier) every wave is past it.
template <bool EARLY>
ANI_RING_HD void ring_before_refill(Ring& r, int vs) { if (!EARLY) r.vr = vs; }
ANI_RING_HD void ring_after_refill(Ring& r, int vs, int n) { r.vr = vs; r.ve = vs + n; }

}  // namespace ani
