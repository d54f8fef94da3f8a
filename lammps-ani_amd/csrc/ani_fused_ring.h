// ani_fused_ring.h — the LDS slots that carry the weight stream of the fused MLP kernel (ani_kernels_mlpf.hip), and the
// walk over the stream's slabs.  Plain integer logic, compiled for the device AND for the host: tests/ring_sim.cpp replays it
// on the CPU for every compiled shape and checks what the kernel's correctness rests on --
//   (1) the consumer's j-th boundary asks for exactly the slab the issuer's j-th step loaded, in the slot it loaded it to, and
//   (2) a load never writes a slot whose slab a wave may still be reading.
// Internal to libani_hip.so.
//
// Scheme (version 5; version 4 kept a ring of variable-size slabs with run-time bookkeeping -- fill levels, wrap rules, early
// and late boundaries, counted waits -- that cost ~150 scalar instructions and a dozen branches per slab on a kernel whose
// single wave per SIMD issues in order): kSlots = 3 slots of kSlot pieces, slab j lives in slot j mod 3 at a fixed offset.
// In front of slab j ("boundary j") every wave waits for its own loads in flight (those of slab j, issued one slab earlier),
// the workgroup barrier follows, and behind it the wave issues its share of slab j + 1 into slot (j + 1) mod 3 -- the slot of
// slab j - 2.  A boundary may stand in front of the LAST block of slab j - 1 (so that slab j's first fragments are requested
// beside that block's MFMAs): whoever is past barrier j has finished every block of slab j - 2, which is all the refill needs.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ANI_RING_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define ANI_RING_HD inline
#endif

namespace ani {

constexpr int kSlots = 3;
constexpr int kSlot = 48;                 // pieces (KB) per slot: the largest slab (two k-steps of an 8-tile first layer, 3 planes)
constexpr int kRing = kSlots * kSlot;     // pieces of LDS the slots take; with the constants' 4 KB: 148 of a CU's 160 KB
constexpr int kChunk = 4;                 // dE/dAEV tiles walked together through all k-steps (accumulators: 16 registers each)

struct Ring {
  const unsigned char* src;   // the tile's weight stream (global), members back to back
  int total;                  // pieces in it
  int qi;                     // next piece of the stream to issue
  int iseg, nseg;             // issuer: segment (see segment()) of the next slab to issue; segments per member
  int ileft, isize;           // issuer: slabs left in that segment, pieces per slab of it
  int islot, cslot;           // slot the next slab is loaded to / the next boundary reads from
  int pq0, pslot, pn, pk;     // the slab whose loads are being issued a piece per block: first stream piece, slot, size, this
                              // wave's next piece of it (pk >= pn: nothing pending)
  int ks0, nt0;               // AEV k-steps / dE/dAEV tiles of the problem (the slab sequence depends on them)
};

// k-steps of the first product per slab: two where such a slab fits a slot, else one
template <int NT1, int P>
struct F1Slab { static constexpr int k = 2 * NT1 * P <= kSlot ? 2 : 1; };

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
  r.src = src; r.total = total; r.qi = 0;
  r.ks0 = ks0; r.nt0 = nt0;
  r.nseg = 6 + (nt0 + kChunk - 1) / kChunk;
  r.iseg = r.nseg - 1;
  next_segment<NT1, NT2, NT3, P>(r);   // -> the first segment with slabs in it
  r.islot = 0; r.cslot = 0;
  r.pq0 = 0; r.pslot = 0; r.pn = 0; r.pk = 0;
}
ANI_RING_HD int next_slot(int s) { return s + 1 == kSlots ? 0 : s + 1; }

// Issuer: the next slab of the stream, if any is left: first stream piece q0, size n, slot; the caller starts the loads.
template <int NT1, int NT2, int NT3, int P>
ANI_RING_HD bool ring_take(Ring& r, int& q0, int& n, int& slot) {
  if (r.qi >= r.total) return false;
  q0 = r.qi; n = r.isize; slot = r.islot;
  r.qi += n;
  r.islot = next_slot(r.islot);
  if (--r.ileft == 0) next_segment<NT1, NT2, NT3, P>(r);
  return true;
}
// Consumer: the slot of the slab the boundary it stands at will read
ANI_RING_HD int ring_consume(Ring& r) {
  const int s = r.cslot;
  r.cslot = next_slot(s);
  return s;
}

}  // namespace ani
