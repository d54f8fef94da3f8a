// ani_kernels_mlpf.hip — the whole MLP of a row tile in ONE workgroup, activations in registers (gfx950).
//
// Replaces, for the usual ANI networks (three hidden layers), the six grouped GEMM launches of ani_kernels_mlp.hip:
// BmmEnsemble forward + energy shifter input + autograd back to dE/dAEV (reference call sites models/lammps_ani.py:110,
// 228-230,197-206).  The six products of a member are still genuine dense contractions on the matrix pipe; what changes is
// where their operands live.  The per-layer kernels wrote every activation to HBM only to read it back (1.1 GB per step at
// 100 002 atoms, 3.9 TB/s, the MFMA pipe 16-23 % busy); here a 128-row species-pure tile is taken through
//     h1 = celu(W0 aev + b0)  h2 = celu(W1 h1 + b1)  h3 = celu(W2 h2 + b2)  e = w3.h3 + b3
//     g3 = w3 celu'(z3)/M     g2 = (W2^T g3) celu'(z2)   g1 = (W1^T g2) celu'(z1)   dE/dAEV = W0^T g1
// by one workgroup of four waves, each wave owning 32 rows (atoms) for the whole pass.  HBM traffic of the MLP is then the
// AEV rows in and the dE/dAEV rows out.
//
// Transposed formulation: every product is  X_out[features][atoms] = W[features][k] * X_in[k][atoms], the WEIGHTS the A
// operand of v_mfma_f32_32x32x16_{bf16,f16}, the activations the B operand.  A 32x32 accumulator tile has its column (the
// atom) on the lane and its rows (features) in the 16 registers, which is exactly the B-operand layout of the next
// product's k-steps (cdna_hip_programming.md section 3, "An accumulator tile as the next MFMA's operand"): registers
// 8s .. 8s+7 of tile t are the fragment of k-step 2t + s, element j of lane half h being feature
//     kappa(ks, h, j) = 16 ks + 8 (j >> 2) + 4 h + (j & 3).
// So an activation never leaves its lane: celu, the celu' masks of the backward pass, the split into 16-bit terms are
// register arithmetic, and there is no barrier between layers.  The weights are permuted to match when the model is
// uploaded (build_stream_kernel): the stream of a (species, member) is a sequence of 1 KB PIECES in exactly the order the
// kernel consumes them -- piece (product, k-step, 32-row tile of outputs, plane) = the A fragment of one MFMA group, 16
// bytes per lane -- which the four waves fetch round-robin with LDS-DMA (global_load_lds_dwordx4: 1 KB per instruction,
// linear in LDS, no registers) into a ring of R pieces and all read back conflict-free (lane l reads bytes 16 l).
// Arithmetic as in ani_kernels_mlp.hip (option mlp_arith): P = 3 exact three-term bf16 splits, six products; P = 2
// two-term fp16 splits of scaled operands, three products.
//
// Synchronisation: a SLAB is the pieces of one k-step of one product.  Before a slab is consumed every wave waits (counted
// vmcnt) for its own pieces of it, then one workgroup barrier says (a) everybody's pieces have landed and (b) everybody has
// finished reading the slab before, whose ring space is refilled right behind the barrier.  One barrier per k-step
// (~1 000-1 500 matrix-pipe cycles); loads run up to R pieces ahead of the consumer.
//
// Register budget (per lane, 512 with one wave per SIMD; largest shape): h1 128 + h2 96 + h3 80 = 304 at the end of the
// forward pass; the backward products write each gradient over the activation it is masked with, their source held as
// 16-bit fragments (g3: 120, g2: 144), so 128 + 96 + 120 + an accumulator while g2 is formed.  One workgroup per CU; tiles
// are handed out from a counter, costliest species first.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <utility>
#include <vector>

#include "ani_fused_ring.h"
#include "ani_kernels.h"
#include "ani_mlp_common.h"

namespace ani {

// timing-only ablation switches (wrong results): ABLF_NOWAIT no wait for the pieces of a slab, ABLF_NOBAR no barrier in
// front of a slab, ABLF_NOMMA no MFMAs, ABLF_NODMA no weight loads
constexpr int kConstBytes = 4096;          // LDS copy of a member's biases / output layer
constexpr int kFusedLds = kRing * 1024 + kConstBytes;
static_assert(kFusedLds + 64 <= 160 * 1024, "ring + constants exceed the LDS of a CU");

// diagnostic build (-DABLF_STAMPS): wave 0 of every workgroup adds the shader-clock cycles it spent in each phase of a tile
// to g_fused_stamps (read by ani_debug_fused_stamps); no stamp executes in the shipped kernel
#ifdef ABLF_STAMPS
__device__ unsigned long long g_fused_stamps[16];
#define FUSED_STAMP(k)                                                                       \
  do {                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    unsigned long long _t;                                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");               \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    if (wave == 0 && lane == 0) atomicAdd(&g_fused_stamps[k], _t - stamp_prev);              \
    stamp_prev = _t;                                                                         \
  } while (0)
#define FUSED_STAMP_INIT()                                                                   \
  unsigned long long stamp_prev;                                                             \
  do {                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");       \
    __builtin_amdgcn_sched_barrier(0);                                                       \
  } while (0)
#else
#define FUSED_STAMP(k) do {} while (0)
#define FUSED_STAMP_INIT() do {} while (0)
#endif

// ---- weight stream of one product, in consumption order -------------------------------------------------------------
// src[row][k] (ld elements per row): the A operand, rows = output features of the product, k = its contraction index.
// Block b of the stream = (k-step ks, output tile nt): b = ks * NT + nt (forward products), b = nt * KS + ks (the two hidden
// backward products) or, chunked (dE/dAEV: the kernel walks 8 tiles at a time through all k-steps), chunk-major.  Block b holds P pieces of 64 lanes x 8 16-bit elements:
// lane l, element j  <-  src[32 nt + (l & 31)][kappa(ks, l >> 5, j)].
__global__ void build_stream_kernel(const float* __restrict__ src, int ld, int rows_valid, int k_valid, int NT, int KS, int chunk,
                                    int P, float scale, unsigned short* __restrict__ dst) {   // chunk < 0: tile-major blocks
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)NT * KS * 512;
  if (idx >= total) return;
  const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
  const int blk = (int)(idx >> 9);
  int ks, nt;
  if (chunk < 0) {        // b = nt * KS + ks: the backward products, one output tile through all its k-steps at a time
    nt = blk / KS; ks = blk - nt * KS;
  } else if (!chunk) {
    ks = blk / NT; nt = blk - ks * NT;
  } else {
    int c = 0, rem = blk;
    for (;;) {
      const int ntc = min(chunk, NT - chunk * c);
      if (rem < KS * ntc) { ks = rem / ntc; nt = chunk * c + (rem - ks * ntc); break; }
      rem -= KS * ntc; c++;
    }
  }
  const int row = 32 * nt + (lane & 31);
  const int k = 16 * ks + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3);
  const float x = (row < rows_valid && k < k_valid) ? src[(long long)row * ld + k] : 0.f;
  unsigned short* d = dst + ((long long)blk * P) * 512 + lane * 8 + j;
  if (P == 3) {
    unsigned h, m, l;
    split3(x, h, m, l);
    d[0] = (unsigned short)(h >> 16);
    d[512] = (unsigned short)(m >> 16);
    d[1024] = (unsigned short)(l >> 16);
  } else {
    unsigned h, l;
    split2(x * scale, 0.f, h, l);
    d[0] = (unsigned short)(h & 0xffffu);
    d[512] = (unsigned short)(l & 0xffffu);
  }
}

// ---- the weight slots (bookkeeping: ani_fused_ring.h) ----------------------------------------------------------------
__device__ __forceinline__ void ring_load_piece(const Ring& r, unsigned char* slot_base, int q0, int k, int lane16) {
#ifndef ABLF_NODMA
  const unsigned char* g = r.src + (size_t)(q0 + k) * 1024 + lane16;
  unsigned char* l = slot_base + (k << 10);
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
#endif
}
// this wave's share (pieces wave, wave + 4, ...) of the next slab of the stream, into the slot after the last one issued.
// An LDS-DMA instruction holds the wave's issue for 60 cycles and more: a slab's nine to twelve in one burst behind the
// barrier leave the matrix pipe idle for as long as twenty MFMAs (measured: 0.331 ms per step at 100 002 atoms).  So the
// loads go one per block (ring_drip, called in front of a block's MFMAs: 0.308 ms; with a burst of the first two or four
// pieces 0.311 / 0.324), and whatever is left at the next boundary is issued there (ring_flush), before its wait.
// -DANI_FUSED_NODRIP restores the burst.
#ifndef ANI_FUSED_NODRIP
#define ANI_FUSED_DRIP
#endif
#ifndef ANI_FUSED_DRIP_HEAD
#define ANI_FUSED_DRIP_HEAD 0
#endif
__device__ __forceinline__ void ring_drip(Ring& r, unsigned char* ring, int lane16) {
#ifdef ANI_FUSED_DRIP
  if (r.pk < r.pn) {
    ring_load_piece(r, ring + r.pslot * (kSlot << 10), r.pq0, r.pk, lane16);
    r.pk += 4;
  }
#endif
}
__device__ __forceinline__ void ring_flush(Ring& r, unsigned char* ring, int lane16) {
#ifdef ANI_FUSED_DRIP
  while (r.pk < r.pn) {
    ring_load_piece(r, ring + r.pslot * (kSlot << 10), r.pq0, r.pk, lane16);
    r.pk += 4;
  }
#endif
}
template <int NT1, int NT2, int NT3, int P>
__device__ __forceinline__ void ring_issue_next(Ring& r, unsigned char* ring, int wave, int lane16) {
  int q0, n, slot;
  if (ring_take<NT1, NT2, NT3, P>(r, q0, n, slot)) {
    unsigned char* base = ring + slot * (kSlot << 10);
#ifdef ANI_FUSED_DRIP
    int k = wave;
    for (int i = 0; i < ANI_FUSED_DRIP_HEAD && k < n; i++, k += 4) ring_load_piece(r, base, q0, k, lane16);
    r.pq0 = q0; r.pslot = slot; r.pn = n; r.pk = k;
#else
    for (int k = wave; k < n; k += 4) ring_load_piece(r, base, q0, k, lane16);
#endif
  }
}
// In front of a slab: returns this lane's read address of the slab's first piece.  Every wave waits for its own loads (the
// slab's pieces among them, issued a slab ago), the barrier says that everybody's have landed and that nobody reads the slab
// two back any more, whose slot the next slab's loads -- issued right behind the barrier -- overwrite.  EARLY marks the calls
// that stand in front of the LAST block of the slab before (the new slab's first fragments are then requested beside that
// block's MFMAs instead of after them); the scheme is the same for both.
__device__ __forceinline__ constexpr bool ring_can_go_early(const Ring&, int) { return true; }
template <int NT1, int NT2, int NT3, int P, bool EARLY>
__device__ __forceinline__ const unsigned char* ring_boundary(Ring& r, unsigned char* ring, int n, int wave, int lane16, int* err_flag) {
  (void)n; (void)err_flag;
  ring_flush(r, ring, lane16);
#ifndef ABLF_NOWAIT
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#ifndef ABLF_NOBAR
  asm volatile("s_barrier" ::: "memory");
#else
  asm volatile("" ::: "memory");
#endif
  const int s = ring_consume(r);
  ring_issue_next<NT1, NT2, NT3, P>(r, ring, wave, lane16);
  return ring + s * (kSlot << 10) + lane16;
}
// The boundary of a slab that follows another of the same product stands in front of the last block of the slab being read
// (FUSED_NEXT_SLAB_EARLY, ahead of that block's MFMAs); FUSED_NEXT_SLAB_LATE, behind them, is where version 4's ring had to
// fall back to when the slab was not on its way yet -- with fixed slots it always is.
#define FUSED_NEXT_SLAB_EARLY(n_, frag_)                                                            \
  base = ring_boundary<RING_T, true>(r, ring, (n_), wave, lane16, err);                             \
  read_frag<P>(base, 0, frag_);                                                                     \
  sched_first_read<P>();
#define FUSED_NEXT_SLAB_LATE(n_, frag_)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int P>
struct FragV { u32x4 p[P]; };
// block b (P pieces) of the slab at `base`
template <int P>
__device__ __forceinline__ void read_frag(const unsigned char* base, int b, FragV<P>& f) {
#ifdef ABLF_NOLDS   // timing experiment only: no fragment reads (a value the compiler cannot fold)
#pragma unroll
  for (int p = 0; p < P; p++) { unsigned x = (unsigned)(size_t)base + b; asm volatile("" : "+v"(x)); f.p[p] = u32x4{x, x, x, x}; }
  return;
#endif
#pragma unroll
  for (int p = 0; p < P; p++) f.p[p] = *reinterpret_cast<const u32x4*>(base + (b * P + p) * 1024);
}
template <int P>
__device__ __forceinline__ void mma_frag(const FragV<P>& a, const FragV<P>& b, f32x16& acc) {
#ifdef ABLF_NOMMA   // timing experiment only: operands kept alive, no matrix instruction
#pragma unroll
  for (int i = 0; i < P; i++) asm volatile("" ::"v"(a.p[i]), "v"(b.p[i]));
  return;
#endif
  if constexpr (P == 3) {   // smallest terms first
    const bf16x8 ah = __builtin_bit_cast(bf16x8, a.p[0]), am = __builtin_bit_cast(bf16x8, a.p[1]), al = __builtin_bit_cast(bf16x8, a.p[2]);
    const bf16x8 bh = __builtin_bit_cast(bf16x8, b.p[0]), bm = __builtin_bit_cast(bf16x8, b.p[1]), bl = __builtin_bit_cast(bf16x8, b.p[2]);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
  } else {
    const f16x8 ah = __builtin_bit_cast(f16x8, a.p[0]), al = __builtin_bit_cast(f16x8, a.p[1]);
    const f16x8 bh = __builtin_bit_cast(f16x8, b.p[0]), bl = __builtin_bit_cast(f16x8, b.p[1]);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
  }
}

// "these values exist here": the compiler may not sink the (register-only) arithmetic that makes them past this point.  It
// does otherwise -- a fragment converted during k-step ks for k-step ks + 1 ends up converted at the top of k-step ks + 1,
// in front of the MFMAs that wait for it, with the matrix pipe idle meanwhile.
template <int P>
__device__ __forceinline__ void pin(FragV<P>& f) {
#pragma unroll
  for (int p = 0; p < P; p++) asm volatile("" : "+v"(f.p[p]));
}
__device__ __forceinline__ void pin(f32x16& t) { asm volatile("" : "+v"(t)); }

// two fp32 values (elements 2 i, 2 i + 1 of a k-step's fragment) -> word i of every plane
template <int P>
__device__ __forceinline__ void split_pair(float x0, float x1, float a_scale, int i, FragV<P>& f) {
#ifdef ABLF_NOSPLIT   // timing experiment only: no conversion arithmetic (one move per plane)
#pragma unroll
  for (int p = 0; p < P; p++) f.p[p][i] = __float_as_uint(x0);
  return;
#endif
  if constexpr (P == 3) {
    unsigned h0, m0, l0, h1, m1, l1;
    split3(x0, h0, m0, l0);
    split3(x1, h1, m1, l1);
    f.p[0][i] = pack_hi16(h0, h1); f.p[1][i] = pack_hi16(m0, m1); f.p[2][i] = pack_hi16(l0, l1);
  } else {
    unsigned h, l;
    split2(x0 * a_scale, x1 * a_scale, h, l);
    f.p[0][i] = h; f.p[1][i] = l;
  }
}
// pair i of k-step ks of X (k-step ks = registers 8 (ks & 1) .. + 7 of tile ks >> 1)
template <int P, int NTX>
__device__ __forceinline__ void split_pair_of(const f32x16 (&X)[NTX], int ks, int i, float a_scale, FragV<P>& f) {
  split_pair<P>(X[ks >> 1][8 * (ks & 1) + 2 * i], X[ks >> 1][8 * (ks & 1) + 2 * i + 1], a_scale, i, f);
}

template <int NT>
__device__ __forceinline__ void zero_tiles(f32x16 (&X)[NT]) {
#pragma unroll
  for (int nt = 0; nt < NT; nt++)
#pragma unroll
    for (int r = 0; r < 16; r++) X[nt][r] = 0.f;
}

#define RING_T NT1, NT2, NT3, P   // the shape parameters the ring's slab sequence depends on

// Order of a block's instructions, pinned (LLVM sched_group_barrier): the P fragment reads of the NEXT block first, then the
// block's MFMAs one at a time with up to four vector instructions behind each.  A wave issues in order and is alone on its
// SIMD: left to itself the compiler (short of registers) reads every fragment right in front of the MFMA that needs it
// -- an LDS round trip per two MFMAs -- and bunches the conversions and epilogues between the MFMA groups, where the matrix
// pipe waits for them.
#ifndef ANI_FUSED_VALU_PER_MFMA
#define ANI_FUSED_VALU_PER_MFMA 4
#endif
template <int P>
__device__ __forceinline__ void sched_first_read() {   // the slab's first block has nobody to be requested behind
#ifndef ABLF_NOSCHED
  __builtin_amdgcn_sched_group_barrier(0x100, P, 0);
#endif
}
template <int P, bool READ>
__device__ __forceinline__ void sched_block() {
#ifndef ABLF_NOSCHED
  if constexpr (READ) __builtin_amdgcn_sched_group_barrier(0x100, P, 0);
#pragma unroll
  for (int i = 0; i < (P == 3 ? 6 : 3); i++) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, ANI_FUSED_VALU_PER_MFMA, 0);
  }
#ifdef ANI_FUSED_BLOCK_FENCE   // experiment: nothing moves across the end of a block
  __builtin_amdgcn_sched_barrier(0);
#endif
#endif
}

// acc[NT] += W (stream) * X  with X in registers: KS k-steps, two per slab.  The fragments of block i + 1 are requested before
// the MFMAs of block i (two register sets, statically alternated), across slabs too; the 16-bit terms of k-step ks + 1 are
// made a pair per block during k-step ks.
template <int NT1, int NT2, int NT3, int KS, int NT, int NTX, int P>
__device__ __forceinline__ void product_reg(Ring& r, unsigned char* ring, const f32x16 (&X)[NTX], f32x16 (&acc)[NT], float a_scale,
                                            int wave, int lane16, int* err) {
  static_assert(KS <= 2 * NTX && KS % 2 == 0, "k-steps beyond the source tiles");
  constexpr int SB = 2 * NT;   // blocks per slab
  FragV<P> bq[2], fa[2];
#pragma unroll
  for (int i = 0; i < 4; i++) split_pair_of<P>(X, 0, i, a_scale, bq[0]);
  const unsigned char* base = ring_boundary<RING_T, false>(r, ring, SB * P, wave, lane16, err);
  read_frag<P>(base, 0, fa[0]);
  sched_first_read<P>();
#pragma unroll
  for (int kp = 0; kp < KS / 2; kp++) {
#pragma unroll
    for (int j = 0; j < SB; j++) {
      const int ks = 2 * kp + j / NT, nt = j % NT, idx = kp * SB + j;
      const bool next_slab = j + 1 == SB && kp + 1 < KS / 2;
      if (j + 1 < SB) read_frag<P>(base, j + 1, fa[(idx + 1) & 1]);
      if (ks + 1 < KS) {
#pragma unroll
        for (int i = 0; i < 4; i++)
          if ((i < NT ? i : NT - 1) == nt) split_pair_of<P>(X, ks + 1, i, a_scale, bq[(ks + 1) & 1]);
      }
      if (next_slab) {
        FUSED_NEXT_SLAB_EARLY(SB * P, fa[(idx + 1) & 1])
        ring_drip(r, ring, lane16); mma_frag<P>(fa[idx & 1], bq[ks & 1], acc[nt]);
        sched_block<P, false>();
        FUSED_NEXT_SLAB_LATE(SB * P, fa[(idx + 1) & 1])
      } else {
        ring_drip(r, ring, lane16); mma_frag<P>(fa[idx & 1], bq[ks & 1], acc[nt]);
        if (j + 1 < SB) sched_block<P, true>(); else sched_block<P, false>();
      }
      if (nt == NT - 1 && ks + 1 < KS) pin<P>(bq[(ks + 1) & 1]);
    }
  }
}

// The hidden backward products, in place:  Y[nt] <- (sum_ks W[nt][ks] X[ks]) * inv * celu'(z[nt]),  Y holding the stored
// activation (through which celu' is known) on entry and the gradient on exit -- no second set of tiles.  The source X is
// split into its 16-bit terms once (it is dead afterwards: 12 registers per k-step instead of 8), during the first output
// tile's pass; then one output tile at a time runs through all k-steps on one accumulator (two, alternated: the masking of
// tile nt - 1 sits among the first MFMAs of tile nt).  A slab is the KS blocks of an output tile.
template <int NT1, int NT2, int NT3, int KS, int NT, int NTX, int P>
__device__ __forceinline__ void product_inplace(Ring& r, unsigned char* ring, const f32x16 (&X)[NTX], f32x16 (&Y)[NT], float a_scale,
                                                float inv, float inv_alpha, int wave, int lane16, int* err) {
  static_assert(KS <= 2 * NTX, "k-steps beyond the source tiles");
  FragV<P> bf[KS], fa[2];
#pragma unroll
  for (int i = 0; i < 4; i++) split_pair_of<P>(X, 0, i, a_scale, bf[0]);
  f32x16 acc[2];
  const unsigned char* base = ring_boundary<RING_T, false>(r, ring, KS * P, wave, lane16, err);
  read_frag<P>(base, 0, fa[0]);
  sched_first_read<P>();
#pragma unroll
  for (int nt = 0; nt < NT; nt++) {
#pragma unroll
    for (int i = 0; i < 16; i++) acc[nt & 1][i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      const int idx = nt * KS + ks;
      const bool next_slab = ks + 1 == KS && nt + 1 < NT;
      if (ks + 1 < KS) read_frag<P>(base, ks + 1, fa[(idx + 1) & 1]);
      if (nt == 0 && ks + 1 < KS) {
#pragma unroll
        for (int i = 0; i < 4; i++) split_pair_of<P>(X, ks + 1, i, a_scale, bf[ks + 1]);
      }
      if (next_slab) {
        FUSED_NEXT_SLAB_EARLY(KS * P, fa[(idx + 1) & 1])
        ring_drip(r, ring, lane16); mma_frag<P>(fa[idx & 1], bf[ks], acc[nt & 1]);
        FUSED_NEXT_SLAB_LATE(KS * P, fa[(idx + 1) & 1])
      } else {
        ring_drip(r, ring, lane16); mma_frag<P>(fa[idx & 1], bf[ks], acc[nt & 1]);
      }
      if (ks == 1 && nt > 0) {
#pragma unroll
        for (int i = 0; i < 16; i++) Y[nt - 1][i] = acc[(nt - 1) & 1][i] * inv * dcelu_from_h(Y[nt - 1][i], inv_alpha);
      }
      if (ks + 1 < KS) sched_block<P, true>(); else sched_block<P, false>();
      if (nt == 0 && ks + 1 < KS) pin<P>(bf[ks + 1]);
      if (ks == 3 && nt > 0) pin(Y[nt - 1]);
    }
  }
#pragma unroll
  for (int i = 0; i < 16; i++) Y[NT - 1][i] = acc[(NT - 1) & 1][i] * inv * dcelu_from_h(Y[NT - 1][i], inv_alpha);
}

// constants of a member in LDS (floats): b0[32 NT1] b1[32 NT2] b2[32 NT3] w3[32 NT3] then {b3, inv[6]}
template <int NT1, int NT2, int NT3>
struct ConstLayout {
  static constexpr int b0 = 0, b1 = 32 * NT1, b2 = b1 + 32 * NT2, w3 = b2 + 32 * NT3, tail = w3 + 32 * NT3, count = tail + 8;
};

template <int NT>
__device__ __forceinline__ void epilogue_celu(f32x16 (&X)[NT], const float* b, int h, float inv, float alpha, float inv_alpha) {
#ifdef ABLF_NOEPI   // timing experiment only
  return;
#endif
#pragma unroll
  for (int nt = 0; nt < NT; nt++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const float4 bv = *reinterpret_cast<const float4*>(b + 32 * nt + 8 * q + 4 * h);
      X[nt][4 * q + 0] = celu_f(fmaf(X[nt][4 * q + 0], inv, bv.x), alpha, inv_alpha);
      X[nt][4 * q + 1] = celu_f(fmaf(X[nt][4 * q + 1], inv, bv.y), alpha, inv_alpha);
      X[nt][4 * q + 2] = celu_f(fmaf(X[nt][4 * q + 2], inv, bv.z), alpha, inv_alpha);
      X[nt][4 * q + 3] = celu_f(fmaf(X[nt][4 * q + 3], inv, bv.w), alpha, inv_alpha);
    }
}

struct TileCtx {
  float alpha, inv_alpha, scale, a_fwd, a_bwd;
  int M;
  int m0, m1;          // members this work item runs
  float* parts;        // member_items: per-member dE/dAEV buffers, else null
  long long part_stride;
  int* err;   // device error word (bit 4: the ring schedule broke)
};

// One 128-row tile of a species bucket, all members.  Shape (NT1, NT2, NT3): 32-feature tiles of the three hidden layers
// (widths padded with zero weights).
template <int NT1, int NT2, int NT3, int P>
__device__ __forceinline__ void fused_tile(const TileCtx& cx, const FusedProb& pr, int tile, int wave, int lane, unsigned char* ring,
                                           float* cst) {
  using CL = ConstLayout<NT1, NT2, NT3>;
  static_assert(CL::count * 4 <= kConstBytes, "constants do not fit their LDS block");
  static_assert(2 * NT2 * P <= kSlot && 2 * NT3 * P <= kSlot && F1Slab<NT1, P>::k * NT1 * P <= kSlot && 2 * kChunk * P <= kSlot,
                "every slab must fit a slot");
  const int c = lane & 31, h = lane >> 5, lane16 = lane * 16;
  int* const err = cx.err;
  const int row = tile * 128 + 32 * wave + c;
  const float valid = pr.centre_of_row[row] >= 0 ? cx.scale : 0.f;
  const float* __restrict__ arow = pr.aev + (size_t)row * pr.aev_stride + 4 * h;
  // dE/dAEV rows: the step's array (members accumulate one after the other), or this member's own copy of it (work items
  // per member: pr.gaev_row0 is the problem's first row in the step's array, the same offset applies in every copy)
  float* __restrict__ grow = (cx.parts ? cx.parts + (size_t)cx.m0 * cx.part_stride + (size_t)pr.gaev_row0 * pr.aev_stride : pr.gaev) +
                             (size_t)row * pr.aev_stride + 4 * h;
  Ring r;
  ring_reset<RING_T>(r, pr.stream + (size_t)cx.m0 * pr.pieces_per_member * 1024, pr.pieces_per_member * (cx.m1 - cx.m0), pr.ks0, pr.nt0);
  FUSED_STAMP_INIT();
  ring_issue_next<RING_T>(r, ring, wave, lane16);   // slab 0; every later slab is issued at the boundary of the one before

  for (int m = cx.m0; m < cx.m1; m++) {
    // the member's constants: one piece per wave, then everything issued so far is waited for (the ring's first slabs among
    // it: they are needed next anyway)
    {
      const int nconst = (CL::count * 4 + 1023) >> 10;
      if (wave < nconst) {
        const unsigned char* g = reinterpret_cast<const unsigned char*>(pr.consts + (size_t)m * pr.consts_per_member) + wave * 1024 + lane16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(reinterpret_cast<unsigned char*>(cst) + wave * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const float inv_f1 = cst[CL::tail + 1], inv_f2 = cst[CL::tail + 2], inv_f3 = cst[CL::tail + 3];
    const float inv_b3 = cst[CL::tail + 4], inv_b2 = cst[CL::tail + 5], inv_b1 = cst[CL::tail + 6];
    const float b3 = cst[CL::tail];
    FUSED_STAMP(0);

    // ---- F1: h1 = celu(W0 aev + b0); the B operand streams from the AEV rows: K1 k-steps (one slab) per iteration, the
    // rows' values requested a slab ahead ----
    f32x16 X1[NT1];
    zero_tiles(X1);
    {
      constexpr int K1 = F1Slab<NT1, P>::k;
      const int ks0 = pr.ks0, nslab = (ks0 + K1 - 1) / K1;
      const float4 z4 = make_float4(0, 0, 0, 0);
      auto ld = [&](int ks, float4& a, float4& b) {
        if (ks < ks0) { a = *reinterpret_cast<const float4*>(arow + 16 * ks); b = *reinterpret_cast<const float4*>(arow + 16 * ks + 8); }
      };
      auto cvt = [&](int i, const float4& a, const float4& b, FragV<P>& f) {   // pair i of the eight values (a, b)
        if (i == 0) split_pair<P>(a.x, a.y, cx.a_fwd, 0, f);
        if (i == 1) split_pair<P>(a.z, a.w, cx.a_fwd, 1, f);
        if (i == 2) split_pair<P>(b.x, b.y, cx.a_fwd, 2, f);
        if (i == 3) split_pair<P>(b.z, b.w, cx.a_fwd, 3, f);
      };
      float4 c0a = z4, c0b = z4, v1a = z4, v1b = z4, v2a = z4, v2b = z4, w1a = z4, w1b = z4, w2a = z4, w2b = z4;
      ld(0, c0a, c0b); ld(1, v1a, v1b);
      if (K1 == 2) ld(2, v2a, v2b);
      FragV<P> b0, b1, b0n, fa[2];
#pragma unroll
      for (int i = 0; i < 4; i++) cvt(i, c0a, c0b, b0);
      b1 = b0; b0n = b0;
      const unsigned char* base = ring_boundary<RING_T, false>(r, ring, min(K1, ks0) * NT1 * P, wave, lane16, cx.err);
      read_frag<P>(base, 0, fa[0]);
      sched_first_read<P>();
      if constexpr (K1 == 2) {
        for (int kp = 0; kp < nslab; kp++) {
          const bool two = 2 * kp + 1 < ks0;           // the slab holds two k-steps (all but an odd last one)
          ld(2 * kp + 3, w1a, w1b); ld(2 * kp + 4, w2a, w2b);
          // k-step 2 kp: blocks 0 .. NT1 - 1; b1 (k-step 2 kp + 1) is made meanwhile
#pragma unroll
          for (int nt = 0; nt < NT1; nt++) {
            if (nt + 1 < NT1 || two) read_frag<P>(base, nt + 1, fa[(nt + 1) & 1]);
            cvt(nt, v1a, v1b, b1);
            ring_drip(r, ring, lane16); mma_frag<P>(fa[nt & 1], b0, X1[nt]);
            sched_block<P, true>();
          }
          pin<P>(b1);
          if (two) {
            // k-step 2 kp + 1: blocks NT1 .. 2 NT1 - 1; b0n (k-step 2 kp + 2) is made meanwhile; the next slab's first
            // fragments are requested in front of the last block
#pragma unroll
            for (int nt = 0; nt < NT1; nt++) {
              const int j = NT1 + nt;
              if (nt + 1 < NT1) read_frag<P>(base, j + 1, fa[(j + 1) & 1]);
              cvt(nt, v2a, v2b, b0n);
              if (nt + 1 == NT1 && kp + 1 < nslab) {
                const int nn = min(2, ks0 - 2 * (kp + 1)) * NT1 * P;
                FUSED_NEXT_SLAB_EARLY(nn, fa[0])      // (j + 1) & 1 == 0: block 0 of the next slab
                ring_drip(r, ring, lane16); mma_frag<P>(fa[j & 1], b1, X1[nt]);
                sched_block<P, false>();
                FUSED_NEXT_SLAB_LATE(nn, fa[0])
              } else {
                ring_drip(r, ring, lane16); mma_frag<P>(fa[j & 1], b1, X1[nt]);
                if (nt + 1 < NT1) sched_block<P, true>(); else sched_block<P, false>();
              }
            }
            pin<P>(b0n);
          }
          b0 = b0n;
          v1a = w1a; v1b = w1b; v2a = w2a; v2b = w2b;
        }
      } else {
        // one k-step per slab (two of them would not fit the ring three times).  NT1 is even here, so block 0 of every slab
        // lands in fragment set 0
        static_assert(K1 == 2 || NT1 % 2 == 0, "fragment sets alternate per block");
        for (int ks = 0; ks < ks0; ks++) {
          ld(ks + 2, w1a, w1b);
#pragma unroll
          for (int nt = 0; nt < NT1; nt++) {
            if (nt + 1 < NT1) read_frag<P>(base, nt + 1, fa[(nt + 1) & 1]);
            cvt(nt, v1a, v1b, b0n);       // k-step ks + 1
            if (nt + 1 == NT1 && ks + 1 < ks0) {
              FUSED_NEXT_SLAB_EARLY(NT1 * P, fa[0])
              ring_drip(r, ring, lane16); mma_frag<P>(fa[nt & 1], b0, X1[nt]);
              sched_block<P, false>();
              FUSED_NEXT_SLAB_LATE(NT1 * P, fa[0])
            } else {
              ring_drip(r, ring, lane16); mma_frag<P>(fa[nt & 1], b0, X1[nt]);
              if (nt + 1 < NT1) sched_block<P, true>(); else sched_block<P, false>();
            }
          }
          pin<P>(b0n);
          b0 = b0n;
          v1a = w1a; v1b = w1b;
        }
      }
    }
    FUSED_STAMP(1);
    epilogue_celu(X1, cst + CL::b0, h, inv_f1, cx.alpha, cx.inv_alpha);
    FUSED_STAMP(2);

    // ---- F2, F3 ----
    f32x16 X2[NT2];
    zero_tiles(X2);
    product_reg<NT1, NT2, NT3, 2 * NT1, NT2, NT1, P>(r, ring, X1, X2, cx.a_fwd, wave, lane16, cx.err);
    FUSED_STAMP(3);
    epilogue_celu(X2, cst + CL::b1, h, inv_f2, cx.alpha, cx.inv_alpha);
    FUSED_STAMP(4);

    f32x16 X3[NT3];
    zero_tiles(X3);
    product_reg<NT1, NT2, NT3, 2 * NT2, NT3, NT2, P>(r, ring, X2, X3, cx.a_fwd, wave, lane16, cx.err);
    FUSED_STAMP(5);
    // last hidden layer fused with the 1-wide output layer and the backward seed dE/dz3 = (1/M) w3 celu'(z3)
    {
      float es = 0.f;
#pragma unroll
      for (int nt = 0; nt < NT3; nt++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const float4 bv = *reinterpret_cast<const float4*>(cst + CL::b2 + 32 * nt + 8 * q + 4 * h);
          const float4 wv = *reinterpret_cast<const float4*>(cst + CL::w3 + 32 * nt + 8 * q + 4 * h);
          const float bb[4] = {bv.x, bv.y, bv.z, bv.w}, ww[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const float hv = celu_f(fmaf(X3[nt][4 * q + i], inv_f3, bb[i]), cx.alpha, cx.inv_alpha);
            es = fmaf(hv, ww[i], es);
            X3[nt][4 * q + i] = valid * ww[i] * dcelu_from_h(hv, cx.inv_alpha);
          }
        }
      es += __shfl_xor(es, 32);
      if (h == 0) pr.e_rows[(size_t)m * pr.sE + row] = valid * (es + b3);
    }

    FUSED_STAMP(6);
    // ---- B3: g2 = (W2^T g3) celu'(z2), written over h2 ; B2: g1 = (W1^T g2) celu'(z1), written over h1 ----
    product_inplace<NT1, NT2, NT3, 2 * NT3, NT2, NT3, P>(r, ring, X3, X2, cx.a_bwd, inv_b3, cx.inv_alpha, wave, lane16, cx.err);
    FUSED_STAMP(7);
    product_inplace<NT1, NT2, NT3, 2 * NT2, NT1, NT2, P>(r, ring, X2, X1, cx.a_bwd, inv_b2, cx.inv_alpha, wave, lane16, cx.err);
    FUSED_STAMP(8);
    f32x16 (&G1)[NT1] = X1;

    // ---- B1: dE/dAEV = W0^T g1, kChunk 32-column tiles at a time, two k-steps per slab; members after the first add to
    // what is there ----
    for (int c0 = 0; c0 < pr.nt0; c0 += kChunk) {
      const int ntc = min(kChunk, pr.nt0 - c0);
      f32x16 acc[kChunk];
      zero_tiles(acc);
      FragV<P> bq[2], fa[2];
#pragma unroll
      for (int i = 0; i < 4; i++) split_pair_of<P>(G1, 0, i, cx.a_bwd, bq[0]);
      const unsigned char* base = ring_boundary<RING_T, false>(r, ring, 2 * ntc * P, wave, lane16, cx.err);
      if (ntc == kChunk) {   // the usual case, free of branches inside a slab
        constexpr int SB = 2 * kChunk;
        read_frag<P>(base, 0, fa[0]);
        sched_first_read<P>();
#pragma unroll
        for (int kp = 0; kp < NT1; kp++) {
#pragma unroll
          for (int j = 0; j < SB; j++) {
            const int ks = 2 * kp + j / kChunk, t = j % kChunk, idx = kp * SB + j;
            if (j + 1 < SB) read_frag<P>(base, j + 1, fa[(idx + 1) & 1]);
            if (ks + 1 < 2 * NT1) split_pair_of<P>(G1, ks + 1, t, cx.a_bwd, bq[(ks + 1) & 1]);
            if (j + 1 == SB && kp + 1 < NT1) {
              FUSED_NEXT_SLAB_EARLY(SB * P, fa[(idx + 1) & 1])
              ring_drip(r, ring, lane16); mma_frag<P>(fa[idx & 1], bq[ks & 1], acc[t]);
              sched_block<P, false>();
              FUSED_NEXT_SLAB_LATE(SB * P, fa[(idx + 1) & 1])
            } else {
              ring_drip(r, ring, lane16); mma_frag<P>(fa[idx & 1], bq[ks & 1], acc[t]);
              if (j + 1 < SB) sched_block<P, true>(); else sched_block<P, false>();
            }
            if (t == kChunk - 1 && ks + 1 < 2 * NT1) pin<P>(bq[(ks + 1) & 1]);
          }
        }
      } else {               // a last, narrower chunk: plain loop, a boundary per slab
#pragma unroll
        for (int kp = 0; kp < NT1; kp++) {
          if (kp > 0) base = ring_boundary<RING_T, false>(r, ring, 2 * ntc * P, wave, lane16, cx.err);
#pragma unroll
          for (int s2 = 0; s2 < 2; s2++) {
            const int ks = 2 * kp + s2;
#pragma unroll
            for (int t = 0; t < kChunk; t++) {
              if (t < ntc) {
                read_frag<P>(base, s2 * ntc + t, fa[0]);
                mma_frag<P>(fa[0], bq[ks & 1], acc[t]);
              }
            }
            if (ks + 1 < 2 * NT1) {
#pragma unroll
              for (int i = 0; i < 4; i++) split_pair_of<P>(G1, ks + 1, i, cx.a_bwd, bq[(ks + 1) & 1]);
            }
          }
        }
      }
#pragma unroll
      for (int t = 0; t < kChunk; t++) {
        if (t < ntc) {
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int f0 = 32 * (c0 + t) + 8 * q;   // + 4 h is in grow
            if (f0 + 4 * h < pr.acols) {
              float4 o = make_float4(acc[t][4 * q] * inv_b1, acc[t][4 * q + 1] * inv_b1, acc[t][4 * q + 2] * inv_b1, acc[t][4 * q + 3] * inv_b1);
              float4* dst = reinterpret_cast<float4*>(grow + f0);
              if (m > cx.m0) { const float4 old = *dst; o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
              *dst = o;
            }
          }
        }
      }
    }
    FUSED_STAMP(9);
  }
}
#undef RING_T

template <int P>
__global__ __launch_bounds__(256, 1) void mlp_fused(FusedArgs G) {
  extern __shared__ uint4 smem4[];
  unsigned char* ring = reinterpret_cast<unsigned char*>(smem4);
  float* cst = reinterpret_cast<float*>(ring + kRing * 1024);
  __shared__ int s_tile;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  TileCtx cx;
  cx.alpha = G.alpha; cx.inv_alpha = G.inv_alpha; cx.scale = G.scale; cx.M = G.M; cx.err = G.err_flag;
  cx.a_fwd = P == 2 ? 16.f : 1.f; cx.a_bwd = P == 2 ? 4096.f : 1.f;
  cx.m0 = 0; cx.m1 = G.M; cx.parts = nullptr; cx.part_stride = 0;
  const int per_tile = G.member_items ? G.M : 1;   // work items per tile (members side by side: they share the AEV rows in L2)
  const int total = G.tile_start[G.nprob] * per_tile;
  int sched_pos = 0;
  for (;;) {
    __syncthreads();   // every wave is done with the tile before (ring, constants, s_tile)
    if (G.sched_items) {   // static schedule: this workgroup's list
      if (threadIdx.x == 0) {
        const int i = G.sched_off[blockIdx.x] + sched_pos;
        s_tile = i < G.sched_off[blockIdx.x + 1] ? G.sched_items[i] : total;
      }
      sched_pos++;
    } else if (threadIdx.x == 0) s_tile = atomicAdd(G.counter, 1);
    __syncthreads();
    const int item = __builtin_amdgcn_readfirstlane(s_tile);
    if (item >= total) break;
    const int t = item / per_tile;
    if (G.member_items) {
      cx.m0 = item - t * per_tile; cx.m1 = cx.m0 + 1;
      cx.parts = G.gaev_parts; cx.part_stride = G.part_stride;
    }
    int pi = 0;
    while (pi + 1 < G.nprob && t >= G.tile_start[pi + 1]) pi++;
    const FusedProb& pr = G.p[pi];
    const int tile = t - G.tile_start[pi];
    switch (pr.shape) {
      case 0: fused_tile<8, 6, 5, P>(cx, pr, tile, wave, lane, ring, cst); break;
      case 1: fused_tile<6, 5, 4, P>(cx, pr, tile, wave, lane, ring, cst); break;
      default: fused_tile<5, 4, 3, P>(cx, pr, tile, wave, lane, ring, cst); break;
    }
  }
}

const int kShapes[3][3] = {{8, 6, 5}, {6, 5, 4}, {5, 4, 3}};

int fused_shape_for(int d1, int d2, int d3) {
  for (int s = 2; s >= 0; s--)
    if (d1 <= 32 * kShapes[s][0] && d2 <= 32 * kShapes[s][1] && d3 <= 32 * kShapes[s][2]) return s;
  return -1;
}
void fused_shape_tiles(int shape, int nt[3]) { for (int k = 0; k < 3; k++) nt[k] = kShapes[shape][k]; }
int fused_consts_floats(int shape) {
  const int* s = kShapes[shape];
  const int n = 32 * (s[0] + s[1] + 2 * s[2]) + 8;
  return (n + 255) / 256 * 256;   // whole 1 KB pieces
}
long long fused_pieces_per_member(int shape, int acols, int P) {
  const int* s = kShapes[shape];
  const long long ks0 = acols / 16, nt0 = (acols + 31) / 32;
  return P * (ks0 * s[0] + 2LL * s[0] * s[1] + 2LL * s[1] * s[2] + 2LL * s[2] * s[1] + 2LL * s[1] * s[0] + 2LL * s[0] * nt0);
}

int fused_read_stamps(unsigned long long* out16, int reset) {
#ifdef ABLF_STAMPS
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_fused_stamps), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_fused_stamps), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 1;
#else
  (void)out16; (void)reset;
  return 0;
#endif
}

__global__ __launch_bounds__(256) void sum_parts_kernel(const float4* __restrict__ parts, long long stride4, int M, float4* __restrict__ dst,
                                                        long long n4) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 a = parts[i];
  for (int m = 1; m < M; m++) {
    const float4 b = parts[(long long)m * stride4 + i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  dst[i] = a;
}
void launch_sum_parts(const float* parts, long long part_stride, int M, float* dst, long long n, hipStream_t st) {
  const long long n4 = n / 4;
  if (n4 <= 0) return;
  hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const float4*>(parts),
                     part_stride / 4, M, reinterpret_cast<float4*>(dst), n4);
}

void launch_build_stream(const float* src, int ld, int rows_valid, int k_valid, int NT, int KS, int chunk, int P, float scale,
                         unsigned short* dst, hipStream_t st) {
  const long long total = (long long)NT * KS * 512;
  if (total <= 0) return;
  hipLaunchKernelGGL(build_stream_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, ld, rows_valid, k_valid, NT, KS,
                     chunk, P, scale, dst);
}

int fused_num_cus() {
  static int ncu = [] {
    int dev = 0, v = 256;
    if (hipGetDevice(&dev) == hipSuccess) note_launch_error(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
    return v > 0 ? v : 256;
  }();
  return ncu;
}

namespace {
// multifit: the smallest makespan T for which first-fit-decreasing packs every item; take[b * ntypes + j] = items of type j in bin b
double fused_pack(int ntypes, const int* count, const double* cost, int bins, std::vector<int>& best) {
  std::vector<int> order(ntypes);
  for (int j = 0; j < ntypes; j++) order[j] = j;
  std::sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
  double total = 0.0, cmax = 0.0;
  for (int j = 0; j < ntypes; j++) { total += count[j] * cost[j]; if (count[j] > 0) cmax = std::max(cmax, cost[j]); }
  std::vector<int> take((size_t)bins * ntypes);
  auto fits = [&](double T) {
    std::fill(take.begin(), take.end(), 0);
    std::vector<double> rem(bins, T);
    for (int jj = 0; jj < ntypes; jj++) {
      const int j = order[jj];
      int left = count[j];
      if (left == 0 || cost[j] <= 0.0) { if (left) { take[j] += left; } continue; }
      for (int b = 0; b < bins && left > 0; b++) {
        const int k = std::min(left, (int)((rem[b] + 1e-9) / cost[j]));
        if (k > 0) { take[(size_t)b * ntypes + j] = k; rem[b] -= k * cost[j]; left -= k; }
      }
      if (left > 0) return false;
    }
    return true;
  };
  double lo = std::max(total / bins, cmax), hi = lo;
  while (!fits(hi)) hi *= 1.25;
  best = take;
  for (int it = 0; it < 24 && hi - lo > 1e-3 * hi; it++) {
    const double mid = 0.5 * (lo + hi);
    if (fits(mid)) { hi = mid; best = take; } else lo = mid;
  }
  return hi;
}
}  // namespace

double fused_schedule(int ntypes, const int* count, const double* cost, int bins, int* items_out, int* off_out) {
  std::vector<int> best, first(ntypes + 1, 0), order(ntypes);
  for (int j = 0; j < ntypes; j++) { order[j] = j; first[j + 1] = first[j] + count[j]; }
  std::sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
  const double T = fused_pack(ntypes, count, cost, bins, best);
  std::vector<int> next(first.begin(), first.end() - 1);
  int n = 0;
  for (int b = 0; b < bins; b++) {
    off_out[b] = n;
    for (int jj = 0; jj < ntypes; jj++) {
      const int j = order[jj];
      for (int k = 0; k < best[(size_t)b * ntypes + j]; k++) items_out[n++] = next[j]++;
    }
  }
  off_out[bins] = n;
  return T;
}

// The same with HALF items (the sixteen-row kernel: item total + 2 i + h = half h of item i, run by the lower half of a
// workgroup's waves at half_ratio of the item's cost -- more than half: the weights stream through the workgroup all the same).
// The last split[j] items of type j are cut in two where that shortens the schedule: the items beyond the last full round of
// workgroups otherwise make a round of their own with most of the chip idle.  split_mode 1: searched (a few candidate counts per
// type, most expensive types first, two sweeps; kept only if the makespan falls by min_gain -- the cost model is good to about
// 5 %: the caller either asks for 8 % or times the candidate against the whole items), 2: every item (tests, measurements).
// items_out holds up to sum(count) + max splits entries.  Returns the makespan; *n_items_out = entries written.
double fused_schedule_halves(int ntypes, const int* count, const double* cost, double half_ratio, int bins, int split_mode, int* split_out,
                             int* items_out, int* off_out, int* n_items_out, double min_gain) {
  std::vector<int> first(ntypes + 1, 0);
  for (int j = 0; j < ntypes; j++) first[j + 1] = first[j] + count[j];
  const int total = first[ntypes];
  std::vector<int> split(ntypes, 0), ecount(2 * ntypes), best;
  std::vector<double> ecost(2 * ntypes);
  for (int j = 0; j < ntypes; j++) { ecost[j] = cost[j]; ecost[ntypes + j] = half_ratio * cost[j]; }
  auto makespan = [&](const std::vector<int>& sp, std::vector<int>& take) {
    for (int j = 0; j < ntypes; j++) { ecount[j] = count[j] - sp[j]; ecount[ntypes + j] = 2 * sp[j]; }
    return fused_pack(2 * ntypes, ecount.data(), ecost.data(), bins, take);
  };
  double T = makespan(split, best);
  if (const char* e = getenv("ANI_FUSED_SPLIT")) {   // experiments: "k0,k1,..." = how many of each type's last items to cut
    int j = 0;
    for (const char* q = e; *q && j < ntypes; j++) {
      split[j] = std::max(0, std::min(count[j], atoi(q)));
      while (*q && *q != ',') q++;
      if (*q == ',') q++;
    }
    T = makespan(split, best);
  } else if (split_mode == 2) {
    for (int j = 0; j < ntypes; j++) split[j] = count[j];
    T = makespan(split, best);
  } else if (split_mode == 1) {
    const double T0 = T;
    std::vector<int> order(ntypes), cur = split, take;
    for (int j = 0; j < ntypes; j++) order[j] = j;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] * count[a] > cost[b] * count[b]; });
    double Tc = T;
    for (int sweep = 0; sweep < 2; sweep++)
      for (int jj = 0; jj < ntypes && jj < 4; jj++) {
        const int j = order[jj];
        if (count[j] == 0) continue;
        const int cap = std::min(count[j], bins);
        const int cand[7] = {0, count[j] % bins, cap / 8, cap / 4, cap / 2, (3 * cap) / 4, cap};
        int keep = cur[j];
        for (int c : cand) {
          if (c < 0 || c > count[j]) continue;
          std::vector<int> trial = cur;
          trial[j] = c;
          const double Tt = makespan(trial, take);
          if (Tt < Tc * (1.0 - 1e-6)) { Tc = Tt; keep = c; }
        }
        cur[j] = keep;
      }
    if (Tc < (1.0 - min_gain) * T0) { split = cur; T = makespan(split, best); }
    else T = makespan(split, best);
  }
  // numbering: type j's whole items first[j] .. first[j] + count[j] - split[j]; the halves of the split[j] items behind them
  std::vector<int> order(2 * ntypes), next(2 * ntypes, 0);
  for (int j = 0; j < 2 * ntypes; j++) order[j] = j;
  std::sort(order.begin(), order.end(), [&](int a, int b) { return ecost[a] > ecost[b]; });
  int n = 0;
  for (int b = 0; b < bins; b++) {
    off_out[b] = n;
    for (int jj = 0; jj < 2 * ntypes; jj++) {
      const int e = order[jj];
      for (int k = 0; k < best[(size_t)b * 2 * ntypes + e]; k++) {
        const int i = next[e]++;
        if (e < ntypes) items_out[n++] = first[e] + i;
        else {
          const int j = e - ntypes;
          items_out[n++] = total + 2 * (first[j] + count[j] - split[j] + (i >> 1)) + (i & 1);
        }
      }
    }
  }
  off_out[bins] = n;
  if (split_out) for (int j = 0; j < ntypes; j++) split_out[j] = split[j];
  if (n_items_out) *n_items_out = n;
  return T;
}

hipError_t launch_mlp_fused(const FusedArgs& G, MlpArith arith, hipStream_t st) {
  const int ncu = fused_num_cus();
  const int total = G.tile_start[G.nprob] * (G.member_items ? G.M : 1);
  if (total <= 0) return hipSuccess;
  hipError_t e = hipSuccess;
  if (!G.sched_items) e = hipMemsetAsync(G.counter, 0, sizeof(int), st);
  if (e != hipSuccess) return e;
  const int pi = arith == MLP_F16X2 ? 1 : 0;
  const void* fn = pi ? (const void*)mlp_fused<2> : (const void*)mlp_fused<3>;
  {
    // raising the dynamic-LDS limit is per kernel and device: once each
    static std::set<std::pair<int, const void*>> raised;
    static std::mutex mtx;
    int dev = 0;
    e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mtx);
    if (!raised.count(std::make_pair(dev, fn))) {
      e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kFusedLds);
      if (e != hipSuccess) return e;
      raised.insert(std::make_pair(dev, fn));
    }
  }
  const int grid = G.sched_items ? G.sched_blocks : (total < ncu ? total : ncu);
  if (pi) hipLaunchKernelGGL(mlp_fused<2>, dim3(grid), dim3(256), kFusedLds, st, G);
  else hipLaunchKernelGGL(mlp_fused<3>, dim3(grid), dim3(256), kFusedLds, st, G);
  return hipGetLastError();
}

}  // namespace ani
