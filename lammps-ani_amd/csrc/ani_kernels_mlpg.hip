// ani_kernels_mlpg.hip — the whole MLP of a row tile in one workgroup, SIXTEEN rows per wave (gfx950).
//
// Same job as ani_kernels_mlpf.hip (BmmEnsemble forward + autograd back to dE/dAEV of a species-pure row tile inside one
// workgroup, activations in registers, weights streamed through LDS; reference call sites models/lammps_ani.py:110,228-230,
// 197-206), different decomposition.  There a wave owns 32 rows, needs 512 registers and is alone on its SIMD: a lone wave
// issues in order, so every LDS-DMA instruction (>= 60 cycles of issue), every wait for a fragment and every barrier skew is
// time the matrix pipe idles (tools/mfma_shadow_probe.hip: one wave hides ~4 vector instructions behind a 32x32x16 MFMA and
// nothing behind a stall; 0.37 of the pipe's peak, profiles/r03_*).  Here a wave owns 16 rows (atoms):
//     v_mfma_f32_16x16x32_{bf16,f16}:  D[16 features][16 atoms] += W[16 features][32 k] * X[32 k][16 atoms]
// the WEIGHTS the A operand (a 1 KB piece = one fragment, 16 bytes per lane), the activations the B operand, an accumulator
// tile four registers.  A wave's activations are then 152 registers for the largest network, the kernel fits 256, and a
// workgroup of EIGHT waves -- two per SIMD -- takes a 128-row tile: while one wave of a SIMD waits (its DMA issue, a fragment,
// its conversions) the other's MFMAs run.  The same code with four waves is a 64-row tile for launches that would leave CUs
// idle (small per-GPU shares of a decomposed box): twice the workgroups, half the latency of a tile.
//
// Accumulator as the next operand, 16-row form: lane (c = lane & 15, g = lane >> 4) of an output tile holds features
// 16 t + 4 g + r (r = 0..3) of atom c; a B fragment of k-step ks wants 8 values per lane at k-slots 8 g + j.  Slot (g, j) of
// k-step ks is DEFINED as feature  kappa(ks, g, j) = 32 ks + 16 (j >> 2) + 4 g + (j & 3): registers 0..3 of tile 2 ks and of
// tile 2 ks + 1 -- no lane movement; the weight stream is permuted to match when the model is uploaded (build_stream16).
// The first product reads its B operand from the AEV rows (8 consecutive columns per lane: identity order).
//
// Weight stream of a (species, member): 1 KB pieces in consumption order; a SLAB is what one barrier hands over:
//   F1, F2, F3 : one k-step (32 deep) of the product = NT blocks (output tiles in order), block = P pieces (planes)
//   B3, B2     : TPS output tiles, each through all its k-steps (TPS = what fits a slot)
//   B1         : dE/dAEV in chunks of 16 (then 8, then the rest) output tiles; a slab = one k-step of the chunk
// Three slots of 48 KB; slab j lives in slot j mod 3; the boundary in front of slab j (wait for own loads, barrier) stands in
// front of the LAST block of slab j - 1 and starts the loads of slab j + 1 (one piece per block and wave).
#include <algorithm>
#include <cstdint>
#include <mutex>
#include <set>
#include <utility>

#include "ani_kernels.h"
#include "ani_mlp_common.h"

namespace ani {

constexpr int kGSlot = 48;                       // pieces (KB) per slot
constexpr int kGSlots = 3;
constexpr int kGConstBytes = 4096;
constexpr int kGLds = kGSlots * kGSlot * 1024 + kGConstBytes;
static_assert(kGLds + 64 <= 160 * 1024, "slots + constants exceed the LDS of a CU");

typedef float f32x4g __attribute__((ext_vector_type(4)));
typedef unsigned u32x4g __attribute__((ext_vector_type(4)));
template <int P>
struct FragG { u32x4g p[P]; };

// ---- weight stream builder --------------------------------------------------------------------------------------------
// src[row][k] (ld elements per row): the A operand.  NT output tiles (16 rows each, starting at tile nt_off) x KS k-steps
// (32 deep).  Block order: ks-major (order 0: b = ks * NT + nt) or tile-major (order 1: b = nt * KS + ks).  identity: k-slot
// (g, j) is column 32 ks + 8 g + j (first product: AEV columns), else kappa(ks, g, j) above.
__global__ void build_stream16_kernel(const float* __restrict__ src, int ld, int rows_valid, int k_valid, int NT, int KS, int order,
                                      int nt_off, int identity, int P, float scale, unsigned short* __restrict__ dst) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)NT * KS * 512;
  if (idx >= total) return;
  const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
  const int blk = (int)(idx >> 9);
  int ks, nt;
  if (order == 0) { ks = blk / NT; nt = blk - ks * NT; } else { nt = blk / KS; ks = blk - nt * KS; }
  const int g = lane >> 4;
  const int row = 16 * (nt_off + nt) + (lane & 15);
  const int k = identity ? 32 * ks + 8 * g + j : 32 * ks + 16 * (j >> 2) + 4 * g + (j & 3);
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
void launch_build_stream16(const float* src, int ld, int rows_valid, int k_valid, int NT, int KS, int order, int nt_off, int identity,
                           int P, float scale, unsigned short* dst, hipStream_t st) {
  const long long total = (long long)NT * KS * 512;
  if (total <= 0) return;
  hipLaunchKernelGGL(build_stream16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, ld, rows_valid, k_valid, NT, KS,
                     order, nt_off, identity, P, scale, dst);
}

// ---- the slots ----------------------------------------------------------------------------------------------------------
struct RingG {
  const unsigned char* src;    // next piece of the stream nobody has asked for yet (global, wave-uniform)
  unsigned char* lds;          // slot 0
  int islot, cslot;            // slot the next slab is loaded to / the next boundary reads from
  const unsigned char* psrc;   // the slab being loaded a piece per block: this wave's next piece of it (global) ...
  unsigned lds_p;              // ... and where it goes (LDS byte address)
  int pk, pn;                  // this wave's next piece index / the slab's size (pk >= pn: nothing pending)
};
__device__ __forceinline__ int g_next_slot(int s) { return s + 1 == kGSlots ? 0 : s + 1; }

// One piece (1 KB: 16 bytes per lane) global -> LDS.  Written as ONE asm statement -- M0 write, a wait state, the load --
// instead of __builtin_amdgcn_global_load_lds: with the builtin the eight-wave kernel in the two-term arithmetic returned,
// about every second launch, one wave (always one of waves 4..7) with wrong rows (tools/mlpg_debug.py, profiles/r04_mlpg_dma_race.log:
// 48 of 100 evaluations; 0 of 100 with this form, with or without the s_nop).  Nothing in the code the compiler made for
// the builtin looked wrong to us; the statement form also keeps the compiler from modelling the LDS write (its own
// s_waitcnt vmcnt in front of every later ds_read that might alias): the waits are the explicit ones at the slab boundaries.
__device__ __forceinline__ void g_dma(const unsigned char* g, unsigned lds_addr) {
#ifndef ABLG_NODMA
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(g) : "memory", "m0");
#endif
}
template <int W>
__device__ __forceinline__ void g_load_piece(RingG& r, unsigned lane16) {
  g_dma(r.psrc + lane16, r.lds_p);
  r.psrc += W * 1024; r.lds_p += W * 1024; r.pk += W;
}
template <int W>
__device__ __forceinline__ void g_drip(RingG& r, unsigned lane16) {
  if (r.pk < r.pn) g_load_piece<W>(r, lane16);
}
template <int W>
__device__ __forceinline__ void g_flush(RingG& r, unsigned lane16) {
  while (r.pk < r.pn) g_load_piece<W>(r, lane16);
}
// start loading the next slab of the stream (n pieces; 0: the stream is at its end)
template <int W>
__device__ __forceinline__ void g_begin(RingG& r, int n, int wave) {
  if (n <= 0) return;
  r.psrc = r.src + wave * 1024;
  r.lds_p = (unsigned)(uintptr_t)r.lds + r.islot * (kGSlot * 1024) + wave * 1024;
  r.pk = wave; r.pn = n;
  r.src += (size_t)n * 1024;
  r.islot = g_next_slot(r.islot);
}
// In front of a slab: everything this wave asked for has landed, then the workgroup barrier (everybody's pieces are there and
// nobody reads the slab two back any more), then the loads of the following slab (n_next pieces) start.  Returns this lane's
// read address of the slab's first piece.
template <int W>
__device__ __forceinline__ const unsigned char* g_boundary(RingG& r, int n_next, int wave, unsigned lane16) {
  g_flush<W>(r, lane16);
#ifndef ABLG_NOWAIT
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  asm volatile("s_barrier" ::: "memory");
#ifdef ABLG_DBLBAR   // experiment: a second rendezvous behind a full drain
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
#ifdef ABLG_SLEEP    // experiment: time between the barrier and the first read of the slab
  asm volatile("s_sleep 2" ::: "memory");
#endif
  const int s = r.cslot;
  r.cslot = g_next_slot(s);
  g_begin<W>(r, n_next, wave);
  return r.lds + s * (kGSlot * 1024) + lane16;
}

template <int P>
__device__ __forceinline__ void g_read(const unsigned char* base, int b, FragG<P>& f) {
#pragma unroll
  for (int p = 0; p < P; p++) f.p[p] = *reinterpret_cast<const u32x4g*>(base + (b * P + p) * 1024);
}
template <int P>
__device__ __forceinline__ void g_mma(const FragG<P>& a, const FragG<P>& b, f32x4g& acc) {
#ifdef ABLG_NOMMA
#pragma unroll
  for (int i = 0; i < P; i++) asm volatile("" ::"v"(a.p[i]), "v"(b.p[i]));
  return;
#endif
  if constexpr (P == 3) {   // smallest terms first
    const bf16x8 ah = __builtin_bit_cast(bf16x8, a.p[0]), am = __builtin_bit_cast(bf16x8, a.p[1]), al = __builtin_bit_cast(bf16x8, a.p[2]);
    const bf16x8 bh = __builtin_bit_cast(bf16x8, b.p[0]), bm = __builtin_bit_cast(bf16x8, b.p[1]), bl = __builtin_bit_cast(bf16x8, b.p[2]);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
  } else {
    const f16x8 ah = __builtin_bit_cast(f16x8, a.p[0]), al = __builtin_bit_cast(f16x8, a.p[1]);
    const f16x8 bh = __builtin_bit_cast(f16x8, b.p[0]), bl = __builtin_bit_cast(f16x8, b.p[1]);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
  }
}
// two fp32 values (elements 2 i, 2 i + 1 of a k-step's fragment) -> word i of every plane
template <int P>
__device__ __forceinline__ void g_split_pair(float x0, float x1, float a_scale, int i, FragG<P>& f) {
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
// pair i (0..3) of k-step ks of X: elements j = 2 i, 2 i + 1 of the fragment = registers 2 (i & 1), + 1 of tile 2 ks + (i >> 1)
template <int P, int NTX>
__device__ __forceinline__ void g_split_of(const f32x4g (&X)[NTX], int ks, int i, float a_scale, FragG<P>& f) {
  g_split_pair<P>(X[2 * ks + (i >> 1)][2 * (i & 1)], X[2 * ks + (i >> 1)][2 * (i & 1) + 1], a_scale, i, f);
}
template <int P>
__device__ __forceinline__ void g_pin(FragG<P>& f) {
#pragma unroll
  for (int p = 0; p < P; p++) asm volatile("" : "+v"(f.p[p]));
}
template <int NT>
__device__ __forceinline__ void g_zero(f32x4g (&X)[NT]) {
#pragma unroll
  for (int nt = 0; nt < NT; nt++) X[nt] = f32x4g{0.f, 0.f, 0.f, 0.f};
}

// Fragments are requested kPF blocks before the block that multiplies with them (kPF + 1 register sets, statically rotated);
// the boundary in front of a slab therefore stands kPF blocks before the slab's first block.  kPF = 2 measured no faster than 1
// (MLP ms at 12 501 / 25 002 / 100 002 atoms: 0.0576 / 0.0806 / 0.231 against 0.0571 / 0.0789 / 0.227, profiles/r04_mlpg_prefetch_abl.log):
// the kernel does not wait for fragments, it is bound by what its waves have to issue.
#ifndef ANI_G_PF
#define ANI_G_PF 1
#endif
constexpr int kPF = ANI_G_PF;

// acc[NT] += W (stream) * X, X in registers: KS k-steps, a slab per k-step.  n_after: size of the slab that follows this
// product's last one in the stream.
// ACT = false: the wave has no rows in this work item (the upper waves of a half item): it keeps its share of the loads and every
// rendezvous -- the calls of g_boundary / g_drip / g_flush are the same statements, in the same order -- and does nothing else.
template <int KS, int NT, int NTX, int P, int W, bool ACT = true>
__device__ __forceinline__ void g_product_fwd(RingG& r, const f32x4g (&X)[NTX], f32x4g (&acc)[NT], float a_scale, int n_after, int wave,
                                              unsigned lane16) {
  static_assert(NTX == 2 * KS && NT >= 4 && NT * P <= kGSlot && NT > kPF, "shape");
  constexpr int N = NT * P, NB = KS * NT;
  FragG<P> bq[2], fa[kPF + 1];
  if constexpr (ACT) {
#pragma unroll
    for (int i = 0; i < 4; i++) g_split_of<P>(X, 0, i, a_scale, bq[0]);
  }
  const unsigned char* base = g_boundary<W>(r, KS > 1 ? N : n_after, wave, lane16);
  if constexpr (ACT) {
#pragma unroll
    for (int q = 0; q < kPF; q++) g_read<P>(base, q, fa[q]);
  }
#pragma unroll
  for (int ks = 0; ks < KS; ks++) {
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      const int idx = ks * NT + nt, nx = idx + kPF;   // nx: the block requested now
      if (nx < NB) {
        const int nks = nx / NT, nnt = nx % NT;
        if (nnt == 0) base = g_boundary<W>(r, nks + 1 < KS ? N : n_after, wave, lane16);   // slab nks starts: slab nks + 1 is asked for
        if constexpr (ACT) g_read<P>(base, nnt, fa[nx % (kPF + 1)]);
      }
      if constexpr (ACT) {
        if (ks + 1 < KS && nt < 4) g_split_of<P>(X, ks + 1, nt, a_scale, bq[(ks + 1) & 1]);
      }
      g_drip<W>(r, lane16);
      if constexpr (ACT) {
        g_mma<P>(fa[idx % (kPF + 1)], bq[ks & 1], acc[nt]);
        if (nt == 3 && ks + 1 < KS) g_pin<P>(bq[(ks + 1) & 1]);
      }
    }
  }
}

// The hidden backward products, in place: Y[nt] <- (sum_ks W[nt][ks] X[ks]) * inv * celu'(z[nt]), Y holding the stored
// activation on entry and the gradient on exit.  X is split into its 16-bit terms once (during the first output tile's pass);
// then one output tile at a time runs through all k-steps.  A slab = TPS output tiles.
template <int KS, int P>
struct GInplace { static constexpr int tps = (kGSlot / (KS * P)) < 1 ? 1 : kGSlot / (KS * P); };
template <int KS, int NT, int P>
__host__ __device__ constexpr int g_inplace_slab_pieces(int s) {   // pieces of slab s of an in-place product (0 beyond the last)
  constexpr int tps = GInplace<KS, P>::tps;
  const int first = s * tps;
  if (first >= NT) return 0;
  return ((NT - first < tps) ? NT - first : tps) * KS * P;
}
template <int KS, int NT, int NTX, int P, int W, bool ACT = true>
__device__ __forceinline__ void g_product_inplace(RingG& r, const f32x4g (&X)[NTX], f32x4g (&Y)[NT], float a_scale, float inv,
                                                  float inv_alpha, int n_after, int wave, unsigned lane16) {
  static_assert(NTX == 2 * KS && KS * P <= kGSlot && KS > kPF, "shape");
  constexpr int TPS = GInplace<KS, P>::tps, NB = NT * KS;
  FragG<P> bf[KS], fa[kPF + 1];
  if constexpr (ACT) {
#pragma unroll
    for (int i = 0; i < 4; i++) g_split_of<P>(X, 0, i, a_scale, bf[0]);
  }
  f32x4g acc[2];
  const unsigned char* base = g_boundary<W>(r, g_inplace_slab_pieces<KS, NT, P>(1) > 0 ? g_inplace_slab_pieces<KS, NT, P>(1) : n_after, wave, lane16);
  if constexpr (ACT) {
#pragma unroll
    for (int q = 0; q < kPF; q++) g_read<P>(base, q, fa[q]);
  }
#pragma unroll
  for (int nt = 0; nt < NT; nt++) {
    if constexpr (ACT) acc[nt & 1] = f32x4g{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      const int idx = nt * KS + ks, nx = idx + kPF;
      if (nx < NB) {
        const int ntn = nx / KS, s = ntn / TPS, lb = nx - s * TPS * KS;   // block nx: slab s, block lb of it
        if (lb == 0) {
          const int n2 = g_inplace_slab_pieces<KS, NT, P>(s + 1);
          base = g_boundary<W>(r, n2 > 0 ? n2 : n_after, wave, lane16);
        }
        if constexpr (ACT) g_read<P>(base, lb, fa[nx % (kPF + 1)]);
      }
      if constexpr (ACT) {
        if (nt == 0 && ks + 1 < KS) {
#pragma unroll
          for (int i = 0; i < 4; i++) g_split_of<P>(X, ks + 1, i, a_scale, bf[ks + 1]);
        }
      }
      g_drip<W>(r, lane16);
      if constexpr (ACT) {
        g_mma<P>(fa[idx % (kPF + 1)], bf[ks], acc[nt & 1]);
        if (nt == 0 && ks + 1 < KS) g_pin<P>(bf[ks + 1]);
        if (ks == (KS > 1 ? 1 : 0) && nt > 0) {
#pragma unroll
          for (int i = 0; i < 4; i++) Y[nt - 1][i] = acc[(nt - 1) & 1][i] * inv * dcelu_from_h(Y[nt - 1][i], inv_alpha);
        }
      }
    }
  }
  if constexpr (ACT) {
#pragma unroll
    for (int i = 0; i < 4; i++) Y[NT - 1][i] = acc[(NT - 1) & 1][i] * inv * dcelu_from_h(Y[NT - 1][i], inv_alpha);
  }
}

// constants of a member in LDS (floats): b0[16 NT1] b1[16 NT2] b2[16 NT3] w3[16 NT3] then {b3, inv[6]} -- the layout of the
// 32-row kernel (fused_consts_floats): 16 NT16 = 32 NT32
template <int NT1, int NT2, int NT3>
struct GConst {
  static constexpr int b0 = 0, b1 = 16 * NT1, b2 = b1 + 16 * NT2, w3 = b2 + 16 * NT3, tail = w3 + 16 * NT3, count = tail + 8;
};

template <int NT>
__device__ __forceinline__ void g_epilogue_celu(f32x4g (&X)[NT], const float* b, int g, float inv, float alpha, float inv_alpha) {
#pragma unroll
  for (int nt = 0; nt < NT; nt++) {
    const float4 bv = *reinterpret_cast<const float4*>(b + 16 * nt + 4 * g);
    X[nt][0] = celu_f(fmaf(X[nt][0], inv, bv.x), alpha, inv_alpha);
    X[nt][1] = celu_f(fmaf(X[nt][1], inv, bv.y), alpha, inv_alpha);
    X[nt][2] = celu_f(fmaf(X[nt][2], inv, bv.z), alpha, inv_alpha);
    X[nt][3] = celu_f(fmaf(X[nt][3], inv, bv.w), alpha, inv_alpha);
  }
}

struct GTileCtx {
  float alpha, inv_alpha, scale, a_fwd, a_bwd;
  int m0, m1;          // members this work item runs
  float* parts;        // member_items: per-member dE/dAEV buffers, else null
  long long part_stride;
};

// dE/dAEV chunks of a problem with nt0 16-column tiles: as many of 16 as fit, then one of 8 if 8 are left, then the rest
__host__ __device__ inline int g_b1_chunks(int nt0) { return nt0 / 16 + ((nt0 % 16) >= 8 ? 1 : 0) + ((nt0 % 8) ? 1 : 0); }
__host__ __device__ inline int g_b1_chunk_tiles(int nt0, int ci) {
  const int full = nt0 / 16;
  if (ci < full) return 16;
  const int rem = nt0 % 16;
  if (ci == full && rem >= 8) return 8;
  return ci < g_b1_chunks(nt0) ? rem % 8 : 0;
}

// one chunk of NTC (compile-time) dE/dAEV tiles: all k-steps of g1, a slab per k-step
template <int NTC, int KS, int NT1, int P, int W, bool ACT = true>
__device__ __forceinline__ void g_b1_chunk(RingG& r, const f32x4g (&G1)[NT1], f32x4g (&acc)[NTC], float a_scale, int n_after, int wave,
                                           unsigned lane16) {
  static_assert(NT1 == 2 * KS && NTC >= 4 && NTC > kPF, "shape");
  constexpr int N = NTC * P, NB = KS * NTC;
  FragG<P> bq[2], fa[kPF + 1];
  if constexpr (ACT) {
#pragma unroll
    for (int i = 0; i < 4; i++) g_split_of<P>(G1, 0, i, a_scale, bq[0]);
  }
  const unsigned char* base = g_boundary<W>(r, KS > 1 ? N : n_after, wave, lane16);
  if constexpr (ACT) {
#pragma unroll
    for (int q = 0; q < kPF; q++) g_read<P>(base, q, fa[q]);
  }
#pragma unroll
  for (int ks = 0; ks < KS; ks++) {
#pragma unroll
    for (int t = 0; t < NTC; t++) {
      const int idx = ks * NTC + t, nx = idx + kPF;
      if (nx < NB) {
        const int nks = nx / NTC, nt = nx % NTC;
        if (nt == 0) base = g_boundary<W>(r, nks + 1 < KS ? N : n_after, wave, lane16);
        if constexpr (ACT) g_read<P>(base, nt, fa[nx % (kPF + 1)]);
      }
      if constexpr (ACT) {
        if (ks + 1 < KS && t < 4) g_split_of<P>(G1, ks + 1, t, a_scale, bq[(ks + 1) & 1]);
      }
      g_drip<W>(r, lane16);
      if constexpr (ACT) {
        g_mma<P>(fa[idx % (kPF + 1)], bq[ks & 1], acc[t]);
        if (t == 3 && ks + 1 < KS) g_pin<P>(bq[(ks + 1) & 1]);
      }
    }
  }
}
// the last, narrower chunk (ntc < 8 tiles, run-time): plain loops, a boundary in front of every slab
template <int KS, int NT1, int P, int W, bool ACT = true>
__device__ __forceinline__ void g_b1_tail(RingG& r, const f32x4g (&G1)[NT1], f32x4g (&acc)[8], int ntc, float a_scale, int n_after, int wave,
                                          unsigned lane16) {
  FragG<P> bq, fa;
#pragma unroll
  for (int ks = 0; ks < KS; ks++) {
    if constexpr (ACT) {
#pragma unroll
      for (int i = 0; i < 4; i++) g_split_of<P>(G1, ks, i, a_scale, bq);
    }
    const unsigned char* base = g_boundary<W>(r, ks + 1 < KS ? ntc * P : n_after, wave, lane16);
    if constexpr (ACT) {
#pragma unroll
      for (int t = 0; t < 8; t++) {
        if (t < ntc) {
          g_read<P>(base, t, fa);
          g_mma<P>(fa, bq, acc[t]);
        }
      }
    }
    g_flush<W>(r, lane16);
  }
}

template <int NTC>
__device__ __forceinline__ void g_store_chunk(const f32x4g (&acc)[NTC], int ntc, float* __restrict__ grow, int c0, int acols, float inv, bool add) {
#pragma unroll
  for (int t = 0; t < NTC; t++) {
    if (t < ntc && 16 * (c0 + t) < acols) {
      float4 o = make_float4(acc[t][0] * inv, acc[t][1] * inv, acc[t][2] * inv, acc[t][3] * inv);
      float4* dst = reinterpret_cast<float4*>(grow + 16 * (c0 + t));   // + 4 g is in grow
      if (add) { const float4 old = *dst; o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
      *dst = o;
    }
  }
}

// One work item of a species bucket: the 16-row strips from row0 on, one per ACTIVE wave (a whole tile: all W waves and
// row0 = tile * 16 W; a half item: the lower W / 2 waves, the others run with ACT = false).  Shape (NT1, NT2, NT3): 16-feature
// tiles of the three hidden layers.
template <int NT1, int NT2, int NT3, int P, int W, bool ACT>
__device__ __forceinline__ void g_tile(const GTileCtx& cx, const FusedProb& pr, int row0, int wave, int lane, unsigned char* ring, float* cst) {
  using CL = GConst<NT1, NT2, NT3>;
  static_assert(CL::count * 4 <= kGConstBytes, "constants do not fit their LDS block");
  static_assert(NT1 % 2 == 0 && NT2 % 2 == 0 && NT3 % 2 == 0, "whole k-steps");
  constexpr int N_F1 = NT1 * P, N_F2 = NT2 * P, N_F3 = NT3 * P;
  constexpr int N_B3 = g_inplace_slab_pieces<NT3 / 2, NT2, P>(0), N_B2 = g_inplace_slab_pieces<NT2 / 2, NT1, P>(0);
  static_assert(N_F1 <= kGSlot && 16 * P <= kGSlot, "every slab must fit a slot");
  const int c = lane & 15, g = lane >> 4;
  const unsigned lane16 = lane * 16;
  const int row = ACT ? row0 + 16 * wave + c : 0;
  float valid = 0.f;
  const float* __restrict__ arow = nullptr;
  float* __restrict__ grow = nullptr;
  if constexpr (ACT) {
    valid = pr.centre_of_row[row] >= 0 ? cx.scale : 0.f;
    arow = pr.aev + (size_t)row * pr.aev_stride + 8 * g;
    grow = (cx.parts ? cx.parts + (size_t)cx.m0 * cx.part_stride + (size_t)pr.gaev_row0 * pr.aev_stride : pr.gaev) +
           (size_t)row * pr.aev_stride + 4 * g;
  }
  const int acols = pr.acols, ks1 = (acols + 31) >> 5, nt0 = (acols + 15) >> 4;
  const int nchunks = g_b1_chunks(nt0);
  RingG r;
  r.src = pr.stream + (size_t)cx.m0 * pr.pieces_per_member * 1024;
  r.lds = ring; r.islot = 0; r.cslot = 0; r.pk = 0; r.pn = 0; r.psrc = r.src; r.lds_p = 0;
  g_begin<W>(r, N_F1, wave);   // slab 0 (its loads leave at the first boundary's flush)

  for (int m = cx.m0; m < cx.m1; m++) {
    {
      constexpr int nconst = (CL::count * 4 + 1023) >> 10;
      if (wave < nconst) {
        const unsigned char* gp = reinterpret_cast<const unsigned char*>(pr.consts + (size_t)m * pr.consts_per_member) + wave * 1024 + lane16;
        g_dma(gp, (unsigned)(uintptr_t)cst + wave * 1024);
      }
      g_flush<W>(r, lane16);
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    float inv_f1 = 0.f, inv_f2 = 0.f, inv_f3 = 0.f, inv_b3 = 0.f, inv_b2 = 0.f, inv_b1 = 0.f, b3 = 0.f;
    if constexpr (ACT) {
      inv_f1 = cst[CL::tail + 1]; inv_f2 = cst[CL::tail + 2]; inv_f3 = cst[CL::tail + 3];
      inv_b3 = cst[CL::tail + 4]; inv_b2 = cst[CL::tail + 5]; inv_b1 = cst[CL::tail + 6];
      b3 = cst[CL::tail];
    }

    // ---- F1: h1 = celu(W0 aev + b0); B operand from the AEV rows, a k-step = 32 columns = 8 per lane, requested two k-steps ahead
    f32x4g X1[NT1];
    if constexpr (ACT) g_zero(X1);
    {
      const float4 z4 = make_float4(0, 0, 0, 0);
      auto ld = [&](int ks, float4& a, float4& b) {
        a = z4; b = z4;
        if constexpr (ACT) {
          if (ks < ks1 && 32 * ks + 8 * g < acols) {
            a = *reinterpret_cast<const float4*>(arow + 32 * ks);
            b = *reinterpret_cast<const float4*>(arow + 32 * ks + 4);
          }
        }
      };
      auto cvt = [&](int i, const float4& a, const float4& b, FragG<P>& f) {
        if constexpr (ACT) {
          if (i == 0) g_split_pair<P>(a.x, a.y, cx.a_fwd, 0, f);
          if (i == 1) g_split_pair<P>(a.z, a.w, cx.a_fwd, 1, f);
          if (i == 2) g_split_pair<P>(b.x, b.y, cx.a_fwd, 2, f);
          if (i == 3) g_split_pair<P>(b.z, b.w, cx.a_fwd, 3, f);
        }
      };
      float4 c0a, c0b, v1a, v1b, v2a, v2b, w2a, w2b;
      ld(0, c0a, c0b); ld(1, v1a, v1b); ld(2, v2a, v2b);
      FragG<P> b0, b1, fa[2];
#pragma unroll
      for (int i = 0; i < 4; i++) cvt(i, c0a, c0b, b0);
      if constexpr (ACT) b1 = b0;
      const unsigned char* base = g_boundary<W>(r, ks1 > 1 ? N_F1 : N_F2, wave, lane16);
      if constexpr (ACT) g_read<P>(base, 0, fa[0]);
      for (int ks = 0; ks < ks1; ks++) {
        ld(ks + 3, w2a, w2b);
#pragma unroll
        for (int nt = 0; nt < NT1; nt++) {
          if (nt + 1 < NT1) {
            if constexpr (ACT) g_read<P>(base, nt + 1, fa[(nt + 1) & 1]);
          } else if (ks + 1 < ks1) {
            base = g_boundary<W>(r, ks + 2 < ks1 ? N_F1 : N_F2, wave, lane16);
            if constexpr (ACT) g_read<P>(base, 0, fa[0]);     // NT1 is even: block 0 of every slab lands in fragment set 0
          }
          if (nt < 4) cvt(nt, v1a, v1b, b1);   // k-step ks + 1
          g_drip<W>(r, lane16);
          if constexpr (ACT) {
            g_mma<P>(fa[nt & 1], b0, X1[nt]);
            if (nt == 3) g_pin<P>(b1);
          }
        }
        if constexpr (ACT) {
          b0 = b1;
          v1a = v2a; v1b = v2b; v2a = w2a; v2b = w2b;
        }
      }
    }
    if constexpr (ACT) g_epilogue_celu(X1, cst + CL::b0, g, inv_f1, cx.alpha, cx.inv_alpha);

    // ---- F2, F3 ----
    f32x4g X2[NT2];
    if constexpr (ACT) g_zero(X2);
    g_product_fwd<NT1 / 2, NT2, NT1, P, W, ACT>(r, X1, X2, cx.a_fwd, N_F3, wave, lane16);
    if constexpr (ACT) g_epilogue_celu(X2, cst + CL::b1, g, inv_f2, cx.alpha, cx.inv_alpha);
    f32x4g X3[NT3];
    if constexpr (ACT) g_zero(X3);
    g_product_fwd<NT2 / 2, NT3, NT2, P, W, ACT>(r, X2, X3, cx.a_fwd, N_B3, wave, lane16);
    // last hidden layer fused with the 1-wide output layer and the backward seed dE/dz3 = (1/M) w3 celu'(z3)
    if constexpr (ACT) {
      float es = 0.f;
#pragma unroll
      for (int nt = 0; nt < NT3; nt++) {
        const float4 bv = *reinterpret_cast<const float4*>(cst + CL::b2 + 16 * nt + 4 * g);
        const float4 wv = *reinterpret_cast<const float4*>(cst + CL::w3 + 16 * nt + 4 * g);
        const float bb[4] = {bv.x, bv.y, bv.z, bv.w}, ww[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const float hv = celu_f(fmaf(X3[nt][i], inv_f3, bb[i]), cx.alpha, cx.inv_alpha);
          es = fmaf(hv, ww[i], es);
          X3[nt][i] = valid * ww[i] * dcelu_from_h(hv, cx.inv_alpha);
        }
      }
      es += __shfl_xor(es, 16);
      es += __shfl_xor(es, 32);
      if (g == 0) pr.e_rows[(size_t)m * pr.sE + row] = valid * (es + b3);
    }

    // ---- B3: g2 = (W2^T g3) celu'(z2) over h2;  B2: g1 = (W1^T g2) celu'(z1) over h1 ----
    g_product_inplace<NT3 / 2, NT2, NT3, P, W, ACT>(r, X3, X2, cx.a_bwd, inv_b3, cx.inv_alpha, N_B2, wave, lane16);
    const int n_b1_first = g_b1_chunk_tiles(nt0, 0) * P;
    g_product_inplace<NT2 / 2, NT1, NT2, P, W, ACT>(r, X2, X1, cx.a_bwd, inv_b2, cx.inv_alpha, n_b1_first, wave, lane16);
    f32x4g (&G1)[NT1] = X1;

    // ---- B1: dE/dAEV = W0^T g1 in chunks of output tiles; members after the first of a work item add to what is there ----
    int c0 = 0;
    for (int ci = 0; ci < nchunks; ci++) {
      const int ntc = g_b1_chunk_tiles(nt0, ci);
      int n_after = g_b1_chunk_tiles(nt0, ci + 1) * P;                       // first slab of the next chunk,
      if (ci + 1 == nchunks) n_after = (m + 1 < cx.m1) ? N_F1 : 0;           // of the next member, or nothing
      if (ntc == 16) {
        f32x4g acc[16];
        if constexpr (ACT) g_zero(acc);
        g_b1_chunk<16, NT1 / 2, NT1, P, W, ACT>(r, G1, acc, cx.a_bwd, n_after, wave, lane16);
        if constexpr (ACT) g_store_chunk<16>(acc, 16, grow, c0, acols, inv_b1, m > cx.m0);
      } else if (ntc == 8) {
        f32x4g acc[8];
        if constexpr (ACT) g_zero(acc);
        g_b1_chunk<8, NT1 / 2, NT1, P, W, ACT>(r, G1, acc, cx.a_bwd, n_after, wave, lane16);
        if constexpr (ACT) g_store_chunk<8>(acc, 8, grow, c0, acols, inv_b1, m > cx.m0);
      } else {
        f32x4g acc[8];
        if constexpr (ACT) g_zero(acc);
        g_b1_tail<NT1 / 2, NT1, P, W, ACT>(r, G1, acc, ntc, cx.a_bwd, n_after, wave, lane16);
        if constexpr (ACT) g_store_chunk<8>(acc, ntc, grow, c0, acols, inv_b1, m > cx.m0);
      }
      c0 += ntc;
    }
  }
}

template <int P, int W>
__global__ __launch_bounds__(64 * W, 1) void mlp_fused16(FusedArgs G) {
  extern __shared__ uint4 smem4g[];
  unsigned char* ring = reinterpret_cast<unsigned char*>(smem4g);
  float* cst = reinterpret_cast<float*>(ring + kGSlots * kGSlot * 1024);
  __shared__ int s_tile;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  GTileCtx cx;
  cx.alpha = G.alpha; cx.inv_alpha = G.inv_alpha; cx.scale = G.scale;
  cx.a_fwd = P == 2 ? 16.f : 1.f; cx.a_bwd = P == 2 ? 4096.f : 1.f;
  cx.m0 = 0; cx.m1 = G.M; cx.parts = nullptr; cx.part_stride = 0;
  const int per_tile = G.member_items ? G.M : 1;
  const int total = G.tile_start[G.nprob] * per_tile;
  int sched_pos = 0;
  for (;;) {
    __syncthreads();   // every wave is done with the tile before (slots, constants, s_tile)
    if (G.sched_items) {
      if (threadIdx.x == 0) {
        const int i = G.sched_off[blockIdx.x] + sched_pos;
        s_tile = i < G.sched_off[blockIdx.x + 1] ? G.sched_items[i] : -1;
      }
      sched_pos++;
    } else if (threadIdx.x == 0) {
      const int i = atomicAdd(G.counter, 1);
      s_tile = i < total ? i : -1;
    }
    __syncthreads();
    int item = __builtin_amdgcn_readfirstlane(s_tile);
    if (item < 0) break;
    // items [0, total): whole tiles; total + 2 i + hf: half hf of item i, run by the lower half of the waves (a static schedule
    // may cut the items that would otherwise make up a last, mostly idle round of workgroups: fused_schedule_halves)
    int hf = -1;
    if (item >= total) { hf = (item - total) & 1; item = (item - total) >> 1; }
    const int t = item / per_tile;
    if (G.member_items) {
      cx.m0 = item - t * per_tile; cx.m1 = cx.m0 + 1;
      cx.parts = G.gaev_parts; cx.part_stride = G.part_stride;
    }
    int pi = 0;
    while (pi + 1 < G.nprob && t >= G.tile_start[pi + 1]) pi++;
    const FusedProb& pr = G.p[pi];
    const int tile = t - G.tile_start[pi];
    const int row0 = tile * (16 * W) + (hf > 0 ? 8 * W : 0);
    // an item without a single real row (the padding of a bucket's last 128 rows): nothing to do
    if (pr.centre_of_row[row0] < 0) continue;
    const bool act = hf < 0 || wave < W / 2;
    if (act) {
      switch (pr.shape) {
        case 0: g_tile<16, 12, 10, P, W, true>(cx, pr, row0, wave, lane, ring, cst); break;
        case 1: g_tile<12, 10, 8, P, W, true>(cx, pr, row0, wave, lane, ring, cst); break;
        default: g_tile<10, 8, 6, P, W, true>(cx, pr, row0, wave, lane, ring, cst); break;
      }
    } else {
      switch (pr.shape) {
        case 0: g_tile<16, 12, 10, P, W, false>(cx, pr, row0, wave, lane, ring, cst); break;
        case 1: g_tile<12, 10, 8, P, W, false>(cx, pr, row0, wave, lane, ring, cst); break;
        default: g_tile<10, 8, 6, P, W, false>(cx, pr, row0, wave, lane, ring, cst); break;
      }
    }
  }
}

long long fused16_pieces_per_member(int shape, int acols, int P) {
  int nt[3];
  fused_shape_tiles(shape, nt);   // 32-feature tiles
  const long long n1 = 2 * nt[0], n2 = 2 * nt[1], n3 = 2 * nt[2];
  const long long ks1 = (acols + 31) / 32, nt0 = (acols + 15) / 16;
  return P * (ks1 * n1 + (n1 / 2) * n2 + (n2 / 2) * n3 + n2 * (n3 / 2) + n1 * (n2 / 2) + nt0 * (n1 / 2));
}
int fused16_b1_chunks(int nt0) { return g_b1_chunks(nt0); }
int fused16_b1_chunk_tiles(int nt0, int ci) { return g_b1_chunk_tiles(nt0, ci); }

hipError_t launch_mlp_fused16(const FusedArgs& G, MlpArith arith, int waves, hipStream_t st) {
  const int ncu = fused_num_cus();
  const int total = G.tile_start[G.nprob] * (G.member_items ? G.M : 1);
  if (total <= 0) return hipSuccess;
  hipError_t e = hipSuccess;
  if (!G.sched_items) e = hipMemsetAsync(G.counter, 0, sizeof(int), st);
  if (e != hipSuccess) return e;
  const bool f16 = arith == MLP_F16X2;
  const void* fn = waves == 8 ? (f16 ? (const void*)mlp_fused16<2, 8> : (const void*)mlp_fused16<3, 8>)
                              : (f16 ? (const void*)mlp_fused16<2, 4> : (const void*)mlp_fused16<3, 4>);
  {
    static std::set<std::pair<int, const void*>> raised;
    static std::mutex mtx;
    int dev = 0;
    e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mtx);
    if (!raised.count(std::make_pair(dev, fn))) {
      e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kGLds);
      if (e != hipSuccess) return e;
      raised.insert(std::make_pair(dev, fn));
    }
  }
  const int grid = G.sched_items ? G.sched_blocks : (total < ncu ? total : ncu);
  if (waves == 8) {
    if (f16) hipLaunchKernelGGL((mlp_fused16<2, 8>), dim3(grid), dim3(512), kGLds, st, G);
    else hipLaunchKernelGGL((mlp_fused16<3, 8>), dim3(grid), dim3(512), kGLds, st, G);
  } else {
    if (f16) hipLaunchKernelGGL((mlp_fused16<2, 4>), dim3(grid), dim3(256), kGLds, st, G);
    else hipLaunchKernelGGL((mlp_fused16<3, 4>), dim3(grid), dim3(256), kGLds, st, G);
  }
  return hipGetLastError();
}

}  // namespace ani
