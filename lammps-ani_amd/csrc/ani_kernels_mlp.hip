// ani_kernels_mlp.hip — species-bucketed MLP ensemble on fp32 MFMA (v_mfma_f32_32x32x2_f32, gfx950).
//
// Replaces BmmEnsemble + autograd through it (reference call sites models/lammps_ani.py:110,228-230,197-206).
// Every product here is a genuine dense contraction  C[rows][N] = A[rows][K] * Bt[N][K]^T :
//   forward  layer l : A = activations of layer l-1, Bt = W_l  ([out][in], torch.nn.Linear layout)
//   backward layer l : A = dE/dz_l,                  Bt = W_l^T ([in][out], transposed copy made at load time)
// so both operands are K-contiguous and one kernel serves all of them.
//
// Grouped launch: all species buckets (and ensemble members) of one layer go into ONE launch; a workgroup finds its
// (problem, row tile, member, column block) from its block index.  MFMA time per workgroup is long (tens of
// microseconds), so load balance across the 256 CUs is decided by tile granularity: WM selects 128-row tiles
// (4 waves stacked in M, each owning all columns) or 64-row tiles (2x2 waves, each owning half the columns; 3
// workgroups per CU) — the launcher picks the one with the smaller estimated makespan.
//
// Inside a workgroup: K is walked in slabs of 32 staged through LDS (row stride 36 floats = 144 B = 9*16 B, so the
// ds_read_b128 fragment reads are bank-conflict free); lane l (r = l&31, h = l>>5) reads 4 consecutive k of row r at
// k = 8*ks + 4*h and feeds them to 4 successive MFMAs — the k order inside a slab is permuted identically for A and
// B, which leaves the dot products unchanged.  The next slab's global loads are issued before the current slab's
// MFMAs (register prefetch).  MFMA C layout (guide §3): col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ani_kernels.h"
#include "ani_mlp_common.h"

namespace ani {


constexpr int LDS_LD = 36;       // floats per staged row (32 + 4 pad)

struct GroupArgs {
  GemmArgs p[kMaxProblems];
  int tile_start[kMaxProblems + 1];  // prefix of workgroups per problem
  int tiles_m[kMaxProblems];
  int nprob;
};

// Epilogues shared by the fp32-MFMA kernel and the split-bf16 kernel (same accumulator layout: col = lane & 31,
// row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)).  `lds` is the workgroup's staging memory, free by now.
// R: rows of the workgroup tile, WN: wave columns, wm: index of this accumulator's 32-row strip inside the tile.
// WT: the results are stored write-through to memory (device-scope stores), for a consumer on another XCD inside the same
// launch (mlp_pipeline_x2): the L2 caches of the eight XCDs are not coherent with each other before a kernel ends
template <bool WT>
__device__ __forceinline__ void store_c(float* p, float v) {
  if constexpr (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <int R, int WN, int EPI, int NTW, bool WT = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x16 (&acc)[NTW], float* lds, int b, int n0, int row0, int t0,
                                              int tcnt, int wm, int wn, int lr, int lh) {
  const int N = g.N;
  float* __restrict__ C = g.C + (long long)b * g.sC;
  const int mbase = row0 + 32 * wm + 4 * lh;
  if constexpr (EPI == EPI_PLAIN) {
#pragma unroll
    for (int nt = 0; nt < NTW; nt++) {
      const int n = n0 + 32 * (t0 + nt) + lr;
      if (nt < tcnt && n < N) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int m = mbase + (r & 3) + 8 * (r >> 2);
          store_c<WT>(&C[(long long)m * g.ldc + n], acc[nt][r]);
        }
      }
    }
  } else if constexpr (EPI == EPI_CELU) {
    const float* bias = g.bias + (long long)b * g.sBias;
#pragma unroll
    for (int nt = 0; nt < NTW; nt++) {
      const int n = n0 + 32 * (t0 + nt) + lr;
      if (nt < tcnt && n < N) {
        const float bv = bias[n];
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int m = mbase + (r & 3) + 8 * (r >> 2);
          store_c<WT>(&C[(long long)m * g.ldc + n], celu_f(acc[nt][r] + bv, g.alpha, g.inv_alpha));
        }
      }
    }
  } else if constexpr (EPI == EPI_BWD) {
    const float* H = g.aux + (long long)b * g.sAux;
#pragma unroll
    for (int nt = 0; nt < NTW; nt++) {
      const int n = n0 + 32 * (t0 + nt) + lr;
      if (nt < tcnt && n < N) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int m = mbase + (r & 3) + 8 * (r >> 2);
          const float h = H[(long long)m * g.ldaux + n];
          store_c<WT>(&C[(long long)m * g.ldc + n], acc[nt][r] * dcelu_from_h(h, g.inv_alpha));
        }
      }
    }
  } else {  // EPI_LAST: last hidden layer fused with the 1-wide output layer and the backward seed
    const float* bias = g.bias + (long long)b * g.sBias;
    const float* w = g.aux + (long long)b * g.sAux;
    float esum[16], valid[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
      esum[r] = 0.f;
      const int m = mbase + (r & 3) + 8 * (r >> 2);
      valid[r] = g.centre_of_row[m] >= 0 ? g.scale : 0.f;
    }
#pragma unroll
    for (int nt = 0; nt < NTW; nt++) {
      const int n = n0 + 32 * (t0 + nt) + lr;
      if (nt < tcnt && n < N) {
        const float bv = bias[n], wv = w[n];
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int m = mbase + (r & 3) + 8 * (r >> 2);
          const float h = celu_f(acc[nt][r] + bv, g.alpha, g.inv_alpha);
          esum[r] = fmaf(h, wv, esum[r]);
          // dE/dz = (1/M) * w_out * celu'(z)
          store_c<WT>(&C[(long long)m * g.ldc + n], valid[r] * wv * dcelu_from_h(h, g.inv_alpha));
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
      float v = esum[r];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 8);
      v += __shfl_xor(v, 4);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 1);
      esum[r] = v;
    }
    const float bl = g.bias_last[b];
    float* e_out = g.e_out + (long long)b * g.sE;
    if constexpr (WN == 1) {
      if (lr == 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int m = mbase + (r & 3) + 8 * (r >> 2);
          e_out[m] = valid[r] * (esum[r] + bl);
        }
      }
    } else {
      // row sums are split over the WN wave columns: combine through LDS (staging buffers are free now)
      __syncthreads();
      float* red = lds;  // [WN][R]
      if (lr == 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) red[wn * R + 32 * wm + 4 * lh + (r & 3) + 8 * (r >> 2)] = esum[r];
      }
      __syncthreads();
      if (wn == 0 && lr == 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int rl = 32 * wm + 4 * lh + (r & 3) + 8 * (r >> 2);
          float v = 0.f;
#pragma unroll
          for (int c = 0; c < WN; c++) v += red[c * R + rl];
          e_out[row0 + rl] = valid[r] * (v + bl);
        }
      }
    }
  }
}

#ifndef ANI_GEMM_LB2
#define ANI_GEMM_LB2 3  // workgroups per CU the 64-row variant is compiled for
#endif
template <int WM, int EPI>
__global__ __launch_bounds__(256, (WM == 4 ? 2 : (WM == 2 ? ANI_GEMM_LB2 : 3))) void gemm_grouped(GroupArgs G) {
  constexpr int WN = 4 / WM;        // waves along N
  constexpr int R = 32 * WM;        // rows per workgroup
  constexpr int NTW = 8 / WN;       // max 32-column tiles per wave
  __shared__ float lds[(R + 256) * LDS_LD];
  float* As = lds;
  float* Bs = lds + R * LDS_LD;

  // ---- which problem / tile ----
  int pi = 0;
  while (pi + 1 < G.nprob && (int)blockIdx.x >= G.tile_start[pi + 1]) pi++;
  const GemmArgs& g = G.p[pi];
  const int tiles_m = G.tiles_m[pi];
  int local = blockIdx.x - G.tile_start[pi];
  const int per_nb = tiles_m * g.batch;
  const int nb = local / per_nb;
  local -= nb * per_nb;
  // members of one row tile sit 8 block ids apart -> same XCD (round-robin dispatch), they share the A tile in L2
  const int grp = local / (8 * g.batch);
  const int rem = local - grp * (8 * g.batch);
  const int gs = min(8, tiles_m - grp * 8);
  const int b = rem / gs;
  const int tile_m = grp * 8 + (rem - b * gs);

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int lr = lane & 31, lh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int n0 = nb * 256;
  const int row0 = g.row0 + tile_m * R;
  const int K = g.K, N = g.N;
  const int ntiles = min(8, (N - n0 + 31) >> 5);           // 32-column tiles of this workgroup
  const int per = (ntiles + WN - 1) / WN;                   // tiles per wave column
  const int t0 = wn * per;                                  // first tile of this wave
  const int tcnt = max(0, min(per, ntiles - t0));           // tiles of this wave (wave-uniform)

  const float* __restrict__ A = g.A + (long long)b * g.sA + (long long)row0 * g.lda;
  const float* __restrict__ Bt = g.Bt + (long long)b * g.sB + (long long)n0 * g.ldb;

  f32x16 acc[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; nt++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[nt][r] = 0.f;

  constexpr int PA = R / 32;  // float4 per thread for the A slab
  float4 pa[PA], pb[8], pm[PA];
  const float* __restrict__ Am = g.Amask ? g.Amask + (long long)b * g.sA + (long long)row0 * g.lda : nullptr;
  const int cr = tid >> 3;         // staged row handled by this thread (per 32-row group)
  const int ck = (tid & 7) * 4;    // k offset inside the slab
  const int brows = 32 * ntiles;

  auto gload = [&](int k0) {
    const int kc = k0 + ck;
    const bool kin = kc < K;
#pragma unroll
    for (int i = 0; i < PA; i++) {
      const int r = cr + 32 * i;
      pa[i] = kin ? *reinterpret_cast<const float4*>(A + (long long)r * g.lda + kc) : make_float4(0, 0, 0, 0);
      if (Am) pm[i] = kin ? *reinterpret_cast<const float4*>(Am + (long long)r * g.lda + kc) : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int r = cr + 32 * i;
      if (r < brows)  // wave-uniform per i
        pb[i] = (kin && (n0 + r) < N) ? *reinterpret_cast<const float4*>(Bt + (long long)r * g.ldb + kc) : make_float4(0, 0, 0, 0);
    }
  };

  const int nkt = (K + 31) >> 5;
  // Every workgroup of a problem streams the SAME Bt slabs; started together they would all hit the same L2 lines
  // (one channel) at the same time.  Each workgroup therefore walks K from its own starting slab and wraps around:
  // the sum over k is the same set of terms in a rotated order.
#ifdef ANI_GEMM_NO_KROT
  const int rot = 0;
#else
  const int rot = (int)((blockIdx.x * 11u) % (unsigned)nkt);
#endif
  auto slab_k0 = [&](int kt) { int t = kt + rot; if (t >= nkt) t -= nkt; return t << 5; };
  gload(slab_k0(0));
  for (int kt = 0; kt < nkt; kt++) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PA; i++) {
      if (Am) {
        pa[i].x *= dcelu_from_h(pm[i].x, g.inv_alpha); pa[i].y *= dcelu_from_h(pm[i].y, g.inv_alpha);
        pa[i].z *= dcelu_from_h(pm[i].z, g.inv_alpha); pa[i].w *= dcelu_from_h(pm[i].w, g.inv_alpha);
      }
      *reinterpret_cast<float4*>(As + (cr + 32 * i) * LDS_LD + ck) = pa[i];
    }
#pragma unroll
    for (int i = 0; i < 8; i++)
      if (cr + 32 * i < brows) *reinterpret_cast<float4*>(Bs + (cr + 32 * i) * LDS_LD + ck) = pb[i];
    __syncthreads();
#ifndef ABL_NO_GLOAD
    if (kt + 1 < nkt) gload(slab_k0(kt + 1));
#endif
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      const float4 a4 = *reinterpret_cast<const float4*>(As + (32 * wm + lr) * LDS_LD + ks * 8 + 4 * lh);
#pragma unroll
      for (int nt = 0; nt < NTW; nt++) {
        if (nt < tcnt) {
          const float4 b4 = *reinterpret_cast<const float4*>(Bs + (32 * (t0 + nt) + lr) * LDS_LD + ks * 8 + 4 * lh);
#ifdef ABL_NO_MFMA
          asm volatile("" ::"v"(a4.x), "v"(b4.x), "v"(b4.w), "v"(a4.w));
          continue;
#endif
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc[nt], 0, 0, 0);
        }
      }
    }
  }

  gemm_epilogue<32 * WM, 4 / WM, EPI>(g, acc, lds, b, n0, row0, t0, tcnt, wm, wn, lr, lh);
}

// ---------------------------------------------------------------------------------------------------------------
// Split-bf16 variant.  Every fp32 number is EXACTLY the sum of three bf16 numbers (8 + 8 + 8 mantissa bits, same
// exponent range): hi = x with the low 16 bits cleared, mid = (x - hi) likewise, lo = x - hi - mid.  A product a*b is
// then the sum of nine bf16 x bf16 products, each exact in fp32; the six of them above 2^-24 relative size
// (hh, hm, mh, hl, lh, mm) are evaluated with v_mfma_f32_32x32x16_bf16 and accumulated in fp32, the three dropped
// ones (ml, lm, ll) sum to less than 2^-23 |a b| -- the size of one fp32 rounding.  On gfx950 a bf16 MFMA does 16x the
// multiply-adds of an fp32 MFMA per cycle, so six of them cost 3/8 of the fp32-input instruction.
// Weights are split once (launch_split_bf16x3); activations are split while they are staged into LDS.
// LDS image of a staged row: [hi 16 k | mid 16 k | lo 16 k] bf16 = 96 B + 16 B pad (stride 112 B: conflict-free
// ds_read_b128).  K is walked in slabs of 16 (one MFMA k-block).
// ---------------------------------------------------------------------------------------------------------------
// Two-term fp16 variant (P = 2 below, the default: option "mlp_arith").  x * 2^s = h + l with two fp16 numbers, both
// rounded to nearest: |x 2^s - h - l| <= 2^-22 |x 2^s| while l is a normal fp16 number (|x 2^s| >= 2^-3), 2^-25 absolute
// below that -- which is why the operands are scaled by powers of two first (exact; undone on the accumulators, exact):
// weights so that the largest of a layer sits below 2^13, activations by 2^4, gradients by 2^12 (GemmArgs::a_scale,
// inv_scale).  Three v_mfma_f32_32x32x16_f16 products (hh, hl, lh; the dropped ll is below 2^-22 |a b|) instead of six:
// operands good to 22-23 bits instead of 24, in sums whose fp32 accumulation over K = 128..1008 terms already carries
// ~sqrt(K) 2^-24.  Against the fp64 oracle on the 100 002-atom box the forces are as close as with the exact split
// (bench.py "parity"; tests/test_hip_properties.py).  Magnitudes beyond 65504 / scale (activations 4094, gradients 16
// Hartree per unit) become inf and the energy NaN: loud, not wrong.  LDS row: [h 16 k | l 16 k] = 64 B + 16 B pad.
// P = planes of a staged operand: 3 = bf16 hi/mid/lo (exact), 2 = fp16 h/l of the scaled value
template <int P>
__device__ __forceinline__ void stage_a4(unsigned char* dst, float4 av, float a_scale) {   // four consecutive k of one row
  if constexpr (P == 3) {
    unsigned h[4], m[4], l[4];
    split3(av.x, h[0], m[0], l[0]); split3(av.y, h[1], m[1], l[1]);
    split3(av.z, h[2], m[2], l[2]); split3(av.w, h[3], m[3], l[3]);
    *reinterpret_cast<uint2*>(dst) = make_uint2(pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3]));
    *reinterpret_cast<uint2*>(dst + 32) = make_uint2(pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3]));
    *reinterpret_cast<uint2*>(dst + 64) = make_uint2(pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3]));
  } else {
    unsigned h0, l0, h1, l1;
    split2(av.x * a_scale, av.y * a_scale, h0, l0);
    split2(av.z * a_scale, av.w * a_scale, h1, l1);
    *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(dst + 32) = make_uint2(l0, l1);
  }
}
template <int P, int NTW>
__device__ __forceinline__ void unscale(f32x16 (&acc)[NTW], float inv_scale) {
  if constexpr (P == 2) {
#pragma unroll
    for (int nt = 0; nt < NTW; nt++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[nt][r] *= inv_scale;
  }
}

// RT: 32-row strips per wave (register blocking in M).  The kernel is bound by LDS traffic, not by the MFMA pipe: with
// RT = 2 every Bt fragment read from LDS feeds two row strips.
template <int WM, int EPI, int KB, int RT, int P>
__global__ __launch_bounds__(256, ((KB == 2 || RT == 2) ? 2 : (WM == 4 ? 2 : 3))) void gemm_grouped_x3(GroupArgs G) {
  constexpr int WN = 4 / WM;
  constexpr int R = 32 * WM * RT;
  constexpr int NTW = 8 / WN;
  constexpr int PB = 32 * P;          // bytes of one k-block of a staged row
  constexpr int ROW = KB * PB + 16;   // bytes per staged row: KB k-blocks of P planes x 16 k, + pad (112 / 80 B: conflict-free)
  constexpr int NC = 2 * P;           // 16-byte chunks per row and k-block
  // ST = 2: two LDS stages -- slab k+1 is written while slab k is multiplied, ONE barrier per slab instead of two;
  // affordable with two planes (2 x 25.6 KB per workgroup: three workgroups per CU still fit), not with three.
  // 100 002 atoms: MLP 0.330 -> 0.318 ms.  (Requesting the loads of slab k+2 at the top of the iteration, a second
  // register set: 0.331.)
#ifndef ANI_X3_STAGES
#define ANI_X3_STAGES 2
#endif
  constexpr int ST = (P == 2 && KB == 1 && RT == 1) ? ANI_X3_STAGES : 1;
  constexpr int STAGE = (R + 256) * ROW;
  __shared__ uint4 lds4[ST * STAGE / 16];
  unsigned char* As = reinterpret_cast<unsigned char*>(lds4);
  unsigned char* Bs = As + R * ROW;

  int pi = 0;
  while (pi + 1 < G.nprob && (int)blockIdx.x >= G.tile_start[pi + 1]) pi++;
  const GemmArgs& g = G.p[pi];
  const int tiles_m = G.tiles_m[pi];
  int local = blockIdx.x - G.tile_start[pi];
  const int per_nb = tiles_m * g.batch;
  const int nb = local / per_nb;
  local -= nb * per_nb;
  const int grp = local / (8 * g.batch);
  const int rem = local - grp * (8 * g.batch);
  const int gs = min(8, tiles_m - grp * 8);
  const int b = rem / gs;
  const int tile_m = grp * 8 + (rem - b * gs);

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int lr = lane & 31, lh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int n0 = nb * 256;
  const int row0 = g.row0 + tile_m * R;
  const int K = g.K, N = g.N;
  const int ntiles = min(8, (N - n0 + 31) >> 5);
  const int per = (ntiles + WN - 1) / WN;
  const int t0 = wn * per;
  const int tcnt = max(0, min(per, ntiles - t0));

  const float* __restrict__ A = g.A + (long long)b * g.sA + (long long)row0 * g.lda;
  // Bt planes are stored k-block major, [kb][N][plane][16]: the slab of one k-block is contiguous over the rows
  const uint4* __restrict__ B3 = reinterpret_cast<const uint4*>(g.Btp + (long long)b * g.sBp);

  f32x16 acc[RT][NTW];
#pragma unroll
  for (int rt = 0; rt < RT; rt++)
#pragma unroll
    for (int nt = 0; nt < NTW; nt++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[rt][nt][r] = 0.f;

  constexpr int PA = R >= 64 ? R / 64 : 1;   // A rows per thread
  float4 pa[PA][KB];
  uint4 pb[NC * KB];
  const int ar = tid >> 2;          // A row (per 64-row group)
  const int ak = (tid & 3) * 4;     // k offset inside a k-block
  const int brows = min(32 * ntiles, N - n0);   // Bt rows staged by this workgroup, as NC * brows 16-byte chunks

  // A (activations) streams from HBM / Infinity Cache, Bt (weights) from L2: A is fetched two slabs ahead, Bt one
  float4 pan[PA][KB], pm[PA][KB], pmn[PA][KB];
  const float* __restrict__ Am = g.Amask ? g.Amask + (long long)b * g.sA + (long long)row0 * g.lda : nullptr;
  auto gloadA = [&](int kb0, float4 (&dst)[PA][KB], float4 (&dstm)[PA][KB]) {
#pragma unroll
    for (int j = 0; j < KB; j++) {
      const int kc = (kb0 + j) * 16 + ak;
      const bool kin = kc < K;
#pragma unroll
      for (int i = 0; i < PA; i++) {
        const int r = ar + 64 * i;
        const bool in = kin && r < R;
        dst[i][j] = in ? *reinterpret_cast<const float4*>(A + (long long)r * g.lda + kc) : make_float4(0, 0, 0, 0);
        if (Am) dstm[i][j] = in ? *reinterpret_cast<const float4*>(Am + (long long)r * g.lda + kc) : make_float4(0, 0, 0, 0);
      }
    }
  };
  auto gloadB = [&](int kb0) {
#pragma unroll
    for (int j = 0; j < KB; j++) {
      const uint4* src = B3 + ((long long)(kb0 + j) * N + n0) * NC;
      const bool bin = kb0 + j < g.kbp;
#pragma unroll
      for (int i = 0; i < NC; i++) {
        const int c = tid + 256 * i;   // consecutive lanes, consecutive 16-byte chunks
        pb[NC * j + i] = (bin && c < NC * brows) ? src[c] : make_uint4(0, 0, 0, 0);
      }
    }
  };

  const int nkt = (g.kbp + KB - 1) / KB;
#ifdef ANI_GEMM_NO_KROT
  const int rot = 0;
#else
  const int rot = (int)((blockIdx.x * 11u) % (unsigned)nkt);
#endif
  auto slab = [&](int kt) { int t = kt + rot; if (t >= nkt) t -= nkt; return t * KB; };
  auto stage_write = [&](unsigned char* As, unsigned char* Bs) {
#pragma unroll
    for (int i = 0; i < PA; i++) {
      const int r = ar + 64 * i;
      if (r < R) {
#pragma unroll
        for (int j = 0; j < KB; j++) {
          float4 av = pa[i][j];
          if (Am) {
            av.x *= dcelu_from_h(pm[i][j].x, g.inv_alpha); av.y *= dcelu_from_h(pm[i][j].y, g.inv_alpha);
            av.z *= dcelu_from_h(pm[i][j].z, g.inv_alpha); av.w *= dcelu_from_h(pm[i][j].w, g.inv_alpha);
          }
          stage_a4<P>(As + r * ROW + j * PB + ak * 2, av, g.a_scale);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < KB; j++)
#pragma unroll
      for (int i = 0; i < NC; i++) {
        const int c = tid + 256 * i;
        const int r = c / NC, q = c - NC * r;
        if (r < 32 * ntiles) *reinterpret_cast<uint4*>(Bs + r * ROW + j * PB + q * 16) = pb[NC * j + i];
      }
  };
  auto multiply = [&](const unsigned char* As, const unsigned char* Bs) {
#pragma unroll
    for (int j = 0; j < KB; j++) {
      Frag<P> af[RT];
#pragma unroll
      for (int rt = 0; rt < RT; rt++) af[rt] = load_frag<P>(As + (32 * (wm * RT + rt) + lr) * ROW + j * PB + lh * 16);
#pragma unroll
      for (int nt = 0; nt < NTW; nt++) {
        if (nt < tcnt) {
          const Frag<P> bf = load_frag<P>(Bs + (32 * (t0 + nt) + lr) * ROW + j * PB + lh * 16);
#pragma unroll
          for (int rt = 0; rt < RT; rt++) mma_planes<P>(af[rt], bf, acc[rt][nt]);
        }
      }
    }
  };
  if constexpr (ST == 2) {
    unsigned char* base = reinterpret_cast<unsigned char*>(lds4);
    gloadA(slab(0), pa, pm);
    gloadB(slab(0));
    stage_write(base, base + R * ROW);
    if (nkt > 1) { gloadA(slab(1), pa, pm); gloadB(slab(1)); }
    __syncthreads();
    for (int kt = 0; kt < nkt; kt++) {
      const unsigned char* cur = base + (kt & 1) * STAGE;
      unsigned char* nxt = base + ((kt + 1) & 1) * STAGE;
      multiply(cur, cur + R * ROW);
      if (kt + 1 < nkt) stage_write(nxt, nxt + R * ROW);     // loads requested one slab ago
      if (kt + 2 < nkt) { gloadA(slab(kt + 2), pa, pm); gloadB(slab(kt + 2)); }
      __syncthreads();
    }
  } else {
  gloadA(slab(0), pa, pm);
  gloadB(slab(0));
  if (nkt > 1) gloadA(slab(1), pan, pmn);
  for (int kt = 0; kt < nkt; kt++) {
    __syncthreads();
    stage_write(As, Bs);
    __syncthreads();
#ifndef ABLX_NO_GLOAD
    if (kt + 1 < nkt) gloadB(slab(kt + 1));
#pragma unroll
    for (int i = 0; i < PA; i++)
#pragma unroll
      for (int j = 0; j < KB; j++) { pa[i][j] = pan[i][j]; pm[i][j] = pmn[i][j]; }
    if (kt + 2 < nkt) gloadA(slab(kt + 2), pan, pmn);
#endif
    multiply(As, Bs);
  }
  __syncthreads();
  }
#pragma unroll
  for (int rt = 0; rt < RT; rt++) unscale<P>(acc[rt], g.inv_scale);
#pragma unroll
  for (int rt = 0; rt < RT; rt++)
    gemm_epilogue<R, WN, EPI>(g, acc[rt], reinterpret_cast<float*>(lds4), b, n0, row0, t0, tcnt, wm * RT + rt, wn, lr, lh);
}

// ---------------------------------------------------------------------------------------------------------------
// Chained MLP for small systems: one workgroup carries a 64-row tile through all the layers (see ani_kernels.h).
// The tile loop of a layer is the one of gemm_grouped_x3<2, *, 1, 1>.
// ---------------------------------------------------------------------------------------------------------------
// WM = 2: 64-row tiles.  NWV = 4: 2x2 waves, two workgroups per CU.  NWV = 8: 2x4 waves, ONE 512-thread workgroup per
// CU, for systems with no more tiles than CUs: every tile streams all the weights from L2 (1.35 MB for the six ANI-2x
// products of a species), so 32-row tiles for twice the workgroups doubled that traffic (5.9 TB/s at 12 500 atoms), while
// a lone 4-wave workgroup on 64 rows ran each wave's instruction stream twice as long; eight waves halve both.
// KB: k-blocks (16 k each) per slab.  The lone eight-wave workgroup of a small system is bound by the length of its
// chain of slab steps (load -> LDS -> barrier -> fragments -> MFMA), not by any throughput: KB = 2 halves the steps.
template <int WM, int NWV, int P, int KB>
__global__ __launch_bounds__(64 * NWV, NWV == 8 ? 1 : 2) void mlp_chain_x3(const GemmArgs* __restrict__ layers, const int* __restrict__ epi,
                                                                           const int* __restrict__ tile_start, int nlayers, int nprob) {
  constexpr int NT = 64 * NWV, WN = NWV / WM, R = 32 * WM, NTW = 8 / WN, PB = 32 * P, ROW = KB * PB + 16, NC = 2 * P;
  constexpr int NB = NC * 256 / NT;   // 16-byte chunks of one k-block of a Bt slab per thread
  constexpr int AL = (R * 4 * KB + NT - 1) / NT;   // float4 items of an A slab (row, k-block, 4 k) per thread
  // two LDS stages: slab k+1 is written while slab k is multiplied and slab k+2 is in flight in registers -- one barrier
  // per slab, and every load has a whole slab to arrive.  (At most two of these workgroups share a CU.)
  constexpr int STAGE = (R + 256) * ROW;
  __shared__ uint4 lds4[2 * STAGE / 16];
  unsigned char* lds = reinterpret_cast<unsigned char*>(lds4);
  int pi = 0;
  while (pi + 1 < nprob && (int)blockIdx.x >= tile_start[pi + 1]) pi++;
  const int tile_m = blockIdx.x - tile_start[pi];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int lr = lane & 31, lh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int ak = (tid & 3) * 4;

  for (int l = 0; l < nlayers; l++) {
    const GemmArgs g = layers[l * nprob + pi];
    const int e = epi[l];
    const int row0 = g.row0 + tile_m * R;
    const int K = g.K, N = g.N;
    const float* __restrict__ A = g.A + (long long)row0 * g.lda;
    const float* __restrict__ Am = g.Amask ? g.Amask + (long long)row0 * g.lda : nullptr;
    const uint4* __restrict__ B3 = reinterpret_cast<const uint4*>(g.Btp);
    const int nkt = (g.kbp + KB - 1) / KB;
    for (int n0 = 0; n0 < N; n0 += 256) {
      const int ntiles = min(8, (N - n0 + 31) >> 5);
      const int per = (ntiles + WN - 1) / WN;
      const int t0 = wn * per;
      const int tcnt = max(0, min(per, ntiles - t0));
      const int brows = min(32 * ntiles, N - n0);
      f32x16 acc[NTW];
#pragma unroll
      for (int nt = 0; nt < NTW; nt++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[nt][r] = 0.f;
      // Two register sets of loads in flight: the set written to LDS in step kt was requested in step kt - 2.  A lone
      // workgroup per CU (small systems) is bound by the latency of these loads, not by the matrix pipe: with one set
      // (one slab of lead) a slab took ~1.2 us whatever the tile height.
      struct Regs { float4 pa[AL], pm[AL]; uint4 pb[KB][NB]; };
      Regs r0, r1;
      // item id of an A slab: 4-k chunk (id & 3), row (id >> 2) % R, k-block id / (4 R)
      auto gload = [&](int kt, Regs& q) {
#pragma unroll
        for (int i = 0; i < AL; i++) {
          const int id = tid + NT * i, ar = (id >> 2) % R, j = id / (4 * R);
          const int kc = (kt * KB + j) * 16 + ak;
          const bool in = kc < K && j < KB;
          q.pa[i] = in ? *reinterpret_cast<const float4*>(A + (long long)ar * g.lda + kc) : make_float4(0, 0, 0, 0);
          if (Am) q.pm[i] = in ? *reinterpret_cast<const float4*>(Am + (long long)ar * g.lda + kc) : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < KB; j++) {
          const int kb = kt * KB + j;
          const uint4* src = B3 + ((long long)kb * N + n0) * NC;
#pragma unroll
          for (int i = 0; i < NB; i++) {
            const int c = tid + NT * i;
            q.pb[j][i] = (kb < g.kbp && c < NC * brows) ? src[c] : make_uint4(0, 0, 0, 0);
          }
        }
      };
      auto stage_write = [&](int st, const Regs& q) {
        unsigned char* As = lds + st * STAGE;
        unsigned char* Bs = As + R * ROW;
#pragma unroll
        for (int i = 0; i < AL; i++) {
          const int id = tid + NT * i, ar = (id >> 2) % R, j = id / (4 * R);
          if (j < KB) {
            float4 av = q.pa[i];
            if (Am) {
              av.x *= dcelu_from_h(q.pm[i].x, g.inv_alpha); av.y *= dcelu_from_h(q.pm[i].y, g.inv_alpha);
              av.z *= dcelu_from_h(q.pm[i].z, g.inv_alpha); av.w *= dcelu_from_h(q.pm[i].w, g.inv_alpha);
            }
            stage_a4<P>(As + ar * ROW + j * PB + ak * 2, av, g.a_scale);
          }
        }
#pragma unroll
        for (int j = 0; j < KB; j++)
#pragma unroll
          for (int i = 0; i < NB; i++) {
            const int c = tid + NT * i;
            const int r = c / NC, q6 = c - NC * r;
            if (r < 32 * ntiles) *reinterpret_cast<uint4*>(Bs + r * ROW + j * PB + q6 * 16) = q.pb[j][i];
          }
      };
      auto multiply = [&](int st) {
        const unsigned char* As = lds + st * STAGE;
        const unsigned char* Bs = As + R * ROW;
#pragma unroll
        for (int j = 0; j < KB; j++) {
          const Frag<P> af = load_frag<P>(As + (32 * wm + lr) * ROW + j * PB + lh * 16);
#pragma unroll
          for (int nt = 0; nt < NTW; nt++) {
            if (nt < tcnt) {
              const Frag<P> bf = load_frag<P>(Bs + (32 * (t0 + nt) + lr) * ROW + j * PB + lh * 16);
              mma_planes<P>(af, bf, acc[nt]);
            }
          }
        }
      };
      gload(0, r0);
      stage_write(0, r0);
      if (nkt > 1) gload(1, r0);
      if (nkt > 2) gload(2, r1);
      __syncthreads();
      for (int kt = 0; kt < nkt; kt += 2) {
        // slab kt sits in stage 0, r0 holds slab kt + 1
        multiply(0);
        if (kt + 1 < nkt) stage_write(1, r0);
        if (kt + 3 < nkt) gload(kt + 3, r0);
        __syncthreads();
        if (kt + 1 >= nkt) break;
        // slab kt + 1 sits in stage 1, r1 holds slab kt + 2
        multiply(1);
        if (kt + 2 < nkt) stage_write(0, r1);
        if (kt + 4 < nkt) gload(kt + 4, r1);
        __syncthreads();
      }
      unscale<P>(acc, g.inv_scale);
      float* ldsf = reinterpret_cast<float*>(lds4);
      if (e == EPI_CELU) gemm_epilogue<R, WN, EPI_CELU>(g, acc, ldsf, 0, n0, row0, t0, tcnt, wm, wn, lr, lh);
      else if (e == EPI_LAST) gemm_epilogue<R, WN, EPI_LAST>(g, acc, ldsf, 0, n0, row0, t0, tcnt, wm, wn, lr, lh);
      else gemm_epilogue<R, WN, EPI_PLAIN>(g, acc, ldsf, 0, n0, row0, t0, tcnt, wm, wn, lr, lh);
      __syncthreads();   // the staging memory is reused (and, for EPI_LAST, was scratch of the epilogue)
    }
    // the next layer of this tile reads what this workgroup just stored: the barrier above carries the workgroup-scope
    // release/acquire (stores complete, same CU, same L1)
  }
}

// ---------------------------------------------------------------------------------------------------------------
// All layers in ONE launch for large systems: persistent workgroups walk the items (layer, 64-row tile) in layer-major
// order, item w to workgroup w mod G.  Tile t of layer l needs only tile t of layer l - 1 (same rows), finished a whole
// layer of items earlier by another workgroup: a flag per item (release / acquire at device scope) carries the
// dependency, and no launch has a tail of its own -- six launches of 1563 tiles on 768 slots are six third rounds of 27
// tiles, this is 9378 items on 768 slots.  One ensemble member, layers no wider than 256, two-term arithmetic.
// Items are drawn from counters (one returning atomic per item, requested while the previous item is being multiplied):
// eight of them, counter c handing out the items c, c + 8, c + 16, ... to the workgroups with blockIdx % 8 == c (one
// counter for all 768 workgroups cost 13 % of the kernel at 50 000 atoms; a static deal, item w to workgroup w mod G, is
// as fast but deadlocks when something else keeps part of the grid from being resident).  Every workgroup takes its items
// in increasing order and waits only for smaller ones, the dispatcher starts workgroups in index order (so every counter
// has running takers): the smallest unfinished item can always proceed, no wait is circular.  A wait is bounded all the
// same (about a second) and then raises the error flag.
// The tile loop is the one of gemm_grouped_x3<2, *, 1, 1, 2> (two LDS stages, one barrier per slab).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 3) void mlp_pipeline_x2(const GemmArgs* __restrict__ layers, const int* __restrict__ epi,
                                                         const int* __restrict__ tile_start, int nlayers, int nprob, int total,
                                                         int* __restrict__ done, int* __restrict__ err_flag) {
  constexpr int P = 2, WN = 2, R = 64, NTW = 8 / WN, PB = 32 * P, ROW = PB + 16, NC = 2 * P, NB = NC;
  constexpr int STAGE = (R + 256) * ROW;
  __shared__ uint4 lds4[2 * STAGE / 16];
  unsigned char* base = reinterpret_cast<unsigned char*>(lds4);
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int lr = lane & 31, lh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int ar = tid >> 2, ak = (tid & 3) * 4;
  __shared__ int s_next;
  const int cq = blockIdx.x & 7;
  int* ctr = done + (size_t)nlayers * total + cq;   // the eight item counters follow the flags
  if (tid == 0) s_next = cq + 8 * atomicAdd(ctr, 1);
  __syncthreads();
  int w = s_next;
  while (w < nlayers * total) {
    int nxt = 0;
    if (tid == 0) nxt = cq + 8 * atomicAdd(ctr, 1);   // the next item of this workgroup: in flight while this one is computed
    const int l = w / total, t = w - l * total;
    int pi = 0;
    while (pi + 1 < nprob && t >= tile_start[pi + 1]) pi++;
    const int tile_m = t - tile_start[pi];
    if (l > 0) {
      if (tid == 0) {
        const int* f = done + (size_t)(l - 1) * total + t;
        int spins = 0;
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && ++spins < (1 << 21)) __builtin_amdgcn_s_sleep(16);
        if (spins >= (1 << 21)) atomicOr(err_flag, 2);
        // ONE agent-scope acquire after the poll has matched (buffer_inv sc1: drops this CU's L1 lines, whatever touched them
        // earlier in the launch), waited for before the barrier releases the other waves' loads
        // (cdna_hip_programming.md, Guideline 16).  The producer side needs no release fence: every result was stored
        // write-through (sc1) and acknowledged (s_waitcnt 0 in every storing wave, then the barrier) before the flag.
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
    }
    const GemmArgs g = layers[l * nprob + pi];
    const int e = epi[l];
    const int row0 = g.row0 + tile_m * R;
    const int K = g.K, N = g.N;
    const int ntiles = min(8, (N + 31) >> 5);
    const int per = (ntiles + WN - 1) / WN;
    const int t0 = wn * per;
    const int tcnt = max(0, min(per, ntiles - t0));
    const int brows = min(32 * ntiles, N);
    const float* __restrict__ A = g.A + (long long)row0 * g.lda;
    const float* __restrict__ Am = g.Amask ? g.Amask + (long long)row0 * g.lda : nullptr;
    const uint4* __restrict__ B3 = reinterpret_cast<const uint4*>(g.Btp);
    f32x16 acc[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; nt++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[nt][r] = 0.f;
    float4 pa, pm;
    uint4 pb[NB];
    auto gload = [&](int kb) {
      const int kc = kb * 16 + ak;
      const bool in = kc < K;
      pa = in ? *reinterpret_cast<const float4*>(A + (long long)ar * g.lda + kc) : make_float4(0, 0, 0, 0);
      if (Am) pm = in ? *reinterpret_cast<const float4*>(Am + (long long)ar * g.lda + kc) : make_float4(0, 0, 0, 0);
      const uint4* src = B3 + (long long)kb * N * NC;
#pragma unroll
      for (int i = 0; i < NB; i++) {
        const int c = tid + 256 * i;
        pb[i] = (kb < g.kbp && c < NC * brows) ? src[c] : make_uint4(0, 0, 0, 0);
      }
    };
    auto stage_write = [&](unsigned char* As) {
      unsigned char* Bs = As + R * ROW;
      float4 av = pa;
      if (Am) {
        av.x *= dcelu_from_h(pm.x, g.inv_alpha); av.y *= dcelu_from_h(pm.y, g.inv_alpha);
        av.z *= dcelu_from_h(pm.z, g.inv_alpha); av.w *= dcelu_from_h(pm.w, g.inv_alpha);
      }
      stage_a4<P>(As + ar * ROW + ak * 2, av, g.a_scale);
#pragma unroll
      for (int i = 0; i < NB; i++) {
        const int c = tid + 256 * i;
        const int r = c / NC, q = c - NC * r;
        if (r < 32 * ntiles) *reinterpret_cast<uint4*>(Bs + r * ROW + q * 16) = pb[i];
      }
    };
    auto multiply = [&](const unsigned char* As) {
      const unsigned char* Bs = As + R * ROW;
      const Frag<P> af = load_frag<P>(As + (32 * wm + lr) * ROW + lh * 16);
#pragma unroll
      for (int nt = 0; nt < NTW; nt++) {
        if (nt < tcnt) {
          const Frag<P> bf = load_frag<P>(Bs + (32 * (t0 + nt) + lr) * ROW + lh * 16);
          mma_planes<P>(af, bf, acc[nt]);
        }
      }
    };
    const int nkt = g.kbp;
    const int rot = (int)(((unsigned)w * 11u) % (unsigned)nkt);
    auto slab = [&](int kt) { int q = kt + rot; if (q >= nkt) q -= nkt; return q; };
    gload(slab(0));
    stage_write(base);
    if (nkt > 1) gload(slab(1));
    __syncthreads();
    for (int kt = 0; kt < nkt; kt++) {
      multiply(base + (kt & 1) * STAGE);
      if (kt + 1 < nkt) stage_write(base + ((kt + 1) & 1) * STAGE);
      if (kt + 2 < nkt) gload(slab(kt + 2));
      __syncthreads();
    }
    unscale<P>(acc, g.inv_scale);
    float* ldsf = reinterpret_cast<float*>(lds4);
    if (e == EPI_CELU) gemm_epilogue<R, WN, EPI_CELU, NTW, true>(g, acc, ldsf, 0, 0, row0, t0, tcnt, wm, wn, lr, lh);
    else if (e == EPI_LAST) gemm_epilogue<R, WN, EPI_LAST, NTW, true>(g, acc, ldsf, 0, 0, row0, t0, tcnt, wm, wn, lr, lh);
    else gemm_epilogue<R, WN, EPI_PLAIN, NTW, true>(g, acc, ldsf, 0, 0, row0, t0, tcnt, wm, wn, lr, lh);
    // this tile of this layer is complete: every thread's (write-through) stores have been acknowledged, then the flag.
    // No device-scope release fence: on this chip it writes back the XCD's whole L2 (measured: the kernel 6x slower)
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt = lgkmcnt = expcnt = 0
    if (tid == 0) s_next = nxt;
    __syncthreads();   // also: the staging memory (scratch of the LAST epilogue) is reused by the next item
    if (tid == 0) __hip_atomic_store(done + (size_t)l * total + t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    w = s_next;
  }
}

int mlp_chain_slots() {
  static const int n = [] {
    int dev = 0, v = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
    return 2 * (v > 0 ? v : 256);
  }();
  return n;
}

void free_chain_plan(ChainPlan& p) {
  if (p.d_done) (void)hipFree(p.d_done);
  p.d_done = nullptr; p.done_n = 0;
  if (p.d_desc) (void)hipFree(p.d_desc);
  p.d_desc = nullptr; p.bytes = 0; p.host.clear();
}

hipError_t launch_mlp_chain(const GemmArgs* layers, const int* epi, int nlayers, int nprob, ChainPlan* plan, hipStream_t st, MlpArith arith,
                            bool pipeline, int* err_flag) {
  std::vector<int> tile_start(nprob + 1, 0);
  static const int forced_r = [] { const char* e = getenv("ANI_CHAIN_WAVES"); return e ? atoi(e) : 0; }();   // experiment knob: 4 / 8
  const int R = 64;
  for (int i = 0; i < nprob; i++) tile_start[i + 1] = tile_start[i] + layers[i].rows / R;
  const int total = tile_start[nprob];
  if (total <= 0) return hipSuccess;
  const size_t b0 = sizeof(GemmArgs) * (size_t)nlayers * nprob, b1 = sizeof(int) * (size_t)nlayers, b2 = sizeof(int) * (size_t)(nprob + 1);
  std::vector<unsigned char> host(b0 + b1 + b2);
  memcpy(host.data(), layers, b0);
  memcpy(host.data() + b0, epi, b1);
  memcpy(host.data() + b0 + b1, tile_start.data(), b2);
  if (host != plan->host) {   // descriptors change only with the list epoch / options
    if (plan->bytes < host.size()) {
      if (plan->d_desc) (void)hipFree(plan->d_desc);
      plan->d_desc = nullptr;
      const hipError_t e = hipMalloc(&plan->d_desc, host.size());
      if (e != hipSuccess) { plan->d_desc = nullptr; plan->bytes = 0; plan->host.clear(); return e; }
      plan->bytes = host.size();
    }
    // a synchronous copy from the pageable vector (a few KB, once per list epoch): the asynchronous form would read
    // plan->host whenever the DMA gets to it, possibly after the next epoch has reassigned it
    const hipError_t e = hipMemcpy(plan->d_desc, host.data(), host.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { plan->host.clear(); return e; }
    plan->host = host;
  }
  const unsigned char* d = reinterpret_cast<const unsigned char*>(plan->d_desc);
  if (pipeline) {
    // The hand-off between workgroups is per 64-row tile of C: no 128-byte line may hold rows of two tiles, or a consumer
    // could keep a line that a second producer completes later.  Row strides that are multiples of 32 floats on 128-byte
    // aligned bases guarantee it; the caller (compute_mlp) selects this path only then, and it is checked again here.
    for (int i = 0; i < nlayers * nprob; i++) {
      const GemmArgs& g = layers[i];
      if ((g.ldc & 31) || (reinterpret_cast<uintptr_t>(g.C) & 127) || (g.lda & 3) || (reinterpret_cast<uintptr_t>(g.A) & 15))
        return hipErrorInvalidValue;
    }
    const size_t nitems = (size_t)nlayers * total, nflags = nitems + 8;   // + the item counters
    if (plan->done_n < nflags) {
      if (plan->d_done) (void)hipFree(plan->d_done);
      plan->d_done = nullptr; plan->done_n = 0;
      const hipError_t e = hipMalloc((void**)&plan->d_done, nflags * sizeof(int));
      if (e != hipSuccess) { plan->d_done = nullptr; return e; }
      plan->done_n = nflags;
    }
    hipError_t e = hipMemsetAsync(plan->d_done, 0, nflags * sizeof(int), st);
    if (e != hipSuccess) return e;
    static const int slots = [] {
      int per_cu = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mlp_pipeline_x2, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
      return per_cu * (mlp_chain_slots() / 2);
    }();
    const int grid = (int)std::max<size_t>(8, std::min<size_t>((size_t)slots, nitems));   // every counter needs a taker
    hipLaunchKernelGGL(mlp_pipeline_x2, dim3(grid), dim3(256), 0, st, reinterpret_cast<const GemmArgs*>(d),
                       reinterpret_cast<const int*>(d + b0), reinterpret_cast<const int*>(d + b0 + b1), nlayers, nprob, total,
                       plan->d_done, err_flag);
    return hipGetLastError();
  }
  // no more tiles than CUs: one eight-wave workgroup per CU; otherwise two four-wave workgroups per CU
  const bool wide = forced_r == 8 || (forced_r != 4 && 2 * total <= mlp_chain_slots());
  auto go = [&](auto kernel, int threads) {
    hipLaunchKernelGGL(kernel, dim3(total), dim3(threads), 0, st, reinterpret_cast<const GemmArgs*>(d),
                       reinterpret_cast<const int*>(d + b0), reinterpret_cast<const int*>(d + b0 + b1), nlayers, nprob);
  };
  static const int kb_wide = [] { const char* e = getenv("ANI_CHAIN_KB"); return e ? atoi(e) : 2; }();   // experiment knob
  if (arith == MLP_F16X2) {
    if (wide && kb_wide == 2) go(mlp_chain_x3<2, 8, 2, 2>, 512);
    else if (wide) go(mlp_chain_x3<2, 8, 2, 1>, 512);
    else go(mlp_chain_x3<2, 4, 2, 1>, 256);
  } else {
    if (wide) go(mlp_chain_x3<2, 8, 3, 1>, 512); else go(mlp_chain_x3<2, 4, 3, 1>, 256);
  }
  return hipGetLastError();
}

// weights -> blocked planes: three bf16 (planes = 3) or two fp16 of the scaled weight (planes = 2)
__global__ void split_planes_kernel(const float* __restrict__ src, long long s_src, int N, int K, int ld, int kb, int planes,
                                    float scale, unsigned short* __restrict__ dst) {
  const long long per = (long long)N * kb * 16;   // (row, k) slots per matrix
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int bi = blockIdx.y;
  if (idx >= per) return;
  const int n = (int)(idx / (kb * 16)), k = (int)(idx % (kb * 16));
  const float x = k < K ? src[(long long)bi * s_src + (long long)n * ld + k] : 0.f;
  unsigned short* d = dst + ((long long)bi * per + ((long long)(k / 16) * N + n) * 16) * planes + (k % 16);
  if (planes == 3) {
    unsigned h, m, l;
    split3(x, h, m, l);
    d[0] = (unsigned short)(h >> 16);
    d[16] = (unsigned short)(m >> 16);
    d[32] = (unsigned short)(l >> 16);
  } else {
    unsigned h, l;
    split2(x * scale, 0.f, h, l);
    d[0] = (unsigned short)(h & 0xffffu);
    d[16] = (unsigned short)(l & 0xffffu);
  }
}

void launch_split_planes(const float* src, int batch, long long s_src, int N, int K, int ld, MlpArith arith, float scale,
                         unsigned short* dst, hipStream_t st) {
  const int kb = (K + 15) / 16;
  const long long per = (long long)N * kb * 16;
  if (per <= 0 || batch <= 0) return;
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((per + 255) / 256), batch), dim3(256), 0, st, src, s_src, N, K, ld, kb,
                     mlp_planes(arith), scale, dst);
}

template <int WM, int KB, int RT, int P>
static void launch_wm_x3(const GroupArgs& G, Epilogue epi, int total, hipStream_t st) {
  const dim3 grid(total), block(256);
  switch (epi) {
    case EPI_PLAIN: hipLaunchKernelGGL((gemm_grouped_x3<WM, EPI_PLAIN, KB, RT, P>), grid, block, 0, st, G); break;
    case EPI_CELU: hipLaunchKernelGGL((gemm_grouped_x3<WM, EPI_CELU, KB, RT, P>), grid, block, 0, st, G); break;
    case EPI_LAST: hipLaunchKernelGGL((gemm_grouped_x3<WM, EPI_LAST, KB, RT, P>), grid, block, 0, st, G); break;
    case EPI_BWD: hipLaunchKernelGGL((gemm_grouped_x3<WM, EPI_BWD, KB, RT, P>), grid, block, 0, st, G); break;
  }
}

template <int WM>
static void launch_wm(const GroupArgs& G, Epilogue epi, int total, hipStream_t st) {
  const dim3 grid(total), block(256);
  switch (epi) {
    case EPI_PLAIN: hipLaunchKernelGGL((gemm_grouped<WM, EPI_PLAIN>), grid, block, 0, st, G); break;
    case EPI_CELU: hipLaunchKernelGGL((gemm_grouped<WM, EPI_CELU>), grid, block, 0, st, G); break;
    case EPI_LAST: hipLaunchKernelGGL((gemm_grouped<WM, EPI_LAST>), grid, block, 0, st, G); break;
    case EPI_BWD: hipLaunchKernelGGL((gemm_grouped<WM, EPI_BWD>), grid, block, 0, st, G); break;
  }
}

// estimated makespan (in units of one CU doing one row x one column-tile x K work) of a tile height R
static double makespan(const GemmArgs* probs, int nprob, int R, int ncu) {
  double total = 0, biggest = 0;
  for (int i = 0; i < nprob; i++) {
    const GemmArgs& g = probs[i];
    if (g.rows <= 0 || g.N <= 0) continue;
    const int tiles_m = (g.rows + R - 1) / R;
    const int nblocks = (g.N + 255) / 256;
    const double per_tile = (double)R * ((g.N + nblocks - 1) / nblocks) * g.K;
    total += per_tile * tiles_m * nblocks * g.batch;
    biggest = per_tile > biggest ? per_tile : biggest;
  }
  // greedy dispatch: every CU is busy until the work runs out, then at most one more tile.  Shorter tiles re-read the
  // Bt slabs more often and have fewer MFMAs per staged byte: charged as a throughput factor (measured ballpark).
  const double eff = R >= 128 ? 1.0 : (R >= 64 ? 0.98 : 0.75);
  return total / (ncu * eff) + biggest;
}

void launch_gemm_group(const GemmArgs* probs, int nprob, Epilogue epi, hipStream_t st, MlpArith arith) {
  const bool split_bf16 = arith != MLP_FP32;
  static const int forced = [] { const char* e = getenv("ANI_GEMM_WM"); return e ? atoi(e) : 0; }();
  static const int ncu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  for (int base = 0; base < nprob; base += kMaxProblems) {
    const int np = nprob - base < kMaxProblems ? nprob - base : kMaxProblems;
    int WM = forced;
    if (WM != 1 && WM != 2 && WM != 4) {
      const double m32 = makespan(probs + base, np, 32, ncu), m64 = makespan(probs + base, np, 64, ncu),
                   m128 = makespan(probs + base, np, 128, ncu);
      WM = (m32 < m64 && m32 < m128) ? 1 : (m64 < m128 ? 2 : 4);
      // the split-bf16 kernel is bound by its staging skeleton, not by MFMA time: 64-row tiles (3 workgroups per CU) won
      // at every size measured (12.5k .. 100k atoms per GPU), the makespan model above was fitted to the fp32-input kernel
      if (split_bf16) WM = 2;
    }
    const int R = 32 * WM;
    GroupArgs G;
    G.nprob = 0;
    int total = 0;
    for (int i = 0; i < np; i++) {
      const GemmArgs& g = probs[base + i];
      if (g.rows <= 0 || g.N <= 0 || g.batch <= 0) continue;
      const int tiles_m = g.rows / R;  // rows is a multiple of kRowTile = 128
      const int nblocks = (g.N + 255) / 256;
      G.p[G.nprob] = g;
      G.tiles_m[G.nprob] = tiles_m;
      G.tile_start[G.nprob] = total;
      total += tiles_m * g.batch * nblocks;
      G.nprob++;
    }
    G.tile_start[G.nprob] = total;
    if (total == 0) continue;
    bool x3 = split_bf16;
    for (int i = 0; i < G.nprob; i++) x3 = x3 && G.p[i].Btp != nullptr;
    if (x3) {
      // 128-row tiles: 2x2 waves with two row strips each (RT = 2), not four waves stacked in M
      if (arith == MLP_F16X2) {
        if (WM == 4) launch_wm_x3<2, 1, 2, 2>(G, epi, total, st);
        else if (WM == 2) launch_wm_x3<2, 1, 1, 2>(G, epi, total, st);
        else launch_wm_x3<1, 1, 1, 2>(G, epi, total, st);
      } else {
        if (WM == 4) launch_wm_x3<2, 1, 2, 3>(G, epi, total, st);
        else if (WM == 2) launch_wm_x3<2, 1, 1, 3>(G, epi, total, st);
        else launch_wm_x3<1, 1, 1, 3>(G, epi, total, st);
      }
    } else {
      if (WM == 4) launch_wm<4>(G, epi, total, st);
      else if (WM == 2) launch_wm<2>(G, epi, total, st);
      else launch_wm<1>(G, epi, total, st);
    }
  }
}

void launch_gemm(const GemmArgs& g, Epilogue epi, hipStream_t st) { launch_gemm_group(&g, 1, epi, st, MLP_FP32); }

}  // namespace ani
