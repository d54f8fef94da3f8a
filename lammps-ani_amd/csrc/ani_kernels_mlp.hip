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

#include "ani_kernels.h"

namespace ani {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int LDS_LD = 36;       // floats per staged row (32 + 4 pad)
constexpr int kMaxProblems = 16; // species buckets per grouped launch

struct GroupArgs {
  GemmArgs p[kMaxProblems];
  int tile_start[kMaxProblems + 1];  // prefix of workgroups per problem
  int tiles_m[kMaxProblems];
  int nprob;
};

__device__ __forceinline__ float celu_f(float z, float alpha, float inv_alpha) {
  return z > 0.f ? z : alpha * expm1f(z * inv_alpha);
}
__device__ __forceinline__ float dcelu_from_h(float h, float inv_alpha) {
  // celu'(z) = 1 (z>0) or exp(z/alpha) = h/alpha + 1
  return h > 0.f ? 1.f : fmaf(h, inv_alpha, 1.f);
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
  float4 pa[PA], pb[8];
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
    for (int i = 0; i < PA; i++) *reinterpret_cast<float4*>(As + (cr + 32 * i) * LDS_LD + ck) = pa[i];
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

  // ---- epilogue -------------------------------------------------------------------------------------
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
          C[(long long)m * g.ldc + n] = acc[nt][r];
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
          C[(long long)m * g.ldc + n] = celu_f(acc[nt][r] + bv, g.alpha, g.inv_alpha);
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
          C[(long long)m * g.ldc + n] = acc[nt][r] * dcelu_from_h(h, g.inv_alpha);
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
          C[(long long)m * g.ldc + n] = valid[r] * wv * dcelu_from_h(h, g.inv_alpha);
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

void launch_gemm_group(const GemmArgs* probs, int nprob, Epilogue epi, hipStream_t st) {
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
    if (WM == 4) launch_wm<4>(G, epi, total, st);
    else if (WM == 2) launch_wm<2>(G, epi, total, st);
    else launch_wm<1>(G, epi, total, st);
  }
}

void launch_gemm(const GemmArgs& g, Epilogue epi, hipStream_t st) { launch_gemm_group(&g, 1, epi, st); }

}  // namespace ani
