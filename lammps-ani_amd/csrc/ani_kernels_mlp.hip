// ani_kernels_mlp.hip — species-bucketed MLP ensemble on fp32 MFMA (v_mfma_f32_32x32x2_f32, gfx950).
//
// Replaces BmmEnsemble + autograd through it (reference call sites models/lammps_ani.py:110,228-230,197-206).
// Every product here is a genuine dense contraction  C[rows][N] = A[rows][K] * Bt[N][K]^T :
//   forward  layer l : A = activations of layer l-1, Bt = W_l  ([out][in], torch.nn.Linear layout)
//   backward layer l : A = dE/dz_l,                  Bt = W_l^T ([in][out], transposed copy made at load time)
// so both operands are K-contiguous and one kernel serves all of them.
//
// Tiling (one workgroup = 4 waves = 128 rows x 32*NT columns):
//   wave w owns rows [32w, 32w+32) and ALL 32*NT columns: NT accumulators of 32x32 (16 VGPRs each);
//   K is walked in slabs of 32 staged through LDS (row stride 36 floats = 144 B = 9*16 B, so the
//   ds_read_b128 fragment reads are bank-conflict free); within a slab, lane l (r = l&31, h = l>>5) reads
//   4 consecutive k of row r at k = 8*ks + 4*h and feeds them to 4 successive MFMAs — the k order inside a
//   slab is permuted identically for A and B, which leaves the dot products unchanged.
//   The next slab's global loads are issued before the current slab's MFMAs (register prefetch).
// MFMA C layout (guide §3): col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#include "ani_kernels.h"

namespace ani {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int LDS_LD = 36;  // floats per staged row (32 + 4 pad)

__device__ __forceinline__ float celu_f(float z, float alpha, float inv_alpha) {
  return z > 0.f ? z : alpha * expm1f(z * inv_alpha);
}
__device__ __forceinline__ float dcelu_from_h(float h, float inv_alpha) {
  // celu'(z) = 1 (z>0) or exp(z/alpha) = h/alpha + 1
  return h > 0.f ? 1.f : fmaf(h, inv_alpha, 1.f);
}

template <int NT, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs g) {
  __shared__ float lds[(kRowTile + 32 * NT) * LDS_LD];
  float* As = lds;
  float* Bs = lds + kRowTile * LDS_LD;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.y;
  const int n0 = blockIdx.z * (32 * NT);
  const int row0 = g.row0 + blockIdx.x * kRowTile;
  const int K = g.K, N = g.N;

  const float* __restrict__ A = g.A + (long long)b * g.sA + (long long)row0 * g.lda;
  const float* __restrict__ Bt = g.Bt + (long long)b * g.sB + (long long)n0 * g.ldb;

  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; nt++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[nt][r] = 0.f;

  float4 pa[4], pb[NT];
  const int cr = tid >> 3;         // staged row handled by this thread (per 32-row group)
  const int ck = (tid & 7) * 4;    // k offset inside the slab

  auto gload = [&](int k0) {
    const int kc = k0 + ck;
    const bool kin = kc < K;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int r = cr + 32 * i;
      pa[i] = kin ? *reinterpret_cast<const float4*>(A + (long long)r * g.lda + kc) : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NT; i++) {
      const int r = cr + 32 * i;
      pb[i] = (kin && (n0 + r) < N) ? *reinterpret_cast<const float4*>(Bt + (long long)r * g.ldb + kc) : make_float4(0, 0, 0, 0);
    }
  };

  const int nkt = (K + 31) >> 5;
  gload(0);
  for (int kt = 0; kt < nkt; kt++) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) *reinterpret_cast<float4*>(As + (cr + 32 * i) * LDS_LD + ck) = pa[i];
#pragma unroll
    for (int i = 0; i < NT; i++) *reinterpret_cast<float4*>(Bs + (cr + 32 * i) * LDS_LD + ck) = pb[i];
    __syncthreads();
    if (kt + 1 < nkt) gload((kt + 1) << 5);
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      const float4 a4 = *reinterpret_cast<const float4*>(As + (32 * wave + lr) * LDS_LD + ks * 8 + 4 * lh);
#pragma unroll
      for (int nt = 0; nt < NT; nt++) {
        const float4 b4 = *reinterpret_cast<const float4*>(Bs + (32 * nt + lr) * LDS_LD + ks * 8 + 4 * lh);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc[nt], 0, 0, 0);
      }
    }
  }

  // ---- epilogue -------------------------------------------------------------------------------------
  float* __restrict__ C = g.C + (long long)b * g.sC;
  const int mbase = row0 + 32 * wave + 4 * lh;
  if constexpr (EPI == EPI_PLAIN) {
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      const int n = n0 + 32 * nt + lr;
      if (n < N) {
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
    for (int nt = 0; nt < NT; nt++) {
      const int n = n0 + 32 * nt + lr;
      if (n < N) {
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
    for (int nt = 0; nt < NT; nt++) {
      const int n = n0 + 32 * nt + lr;
      if (n < N) {
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
    float esum[16];
#pragma unroll
    for (int r = 0; r < 16; r++) esum[r] = 0.f;
    float valid[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int m = mbase + (r & 3) + 8 * (r >> 2);
      valid[r] = g.centre_of_row[m] >= 0 ? g.scale : 0.f;
    }
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      const int n = n0 + 32 * nt + lr;
      if (n < N) {
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
    const float bl = g.bias_last[b];
    float* e_out = g.e_out + (long long)b * g.sE;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      float v = esum[r];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 8);
      v += __shfl_xor(v, 4);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 1);
      if (lr == 0) {
        const int m = mbase + (r & 3) + 8 * (r >> 2);
        e_out[m] = valid[r] * (v + bl);
      }
    }
  }
}

template <int EPI>
static void launch_nt(const GemmArgs& g, int NT, dim3 grid, hipStream_t st) {
  switch (NT) {
    case 1: hipLaunchKernelGGL((gemm_kernel<1, EPI>), grid, dim3(256), 0, st, g); break;
    case 2: hipLaunchKernelGGL((gemm_kernel<2, EPI>), grid, dim3(256), 0, st, g); break;
    case 3: hipLaunchKernelGGL((gemm_kernel<3, EPI>), grid, dim3(256), 0, st, g); break;
    case 4: hipLaunchKernelGGL((gemm_kernel<4, EPI>), grid, dim3(256), 0, st, g); break;
    case 5: hipLaunchKernelGGL((gemm_kernel<5, EPI>), grid, dim3(256), 0, st, g); break;
    case 6: hipLaunchKernelGGL((gemm_kernel<6, EPI>), grid, dim3(256), 0, st, g); break;
    case 7: hipLaunchKernelGGL((gemm_kernel<7, EPI>), grid, dim3(256), 0, st, g); break;
    default: hipLaunchKernelGGL((gemm_kernel<8, EPI>), grid, dim3(256), 0, st, g); break;
  }
}

void launch_gemm(const GemmArgs& g, Epilogue epi, hipStream_t st) {
  if (g.rows <= 0 || g.N <= 0) return;
  const int ntiles = (g.N + 31) / 32;
  const int NT = ntiles > 8 ? 8 : ntiles;
  const int nblocks = (ntiles + NT - 1) / NT;
  dim3 grid(g.rows / kRowTile, g.batch, nblocks);
  switch (epi) {
    case EPI_PLAIN: launch_nt<EPI_PLAIN>(g, NT, grid, st); break;
    case EPI_CELU: launch_nt<EPI_CELU>(g, NT, grid, st); break;
    case EPI_LAST: launch_nt<EPI_LAST>(g, NT, grid, st); break;
    case EPI_BWD: launch_nt<EPI_BWD>(g, NT, grid, st); break;
  }
}

}  // namespace ani
