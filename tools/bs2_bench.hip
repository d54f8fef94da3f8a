// bs2_bench.hip — bs_bench.hip with the two-term fp16 split (three MFMA products) instead of the three-term bf16 one:
// a workgroup keeps the hi/mid/lo planes of 64 columns of Bt for the whole K in LDS for its lifetime; its 8 waves walk
// 32-row strips of A independently: A goes global -> registers in MFMA operand layout (32 B per lane and k-block), is
// split in registers, and meets Bt fragments read from LDS.  No barrier after the Bt fill, no A staging.
//   hipcc -O3 --offload-arch=gfx950 -o bs_bench bs_bench.hip && ./bs_bench [rows N K]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int ROW = 80;  // bytes per (k-block, column) row in LDS: 2 planes x 32 B + pad

__device__ __forceinline__ void split8(const float4 lo4, const float4 hi4, uint4& h, uint4& l) {
  const float x[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
  unsigned hh[4], ll[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const f16x2 hv = {(_Float16)x[2 * i], (_Float16)x[2 * i + 1]};
    const f16x2 lv = {(_Float16)(x[2 * i] - (float)hv[0]), (_Float16)(x[2 * i + 1] - (float)hv[1])};
    hh[i] = __builtin_bit_cast(unsigned, hv); ll[i] = __builtin_bit_cast(unsigned, lv);
  }
  h = make_uint4(hh[0], hh[1], hh[2], hh[3]);
  l = make_uint4(ll[0], ll[1], ll[2], ll[3]);
}

// TR = 1: activations feature-major in and out (At[k][rows], Ct[n][rows]): every fragment load is a dword per lane with
// the 32 lanes of a half-wave on 128 contiguous bytes, and so is every store (the weights are the MFMA's A operand)
template <int D, int TR, int ABL = 0>
__global__ __launch_bounds__(512, 1) void bs(const float* __restrict__ A, int lda, const uint4* __restrict__ B3, float* __restrict__ C,
                                            int ldc, int rows, int N, int K, int ncb) {
  extern __shared__ uint4 lds4[];
  unsigned char* Bs = reinterpret_cast<unsigned char*>(lds4);
  const int nkb = (K + 15) >> 4;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
  // XCD-aware: workgroups are dealt round-robin to the 8 XCDs; the ncb workgroups that walk the same rows (one per
  // column block) sit on ONE XCD, so that its L2 serves the A re-reads
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per_xcd = gridDim.x >> 3, gx = per_xcd / ncb;
  if (j >= gx * ncb) return;
  const int cb = j % ncb, g = (j / ncb) * 8 + xcd, G = gx * 8;
  const int ncols = min(64, N - 64 * cb);
  // Bt block -> LDS, once
  for (int c = tid; c < nkb * 64 * 4; c += 512) {
    const int q = c % 4, col = (c / 4) & 63, kb = c / (4 * 64);
    const uint4 v = col < ncols ? B3[((long long)kb * N + 64 * cb + col) * 4 + q] : make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(Bs + (kb * 64 + col) * ROW + q * 16) = v;
  }
  __syncthreads();

  const int ntiles = rows >> 5;
  const int tstep = G * 8;
  int ct = g * 8 + wave;               // tile being multiplied
  int lt = ct, lkb = 0;                // next (tile, k-block) to request
  const int mine = ct < ntiles ? (ntiles - ct + tstep - 1) / tstep : 0;
  const int total = mine * nkb;
  float4 pa[D][2];
  auto issue = [&](int slot) {
    if (lt < ntiles && !((ABL & 1) && lt != g * 8 + wave)) {
      if constexpr (TR) {
        const float* p = A + (long long)(lkb * 16 + lh * 8) * lda + 32 * lt + lr;   // lda = padded row count
        pa[slot][0] = make_float4(p[0], p[lda], p[2 * (long long)lda], p[3 * (long long)lda]);
        pa[slot][1] = make_float4(p[4 * (long long)lda], p[5 * (long long)lda], p[6 * (long long)lda], p[7 * (long long)lda]);
      } else {
        const float* p = A + (long long)(32 * lt + lr) * lda + lkb * 16 + lh * 8;
        pa[slot][0] = *reinterpret_cast<const float4*>(p);
        pa[slot][1] = *reinterpret_cast<const float4*>(p + 4);
      }
    }
    if (++lkb == nkb) { lkb = 0; lt += tstep; }
  };
#pragma unroll
  for (int d = 0; d < D; d++) issue(d);
  f32x16 acc[2];
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
  int ckb = 0;
  const int ntc = (ncols + 31) >> 5;
  for (int i = 0; i < total; i += D) {
#pragma unroll
    for (int u = 0; u < D; u++) {
      if (i + u < total) {
        uint4 ah, al;
        if constexpr (ABL & 2) { ah = __builtin_bit_cast(uint4, pa[u][0]); al = __builtin_bit_cast(uint4, pa[u][1]); }
        else split8(pa[u][0], pa[u][1], ah, al);
        issue(u);
        const f16x8 Ah = __builtin_bit_cast(f16x8, ah), Al = __builtin_bit_cast(f16x8, al);
#pragma unroll
        for (int c = 0; c < 2; c++) {
          if (c < ntc) {
            const unsigned char* bp = Bs + (((ABL & 4) ? 0 : ckb) * 64 + 32 * c + lr) * ROW + lh * 16;
            const f16x8 bh = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(bp));
            const f16x8 bl = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(bp + 32));
            if constexpr (TR) {
              acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, Al, acc[c], 0, 0, 0);
              acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, Ah, acc[c], 0, 0, 0);
              acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, Ah, acc[c], 0, 0, 0);
            } else {
              acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, bh, acc[c], 0, 0, 0);
              acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, bl, acc[c], 0, 0, 0);
              acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, bh, acc[c], 0, 0, 0);
            }
          }
        }
        if (++ckb == nkb) {
          const int mbase = 32 * ct + 4 * lh;
#pragma unroll
          for (int c = 0; c < 2; c++) {
            const int col = 64 * cb + 32 * c + lr;
            if constexpr (TR) {
              if (c < ntc) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                  const int n = 64 * cb + 32 * c + 4 * lh + (r & 3) + 8 * (r >> 2);
                  if (n < N) C[(long long)n * ldc + 32 * ct + lr] = acc[c][r];   // ldc = padded row count
                }
              }
            } else if (c < ntc && col < N) {
#pragma unroll
              for (int r = 0; r < 16; r++) C[(long long)(mbase + (r & 3) + 8 * (r >> 2)) * ldc + col] = acc[c][r];
            }
#pragma unroll
            for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
          }
          ckb = 0;
          ct += tstep;
        }
      }
    }
  }
}

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 66688;
  const int N = argc > 2 ? atoi(argv[2]) : 256;
  const int K = argc > 3 ? atoi(argv[3]) : 256;
  const int reps = 20, nkb = (K + 15) / 16;
  std::vector<float> hA((size_t)rows * K), hB((size_t)N * K);
  for (auto& v : hA) v = (float)rand() / RAND_MAX - 0.5f;
  for (auto& v : hB) v = (float)rand() / RAND_MAX - 0.5f;
  std::vector<unsigned short> hB3((size_t)nkb * N * 32, 0);
  for (int n = 0; n < N; n++)
    for (int k = 0; k < K; k++) {
      const float x = hB[(size_t)n * K + k];
      const _Float16 h = (_Float16)x, l = (_Float16)(x - (float)h);
      unsigned short* d = &hB3[((size_t)(k / 16) * N + n) * 32 + (k % 16)];
      memcpy(&d[0], &h, 2); memcpy(&d[16], &l, 2);
    }
  float *A, *C; uint4* B3;
  hipMalloc(&A, hA.size() * 4); hipMalloc(&C, (size_t)rows * N * 4); hipMalloc(&B3, hB3.size() * 2);
  hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(B3, hB3.data(), hB3.size() * 2, hipMemcpyHostToDevice);
  const size_t lds = (size_t)nkb * 64 * ROW;
  hipFuncSetAttribute((const void*)bs<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)bs<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)bs<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  std::vector<float> hAt((size_t)rows * K);
  for (int m = 0; m < rows; m++) for (int k = 0; k < K; k++) hAt[(size_t)k * rows + m] = hA[(size_t)m * K + k];
  float* At; hipMalloc(&At, hAt.size() * 4); hipMemcpy(At, hAt.data(), hAt.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int ncb = (N + 63) / 64;
  const int grid = 256;
  hipFuncSetAttribute((const void*)bs<2, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)bs<2, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)bs<2, 0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)bs<2, 0, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int variant = 0; variant < 7; variant++) {
    auto launch = [&]() {
      if (variant == 0) hipLaunchKernelGGL((bs<2, 0>), dim3(grid), dim3(512), lds, 0, A, K, B3, C, N, rows, N, K, ncb);
      else if (variant == 1) hipLaunchKernelGGL((bs<2, 1>), dim3(grid), dim3(512), lds, 0, At, rows, B3, C, rows, rows, N, K, ncb);
      else if (variant == 2) hipLaunchKernelGGL((bs<4, 1>), dim3(grid), dim3(512), lds, 0, At, rows, B3, C, rows, rows, N, K, ncb);
      else if (variant == 3) hipLaunchKernelGGL((bs<2, 0, 1>), dim3(grid), dim3(512), lds, 0, A, K, B3, C, N, rows, N, K, ncb);
      else if (variant == 4) hipLaunchKernelGGL((bs<2, 0, 2>), dim3(grid), dim3(512), lds, 0, A, K, B3, C, N, rows, N, K, ncb);
      else if (variant == 5) hipLaunchKernelGGL((bs<2, 0, 4>), dim3(grid), dim3(512), lds, 0, A, K, B3, C, N, rows, N, K, ncb);
      else hipLaunchKernelGGL((bs<2, 0, 7>), dim3(grid), dim3(512), lds, 0, A, K, B3, C, N, rows, N, K, ncb);
    };
    hipMemset(C, 0, (size_t)rows * N * 4);
    for (int i = 0; i < 3; i++) launch();
    hipDeviceSynchronize();
    printf("launch: %s\n", hipGetErrorString(hipGetLastError()));
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; i++) launch();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    std::vector<float> hC((size_t)rows * N);
    hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int t = 0; t < 512; t++) {
      const int m = (int)(((long long)t * 7919) % rows), n = (t * 104729) % N;
      double acc = 0;
      for (int k = 0; k < K; k++) acc += (double)hA[(size_t)m * K + k] * hB[(size_t)n * K + k];
      maxerr = fmax(maxerr, fabs(acc - ((variant == 0 || variant >= 3) ? hC[(size_t)m * N + n] : hC[(size_t)n * rows + m])));
    }
    const double fl = 2.0 * rows * N * K;
    printf("%s rows=%d N=%d K=%d  %.3f ms  %.1f TFLOP/s algorithmic  maxerr=%.2e  (A x ncb + C: %.0f MB -> %.2f TB/s)\n", variant == 0 ? "row-major D=2" : variant == 1 ? "feat-major D=2" : variant == 2 ? "feat-major D=4" : variant == 3 ? "ABL no A loads" : variant == 4 ? "ABL no split" : variant == 5 ? "ABL B frag fixed" : variant == 6 ? "ABL all three" : "dynamic strip queue",
           rows, N, K, ms, fl / ms / 1e9, maxerr, (rows * (double)(K * ncb + N) * 4) / 1e6, rows * (double)(K * ncb + N) * 4 / ms / 1e9);
  }
  return 0;
}
