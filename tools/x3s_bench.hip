// x3s_bench.hip — prototype of a Bt-stationary split-bf16 GEMM (development tool, not shipped):
// every wave keeps the hi/mid/lo planes of ITS 32 columns of Bt for the whole K in registers and streams 32-row strips
// of A through LDS; no Bt traffic through LDS, one barrier per strip instead of two per 16-deep slab.
//   hipcc -O3 --offload-arch=gfx950 -o x3s_bench x3s_bench.hip && ./x3s_bench [rows N K]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
  const unsigned xb = __float_as_uint(x);
  h = xb & 0xffff0000u;
  const float r1 = x - __uint_as_float(h);
  m = __float_as_uint(r1) & 0xffff0000u;
  l = __float_as_uint(r1 - __uint_as_float(m));
}
__device__ __forceinline__ unsigned pack_hi16(unsigned a, unsigned b) { return (a >> 16) | (b & 0xffff0000u); }

template <int KBMAX>
__global__ __launch_bounds__(512, 1) void x3s(const float* __restrict__ A, int lda, const uint4* __restrict__ B3, float* __restrict__ C,
                                             int ldc, int rows, int N, int K) {
  extern __shared__ uint4 lds4[];
  const int nkb = (K + 15) >> 4;
  const int rowa = nkb * 96 + 16;  // bytes per staged A row
  unsigned char* As = reinterpret_cast<unsigned char*>(lds4);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
  const int ntiles = (N + 31) >> 5;
  const bool active = wave < ntiles;
  const int col = 32 * wave + lr;

  // Bt of this wave's 32 columns, whole K, three planes: registers
  uint4 bq[KBMAX][3];
#pragma unroll
  for (int kb = 0; kb < KBMAX; kb++)
#pragma unroll
    for (int p = 0; p < 3; p++)
      bq[kb][p] = (active && kb < nkb && col < N) ? B3[((long long)kb * N + col) * 6 + p * 2 + lh] : make_uint4(0, 0, 0, 0);

  const int nstrips = rows >> 5;
  const int k4 = K >> 2;              // float4 per A row
  const int per = (32 * k4 + 511) / 512;  // float4 per thread and strip (<= 4 for K <= 256)
  float4 pa[4];
  auto gload = [&](int strip) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int idx = tid + 512 * i;
      const int r = idx / k4, c4 = idx - r * k4;
      pa[i] = (i < per && r < 32 && strip < nstrips) ? *reinterpret_cast<const float4*>(A + (long long)(32 * strip + r) * lda + 4 * c4)
                                                      : make_float4(0, 0, 0, 0);
    }
  };
  auto stage = [&](int buf) {
    unsigned char* dstb = As + buf * 32 * rowa;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int idx = tid + 512 * i;
      const int r = idx / k4, c4 = idx - r * k4;
      if (i < per && r < 32) {
        unsigned h[4], m[4], l[4];
        split3(pa[i].x, h[0], m[0], l[0]); split3(pa[i].y, h[1], m[1], l[1]);
        split3(pa[i].z, h[2], m[2], l[2]); split3(pa[i].w, h[3], m[3], l[3]);
        const int k = 4 * c4;
        unsigned char* d = dstb + r * rowa + (k >> 4) * 96 + (k & 15) * 2;
        *reinterpret_cast<uint2*>(d) = make_uint2(pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3]));
        *reinterpret_cast<uint2*>(d + 32) = make_uint2(pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3]));
        *reinterpret_cast<uint2*>(d + 64) = make_uint2(pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3]));
      }
    }
  };

  int strip = blockIdx.x;
  gload(strip);
  stage(0);
  __syncthreads();
  int buf = 0;
  for (; strip < nstrips; strip += gridDim.x) {
    gload(strip + gridDim.x);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    if (active) {
      const unsigned char* ap = As + buf * 32 * rowa + lr * rowa + lh * 16;
#pragma unroll
      for (int kb = 0; kb < KBMAX; kb++) {
        if (kb < nkb) {
          const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ap + kb * 96));
          const bf16x8 am = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ap + kb * 96 + 32));
          const bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ap + kb * 96 + 64));
          const bf16x8 bh = __builtin_bit_cast(bf16x8, bq[kb][0]);
          const bf16x8 bm = __builtin_bit_cast(bf16x8, bq[kb][1]);
          const bf16x8 bl = __builtin_bit_cast(bf16x8, bq[kb][2]);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        }
      }
      if (col < N) {
        const int mbase = 32 * strip + 4 * lh;
#pragma unroll
        for (int r = 0; r < 16; r++) C[(long long)(mbase + (r & 3) + 8 * (r >> 2)) * ldc + col] = acc[r];
      }
    }
    stage(buf ^ 1);
    __syncthreads();
    buf ^= 1;
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
  std::vector<unsigned short> hB3((size_t)nkb * N * 48, 0);
  for (int n = 0; n < N; n++)
    for (int k = 0; k < K; k++) {
      const float x = hB[(size_t)n * K + k];
      unsigned xb; memcpy(&xb, &x, 4);
      unsigned h = xb & 0xffff0000u; float hf; memcpy(&hf, &h, 4);
      float r1 = x - hf; unsigned mb; memcpy(&mb, &r1, 4); mb &= 0xffff0000u; float mf; memcpy(&mf, &mb, 4);
      float r2 = r1 - mf; unsigned lb; memcpy(&lb, &r2, 4);
      unsigned short* d = &hB3[((size_t)(k / 16) * N + n) * 48 + (k % 16)];
      d[0] = h >> 16; d[16] = mb >> 16; d[32] = lb >> 16;
    }
  float *A, *C; uint4* B3;
  hipMalloc(&A, hA.size() * 4); hipMalloc(&C, (size_t)rows * N * 4); hipMalloc(&B3, hB3.size() * 2);
  hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(B3, hB3.data(), hB3.size() * 2, hipMemcpyHostToDevice);
  const size_t lds = 2 * 32 * (size_t)(nkb * 96 + 16);
  hipFuncSetAttribute((const void*)x3s<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256;
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL((x3s<16>), dim3(grid), dim3(512), lds, 0, A, K, B3, C, N, rows, N, K);
  hipDeviceSynchronize();
  printf("launch: %s\n", hipGetErrorString(hipGetLastError()));
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL((x3s<16>), dim3(grid), dim3(512), lds, 0, A, K, B3, C, N, rows, N, K);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  std::vector<float> hC((size_t)rows * N);
  hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int t = 0; t < 256; t++) {
    const int m = (int)(((long long)t * 7919) % rows), n = (t * 104729) % N;
    double acc = 0;
    for (int k = 0; k < K; k++) acc += (double)hA[(size_t)m * K + k] * hB[(size_t)n * K + k];
    maxerr = fmax(maxerr, fabs(acc - hC[(size_t)m * N + n]));
  }
  const double fl = 2.0 * rows * N * K;
  printf("rows=%d N=%d K=%d  %.3f ms  %.1f TFLOP/s algorithmic  maxerr=%.2e  (bytes A+C %.0f MB -> %.2f TB/s)\n", rows, N, K, ms,
         fl / ms / 1e9, maxerr, (rows * (double)(K + N) * 4) / 1e6, rows * (double)(K + N) * 4 / ms / 1e9);
  return 0;
}
