// mfma_bf16_probe.hip — issue interval of v_mfma_f32_32x32x16_bf16 in the patterns the fused MLP kernel uses, one wave per
// SIMD (development probe, not shipped).  Prints shader cycles per MFMA (s_memtime) for: one dependent chain on one
// accumulator; the kernel's pattern, six in a row per accumulator over NA accumulators; the same with independent VALU
// work between the MFMAs; the same with the A fragments read from LDS one block ahead.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NA>
__global__ __launch_bounds__(256, 1) void probe(float* out, unsigned long long* cyc, int iters) {
  extern __shared__ uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += 256) lds[i] = make_uint4(0x3f803f80u + i, 0x3f003f00u, 0x3e803e80u, 0x3f803f80u);
  __syncthreads();
  f32x16 acc[NA];
  for (int n = 0; n < NA; n++) for (int r = 0; r < 16; r++) acc[n][r] = 0.f;
  u32x4 a0 = {0x3f803f80u + (unsigned)lane, 0x3f003f00u, 0x3e803e80u, 0x3f803f80u}, b0 = {0x3e003e00u, 0x3f803f80u + (unsigned)lane, 0x3f003f00u, 0x3e803e80u};
  float v0 = lane * 0.001f, v1 = 1.0f, v2 = 0.5f, v3 = 0.25f;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  const unsigned char* base = reinterpret_cast<const unsigned char*>(lds) + lane * 16;
  u32x4 fa[2][3];
  if (MODE == 3) for (int p = 0; p < 3; p++) fa[0][p] = *reinterpret_cast<const u32x4*>(base + p * 1024);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int n = 0; n < NA; n++) {
      u32x4 a[3] = {a0, a0, a0};
      if (MODE == 3) {
        if (true) for (int p = 0; p < 3; p++) fa[(n + 1) & 1][p] = *reinterpret_cast<const u32x4*>(base + ((n + 1) * 3 + p) * 1024);
        for (int p = 0; p < 3; p++) a[p] = fa[n & 1][p];
      }
#pragma unroll
      for (int k = 0; k < 6; k++) {
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[k % 3]), __builtin_bit_cast(bf16x8, b0), acc[n], 0, 0, 0);
        if (MODE == 2) {   // four independent VALU per MFMA
          v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 0.9999f, 0.25f); v2 = fmaf(v2, 1.0002f, 0.125f); v3 = fmaf(v3, 0.9998f, 0.0625f);
        }
      }
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = v0 + v1 + v2 + v3;
  for (int n = 0; n < NA; n++) for (int r = 0; r < 16; r++) s += acc[n][r];
  out[blockIdx.x * 256 + tid] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int MODE, int NA>
void run(const char* name, float* out, unsigned long long* cyc, int blocks) {
  const int iters = 200;
  (void)hipFuncSetAttribute((const void*)probe<MODE, NA>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<MODE, NA>), dim3(blocks), dim3(256), 100 * 1024, 0, out, cyc, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((probe<MODE, NA>), dim3(blocks), dim3(256), 100 * 1024, 0, out, cyc, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[4];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double n = (double)iters * NA * 6;
  printf("%-58s blocks=%4d  %7.1f cycles/MFMA (wave 0)  wall %.3f ms = %.1f ns/MFMA\n", name, blocks, h[0] / n, ms, ms * 1e6 / n);
}

int main() {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, sizeof(float) * 256 * 256);
  (void)hipMalloc(&cyc, sizeof(unsigned long long) * 256 * 4);
  for (int blocks : {1, 256}) {
    run<0, 1>("one accumulator, dependent chain", out, cyc, blocks);
    run<0, 2>("six in a row per accumulator, 2 accumulators", out, cyc, blocks);
    run<0, 6>("six in a row per accumulator, 6 accumulators", out, cyc, blocks);
    run<2, 6>("  + four independent v_fma between the MFMAs", out, cyc, blocks);
    run<3, 6>("  + A fragments from LDS, one block ahead", out, cyc, blocks);
  }
  return 0;
}
