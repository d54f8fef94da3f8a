// mfma_gap_probe2.hip — like mfma_gap_probe.hip with the order fixed in ONE asm statement per MFMA: the MFMA, then N
// independent fillers (the compiler regrouped the separate statements of the first probe: MFMAs together, fillers
// together, which is exactly what it does to the fused MLP kernel).  Shader cycles per MFMA, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define F1 "v_fma_f32 %1, %2, %3, %1\n"
#define F2 F1 "v_fma_f32 %2, %1, %3, %2\n"
#define F3 F2 "v_fma_f32 %4, %1, %3, %4\n"
#define F4 F3 "v_fma_f32 %1, %4, %3, %1\n"
#define F5 F4 "v_fma_f32 %2, %4, %3, %2\n"
#define F6 F5 "v_fma_f32 %4, %2, %3, %4\n"
#define F8 F6 "v_fma_f32 %1, %2, %3, %1\nv_fma_f32 %2, %1, %3, %2\n"
#define E2 "v_exp_f32 %1, %1\nv_exp_f32 %2, %2\n" F3
template <int N>
__global__ __launch_bounds__(256, 1) void probe(float* out, unsigned long long* cyc, int iters) {
  extern __shared__ uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  f32x16 acc;
  for (int r = 0; r < 16; r++) acc[r] = 0.f;
  u32x4 a0 = {0x3f803f80u + (unsigned)lane, 0x3f003f00u, 0x3e803e80u, 0x3f803f80u}, b0 = {0x3e003e00u, 0x3f803f80u + (unsigned)lane, 0x3f003f00u, 0x3e803e80u};
  float v0 = lane * 0.001f, v1 = 1.0f, v2 = 0.5f, v3 = 0.25f;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 12; k++) {
#define ST(fill) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %5, %6, %0\n" fill : "+v"(acc), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a0), "v"(b0));
      if (N == 0) { ST("") }
      if (N == 1) { ST(F1) }
      if (N == 2) { ST(F2) }
      if (N == 3) { ST(F3) }
      if (N == 4) { ST(F4) }
      if (N == 5) { ST(F5) }
      if (N == 6) { ST(F6) }
      if (N == 8) { ST(F8) }
      if (N == 10) { ST(E2) }
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = v0 + v1 + v2 + v3;
  for (int r = 0; r < 16; r++) s += acc[r];
  out[blockIdx.x * 256 + tid] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}
template <int N>
void run(float* out, unsigned long long* cyc) {
  const int iters = 200, blocks = 256;
  (void)hipFuncSetAttribute((const void*)probe<N>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipLaunchKernelGGL((probe<N>), dim3(blocks), dim3(256), 100 * 1024, 0, out, cyc, iters);
  (void)hipDeviceSynchronize();
  unsigned long long h[4];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("MFMA + %2d fillers in program order: %6.1f cycles/MFMA%s\n", N == 10 ? 5 : N, h[0] / (double)(iters * 12), N == 10 ? "  (2 v_exp + 3 v_fma)" : "");
}
int main() {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, sizeof(float) * 256 * 256);
  (void)hipMalloc(&cyc, sizeof(unsigned long long) * 256 * 4);
  run<0>(out, cyc); run<1>(out, cyc); run<2>(out, cyc); run<3>(out, cyc); run<4>(out, cyc); run<5>(out, cyc); run<6>(out, cyc); run<8>(out, cyc); run<10>(out, cyc);
  return 0;
}
