// pk_probe.hip — VALU issue rate of v_fma_f32 vs v_pk_fma_f32 on gfx950 (development tool):
// many waves per SIMD, 16 independent accumulator chains per lane, no memory traffic in the loop.
//   hipcc -O3 --offload-arch=gfx950 -o pk_probe pk_probe.hip && ./pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_scalar(float* out, float a, float b, int iters) {
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = __builtin_fmaf(acc[i], a, b);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_packed(float* out, float a, float b, int iters) {
  f2 acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = (f2){threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f + i};
  const f2 av = {a, a}, bv = {b, b};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = __builtin_elementwise_fma(acc[i], av, bv);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s += acc[i].x + acc[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  float* out; hipMalloc(&out, 4 * 256 * 256 * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int blocks_per_cu : {1, 2, 4, 8}) {
    const int grid = 256 * blocks_per_cu;
    for (int v = 0; v < 2; v++) {
      for (int w = 0; w < 2; w++) {
        if (w) hipEventRecord(e0, 0);
        if (v == 0) hipLaunchKernelGGL(k_scalar, dim3(grid), dim3(256), 0, 0, out, 0.999f, 0.001f, iters);
        else hipLaunchKernelGGL(k_packed, dim3(grid), dim3(256), 0, 0, out, 0.999f, 0.001f, iters);
        if (w) { hipEventRecord(e1, 0); hipEventSynchronize(e1); }
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double fma = (double)grid * 256 * 16 * iters;   // scalar-equivalent FMAs
      printf("%s  %d waves/SIMD: %.3f ms  %.1f TFLOP/s  (%.2f cycles at 2.4 GHz per wave-instruction)\n", v ? "v_pk_fma_f32" : "v_fma_f32   ",
             blocks_per_cu, ms, 2 * fma / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)iters * (v ? 8 : 16) * blocks_per_cu));
    }
  }
  return 0;
}
