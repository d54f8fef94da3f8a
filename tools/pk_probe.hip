// pk_probe.hip -- issue rate of v_pk_fma_f32 / v_pk_mul_f32 against v_fma_f32 on gfx950 (one number decides whether packing the
// AEV kernels' fp32 arithmetic can pay).  Build: hipcc --offload-arch=gfx950 -O3 -o pk_probe pk_probe.hip ; run: ./pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP 64
template <int MODE>
__global__ __launch_bounds__(1024) void probe(float* out, unsigned long long* cyc, float seed) {
  v2f a[8], b[8], c[8];
  for (int i = 0; i < 8; i++) { a[i] = v2f{seed + i, seed - i}; b[i] = v2f{1.0f + 1e-7f * i, 1.0f - 1e-7f * i}; c[i] = v2f{1e-9f * threadIdx.x, 2e-9f}; }
  unsigned long long t0, t1;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int r = 0; r < REP; r++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (MODE == 0) {   // 16 scalar fma: three distinct VGPR sources
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(b[i].x), "v"(c[i].x));
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].y) : "v"(b[i].y), "v"(c[i].y));
        } else if (MODE == 1) {   // 8 packed fma, three distinct VGPR pairs
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
        } else if (MODE == 2) {   // packed fma, two sources the same pair
          asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a[i]) : "v"(b[i]));
        } else if (MODE == 3) {   // packed mul
          asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
        } else if (MODE == 4) {   // scalar mul x2
          asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[i].x) : "v"(b[i].x), "v"(c[i].x));
          asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[i].y) : "v"(b[i].y), "v"(c[i].y));
        } else if (MODE == 5) {   // packed fma with a broadcast source (op_sel_hi 0 on src0)
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
        } else if (MODE == 6) {   // packed add
          asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
        }
      }
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
  for (int i = 0; i < 8; i++) s += a[i].x + a[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) atomicMax(cyc, t1 - t0);   // the slowest wave: the issue arbiter favours the oldest one
}
template <int MODE>
static void run(const char* name, int threads, int ninstr_per_iter, int blocks = 1) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, (size_t)4 * 1024 * 1024); hipMalloc(&cyc, 8);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0f);
  hipDeviceSynchronize(); hipMemset(cyc, 0, 8);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0f);
  unsigned long long h; if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { printf("launch failed\n"); return; } hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double n = (double)REP * 4 * ninstr_per_iter;
  // s_memtime counts at 100 MHz; print raw ticks per instruction per wave-on-a-SIMD as a RATIO between modes
  printf("%-44s threads %4d  ticks %8llu  ticks/instr %.4f  waves/SIMD %d  cycles per wave-instruction and SIMD %.3f\n", name, threads, h, h / n, blocks > 1 ? 8 : threads / 256, h / n / (blocks > 1 ? 8 : threads / 256));
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int threads : {256, 512, 1024}) {
    run<0>("v_fma_f32 x16 (per 8 pairs)", threads, 16);
    run<1>("v_pk_fma_f32 x8, distinct sources", threads, 8);
    run<2>("v_pk_fma_f32 x8, src0 == src1", threads, 8);
    run<5>("v_pk_fma_f32 x8, broadcast src0", threads, 8);
    run<4>("v_mul_f32 x16", threads, 16);
    run<3>("v_pk_mul_f32 x8", threads, 8);
    run<6>("v_pk_add_f32 x8", threads, 8);
  }
  // two workgroups of 16 waves per CU: 8 waves per SIMD (the slowest wave of the whole grid)
  run<0>("v_fma_f32 x16, 512 workgroups", 1024, 16, 512);
  run<1>("v_pk_fma_f32 x8, 512 workgroups", 1024, 8, 512);
  run<4>("v_mul_f32 x16, 512 workgroups", 1024, 16, 512);
  run<3>("v_pk_mul_f32 x8, 512 workgroups", 1024, 8, 512);
  return 0;
}
