// mfma_gap_probe3.hip — which KINDS of instruction hide behind a v_mfma_f32_32x32x16_bf16 of the same wave (one wave per
// SIMD): the MFMA and four fillers of one kind in ONE asm statement (program order fixed).  Cycles per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(256, 1) void probe(float* out, unsigned long long* cyc, int iters) {
  extern __shared__ uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  f32x16 acc;
  float spare = 1.f;
  for (int r = 0; r < 16; r++) acc[r] = 0.f;
  u32x4 a0 = {0x3f803f80u + (unsigned)lane, 0x3f003f00u, 0x3e803e80u, 0x3f803f80u}, b0 = {0x3e003e00u, 0x3f803f80u + (unsigned)lane, 0x3f003f00u, 0x3e803e80u};
  float v0 = lane * 0.001f, v1 = 1.0f, v2 = 0.5f, v3 = 0.25f;
  unsigned ldsaddr = lane * 8;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 12; k++) {
#define ST(fill) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %6, %7, %0\n" fill : "+a"(acc), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+a"(spare) : "v"(a0), "v"(b0), "v"(ldsaddr) : "vcc");
      if (KIND == 0) { ST("") }
      if (KIND == 1) { ST("v_fma_f32 %1, %2, %3, %1\nv_fma_f32 %2, %1, %3, %2\nv_fma_f32 %3, %1, %2, %3\nv_fma_f32 %4, %1, %3, %4\n") }
      if (KIND == 2) { ST("v_accvgpr_read_b32 %1, %5\nv_accvgpr_read_b32 %2, %5\nv_accvgpr_read_b32 %3, %5\nv_accvgpr_read_b32 %4, %5\n") }
      if (KIND == 3) { ST("v_accvgpr_write_b32 %5, %1\nv_accvgpr_write_b32 %5, %2\nv_accvgpr_write_b32 %5, %3\nv_accvgpr_write_b32 %5, %4\n") }
      if (KIND == 4) { ST("v_exp_f32 %1, %1\nv_exp_f32 %2, %2\nv_mul_f32 %3, %3, %4\nv_mul_f32 %4, %4, %3\n") }
      if (KIND == 5) { ST("v_cmp_lt_f32 vcc, 0, %1\nv_cndmask_b32 %2, %3, %4, vcc\nv_cmp_lt_f32 vcc, 0, %3\nv_cndmask_b32 %4, %1, %2, vcc\n") }
      if (KIND == 6) { ST("v_and_b32 %1, 0xffff0000, %1\nv_and_b32 %2, 0xffff0000, %2\nv_sub_f32 %3, %3, %1\nv_sub_f32 %4, %4, %2\n") }
      if (KIND == 9) { ST("v_perm_b32 %1, %2, %3, %4\nv_perm_b32 %2, %1, %3, %4\nv_lshrrev_b32 %3, 16, %3\nv_and_or_b32 %4, %1, %2, %3\n") }
      if (KIND == 10) { ST("s_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\n") }
      if (KIND == 11) { ST("v_mov_b32 %1, %2\nv_mov_b32 %2, %3\nv_mov_b32 %3, %4\nv_mov_b32 %4, %1\n") }
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = v0 + v1 + v2 + v3;
  s += spare;
  for (int r = 0; r < 16; r++) s += acc[r];
  out[blockIdx.x * 256 + tid] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}
template <int KIND>
void run(const char* name, float* out, unsigned long long* cyc) {
  const int iters = 200, blocks = 256;
  (void)hipFuncSetAttribute((const void*)probe<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipLaunchKernelGGL((probe<KIND>), dim3(blocks), dim3(256), 100 * 1024, 0, out, cyc, iters);
  (void)hipDeviceSynchronize();
  unsigned long long h[4];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-52s %6.1f cycles/MFMA\n", name, h[0] / (double)(iters * 12));
}
int main() {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, sizeof(float) * 256 * 256);
  (void)hipMalloc(&cyc, sizeof(unsigned long long) * 256 * 4);
  run<0>("MFMA alone (accumulator in AGPRs)", out, cyc);
  run<1>("+ 4 v_fma_f32", out, cyc);
  run<2>("+ 4 v_accvgpr_read_b32 (other AGPRs)", out, cyc);
  run<3>("+ 4 v_accvgpr_write_b32 (other AGPRs)", out, cyc);
  run<4>("+ 2 v_exp_f32 + 2 v_mul_f32", out, cyc);
  run<5>("+ 2 (v_cmp + v_cndmask)", out, cyc);
  run<6>("+ 2 v_and (literal) + 2 v_sub", out, cyc);
  run<9>("+ 2 v_perm + v_lshrrev + v_and_or", out, cyc);
  run<10>("+ 4 s_nop 0", out, cyc);
  run<11>("+ 4 v_mov_b32", out, cyc);
  return 0;
}
