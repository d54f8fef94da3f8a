// issue_probe.hip -- what a gfx950 SIMD's issue time goes to when 1 / 2 / 4 waves share it: cost, in s_memtime ticks per
// wave-iteration and SIMD, of adding matrix (4x4x1 fp32), transcendental, DPP, swap and LDS instructions to a block of
// sixteen independent v_fma_f32.  The AEV kernels run at 4-5 waves per SIMD with exactly this mix.
// Build: hipcc --offload-arch=gfx950 -O3 -o issue_probe issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define REP 64
#define FMA4(o) \
  asm volatile("v_fma_f32 %0, %4, %5, %0\nv_fma_f32 %1, %4, %5, %1\nv_fma_f32 %2, %4, %5, %2\nv_fma_f32 %3, %4, %5, %3" \
               : "+v"(a[o]), "+v"(a[o + 1]), "+v"(a[o + 2]), "+v"(a[o + 3]) : "v"(b), "v"(c));
template <int MODE>
__global__ __launch_bounds__(1024) void probe(float* out, unsigned long long* cyc, float seed) {
  __shared__ float lds[4096];
  float a[16];
  for (int i = 0; i < 16; i++) a[i] = seed + i;
  float b = 1.0f + 1e-7f * threadIdx.x, c = 1e-9f;
  f32x4 d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;
  float4 q = {0, 0, 0, 0};
  float e0 = 0.5f, e1 = 0.25f, e2 = 0.125f, e3 = 0.75f;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 0.f;
  const unsigned addr = (threadIdx.x & 1023) * 16, addr4 = (threadIdx.x & 1023) * 4;
  unsigned long long t0, t1;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int r = 0; r < REP; r++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (MODE != 3 && MODE != 6 && MODE != 8 && MODE != 9 && MODE != 11) { FMA4(0) FMA4(4) FMA4(8) FMA4(12) } else { FMA4(0) FMA4(4) FMA4(8) }
      if (MODE == 1 || MODE == 3) {   // four independent 4x4x1 MFMAs
        asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %4, %5, %0\nv_mfma_f32_4x4x1_16b_f32 %1, %4, %5, %1\nv_mfma_f32_4x4x1_16b_f32 %2, %4, %5, %2\nv_mfma_f32_4x4x1_16b_f32 %3, %4, %5, %3"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b), "v"(c));
      }
      if (MODE == 2) {   // four MFMAs on one accumulator
        asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0\nv_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0\nv_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0\nv_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0"
                     : "+v"(d0) : "v"(b), "v"(c));
      }
      if (MODE == 4) asm volatile("ds_read_b128 %0, %1\nds_read_b128 %0, %1 offset:16384\ns_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(addr) : "memory");
      if (MODE == 5) asm volatile("ds_add_f32 %0, %1\nds_add_f32 %0, %1 offset:4096\nds_add_f32 %0, %1 offset:8192" ::"v"(addr4), "v"(c) : "memory");
      if (MODE == 6) asm volatile("v_exp_f32 %0, %0\nv_exp_f32 %1, %1\nv_exp_f32 %2, %2\nv_exp_f32 %3, %3" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));
      if (MODE == 8)
        asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_mov_b32_dpp %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                     "v_mov_b32_dpp %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_mov_b32_dpp %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                     : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));
      if (MODE == 9) asm volatile("v_permlane32_swap_b32 %0, %1\nv_permlane32_swap_b32 %2, %3\nv_permlane32_swap_b32 %0, %2\nv_permlane32_swap_b32 %1, %3" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));
      if (MODE == 10) asm volatile("ds_add_f32 %0, %1\nds_add_f32 %0, %1 offset:4096\nds_add_f32 %0, %1 offset:8192" ::"v"(addr4 & 0x7c), "v"(c) : "memory");   // 32 addresses: 2 lanes each... per half-wave one
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = q.x + q.w + e0 + e1 + e2 + e3 + d0[0] + d1[1] + d2[2] + d3[3] + lds[threadIdx.x];
  for (int i = 0; i < 16; i++) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) atomicMax(cyc, t1 - t0);
}
template <int MODE>
static void run(const char* name) {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 1 << 22); (void)hipMalloc(&cyc, 8);
  printf("%-52s", name);
  for (int threads : {256, 512, 1024}) {
    hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(threads), 0, 0, out, cyc, 1.0f);
    (void)hipDeviceSynchronize(); (void)hipMemset(cyc, 0, 8);
    hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(threads), 0, 0, out, cyc, 1.0f);
    if (hipDeviceSynchronize() != hipSuccess) { printf(" launch failed"); continue; }
    unsigned long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("  %dw: %7.1f", threads / 256, (double)h / (REP * 4) / (threads / 256));   // ticks per wave-iteration and SIMD
  }
  printf("\n");
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  printf("ticks per wave-iteration and SIMD at 1 / 2 / 4 waves per SIMD\n");
  run<0>("16 fma");
  run<11>("12 fma");
  run<3>("12 fma + 4 mfma4x4x1 (independent)");
  run<1>("16 fma + 4 mfma4x4x1 (independent)");
  run<2>("16 fma + 4 mfma4x4x1 (one accumulator)");
  run<6>("12 fma + 4 v_exp_f32");
  run<8>("12 fma + 4 v_mov_b32_dpp");
  run<9>("12 fma + 4 v_permlane32_swap");
  run<4>("16 fma + 2 ds_read_b128 + wait");
  run<5>("16 fma + 3 ds_add_f32 (64 addresses)");
  run<10>("16 fma + 3 ds_add_f32 (32 addresses, 2 lanes each)");
  return 0;
}
