// mfma_gap_probe.hip — what fits beside a v_mfma_f32_32x32x16_bf16 of the same wave, one wave per SIMD (development probe).
// Between consecutive (dependent) MFMAs: N instructions of one kind (inline asm, register operands, no dependence on the
// MFMA); prints shader cycles per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define REP1(x) x
#define REP2(x) x x
#define REP3(x) x x x
#define REP4(x) x x x x
#define REP6(x) x x x x x x
#define REP8(x) x x x x x x x x

template <int KIND, int N>
__device__ __forceinline__ void filler(float& v0, float& v1, float& v2, float& v3, unsigned& u0, unsigned& u1, int& s0, int& s1) {
#define EMIT(str) \
  if (N == 1) asm volatile(REP1(str) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(u0), "+v"(u1), "+s"(s0), "+s"(s1)); \
  if (N == 2) asm volatile(REP2(str) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(u0), "+v"(u1), "+s"(s0), "+s"(s1)); \
  if (N == 3) asm volatile(REP3(str) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(u0), "+v"(u1), "+s"(s0), "+s"(s1)); \
  if (N == 4) asm volatile(REP4(str) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(u0), "+v"(u1), "+s"(s0), "+s"(s1)); \
  if (N == 6) asm volatile(REP6(str) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(u0), "+v"(u1), "+s"(s0), "+s"(s1)); \
  if (N == 8) asm volatile(REP8(str) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(u0), "+v"(u1), "+s"(s0), "+s"(s1));
  if (KIND == 0) { EMIT("v_fma_f32 %0, %1, %2, %3\n") }                    // plain VALU, registers only
  if (KIND == 1) { EMIT("v_and_b32 %4, 0xffff0000, %5\n") }                // VALU with a 32-bit literal
  if (KIND == 3) { EMIT("s_add_i32 %6, %6, %7\n") }                        // SALU
  if (KIND == 4) { EMIT("v_exp_f32 %0, %1\n") }                            // transcendental
  if (KIND == 5) { EMIT("v_accvgpr_write_b32 a200, %1\n") }                // VGPR -> AGPR
  if (KIND == 6) { EMIT("v_mov_b32 %0, %1\n") }
  if (KIND == 7) { EMIT("s_nop 0\n") }
  if (KIND == 8) { EMIT("v_cndmask_b32 %0, %1, %2, vcc\n") }
#undef EMIT
}

template <int KIND, int N>
__global__ __launch_bounds__(256, 1) void probe(float* out, unsigned long long* cyc, int iters) {
  extern __shared__ uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  f32x16 acc;
  for (int r = 0; r < 16; r++) acc[r] = 0.f;
  u32x4 a0 = {0x3f803f80u + (unsigned)lane, 0x3f003f00u, 0x3e803e80u, 0x3f803f80u}, b0 = {0x3e003e00u, 0x3f803f80u + (unsigned)lane, 0x3f003f00u, 0x3e803e80u};
  float v0 = lane * 0.001f, v1 = 1.0f, v2 = 0.5f, v3 = 0.25f;
  unsigned u0 = lane, u1 = 77;
  int s0 = 1, s1 = 3;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 12; k++) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b0), acc, 0, 0, 0);
      filler<KIND, N>(v0, v1, v2, v3, u0, u1, s0, s1);
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = v0 + v1 + v2 + v3 + (float)(u0 + u1) + (float)(s0 + s1);
  for (int r = 0; r < 16; r++) s += acc[r];
  out[blockIdx.x * 256 + tid] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int KIND, int N>
void run(const char* name, float* out, unsigned long long* cyc) {
  const int iters = 200, blocks = 256;
  (void)hipFuncSetAttribute((const void*)probe<KIND, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipLaunchKernelGGL((probe<KIND, N>), dim3(blocks), dim3(256), 100 * 1024, 0, out, cyc, iters);
  (void)hipDeviceSynchronize();
  unsigned long long h[4];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-36s %d per MFMA: %6.1f cycles/MFMA\n", name, N, h[0] / (double)(iters * 12));
}
#define ALLN(K, name) run<K, 1>(name, out, cyc); run<K, 2>(name, out, cyc); run<K, 3>(name, out, cyc); run<K, 4>(name, out, cyc); run<K, 6>(name, out, cyc); run<K, 8>(name, out, cyc);
int main() {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, sizeof(float) * 256 * 256);
  (void)hipMalloc(&cyc, sizeof(unsigned long long) * 256 * 4);
  ALLN(0, "v_fma_f32 (registers)")
  ALLN(1, "v_and_b32 with 32-bit literal")
  ALLN(3, "s_add_i32")
  ALLN(4, "v_exp_f32")
  ALLN(5, "v_accvgpr_write_b32")
  ALLN(6, "v_mov_b32")
  ALLN(7, "s_nop 0")
  ALLN(8, "v_cndmask_b32")
  return 0;
}
