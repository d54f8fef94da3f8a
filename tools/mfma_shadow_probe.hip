// mfma_shadow_probe.hip — how much of a wave's other work hides behind its own MFMAs (development probe, gfx950).
// Every test is ONE asm statement per MFMA (program order fixed): the MFMA followed by N fillers of one kind, 12 statements
// per loop iteration, W waves per SIMD (W = 1: 256 threads, W = 2: 512 threads per workgroup, one workgroup per CU).
// Prints shader cycles (s_memtime) per MFMA of wave 0.
//   hipcc -O3 --offload-arch=gfx950 -o tools/abl/mfma_shadow_probe tools/mfma_shadow_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define R1(x) x
#define R2(x) x x
#define R3(x) x x x
#define R4(x) x x x x
#define R6(x) R4(x) R2(x)
#define R8(x) R4(x) R4(x)
#define R12(x) R8(x) R4(x)

#define M32 "v_mfma_f32_32x32x16_bf16 %0, %6, %7, %0\n"
#define M32B "v_mfma_f32_32x32x16_bf16 %9, %6, %7, %9\n"
#define M16 "v_mfma_f32_16x16x32_bf16 %10, %6, %7, %10\n"
#define M16B "v_mfma_f32_16x16x32_bf16 %11, %6, %7, %11\n"
#define FMA "v_fma_f32 %1, %2, %3, %1\n"
#define FMA4 "v_fma_f32 %1, %2, %3, %1\nv_fma_f32 %2, %1, %3, %2\nv_fma_f32 %3, %1, %2, %3\nv_fma_f32 %4, %1, %3, %4\n"
#define SADD "s_add_i32 %12, %12, 3\n"
#define DSR "ds_read_b128 %13, %8\n"
#define ACR "v_accvgpr_read_b32 %1, %5\n"
#define ACW "v_accvgpr_write_b32 %5, %1\n"
#define MOV "v_mov_b32 %1, %2\n"
#define IND4 "v_fma_f32 %1, %14, %15, %1\nv_fma_f32 %2, %14, %15, %2\nv_fma_f32 %3, %16, %17, %3\nv_fma_f32 %4, %16, %17, %4\n"   // independent
#define EXP "v_exp_f32 %14, %15\n"
#define PK "v_pk_mul_f32 %[p0], %[p0], %[p1]\n"
// a block of the fused MLP kernel in miniature: P fragment reads, six dependent MFMAs with the fillers spread behind them
#define BLK(f) DSR DSR DSR M32 f M32 f M32 f M32 f M32 f M32 f

template <int KIND, int W>
__global__ __launch_bounds__(256 * W, 1) void probe(float* out, unsigned long long* cyc, int iters, const unsigned char* gsrc) {
  extern __shared__ uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  f32x16 acc, acc2;
  f32x4 c16a = {0, 0, 0, 0}, c16b = {0, 0, 0, 0};
  float spare = 1.f;
  for (int r = 0; r < 16; r++) { acc[r] = 0.f; acc2[r] = 0.f; }
  u32x4 a0 = {0x3f803f80u + (unsigned)lane, 0x3f003f00u, 0x3e803e80u, 0x3f803f80u}, b0 = {0x3e003e00u, 0x3f803f80u + (unsigned)lane, 0x3f003f00u, 0x3e803e80u};
  u32x4 frag = {0, 0, 0, 0};
  float v0 = lane * 0.001f, v1 = 1.0f, v2 = 0.5f, v3 = 0.25f, w0 = 0.1f, w1 = 0.2f, w2 = 0.3f, w3 = 0.4f;
  unsigned ldsaddr = lane * 16;
  int s0 = 1;
  unsigned long long t0, t1;
  lds[tid] = make_uint4(tid, 1, 2, 3);
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 12; k++) {
#define ST(body)                                                                                                              \
  asm volatile(body : "+a"(acc), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+a"(spare), "+v"(a0), "+v"(b0), "+v"(ldsaddr),         \
                      "+a"(acc2), "+a"(c16a), "+a"(c16b), "+s"(s0), "+v"(frag), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3)         \
               :: "vcc", "memory");
      // acc2 / c16a / c16b / s0 / frag are read-modify-written by the asm through input operands on purpose (their values are
      // never looked at: this is a timing probe)
      if (KIND == 0) { ST(M32) }                               // dependent chain
      if (KIND == 1) { ST(M32 M32B) }                          // two independent chains alternating (per 2 MFMAs)
      if (KIND == 2) { ST(M32 R2(FMA)) }
      if (KIND == 3) { ST(M32 FMA4) }
      if (KIND == 4) { ST(M32 FMA4 R2(FMA)) }
      if (KIND == 5) { ST(M32 FMA4 FMA4) }
      if (KIND == 6) { ST(M32 FMA4 FMA4 FMA4) }
      if (KIND == 7) { ST(M32 R4(SADD)) }
      if (KIND == 8) { ST(M32 R8(SADD)) }
      if (KIND == 9) { ST(M32 DSR) }
      if (KIND == 10) { ST(M32 R3(DSR)) }
      if (KIND == 11) { ST(M32 R4(ACR)) }
      if (KIND == 12) { ST(M32 R4(ACW)) }
      if (KIND == 13) { ST(M32 R4(MOV)) }
      if (KIND == 14) { ST(M32 FMA4 R4(SADD)) }                // mixed: 4 VALU + 4 SALU
      if (KIND == 15) { ST(M32 FMA4 R4(SADD) DSR) }            // 4 VALU + 4 SALU + 1 LDS read
      if (KIND == 16) { ST(M16) }                              // 16x16x32 dependent chain
      if (KIND == 17) { ST(M16 M16B) }
      if (KIND == 18) { ST(M16 R2(FMA)) }
      if (KIND == 19) { ST(M16 FMA4) }
      if (KIND == 20) { ST(M16 DSR) }
      if (KIND == 21) { ST(M16 FMA4 DSR) }
      if (KIND == 22) { ST(FMA4 FMA4) }                        // no MFMA at all: 8 VALU
      if (KIND == 23) { ST(R8(SADD)) }                         // 8 SALU
      if (KIND == 24) { ST(R3(DSR)) }                          // 3 LDS reads (no wait)
      if (KIND == 25) { ST(M32 "s_nop 7\n") }
      if (KIND == 26) { ST(M32 M32B FMA4 FMA4) }               // two chains, 8 VALU per two MFMAs
      if (KIND == 27) { ST(M32 IND4) }
      if (KIND == 28) { ST(M32 IND4 IND4) }
      if (KIND == 29) { ST(M32 R2(EXP)) }
      if (KIND == 30) { ST(M32 R4(EXP)) }
      if (KIND == 31) { ST(BLK("")) }
      if (KIND == 32) { ST(BLK(R2(FMA))) }
      if (KIND == 33) { ST(BLK(IND4)) }
      if (KIND == 34) { ST(BLK(IND4 R2(FMA))) }
      if (KIND == 35) { ST(BLK(IND4 IND4)) }
      if (KIND == 36) { ST(BLK(IND4 R4(SADD))) }
      if (KIND == 37) { ST(BLK(IND4 R8(SADD))) }
      if (KIND == 38) { ST(IND4 IND4) }
      if (KIND == 39) { ST(R4(EXP)) }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = v0 + v1 + v2 + v3 + spare;
  for (int r = 0; r < 16; r++) s += acc[r];
  out[blockIdx.x * 256 * W + tid] = s;
  s += w0 + w1 + w2 + w3 + (float)s0 + __uint_as_float(frag[0]) + __uint_as_float(a0[0] ^ b0[0] ^ ldsaddr) + acc2[0] + c16a[0] + c16b[0];
  out[blockIdx.x * 256 * W + tid] = s;
  if (lane == 0 && (tid >> 6) < 8) { cyc[blockIdx.x * 16 + (tid >> 6)] = t0; cyc[blockIdx.x * 16 + 8 + (tid >> 6)] = t1; }
  (void)gsrc;
}
template <int KIND, int W>
void run(const char* name, int per, float* out, unsigned long long* cyc) {
  const int iters = 200, blocks = 256;
  (void)hipFuncSetAttribute((const void*)probe<KIND, W>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipLaunchKernelGGL((probe<KIND, W>), dim3(blocks), dim3(256 * W), 100 * 1024, 0, out, cyc, iters, nullptr);
  (void)hipDeviceSynchronize();
  unsigned long long h[16];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  // wave 0 alone, and the workgroup as a whole (first start to last end): with two waves per SIMD the older wave has priority
  unsigned long long lo = h[0], hi = h[8];
  for (int w = 0; w < 4 * W; w++) { if (h[w] < lo) lo = h[w]; if (h[8 + w] > hi) hi = h[8 + w]; }
  printf("W=%d %-58s wave0 %7.1f  all waves %7.1f cycles per statement and wave-slot (%d MFMA)\n", W, name, (h[8] - h[0]) / (double)(iters * 12),
         (hi - lo) / (double)(iters * 12) / W, per);
}
#define BOTH(K, name, per) run<K, 1>(name, per, out, cyc); run<K, 2>(name, per, out, cyc);
int main() {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, sizeof(float) * 256 * 512);
  (void)hipMalloc(&cyc, sizeof(unsigned long long) * 256 * 16);
  BOTH(0, "32x32x16 dependent chain", 1)
  BOTH(1, "32x32x16 two chains (2 MFMA per statement)", 2)
  BOTH(2, "32x32x16 + 2 v_fma", 1)
  BOTH(3, "32x32x16 + 4 v_fma", 1)
  BOTH(4, "32x32x16 + 6 v_fma", 1)
  BOTH(5, "32x32x16 + 8 v_fma", 1)
  BOTH(6, "32x32x16 + 12 v_fma", 1)
  BOTH(7, "32x32x16 + 4 s_add", 1)
  BOTH(8, "32x32x16 + 8 s_add", 1)
  BOTH(9, "32x32x16 + 1 ds_read_b128", 1)
  BOTH(10, "32x32x16 + 3 ds_read_b128", 1)
  BOTH(11, "32x32x16 + 4 v_accvgpr_read", 1)
  BOTH(12, "32x32x16 + 4 v_accvgpr_write", 1)
  BOTH(13, "32x32x16 + 4 v_mov", 1)
  BOTH(14, "32x32x16 + 4 v_fma + 4 s_add", 1)
  BOTH(15, "32x32x16 + 4 v_fma + 4 s_add + 1 ds_read", 1)
  BOTH(16, "16x16x32 dependent chain", 1)
  BOTH(17, "16x16x32 two chains (2 MFMA per statement)", 2)
  BOTH(18, "16x16x32 + 2 v_fma", 1)
  BOTH(19, "16x16x32 + 4 v_fma", 1)
  BOTH(20, "16x16x32 + 1 ds_read_b128", 1)
  BOTH(21, "16x16x32 + 4 v_fma + 1 ds_read_b128", 1)
  BOTH(22, "8 v_fma, no MFMA", 0)
  BOTH(23, "8 s_add, no MFMA", 0)
  BOTH(24, "3 ds_read_b128, no MFMA", 0)
  BOTH(25, "32x32x16 + s_nop 7", 1)
  BOTH(26, "two chains + 8 v_fma (2 MFMA per statement)", 2)
  BOTH(27, "32x32x16 + 4 independent v_fma", 1)
  BOTH(28, "32x32x16 + 8 independent v_fma", 1)
  BOTH(29, "32x32x16 + 2 v_exp", 1)
  BOTH(30, "32x32x16 + 4 v_exp", 1)
  BOTH(38, "8 independent v_fma, no MFMA", 0)
  BOTH(39, "4 v_exp, no MFMA", 0)
  BOTH(31, "block: 3 ds_read + 6 MFMA", 6)
  BOTH(32, "block: 3 ds_read + 6 x (MFMA + 2 v_fma)", 6)
  BOTH(33, "block: 3 ds_read + 6 x (MFMA + 4 ind v_fma)", 6)
  BOTH(34, "block: 3 ds_read + 6 x (MFMA + 6 v_fma)", 6)
  BOTH(35, "block: 3 ds_read + 6 x (MFMA + 8 ind v_fma)", 6)
  BOTH(36, "block: 3 ds_read + 6 x (MFMA + 4 v_fma + 4 s_add)", 6)
  BOTH(37, "block: 3 ds_read + 6 x (MFMA + 4 v_fma + 8 s_add)", 6)
  return 0;
}
