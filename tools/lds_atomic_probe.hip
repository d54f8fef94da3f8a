// lds_atomic_probe.hip -- what an LDS read-modify-write costs on gfx950, per wave-instruction, when all waves of a CU issue
// them back to back (the LDS is one unit per CU): float add against integer add, 64-bit add, plain write and read, with 64 /
// 32 / 8 active lanes and with lanes sharing addresses.  The AEV backward kernel's pair loop issues six ds_add_f32 per step.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic_probe lds_atomic_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 256
template <int MODE>
__global__ __launch_bounds__(1024) void probe(float* out, unsigned long long* cyc, int active, int share) {
  __shared__ float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // every wave its own 1 KB; `share` lanes per address
  const unsigned addr = (wave * 256 + (lane / share)) * 4;
  float c = 1e-6f * lane, r0 = 0.f;
  unsigned long long c64 = lane, t0, t1;
  double d64 = 1e-6 * lane;
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  if (lane < active) {
    for (int r = 0; r < REP; r++) {
      if (MODE == 0) asm volatile("ds_add_f32 %0, %1\nds_add_f32 %0, %1 offset:16384\nds_add_f32 %0, %1 offset:32768\nds_add_f32 %0, %1 offset:49152" ::"v"(addr), "v"(c) : "memory");
      if (MODE == 1) asm volatile("ds_add_u32 %0, %1\nds_add_u32 %0, %1 offset:16384\nds_add_u32 %0, %1 offset:32768\nds_add_u32 %0, %1 offset:49152" ::"v"(addr), "v"(c) : "memory");
      if (MODE == 2) asm volatile("ds_write_b32 %0, %1\nds_write_b32 %0, %1 offset:16384\nds_write_b32 %0, %1 offset:32768\nds_write_b32 %0, %1 offset:49152" ::"v"(addr), "v"(c) : "memory");
      if (MODE == 3) asm volatile("ds_read_b32 %0, %1\nds_read_b32 %0, %1 offset:16384\nds_read_b32 %0, %1 offset:32768\nds_read_b32 %0, %1 offset:49152\ns_waitcnt lgkmcnt(0)" : "=v"(r0) : "v"(addr) : "memory");
      if (MODE == 4) asm volatile("ds_add_u64 %0, %1\nds_add_u64 %0, %1 offset:16384\nds_add_u64 %0, %1 offset:32768\nds_add_u64 %0, %1 offset:49152" ::"v"(addr * 2 & 0x3fff), "v"(c64) : "memory");
      if (MODE == 8) asm volatile("ds_add_f64 %0, %1\nds_add_f64 %0, %1 offset:16384\nds_add_f64 %0, %1 offset:32768\nds_add_f64 %0, %1 offset:49152" ::"v"(addr * 2 & 0x3fff), "v"(d64) : "memory");
      if (MODE == 5) asm volatile("ds_add_rtn_f32 %0, %1, %2\nds_add_rtn_f32 %0, %1, %2 offset:16384\nds_add_rtn_f32 %0, %1, %2 offset:32768\nds_add_rtn_f32 %0, %1, %2 offset:49152\ns_waitcnt lgkmcnt(0)" : "=v"(r0) : "v"(addr), "v"(c) : "memory");
      if (MODE == 6) asm volatile("ds_max_f32 %0, %1\nds_max_f32 %0, %1 offset:16384\nds_max_f32 %0, %1 offset:32768\nds_max_f32 %0, %1 offset:49152" ::"v"(addr), "v"(c) : "memory");
      if (MODE == 7) asm volatile("ds_pk_add_f16 %0, %1\nds_pk_add_f16 %0, %1 offset:16384\nds_pk_add_f16 %0, %1 offset:32768\nds_pk_add_f16 %0, %1 offset:49152" ::"v"(addr), "v"(c) : "memory");
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + lds[threadIdx.x];
  if (lane == 0) atomicMax(cyc, t1 - t0);
}
template <int MODE>
static void run(const char* name, int active, int share) {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 1 << 22); (void)hipMalloc(&cyc, 8);
  printf("%-24s active %2d, %d lane(s)/address:", name, active, share);
  for (int threads : {256, 1024}) {
    hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(threads), 0, 0, out, cyc, active, share);
    (void)hipDeviceSynchronize(); (void)hipMemset(cyc, 0, 8);
    hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(threads), 0, 0, out, cyc, active, share);
    if (hipDeviceSynchronize() != hipSuccess) { printf(" launch failed"); continue; }
    unsigned long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("  %2d waves: %7.1f ticks per wave-instruction", threads / 64, (double)h / (REP * 4) / (threads / 64));
  }
  printf("\n");
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  printf("LDS instructions issued back to back by all waves of one CU; ticks (s_memtime) per wave-instruction of the CU's LDS\n");
  run<0>("ds_add_f32", 64, 1); run<0>("ds_add_f32", 32, 1); run<0>("ds_add_f32", 8, 1); run<0>("ds_add_f32", 64, 2); run<0>("ds_add_f32", 64, 8);
  run<1>("ds_add_u32", 64, 1); run<1>("ds_add_u32", 32, 1); run<1>("ds_add_u32", 8, 1); run<1>("ds_add_u32", 64, 8);
  run<4>("ds_add_u64", 64, 1); run<4>("ds_add_u64", 8, 1);
  run<8>("ds_add_f64", 64, 1); run<8>("ds_add_f64", 8, 1);
  run<5>("ds_add_rtn_f32", 64, 1);
  run<6>("ds_max_f32", 64, 1);
  run<7>("ds_pk_add_f16", 64, 1);
  run<2>("ds_write_b32", 64, 1); run<2>("ds_write_b32", 8, 1);
  run<3>("ds_read_b32", 64, 1);
  return 0;
}
