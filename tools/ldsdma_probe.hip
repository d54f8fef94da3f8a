// ldsdma_probe.hip — issue cost and latency of global_load_lds_dwordx4 bursts (development probe).  Every wave of a 256-thread
// workgroup per CU issues a burst of B one-KB pieces from an L2-sized table, then waits for all of them; prints the shader
// cycles the issuing took (per piece) and the cycles until the last piece had landed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int B>
__global__ __launch_bounds__(256, 1) void probe(const unsigned char* tab, int npieces, unsigned long long* out, int iters) {
  extern __shared__ uint4 lds[];
  unsigned char* ring = reinterpret_cast<unsigned char*>(lds);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned long long issue = 0, wait = 0;
  unsigned q = blockIdx.x * 977u + wave * 131u;
  for (int it = 0; it < iters; it++) {
    unsigned long long t0, t1, t2;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll
    for (int k = 0; k < B; k++) {
      q = (q * 1664525u + 1013904223u);
      const unsigned piece = (q >> 8) % (unsigned)npieces;
      const unsigned char* g = tab + (size_t)piece * 1024 + lane * 16;
      unsigned char* l = ring + ((wave * B + k) << 10);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
    issue += t1 - t0; wait += t2 - t1;
  }
  if (lane == 0) { out[(blockIdx.x * 4 + wave) * 2] = issue; out[(blockIdx.x * 4 + wave) * 2 + 1] = wait; }
}
template <int B>
void run(const unsigned char* tab, int npieces, unsigned long long* out, int blocks) {
  const int iters = 50;
  (void)hipFuncSetAttribute((const void*)probe<B>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
  hipLaunchKernelGGL((probe<B>), dim3(blocks), dim3(256), 144 * 1024, 0, tab, npieces, out, iters);
  hipLaunchKernelGGL((probe<B>), dim3(blocks), dim3(256), 144 * 1024, 0, tab, npieces, out, iters);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 8);
  (void)hipMemcpy(h.data(), out, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
  double is = 0, wt = 0;
  for (int i = 0; i < blocks * 4; i++) { is += h[2 * i]; wt += h[2 * i + 1]; }
  is /= blocks * 4.0 * iters; wt /= blocks * 4.0 * iters;
  printf("burst of %2d pieces per wave (4 waves/CU, %3d CUs): issue %7.0f cycles (%5.0f per piece), then %7.0f cycles until all landed; %.1f B/cycle/CU\n",
         B, blocks, is, is / B, wt, 4.0 * B * 1024 / (is + wt));
}
int main() {
  const int npieces = 2304;   // 2.3 MB table: what the two water species' streams are
  unsigned char* tab; unsigned long long* out;
  (void)hipMalloc(&tab, (size_t)npieces * 1024);
  (void)hipMemset(tab, 1, (size_t)npieces * 1024);
  (void)hipMalloc(&out, sizeof(unsigned long long) * 256 * 8);
  for (int blocks : {1, 256}) {
    run<1>(tab, npieces, out, blocks); run<2>(tab, npieces, out, blocks); run<4>(tab, npieces, out, blocks); run<8>(tab, npieces, out, blocks);
    run<12>(tab, npieces, out, blocks); run<16>(tab, npieces, out, blocks); run<24>(tab, npieces, out, blocks); run<32>(tab, npieces, out, blocks);
  }
  return 0;
}
