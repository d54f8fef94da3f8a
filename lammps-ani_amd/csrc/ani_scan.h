// ani_scan.h — block-wide exclusive scan shared by the rebuild-time kernels (device code only).
#pragma once
#include <hip/hip_runtime.h>

namespace ani {

// exclusive scan of one int per thread over a block of up to 1024 threads (16 waves); returns the block total in
// `total`.  wave_sums: 16 ints of LDS.
__device__ __forceinline__ int block_exclusive_scan(int v, int& total, int* wave_sums) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off);
    if (lane >= off) incl += t;
  }
  __syncthreads();
  if (lane == 63) wave_sums[wave] = incl;
  __syncthreads();
  int base = 0, tot = 0;
  const int nw = blockDim.x >> 6;
  for (int w = 0; w < nw; w++) {
    const int s = wave_sums[w];
    if (w < wave) base += s;
    tot += s;
  }
  total = tot;
  return base + incl - v;
}

}  // namespace ani
