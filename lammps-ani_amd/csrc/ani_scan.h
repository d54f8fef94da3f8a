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

// exclusive scan of n values by ONE block, VPT consecutive values per thread and round (a 1024-thread block covers
// 8192 values per round at VPT = 8: two barriers per 8192 values instead of per 1024).  load(i) gives value i,
// store(i, exclusive_prefix, value) receives the result; returns the total.
template <int VPT, typename Load, typename Store>
__device__ __forceinline__ int block_scan_rounds(int n, int* wave_sums, Load load, Store store) {
  int carry = 0;
  for (int base = 0; base < n; base += blockDim.x * VPT) {
    const int i0 = base + threadIdx.x * VPT;
    int v[VPT], s = 0;
#pragma unroll
    for (int k = 0; k < VPT; k++) {
      v[k] = (i0 + k < n) ? load(i0 + k) : 0;
      s += v[k];
    }
    int total;
    int ex = carry + block_exclusive_scan(s, total, wave_sums);
#pragma unroll
    for (int k = 0; k < VPT; k++) {
      if (i0 + k < n) store(i0 + k, ex, v[k]);
      ex += v[k];
    }
    carry += total;
  }
  return carry;
}

}  // namespace ani
