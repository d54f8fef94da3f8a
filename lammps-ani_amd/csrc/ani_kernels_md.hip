// ani_kernels_md.hip — the timestep loop's own kernels (include/ani_md.h): what LAMMPS' fix nve, fix langevin,
// Neighbor::check_distance and Comm do around PairANI::compute in the reference's runs
// (examples/benchmark/in.lammps:24-27,54-72).  Test / bench infrastructure of the LAMMPS-free loop, not the pair style.
#include <hip/hip_runtime.h>

#include "../../include/ani_md.h"

namespace {

__global__ __launch_bounds__(256) void initial_integrate_kernel(double* __restrict__ x, double* __restrict__ v,
                                                                const double* __restrict__ f, const double* __restrict__ dtfm,
                                                                double dt, int n, const double* __restrict__ xb,
                                                                double* __restrict__ d2max) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double d2 = 0.0;
  if (i < n) {
    const double s = dtfm[i];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const double vk = v[3 * i + k] + s * f[3 * i + k];
      const double xk = x[3 * i + k] + dt * vk;
      v[3 * i + k] = vk;
      x[3 * i + k] = xk;
      const double d = xk - xb[3 * i + k];
      d2 += d * d;
    }
  }
  // workgroup maximum, then at most one atomic per workgroup -- and none once the running maximum is larger (it only
  // grows between two looks of the host: a stale read costs an atomic, never a missed one).  1 563 same-address atomics
  // made this kernel 21 us at 100 002 atoms.  Non-negative doubles order like their bit patterns.
  for (int off = 32; off > 0; off >>= 1) d2 = fmax(d2, __shfl_xor(d2, off));
  __shared__ double wmax[4];
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = d2;
  __syncthreads();
  if (threadIdx.x == 0) {
    d2 = fmax(fmax(wmax[0], wmax[1]), fmax(wmax[2], wmax[3]));
    if (d2 > *reinterpret_cast<volatile double*>(d2max))
      atomicMax(reinterpret_cast<unsigned long long*>(d2max), (unsigned long long)__double_as_longlong(d2));
  }
}

__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {   // splitmix64 finaliser
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void final_integrate_kernel(double* __restrict__ v, double* __restrict__ f,
                                                              const double* __restrict__ dtfm, int n, int langevin,
                                                              const double* __restrict__ g1, const double* __restrict__ g2,
                                                              const long long* __restrict__ tag, unsigned long long seed,
                                                              unsigned long long step) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double s = dtfm[i];
  const unsigned long long key = mix64(seed + 0x9E3779B97F4A7C15ULL * (step + 1)) ^ (0xD1B54A32D192ED03ULL * (unsigned long long)(tag ? tag[i] : i));
#pragma unroll
  for (int k = 0; k < 3; k++) {
    double fk = f[3 * i + k];
    const double vk = v[3 * i + k];
    if (langevin) {
      const unsigned long long h = mix64(key + 0x9E3779B97F4A7C15ULL * (k + 1));
      const double r = (double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5;   // uniform in [-0.5, 0.5)
      fk += g1[i] * vk + g2[i] * r;
      f[3 * i + k] = fk;
    }
    v[3 * i + k] = vk + s * fk;
  }
}

// final_integrate of step n and initial_integrate of step n + 1 in one pass over the owned atoms (nothing happens between
// them in a run without per-step output): f += thermostat;  v += s f  (the full-step velocity, not stored);  v += s f;  x += dt v
__global__ __launch_bounds__(256) void final_initial_integrate_kernel(double* __restrict__ x, double* __restrict__ v, double* __restrict__ f,
                                                                      const double* __restrict__ dtfm, double dt, int n, int langevin,
                                                                      const double* __restrict__ g1, const double* __restrict__ g2,
                                                                      const long long* __restrict__ tag, unsigned long long seed,
                                                                      unsigned long long step, const double* __restrict__ xb,
                                                                      double* __restrict__ d2max) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double d2 = 0.0;
  if (i < n) {
    const double s = dtfm[i];
    const unsigned long long key = mix64(seed + 0x9E3779B97F4A7C15ULL * (step + 1)) ^ (0xD1B54A32D192ED03ULL * (unsigned long long)(tag ? tag[i] : i));
#pragma unroll
    for (int k = 0; k < 3; k++) {
      double fk = f[3 * i + k];
      double vk = v[3 * i + k];
      if (langevin) {
        const unsigned long long h = mix64(key + 0x9E3779B97F4A7C15ULL * (k + 1));
        const double r = (double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5;
        fk += g1[i] * vk + g2[i] * r;
        f[3 * i + k] = fk;
      }
      vk += s * fk;            // final_integrate of the step that ends
      vk += s * fk;            // initial_integrate of the step that begins (same rounding as the two kernels)
      const double xk = x[3 * i + k] + dt * vk;
      v[3 * i + k] = vk;
      x[3 * i + k] = xk;
      const double d = xk - xb[3 * i + k];
      d2 += d * d;
    }
  }
  for (int off = 32; off > 0; off >>= 1) d2 = fmax(d2, __shfl_xor(d2, off));
  __shared__ double wmax[4];
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = d2;
  __syncthreads();
  if (threadIdx.x == 0) {
    d2 = fmax(fmax(wmax[0], wmax[1]), fmax(wmax[2], wmax[3]));
    if (d2 > *reinterpret_cast<volatile double*>(d2max))
      atomicMax(reinterpret_cast<unsigned long long*>(d2max), (unsigned long long)__double_as_longlong(d2));
  }
}

__global__ __launch_bounds__(256) void forward_ghosts_kernel(double* __restrict__ x, const long long* __restrict__ owner,
                                                             const double* __restrict__ shift, int nlocal, int nghost) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 3 * nghost) return;
  const int g = t / 3, k = t - 3 * g;
  x[3 * (size_t)nlocal + t] = x[3 * owner[g] + k] + shift[t];
}

__global__ __launch_bounds__(256) void pack_ghosts_kernel(const double* __restrict__ x, const long long* __restrict__ owner,
                                                          const double* __restrict__ shift, int nsend, double* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 3 * nsend) return;
  const int g = t / 3, k = t - 3 * g;
  out[t] = x[3 * owner[g] + k] + shift[t];
}

__global__ __launch_bounds__(256) void unpack_reverse_kernel(double* __restrict__ f, const long long* __restrict__ owner, int nsend,
                                                             const double* __restrict__ in) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 3 * nsend) return;
  const int g = t / 3, k = t - 3 * g;
  atomicAdd(&f[3 * owner[g] + k], in[t]);   // an atom can be a ghost of several bricks / images
}

__global__ __launch_bounds__(256) void reverse_ghosts_kernel(double* __restrict__ f, const long long* __restrict__ owner,
                                                             int nlocal, int nghost) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 3 * nghost) return;
  const int g = t / 3, k = t - 3 * g;
  atomicAdd(&f[3 * owner[g] + k], f[3 * (size_t)nlocal + t]);
}

// the same with the ghost block in the caller's order: message slot g is ghost ghost_of[g]
__global__ __launch_bounds__(256) void reverse_ghosts_ordered_kernel(double* __restrict__ f, const long long* __restrict__ owner,
                                                                     const long long* __restrict__ ghost_of, int nlocal, int nghost) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 3 * nghost) return;
  const int g = t / 3, k = t - 3 * g;
  atomicAdd(&f[3 * owner[g] + k], f[3 * ((size_t)nlocal + (size_t)ghost_of[g]) + k]);
}

// rows[k] = src[idx[k]] (gather) / dst[idx[k]] = rows[k] (scatter): three doubles per row
__global__ __launch_bounds__(256) void gather_rows_kernel(const double* __restrict__ src, const long long* __restrict__ idx, int n,
                                                          double* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 3 * n) return;
  const int g = t / 3, k = t - 3 * g;
  out[t] = src[3 * idx[g] + k];
}
__global__ __launch_bounds__(256) void scatter_rows_kernel(double* __restrict__ dst, const long long* __restrict__ idx, int n,
                                                           const double* __restrict__ in) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 3 * n) return;
  const int g = t / 3, k = t - 3 * g;
  dst[3 * idx[g] + k] = in[t];
}


// ---- re-neighbouring of the stand-in loop, natively (LAMMPS: Domain::pbc, Comm::borders) ---------------------------------
__global__ __launch_bounds__(256) void wrap_kernel(double* __restrict__ x, int n, double lo0, double lo1, double lo2, double l0, double l1,
                                                   double l2, int pmask) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 3 * n) return;
  const int k = t % 3;
  if (!((pmask >> k) & 1)) return;
  const double lo = k == 0 ? lo0 : (k == 1 ? lo1 : lo2), L = k == 0 ? l0 : (k == 1 ? l1 : l2);
  double v = x[t];
  v -= floor((v - lo) / L) * L;
  if (v >= lo + L) v = lo;   // a coordinate that rounds up onto the upper face
  x[t] = v;
}

// Ghost shell: atom a is a ghost of combination c (a brick of the decomposition and an image shift) when its UNSHIFTED position
// lies in [clo[c], chi[c]) in every dimension.  Hits are listed combination-major, atoms ascending (the order the exchange and
// the receiver's ghost block agree on): count per (combination, block of 256 atoms), scan, fill.
__device__ __forceinline__ bool shell_hit(const double* __restrict__ x, int a, int n, const double* __restrict__ clo,
                                          const double* __restrict__ chi, int c) {
  if (a >= n) return false;
  bool in = true;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const double v = x[3 * a + k];
    in = in && v >= clo[3 * c + k] && v < chi[3 * c + k];
  }
  return in;
}
__global__ __launch_bounds__(256) void shell_count_kernel(const double* __restrict__ x, int n, const double* __restrict__ clo,
                                                          const double* __restrict__ chi, int nblk, int* __restrict__ blk_cnt) {
  const int b = blockIdx.x, c = blockIdx.y;
  const bool hit = shell_hit(x, b * 256 + threadIdx.x, n, clo, chi, c);
  __shared__ int wc[4];
  const int cnt = __popcll(__ballot(hit));
  if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) blk_cnt[c * nblk + b] = wc[0] + wc[1] + wc[2] + wc[3];
}
// exclusive scan of blk_cnt[ncombo * nblk] by one block; out_counts[0] = total, out_counts[1 + c] = hits of combination c
__global__ __launch_bounds__(1024) void shell_scan_kernel(const int* __restrict__ blk_cnt, int* __restrict__ blk_off, int ncombo, int nblk,
                                                          int* __restrict__ out_counts) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int total_n = ncombo * nblk;
  for (int base = 0; base < total_n; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < total_n ? blk_cnt[i] : 0;
    // inclusive scan inside the wave, then over the waves
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(incl, off);
      if ((threadIdx.x & 63) >= off) incl += t;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    int before = carry_s, tot = 0;
    for (int w = 0; w < 16; w++) { if (w < (int)(threadIdx.x >> 6)) before += wsum[w]; tot += wsum[w]; }
    if (i < total_n) blk_off[i] = before + incl - v;
    __syncthreads();
    if (threadIdx.x == 0) carry_s += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) out_counts[0] = carry_s;
  __syncthreads();
  for (int c = threadIdx.x; c < ncombo; c += 1024) {
    const int beg = blk_off[c * nblk];
    const int end = c + 1 < ncombo ? blk_off[(c + 1) * nblk] : carry_s;
    out_counts[1 + c] = end - beg;
  }
}
__global__ __launch_bounds__(256) void shell_fill_kernel(const double* __restrict__ x, int n, const double* __restrict__ clo,
                                                         const double* __restrict__ chi, const double* __restrict__ cshift, int nblk,
                                                         const int* __restrict__ blk_off, long long* __restrict__ send_idx,
                                                         double* __restrict__ send_shift) {
  const int b = blockIdx.x, c = blockIdx.y, a = b * 256 + threadIdx.x;
  const bool hit = shell_hit(x, a, n, clo, chi, c);
  __shared__ int wc[4];
  const unsigned long long m = __ballot(hit);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wc[wave] = __popcll(m);
  __syncthreads();
  if (!hit) return;
  int pos = blk_off[c * nblk + b] + __popcll(m & ((1ULL << lane) - 1ULL));
  for (int w = 0; w < wave; w++) pos += wc[w];
  send_idx[pos] = a;
  send_shift[3 * pos] = cshift[3 * c]; send_shift[3 * pos + 1] = cshift[3 * c + 1]; send_shift[3 * pos + 2] = cshift[3 * c + 2];
}
// one rank: the ghosts are images of owned atoms -- positions and species appended behind the owned block
__global__ __launch_bounds__(256) void append_ghosts_kernel(double* __restrict__ x, int* __restrict__ species, int nlocal,
                                                            const long long* __restrict__ owner, const double* __restrict__ shift, int nghost) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nghost) return;
  const long long o = owner[g];
#pragma unroll
  for (int k = 0; k < 3; k++) x[3 * (size_t)(nlocal + g) + k] = x[3 * o + k] + shift[3 * g + k];
  species[nlocal + g] = species[o];
}
// Neighbor::decide's look at the loop: *out = the running displacement maximum (then zeroed), or +inf when the last energy is not
// finite (a capacity overflow or a blown-up step: the device entry points cannot return an error)
__global__ void check_kernel(double* __restrict__ d2max, const double* __restrict__ ev, double* __restrict__ out) {
  const double e = ev[0];
  const bool finite = e == e && e - e == 0.0;
  out[0] = finite ? d2max[0] : __longlong_as_double(0x7ff0000000000000LL);
  d2max[0] = 0.0;
}

}  // namespace

extern "C" {

int ani_md_gather_rows(const double* src, const int64_t* idx, int n, double* out, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((3 * n + 255) / 256), dim3(256), 0, (hipStream_t)stream, src,
                     reinterpret_cast<const long long*>(idx), n, out);
  return (int)hipGetLastError();
}

int ani_md_scatter_rows(double* dst, const int64_t* idx, int n, const double* in, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3((3 * n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dst,
                     reinterpret_cast<const long long*>(idx), n, in);
  return (int)hipGetLastError();
}

int ani_md_initial_integrate(double* x, double* v, const double* f, const double* dtfm, double dt, int nlocal,
                             const double* x_built, double* d2max, void* stream) {
  if (nlocal <= 0) return 0;
  hipLaunchKernelGGL(initial_integrate_kernel, dim3((nlocal + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, v, f, dtfm, dt,
                     nlocal, x_built, d2max);
  return (int)hipGetLastError();
}

int ani_md_final_integrate(double* v, double* f, const double* dtfm, int nlocal, int langevin, const double* g1,
                           const double* g2, const int64_t* tag, uint64_t seed, uint64_t step, void* stream) {
  if (nlocal <= 0) return 0;
  hipLaunchKernelGGL(final_integrate_kernel, dim3((nlocal + 255) / 256), dim3(256), 0, (hipStream_t)stream, v, f, dtfm, nlocal,
                     langevin, g1, g2, reinterpret_cast<const long long*>(tag), (unsigned long long)seed, (unsigned long long)step);
  return (int)hipGetLastError();
}

int ani_md_final_initial_integrate(double* x, double* v, double* f, const double* dtfm, double dt, int nlocal, int langevin,
                                   const double* g1, const double* g2, const int64_t* tag, uint64_t seed, uint64_t step,
                                   const double* x_built, double* d2max, void* stream) {
  if (nlocal <= 0) return 0;
  hipLaunchKernelGGL(final_initial_integrate_kernel, dim3((nlocal + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, v, f, dtfm, dt,
                     nlocal, langevin, g1, g2, reinterpret_cast<const long long*>(tag), (unsigned long long)seed,
                     (unsigned long long)step, x_built, d2max);
  return (int)hipGetLastError();
}

int ani_md_forward_ghosts(double* x, const int64_t* owner, const double* shift, int nlocal, int nghost, void* stream) {
  if (nghost <= 0) return 0;
  hipLaunchKernelGGL(forward_ghosts_kernel, dim3((3 * nghost + 255) / 256), dim3(256), 0, (hipStream_t)stream, x,
                     reinterpret_cast<const long long*>(owner), shift, nlocal, nghost);
  return (int)hipGetLastError();
}

int ani_md_reverse_ghosts(double* f, const int64_t* owner, int nlocal, int nghost, void* stream) {
  if (nghost <= 0) return 0;
  hipLaunchKernelGGL(reverse_ghosts_kernel, dim3((3 * nghost + 255) / 256), dim3(256), 0, (hipStream_t)stream, f,
                     reinterpret_cast<const long long*>(owner), nlocal, nghost);
  return (int)hipGetLastError();
}

int ani_md_reverse_ghosts_ordered(double* f, const int64_t* owner, const int64_t* ghost_of, int nlocal, int nghost, void* stream) {
  if (nghost <= 0) return 0;
  if (!ghost_of) return ani_md_reverse_ghosts(f, owner, nlocal, nghost, stream);
  hipLaunchKernelGGL(reverse_ghosts_ordered_kernel, dim3((3 * nghost + 255) / 256), dim3(256), 0, (hipStream_t)stream, f,
                     reinterpret_cast<const long long*>(owner), reinterpret_cast<const long long*>(ghost_of), nlocal, nghost);
  return (int)hipGetLastError();
}

int ani_md_pack_ghosts(const double* x, const int64_t* owner, const double* shift, int nsend, double* out, void* stream) {
  if (nsend <= 0) return 0;
  hipLaunchKernelGGL(pack_ghosts_kernel, dim3((3 * nsend + 255) / 256), dim3(256), 0, (hipStream_t)stream, x,
                     reinterpret_cast<const long long*>(owner), shift, nsend, out);
  return (int)hipGetLastError();
}

int ani_md_unpack_reverse(double* f, const int64_t* owner, int nsend, const double* in, void* stream) {
  if (nsend <= 0) return 0;
  hipLaunchKernelGGL(unpack_reverse_kernel, dim3((3 * nsend + 255) / 256), dim3(256), 0, (hipStream_t)stream, f,
                     reinterpret_cast<const long long*>(owner), nsend, in);
  return (int)hipGetLastError();
}

}  // extern "C"

int ani_md_wrap_positions(double* x, int n, const double* lo3, const double* len3, int periodic_mask, void* stream) {
  if (n <= 0 || !periodic_mask) return 0;
  hipLaunchKernelGGL(wrap_kernel, dim3((3 * n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, n, lo3[0], lo3[1], lo3[2], len3[0], len3[1],
                     len3[2], periodic_mask);
  return (int)hipGetLastError();
}

int ani_md_ghost_shell_count(const double* x, int n, const double* clo, const double* chi, int ncombo, int* blk_cnt, int* blk_off,
                             int* out_counts, void* stream) {
  if (ncombo <= 0) return 0;
  const int nblk = n > 0 ? (n + 255) / 256 : 1;
  hipLaunchKernelGGL(shell_count_kernel, dim3(nblk, ncombo), dim3(256), 0, (hipStream_t)stream, x, n, clo, chi, nblk, blk_cnt);
  hipLaunchKernelGGL(shell_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, blk_cnt, blk_off, ncombo, nblk, out_counts);
  return (int)hipGetLastError();
}

int ani_md_ghost_shell_fill(const double* x, int n, const double* clo, const double* chi, const double* cshift, int ncombo,
                            const int* blk_off, int64_t* send_idx, double* send_shift, void* stream) {
  if (ncombo <= 0) return 0;
  const int nblk = n > 0 ? (n + 255) / 256 : 1;
  hipLaunchKernelGGL(shell_fill_kernel, dim3(nblk, ncombo), dim3(256), 0, (hipStream_t)stream, x, n, clo, chi, cshift, nblk, blk_off,
                     reinterpret_cast<long long*>(send_idx), send_shift);
  return (int)hipGetLastError();
}

int ani_md_append_ghosts(double* x, int* species, int nlocal, const int64_t* owner, const double* shift, int nghost, void* stream) {
  if (nghost <= 0) return 0;
  hipLaunchKernelGGL(append_ghosts_kernel, dim3((nghost + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, species, nlocal,
                     reinterpret_cast<const long long*>(owner), shift, nghost);
  return (int)hipGetLastError();
}

int ani_md_check(double* d2max, const double* ev, double* out, void* stream) {
  hipLaunchKernelGGL(check_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, d2max, ev, out);
  return (int)hipGetLastError();
}
