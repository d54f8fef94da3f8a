// ani_kernels_f64.hip — the `double` precision path of pair_style ani (reference: src/pair_ani.cpp:326-337 "single|double",
// src/ani_csrc/ani.cpp:52-61 module_to_dtype(model, kFloat64)).
//
// Correctness path, not a fast path: the reference uses fp64 for its golden vectors (tests/lammps-unittest/
// test_ani2x_nocuaev_double_half, src/ani_csrc/test_model.cpp:164 threshold 1e-8), and so do we — this path lets the
// GPU be checked against the fp64 oracle at 1e-8 instead of fp32 tolerances.  Same algorithm and data layout as the
// fp32 path (species-bucketed rows, columns of absent species pruned, half/full lists through the same list code);
// the kernels are the straightforward ones: one wave per centre with LDS double atomics for the AEV passes, and an
// LDS-tiled FMA GEMM (v_fma_f64; 64x64 tiles, 4x4 outputs per thread) for the MLP.  libm transcendentals, acos-free
// angle handling as in the fp32 kernels.
#include "ani_kernels.h"

namespace ani {

namespace {
constexpr int kW64 = 2;  // centres per workgroup

struct Wave64 {
  double dx[kMaxRad], dy[kMaxRad], dz[kMaxRad], r[kMaxRad], fc[kMaxRad];
  int sp[kMaxRad], j[kMaxRad];
  int ang[kMaxAng];
  double fca[kMaxAng];
  double row[1024];
  double gd[3 * kMaxRad];
};

__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ int below(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}
__device__ __forceinline__ void pair_of(int t, int n, int& a, int& b) {
  int aa = 0, start = 0;
  while (start + (n - 1 - aa) <= t) { start += n - 1 - aa; aa++; }
  a = aa;
  b = aa + 1 + (t - start);
}
__device__ __forceinline__ int triu64(int s1, int s2, int S) {
  const int lo = s1 < s2 ? s1 : s2, hi = s1 < s2 ? s2 : s1;
  return lo * S - lo * (lo - 1) / 2 + (hi - lo);
}

__device__ __forceinline__ void compact64(const Aev64Params& p, const Aev64Args& a, const int4 info, int lane, Wave64& L, int& nrad,
                                          int& nang, bool& over) {
  const int i = info.x, beg = info.y, n = info.z;
  const double xi = a.x[3 * i], yi = a.x[3 * i + 1], zi = a.x[3 * i + 2];
  nrad = 0;
  nang = 0;
  over = false;
  for (int base = 0; base < n; base += 64) {
    const int q = base + lane;
    const bool valid = q < n;
    const int j = valid ? a.jlist[beg + q] : i;
    const double dx = a.x[3 * j] - xi, dy = a.x[3 * j + 1] - yi, dz = a.x[3 * j + 2] - zi;
    const double r = sqrt(dx * dx + dy * dy + dz * dz);
    const bool in_r = valid && (p.compat || r <= p.Rcr);
    const bool in_a = valid && r <= p.Rca;
    const unsigned long long mr = __ballot(in_r), ma = __ballot(in_a);
    const int pos = nrad + below(mr);
    if (in_r && pos < kMaxRad) {
      L.dx[pos] = dx; L.dy[pos] = dy; L.dz[pos] = dz; L.r[pos] = r;
      L.fc[pos] = 0.5 * cos(r * (M_PI / p.Rcr)) + 0.5;
      L.sp[pos] = a.cmap.m[a.species[j]];
      L.j[pos] = j;
    }
    const int posa = nang + below(ma);
    if (in_a && posa < kMaxAng && pos < kMaxRad) {
      L.ang[posa] = pos;
      L.fca[posa] = 0.5 * cos(r * (M_PI / p.Rca)) + 0.5;
    }
    nrad += __popcll(mr);
    nang += __popcll(ma);
  }
  if (nrad > kMaxRad) { nrad = kMaxRad; over = true; }
  if (nang > kMaxAng) { nang = kMaxAng; over = true; }
}

__global__ __launch_bounds__(64 * kW64) void aev64_forward(Aev64Params p, Aev64Args a) {
  __shared__ Wave64 lds[kW64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * kW64 + wave;
  if (row >= a.nrows) return;
  const int4 info = a.row_info[row];
  if (info.x < 0) return;
  Wave64& L = lds[wave];
  for (int e = lane; e < p.aev_stride; e += 64) L.row[e] = 0.0;
  int nrad, nang;
  bool over;
  compact64(p, a, info, lane, L, nrad, nang, over);
  if (over && lane == 0) atomicOr(a.err_flag, 1);
  wsync();
  for (int t = lane; t < nrad * p.nR; t += 64) {
    const int q = t / p.nR, k = t - q * p.nR;
    const double dr = L.r[q] - p.ShfR[k];
    atomicAdd(&L.row[L.sp[q] * p.nR + k], 0.25 * exp(-p.EtaR * dr * dr) * L.fc[q]);
  }
  const int npair = nang * (nang - 1) / 2;
  for (int t = lane; t < npair; t += 64) {
    int ia, ib;
    pair_of(t, nang, ia, ib);
    const int qa = L.ang[ia], qb = L.ang[ib];
    const double ra = L.r[qa], rb = L.r[qb];
    const double dot = L.dx[qa] * L.dx[qb] + L.dy[qa] * L.dy[qb] + L.dz[qa] * L.dz[qb];
    const double c = 0.95 * dot / fmax(ra * rb, 1e-10);
    const double s = sqrt(fmax(1.0 - c * c, 0.0));
    const double w = 2.0 * L.fca[ia] * L.fca[ib];
    const double rho = 0.5 * (ra + rb);
    double* out = &L.row[p.radial_len + triu64(L.sp[qa], L.sp[qb], p.S) * p.nA * p.nZ];
    for (int sa = 0; sa < p.nA; sa++) {
      const double dr = rho - p.ShfA[sa];
      const double f2 = exp(-p.EtaA * dr * dr);
      for (int z = 0; z < p.nZ; z++) {
        const double base = 0.5 * (1.0 + c * p.cosZ[z] + s * p.sinZ[z]);
        atomicAdd(&out[sa * p.nZ + z], w * f2 * pow(fmax(base, 0.0), p.Zeta));
      }
    }
  }
  wsync();
  double* dst = a.aev + (long long)row * p.aev_stride;
  for (int e = lane; e < p.aev_stride; e += 64) dst[e] = L.row[e];
}

__global__ __launch_bounds__(64 * kW64) void aev64_backward(Aev64Params p, Aev64Args a) {
  __shared__ Wave64 lds[kW64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * kW64 + wave;
  if (row >= a.nrows) return;
  const int4 info = a.row_info[row];
  if (info.x < 0) return;
  Wave64& L = lds[wave];
  const double* g = a.gaev + (long long)row * p.aev_stride;
  for (int e = lane; e < p.aev_stride; e += 64) L.row[e] = g[e];
  int nrad, nang;
  bool over;
  compact64(p, a, info, lane, L, nrad, nang, over);
  if (over && lane == 0) atomicOr(a.err_flag, 1);
  wsync();
  for (int q = lane; q < nrad; q += 64) {
    const double r = L.r[q], fc = L.fc[q];
    const double dfc = -0.5 * (M_PI / p.Rcr) * sin(r * (M_PI / p.Rcr));
    const double* gg = &L.row[L.sp[q] * p.nR];
    double dEdr = 0.0;
    for (int k = 0; k < p.nR; k++) {
      const double dr = r - p.ShfR[k];
      const double e = 0.25 * exp(-p.EtaR * dr * dr);
      dEdr += gg[k] * e * (dfc - 2.0 * p.EtaR * dr * fc);
    }
    const double sc = dEdr / r;
    L.gd[3 * q] = sc * L.dx[q]; L.gd[3 * q + 1] = sc * L.dy[q]; L.gd[3 * q + 2] = sc * L.dz[q];
  }
  wsync();
  const int npair = nang * (nang - 1) / 2;
  for (int t = lane; t < npair; t += 64) {
    int ia, ib;
    pair_of(t, nang, ia, ib);
    const int qa = L.ang[ia], qb = L.ang[ib];
    const double ra = L.r[qa], rb = L.r[qb];
    const double ax = L.dx[qa], ay = L.dy[qa], az = L.dz[qa], bx = L.dx[qb], by = L.dy[qb], bz = L.dz[qb];
    const double rr = ra * rb;
    const double cosv = (ax * bx + ay * by + az * bz) / rr;
    const double c = 0.95 * cosv;
    const double s = sqrt(fmax(1.0 - c * c, 1e-300));
    const double fa = L.fca[ia], fb = L.fca[ib];
    const double dfa = -0.5 * (M_PI / p.Rca) * sin(ra * (M_PI / p.Rca));
    const double dfb = -0.5 * (M_PI / p.Rca) * sin(rb * (M_PI / p.Rca));
    const double P = fa * fb, rho = 0.5 * (ra + rb);
    const double* gg = &L.row[p.radial_len + triu64(L.sp[qa], L.sp[qb], p.S) * p.nA * p.nZ];
    double A = 0, B = 0, C = 0;
    for (int sa = 0; sa < p.nA; sa++) {
      const double dr = rho - p.ShfA[sa];
      const double f2 = exp(-p.EtaA * dr * dr);
      const double df2 = -2.0 * p.EtaA * dr * f2;
      for (int z = 0; z < p.nZ; z++) {
        const double base = fmax(0.5 * (1.0 + c * p.cosZ[z] + s * p.sinZ[z]), 0.0);
        const double pm1 = pow(base, p.Zeta - 1.0);
        const double f1 = pm1 * base;
        const double df1 = p.Zeta * pm1 * 0.5 * (s * p.cosZ[z] - c * p.sinZ[z]) / s;
        const double gv = gg[sa * p.nZ + z];
        A += gv * f2 * df1;
        B += gv * df2 * f1;
        C += gv * f2 * f1;
      }
    }
    A *= 2.0 * P * 0.95;
    B *= P;
    C *= 2.0;
    const double ca = A / rr;
    const double ta = (B + C * dfa * fb) / ra - A * cosv / (ra * ra);
    const double tb = (B + C * fa * dfb) / rb - A * cosv / (rb * rb);
    atomicAdd(&L.gd[3 * qa + 0], ca * bx + ta * ax);
    atomicAdd(&L.gd[3 * qa + 1], ca * by + ta * ay);
    atomicAdd(&L.gd[3 * qa + 2], ca * bz + ta * az);
    atomicAdd(&L.gd[3 * qb + 0], ca * ax + tb * bx);
    atomicAdd(&L.gd[3 * qb + 1], ca * ay + tb * by);
    atomicAdd(&L.gd[3 * qb + 2], ca * az + tb * bz);
  }
  wsync();
  double fx = 0, fy = 0, fz = 0, v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int q = lane; q < nrad; q += 64) {
    const double gx = L.gd[3 * q], gy = L.gd[3 * q + 1], gz = L.gd[3 * q + 2];
    const int j = L.j[q];
    atomicAdd(&a.fbuf[3 * j + 0], -gx);
    atomicAdd(&a.fbuf[3 * j + 1], -gy);
    atomicAdd(&a.fbuf[3 * j + 2], -gz);
    fx += gx; fy += gy; fz += gz;
    if (a.virial) {
      const double dx = L.dx[q], dy = L.dy[q], dz = L.dz[q];
      v[0] += gx * dx; v[1] += gx * dy; v[2] += gx * dz;
      v[3] += gy * dx; v[4] += gy * dy; v[5] += gy * dz;
      v[6] += gz * dx; v[7] += gz * dy; v[8] += gz * dz;
    }
  }
  for (int off = 32; off > 0; off >>= 1) { fx += __shfl_xor(fx, off); fy += __shfl_xor(fy, off); fz += __shfl_xor(fz, off); }
  if (lane == 0) {
    atomicAdd(&a.fbuf[3 * info.x + 0], fx);
    atomicAdd(&a.fbuf[3 * info.x + 1], fy);
    atomicAdd(&a.fbuf[3 * info.x + 2], fz);
  }
  if (a.virial) {
    for (int k = 0; k < 9; k++) {
      double sv = v[k];
      for (int off = 32; off > 0; off >>= 1) sv += __shfl_xor(sv, off);
      if (lane == 0) atomicAdd(&a.virial[k], -sv);
    }
  }
}

// C[rows][N] = epi(A[rows][K] * Bt[N][K]^T), 64x64 tile, 256 threads, 4x4 outputs per thread, K slabs of 16
__device__ __forceinline__ double celu64(double z, double alpha) { return z > 0 ? z : alpha * expm1(z / alpha); }
__device__ __forceinline__ double dcelu64(double h, double alpha) { return h > 0 ? 1.0 : h / alpha + 1.0; }

template <int EPI>
__global__ __launch_bounds__(256) void gemm64_kernel(Gemm64Args g) {
  __shared__ double As[16][65], Bs[16][65];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int b = blockIdx.z;
  const int row0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  const double* A = g.A + (long long)b * g.sA;
  const double* Bt = g.Bt + (long long)b * g.sB;
  double acc[4][4] = {};
  for (int k0 = 0; k0 < g.K; k0 += 16) {
    for (int e = threadIdx.x; e < 64 * 16; e += 256) {
      const int r = e >> 4, k = e & 15;
      As[k][r] = (row0 + r < g.rows && k0 + k < g.K) ? A[(long long)(row0 + r) * g.lda + k0 + k] : 0.0;
      Bs[k][r] = (n0 + r < g.N && k0 + k < g.K) ? Bt[(long long)(n0 + r) * g.ldb + k0 + k] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) {
      double av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; i++) { av[i] = As[k][ty * 4 + i]; bv[i] = Bs[k][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = fma(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
  double* C = g.C + (long long)b * g.sC;
  for (int i = 0; i < 4; i++) {
    const int m = row0 + ty * 4 + i;
    if (m >= g.rows) continue;
    double esum = 0.0;
    const double valid = (EPI == EPI_LAST) ? (g.centre_of_row[m] >= 0 ? g.scale : 0.0) : 1.0;
    for (int j = 0; j < 4; j++) {
      const int n = n0 + tx * 4 + j;
      if (n >= g.N) continue;
      double v = acc[i][j];
      if (EPI == EPI_PLAIN) {
        C[(long long)m * g.ldc + n] = v;
      } else if (EPI == EPI_CELU) {
        C[(long long)m * g.ldc + n] = celu64(v + g.bias[(long long)b * g.sBias + n], g.alpha);
      } else if (EPI == EPI_BWD) {
        const double h = g.aux[(long long)b * g.sAux + (long long)m * g.ldaux + n];
        C[(long long)m * g.ldc + n] = v * dcelu64(h, g.alpha);
      } else {
        const double h = celu64(v + g.bias[(long long)b * g.sBias + n], g.alpha);
        const double wv = g.aux[(long long)b * g.sAux + n];
        esum += h * wv;
        C[(long long)m * g.ldc + n] = valid * wv * dcelu64(h, g.alpha);
      }
    }
    if (EPI == EPI_LAST) {
      // e_out is zeroed before the launch; the output-layer bias is added once (by the first column block)
      if (blockIdx.y == 0 && tx == 0) esum += g.bias_last[b];
      atomicAdd(&g.e_out[(long long)b * g.sE + m], valid * esum);
    }
  }
}

__global__ void finish64_energy(const double* e_rows, int M, int nrows, const int* centre_of_row, const int* ilist, const int* species,
                                Sae64 sae, double* eatom, double* ev) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < nrows; row += gridDim.x * blockDim.x) {
    const int ii = centre_of_row[row];
    if (ii < 0) continue;
    double e = 0.0;
    for (int m = 0; m < M; m++) e += e_rows[(long long)m * nrows + row];
    const double ea = e + sae.v[species[ilist[ii]]];
    if (eatom) eatom[ii] = ea * 627.5094738898777;
    acc += ea;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(&ev[0], red[0] * 627.5094738898777);
}

__global__ void finish64_rest(const double* fbuf, int n3, double* f_out, int accumulate, const double* vir, double* ev, const int* err_flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n3) {
    const double v = fbuf[i] * 627.5094738898777;
    f_out[i] = accumulate ? f_out[i] + v : v;
  }
  if (i < 9) {
    const int k = i / 3, l = i % 3;
    ev[1 + i] = vir ? 0.5 * (vir[3 * k + l] + vir[3 * l + k]) * 627.5094738898777 : 0.0;
  }
  if (i == 0 && err_flag && *err_flag) ev[0] = __longlong_as_double(0x7ff8000000000000LL);
}
}  // namespace

__global__ void cvt64_kernel(const float* __restrict__ src, double* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (double)src[i];
}
void launch_cvt_f32_f64(const float* src, double* dst, size_t n, hipStream_t st) {
  if (n == 0) return;
  hipLaunchKernelGGL(cvt64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, n);
}

void launch_aev64_forward(const Aev64Params& p, const Aev64Args& a, hipStream_t st) {
  if (a.nrows <= 0) return;
  hipLaunchKernelGGL(aev64_forward, dim3((a.nrows + kW64 - 1) / kW64), dim3(64 * kW64), 0, st, p, a);
}
void launch_aev64_backward(const Aev64Params& p, const Aev64Args& a, hipStream_t st) {
  if (a.nrows <= 0) return;
  hipLaunchKernelGGL(aev64_backward, dim3((a.nrows + kW64 - 1) / kW64), dim3(64 * kW64), 0, st, p, a);
}
void launch_gemm64(const Gemm64Args& g, Epilogue epi, hipStream_t st) {
  if (g.rows <= 0 || g.N <= 0) return;
  const dim3 grid((g.rows + 63) / 64, (g.N + 63) / 64, g.batch), block(256);
  switch (epi) {
    case EPI_PLAIN: hipLaunchKernelGGL(gemm64_kernel<EPI_PLAIN>, grid, block, 0, st, g); break;
    case EPI_CELU: hipLaunchKernelGGL(gemm64_kernel<EPI_CELU>, grid, block, 0, st, g); break;
    case EPI_LAST: hipLaunchKernelGGL(gemm64_kernel<EPI_LAST>, grid, block, 0, st, g); break;
    case EPI_BWD: hipLaunchKernelGGL(gemm64_kernel<EPI_BWD>, grid, block, 0, st, g); break;
  }
}
void launch_finish64(const double* e_rows, int M, int nrows, const int* centre_of_row, const int* ilist, const int* species,
                     const Sae64& sae, const double* fbuf, int ntotal, const double* vir, double* f_out, int accumulate, double* ev,
                     double* eatom, const int* err_flag, hipStream_t st) {
  note_launch_error(hipMemsetAsync(ev, 0, sizeof(double) * 10, st));
  hipLaunchKernelGGL(finish64_energy, dim3(64), dim3(256), 0, st, e_rows, M, nrows, centre_of_row, ilist, species, sae, eatom, ev);
  const int n3 = ntotal * 3 > 9 ? ntotal * 3 : 9;
  hipLaunchKernelGGL(finish64_rest, dim3((n3 + 255) / 256), dim3(256), 0, st, fbuf, ntotal * 3, f_out, accumulate, vir, ev, err_flag);
}

}  // namespace ani
