// ani_kernels_aev.hip — AEV (radial + angular symmetry functions) forward and analytic backward over the
// LAMMPS full neighbour list.  Replaces torchani's cuaev / pyaev (not in the reference tree; call sites
// models/lammps_ani.py:277-296) and the autograd pass through it (models/lammps_ani.py:197-206).
//
// One wavefront (64 lanes) per centre atom, four centres per workgroup, no workgroup barriers: each wave owns a
// private LDS region holding the screened neighbour list of its centre and the centre's AEV row (forward) or
// dE/dAEV row + per-neighbour gradient accumulators (backward).
//
//   forward : list pairs are streamed with coalesced jlist loads, positions gathered as float4, screened and
//             compacted into LDS with ballot/popcount; radial terms are spread over (neighbour, shift) lanes,
//             angular terms over (j,k) pair lanes; both accumulate into the LDS row, which is then written once.
//   backward: the same compaction, then dE/dr per neighbour (radial) and dE/d(d_j), dE/d(d_k) per pair
//             (angular) accumulate into LDS per-neighbour vectors; one global float atomic per neighbour
//             component scatters the force, the centre gets minus the sum; the virial is reduced per wave.
//
// Formulas and their derivatives are the ones restated in oracle/ani_oracle.c.
#include "ani_kernels.h"

namespace ani {

constexpr int kWaves = 4;
constexpr int kAevMax = 1024;  // LDS floats reserved for one AEV row (ANI-2x: 1008)

struct WaveLds {
  float dx[kMaxRad], dy[kMaxRad], dz[kMaxRad], r[kMaxRad], fc[kMaxRad];
  int sp[kMaxRad], j[kMaxRad];
  int ang[kMaxAng];
  float fca[kMaxAng];
  float row[kAevMax];  // forward: AEV accumulators; backward: dE/dAEV of this centre
  float gd[3 * kMaxRad];  // backward only
};

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int lanes_below(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

// Screen + compact the neighbours of one centre into the wave's LDS region.  Returns counts through nrad/nang
// (clamped to capacity; *over is set when clamping happened).
__device__ __forceinline__ void compact_neighbours(const AevParams& p, const AevArgs& a, int ii, int lane, WaveLds& L,
                                                   int& nrad, int& nang, bool& over) {
  const int i = a.ilist[ii];
  const float4 xi = a.xyzs[i];
  const int beg = a.nbr_off[ii];
  const int n = a.numneigh[ii];
  nrad = 0;
  nang = 0;
  over = false;
  for (int base = 0; base < n; base += 64) {
    const int q = base + lane;
    const bool valid = q < n;
    const int j = valid ? a.jlist[beg + q] : i;
    const float4 xj = a.xyzs[j];
    const float dx = xj.x - xi.x, dy = xj.y - xi.y, dz = xj.z - xi.z;
    const float r = sqrtf(dx * dx + dy * dy + dz * dz);
    const bool in_r = valid && (p.compat || r <= p.Rcr);
    const bool in_a = valid && r <= p.Rca;
    const unsigned long long mr = __ballot(in_r);
    const unsigned long long ma = __ballot(in_a);
    const int pos = nrad + lanes_below(mr);
    if (in_r && pos < kMaxRad) {
      L.dx[pos] = dx; L.dy[pos] = dy; L.dz[pos] = dz; L.r[pos] = r;
      L.fc[pos] = 0.5f * cosf(r * p.pi_over_Rcr) + 0.5f;
      L.sp[pos] = __float_as_int(xj.w);
      L.j[pos] = j;
    }
    const int posa = nang + lanes_below(ma);
    if (in_a && posa < kMaxAng && pos < kMaxRad) {
      L.ang[posa] = pos;
      L.fca[posa] = 0.5f * cosf(r * p.pi_over_Rca) + 0.5f;
    }
    nrad += __popcll(mr);
    nang += __popcll(ma);
  }
  if (nrad > kMaxRad) { nrad = kMaxRad; over = true; }
  if (nang > kMaxAng) { nang = kMaxAng; over = true; }
}

// unordered pair index t -> (a, b), a < b < n, row-major over the strict upper triangle
__device__ __forceinline__ void decode_pair(int t, int n, int& a, int& b) {
  const float fn = (float)(2 * n - 1);
  int aa = (int)floorf((fn - sqrtf(fn * fn - 8.f * (float)t)) * 0.5f);
  if (aa < 0) aa = 0;
  if (aa > n - 2) aa = n - 2;
  // first pair index of row aa: aa*(2n-aa-1)/2
  while (aa > 0 && aa * (2 * n - aa - 1) / 2 > t) aa--;
  while ((aa + 1) * (2 * n - aa - 2) / 2 <= t) aa++;
  a = aa;
  b = aa + 1 + (t - aa * (2 * n - aa - 1) / 2);
}

__device__ __forceinline__ int triu_index(int s1, int s2, int S) {
  const int lo = s1 < s2 ? s1 : s2, hi = s1 < s2 ? s2 : s1;
  return lo * S - lo * (lo - 1) / 2 + (hi - lo);
}

__global__ __launch_bounds__(64 * kWaves) void aev_forward_kernel(AevParams p, AevArgs a) {
  __shared__ WaveLds lds[kWaves];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * kWaves + wave;
  if (row >= a.nrows) return;
  const int ii = a.centre_of_row[row];
  if (ii < 0) return;  // bucket padding: row stays zero (cleared at rebuild)
  WaveLds& L = lds[wave];

  for (int e = lane; e < p.aev_stride; e += 64) L.row[e] = 0.f;
  int nrad, nang;
  bool over;
  compact_neighbours(p, a, ii, lane, L, nrad, nang, over);
  if (over && lane == 0) atomicOr(a.err_flag, 1);
  wave_sync();

  // radial: items (neighbour q, shift k)
  const int nR = p.nR;
  for (int t = lane; t < nrad * nR; t += 64) {
    const int q = t / nR, k = t - q * nR;
    const float dr = L.r[q] - p.ShfR[k];
    const float v = 0.25f * expf(-p.EtaR * dr * dr) * L.fc[q];
    atomicAdd(&L.row[L.sp[q] * nR + k], v);
  }

  // angular: items = unordered pairs of the angular list
  const int npair = nang * (nang - 1) / 2;
  for (int t = lane; t < npair; t += 64) {
    int ia, ib;
    decode_pair(t, nang, ia, ib);
    const int qa = L.ang[ia], qb = L.ang[ib];
    const float ra = L.r[qa], rb = L.r[qb];
    const float dot = L.dx[qa] * L.dx[qb] + L.dy[qa] * L.dy[qb] + L.dz[qa] * L.dz[qb];
    const float c = 0.95f * dot / fmaxf(ra * rb, 1e-10f);
    const float s = sqrtf(fmaxf(1.f - c * c, 0.f));
    const float w = 2.f * L.fca[ia] * L.fca[ib];
    const float rho = 0.5f * (ra + rb);
    float* out = &L.row[p.radial_len + triu_index(L.sp[qa], L.sp[qb], p.S) * p.nAZ];
    float f1[kMaxShfZ];
#pragma unroll 4
    for (int z = 0; z < p.nZ; z++) {
      const float base = 0.5f * (1.f + c * p.cosZ[z] + s * p.sinZ[z]);
      f1[z] = w * powf(fmaxf(base, 0.f), p.Zeta);
    }
    for (int sa = 0; sa < p.nA; sa++) {
      const float dr = rho - p.ShfA[sa];
      const float f2 = expf(-p.EtaA * dr * dr);
      for (int z = 0; z < p.nZ; z++) atomicAdd(&out[sa * p.nZ + z], f1[z] * f2);
    }
  }
  wave_sync();
  float* dst = a.aev + (long long)row * p.aev_stride;
  for (int e = lane; e < p.aev_stride; e += 64) dst[e] = L.row[e];
}

__global__ __launch_bounds__(64 * kWaves) void aev_backward_kernel(AevParams p, AevArgs a) {
  __shared__ WaveLds lds[kWaves];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * kWaves + wave;
  if (row >= a.nrows) return;
  const int ii = a.centre_of_row[row];
  if (ii < 0) return;
  WaveLds& L = lds[wave];

  const float* g = a.gaev + (long long)row * p.aev_stride;
  for (int e = lane; e < p.aev_stride; e += 64) L.row[e] = g[e];
  int nrad, nang;
  bool over;
  compact_neighbours(p, a, ii, lane, L, nrad, nang, over);
  if (over && lane == 0) atomicOr(a.err_flag, 1);
  wave_sync();

  // radial: one lane per neighbour, dE/dr summed over shifts
  const int nR = p.nR;
  for (int q = lane; q < nrad; q += 64) {
    const float r = L.r[q], fc = L.fc[q];
    const float dfc = -0.5f * p.pi_over_Rcr * sinf(r * p.pi_over_Rcr);
    const float* gg = &L.row[L.sp[q] * nR];
    float dEdr = 0.f;
    for (int k = 0; k < nR; k++) {
      const float dr = r - p.ShfR[k];
      const float e = 0.25f * expf(-p.EtaR * dr * dr);
      dEdr += gg[k] * e * (dfc - 2.f * p.EtaR * dr * fc);
    }
    const float s = dEdr / r;
    L.gd[3 * q + 0] = s * L.dx[q];
    L.gd[3 * q + 1] = s * L.dy[q];
    L.gd[3 * q + 2] = s * L.dz[q];
  }
  wave_sync();

  // angular: one lane per unordered pair
  const int npair = nang * (nang - 1) / 2;
  for (int t = lane; t < npair; t += 64) {
    int ia, ib;
    decode_pair(t, nang, ia, ib);
    const int qa = L.ang[ia], qb = L.ang[ib];
    const float ra = L.r[qa], rb = L.r[qb];
    const float ax = L.dx[qa], ay = L.dy[qa], az = L.dz[qa];
    const float bx = L.dx[qb], by = L.dy[qb], bz = L.dz[qb];
    const float rr = ra * rb;
    const float cosv = (ax * bx + ay * by + az * bz) / rr;
    const float c = 0.95f * cosv;
    const float s = sqrtf(fmaxf(1.f - c * c, 1e-12f));
    const float fa = L.fca[ia], fb = L.fca[ib];
    const float dfa = -0.5f * p.pi_over_Rca * sinf(ra * p.pi_over_Rca);
    const float dfb = -0.5f * p.pi_over_Rca * sinf(rb * p.pi_over_Rca);
    const float P = fa * fb, rho = 0.5f * (ra + rb);
    const float* gg = &L.row[p.radial_len + triu_index(L.sp[qa], L.sp[qb], p.S) * p.nAZ];
    float f1[kMaxShfZ], df1[kMaxShfZ];
#pragma unroll 4
    for (int z = 0; z < p.nZ; z++) {
      // base = (1 + cos(theta - ShfZ))/2 ; sin(theta - ShfZ) = s cosZ - c sinZ
      const float base = fmaxf(0.5f * (1.f + c * p.cosZ[z] + s * p.sinZ[z]), 0.f);
      const float pm1 = powf(base, p.Zeta - 1.f);
      f1[z] = pm1 * base;
      df1[z] = p.Zeta * pm1 * 0.5f * (s * p.cosZ[z] - c * p.sinZ[z]) / s;
    }
    float A = 0.f, B = 0.f, C = 0.f;
    for (int sa = 0; sa < p.nA; sa++) {
      const float dr = rho - p.ShfA[sa];
      const float f2 = expf(-p.EtaA * dr * dr);
      const float df2 = -2.f * p.EtaA * dr * f2;
      for (int z = 0; z < p.nZ; z++) {
        const float gv = gg[sa * p.nZ + z];
        A = fmaf(gv * f2, df1[z], A);
        B = fmaf(gv * df2, f1[z], B);
        C = fmaf(gv * f2, f1[z], C);
      }
    }
    A *= 2.f * P * 0.95f;
    B *= 2.f * P * 0.5f;
    C *= 2.f;
    const float ca = A / rr, ia2 = A * cosv / (ra * ra), ib2 = A * cosv / (rb * rb);
    const float ta = (B + C * dfa * fb) / ra, tb = (B + C * fa * dfb) / rb;
    atomicAdd(&L.gd[3 * qa + 0], ca * bx + (ta - ia2) * ax);
    atomicAdd(&L.gd[3 * qa + 1], ca * by + (ta - ia2) * ay);
    atomicAdd(&L.gd[3 * qa + 2], ca * bz + (ta - ia2) * az);
    atomicAdd(&L.gd[3 * qb + 0], ca * ax + (tb - ib2) * bx);
    atomicAdd(&L.gd[3 * qb + 1], ca * ay + (tb - ib2) * by);
    atomicAdd(&L.gd[3 * qb + 2], ca * az + (tb - ib2) * bz);
  }
  wave_sync();

  // scatter: F_j -= gd_j ; F_i += sum_j gd_j ; virial -= gd (x) d
  float fx = 0.f, fy = 0.f, fz = 0.f;
  float v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int q = lane; q < nrad; q += 64) {
    const float gx = L.gd[3 * q], gy = L.gd[3 * q + 1], gz = L.gd[3 * q + 2];
    const int j = L.j[q];
    atomicAdd(&a.fbuf[3 * j + 0], -gx);
    atomicAdd(&a.fbuf[3 * j + 1], -gy);
    atomicAdd(&a.fbuf[3 * j + 2], -gz);
    fx += gx; fy += gy; fz += gz;
    if (a.virial) {
      const float dx = L.dx[q], dy = L.dy[q], dz = L.dz[q];
      v[0] += gx * dx; v[1] += gx * dy; v[2] += gx * dz;
      v[3] += gy * dx; v[4] += gy * dy; v[5] += gy * dz;
      v[6] += gz * dx; v[7] += gz * dy; v[8] += gz * dz;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    fx += __shfl_xor(fx, off);
    fy += __shfl_xor(fy, off);
    fz += __shfl_xor(fz, off);
  }
  const int i = a.ilist[ii];
  if (lane == 0) {
    atomicAdd(&a.fbuf[3 * i + 0], fx);
    atomicAdd(&a.fbuf[3 * i + 1], fy);
    atomicAdd(&a.fbuf[3 * i + 2], fz);
  }
  if (a.virial) {
#pragma unroll
    for (int k = 0; k < 9; k++) {
      float s = v[k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
      if (lane == 0) atomicAdd(&a.virial[k], -(double)s);
    }
  }
}

void launch_aev_forward(const AevParams& p, const AevArgs& a, hipStream_t st) {
  if (a.nrows <= 0) return;
  hipLaunchKernelGGL(aev_forward_kernel, dim3((a.nrows + kWaves - 1) / kWaves), dim3(64 * kWaves), 0, st, p, a);
}
void launch_aev_backward(const AevParams& p, const AevArgs& a, hipStream_t st) {
  if (a.nrows <= 0) return;
  hipLaunchKernelGGL(aev_backward_kernel, dim3((a.nrows + kWaves - 1) / kWaves), dim3(64 * kWaves), 0, st, p, a);
}

}  // namespace ani
