// ani_kernels_rep.hip — optional pairwise repulsion.
//
// The reference attaches torchani's RepulsionXTB(cutoff = 5.1, cutoff_fn = "smooth") to reactive models
// (models/ani_models.py:50-53) and adds its energy in LammpsANI.forward (models/lammps_ani.py:186-193,300-330).
// torchani itself is not in the reference tree; the functional form below is restated from it [RECALL] and is the one
// oracle/ani_oracle.c:rep_pair follows:
//     d = r in Bohr;   e(r) = y_ab / d * exp(-sqrt_alpha_ab * d^k_ab) * fc(r),   fc(r) = exp(1 - 1 / (1 - (r/Rc)^2)), r < Rc
// A rank's energy share: HALF of e per (centre, neighbour) entry of the full list — a pair of two owned atoms is met
// from both ends, an owned-ghost pair from one end only, which is what the ghost_flags weighting of
// compute_from_neighbors amounts to.  Forces go to owned and ghost atoms like every other term of the pair style.
//
// One wave per centre over its (species-sorted) list segment, pair arithmetic in fp64 (5.5 M pairs per step at 100 k
// atoms: nothing next to the AEV passes), accumulation into the same force / virial accumulators as the AEV backward.
#include "ani_kernels.h"

namespace ani {

namespace {

constexpr double kAng2Bohr = 1.8897261258369282;

// F = float: force accumulators float4 {fx,fy,fz,-};  F = double: double[3].  Positions are the caller's fp64 array.
template <typename F>
__global__ __launch_bounds__(256) void repulsion_kernel(RepArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= a.nrows) return;
  const int4 info = a.row_info[row];
  if (info.x < 0) return;
  const int i = info.x, si = a.species[i];
  // positions in fp64 in both precisions: the wall is steep (d2e/dr2 ~ 1e3 kcal/mol/A^2 at 1 A), a float position in a
  // 100 A box (ulp 4e-6 A) would alone cost several 1e-3 kcal/mol/A of force
  const double xi[3] = {a.pos[3 * (long long)i], a.pos[3 * (long long)i + 1], a.pos[3 * (long long)i + 2]};
  const int S = a.S;
  const double* ty = a.tables + (long long)si * S;           // y_ab row of the centre species
  const double* tsa = a.tables + (long long)S * S + (long long)si * S;
  const double* tk = a.tables + 2LL * S * S + (long long)si * S;
  double er = 0.0, fi[3] = {0.0, 0.0, 0.0}, v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int q = lane; q < info.z; q += 64) {
    const int j = a.jlist[info.y + q];
    const double* pj = a.pos + 3 * (long long)j;
    const double d[3] = {pj[0] - xi[0], pj[1] - xi[1], pj[2] - xi[2]};
    const double r = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (r >= a.cutoff || r <= 0.0) continue;
    const int sj = a.species[j];
    double e_half, sc;   // half the pair energy (Hartree); dE/d(d_k) = sc * d_k (Hartree/Angstrom^2 * Angstrom)
    if constexpr (sizeof(F) == 4) {
      // precision 'single': r comes from the fp64 positions, the pair function itself in fp32 with the hardware
      // exp2/log2 (the fp64 libm pow/exp made this kernel cost more than the whole AEV backward)
      const float rf = (float)r, rc = (float)a.cutoff;
      const float x = rf / rc, den = 1.f - x * x;
      if (den <= 1e-10f) continue;
      const float fc = __builtin_amdgcn_exp2f((1.f - 1.f / den) * 1.4426950408889634f);
      const float dfc = fc * (-(2.f * x / rc) / (den * den));
      const float db = rf * (float)kAng2Bohr, al = (float)tsa[sj], kk = (float)tk[sj];
      const float pk1 = __builtin_amdgcn_exp2f((kk - 1.f) * __builtin_amdgcn_logf(db));   // db^(k-1)
      const float g = (float)ty[sj] / db * __builtin_amdgcn_exp2f(-al * pk1 * db * 1.4426950408889634f);
      const float dg = (float)kAng2Bohr * g * (-1.f / db - al * kk * pk1);
      e_half = 0.5 * (double)(g * fc);
      sc = 0.5 * (double)(dg * fc + g * dfc) / r;
    } else {
      const double x = r / a.cutoff, den = 1.0 - x * x;
      if (den <= 1e-10) continue;
      const double fc = exp(1.0 - 1.0 / den), dfc = fc * (-(2.0 * x / a.cutoff) / (den * den));
      const double db = r * kAng2Bohr, al = tsa[sj], kk = tk[sj];
      const double g = ty[sj] / db * exp(-al * pow(db, kk));
      const double dg = kAng2Bohr * g * (-1.0 / db - al * kk * pow(db, kk - 1.0));
      e_half = 0.5 * g * fc;
      sc = 0.5 * (dg * fc + g * dfc) / r;
    }
    er += e_half;
    double gk[3] = {sc * d[0], sc * d[1], sc * d[2]};
    if constexpr (sizeof(F) == 4) {
      float* fb = reinterpret_cast<float*>(a.fbuf) + 4 * (long long)j;
      atomicAdd(fb + 0, (float)-gk[0]); atomicAdd(fb + 1, (float)-gk[1]); atomicAdd(fb + 2, (float)-gk[2]);
    } else {
      double* fb = reinterpret_cast<double*>(a.fbuf) + 3 * (long long)j;
      atomicAdd(fb + 0, -gk[0]); atomicAdd(fb + 1, -gk[1]); atomicAdd(fb + 2, -gk[2]);
    }
    fi[0] += gk[0]; fi[1] += gk[1]; fi[2] += gk[2];
    if (a.virial) {
#pragma unroll
      for (int k = 0; k < 3; k++)
#pragma unroll
        for (int l = 0; l < 3; l++) v[3 * k + l] += gk[k] * d[l];
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    er += __shfl_xor(er, off);
#pragma unroll
    for (int k = 0; k < 3; k++) fi[k] += __shfl_xor(fi[k], off);
  }
  if (a.virial) {
#pragma unroll
    for (int k = 0; k < 9; k++)
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_xor(v[k], off);
  }
  if (lane == 0) {
    const int slot = row & (a.nslots - 1);
    atomicAdd(&a.erep[slot], er);
    if constexpr (sizeof(F) == 4) {
      float* fb = reinterpret_cast<float*>(a.fbuf) + 4 * (long long)i;
      atomicAdd(fb + 0, (float)fi[0]); atomicAdd(fb + 1, (float)fi[1]); atomicAdd(fb + 2, (float)fi[2]);
    } else {
      double* fb = reinterpret_cast<double*>(a.fbuf) + 3 * (long long)i;
      atomicAdd(fb + 0, fi[0]); atomicAdd(fb + 1, fi[1]); atomicAdd(fb + 2, fi[2]);
    }
    if (a.virial)
      for (int k = 0; k < 9; k++) atomicAdd(&a.virial[9 * (row & (a.vslots - 1)) + k], -v[k]);
  }
}

// ev[0] += sum(erep) in kcal/mol; launched after the finish kernels of either precision
__global__ void repulsion_energy_kernel(const double* __restrict__ erep, int nslots, double* __restrict__ ev) {
  __shared__ double red[256];
  double t = 0.0;
  for (int s = threadIdx.x; s < nslots; s += blockDim.x) t += erep[s];
  red[threadIdx.x] = t;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) ev[0] += red[0] * 627.5094738898777;
}

}  // namespace

void launch_repulsion(const RepArgs& a, bool fp64, hipStream_t st) {
  if (a.nrows <= 0) return;
  const dim3 grid((a.nrows + 3) / 4), block(256);
  if (fp64) hipLaunchKernelGGL(repulsion_kernel<double>, grid, block, 0, st, a);
  else hipLaunchKernelGGL(repulsion_kernel<float>, grid, block, 0, st, a);
}

void launch_repulsion_energy(const double* erep, int nslots, double* d_ev, hipStream_t st) {
  hipLaunchKernelGGL(repulsion_energy_kernel, dim3(1), dim3(256), 0, st, erep, nslots, d_ev);
}

}  // namespace ani
