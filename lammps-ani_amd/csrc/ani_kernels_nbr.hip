// ani_kernels_nbr.hip — device-side construction of the LAMMPS-style full neighbour list.
//
// What LAMMPS core does on the host for `neighbor 2.0 bin` + a full list request (src/pair_ani.cpp:219-223 asks for
// NeighConst::REQ_FULL): bin the nlocal + nghost atoms of the rank into cells no smaller than the neighbour cutoff
// (force cutoff + skin), then for every owned atom collect the atoms of the 27 surrounding cells closer than that
// cutoff.  Ghost atoms carry the periodic images, so there is no wrapping here, exactly as in LAMMPS' binned builds.
//
// Layout: atoms are counting-sorted by cell (`order`, ascending atom index inside a cell, so the list is
// deterministic) and their positions copied in that order (`xs`), which makes the three x-adjacent cells of a
// (y,z) column one contiguous range.  Sixteen lanes per atom in sorted order walk each range together.  Two passes
// (count, fill) around one scan; the list is HBM-bound integer work (4 B per pair written once).
#include "ani_kernels.h"
#include "ani_scan.h"

namespace ani {

namespace {

__device__ __forceinline__ int clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

__device__ __forceinline__ void cell_coords(const NbrGrid& g, double x, double y, double z, int& cx, int& cy, int& cz) {
  // atoms outside [lo,hi) are clamped into the boundary cells: two atoms closer than one cell edge still land in
  // cells whose (clamped) indices differ by at most one, so the 27-cell search stays exact
  cx = clampi((int)floor((x - g.lo[0]) * g.inv[0]), g.nc[0] - 1);
  cy = clampi((int)floor((y - g.lo[1]) * g.inv[1]), g.nc[1] - 1);
  cz = clampi((int)floor((z - g.lo[2]) * g.inv[2]), g.nc[2] - 1);
}

__global__ void nbr_bin_count_kernel(const double* __restrict__ x, int ntotal, NbrGrid g, int* __restrict__ cell_id,
                                     int* __restrict__ cell_count) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= ntotal) return;
  int cx, cy, cz;
  cell_coords(g, x[3 * a], x[3 * a + 1], x[3 * a + 2], cx, cy, cz);
  const int c = (cz * g.nc[1] + cy) * g.nc[0] + cx;
  cell_id[a] = c;
  atomicAdd(&cell_count[c], 1);
}

// out[0..n] = exclusive scan of in[0..n), single block
__global__ __launch_bounds__(1024) void nbr_scan_kernel(const int* __restrict__ in, int* __restrict__ out, int n) {
  __shared__ int wave_sums[16];
  const int total = block_scan_rounds<8>(n, wave_sums, [&](int i) { return in[i]; }, [&](int i, int ex, int) { out[i] = ex; });
  if (threadIdx.x == 0) out[n] = total;
}

// the same scan over many blocks, for arrays with one entry per atom: chunk-relative prefixes and chunk totals, then
// the totals of the chunks before are added (one block scanning 100 000 entries round after round took ~0.1 ms)
constexpr int kScanVpt = 4, kScanChunk = 1024 * kScanVpt;

__global__ __launch_bounds__(1024) void scan_chunk_kernel(const int* __restrict__ in, int* __restrict__ out, int n,
                                                           int* __restrict__ chunk_tot) {
  __shared__ int wave_sums[16];
  const int i0 = blockIdx.x * kScanChunk + threadIdx.x * kScanVpt;
  int v[kScanVpt], s = 0;
#pragma unroll
  for (int k = 0; k < kScanVpt; k++) {
    v[k] = i0 + k < n ? in[i0 + k] : 0;
    s += v[k];
  }
  int total;
  int ex = block_exclusive_scan(s, total, wave_sums);
#pragma unroll
  for (int k = 0; k < kScanVpt; k++) {
    if (i0 + k < n) out[i0 + k] = ex;
    ex += v[k];
  }
  if (threadIdx.x == 0) chunk_tot[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void scan_add_kernel(int* __restrict__ out, int n, const int* __restrict__ chunk_tot, int nchunks) {
  __shared__ int before;
  const int chunk = (blockIdx.x * blockDim.x) / kScanChunk;   // kScanChunk is a multiple of the block size
  if (threadIdx.x < 64) {
    int acc = 0;
    for (int c = threadIdx.x; c < chunk; c += 64) acc += chunk_tot[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (threadIdx.x == 0) before = acc;
  }
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] += before;
  else if (i == n) {
    int acc = 0;
    for (int c = 0; c < nchunks; c++) acc += chunk_tot[c];
    out[n] = acc;
  }
}

__global__ void nbr_bin_fill_kernel(const int* __restrict__ cell_id, const int* __restrict__ cell_start,
                                    int* __restrict__ cursor, int* __restrict__ order, int ntotal) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= ntotal) return;
  const int c = cell_id[a];
  order[cell_start[c] + atomicAdd(&cursor[c], 1)] = a;
}

// ascending atom index inside every cell (the atomics above fill in arbitrary order).  One wave per cell: every lane
// holds up to four entries and ranks them by counting the smaller ones (atom indices are distinct), then writes each
// entry to its rank -- all reads of the cell precede the writes in the wave's program order, so this is in place.
__global__ __launch_bounds__(256) void nbr_bin_sort_kernel(const int* __restrict__ cell_start, int* __restrict__ order, int ncell) {
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (c >= ncell) return;
  const int beg = cell_start[c], n = cell_start[c + 1] - beg;
  if (n <= 1) return;
  if (n > 256) {   // denser than any physical system at these cell sizes: serial insertion sort
    if (lane == 0)
      for (int p = beg + 1; p < beg + n; p++) {
        const int v = order[p];
        int q = p - 1;
        while (q >= beg && order[q] > v) { order[q + 1] = order[q]; q--; }
        order[q + 1] = v;
      }
    return;
  }
  int v[4], rk[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int idx = lane + 64 * k;
    v[k] = idx < n ? order[beg + idx] : 0x7fffffff;
    rk[k] = 0;
  }
  for (int q = 0; q < n; q++) {
    const int e = order[beg + q];
#pragma unroll
    for (int k = 0; k < 4; k++) rk[k] += e < v[k] ? 1 : 0;
  }
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (lane + 64 * k < n) order[beg + rk[k]] = v[k];
}

__global__ void nbr_gather_kernel(const double* __restrict__ x, const int* __restrict__ order, int ntotal,
                                  double* __restrict__ xs) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= ntotal) return;
  const int a = order[p];
  xs[3 * p] = x[3 * a]; xs[3 * p + 1] = x[3 * a + 1]; xs[3 * p + 2] = x[3 * a + 2];
}

// the same, plus what the one-kernel build (nbr_search_sorted_kernel) reads per candidate in ONE 16-byte load: the position in
// fp32 relative to the grid origin and, in the fourth word, the atom index with the species in the four bits above kIdxBits
constexpr int kIdxBits = 28;
constexpr unsigned kIdxMask = (1u << kIdxBits) - 1u;

__global__ void nbr_gather_packed_kernel(const double* __restrict__ x, const int* __restrict__ order, const int* __restrict__ species,
                                         int ntotal, NbrGrid g, double* __restrict__ xs, float4* __restrict__ xq) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= ntotal) return;
  const int a = order[p];
  const double px = x[3 * a], py = x[3 * a + 1], pz = x[3 * a + 2];
  xs[3 * p] = px; xs[3 * p + 1] = py; xs[3 * p + 2] = pz;
  const unsigned w = (unsigned)a | (((unsigned)species[a] & 15u) << kIdxBits);
  xq[p] = make_float4((float)(px - g.lo[0]), (float)(py - g.lo[1]), (float)(pz - g.lo[2]), __uint_as_float(w));
}

// kSearchLanes lanes per atom: the candidates of a cell range are taken kSearchLanes at a time (contiguous positions:
// coalesced), hits are placed by a ballot over the group -- the order of a list is the serial order (cell ranges, then
// sorted position), whatever the lane count.  One lane per atom left the chip latency-bound (2 waves per SIMD, a
// dependent L2 load per candidate).
constexpr int kSearchLanes = 16;

// MODE 0: count (numneigh);  1: fill dense segments at nbr_off;  2: ONE pass -- count and fill rows of a fixed capacity `cap`
// (entry k of centre i at jlist[i * cap + k]; entries beyond cap are counted, not written, and *ovf is set)
template <int MODE>
__global__ __launch_bounds__(256) void nbr_search_kernel(const double* __restrict__ xs, const int* __restrict__ order,
                                                          const int* __restrict__ cell_start, NbrGrid g, int nlocal,
                                                          int ntotal, double cut2, int* __restrict__ numneigh,
                                                          const int* __restrict__ nbr_off, int* __restrict__ jlist, int cap,
                                                          int* __restrict__ ovf) {
  constexpr int L = kSearchLanes;
  constexpr bool FILL = MODE != 0;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int p = (int)(t / L), l = threadIdx.x & (L - 1);
  const int grp_shift = (threadIdx.x & 63) & ~(L - 1);   // where this group's bits sit in the wave ballot
  if (p >= ntotal) return;
  const int i = order[p];
  if (i >= nlocal) return;  // ghosts are neighbours only
  const double xi = xs[3 * p], yi = xs[3 * p + 1], zi = xs[3 * p + 2];
  int cx, cy, cz;
  cell_coords(g, xi, yi, zi, cx, cy, cz);
  const int R = g.reach;
  const int x0 = cx - R > 0 ? cx - R : 0, x1 = cx + R < g.nc[0] - 1 ? cx + R : g.nc[0] - 1;
  int n = 0;
  int* out = MODE == 1 ? jlist + nbr_off[i] : (MODE == 2 ? jlist + (size_t)i * cap : nullptr);
  for (int dz = -R; dz <= R; dz++) {
    const int z = cz + dz;
    if (z < 0 || z >= g.nc[2]) continue;
    for (int dy = -R; dy <= R; dy++) {
      const int y = cy + dy;
      if (y < 0 || y >= g.nc[1]) continue;
      const int rowc = (z * g.nc[1] + y) * g.nc[0];
      const int beg = cell_start[rowc + x0], end = cell_start[rowc + x1 + 1];
      for (int q0 = beg; q0 < end; q0 += L) {
        const int q = q0 + l;
        bool hit = false;
        if (q < end) {
          const double ddx = xs[3 * q] - xi, ddy = xs[3 * q + 1] - yi, ddz = xs[3 * q + 2] - zi;
          const double r2 = ddx * ddx + ddy * ddy + ddz * ddz;
          hit = r2 <= cut2 && q != p;  // rsq <= cutneighsq, as LAMMPS' npair full/bin
        }
        const unsigned bits = (unsigned)(__ballot(hit) >> grp_shift) & ((1u << L) - 1u);
        if (FILL && hit) {
          const int k = n + __popc(bits & ((1u << l) - 1u));
          if (MODE == 1 || k < cap) out[k] = order[q];
        }
        n += __popc(bits);
      }
    }
  }
  if (MODE != 1 && l == 0) {
    numneigh[i] = n;
    if (MODE == 2 && n > cap) *ovf = 1;
  }
}

// The list build in ONE kernel: search, and every centre's entries grouped by neighbour species (what sort_jlist_kernel does to a
// caller's list), written to rows of `cap` entries at jrows[i * cap].  Sixteen lanes per centre as above.  Per candidate one
// 16-byte load (nbr_gather_packed_kernel); the cutoff test runs in fp32 on positions relative to the grid origin and falls back
// to the fp64 test of the kernels above -- rsq <= cutneighsq on the caller's doubles, as LAMMPS' npair full/bin -- whenever the
// fp32 value lies within `band` of the cutoff, band being twice the worst rounding error of the fp32 value: the list is the
// same set.  Hits are parked in LDS (one row per centre) in the serial order of the kernels above, counted per species, and
// written out species by species, order kept inside a species: the result equals sort_jlist_kernel applied to the rows of
// nbr_search_kernel<2>.  The cell ranges ((2R+1)^2 runs of 2R+1 x-adjacent cells) are fetched by the group's lanes side by
// side and handed round with a shuffle instead of one dependent load pair per run.  R = g.reach: 1 for cells no smaller than
// the cutoff, 2 for cells of half that edge (fewer candidates per hit: 0.59 of the volume).
template <int R>
__global__ __launch_bounds__(256) void nbr_search_sorted_kernel(const float4* __restrict__ xq, const double* __restrict__ xs,
                                                                 const int* __restrict__ cell_start, NbrGrid g, int nlocal, int ntotal,
                                                                 double cut2, float cutf, int S, int cap, int* __restrict__ numneigh,
                                                                 int* __restrict__ ilist, int* __restrict__ jrows, int* __restrict__ ovf) {
  extern __shared__ int rows_lds[];   // [16 centres][cap]
  constexpr int L = kSearchLanes, W = 2 * R + 1, NRUN = W * W, RPL = (NRUN + L - 1) / L;
  static_assert(L == 16, "group ballots below assume 16 lanes per centre");
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int p = (int)(t / L), l = threadIdx.x & (L - 1);
  const int grp_shift = (threadIdx.x & 63) & ~(L - 1);
  if (p >= ntotal) return;
  const float4 me = xq[p];
  const int i = (int)(__float_as_uint(me.w) & kIdxMask);
  if (i >= nlocal) return;  // ghosts are neighbours only
  const double xi = xs[3 * p], yi = xs[3 * p + 1], zi = xs[3 * p + 2];
  int cx, cy, cz;
  cell_coords(g, xi, yi, zi, cx, cy, cz);
  const int x0 = cx - R > 0 ? cx - R : 0, x1 = cx + R < g.nc[0] - 1 ? cx + R : g.nc[0] - 1;
  int rb[RPL], re[RPL];
#pragma unroll
  for (int k = 0; k < RPL; k++) {
    const int r = l + L * k;
    rb[k] = re[k] = 0;
    const int z = cz + r / W - R, y = cy + r % W - R;
    if (r < NRUN && z >= 0 && z < g.nc[2] && y >= 0 && y < g.nc[1]) {
      const int rowc = (z * g.nc[1] + y) * g.nc[0];
      rb[k] = cell_start[rowc + x0];
      re[k] = cell_start[rowc + x1 + 1];
    }
  }
  // |fp32 rsq - rsq| <= 2 (|dx|+|dy|+|dz|) M 2^-23 + rsq 2^-21 with M the largest coordinate in play: 3.5 cut M 2^-23 near the cutoff
  const float cut2f = cutf * cutf;
  const float M = fmaxf(fmaxf(fabsf(me.x), fabsf(me.y)), fabsf(me.z)) + 2.f * cutf;
  const float band = 8.f * cutf * M * 1.1920929e-7f + 1e-6f * cut2f;
  const float c_in = cut2f - band, c_out = cut2f + band;
  int* row = rows_lds + (threadIdx.x >> 4) * cap;
  int n = 0;
#pragma unroll
  for (int k = 0; k < RPL; k++) {
    const int nrun = NRUN - L * k < L ? NRUN - L * k : L;
    for (int r = 0; r < nrun; r++) {
      const int beg = __shfl(rb[k], grp_shift + r), end = __shfl(re[k], grp_shift + r);
      for (int q0 = beg; q0 < end; q0 += L) {
        const int q = q0 + l;
        bool hit = false;
        unsigned e = 0;
        if (q < end) {
          const float4 c = xq[q];
          const float dx = c.x - me.x, dy = c.y - me.y, dz = c.z - me.z;
          const float r2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
          hit = r2 < c_in;
          if (!hit && r2 <= c_out) {
            const double ddx = xs[3 * q] - xi, ddy = xs[3 * q + 1] - yi, ddz = xs[3 * q + 2] - zi;
            hit = ddx * ddx + ddy * ddy + ddz * ddz <= cut2;
          }
          hit = hit && q != p;
          e = __float_as_uint(c.w);
        }
        const unsigned bits = (unsigned)(__ballot(hit) >> grp_shift) & 0xffffu;
        if (hit) {
          const int kk = n + __popc(bits & ((1u << l) - 1u));
          if (kk < cap) row[kk] = (int)e;
        }
        n += __popc(bits);
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int nn = n < cap ? n : cap;
  int cnt[kMaxSpecies];
#pragma unroll
  for (int s = 0; s < kMaxSpecies; s++) cnt[s] = 0;
  for (int k0 = 0; k0 < nn; k0 += L) {
    const int sp = k0 + l < nn ? (int)((unsigned)row[k0 + l] >> kIdxBits) : -1;
#pragma unroll
    for (int s = 0; s < kMaxSpecies; s++)
      if (s < S) cnt[s] += __popc((unsigned)(__ballot(sp == s) >> grp_shift) & 0xffffu);
  }
  int acc = 0;
#pragma unroll
  for (int s = 0; s < kMaxSpecies; s++) {
    const int c = cnt[s];
    cnt[s] = acc;   // from here on: where the next entry of species s goes
    acc += c;
  }
  int* out = jrows + (size_t)i * cap;
  for (int k0 = 0; k0 < nn; k0 += L) {
    const unsigned e = k0 + l < nn ? (unsigned)row[k0 + l] : 0u;
    const int sp = k0 + l < nn ? (int)(e >> kIdxBits) : -1;
#pragma unroll
    for (int s = 0; s < kMaxSpecies; s++)
      if (s < S) {
        const unsigned b = (unsigned)(__ballot(sp == s) >> grp_shift) & 0xffffu;
        if (sp == s) out[cnt[s] + __popc(b & ((1u << l) - 1u))] = (int)(e & kIdxMask);
        cnt[s] += __popc(b);
      }
  }
  if (l == 0) {
    numneigh[i] = n;
    ilist[i] = i;
    if (n > cap) *ovf = 1;
  }
}

__global__ void iota_kernel(int* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = i;
}

}  // namespace

void launch_nbr_bin(const double* d_x, int ntotal, const NbrGrid& g, const NbrScratch& s, hipStream_t st, const int* d_species) {
  note_launch_error(hipMemsetAsync(s.cell_count, 0, sizeof(int) * (size_t)(g.ncell + 1), st));
  note_launch_error(hipMemsetAsync(s.cursor, 0, sizeof(int) * (size_t)g.ncell, st));
  if (ntotal <= 0) {
    note_launch_error(hipMemsetAsync(s.cell_start, 0, sizeof(int) * (size_t)(g.ncell + 1), st));
    return;
  }
  const dim3 grid((ntotal + 255) / 256), block(256);
  hipLaunchKernelGGL(nbr_bin_count_kernel, grid, block, 0, st, d_x, ntotal, g, s.cell_id, s.cell_count);
  hipLaunchKernelGGL(nbr_scan_kernel, dim3(1), dim3(1024), 0, st, s.cell_count, s.cell_start, g.ncell);
  hipLaunchKernelGGL(nbr_bin_fill_kernel, grid, block, 0, st, s.cell_id, s.cell_start, s.cursor, s.order, ntotal);
  hipLaunchKernelGGL(nbr_bin_sort_kernel, dim3((g.ncell + 3) / 4), block, 0, st, s.cell_start, s.order, g.ncell);
  if (d_species && s.xq) hipLaunchKernelGGL(nbr_gather_packed_kernel, grid, block, 0, st, d_x, s.order, d_species, ntotal, g, s.xs, s.xq);
  else hipLaunchKernelGGL(nbr_gather_kernel, grid, block, 0, st, d_x, s.order, ntotal, s.xs);
}

namespace {
void nbr_scan_offsets(int nlocal, int ntotal, const NbrScratch& s, int* d_numneigh, int* d_nbr_off, hipStream_t st) {
  // cell_id is free once the atoms are binned: scratch for the chunk totals (nlocal / 4096 + 1 <= ntotal entries)
  const int nchunks = nlocal / kScanChunk + 1;
  if (nchunks > ntotal) { hipLaunchKernelGGL(nbr_scan_kernel, dim3(1), dim3(1024), 0, st, d_numneigh, d_nbr_off, nlocal); return; }
  hipLaunchKernelGGL(scan_chunk_kernel, dim3(nchunks), dim3(1024), 0, st, d_numneigh, d_nbr_off, nlocal, s.cell_id);
  hipLaunchKernelGGL(scan_add_kernel, dim3((nlocal + 256) / 256), dim3(256), 0, st, d_nbr_off, nlocal, s.cell_id, nchunks);
}
}  // namespace

void launch_nbr_count(int nlocal, int ntotal, const NbrGrid& g, const NbrScratch& s, double cutneigh, int* d_numneigh,
                      int* d_nbr_off, hipStream_t st) {
  if (nlocal > 0 && ntotal > 0)
    hipLaunchKernelGGL(nbr_search_kernel<0>, dim3((unsigned)(((long long)ntotal * kSearchLanes + 255) / 256)), dim3(256), 0, st, s.xs, s.order, s.cell_start, g,
                       nlocal, ntotal, cutneigh * cutneigh, d_numneigh, nullptr, nullptr, 0, nullptr);
  nbr_scan_offsets(nlocal, ntotal, s, d_numneigh, d_nbr_off, st);
}

void launch_nbr_fill(int nlocal, int ntotal, const NbrGrid& g, const NbrScratch& s, double cutneigh, const int* d_nbr_off,
                     int* d_jlist, int* d_ilist, hipStream_t st) {
  if (nlocal <= 0 || ntotal <= 0) return;
  hipLaunchKernelGGL(nbr_search_kernel<1>, dim3((unsigned)(((long long)ntotal * kSearchLanes + 255) / 256)), dim3(256), 0, st, s.xs, s.order, s.cell_start, g,
                     nlocal, ntotal, cutneigh * cutneigh, nullptr, d_nbr_off, d_jlist, 0, nullptr);
  hipLaunchKernelGGL(iota_kernel, dim3((nlocal + 255) / 256), dim3(256), 0, st, d_ilist, nlocal);
}

// count and fill in one pass into rows of `cap` entries (d_jrows[nlocal * cap]); d_numneigh holds the TRUE counts, d_nbr_off
// their scan (the dense offsets the sorted list is written at), *d_ovf = 1 if a centre has more than cap neighbours (the rows
// are then incomplete: the caller fills a dense list with launch_nbr_fill from the counts it already has)
void launch_nbr_onepass(int nlocal, int ntotal, const NbrGrid& g, const NbrScratch& s, double cutneigh, int cap, int* d_numneigh,
                        int* d_nbr_off, int* d_jrows, int* d_ilist, int* d_ovf, hipStream_t st) {
  note_launch_error(hipMemsetAsync(d_ovf, 0, sizeof(int), st));
  if (nlocal > 0 && ntotal > 0) {
    hipLaunchKernelGGL(nbr_search_kernel<2>, dim3((unsigned)(((long long)ntotal * kSearchLanes + 255) / 256)), dim3(256), 0, st, s.xs, s.order, s.cell_start, g,
                       nlocal, ntotal, cutneigh * cutneigh, d_numneigh, nullptr, d_jrows, cap, d_ovf);
    hipLaunchKernelGGL(iota_kernel, dim3((nlocal + 255) / 256), dim3(256), 0, st, d_ilist, nlocal);
  }
  nbr_scan_offsets(nlocal, ntotal, s, d_numneigh, d_nbr_off, st);
}

// the whole build in one kernel (nbr_search_sorted_kernel): rows of `cap` entries grouped by species, true counts, identity
// ilist, *d_ovf = 1 if a centre has more than cap neighbours.  The bins must come from launch_nbr_bin WITH species (s.xq).
bool nbr_sorted_rows_supported(int ntotal, int S, int cap) {
  return ntotal < (1 << kIdxBits) && S <= kMaxSpecies && S <= 16 && cap > 0 && (size_t)cap * 16 * sizeof(int) <= 64 * 1024;
}
void launch_nbr_sorted_rows(int nlocal, int ntotal, const NbrGrid& g, const NbrScratch& s, double cutneigh, int S, int cap,
                            int* d_numneigh, int* d_ilist, int* d_jrows, int* d_ovf, hipStream_t st) {
  note_launch_error(hipMemsetAsync(d_ovf, 0, sizeof(int), st));
  if (nlocal <= 0 || ntotal <= 0) return;
  const dim3 grid((unsigned)(((long long)ntotal * kSearchLanes + 255) / 256)), block(256);
  const size_t lds = (size_t)cap * 16 * sizeof(int);
  if (g.reach == 2)
    hipLaunchKernelGGL(nbr_search_sorted_kernel<2>, grid, block, lds, st, s.xq, s.xs, s.cell_start, g, nlocal, ntotal, cutneigh * cutneigh,
                       (float)cutneigh, S, cap, d_numneigh, d_ilist, d_jrows, d_ovf);
  else
    hipLaunchKernelGGL(nbr_search_sorted_kernel<1>, grid, block, lds, st, s.xq, s.xs, s.cell_start, g, nlocal, ntotal, cutneigh * cutneigh,
                       (float)cutneigh, S, cap, d_numneigh, d_ilist, d_jrows, d_ovf);
}

}  // namespace ani
