// ani_kernels_misc.hip — the small kernels around the two hot passes: position packing, rebuild-time list
// preparation (what src/ani_csrc/ani.cpp:213-229 and models/lammps_ani.py:156-166 do with torch ops every
// rebuild / every step), and the final reductions / unit conversion (src/ani_csrc/ani.cpp:246-262).
#include "ani_kernels.h"
#include "ani_scan.h"

namespace ani {

// Midpoint of the bounding box of all atoms of the rank, once per list epoch.  The fp32 copies of the positions are
// taken relative to it: a sub-domain that sits 150 A from the origin of the simulation box would otherwise carry four
// times the rounding error of one that contains it (fp32 ulp 1.5e-5 A against 3.8e-6 A; the reference converts absolute
// coordinates, src/ani_csrc/ani.cpp:206-207).  Only differences of positions enter the AEVs, so nothing else changes.
__global__ __launch_bounds__(1024) void origin_kernel(const double* __restrict__ x, int ntotal, double* __restrict__ origin) {
  __shared__ double lo[3][16], hi[3][16];
  double a[3] = {1e300, 1e300, 1e300}, b[3] = {-1e300, -1e300, -1e300};
  // any point inside the cloud serves: beyond 8192 atoms a strided sample of them gives the box (one block scanning all
  // 145 000 atoms of the benchmark's rank took 69 us of every re-neighbouring step)
  const int stride = ntotal > 8192 ? ntotal >> 13 : 1;
  for (long long i = (long long)threadIdx.x * stride; i < ntotal; i += (long long)blockDim.x * stride)
    for (int k = 0; k < 3; k++) {
      const double v = x[3 * i + k];
      a[k] = v < a[k] ? v : a[k];
      b[k] = v > b[k] ? v : b[k];
    }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = 0; k < 3; k++) {
    for (int off = 32; off > 0; off >>= 1) {
      const double oa = __shfl_xor(a[k], off), ob = __shfl_xor(b[k], off);
      a[k] = oa < a[k] ? oa : a[k];
      b[k] = ob > b[k] ? ob : b[k];
    }
    if (lane == 0) { lo[k][wave] = a[k]; hi[k][wave] = b[k]; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int k = threadIdx.x;
    double m = lo[k][0], M = hi[k][0];
    for (int w = 1; w < (int)(blockDim.x >> 6); w++) { m = lo[k][w] < m ? lo[k][w] : m; M = hi[k][w] > M ? hi[k][w] : M; }
    origin[k] = ntotal > 0 ? 0.5 * (m + M) : 0.0;
  }
}

void launch_origin(const double* d_x, int ntotal, double* d_origin, hipStream_t st) {
  hipLaunchKernelGGL(origin_kernel, dim3(1), dim3(1024), 0, st, d_x, ntotal, d_origin);
}

__global__ void pack_kernel(const double* __restrict__ x, const int* __restrict__ species, int i0, int i1, SpeciesMap cmap,
                            float4* __restrict__ out, float* __restrict__ fbuf, double* __restrict__ virial_acc,
                            double* __restrict__ ev_zero, const double* __restrict__ origin, GhostFold gf) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (virial_acc && t < 9 * kVirialSlots) virial_acc[t] = 0.0;
  if (ev_zero && t < 10) ev_zero[t] = 0.0;   // the finish kernel ADDS its block sums of the energy
  const int i = i0 + t;
  if (i >= i1) return;
  // coordinates.to(dtype): src/ani_csrc/ani.cpp:206-207, relative to the epoch's origin.  The species stored next to the
  // position is the index the AEV kernels use (compact index among the species present, see ani_hip.cpp:specialize).
  const int sp = species[i];
  const int cs = (sp >= 0 && sp < kMaxSpecies) ? cmap.m[sp] : 0;
  double px, py, pz;
  if (gf.owner && i >= gf.nlocal) {
    // ghost fold (ani_set_ghost_fold): the ghost is an image of an owned atom of this very rank -- its position is the owner's
    // plus the image shift, which is also written back to the caller's array (the forward communication of the step)
    const int g = i - gf.nlocal;
    const long long o = gf.owner[g];
    px = x[3 * o] + gf.shift[3 * g]; py = x[3 * o + 1] + gf.shift[3 * g + 1]; pz = x[3 * o + 2] + gf.shift[3 * g + 2];
    double* xw = const_cast<double*>(x);
    xw[3 * i] = px; xw[3 * i + 1] = py; xw[3 * i + 2] = pz;
  } else {
    px = x[3 * i]; py = x[3 * i + 1]; pz = x[3 * i + 2];
  }
  out[i] = make_float4((float)(px - origin[0]), (float)(py - origin[1]), (float)(pz - origin[2]), __int_as_float(cs));
  reinterpret_cast<float4*>(fbuf)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

void launch_pack(const double* d_x, const int* d_species, int i0, int i1, const SpeciesMap& cmap, float4* xyzs, float* fbuf,
                 double* virial_acc, double* ev_zero, const double* d_origin, hipStream_t st, const GhostFold* gf) {
  int nthr = i1 - i0;
  if (virial_acc && nthr < 9 * kVirialSlots) nthr = 9 * kVirialSlots;
  if (ev_zero && nthr < 10) nthr = 10;
  if (nthr <= 0) return;
  hipLaunchKernelGGL(pack_kernel, dim3((nthr + 255) / 256), dim3(256), 0, st, d_x, d_species, i0, i1, cmap,
                     xyzs, fbuf, virial_acc, ev_zero, d_origin, gf ? *gf : GhostFold{});
}

// images of an owned atom as a chain: head[owner] -> ghost -> next[ghost] -> ... -> -1 (order = arrival of the atomics)
__global__ __launch_bounds__(256) void ghost_chain_kernel(const long long* __restrict__ owner, int nghost, int nlocal, int* __restrict__ head,
                                                          int* __restrict__ next, int* __restrict__ bad) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nghost) return;
  const long long o = owner[g];
  if (o < 0 || o >= nlocal) { *bad = 1; next[g] = -1; return; }
  next[g] = atomicExch(head + o, g);
}
void launch_ghost_chain(const long long* d_owner, int nghost, int nlocal, int* d_head, int* d_next, int* d_bad, hipStream_t st) {
  note_launch_error(hipMemsetAsync(d_head, 0xff, sizeof(int) * (size_t)(nlocal > 0 ? nlocal : 1), st));
  note_launch_error(hipMemsetAsync(d_bad, 0, sizeof(int), st));
  if (nghost > 0) hipLaunchKernelGGL(ghost_chain_kernel, dim3((nghost + 255) / 256), dim3(256), 0, st, d_owner, nghost, nlocal, d_head, d_next, d_bad);
}

// energy, virial and the error word of a step to page-locked host memory in ONE small kernel: two hipMemcpyAsync into the
// caller's stack cost a staging copy each and ~20 us of gaps between copy-engine commands and kernels (100 002 atoms)
__global__ void step_tail_kernel(const double* __restrict__ ev, const int* __restrict__ flag, volatile double* __restrict__ host_out,
                                 double stamp) {
  const int t = threadIdx.x;
  if (t < 10) host_out[t] = ev[t];
  if (t == 10) host_out[10] = (double)*flag;
  __threadfence_system();
  __syncthreads();
  if (t == 0) { host_out[11] = stamp; __threadfence_system(); }   // a host that polls sees the stamp behind the values
}
void launch_step_tail(const double* d_ev, const int* d_flag, double* host_out, double stamp, hipStream_t st) {
  hipLaunchKernelGGL(step_tail_kernel, dim3(1), dim3(64), 0, st, d_ev, d_flag, host_out, stamp);
}

// Forces to page-locked host memory by a kernel, chunk after chunk, each chunk announced by a word the host polls: the host adds
// chunk c into the caller's array while chunk c + 1 is on its way (finish_host, option out_force_accumulate).  A few workgroups
// walk the chunks together so that they complete in order; the last workgroup to finish a chunk (device counter, which only
// ever grows: epoch * gridDim.x when everybody is through) writes the chunk's word.  Every thread fences its stores system-wide
// before the workgroup reports, so the word is behind the data.
__global__ __launch_bounds__(256) void copy_out_kernel(const double* __restrict__ f, double* __restrict__ host, long long n, int nchunks,
                                                       unsigned long long* __restrict__ ctr, volatile unsigned* __restrict__ host_flags,
                                                       unsigned epoch) {
  const double2* __restrict__ f2 = reinterpret_cast<const double2*>(f);
  double2* __restrict__ h2 = reinterpret_cast<double2*>(host);
  for (int c = 0; c < nchunks; c++) {
    const long long a = (n * c / nchunks) & ~1LL, b = c + 1 == nchunks ? n : ((n * (c + 1) / nchunks) & ~1LL);
    const long long len2 = (b - a) >> 1;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < len2; k += (long long)gridDim.x * blockDim.x)
      h2[(a >> 1) + k] = f2[(a >> 1) + k];
    if (((b - a) & 1) && blockIdx.x == 0 && threadIdx.x == 0) host[b - 1] = f[b - 1];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned long long done = atomicAdd(&ctr[c], 1ULL);
      if (done + 1 == (unsigned long long)epoch * gridDim.x) {
        host_flags[c] = epoch;
        __threadfence_system();
      }
    }
  }
}
void launch_copy_out(const double* d_f, double* host, long long n, int nchunks, unsigned long long* d_ctr, unsigned* host_flags,
                     unsigned epoch, int nblocks, hipStream_t st) {
  hipLaunchKernelGGL(copy_out_kernel, dim3(nblocks), dim3(256), 0, st, d_f, host, n, nchunks, d_ctr, host_flags, epoch);
}

// ---- rebuild-time preparation ------------------------------------------------------------------------
// Two kernels over chunks of kPrepChunk centres.  The first scans, inside every chunk, the list lengths (-> offsets
// relative to the chunk) and the species flags (-> stable rank of a centre among the centres of its species in the
// chunk), and leaves the chunk totals; the second adds the totals of the chunks before and writes the rows.  (One block
// scanning all centres round after round was latency-bound: ~0.1 ms per 100 000 centres and scan.)
constexpr int kPrepVpt = 4;
constexpr int kPrepChunk = 1024 * kPrepVpt;

__global__ __launch_bounds__(1024) void prepare_count_kernel(const int* __restrict__ species, const int* __restrict__ ilist,
                                                              const int* __restrict__ numneigh, int nlocal, int ntotal, int S,
                                                              PrepOut o, int* __restrict__ rank_in_species,
                                                              int* __restrict__ chunk_tot) {
  __shared__ int wave_sums[16];
  const int b = blockIdx.x, i0 = b * kPrepChunk + threadIdx.x * kPrepVpt;
  int* tot = chunk_tot + (size_t)b * (kMaxSpecies + 1);
  // list lengths
  int nn[kPrepVpt], sp[kPrepVpt], sum = 0, vmax = 0, bad = 0;
#pragma unroll
  for (int k = 0; k < kPrepVpt; k++) {
    const int ii = i0 + k;
    nn[k] = ii < nlocal ? numneigh[ii] : 0;
    sum += nn[k];
    vmax = nn[k] > vmax ? nn[k] : vmax;
    sp[k] = -1;
    if (ii < nlocal) {
      const int i = ilist[ii];
      if (i < 0 || i >= ntotal) bad = 1;
      else sp[k] = species[i];
    }
  }
  int total;
  int ex = block_exclusive_scan(sum, total, wave_sums);
#pragma unroll
  for (int k = 0; k < kPrepVpt; k++) {
    if (i0 + k < nlocal) o.nbr_off[i0 + k] = ex;
    ex += nn[k];
  }
  if (threadIdx.x == 0) {
    tot[0] = total;
    if (total) atomicAdd(&o.bucket_info[2 * kMaxSpecies + 4], total);   // all pairs of the list (fits: the callers bound it by 2^31)
  }
  if (vmax) atomicMax(&o.bucket_info[2 * kMaxSpecies + 2], vmax);
  // stable rank inside the species, chunk-relative
  for (int s = 0; s < S; s++) {
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < kPrepVpt; k++) cnt += sp[k] == s ? 1 : 0;
    int r = block_exclusive_scan(cnt, total, wave_sums);
#pragma unroll
    for (int k = 0; k < kPrepVpt; k++)
      if (sp[k] == s) rank_in_species[i0 + k] = r++;
    if (threadIdx.x == 0) {
      tot[1 + s] = total;
      if (total) atomicAdd(&o.bucket_info[s], total);
    }
  }
  // every atom, ghosts included: neighbour species index the AEV row (a species outside [0,S) is flagged: the reference
  // would fail inside the network lookup); also record which species occur at all
  int present = 0;
  for (int i = b * blockDim.x + threadIdx.x; i < ntotal; i += gridDim.x * blockDim.x) {
    const int q = species[i];
    if (q < 0 || q >= S) bad = 1;
    else present |= 1 << q;
  }
  if (present) atomicOr(&o.bucket_info[2 * kMaxSpecies + 3], present);
  if (bad) o.bucket_info[2 * kMaxSpecies + 1] = 1;
}

__global__ __launch_bounds__(256) void prepare_rows_kernel(const int* __restrict__ species, const int* __restrict__ ilist,
                                                            const int* __restrict__ numneigh, int nlocal, int ntotal, int S,
                                                            int nrows_cap, PrepOut o, const int* __restrict__ rank_in_species,
                                                            const int* __restrict__ chunk_tot, int row_stride) {
  __shared__ int row_start[kMaxSpecies + 1];
  __shared__ int before[kMaxSpecies + 1];   // totals of the chunks before this block's chunk: pairs, centres per species
  const int chunk = (blockIdx.x * blockDim.x) / kPrepChunk;   // kPrepChunk is a multiple of the block size
  if (threadIdx.x <= (unsigned)S) {
    int acc = 0;
    for (int c = 0; c < chunk; c++) acc += chunk_tot[(size_t)c * (kMaxSpecies + 1) + threadIdx.x];
    before[threadIdx.x] = acc;
  }
  if (threadIdx.x == 0) {
    int r = 0;
    for (int s = 0; s < S; s++) {
      row_start[s] = r;
      r += (o.bucket_info[s] + kRowTile - 1) / kRowTile * kRowTile;
    }
    row_start[S] = r;
    if (blockIdx.x == 0) {
      for (int s = 0; s < S; s++) o.bucket_info[kMaxSpecies + s] = row_start[s];
      o.bucket_info[2 * kMaxSpecies] = r;
    }
  }
  __syncthreads();
  const int ii = blockIdx.x * blockDim.x + threadIdx.x;
  if (ii >= nlocal) {
    if (nlocal == 0 && ii == 0) o.nbr_off[0] = 0;
    return;
  }
  // dense segments, or rows of row_stride entries (the list built by launch_nbr_sorted_rows; ilist is the identity there)
  const int len = numneigh[ii];
  const int off = row_stride > 0 ? ii * row_stride : before[0] + o.nbr_off[ii];
  o.nbr_off[ii] = off;
  if (ii == nlocal - 1) o.nbr_off[nlocal] = row_stride > 0 ? nlocal * row_stride : off + len;
  const int i = ilist[ii];
  const int sp = (i >= 0 && i < ntotal) ? species[i] : -1;   // a bad index was flagged by the first kernel: the caller stops
  if (sp < 0 || sp >= S) { o.row_of_centre[ii] = -1; return; }
  const int row = row_start[sp] + before[1 + sp] + rank_in_species[ii];
  o.row_of_centre[ii] = row;
  if (row < nrows_cap) {
    o.centre_of_row[row] = ii;
    o.row_info[row] = make_int4(i, off, len, ii);
    if (o.row_of_atom) o.row_of_atom[i] = row;
  }
}

size_t prepare_scratch_ints(int nlocal) {   // row_of_centre: [nlocal] rows, [nlocal] ranks, chunk totals
  return (size_t)2 * nlocal + ((size_t)nlocal / kPrepChunk + 1) * (kMaxSpecies + 1) + 2;
}

void launch_prepare(const int* d_species, const int* d_ilist, const int* d_numneigh, int nlocal, int ntotal, int S, int nrows_cap,
                    const PrepOut& o, hipStream_t st, int row_stride) {
  // row_of_centre is sized prepare_scratch_ints(nlocal) by the caller
  int* rank = o.row_of_centre + nlocal;
  int* chunk_tot = o.row_of_centre + 2 * (size_t)nlocal;
  const int nchunks = nlocal / kPrepChunk + 1;
  note_launch_error(hipMemsetAsync(o.centre_of_row, 0xff, sizeof(int) * (size_t)nrows_cap, st));
  note_launch_error(hipMemsetAsync(o.row_info, 0xff, sizeof(int4) * (size_t)nrows_cap, st));
  note_launch_error(hipMemsetAsync(o.bucket_info, 0, sizeof(int) * kBucketInfoInts, st));
  if (o.row_of_atom) note_launch_error(hipMemsetAsync(o.row_of_atom, 0xff, sizeof(int) * (size_t)(ntotal > 0 ? ntotal : 1), st));
  hipLaunchKernelGGL(prepare_count_kernel, dim3(nchunks), dim3(1024), 0, st, d_species, d_ilist, d_numneigh, nlocal, ntotal, S, o,
                     rank, chunk_tot);
  hipLaunchKernelGGL(prepare_rows_kernel, dim3(nlocal > 0 ? (nlocal + 255) / 256 : 1), dim3(256), 0, st, d_species, d_ilist,
                     d_numneigh, nlocal, ntotal, S, nrows_cap, o, rank, chunk_tot, row_stride);
}

// ---- is the list symmetric? ------------------------------------------------------------------------------------------
// The backward kernel's symmetric radial collection (AevArgs::row_of_atom) takes both radial terms of a pair on the centre's
// side and scatters nothing to a neighbour that is itself a centre: right only if j is in i's list exactly when i is in j's.
// A list built here is (one cutoff test in fp64 for both directions); a caller's list may not be (exclusions applied to one
// direction, a truncated list).  Check of a caller's list, once per epoch: every entry i -> j between two centres adds
// g(i, j) = g(j, i) to i's sum and subtracts it from j's; all sums end at zero if (not only if: a 32-bit hash sum) the
// entries come in mirrored pairs.  One wave per centre, one atomic per entry.
__device__ __forceinline__ unsigned pair_hash(unsigned a, unsigned b) {
  const unsigned lo = a < b ? a : b, hi = a < b ? b : a;
  unsigned h = lo * 0x9E3779B1u ^ (hi + 0x7F4A7C15u) * 0x85EBCA77u;
  h ^= h >> 15; h *= 0xC2B2AE3Du; h ^= h >> 13;
  return h | 1u;
}
__global__ __launch_bounds__(256) void list_symmetry_kernel(const int* __restrict__ ilist, const int* __restrict__ nbr_off,
                                                            const int* __restrict__ numneigh, const int* __restrict__ jraw,
                                                            const int* __restrict__ row_of_atom, int nlocal, int ntotal,
                                                            unsigned* __restrict__ acc) {
  const int ii = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (ii >= nlocal) return;
  const int i = ilist[ii], n = numneigh[ii];   // numneigh[ii] belongs to centre ilist[ii]
  const int* seg = jraw + nbr_off[ii];
  unsigned own = 0;
  for (int k = lane; k < n; k += 64) {
    const int j = seg[k];
    if (j < 0 || j >= ntotal || row_of_atom[j] < 0) continue;   // ghosts and padding have no list of their own
    const unsigned g = pair_hash((unsigned)i, (unsigned)j);
    own += g;
    atomicSub(acc + j, g);
  }
  for (int o = 32; o > 0; o >>= 1) own += __shfl_xor(own, o);
  if (lane == 0 && own) atomicAdd(acc + i, own);
}
__global__ __launch_bounds__(256) void list_symmetry_verdict_kernel(const unsigned* __restrict__ acc, int ntotal, int* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ntotal && acc[i] != 0u) atomicOr(out, 1);
}
void launch_list_symmetry(const int* d_ilist, const int* d_nbr_off, const int* d_numneigh, const int* d_jraw, const int* d_row_of_atom,
                          int nlocal, int ntotal, unsigned* d_acc, int* d_out, hipStream_t st) {
  note_launch_error(hipMemsetAsync(d_acc, 0, sizeof(unsigned) * (size_t)(ntotal > 0 ? ntotal : 1), st));
  note_launch_error(hipMemsetAsync(d_out, 0, sizeof(int), st));
  if (nlocal <= 0) return;
  hipLaunchKernelGGL(list_symmetry_kernel, dim3((nlocal + 3) / 4), dim3(256), 0, st, d_ilist, d_nbr_off, d_numneigh, d_jraw, d_row_of_atom,
                     nlocal, ntotal, d_acc);
  hipLaunchKernelGGL(list_symmetry_verdict_kernel, dim3((ntotal + 255) / 256), dim3(256), 0, st, d_acc, ntotal, d_out);
}

// ---- final reductions ------------------------------------------------------------------------------------
// ONE launch: blocks [0, kFinishBlocks) sum the row energies (enough blocks that each walks its rows in a few dependent loads) and ADD their share to ev_out[0] (zeroed by pack_kernel;
// a few double atomics on one address), block kFinishBlocks reduces the virial rows, the rest convert the force
// accumulators.  Three dependent launches of a few microseconds each were 6 % of a 12 500-atom step.
constexpr int kFinishBlocks = 128;

__global__ __launch_bounds__(256) void finish_kernel(FinishArgs a) {
  __shared__ double red[256];
  const int b = a.energy ? blockIdx.x : blockIdx.x + kFinishBlocks + 1;   // without the energy / virial blocks
  if (b < kFinishBlocks) {
    double acc = 0.0;
    for (int row = b * blockDim.x + threadIdx.x; row < a.nrows; row += kFinishBlocks * blockDim.x) {
      const int ii = a.centre_of_row[row];
      if (ii < 0) continue;
      float e = 0.f;
      for (int m = 0; m < a.M; m++) e += a.e_rows[(long long)m * a.nrows_ld + row];
      // energy_shifter (models/lammps_ani.py:230,250), self energy added in fp64
      const double ea = (double)e + a.sae[a.species[a.ilist[ii]]];
      if (a.eatom_out) a.eatom_out[ii] = ea * 627.5094738898777;
      acc += ea;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      // a capacity overflow turns the energy into NaN so that device-resident callers notice
      const bool bad = b == 0 && a.err_flag && *a.err_flag;
      atomicAdd(&a.ev_out[0], bad ? __longlong_as_double(0x7ff8000000000000LL) : red[0] * 627.5094738898777);
    }
    return;
  }
  if (b == kFinishBlocks) {
    // virial: sum the partial rows, then (virial.t() + virial)/2 : models/lammps_ani.py:200
    if (!a.virial_acc) return;   // ev_out[1..9] stay zero
    __shared__ double vsum[9];
    for (int c = 0; c < 9; c++) {
      double t = 0.0;
      for (int sl = threadIdx.x; sl < kVirialSlots; sl += blockDim.x) t += a.virial_acc[9 * sl + c];
      red[threadIdx.x] = t;
      __syncthreads();
      for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if ((int)threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
        __syncthreads();
      }
      if (threadIdx.x == 0) vsum[c] = red[0];
      __syncthreads();
    }
    if (threadIdx.x < 9) {
      const int k = threadIdx.x / 3, l = threadIdx.x % 3;
      a.ev_out[1 + threadIdx.x] = 0.5 * (vsum[3 * k + l] + vsum[3 * l + k]) * 627.5094738898777;
    }
    return;
  }
  const int i = 3 * a.atom0 + (b - kFinishBlocks - 1) * blockDim.x + threadIdx.x;
  if (i >= 3 * a.atom1) return;
  const int at = i / 3, c = i - 3 * at;
  float acc = a.fbuf[4 * at + c];  // accumulators are one float4 per atom
  if (a.fold_head) {               // ghost fold: the rows of the atom's images come home here (the reverse communication)
    for (int g = a.fold_head[at]; g >= 0; g = a.fold_next[g]) acc += a.fbuf[4 * (a.fold_nlocal + g) + c];
  }
  const double v = (double)acc * 627.5094738898777;
  a.f_out[i] = a.f_accumulate ? a.f_out[i] + v : v;
}

void launch_finish(const FinishArgs& a, hipStream_t st) {
  const int n3 = a.f_out ? (a.atom1 - a.atom0) * 3 : 0;
  const int blocks = (a.energy ? kFinishBlocks + 1 : 0) + (n3 + 255) / 256;
  if (blocks <= 0) return;
  hipLaunchKernelGGL(finish_kernel, dim3(blocks), dim3(256), 0, st, a);
}

// ---- rows with and without a ghost atom among their candidates (rebuild time) ---------------------------------
// flag[row] = 1 if any entry of the row's list is a ghost (index >= nlocal); one wave per row
__global__ __launch_bounds__(256) void classify_rows_kernel(const int4* __restrict__ row_info, const int* __restrict__ jlist, int nrows,
                                                            int nlocal, int* __restrict__ flag) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= nrows) return;
  const int4 info = row_info[row];
  bool ghost = false;
  if (info.x >= 0)
    for (int q = lane; q < info.z; q += 64) ghost = ghost || jlist[info.y + q] >= nlocal;
  const bool any = __ballot(ghost) != 0ULL;
  if (lane == 0) flag[row] = any ? 1 : 0;
}

// stable partition by one workgroup: list = the flagged rows in ascending order, then the others; count[0] = flagged
__global__ __launch_bounds__(1024) void partition_rows_kernel(const int* __restrict__ flag, int nrows, int* __restrict__ list,
                                                              int* __restrict__ count) {
  __shared__ int wsum[16], tot;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  auto block_scan = [&](bool f, int& before, int& total) {   // exclusive rank of this thread's flag inside the block
    const unsigned long long m = __ballot(f);
    const int inw = __popcll(m & ((1ULL << lane) - 1ULL));
    __syncthreads();
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int base = 0, t = 0;
    for (int w = 0; w < 16; w++) { if (w < wave) base += wsum[w]; t += wsum[w]; }
    before = base + inw;
    total = t;
  };
  int nb = 0;
  for (int r0 = 0; r0 < nrows; r0 += 1024) {
    const int r = r0 + threadIdx.x;
    int before, total;
    block_scan(r < nrows && flag[r] != 0, before, total);
    nb += total;
  }
  if (threadIdx.x == 0) { tot = nb; count[0] = nb; }
  __syncthreads();
  nb = tot;
  int doneB = 0, doneI = 0;
  for (int r0 = 0; r0 < nrows; r0 += 1024) {
    const int r = r0 + threadIdx.x;
    const bool in = r < nrows, f = in && flag[r] != 0;
    int before, total;
    block_scan(f, before, total);
    if (f) list[doneB + before] = r;
    else if (in) list[nb + doneI + (threadIdx.x - before)] = r;
    doneB += total;
    doneI += min(1024, nrows - r0) - total;
  }
}

void launch_row_classes(const int4* row_info, const int* jlist, int nrows, int nlocal, int* flag, int* list, int* count, hipStream_t st) {
  if (nrows <= 0) { note_launch_error(hipMemsetAsync(count, 0, sizeof(int), st)); return; }
  hipLaunchKernelGGL(classify_rows_kernel, dim3((nrows + 3) / 4), dim3(256), 0, st, row_info, jlist, nrows, nlocal, flag);
  hipLaunchKernelGGL(partition_rows_kernel, dim3(1), dim3(1024), 0, st, flag, nrows, list, count);
}

}  // namespace ani
