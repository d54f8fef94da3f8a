// ani_kernels_aev.hip — AEV (radial + angular symmetry functions) forward and analytic backward over the
// LAMMPS full neighbour list.  Replaces torchani's cuaev / pyaev (not in the reference tree; call sites
// models/lammps_ani.py:277-296) and the autograd pass through it (models/lammps_ani.py:197-206).
//
// One wavefront (64 lanes) per centre atom, four centres per workgroup, no workgroup barriers: each wave owns a
// private slice of LDS.  At rebuild time every centre's neighbour segment is sorted by neighbour species
// (sort_jlist_kernel), so the screened lists built per step come out grouped by species and every
// (species) / (species pair) block of the AEV row has exactly one contiguous group of contributors.
//
// Fast path (NR = 16 radial shifts, NA x NZ = 8x4 (ANI-2x) or 4x8 (ANI-1x) angular grid, <= 8 species):
//   compaction : coalesced jlist loads, float4 position gathers, ballot/popcount compaction into LDS.
//   radial     : per species group, lanes = (4 neighbour slots) x (16 shifts); 2 xor-shuffles; plain store.
//   angular    : the pairs of all non-empty species-pair buckets form ONE padded stream (bucket starts are
//                multiples of Q = 64/NA).  Per chunk of 64 pairs: phase 1, lane = pair, computes the NA radial
//                and NZ angular factors once and parks them in LDS; phase 2, lane = (slot q, shift a), walks
//                the chunk in NA steps with NZ register accumulators that are flushed (log2(Q) xor-shuffles,
//                plain 16-byte stores) whenever the bucket changes.  No LDS atomics, bitwise reproducible.
//   backward   : same stream; lane = pair contracts the 32 dE/dAEV entries of its bucket against the factor
//                derivatives and adds the two gradient vectors to per-neighbour LDS accumulators; one global
//                float atomic per neighbour component scatters the force; the virial is reduced per wave.
// Generic path (any NR/NA/NZ/S): the first-generation kernels with LDS float atomics, kept for odd model
// shapes (e.g. the unit-test "tiny" model).
//
// Transcendentals on the fast path are the hardware ones (v_exp_f32, v_log_f32, v_cos_f32, v_sin_f32, v_rcp_f32,
// v_rsq_f32), as the reference's cuaev is built with -use_fast_math (src/ani_csrc/CMakeLists.txt:12-20).
// Formulas and their derivatives are the ones restated in oracle/ani_oracle.c.
#include <algorithm>
#include <cstdlib>

#include <map>
#include <mutex>
#include <set>
#include <utility>

#include "ani_kernels.h"

namespace ani {

constexpr int kWaves = 4;    // centres per workgroup: forward fast path and generic kernels
constexpr int kWavesB = 4;   // backward fast path
#ifndef ANI_BWD_MINW
#define ANI_BWD_MINW 4
#endif
#ifndef ANI_FWD_MINW
#define ANI_FWD_MINW 4
#endif
#ifndef ANI_FUSED_MINW
#define ANI_FUSED_MINW 4
#endif
constexpr int kAevMax = 1024;   // LDS floats reserved for one AEV row (ANI-2x: 1008)
constexpr int kMaxBuckets = 36; // species pairs on the fast path (S <= 8)

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ int lanes_below(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}
// inclusive prefix sum over the 64 lanes without LDS: four DPP shifts inside each row of 16 lanes, then the row totals
// passed on with row_bcast:15 (rows 1, 3) and row_bcast:31 (rows 2, 3)
__device__ __forceinline__ int wave_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
  return v;
}
#ifdef ABL_FAKE_TRANS   // timing experiment only (wrong results): every transcendental replaced by one multiply-add
__device__ __forceinline__ float fexp2(float x) { return fmaf(x, 0.001f, 1.0f); }
__device__ __forceinline__ float flog2(float x) { return fmaf(x, 0.5f, -0.5f); }
__device__ __forceinline__ float frcp(float x) { return fmaf(x, -0.1f, 1.0f); }
__device__ __forceinline__ float frsq(float x) { return fmaf(x, -0.1f, 1.0f); }
__device__ __forceinline__ float fcos_rev(float rev) { return fmaf(rev, -0.5f, 1.0f); }
__device__ __forceinline__ float fsin_rev(float rev) { return fmaf(rev, 0.5f, 0.1f); }
#else
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float flog2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float frsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fcos_rev(float rev) { return __builtin_amdgcn_cosf(rev); }  // cos(2 pi rev)
__device__ __forceinline__ float fsin_rev(float rev) { return __builtin_amdgcn_sinf(rev); }
#endif
constexpr float kLog2e = 1.4426950408889634f;

// diagnostic build (-DABLB_STAMPS): wave 0 of every backward workgroup adds the shader-clock cycles it spent in each phase
// of a centre to g_bwd_stamps (read through ani_debug_fused_stamps); no stamp executes in the shipped kernel
#ifdef ABLB_STAMPS
__device__ unsigned long long g_bwd_stamps[32];
#define BWD_STAMP(k)                                                                          \
  do {                                                                                        \
    unsigned long long _t;                                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory"); \
    stamp_acc[k] += _t - stamp_prev;                                                          \
    stamp_prev = _t;                                                                          \
  } while (0)
#define BWD_STAMP_VM(k)                                                                       \
  do {                                                                                        \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                          \
    BWD_STAMP(k);                                                                             \
  } while (0)
#define BWD_STAMP_PARAMS , unsigned long long& stamp_prev, unsigned long long (&stamp_acc)[8]
#define BWD_STAMP_ARGS , stamp_prev, stamp_acc
#else
#define BWD_STAMP(k) do {} while (0)
#define BWD_STAMP_VM(k) do {} while (0)
#define BWD_STAMP_PARAMS
#define BWD_STAMP_ARGS
#endif

// unordered pair index t -> (a, b), a < b < n, row-major over the strict upper triangle
__device__ __forceinline__ void decode_pair(int t, int n, int& a, int& b) {
  // closed form + one branch-free correction step each way; all quantities < 2^14, so 24-bit multiplies are exact
  const float fn = (float)(2 * n - 1);
  int aa = (int)((fn - __builtin_amdgcn_sqrtf(fn * fn - 8.f * (float)t)) * 0.5f);
  aa = max(0, min(aa, n - 2));
  const int n2 = 2 * n - 1;
  int s0 = __mul24(aa, n2 - aa) >> 1;              // first pair index of row aa
  const bool dn = s0 > t;
  aa -= dn ? 1 : 0;
  s0 = dn ? (__mul24(aa, n2 - aa) >> 1) : s0;
  const int s1 = __mul24(aa + 1, n2 - aa - 1) >> 1;  // first pair index of row aa + 1
  const bool up = s1 <= t;
  aa += up ? 1 : 0;
  s0 = up ? s1 : s0;
  a = aa;
  b = aa + 1 + (t - s0);
}
__device__ __forceinline__ int triu_index(int s1, int s2, int S) {
  const int lo = s1 < s2 ? s1 : s2, hi = s1 < s2 ? s2 : s1;
  return lo * S - lo * (lo - 1) / 2 + (hi - lo);
}

// =====================================================================================================
// rebuild time: stable sort of every centre's neighbour segment by neighbour species
// =====================================================================================================
__global__ __launch_bounds__(256) void sort_jlist_kernel(const int* __restrict__ species, const int* __restrict__ nbr_off,
                                                         const int* __restrict__ numneigh, const int* __restrict__ jin,
                                                         int* __restrict__ jout, int nlocal, int S, int present, int in_stride) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ii = blockIdx.x * 4 + wave;
  if (ii >= nlocal) return;
  const int beg = nbr_off[ii], n = numneigh[ii];
  // the unsorted entries: dense segments like the output, or rows of a fixed capacity (the one-pass list build)
  jin += in_stride ? (long long)ii * in_stride - beg : 0LL;
  int outpos = 0;
  for (int s = 0; s < S; s++) {
    if (!((present >> s) & 1)) continue;   // a species that occurs nowhere in the system: no pass over the list for it
    for (int base = 0; base < n; base += 64) {
      const int q = base + lane;
      const int j = q < n ? jin[beg + q] : 0;
      const bool hit = q < n && species[j] == s;
      const unsigned long long m = __ballot(hit);
      if (hit) jout[beg + outpos + lanes_below(m)] = j;
      outpos += __popcll(m);
    }
  }
}

void launch_sort_jlist(const int* d_species, const int* d_nbr_off, const int* d_numneigh, const int* d_jin, int* d_jout,
                       int nlocal, int S, int present_mask, hipStream_t st, int in_stride) {
  if (nlocal <= 0) return;
  hipLaunchKernelGGL(sort_jlist_kernel, dim3((nlocal + 3) / 4), dim3(256), 0, st, d_species, d_nbr_off, d_numneigh, d_jin, d_jout,
                     nlocal, S, present_mask, in_stride);
}

// =====================================================================================================
// fast path
// =====================================================================================================
// The backward kernel's per-neighbour gradient accumulators in LDS take several adds per pair.  On gfx950 an LDS fp32 add
// (ds_add_f32, also ds_pk_add_f16) is executed ONE LANE AT A TIME, ~3 cycles per active lane -- 192 cycles for a full wave,
// the CU's LDS blocked meanwhile -- whereas ds_add_f64 takes 8 cycles per wave-instruction, like the integer adds
// (tools/lds_atomic_probe.hip, profiles/r03_lds_atomic_probe.log).  So the accumulators are doubles: one v_cvt_f64_f32 per
// value added, 6 instead of 3 words per neighbour, and sums that are more accurate on top.
#ifdef ANI_GD_F32
typedef float gd_t;
#else
typedef double gd_t;
#endif
constexpr int kGdWords = 3 * (int)(sizeof(gd_t) / 4);   // LDS words per angular neighbour

struct FastLds {
  // carved from dynamic LDS, per wave
  float4* ad;     // [kMaxAng] angular neighbours dx,dy,dz,r
  float* afc;     // [kMaxAng] fc(r; Rca)
  float* row;     // [rowf] forward: AEV row; backward: dE/dAEV row
  float* pf2;     // [64*NA] forward
  float* pf1;     // [64*NZ] forward
  int* tb;        // [kMaxBuckets*8] bucket table
  int* rstart;    // [kMaxSpecies+1]
  int* astart;    // [kMaxSpecies+1]
  float* rr;      // [cap] forward: radial list r
  float* rfc;     // [cap] forward: fc(r; Rcr)
  int* aj;        // [kMaxAng] backward: atom index of the angular neighbours
  gd_t* gd;       // [3*kMaxAng] backward: dE/d(displacement) of the angular neighbours (fp64: see gd_t)
  float* gt;      // [3*64] backward: staging of one chunk of radial-only gradients for the force scatter
  int* jt;        // [64]   ... and of their atom indices
  int4* rowd;     // [2*32] backward: descriptors of 32 rows of the pair stream (build_row_descriptors)
  float4* rowacc; // [2*64] backward, experiment ANI_PARK_ROWS only: per-row sums of the gradients w.r.t. the rows' neighbours
};

#ifdef ANI_PARK_ROWS
constexpr int kParkWords = 256;
#else
constexpr int kParkWords = 0;
#endif
// lanes per row of the backward pair stream (a DPP quad): narrow rows waste few lanes on buckets of 5..12 columns
#ifndef ANI_ROW_W
#define ANI_ROW_W 16
#endif
constexpr int kRowW = ANI_ROW_W;
#ifdef ANI_BWD_ROWS
constexpr int kRowdWords = 256;   // descriptors of 32 rows of the row-layout pair stream
#else
constexpr int kRowdWords = 0;
#endif
__host__ __device__ constexpr int fast_wave_floats(int cap, bool bwd) {
  // both: ad 4*kMaxAng + row kAevMax + tb kMaxBuckets*8 + starts 2*24
  // forward adds afc, the phase-1 factor buffers pf (64*12) and r, fc per radial neighbour;
  // backward adds afc, aj and gd[3] (fp64) per ANGULAR neighbour and the 4*64 staging words (the radial-only neighbours never
  // touch LDS)
  return 4 * kMaxAng + kAevMax + kMaxBuckets * 8 + 48 + (bwd ? (2 + kGdWords) * kMaxAng + 256 + kRowdWords + kParkWords : kMaxAng + 64 * 12 + 2 * cap);
}
// same with the AEV row sized for the columns actually in use (rowf floats, multiple of 64)
// ... and the bucket table for the species pairs of the model in use (S (S + 1) / 2 entries of 8 words: 3 for pruned water, 28 for
// seven species): the 264 words a water run does not need are the difference between five and six forward workgroups per CU
__host__ __device__ constexpr int table_words(int S) { return 8 * (S * (S + 1) / 2); }
__host__ __device__ constexpr int fast_wave_floats_row(int cap, bool bwd, int rowf, int S) {
  return fast_wave_floats(cap, bwd) - kAevMax + rowf - kMaxBuckets * 8 + table_words(S);
}

template <int NA, int NZ>
__device__ __forceinline__ FastLds carve(float* base, int cap, bool bwd, int rowf) {
  FastLds L{};
  float* p = base;
  // fixed-size pieces first so that their offsets from the wave base are compile-time immediates
  L.ad = reinterpret_cast<float4*>(p); p += 4 * kMaxAng;
  L.rstart = reinterpret_cast<int*>(p); p += 24;
  L.astart = reinterpret_cast<int*>(p); p += 24;
  if (!bwd) {
    L.afc = p; p += kMaxAng;
    L.pf2 = p; p += 64 * NA;
    L.pf1 = p; p += 64 * NZ;
    L.row = p; p += rowf;
    L.rr = p; p += cap;
    L.rfc = p; p += cap;
    L.tb = reinterpret_cast<int*>(p);   // sized by the model's species count (table_words): last
  } else {
    L.aj = reinterpret_cast<int*>(p); p += kMaxAng;
    L.gd = reinterpret_cast<gd_t*>(p); p += kGdWords * kMaxAng;   // 8-byte aligned: every piece in front of it is an even number of words
    L.afc = p; p += kMaxAng;
    L.gt = p; p += 3 * 64;
    L.jt = reinterpret_cast<int*>(p); p += 64;
#ifdef ANI_BWD_ROWS
    L.rowd = reinterpret_cast<int4*>(p); p += 256;
#endif
#ifdef ANI_PARK_ROWS
    L.rowacc = reinterpret_cast<float4*>(p); p += 256;
#endif
    L.row = p; p += rowf;
    L.tb = reinterpret_cast<int*>(p);
  }
  return L;
}

// ---- per-step compaction (its own kernel) -------------------------------------------------------------------
// nbr_compact_kernel screens the (species-sorted) candidate list of every centre ONCE per step and leaves, per AEV row,
//   cl_hdr[2*row + 0] = {centre atom i (-1: padding row / skipped), nrad | nang << 16 | centre species << 24,
//                        angular neighbours of species 0..3 (u8 each), of species 4..7}
//   cl_hdr[2*row + 1] = radial-only neighbours (Rca < r <= Rcr) per species, 8 x u16
//   cl_xyz[row*stride + t]            , t < nang : {dx, dy, dz, r} of the neighbours inside Rca, in list (= species) order
//   cl_xyz[row*stride + kMaxAng + u]  , u < nrad - nang : the same for the radial-only neighbours
//   cl_j  [...]                       their atom indices (backward pass: force scatter)
// Two append-only streams per row, so the kernel is ONE pass over the candidates with two ballots per 64 of them.
// The forward and the backward kernel both start from these lists (coalesced 16-byte loads addressed by row: one
// centre ahead is all the prefetch they need) instead of each walking the ~150 candidates again: the walk is a chain of
// dependent loads (list -> positions) with little arithmetic, exactly what a small kernel at eight waves per SIMD
// hides and a 100-170 VGPR compute kernel at three or four does not.
// Persistent waves, software-pipelined across rows: wave w walks rows w, w + W, ...; while row c is screened, the
// position gathers of c+1, the list loads of c+2 and the row_info of c+3 are in flight.  NCH 64-entry chunks of a list
// are held per stage; longer lists take further super-chunks loaded in place.
// vmcnt counts loads and stores together, in order: a wait for prefetched data that is placed AFTER a row's stores
// waits for those stores as well (a full write round trip per row).  So every value a later row needs is waited for
// (touch()) before the current row's stores are issued, and the stores come last in the iteration; row_info is loaded
// through an index the compiler cannot prove uniform, or it would load + v_readfirstlane it on the spot (draining
// every gather in flight) -- it is made scalar only when its row is started.
constexpr int kWavesC = 4;
// a compact-list entry of cl_j: the neighbour's atom index with its (compact) species in the top four bits, so that the backward
// kernel's radial stage knows a neighbour's dE/dAEV row without walking the per-species counts (27 scalars that spilled)
constexpr int kClIndexBits = 28;
__device__ __forceinline__ int cl_pack(int j, int sp) { return j | (sp << kClIndexBits); }
__device__ __forceinline__ int cl_index(int e) { return e & ((1 << kClIndexBits) - 1); }
__device__ __forceinline__ int cl_species(int e) { return (unsigned)e >> kClIndexBits; }
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ void touch(int v) { asm volatile("" ::"v"(v)); }
__device__ __forceinline__ void touch(float v) { asm volatile("" ::"v"(v)); }
__device__ __forceinline__ void touch(const float4& v) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); }
__device__ __forceinline__ void touch(const int4& v) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); }
__device__ __forceinline__ int4 uniform4(const int4& v) {
  return make_int4(__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y), __builtin_amdgcn_readfirstlane(v.z),
                   __builtin_amdgcn_readfirstlane(v.w));
}
// The fast kernels walk the k-th row of a range, k in [0, a.kcount): row = a.row_list[a.k0 + k], or a.k0 + k without a
// list (every row: k0 = 0, kcount = nrows).  Lists: the rows with / without a ghost atom among their candidates, so
// that a step can be cut where the ghost exchange has to happen (ani_step_* in ani_hip.h).
__device__ __forceinline__ int4 load_info(const AevArgs& a, int k) {   // per-lane copy; k past the end reads the last row
  const int kk = opaque(k < a.kcount ? k : a.kcount - 1);
  const int r = a.row_list ? a.row_list[a.k0 + kk] : a.k0 + kk;
  int4 info = a.row_info[r];
  info.w = r;   // the row travels with its stage (the centre's position in ilist, which sat here, is not used below)
  return info;
}
template <int NCH>
__device__ __forceinline__ void load_j(const AevArgs& a, const int4 info, int lane, int (&jj)[NCH]) {
  const int i = info.x < 0 ? 0 : info.x;
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const int q = 64 * c + lane;
    jj[c] = q < info.z ? a.jlist[info.y + q] : i;
  }
}
template <int NCH>
__device__ __forceinline__ void gather_x(const AevArgs& a, const int4 info, const int (&jj)[NCH], float4& xi, float4 (&xx)[NCH]) {
  xi = a.xyzs[info.x < 0 ? 0 : info.x];
#pragma unroll
  for (int c = 0; c < NCH; c++) xx[c] = a.xyzs[jj[c]];
}

template <int NCH>
__global__ __launch_bounds__(64 * kWavesC, NCH >= 4 ? 4 : 5) void nbr_compact_kernel(AevParams p, AevArgs a, int cap) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int nw = gridDim.x * kWavesC;
  int k = blockIdx.x * kWavesC + wave;
  if (k >= a.kcount) return;
  const int cap2 = a.cl_stride - kMaxAng;   // room of the radial-only stream
  int4 info = load_info(a, k), info1 = load_info(a, k + nw), info2 = load_info(a, k + 2 * nw);
  int jj[NCH], jj1[NCH];
  float4 xi, xx[NCH];
  load_j(a, info, lane, jj);
  load_j(a, info1, lane, jj1);
  gather_x(a, info, jj, xi, xx);
  while (k < a.kcount) {
    // next stages
    float4 xi1, xx1[NCH];
    gather_x(a, info1, jj1, xi1, xx1);
    int jj2[NCH];
    load_j(a, info2, lane, jj2);
    const int4 info3 = load_info(a, k + 3 * nw);

    const int4 inf = uniform4(info);
    const int row = inf.w;
    int4* hdr = a.cl_hdr + 2 * (size_t)row;
    const int i = inf.x, beg = inf.y, n = inf.x < 0 ? 0 : inf.z;
    float4* oxyz = a.cl_xyz + (size_t)row * a.cl_stride;
    int* oj = a.cl_j + (size_t)row * a.cl_stride;
    int nA = 0, nR = 0;    // wave-uniform stream lengths
    int ca = 0, c2 = 0;    // lane s: entries of species s in either stream
    float4 dd[NCH];
    int pos[NCH];          // output slot of this lane's candidate of chunk c, -1: screened out
    auto screen = [&](int c, bool valid, const float4& xj, float4& d, int& sp) -> int {
      d.x = xj.x - xi.x; d.y = xj.y - xi.y; d.z = xj.z - xi.z;
      d.w = __builtin_amdgcn_sqrtf(d.x * d.x + d.y * d.y + d.z * d.z);
      sp = __float_as_int(xj.w);
      const bool in_a = valid && d.w <= p.Rca;
      const bool in_2 = valid && !in_a && (p.compat || d.w <= p.Rcr);
      const unsigned long long mA = __ballot(in_a), m2 = __ballot(in_2);
      const int pA = nA + lanes_below(mA), p2 = nR + lanes_below(m2);
      for (int s = 0; s < p.S; s++) {
        const int k1 = __popcll(__ballot(in_a && sp == s)), k2 = __popcll(__ballot(in_2 && sp == s));
        if (lane == s) { ca += k1; c2 += k2; }
      }
      nA += __popcll(mA);
      nR += __popcll(m2);
      return (in_a && pA < kMaxAng) ? pA : ((in_2 && p2 < cap2) ? kMaxAng + p2 : -1);
    };
    int spv[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      pos[c] = -1;
      spv[c] = 0;
      dd[c] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (64 * c < n) pos[c] = screen(c, 64 * c + lane < n, xx[c], dd[c], spv[c]);   // wave-uniform test
    }
    // everything the later rows need has now to be in registers: the stores below must not be waited for
    touch(xi1); touch(info3);
#pragma unroll
    for (int c = 0; c < NCH; c++) { touch(xx1[c]); touch(jj2[c]); }
#pragma unroll
    for (int c = 0; c < NCH; c++)
      if (pos[c] >= 0) { oxyz[pos[c]] = dd[c]; oj[pos[c]] = cl_pack(jj[c], spv[c]); }
    for (int base0 = 64 * NCH; base0 < n; base0 += 64) {   // more than 64 * NCH list entries: rare, loaded in place
      const int q = base0 + lane;
      const int j = q < n ? a.jlist[beg + q] : i;
      const float4 xj = a.xyzs[j];
      float4 d;
      int sp;
      const int ps = screen(0, q < n, xj, d, sp);
      if (ps >= 0) { oxyz[ps] = d; oj[ps] = cl_pack(j, sp); }
    }
    // capacity of the consumers' LDS lists: never a silent truncation -- the row is skipped and the error flag raised
    const bool over = nA > kMaxAng || nR > cap2 || nA + nR > cap;
    if (lane < 8) {
      reinterpret_cast<unsigned char*>(hdr)[8 + lane] = (unsigned char)ca;
      reinterpret_cast<unsigned short*>(hdr)[8 + lane] = (unsigned short)c2;
    }
    if (lane == 0) {
      reinterpret_cast<int*>(hdr)[0] = (over || inf.x < 0) ? -1 : i;
      reinterpret_cast<int*>(hdr)[1] = (nA + nR) | (nA << 16) | (__float_as_int(xi.w) << 24);
      if (over) atomicOr(a.err_flag, 1);
    }
    info = info1; info1 = info2; info2 = info3;
    xi = xi1;
#pragma unroll
    for (int c = 0; c < NCH; c++) { jj[c] = jj1[c]; xx[c] = xx1[c]; jj1[c] = jj2[c]; }
    k += nw;
  }
}

// What the forward / backward kernels load for a centre: first the header (wave-uniform, two 16-byte words), then the
// first NCH 64-entry chunks of its compact list in slot order -- slots [0, nang) are the angular region, slots
// [nang, nrad) the radial-only one -- so that a typical centre (nrad < 64) is ONE chunk with every lane busy instead
// of an angular chunk (a quarter of the lanes) and a radial-only chunk; for the backward pass also the atom indices and
// the dE/dAEV row.  No prefetch across centres (see ANI_PERSISTENT_LOOP).
typedef int hdr_t __attribute__((ext_vector_type(8)));
template <int NCH, bool BWD, int GR>
struct Loaded {
  float4 xx[NCH];
  int jj[BWD ? NCH : 1];
  float4 grow[BWD ? GR : 1];
};
__device__ __forceinline__ hdr_t load_header(const AevArgs& a, int row) {   // row is wave-uniform
  const int4* hp = a.cl_hdr + 2 * (size_t)row;
  const int4 h0 = hp[0], h1 = hp[1];
  hdr_t h;
  h[0] = __builtin_amdgcn_readfirstlane(h0.x); h[1] = __builtin_amdgcn_readfirstlane(h0.y);
  h[2] = __builtin_amdgcn_readfirstlane(h0.z); h[3] = __builtin_amdgcn_readfirstlane(h0.w);
  h[4] = __builtin_amdgcn_readfirstlane(h1.x); h[5] = __builtin_amdgcn_readfirstlane(h1.y);
  h[6] = __builtin_amdgcn_readfirstlane(h1.z); h[7] = __builtin_amdgcn_readfirstlane(h1.w);
  return h;
}
// The same through the scalar cache: the header is wave-uniform, a scalar load leaves eight vector moves out -- and it is counted
// by lgkmcnt, not by the in-order vmcnt queue, so it does not wait for the force atomics the wave still has in flight.
__device__ __forceinline__ hdr_t load_header_scalar(const AevArgs& a, int row) {   // row is wave-uniform
  const int4* hp = a.cl_hdr + 2 * (size_t)row;
  hdr_t h;
  asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(h) : "s"(hp) : "memory");
  return h;
}
// compact-list slot t (angular entries first, then the radial-only ones) -> index in the row's cl_xyz / cl_j
__device__ __forceinline__ int slot_index(int t, int nang) { return t < nang ? t : kMaxAng + (t - nang); }
template <int NCH, bool BWD, int GR>
__device__ __forceinline__ void load_lists(const AevParams& p, const AevArgs& a, int row, int nrad, int nang, int lane,
                                           Loaded<NCH, BWD, GR>& ld) {
  const float4* px = a.cl_xyz + (size_t)row * a.cl_stride;
  const int* pj = a.cl_j + (size_t)row * a.cl_stride;
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const int t = 64 * c + lane;
    ld.xx[c] = t < nrad ? px[slot_index(t, nang)] : make_float4(0.f, 0.f, 0.f, 1.f);
    if constexpr (BWD) ld.jj[c] = t < nrad ? pj[slot_index(t, nang)] : 0;
  }
  if constexpr (BWD) {
    const int n4 = p.aev_stride >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(a.gaev + (size_t)row * p.aev_stride);
#pragma unroll
    for (int c = 0; c < GR; c++) ld.grow[c] = lane + 64 * c < n4 ? g4[lane + 64 * c] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
__device__ __forceinline__ int hdr_nrad(const hdr_t& h) { return h[0] < 0 ? 0 : (h[1] & 0xffff); }
__device__ __forceinline__ int hdr_nang(const hdr_t& h) { return h[0] < 0 ? 0 : ((h[1] >> 16) & 0xff); }
__device__ __forceinline__ int hdr_species(const hdr_t& h) { return (h[1] >> 24) & 0xf; }

// scalar bookkeeping of a centre's compact list: starts of species k in the angular stream (as), in the radial-only
// stream (r2) and in the species-grouped radial order (rs); entry 8 holds the totals
struct Groups { int as[9], r2[9], rs[9]; };
__device__ __forceinline__ Groups unpack_groups(const hdr_t& h) {
  Groups g;
  g.as[0] = g.r2[0] = g.rs[0] = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int cak = (h[2 + (k >> 2)] >> (8 * (k & 3))) & 0xff, c2k = (h[4 + (k >> 1)] >> (16 * (k & 1))) & 0xffff;
    g.as[k + 1] = g.as[k] + cak;
    g.r2[k + 1] = g.r2[k] + c2k;
    g.rs[k + 1] = g.rs[k] + cak + c2k;
  }
  return g;
}
__device__ __forceinline__ void store_starts(const AevParams& p, const Groups& g, int nrad, int nang, int lane, FastLds& L) {
  int vr = nrad, va = nang;   // entries S.. of the start tables hold the totals
#pragma unroll
  for (int k = 7; k >= 0; k--)
    if (lane == k && k < p.S) { vr = g.rs[k]; va = g.as[k]; }
  if (lane <= p.S) { L.rstart[lane] = vr; L.astart[lane] = va; }
}

// Forward: fill the wave's LDS lists from the prefetched compact list: the radial list {r, fc} grouped by species (inside
// a group the angular neighbours first), the angular list (the angular region as it stands, already in species order),
// group starts in L.rstart / L.astart.  All per-species bookkeeping is scalar.
template <int NCH>
__device__ __forceinline__ void unpack_lists(const AevParams& p, const AevArgs& a, int row, const hdr_t& h,
                                             const Loaded<NCH, false, 1>& ld, int lane, FastLds& L, int& nrad, int& nang) {
  nrad = hdr_nrad(h);
  nang = hdr_nang(h);
  const Groups g = unpack_groups(h);
  store_starts(p, g, nrad, nang, lane, L);
  const float half_inv_Rcr = 0.5f * p.pi_over_Rcr * 0.3183098861837907f;  // r/(2 Rcr) revolutions
  const float half_inv_Rca = 0.5f * p.pi_over_Rca * 0.3183098861837907f;
  const float4* __restrict__ px = a.cl_xyz + (size_t)row * a.cl_stride;
  for (int base = 0; base < nrad; base += 64) {
    const int c = base >> 6, t = base + lane;
    float4 v = make_float4(0.f, 0.f, 0.f, 1.f);
    if (c < NCH) {
#pragma unroll
      for (int k = 0; k < NCH; k++)
        if (c == k) v = ld.xx[k];
    } else if (t < nrad) {   // lists longer than the chunks held in registers: loaded in place
      v = px[slot_index(t, nang)];
    }
    if (t < nrad) {
      // w-th entry of the angular (ang) or the radial-only stream -> position in the LDS radial list
      const bool ang = t < nang;
      const int w = ang ? t : t - nang;
      int pos = w;   // species 0: the radial-only entries follow its as[1] angular ones
      int lead = g.as[1];
#pragma unroll
      for (int k = 1; k < 8; k++)
        if (k < p.S) {
          const int st = ang ? g.as[k] : g.r2[k];
          if (w >= st) { pos = g.rs[k] + (w - st); lead = g.as[k + 1] - g.as[k]; }
        }
      if (!ang) pos += lead;
      L.rr[pos] = v.w;
      L.rfc[pos] = 0.5f * fcos_rev(v.w * half_inv_Rcr) + 0.5f;
      if (ang) {
        L.ad[w] = v;
        L.afc[w] = 0.5f * fcos_rev(v.w * half_inv_Rca) + 0.5f;
      }
    }
  }
}

// Build the table of non-empty species-pair buckets.  Returns the padded stream length; nbk = number of entries.
// entry: {tstart, a1, n1, a2, n2, outoff, tri, npairs}
template <int NA, int NZ>
__device__ __forceinline__ int build_bucket_table(const AevParams& p, int lane, FastLds& L, int& nbk) {
  constexpr int Q = 64 / NA;
  const int nb_all = p.S * (p.S + 1) / 2;
  int s1 = 0, s2 = 0, np = 0, n1 = 0, n2 = 0;
  if (lane < nb_all) {
    int rem = lane;
    while (rem >= p.S - s1) { rem -= p.S - s1; s1++; }
    s2 = s1 + rem;
    n1 = L.astart[s1 + 1] - L.astart[s1];
    n2 = L.astart[s2 + 1] - L.astart[s2];
    np = (s1 == s2) ? n1 * (n1 - 1) / 2 : n1 * n2;
  }
  const int npad = (np + Q - 1) / Q * Q;
  const int incl = wave_incl_scan(npad);
  const unsigned long long m = __ballot(np > 0);
  if (np > 0) {
    int* e = L.tb + 8 * lanes_below(m);
    e[0] = incl - npad; e[1] = L.astart[s1]; e[2] = n1; e[3] = L.astart[s2]; e[4] = n2;
    e[5] = p.radial_len + lane * (NA * NZ); e[6] = (s1 == s2) ? 1 : 0; e[7] = np;
  }
  nbk = __popcll(m);
  return __builtin_amdgcn_readlane(incl, 63);
}

// lane -> its pair of the padded stream.  valid = false for padding slots (indices then point at neighbour 0 of
// the group, a harmless geometry).
__device__ __forceinline__ void stream_pair(const FastLds& L, int nbk, int t, int& ia, int& ib, int& outoff, bool& valid) {
  int e = 0;
  for (int k = 1; k < nbk; k++)
    if (t >= L.tb[8 * k]) e = k;
  const int4 e0 = *reinterpret_cast<const int4*>(L.tb + 8 * e);
  const int4 e1 = *reinterpret_cast<const int4*>(L.tb + 8 * e + 4);
  const int u = t - e0.x;
  valid = u < e1.w;
  outoff = e1.y;
  int a = 0, b = 0;
  if (valid) {
    if (e1.z) {
      decode_pair(u, e0.z, a, b);
    } else {
      a = (int)(((float)u + 0.5f) * frcp((float)e1.x));
      b = u - a * e1.x;
    }
  } else if (e1.z) {
    b = 1;  // distinct neighbours keep the padded geometry finite
  }
  ia = e0.y + a;
  ib = e0.w + b;
}

// Backward, dense pair stream in two half-waves.  Lanes 0..31 and 32..63 walk the SAME sequence of 32 "column slots" per
// step, the lower half on the even row neighbours of a bucket, the upper half on the odd ones: lane l and lane l + 32 hold
// pairs (a, b) and (a + 1, b) with the same column neighbour b, so a column's gradient leaves the step as ONE LDS add per two
// pairs (the two halves are summed with a half-wave swap; LDS float adds cost about a cycle per active lane, and at one add
// per pair they made the LDS, not the vector unit, the busiest part of the CU).  Rows come in BANDS of two: a rectangular
// bucket (n1 rows x n2 columns; the larger species group is the columns) has ceil(n1 / 2) bands of n2 slots (an odd last row
// leaves its upper half idle); a triangular one (pairs a < b of n neighbours) has bands k = 0 .. n/2 - 1 of rows 2k, 2k + 1
// with n - 1 - 2k slots, column b = 2k + 1 + slot -- the upper half idles at slot 0, where its row meets itself -- floor(n^2/4)
// slots in all.  A water centre: 36 + 36 + 9 = 81 slots = 3 steps for its 153 pairs.
// Table entry: {vstart, ra, nr, ca, nc, outoff, tri, slots}; returns the number of slots.
template <int NA, int NZ>
__device__ __forceinline__ int build_pair_table(const AevParams& p, int lane, FastLds& L, int& nbk) {
  const int nb_all = p.S * (p.S + 1) / 2;
  int s1 = 0, s2 = 0, ns = 0, n1 = 0, n2 = 0;
  bool sw = false;
  if (lane < nb_all) {
    int rem = lane;
    while (rem >= p.S - s1) { rem -= p.S - s1; s1++; }
    s2 = s1 + rem;
    n1 = L.astart[s1 + 1] - L.astart[s1];
    n2 = L.astart[s2 + 1] - L.astart[s2];
    if (s1 == s2) {
      ns = (n1 * n1) >> 2;                       // 0 for n1 < 2
    } else if (n1 > 0 && n2 > 0) {
      sw = n2 < n1;                              // rows = the smaller group
      const int nr = sw ? n2 : n1, nc = sw ? n1 : n2;
      ns = ((nr + 1) >> 1) * nc;
    }
  }
  const int incl = wave_incl_scan(ns);
  const unsigned long long m = __ballot(ns > 0);
  if (ns > 0) {
    int* e = L.tb + 8 * lanes_below(m);
    e[0] = incl - ns; e[1] = L.astart[sw ? s2 : s1]; e[2] = sw ? n2 : n1; e[3] = L.astart[sw ? s1 : s2]; e[4] = sw ? n1 : n2;
    e[5] = p.radial_len + lane * (NA * NZ); e[6] = (s1 == s2) ? 1 : 0; e[7] = ns;
  }
  nbk = __popcll(m);
  return __builtin_amdgcn_readlane(incl, 63);
}
// lane -> its pair: slot v of the stream, half h (0: the band's even row, 1: its odd row).  pos = the pair's position in its
// run (the consecutive lanes of a half-wave that share the row neighbour ia).  Idle lanes (past the end, an odd last row's
// upper half, the upper half at a triangular band's first slot): valid = false, pos = 0, and the indices of a harmless pair.
__device__ __forceinline__ void stream_pair_half(const FastLds& L, int nbk, int v, int h, int& ia, int& ib, int& outoff, int& pos,
                                                 bool& valid) {
  int e = 0;
  for (int k = 1; k < nbk; k++)
    if (v >= L.tb[8 * k]) e = k;
  const int4 e0 = *reinterpret_cast<const int4*>(L.tb + 8 * e);       // vstart, ra, nr, ca
  const int4 e1 = *reinterpret_cast<const int4*>(L.tb + 8 * e + 4);   // nc, outoff, tri, slots
  const int u = v - e0.x;
  outoff = e1.y;
  int a = 0, b = 0;
  pos = 0;
  valid = nbk > 0 && u < e1.w;
  if (valid) {
    if (e1.z) {
      // band k: the largest k with k (n - k) <= u; closed form + one correction step each way (all quantities < 2^14)
      const int n = e0.z;
      const float fn = (float)n;
      int k = (int)((fn - __builtin_amdgcn_sqrtf(fmaxf(fn * fn - 4.f * (float)u, 0.f))) * 0.5f);
      k = max(0, min(k, (n >> 1) - 1));
      if (__mul24(k, n - k) > u) k--;
      if (__mul24(k + 1, n - k - 1) <= u) k++;
      const int c = u - __mul24(k, n - k);
      a = 2 * k + h;
      b = 2 * k + 1 + c;
      pos = c - h;
      valid = h == 0 || c >= 1;
    } else {
      const int k = (int)(((float)u + 0.5f) * frcp((float)e1.x));
      const int c = u - k * e1.x;
      a = 2 * k + h;
      b = c;
      pos = c;
      valid = a < e0.z;
    }
  }
  if (!valid) { a = 0; b = e1.z ? 1 : 0; pos = 0; }   // two distinct neighbours of a non-empty bucket (a triangular one has n >= 2)
  ia = e0.y + a;
  ib = e0.w + b;
}

#ifdef ABL_NO_TWRITE
#define TILE_ADD(p, v) asm volatile("" ::"v"(p), "v"(v))
#else
#define TILE_ADD(p, v) atomicAdd(p, (gd_t)(v))
#endif
// v + v[lane ^ OFF] without an LDS round trip: OFF = 1, 2, 4, 8 as DPP moves inside a row of 16 lanes, OFF = 16 / 32
// with the gfx950 row / half-wave swaps (both operands the same register: one result holds the even rows or the lower
// half everywhere, the other the odd rows or the upper half)
template <int OFF>
__device__ __forceinline__ float xor_sum(float v) {
  const int x = __float_as_int(v);
  if constexpr (OFF == 1) {
    return v + __int_as_float(__builtin_amdgcn_mov_dpp(x, 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  } else if constexpr (OFF == 2) {
    return v + __int_as_float(__builtin_amdgcn_mov_dpp(x, 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  } else if constexpr (OFF == 4) {
    int y = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xf, 0x5, false);  // row_shl:4 into banks 0,2
    y = __builtin_amdgcn_update_dpp(y, x, 0x114, 0xf, 0xa, false);      // row_shr:4 into banks 1,3
    return v + __int_as_float(y);
  } else if constexpr (OFF == 8) {
    return v + __int_as_float(__builtin_amdgcn_mov_dpp(x, 0x128, 0xf, 0xf, true));  // row_ror:8
  } else if constexpr (OFF == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  } else {
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
}
// Backward enumeration.  Every non-empty species-pair bucket is laid out as ROWS of kRowW lanes (4: one DPP quad):
//   s1 != s2 : row = neighbour of one species, column = neighbour of the other, oriented so that the bucket needs the
//              fewer rows of the stream;
//   s1 == s2 : the strict upper triangle of n x n folded into nn/2 rows x nn columns (nn = n rounded up to even):
//              lane (r, c) holds the pair (r, c) if c > r and the pair (nn-1-r, nn-1-c) if c < r;
//   more than kRowW columns: the row is cut into column blocks of kRowW, each its own row of the stream.
// The rows of all buckets form one stream; a step of the pair loop takes 64 / kRowW consecutive rows (64 lanes),
// whatever buckets they belong to.  A pair's gradient with respect to its ROW neighbour is summed over the row with
// DPP adds (no LDS), lane 0 of the row adds the sum to that neighbour's LDS accumulator; the gradient with respect to
// the COLUMN neighbour goes to its accumulator with one LDS float add per lane.  Rows were 16 lanes wide at first: with
// the 5..12 neighbours per species of a typical centre that left 40 % of the lanes of a step idle (water: 153 pairs in
// 4 steps of 64); quads waste at most three lanes per row (the same 153 pairs in 3 steps), and the arithmetic of the
// pairs, not the bookkeeping around it, is what this pass spends its time on.  (The very first version issued six LDS
// float atomics per pair with ~10-way address conflicts, which cost more than all of its arithmetic.)
// entry: {row0, ra, nr, ca, nc, outoff, tri, column blocks}.  Returns the number of rows in the stream.
template <int NA, int NZ>
__device__ __forceinline__ int build_row_table(const AevParams& p, int lane, FastLds& L, int& nbk) {
  const int nb_all = p.S * (p.S + 1) / 2;
  int s1 = 0, s2 = 0, ra = 0, nr = 0, ca = 0, nc = 0, nrow = 0, ncb = 1, tri = 0;
  if (lane < nb_all) {
    int rem = lane;
    while (rem >= p.S - s1) { rem -= p.S - s1; s1++; }
    s2 = s1 + rem;
    const int a1 = L.astart[s1], a2 = L.astart[s2];
    const int n1 = L.astart[s1 + 1] - a1, n2 = L.astart[s2 + 1] - a2;
    int nrows = 0, ncols = 0;
    if (s1 == s2) {
      tri = 1;
      ra = ca = a1; nr = nc = n1;
      const int nn = (n1 + 1) & ~1;
      if (n1 >= 2) { nrows = nn >> 1; ncols = nn; }
    } else if (n1 > 0 && n2 > 0) {
      // the orientation that takes fewer rows of kRowW lanes
      const bool sw = n2 * ((n1 + kRowW - 1) / kRowW) < n1 * ((n2 + kRowW - 1) / kRowW);
      ra = sw ? a2 : a1; nr = sw ? n2 : n1;
      ca = sw ? a1 : a2; nc = sw ? n1 : n2;
      nrows = nr; ncols = nc;
    }
    if (nrows > 0) {
      ncb = (ncols + kRowW - 1) / kRowW;
      nrow = nrows * ncb;
    }
  }
  const int incl = wave_incl_scan(nrow);
  const unsigned long long m = __ballot(nrow > 0);
  if (nrow > 0) {
    int* e = L.tb + 8 * lanes_below(m);
    e[0] = incl - nrow; e[1] = ra; e[2] = nr; e[3] = ca; e[4] = nc;
    e[5] = p.radial_len + lane * (NA * NZ); e[6] = tri; e[7] = ncb;
  }
  nbk = __popcll(m);
  return __builtin_amdgcn_readlane(incl, 63);
}

// Descriptors of rows RB .. RB+31 of the backward pair stream, one lane per row, so that the per-pair lanes of a step
// (4 rows x 16 lanes) read what they need about their row with two 16-byte LDS loads instead of each repeating the
// bucket search, the division by the column-block count and the fold arithmetic (that prologue was ~90 of the ~480
// vector instructions of a step):
//   d0 = {row neighbour of the pairs above the diagonal (all pairs of a rectangular bucket), row neighbour of the folded
//         pairs below it (-1: none), first column neighbour of the bucket, number of columns}
//   d1 = {first column of this row's block, r | tri << 16 | valid << 17, nn (columns of the folded triangle), offset of
//         the bucket's dE/dAEV block}
constexpr int kRowBlock = 32;   // rows of the pair stream whose descriptors are held at a time
__device__ __forceinline__ void build_row_descriptors(FastLds& L, int nbk, int nrows_stream, int RB, int lane) {
  const int R = RB + lane;
  int4 d0 = make_int4(0, -1, 0, 0), d1 = make_int4(0, 0, 0, 0);
  if (R < nrows_stream) {
    int e = 0;
    for (int k = 1; k < nbk; k++)
      if (R >= L.tb[8 * k]) e = k;
    const int4 e0 = *reinterpret_cast<const int4*>(L.tb + 8 * e);       // row0, ra, nr, ca
    const int4 e1 = *reinterpret_cast<const int4*>(L.tb + 8 * e + 4);   // nc, outoff, tri, ncb
    const int u = R - e0.x;
    const int r = e1.w > 1 ? u / e1.w : u;
    const int nn = (e0.z + 1) & ~1;
    const int rb = nn - 1 - r;
    d0 = make_int4(e0.y + r, (e1.z && rb < e0.z) ? e0.y + rb : -1, e0.w, e1.x);
    d1 = make_int4((u - r * e1.w) * kRowW, r | (e1.z ? 1 << 16 : 0) | (1 << 17), nn, e1.y);
  }
  if (lane < kRowBlock) {
    L.rowd[2 * lane] = d0;
    L.rowd[2 * lane + 1] = d1;
  }
}

// forward_compute: the AEV row of one centre from the wave's LDS lists (radial list grouped by species, angular list, group
// starts) -- filled from the compact list by forward_centre, or straight from the candidate list by the fused kernel
template <int NA, int NZ>
__device__ __forceinline__ void forward_compute(const AevParams& p, const AevArgs& a, FastLds& L, int row, int lane BWD_STAMP_PARAMS);

template <int NA, int NZ, int NCH>
__device__ __forceinline__ void forward_centre(const AevParams& p, const AevArgs& a, FastLds& L, int row, const hdr_t& h,
                                               const Loaded<NCH, false, 1>& pf, int lane BWD_STAMP_PARAMS) {
  for (int e = lane; e < (p.aev_stride >> 2); e += 64) reinterpret_cast<float4*>(L.row)[e] = make_float4(0, 0, 0, 0);
  int nrad, nang;
  unpack_lists<NCH>(p, a, row, h, pf, lane, L, nrad, nang);
  wave_sync();
  BWD_STAMP(2);   // unpack
  forward_compute<NA, NZ>(p, a, L, row, lane BWD_STAMP_ARGS);
}

template <int NA, int NZ>
__device__ __forceinline__ void forward_compute(const AevParams& p, const AevArgs& a, FastLds& L, int row, int lane BWD_STAMP_PARAMS) {
  constexpr int NR = 16, Q = 64 / NA;

  // ---- radial: per species group, lanes = (slot q, shift k) ----
#ifndef ABLF_NO_RAD
  {
    const int q = lane >> 4, k = lane & 15;
    const float shf = fmaf((float)k, p.dShfR, p.ShfR0);
    const float c = -p.EtaR * kLog2e;
    for (int s = 0; s < p.S; s++) {
      const int b0 = L.rstart[s], b1 = L.rstart[s + 1];
      if (b1 <= b0) continue;
      float acc = 0.f, acc2 = 0.f;
      int t = b0 + q;
      for (; t + 4 < b1; t += 8) {   // two neighbours per trip: their LDS reads and exponentials overlap
        const float r0 = L.rr[t], r1 = L.rr[t + 4], f0 = L.rfc[t], f1 = L.rfc[t + 4];
        const float d0 = r0 - shf, d1 = r1 - shf;
        acc = fmaf(fexp2(c * d0 * d0), f0, acc);
        acc2 = fmaf(fexp2(c * d1 * d1), f1, acc2);
      }
      if (t < b1) {
        const float dr = L.rr[t] - shf;
        acc = fmaf(fexp2(c * dr * dr), L.rfc[t], acc);
      }
      acc = xor_sum<32>(xor_sum<16>(acc + acc2));   // row / half-wave swaps, no LDS round trip
      if (q == 0) L.row[s * NR + k] = 0.25f * acc;
    }
  }
#endif

  BWD_STAMP(3);   // radial
  // ---- angular ----
  int nbk;
  const int total = build_bucket_table<NA, NZ>(p, lane, L, nbk);
  wave_sync();
  BWD_STAMP(4);   // bucket table
  float acc[NZ];
#pragma unroll
  for (int z = 0; z < NZ; z++) acc[z] = 0.f;
  int cur_off = -1;
  const int la = lane % NA, lq = lane / NA;
  const float cA = -p.EtaA * kLog2e;

  // A bucket's sums over the 64 / NA slot lanes of each shift.  Cross-lane moves are the expensive vector instructions here
  // (tools/issue_probe.hip: a half-wave or row swap takes the issue time of ~5 fp32 FMAs, a DPP move of ~3), so the NZ values
  // are not reduced one by one (NZ x 3..4 levels): each swap level HALVES the number of values a lane carries -- the pair
  // (acc[i], acc[i + NZ/2]) goes through ONE v_permlane32_swap whose two results add up to the lower half-wave's total of
  // acc[i] in the lower lanes and the upper half-wave's total of acc[i + NZ/2] in the upper ones; the same with the row swap
  // -- and the lane bits above the shift index end up selecting the section: lane = (z, [spare bit,] shift).
  auto flush = [&]() {
    if (cur_off >= 0) {
      float t[NZ / 2];
#pragma unroll
      for (int i = 0; i < NZ / 2; i++) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i]), __float_as_uint(acc[i + NZ / 2]), false, false);
        t[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
      }
      float u[NZ / 4];
#pragma unroll
      for (int i = 0; i < NZ / 4; i++) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(t[i]), __float_as_uint(t[i + NZ / 4]), false, false);
        u[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
      }
      float v;
      if constexpr (NA == 8) {        // NZ = 4: lane = (z: 2 bits, spare bit 3, shift: 3 bits)
        v = xor_sum<8>(u[0]);
      } else {                        // NA = 4, NZ = 8: lane = (z: 3 bits, spare bit 2, shift: 2 bits)
        const bool odd = lane & 8;
        const float keep = odd ? u[1] : u[0], send = odd ? u[0] : u[1];
        v = keep + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send), 0x128, 0xf, 0xf, true));   // row_ror:8
        v = xor_sum<4>(v);
      }
      if ((lane & NA) == 0) L.row[cur_off + la * NZ + (lane >> (NA == 8 ? 4 : 3))] = v;
    }
#pragma unroll
    for (int z = 0; z < NZ; z++) acc[z] = 0.f;
  };

#ifdef ABLF_NO_ANG
  for (int base = 0; base < 0; base += 64) {
#else
  for (int base = 0; base < total; base += 64) {
#endif
    // phase 1: lane = pair
    int outoff;
    {
      int ia, ib;
      bool valid;
      stream_pair(L, nbk, base + lane, ia, ib, outoff, valid);
      const float4 A = L.ad[ia], B = L.ad[ib];
      const float dot = A.x * B.x + A.y * B.y + A.z * B.z;
      const float cth = 0.95f * dot * frcp(fmaxf(A.w * B.w, 1e-10f));
      const float sth = __builtin_amdgcn_sqrtf(fmaxf(1.f - cth * cth, 0.f));
      const float w = valid ? 2.f * L.afc[ia] * L.afc[ib] : 0.f;
      const float rho = 0.5f * (A.w + B.w);
      // 16-byte stores (a lane's NZ / NA factors are contiguous): scalar stores at a 16- or 32-byte lane stride would
      // be 4- and 8-way bank conflicts
      float t1[NZ], t2[NA];
#pragma unroll
      for (int z = 0; z < NZ; z++) {
        const float basez = fmaxf(0.5f * (1.f + cth * p.cosZ[z] + sth * p.sinZ[z]), 0.f);
        t1[z] = w * fexp2(p.Zeta * flog2(basez));
      }
      float dr = rho - p.ShfA0;   // equidistant shifts: one subtraction per step instead of a table of NA scalars
#pragma unroll
      for (int s = 0; s < NA; s++) {
        t2[s] = fexp2(cA * dr * dr);
        dr -= p.dShfA;
      }
#pragma unroll
      for (int z = 0; z < NZ; z += 4)
        *reinterpret_cast<float4*>(L.pf1 + lane * NZ + z) = make_float4(t1[z], t1[z + 1], t1[z + 2], t1[z + 3]);
#pragma unroll
      for (int s = 0; s < NA; s += 4)
        *reinterpret_cast<float4*>(L.pf2 + lane * NA + s) = make_float4(t2[s], t2[s + 1], t2[s + 2], t2[s + 3]);
    }
    wave_sync();
    // phase 2: lane = (slot lq, shift la); group g covers slots g*Q .. g*Q+Q-1, all of one bucket (its output offset is
    // read from the register of the group's first pair: no LDS round trip in front of the flush test).  The factors of
    // group g+1 are requested before group g is accumulated.
    const int ngroups = min(NA, (total - base + Q - 1) / Q);
    float f2n = L.pf2[lq * NA + la];
    float4 qn[NZ / 4];
#pragma unroll
    for (int z4 = 0; z4 < NZ / 4; z4++) qn[z4] = *reinterpret_cast<const float4*>(L.pf1 + lq * NZ + 4 * z4);
#pragma unroll
    for (int g = 0; g < NA; g++) {
      if (g < ngroups) {   // wave-uniform
        const float f2 = f2n;
        float4 q[NZ / 4];
#pragma unroll
        for (int z4 = 0; z4 < NZ / 4; z4++) q[z4] = qn[z4];
        if (g + 1 < NA && g + 1 < ngroups) {
          const int slot = (g + 1) * Q + lq;
          f2n = L.pf2[slot * NA + la];
#pragma unroll
          for (int z4 = 0; z4 < NZ / 4; z4++) qn[z4] = *reinterpret_cast<const float4*>(L.pf1 + slot * NZ + 4 * z4);
        }
        const int off = __builtin_amdgcn_readlane(outoff, g * Q);
        if (off != cur_off) { flush(); cur_off = off; }
#pragma unroll
        for (int z4 = 0; z4 < NZ / 4; z4++) {
          acc[4 * z4] = fmaf(f2, q[z4].x, acc[4 * z4]); acc[4 * z4 + 1] = fmaf(f2, q[z4].y, acc[4 * z4 + 1]);
          acc[4 * z4 + 2] = fmaf(f2, q[z4].z, acc[4 * z4 + 2]); acc[4 * z4 + 3] = fmaf(f2, q[z4].w, acc[4 * z4 + 3]);
        }
      }
    }
    wave_sync();
  }
  flush();
  wave_sync();
  BWD_STAMP(5);   // angular

  float4* dst = reinterpret_cast<float4*>(a.aev + (long long)row * p.aev_stride);
  const int n4 = p.aev_stride >> 2;
  for (int e = lane; e < n4; e += 64) dst[e] = reinterpret_cast<const float4*>(L.row)[e];
  wave_sync();  // the LDS slice is reused by this wave's next centre
  BWD_STAMP(6);   // row store
}

// Persistent waves: wave w handles rows w, w + W, w + 2W, ...  A centre's inputs (header, list chunks, dE/dAEV row: all
// addressed by the row alone) are requested when the centre is started and waited for on the spot: the kernels are
// bound by what their resident waves execute, not by one wave's latency, so the other waves cover the wait.  Requesting
// them one centre ahead (two register stages used alternately, secured before the centre's stores and atomics entered
// the in-order vmcnt queue) was built and measured: forward unchanged, backward 4 % SLOWER -- the second stage costs
// 22 registers that the pair loop uses better.
// Which rows a wave takes: by ticket when the launch has counters (AevArgs::row_counter) -- centres cost differently, and with a
// fixed stride a kernel ends with its unluckiest wave (forward pass with the compaction inside: 0.252 -> 0.224 ms) -- else a
// fixed stride.  The workgroups form kTicketGroups groups (blockIdx mod groups), each with a contiguous share of the rows and
// a counter of its own (ONE counter would serialise 100 000 same-address atomics: measured 1.3 ms).  A wave holds the tickets
// of its current and its next centre and has two more on their way: the one drawn a centre ago is taken behind the centre's
// first wait for loaded data, at no cost (the memory queue is in order).  A group's last wave out leaves the counters at zero for the next launch on the stream.
#define ANI_TICKETS_BEGIN(KW)                                                                              \
  const int nw = gridDim.x * KW;                                                                          \
  const int tk_ngrp = min((int)gridDim.x, kTicketGroups), tk_grp = blockIdx.x % tk_ngrp;                  \
  const int tk_share = (a.kcount + tk_ngrp - 1) / tk_ngrp, tk_base = tk_grp * tk_share;                   \
  int* const tk_ctr = a.row_counter ? a.row_counter + kTicketStride * tk_grp : nullptr;                   \
  const int kend = tk_ctr ? min(a.kcount, tk_base + tk_share) : a.kcount;                                 \
  int k = blockIdx.x * KW + wave, k1 = k + nw, tk_pending = 0;                                            \
  if (tk_ctr) {                                                                                           \
    int t0 = 0, t1 = 0;                                                                                   \
    if (lane == 0) { t0 = atomicAdd(tk_ctr, 1); t1 = atomicAdd(tk_ctr, 1); tk_pending = atomicAdd(tk_ctr, 1); } \
    k = tk_base + __builtin_amdgcn_readfirstlane(t0);                                                     \
    k1 = tk_base + __builtin_amdgcn_readfirstlane(t1);                                                    \
  }
/* at the top of a centre: the ticket three centres ahead leaves ... */
#define ANI_TICKET_DRAW(tn)                                                                                \
  int tn = 0;                                                                                             \
  if (tk_ctr && lane == 0) tn = atomicAdd(tk_ctr, 1);
/* ... and behind the centre's first wait for loaded data (which, the memory queue being in order, is also a wait for everything
   the wave issued before): the ticket drawn a centre ago is taken into a scalar register */
#define ANI_TICKET_TAKE(k2)                                                                                \
  touch(tk_pending);                                                                                      \
  const int k2 = tk_base + __builtin_amdgcn_readfirstlane(tk_pending);
#define ANI_TICKET_NEXT(tn, k2)                                                                            \
  k = k1;                                                                                                 \
  k1 = tk_ctr ? k2 : k1 + nw;                                                                             \
  tk_pending = tn;
#define ANI_TICKETS_END(KW)                                                                                \
  if (tk_ctr && lane == 0 && atomicAdd(tk_ctr + 1, 1) == ((int)gridDim.x - tk_grp + tk_ngrp - 1) / tk_ngrp * KW - 1) { \
    __atomic_store_n(tk_ctr, 0, __ATOMIC_RELAXED);                                                        \
    __atomic_store_n(tk_ctr + 1, 0, __ATOMIC_RELAXED);                                                    \
  }
#define ANI_PERSISTENT_LOOP(KW, NCH, BWD, GR, CENTRE)                                                      \
  ANI_TICKETS_BEGIN(KW)                                                                                   \
  while (k < ((tk_ctr && BWD) ? tk_base + tk_share : kend)) {                                             \
    ANI_TICKET_DRAW(tk_tn)                                                                                \
    const int kk = (tk_ctr && BWD) ? (k - tk_base) * tk_ngrp + tk_grp : k;   /* backward: a group's rows interleaved with the others' */ \
    const int kc = kk < a.kcount ? kk : a.kcount - 1;                                                     \
    const int row = a.row_list ? __builtin_amdgcn_readfirstlane(a.row_list[a.k0 + kc]) : a.k0 + kc;       \
    const hdr_t hc = load_header(a, row);                                                                 \
    ANI_TICKET_TAKE(tk_k2)                                                                                \
    if (hc[0] >= 0 && kk < a.kcount) {                                                                    \
      Loaded<NCH, BWD, GR> cur;                                                                           \
      load_lists(p, a, row, hdr_nrad(hc), hdr_nang(hc), lane, cur);                                       \
      CENTRE;                                                                                             \
    }                                                                                                     \
    ANI_TICKET_NEXT(tk_tn, tk_k2)                                                                         \
  }                                                                                                       \
  ANI_TICKETS_END(KW)

template <int NA, int NZ, int NCH>
__global__ __launch_bounds__(64 * kWaves, ANI_FWD_MINW) void aev_forward_fast(AevParams p, AevArgs a, int cap, int rowf) {
  extern __shared__ float4 smem4[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  FastLds L = carve<NA, NZ>(reinterpret_cast<float*>(smem4) + wave * fast_wave_floats_row(cap, false, rowf, p.S), cap, false, rowf);
#ifdef ABLB_STAMPS
  unsigned long long stamp_prev, stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
  const int nw = gridDim.x * kWaves;
  for (int k = blockIdx.x * kWaves + wave; k < a.kcount; k += nw) {
    const int row = a.row_list ? __builtin_amdgcn_readfirstlane(a.row_list[a.k0 + k]) : a.k0 + k;
    const hdr_t hc = load_header(a, row);
    BWD_STAMP(0);
    if (hc[0] < 0) continue;
    Loaded<NCH, false, 1> cur;
    load_lists(p, a, row, hdr_nrad(hc), hdr_nang(hc), lane, cur);
    BWD_STAMP(1);
    forward_centre<NA, NZ, NCH>(p, a, L, row, hc, cur, lane, stamp_prev, stamp_acc);
    stamp_acc[7] += 1;
  }
  if (wave == 0 && lane == 0) {
    for (int q = 0; q < 7; q++) atomicAdd(&g_bwd_stamps[16 + q], stamp_acc[q]);
    atomicAdd(&g_bwd_stamps[24], stamp_acc[7]);
  }
#else
  ANI_PERSISTENT_LOOP(kWaves, NCH, false, 1, (forward_centre<NA, NZ, NCH>(p, a, L, row, hc, cur, lane)))
#endif
}

// ---- the forward pass with the per-step compaction inside ("aev_fused") ----------------------------------------------------
// nbr_compact_kernel is bound by latency (list -> positions -> stores, little arithmetic: 0.09 ms at 100 000 centres at a third
// of the memory system's rate), aev_forward_fast by vector issue; run one after the other, each leaves the other's resource
// idle.  Here the wave that computes a centre's AEV row screens its candidate list itself: the gathers of the centre in hand
// are covered by the arithmetic of the five other waves of the SIMD, the descriptor and the candidate indices of the wave's
// NEXT centre travel during the arithmetic of this one (three registers), the LDS lists are filled straight from the
// screened candidates (no compact list read back, no unpack from it), and the compact list and its header are still
// written -- the backward kernel starts from them.  Candidate lists longer than 64 NCHC entries, or AEV shapes off the fast
// path, take the two kernels.
template <int NA, int NZ, int NCHC>
__global__ __launch_bounds__(64 * kWaves, ANI_FUSED_MINW) void aev_forward_fused(AevParams p, AevArgs a, int cap, int rowf) {
  extern __shared__ float4 smem4[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  FastLds L = carve<NA, NZ>(reinterpret_cast<float*>(smem4) + wave * fast_wave_floats_row(cap, false, rowf, p.S), cap, false, rowf);
  ANI_TICKETS_BEGIN(kWaves)   // rows by ticket (see ANI_PERSISTENT_LOOP)
  const int cap2 = a.cl_stride - kMaxAng;   // room of the radial-only stream
  const float half_inv_Rcr = 0.5f * p.pi_over_Rcr * 0.3183098861837907f;  // r/(2 Rcr) revolutions
  const float half_inv_Rca = 0.5f * p.pi_over_Rca * 0.3183098861837907f;
  int4 info = load_info(a, k);
  int jj[NCHC];
  load_j(a, info, lane, jj);
  while (k < kend) {   // a group's rows are contiguous here (interleaved with the other groups' they run no faster or slower)
    ANI_TICKET_DRAW(tn)
    const int4 info1 = load_info(a, k1);       // in flight beside the gathers
    float4 xi, xx[NCHC];
    gather_x(a, info, jj, xi, xx);
    int jj1[NCHC];
    load_j(a, info1, lane, jj1);               // needs info1 only: the gathers stay in flight
    const int4 inf = uniform4(info);
    const int row = inf.w, i = inf.x, n = inf.x < 0 ? 0 : inf.z;
    for (int e = lane; e < (p.aev_stride >> 2); e += 64) reinterpret_cast<float4*>(L.row)[e] = make_float4(0, 0, 0, 0);
    int4* hdr = a.cl_hdr + 2 * (size_t)row;
    float4* oxyz = a.cl_xyz + (size_t)row * a.cl_stride;
    int* oj = a.cl_j + (size_t)row * a.cl_stride;
    // ---- screen (as nbr_compact_kernel): two append-only streams, per-species counts in lane s ----
    int nA = 0, nR = 0;
    int cc = 0;      // lane s: entries of species s in the angular stream (low half) and in the radial-only stream (high half)
    int spc[NCHC];   // the candidate's (compact) species
    float4 dd[NCHC];
    int pos[NCHC];   // slot in the row's compact list: [0, kMaxAng) angular stream, kMaxAng + .. radial-only stream, -1 screened out
#pragma unroll
    for (int c = 0; c < NCHC; c++) {
      pos[c] = -1;
      spc[c] = 0;
      dd[c] = make_float4(0.f, 0.f, 0.f, 1.f);
      if (64 * c < n) {   // wave-uniform
        const bool valid = 64 * c + lane < n;
        const float4 xj = xx[c];
        float4 d;
        d.x = xj.x - xi.x; d.y = xj.y - xi.y; d.z = xj.z - xi.z;
        d.w = __builtin_amdgcn_sqrtf(d.x * d.x + d.y * d.y + d.z * d.z);
        const int sp = __float_as_int(xj.w);
        const bool in_a = valid && d.w <= p.Rca;
        const bool in_2 = valid && !in_a && (p.compat || d.w <= p.Rcr);
        const unsigned long long mA = __ballot(in_a), m2 = __ballot(in_2);
        const int pA = nA + lanes_below(mA), p2 = nR + lanes_below(m2);
        for (int s = 0; s < p.S; s++) {
          const unsigned long long ms = __ballot(sp == s);
          const int k12 = __popcll(mA & ms) | (__popcll(m2 & ms) << 16);
          if (lane == s) cc += k12;
        }
        nA += __popcll(mA);
        nR += __popcll(m2);
        spc[c] = sp;
        dd[c] = d;
        pos[c] = (in_a && pA < kMaxAng) ? pA : ((in_2 && p2 < cap2) ? kMaxAng + p2 : -1);
      }
    }
    // the next centre's candidate indices are in registers before this centre's stores enter the (in-order) memory queue
#pragma unroll
    for (int c = 0; c < NCHC; c++) touch(jj1[c]);
    ANI_TICKET_TAKE(k2)
    const bool over = nA > kMaxAng || nR > cap2 || nA + nR > cap;   // never a silent truncation: row skipped, flag raised
#pragma unroll
    for (int c = 0; c < NCHC; c++)
      if (pos[c] >= 0) { oxyz[pos[c]] = dd[c]; oj[pos[c]] = cl_pack(jj[c], spc[c]); }
    if (lane < 8) {
      reinterpret_cast<unsigned char*>(hdr)[8 + lane] = (unsigned char)(cc & 0xffff);
      reinterpret_cast<unsigned short*>(hdr)[8 + lane] = (unsigned short)(cc >> 16);
    }
    if (lane == 0) {
      reinterpret_cast<int*>(hdr)[0] = (over || inf.x < 0) ? -1 : i;
      reinterpret_cast<int*>(hdr)[1] = (nA + nR) | (nA << 16) | (__float_as_int(xi.w) << 24);
      if (over) atomicOr(a.err_flag, 1);
    }
    if (!over && inf.x >= 0) {
      // ---- LDS lists straight from the screened candidates.  The candidates come sorted by species, so a candidate's place
      // in the species-grouped radial list is its stream index plus ONE number of its species: an angular entry of species s
      // sits behind the radial-only entries of the species before it (exclusive prefix of the radial-only counts), a
      // radial-only entry behind the angular entries up to and including its own species (inclusive prefix of the angular
      // counts).  The prefixes are a three-level DPP scan over the lanes that hold the counts -- no scalar tables (unpack_lists
      // walks 27 of them, which here spilled) -- and a candidate fetches its species' pair with one ds_bpermute. ----
      int inc = cc;
      inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, true);   // row_shr:1
      inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, true);   // row_shr:2
      inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, true);   // row_shr:4
      const int exc = inc - cc;   // lane s: counts of the species before s; lane S: the totals
      if (lane <= p.S) { L.rstart[lane] = (exc & 0xffff) + (exc >> 16); L.astart[lane] = exc & 0xffff; }
      const int tab = (exc >> 16) | (inc << 16);   // {radial-only entries before species s, angular entries up to and including s}
#pragma unroll
      for (int c = 0; c < NCHC; c++) {
        const int t = __shfl(tab, spc[c]);
        if (pos[c] >= 0) {
          const bool ang = pos[c] < kMaxAng;
          const int w = ang ? pos[c] : pos[c] - kMaxAng;
          const int pr = w + (ang ? (t & 0xffff) : ((unsigned)t >> 16));
          const float4 v = dd[c];
          L.rr[pr] = v.w;
          L.rfc[pr] = 0.5f * fcos_rev(v.w * half_inv_Rcr) + 0.5f;
          if (ang) {
            L.ad[w] = v;
            L.afc[w] = 0.5f * fcos_rev(v.w * half_inv_Rca) + 0.5f;
          }
        }
      }
      wave_sync();
#ifdef ABLB_STAMPS
      unsigned long long stamp_prev = 0, stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      forward_compute<NA, NZ>(p, a, L, row, lane, stamp_prev, stamp_acc);
#else
      forward_compute<NA, NZ>(p, a, L, row, lane);
#endif
    }
    info = info1;
#pragma unroll
    for (int c = 0; c < NCHC; c++) jj[c] = jj1[c];
    ANI_TICKET_NEXT(tn, k2)
  }
  ANI_TICKETS_END(kWaves)
}

// coalesced force scatter of `cnt` neighbours whose gradients sit in LDS as g[3*q + k] with atom indices jx[q]:
// lane = (list slot, component), so the three adds of one atom (one float4 of fbuf) leave from adjacent lanes of ONE
// instruction.  Global float atomics execute at the memory side as 64-byte requests (MI355X_MICROARCH.md): three
// separate instructions per neighbour were three requests, this is one, and atoms adjacent both in memory and in the
// (spatially ordered) list share requests too.
template <typename T>
__device__ __forceinline__ void scatter_neighbours(const AevArgs& a, const T* g, const int* jx, int cnt, int lane) {
#ifndef ABL_NO_GATOM
  for (int base = 0; base < cnt; base += 16) {
    const int q = base + (lane >> 2), k = lane & 3;
    if (q < cnt && k < 3) atomicAdd(&a.fbuf[4 * jx[q] + k], -(float)g[3 * q + k]);
  }
#endif
}

template <int NA, int NZ, int NCH, int GR, bool VIR, typename PF>
__device__ __forceinline__ void backward_centre(const AevParams& p, const AevArgs& a, FastLds& L, int row, const hdr_t h,
                                                const Loaded<NCH, true, GR> pf, int lane, float (&wv)[9],
                                                const RepTab& rep, float& er, PF&& before_final_scatter BWD_STAMP_PARAMS) {
  constexpr int NR = 16;
  const int centre = h[0];
  const int nrad = hdr_nrad(h), nang = hdr_nang(h);
  // the rows of the neighbours that are centres themselves (symmetric radial collection, below): asked for here, in front of
  // the centre's set-up, instead of inside the radial stage -- where the lookup stood between the entry and the row it selects,
  // two dependent round trips in a row
  int rjp[NCH];
#pragma unroll
  for (int k = 0; k < NCH; k++) {
    rjp[k] = -1;
#ifndef ANI_RAD_NOREC
    if (a.row_of_atom && 64 * k + lane < nrad) rjp[k] = a.row_of_atom[cl_index(pf.jj[k])];
#endif
  }
#pragma unroll
  for (int c = 0; c < GR; c++)
    if (lane + 64 * c < (p.aev_stride >> 2)) reinterpret_cast<float4*>(L.row)[lane + 64 * c] = pf.grow[c];
  {
    // starts of the species groups in the angular list (the pair table is built from them): lane s takes its count out of the
    // header and a three-level DPP scan makes the prefix; lane S ends up with the total
    const int word = lane < 4 ? h[2] : h[3];
    const int ca = lane < 8 ? (word >> (8 * (lane & 3))) & 0xff : 0;
    int inc = ca;
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, true);   // row_shr:1
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, true);   // row_shr:2
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, true);   // row_shr:4
    if (lane <= p.S) L.astart[lane] = inc - ca;
  }
  wave_sync();   // the dE/dAEV row is read below

  // ---- radial stage, one lane per neighbour, straight from the prefetched registers.  A radial-only neighbour
  // (Rca < r <= Rcr) is finished here: its gradient goes to the centre's sums and, through a 64-slot staging buffer,
  // to the force scatter.  A neighbour inside Rca parks its displacement, atom index and radial gradient in LDS, where
  // the angular stage adds to it. ----
  float fx = 0.f, fy = 0.f, fz = 0.f;   // this lane's share of sum_j g_j (force on the centre)
  const float cR = -p.EtaR * kLog2e;
  const float rev = 0.5f * p.pi_over_Rcr * 0.3183098861837907f;
  const float revA = 0.5f * p.pi_over_Rca * 0.3183098861837907f;
  const float4* __restrict__ px = a.cl_xyz + (size_t)row * a.cl_stride;
  const int* __restrict__ pj = a.cl_j + (size_t)row * a.cl_stride;
  // chunks of the compact list in slot order: lanes with t < nang hold angular neighbours, the others radial-only ones
  for (int base = 0; base < nrad; base += 64) {
    const int c = base >> 6, t0 = base + lane;
    const bool live = t0 < nrad, ang = t0 < nang;
    const int w = ang ? t0 : t0 - nang;   // index in its stream
    float4 v = make_float4(0.f, 0.f, 0.f, 1.f);
    int j = 0;
    if (c < NCH) {
#pragma unroll
      for (int k = 0; k < NCH; k++)
        if (c == k) { v = pf.xx[k]; j = pf.jj[k]; }
    } else if (live) {  // lists longer than the chunks held in registers: loaded in place
      v = px[slot_index(t0, nang)];
      j = pj[slot_index(t0, nang)];
    }
    const int t = w;   // angular neighbours: index of the LDS accumulators
    float gx = 0.f, gy = 0.f, gz = 0.f;
#ifndef ABL_NO_RAD
    const int s = cl_species(j);   // the entry carries its species (cl_pack)
    j = cl_index(j);
    // A pair's two radial terms -- E_centre's and E_neighbour's dependence on their distance -- differ only in the dE/dAEV
    // row they are weighted with (same Gaussians, same cutoff factor).  When the neighbour is a centre of this launch too, the
    // centre reads the neighbour's row (its block for the centre's species) and takes BOTH terms' force on itself; the
    // neighbour does the same from its side, and neither scatters a radial gradient to the other: two thirds of the kernel's
    // global atomics were those (it waits for its atomics, not for its arithmetic).  Ghosts have no row here: their term is
    // scattered as before and comes back through the ghost exchange.  sym: this lane's neighbour is handled that way.
    float gsx = 0.f, gsy = 0.f, gsz = 0.f;   // sym: what the centre adds to its own force beyond (gx, gy, gz)
    int rj = -1;
#ifndef ANI_RAD_NOREC
    if (c < NCH) {
#pragma unroll
      for (int k = 0; k < NCH; k++)
        if (c == k) rj = rjp[k];
    } else if (a.row_of_atom && live) rj = a.row_of_atom[j];
#endif
    const bool sym = rj >= 0;
    if (live) {
      const float4* gg4 = reinterpret_cast<const float4*>(L.row + s * NR);
      const float4* qq4 = reinterpret_cast<const float4*>(a.gaev + (size_t)(sym ? rj : row) * p.aev_stride + hdr_species(h) * NR);
      const float r = v.w;
      const float fc = 0.5f * fcos_rev(r * rev) + 0.5f;
      const float dfc = -0.5f * p.pi_over_Rcr * fsin_rev(r * rev);
      float dEdr = 0.f, dEdn = 0.f;   // d/dr of the centre's energy, of the neighbour's
#ifndef ANI_RAD_NOREC
      // The sixteen Gaussians exp2(cR (r - ShfR_k)^2) from six exponentials: in each half of the shifts, around its midpoint
      // sc, d_k = dc - m Delta (m = k - 3.5) and exp2(cR d_k^2) = E H^(2m) C_|m| with E = exp2(cR dc^2), H = exp2(-cR dc Delta),
      // C_m = exp2(cR m^2 Delta^2) (wave-uniform).  A transcendental takes the issue time of ~5 FMAs (tools/issue_probe.hip).
      // E H^(2m) <= 1 / C_m; when E underflows, the half's largest term is below exp(-EtaR (|dc| - 3.5 Delta)^2) ~ 1e-12.
      {
        const float cd = cR * p.dShfR * p.dShfR;
        const float C0 = fexp2(0.25f * cd), C1 = fexp2(2.25f * cd), C2 = fexp2(6.25f * cd), C3 = fexp2(12.25f * cd);
        const float kf = -2.f * p.EtaR * fc;
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
          const float dc = r - (p.ShfR0 + (3.5f + 8.f * hf) * p.dShfR);
          const float x = cR * dc;
          const float E = fexp2(x * dc), H = fexp2(-x * p.dShfR), Hi = fexp2(x * p.dShfR);
          const float R = H * H, Ri = Hi * Hi;
          float e[8];
          float up = E * H, dn = E * Hi;
          e[4] = up * C0; up *= R; e[5] = up * C1; up *= R; e[6] = up * C2; up *= R; e[7] = up * C3;
          e[3] = dn * C0; dn *= Ri; e[2] = dn * C1; dn *= Ri; e[1] = dn * C2; dn *= Ri; e[0] = dn * C3;
          const float4 ga = gg4[2 * hf], gb = gg4[2 * hf + 1];
          const float gk[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
          float4 qa = make_float4(0.f, 0.f, 0.f, 0.f), qb = qa;
          if (sym) { qa = qq4[2 * hf]; qb = qq4[2 * hf + 1]; }
          const float qk[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
          float dr = dc + 3.5f * p.dShfR;   // r - ShfR[8 hf]
#pragma unroll
          for (int kk = 0; kk < 8; kk++) {
            const float tk = e[kk] * fmaf(kf, dr, dfc);
            dEdr = fmaf(gk[kk], tk, dEdr);
            dEdn = fmaf(qk[kk], tk, dEdn);
            dr -= p.dShfR;
          }
        }
      }
#else
      float dr = r - p.ShfR0;   // equidistant shifts
#pragma unroll
      for (int k4 = 0; k4 < NR / 4; k4++) {
        const float4 gv = gg4[k4];
        const float gk[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          const float e = fexp2(cR * dr * dr);
          dEdr = fmaf(gk[kk] * e, fmaf(-2.f * p.EtaR * dr, fc, dfc), dEdr);
          dr -= p.dShfR;
        }
      }
#endif
      const float inv_r = frcp(r);
      const float sc = 0.25f * dEdr * inv_r, scn = 0.25f * dEdn * inv_r;
      gx = sc * v.x; gy = sc * v.y; gz = sc * v.z;
      gsx = scn * v.x; gsy = scn * v.y; gsz = scn * v.z;
      if (rep.on && r < rep.cutoff) {
        // pairwise repulsion (ani_kernels_rep.hip has the formulas): half of e(r) per list entry.  The pair function in
        // fp32 like the rest of this precision mode, but its argument from the fp64 positions: the wall is steep
        const int ti = 8 * hdr_species(h) + s;
        const double* xj = rep.x64 + 3 * (long long)j;
        const double* xc = rep.x64 + 3 * (long long)centre;
        const double ddx = xj[0] - xc[0], ddy = xj[1] - xc[1], ddz = xj[2] - xc[2];
        const double r64 = sqrt(ddx * ddx + ddy * ddy + ddz * ddz);
        const float rr = (float)r64;
        const float x = rr * frcp(rep.cutoff), den = 1.f - x * x;
        if (den > 1e-10f) {
          const float fcr = fexp2((1.f - frcp(den)) * kLog2e);
          const float dfcr = fcr * (-(2.f * x * frcp(rep.cutoff)) * frcp(den * den));
          const float db = rr * 1.8897261258369282f, al = rep.sa[ti], kk = rep.k[ti];
          const float pk1 = fexp2((kk - 1.f) * flog2(db));
          const float g = rep.y[ti] * frcp(db) * fexp2(-al * pk1 * db * kLog2e);
          const float dg = 1.8897261258369282f * g * (-frcp(db) - al * kk * pk1);
          er += 0.5f * g * fcr;
          const float sr = 0.5f * (dg * fcr + g * dfcr);
          const double inv = 1.0 / r64;
          const float rx = sr * (float)(ddx * inv), ry = sr * (float)(ddy * inv), rz = sr * (float)(ddz * inv);
          gx += rx; gy += ry; gz += rz;
          if (sym) { gsx += rx; gsy += ry; gsz += rz; }   // the neighbour's entry for this pair holds the same half of the pair term
        }
      }
    }
#endif
    if (live && ang) {
      L.ad[t] = v;
      L.afc[t] = 0.5f * fcos_rev(v.w * revA) + 0.5f;
      L.aj[t] = j;
      if (sym) {   // radial terms settled here on the centre's side; the accumulator starts empty for the angular stage
        fx += gx + gsx; fy += gy + gsy; fz += gz + gsz;
        if constexpr (VIR) {
          wv[0] += gx * v.x; wv[1] += gx * v.y; wv[2] += gx * v.z;
          wv[3] += gy * v.x; wv[4] += gy * v.y; wv[5] += gy * v.z;
          wv[6] += gz * v.x; wv[7] += gz * v.y; wv[8] += gz * v.z;
        }
        gx = gy = gz = 0.f;
      }
      L.gd[3 * t] = gx; L.gd[3 * t + 1] = gy; L.gd[3 * t + 2] = gz;
    }
    // radial-only neighbours: finished here (scattered at once; ONE merged scatter per centre -- angular + radial-only +
    // centre, 21 atoms per atomic instruction instead of 16 -- was measured: 5 % slower, the atomics then leave in one
    // burst at the end of the centre instead of two spread over it)
    const int lo = min(max(nang - base, 0), 64), hi = min(nrad - base, 64);   // radial-only lanes of this chunk: [lo, hi)
    if (hi > lo) {   // wave-uniform
      const bool out = live && !ang && !sym;            // a radial-only neighbour whose gradient has to travel
      const unsigned long long mout = __ballot(out);
      if (live && !ang) {
        fx += gx + gsx; fy += gy + gsy; fz += gz + gsz;   // (gs = 0 unless sym)
        if constexpr (VIR) {
          wv[0] += gx * v.x; wv[1] += gx * v.y; wv[2] += gx * v.z;
          wv[3] += gy * v.x; wv[4] += gy * v.y; wv[5] += gy * v.z;
          wv[6] += gz * v.x; wv[7] += gz * v.y; wv[8] += gz * v.z;
        }
        if (out) {
          const int q = lanes_below(mout);
          L.gt[3 * q] = gx; L.gt[3 * q + 1] = gy; L.gt[3 * q + 2] = gz;
          L.jt[q] = j;
        }
      }
      wave_sync();
      scatter_neighbours(a, L.gt, L.jt, __popcll(mout), lane);
      wave_sync();   // the staging buffer is reused by the next chunk
    }
  }
  BWD_STAMP(2);   // radial stage incl. the radial-only scatter
#ifndef ANI_BWD_ROWS
  // ---- angular, dense pair stream (round 3).  The pairs of all non-empty species-pair buckets form ONE stream, bucket after
  // bucket, that a step takes 64 lanes of whatever buckets they belong to: a water centre's 153 pairs are 3 steps (the row
  // layout kept below under ANI_BWD_ROWS, one 16-lane row per row neighbour, needed 4: 60 % of its lanes held a pair).  The
  // two half-waves walk the stream's column slots together, on two adjacent row neighbours (build_pair_table).  The pairs of
  // a half-wave that share their ROW neighbour are consecutive lanes (a run): its gradient is summed over the run by a
  // segmented scan inside each 16-lane DPP row -- level d adds the value d lanes down if that lane belongs to the same run --
  // and the last lane of the run in its DPP row adds the partial sum to the neighbour's LDS accumulator.  The COLUMN
  // neighbour's gradient: the two half-waves' values summed with a half-wave swap, one LDS add per lane of the lower half.
  int nbk;
  const int total_pairs = build_pair_table<NA, NZ>(p, lane, L, nbk);
  wave_sync();
  BWD_STAMP(3);   // pair table
  const float cA = -p.EtaA * kLog2e;
#ifdef ABL_NO_ANG
  for (int base = 0; base < 0; base += 64) {
#else
  for (int base = 0; base < total_pairs; base += 32) {
#endif
    int ia, ib, outoff, pos;
    bool valid;
    stream_pair_half(L, nbk, base + (lane & 31), lane >> 5, ia, ib, outoff, pos, valid);
    const float4 A = L.ad[ia], B = L.ad[ib];
    const float inv_ra = frcp(A.w), inv_rb = frcp(B.w);
    const float inv_rr = inv_ra * inv_rb;
    const float cosv = (A.x * B.x + A.y * B.y + A.z * B.z) * inv_rr;
    const float c = 0.95f * cosv;
    const float s2 = fmaxf(1.f - c * c, 1e-12f);
    const float inv_s = frsq(s2);
    const float sn = s2 * inv_s;
    const float fa = L.afc[ia], fb = L.afc[ib];
    const float dfa = -0.5f * p.pi_over_Rca * fsin_rev(A.w * revA);
    const float dfb = -0.5f * p.pi_over_Rca * fsin_rev(B.w * revA);
    const float rho = 0.5f * (A.w + B.w);
    float f1[NZ], df1[NZ];
#pragma unroll
    for (int z = 0; z < NZ; z++) {
      const float bz = fmaxf(0.5f * (1.f + c * p.cosZ[z] + sn * p.sinZ[z]), 0.f);
      const float pm1 = fexp2((p.Zeta - 1.f) * flog2(bz));
      f1[z] = pm1 * bz;
      df1[z] = p.Zeta * pm1 * 0.5f * (sn * p.cosZ[z] - c * p.sinZ[z]) * inv_s;
    }
    const float4* gg4 = reinterpret_cast<const float4*>(L.row + outoff);  // outoff is a multiple of NA*NZ
    float Aq = 0.f, Bq = 0.f, Cq = 0.f;
    float drA = rho - p.ShfA0;
#pragma unroll
    for (int sa = 0; sa < NA; sa++) {
#ifdef ANI_GG_SPLIT
      // keep the compiler from loading all NA x NZ dE/dAEV values of the bucket before the first is used (32 registers):
      // the second half of the shifts is fetched after the first half has been contracted
      if (sa == NA / 2) asm volatile("" ::: "memory");
#endif
      const float dr = drA;
      drA -= p.dShfA;   // equidistant shifts
      const float f2 = fexp2(cA * dr * dr);
      const float df2 = -2.f * p.EtaA * dr * f2;
      float g1 = 0.f, gd1 = 0.f;
#pragma unroll
      for (int z4 = 0; z4 < NZ / 4; z4++) {
        const float4 gv = gg4[sa * (NZ / 4) + z4];
        g1 = fmaf(gv.x, f1[4 * z4], g1); gd1 = fmaf(gv.x, df1[4 * z4], gd1);
        g1 = fmaf(gv.y, f1[4 * z4 + 1], g1); gd1 = fmaf(gv.y, df1[4 * z4 + 1], gd1);
        g1 = fmaf(gv.z, f1[4 * z4 + 2], g1); gd1 = fmaf(gv.z, df1[4 * z4 + 2], gd1);
        g1 = fmaf(gv.w, f1[4 * z4 + 3], g1); gd1 = fmaf(gv.w, df1[4 * z4 + 3], gd1);
      }
      Aq = fmaf(f2, gd1, Aq);
      Cq = fmaf(f2, g1, Cq);
      Bq = fmaf(df2, g1, Bq);
    }
    // gradient of this pair w.r.t. the two neighbour displacements; zero for masked lanes
    const float P = valid ? fa * fb : 0.f;
    Aq *= 2.f * P * 0.95f;
    Bq *= P;       // 2 * P * 0.5
    Cq *= valid ? 2.f : 0.f;
    const float cc = Aq * inv_rr;
    const float ta = (Bq + Cq * dfa * fb) * inv_ra - Aq * cosv * inv_ra * inv_ra;
    const float tb = (Bq + Cq * fa * dfb) * inv_rb - Aq * cosv * inv_rb * inv_rb;
    const float va[3] = {cc * B.x + ta * A.x, cc * B.y + ta * A.y, cc * B.z + ta * A.z};  // d/d(neighbour ia)
    const float vb[3] = {cc * A.x + tb * B.x, cc * A.y + tb * B.y, cc * A.z + tb * B.z};  // d/d(neighbour ib)
    // column neighbour: lanes l and l + 32 hold the same one
    {
#ifndef ANI_COL_SWAP
      // one fp64 LDS add per lane and component (the two half-waves' lanes l, l + 32 share the address: a two-way conflict costs
      // less than the half-wave swap-add that used to halve the number of slow fp32 adds)
      if (valid) { TILE_ADD(&L.gd[3 * ib], vb[0]); TILE_ADD(&L.gd[3 * ib + 1], vb[1]); TILE_ADD(&L.gd[3 * ib + 2], vb[2]); }
#else
      const float c0 = xor_sum<32>(vb[0]), c1 = xor_sum<32>(vb[1]), c2 = xor_sum<32>(vb[2]);
#ifdef ABL_NO_COLADD2
      asm volatile("" ::"v"(c0), "v"(c1), "v"(c2));
#else
      if (valid && lane < 32) { TILE_ADD(&L.gd[3 * ib], c0); TILE_ADD(&L.gd[3 * ib + 1], c1); TILE_ADD(&L.gd[3 * ib + 2], c2); }
#endif
#endif
    }
    // row neighbour: segmented inclusive scan over the run, inside the lane's 16-lane DPP row
    float rs[3] = {va[0], va[1], va[2]};
#pragma unroll
    for (int k = 0; k < 3; k++) {
      // the DPP move is executed by every lane (it is a cross-lane read: inside a conditional it would see the lanes the
      // condition switched off as invalid sources); the condition only selects what is added
      float t;
      t = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(rs[k]), 0x111, 0xf, 0xf, true)); rs[k] += pos >= 1 ? t : 0.f;   // row_shr:1
      t = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(rs[k]), 0x112, 0xf, 0xf, true)); rs[k] += pos >= 2 ? t : 0.f;   // row_shr:2
      t = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(rs[k]), 0x114, 0xf, 0xf, true)); rs[k] += pos >= 4 ? t : 0.f;   // row_shr:4
      t = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(rs[k]), 0x118, 0xf, 0xf, true)); rs[k] += pos >= 8 ? t : 0.f;   // row_shr:8
    }
    // the run ends here (for this DPP row) if the next lane starts a run or lies in the next DPP row
    const int pos_next = __builtin_amdgcn_update_dpp(0, pos, 0x101, 0xf, 0xf, true);   // row_shl:1: lane 15 of a row reads 0
    if (valid && pos_next == 0) { TILE_ADD(&L.gd[3 * ia], rs[0]); TILE_ADD(&L.gd[3 * ia + 1], rs[1]); TILE_ADD(&L.gd[3 * ia + 2], rs[2]); }
  }
  wave_sync();   // gd is complete
#else
  int nbk;
  const int nrows_stream = build_row_table<NA, NZ>(p, lane, L, nbk);
  wave_sync();
  BWD_STAMP(3);   // row table

  // ---- angular: lane = pair; 4 rows of the stream per step (see build_row_table) ----
  const float cA = -p.EtaA * kLog2e;
#ifdef ABL_NO_ANG
  for (int RB = 0; RB < 0; RB += kRowBlock) {
#else
  for (int RB = 0; RB < nrows_stream; RB += kRowBlock) {
#endif
   build_row_descriptors(L, nbk, nrows_stream, RB, lane);
   wave_sync();
   for (int R0 = RB; R0 < min(RB + kRowBlock, nrows_stream); R0 += 64 / kRowW) {
    const int rl = R0 - RB + lane / kRowW;
    const int4 d0 = L.rowd[2 * rl], d1 = L.rowd[2 * rl + 1];
    const bool row_ok = (d1.y >> 17) & 1, tri = (d1.y >> 16) & 1;
    const int r = d1.y & 0xffff, nn = d1.z;
    const int col = d1.x + lane % kRowW;
    const bool above = !tri || col > r;   // tri: which of the two folded pairs this lane holds
    const int b = above ? col : nn - 1 - col;
    const bool valid = row_ok && b < d0.w && (!tri || col != r) && (above || d0.y >= 0);
    // masked lanes: two distinct neighbours keep the geometry finite
    const int ia = (valid && !above) ? d0.y : d0.x;
    const int ib = d0.z + (valid ? b : ((tri && r == 0) ? 1 : 0));
    const float4 A = L.ad[ia], B = L.ad[ib];
    const float inv_ra = frcp(A.w), inv_rb = frcp(B.w);
    const float inv_rr = inv_ra * inv_rb;
    const float cosv = (A.x * B.x + A.y * B.y + A.z * B.z) * inv_rr;
    const float c = 0.95f * cosv;
    const float s2 = fmaxf(1.f - c * c, 1e-12f);
    const float inv_s = frsq(s2);
    const float sn = s2 * inv_s;
    const float fa = L.afc[ia], fb = L.afc[ib];
    const float dfa = -0.5f * p.pi_over_Rca * fsin_rev(A.w * revA);
    const float dfb = -0.5f * p.pi_over_Rca * fsin_rev(B.w * revA);
    const float rho = 0.5f * (A.w + B.w);
    float f1[NZ], df1[NZ];
#pragma unroll
    for (int z = 0; z < NZ; z++) {
      const float bz = fmaxf(0.5f * (1.f + c * p.cosZ[z] + sn * p.sinZ[z]), 0.f);
      const float pm1 = fexp2((p.Zeta - 1.f) * flog2(bz));
      f1[z] = pm1 * bz;
      df1[z] = p.Zeta * pm1 * 0.5f * (sn * p.cosZ[z] - c * p.sinZ[z]) * inv_s;
    }
    const float4* gg4 = reinterpret_cast<const float4*>(L.row + d1.w);  // outoff is a multiple of NA*NZ
    float Aq = 0.f, Bq = 0.f, Cq = 0.f;
    float drA = rho - p.ShfA0;
#pragma unroll
    for (int sa = 0; sa < NA; sa++) {
#ifdef ANI_GG_SPLIT
      // keep the compiler from loading all NA x NZ dE/dAEV values of the bucket before the first is used (32 registers):
      // the second half of the shifts is fetched after the first half has been contracted
      if (sa == NA / 2) asm volatile("" ::: "memory");
#endif
      const float dr = drA;
      drA -= p.dShfA;   // equidistant shifts
      const float f2 = fexp2(cA * dr * dr);
      const float df2 = -2.f * p.EtaA * dr * f2;
      float g1 = 0.f, gd1 = 0.f;
#pragma unroll
      for (int z4 = 0; z4 < NZ / 4; z4++) {
        const float4 gv = gg4[sa * (NZ / 4) + z4];
        g1 = fmaf(gv.x, f1[4 * z4], g1); gd1 = fmaf(gv.x, df1[4 * z4], gd1);
        g1 = fmaf(gv.y, f1[4 * z4 + 1], g1); gd1 = fmaf(gv.y, df1[4 * z4 + 1], gd1);
        g1 = fmaf(gv.z, f1[4 * z4 + 2], g1); gd1 = fmaf(gv.z, df1[4 * z4 + 2], gd1);
        g1 = fmaf(gv.w, f1[4 * z4 + 3], g1); gd1 = fmaf(gv.w, df1[4 * z4 + 3], gd1);
      }
      Aq = fmaf(f2, gd1, Aq);
      Cq = fmaf(f2, g1, Cq);
      Bq = fmaf(df2, g1, Bq);
    }
    // gradient of this pair w.r.t. the two neighbour displacements; zero for masked lanes
    const float P = valid ? fa * fb : 0.f;
    Aq *= 2.f * P * 0.95f;
    Bq *= P;       // 2 * P * 0.5
    Cq *= valid ? 2.f : 0.f;
    const float cc = Aq * inv_rr;
    const float ta = (Bq + Cq * dfa * fb) * inv_ra - Aq * cosv * inv_ra * inv_ra;
    const float tb = (Bq + Cq * fa * dfb) * inv_rb - Aq * cosv * inv_rb * inv_rb;
    const float va[3] = {cc * B.x + ta * A.x, cc * B.y + ta * A.y, cc * B.z + ta * A.z};  // d/d(neighbour ia)
    const float vb[3] = {cc * A.x + tb * B.x, cc * A.y + tb * B.y, cc * A.z + tb * B.z};  // d/d(neighbour ib)
#ifndef ABL_NO_LATOM
    // column neighbour: one LDS add per lane and component
#ifndef ABL_NO_COLADD
#if ANI_ROW_W == 16 && defined(ANI_COLREDUCE)   // measured slower (0.434 against 0.407 ms with registers to spare, worse when it spills)
    // A ds_add_f32 whose lanes hit the same address is serialised per address, and the four rows of a step that belong to
    // the same bucket and column block share their 16 columns: a 4-way conflict on every column add, ~2/3 of the LDS
    // time of this pass.  Such a step (wave-uniform test on the rows' keys) sums the four rows in registers first -- two
    // row / half-wave swaps per value -- and lets ONE row of lanes add, conflict-free.  cA: pairs whose column neighbour
    // is `col` (all pairs of a rectangular bucket, the above-diagonal ones of a folded triangle); cB: the
    // below-diagonal ones, whose column neighbour is nn-1-col.
    const int key = row_ok ? ((d1.w << 8) | (d1.x >> 4)) : -1;   // bucket (its output offset) and column block
    const int k0 = __builtin_amdgcn_readlane(key, 0), k1 = __builtin_amdgcn_readlane(key, 16),
              k2 = __builtin_amdgcn_readlane(key, 32), k3 = __builtin_amdgcn_readlane(key, 48);
    if ((k1 == k0 || k1 < 0) && (k2 == k0 || k2 < 0) && (k3 == k0 || k3 < 0)) {   // wave-uniform
      const bool tri0 = __builtin_amdgcn_readlane(d1.y, 0) & (1 << 16);
      float cA[3];
#pragma unroll
      for (int k = 0; k < 3; k++) cA[k] = xor_sum<32>(xor_sum<16>((valid && above) ? vb[k] : 0.f));
      if (lane < 16 && col < d0.w) {
        const int q = d0.z + col;
        TILE_ADD(&L.gd[3 * q], cA[0]); TILE_ADD(&L.gd[3 * q + 1], cA[1]); TILE_ADD(&L.gd[3 * q + 2], cA[2]);
      }
      if (tri0) {
        float cB[3];
#pragma unroll
        for (int k = 0; k < 3; k++) cB[k] = xor_sum<32>(xor_sum<16>((valid && !above) ? vb[k] : 0.f));
        const int colB = nn - 1 - col;
        if (lane < 16 && colB >= 0 && colB < d0.w) {
          const int q = d0.z + colB;
          TILE_ADD(&L.gd[3 * q], cB[0]); TILE_ADD(&L.gd[3 * q + 1], cB[1]); TILE_ADD(&L.gd[3 * q + 2], cB[2]);
        }
      }
    } else
#endif
    if (valid) {   // angular list index = index of the LDS accumulators
      TILE_ADD(&L.gd[3 * ib], vb[0]); TILE_ADD(&L.gd[3 * ib + 1], vb[1]); TILE_ADD(&L.gd[3 * ib + 2], vb[2]);
    }
#else
    asm volatile("" ::"v"(vb[0]), "v"(vb[1]), "v"(vb[2]));
#endif
    // row neighbour(s): sums over the 16 lanes of the row, [0..2] for the pairs whose ia is r (all pairs of a
    // rectangular bucket, the above-diagonal ones of a folded triangle), [3..5] for those whose ia is nn-1-r
    float rw[6];
#pragma unroll
    for (int k = 0; k < 3; k++) { rw[k] = above ? va[k] : 0.f; rw[3 + k] = above ? 0.f : va[k]; }
#ifndef ABL_NO_ROWDPP
#pragma unroll
    for (int k = 0; k < 6; k++) {
      rw[k] = xor_sum<2>(xor_sum<1>(rw[k]));
      if constexpr (kRowW >= 8) rw[k] = xor_sum<4>(rw[k]);
      if constexpr (kRowW >= 16) rw[k] = xor_sum<8>(rw[k]);
    }
#endif
#ifdef ANI_PARK_ROWS
    // experiment (slower, kept for the record): row sums parked per row and added once per block of rows
    if (lane % kRowW == 0) {
      L.rowacc[2 * rl] = make_float4(rw[0], rw[1], rw[2], 0.f);
      L.rowacc[2 * rl + 1] = make_float4(rw[3], rw[4], rw[5], 0.f);
    }
#elif !defined(ABL_NO_ROWADD)
    if (lane % kRowW == 0 && row_ok) {
      {
        const int q = d0.x;
        TILE_ADD(&L.gd[3 * q], rw[0]); TILE_ADD(&L.gd[3 * q + 1], rw[1]); TILE_ADD(&L.gd[3 * q + 2], rw[2]);
      }
      if (d0.y >= 0) {
        const int q = d0.y;
        TILE_ADD(&L.gd[3 * q], rw[3]); TILE_ADD(&L.gd[3 * q + 1], rw[4]); TILE_ADD(&L.gd[3 * q + 2], rw[5]);
      }
    }
#else
    asm volatile("" ::"v"(rw[0]), "v"(rw[1]), "v"(rw[2]), "v"(rw[3]), "v"(rw[4]), "v"(rw[5]));
#endif
#else
    asm volatile("" ::"v"(va[0]), "v"(va[1]), "v"(va[2]), "v"(vb[0]), "v"(vb[1]), "v"(vb[2]));
#endif
   }
   wave_sync();
#if defined(ANI_PARK_ROWS) && !defined(ABL_NO_LATOM)
   if (lane < kRowBlock && RB + lane < nrows_stream) {   // row sums -> accumulators of the rows' neighbours
     const int4 d0 = L.rowd[2 * lane];
     const float4 sa = L.rowacc[2 * lane];
     TILE_ADD(&L.gd[3 * d0.x], sa.x); TILE_ADD(&L.gd[3 * d0.x + 1], sa.y); TILE_ADD(&L.gd[3 * d0.x + 2], sa.z);
     if (d0.y >= 0) {
       const float4 sb = L.rowacc[2 * lane + 1];
       TILE_ADD(&L.gd[3 * d0.y], sb.x); TILE_ADD(&L.gd[3 * d0.y + 1], sb.y); TILE_ADD(&L.gd[3 * d0.y + 2], sb.z);
     }
   }
#endif
   wave_sync();   // the tables are rewritten by the next block of rows; after the last block: gd is complete
  }

#endif
  BWD_STAMP(4);   // angular stage
  // ---- the angular neighbours: F_j -= gd_j ; F_i += sum_j gd_j ; virial -= gd (x) d ----
  for (int q = lane; q < nang; q += 64) {
    const float gx = (float)L.gd[3 * q], gy = (float)L.gd[3 * q + 1], gz = (float)L.gd[3 * q + 2];
    fx += gx; fy += gy; fz += gz;
    if constexpr (VIR) {
      // virial: per-lane partial sums kept across all the centres of this wave; the nine totals are reduced over the
      // lanes and added to the global accumulator ONCE per wave at the end of the kernel (nine double atomics per centre
      // on nine addresses serialise at the memory side: 10 ms per step at 100 000 centres)
      const float4 d = L.ad[q];
      wv[0] += gx * d.x; wv[1] += gx * d.y; wv[2] += gx * d.z;
      wv[3] += gy * d.x; wv[4] += gy * d.y; wv[5] += gy * d.z;
      wv[6] += gz * d.x; wv[7] += gz * d.y; wv[8] += gz * d.z;
    }
  }
  before_final_scatter();   // the caller's loads for the wave's next centre: in front of the atomics in the in-order memory queue
  scatter_neighbours(a, L.gd, L.aj, nang, lane);
  fx = xor_sum<32>(xor_sum<16>(xor_sum<8>(xor_sum<4>(xor_sum<2>(xor_sum<1>(fx))))));
  fy = xor_sum<32>(xor_sum<16>(xor_sum<8>(xor_sum<4>(xor_sum<2>(xor_sum<1>(fy))))));
  fz = xor_sum<32>(xor_sum<16>(xor_sum<8>(xor_sum<4>(xor_sum<2>(xor_sum<1>(fz))))));
  if (lane < 3) atomicAdd(&a.fbuf[4 * centre + lane], lane == 0 ? fx : (lane == 1 ? fy : fz));
  wave_sync();  // the LDS slice is reused by this wave's next centre
  BWD_STAMP(5);   // final scatter
}

template <int NA, int NZ, int NCH, int GR, bool VIR>
__global__ __launch_bounds__(64 * kWavesB, ANI_BWD_MINW) void aev_backward_fast(AevParams p, AevArgs a, int cap, int rowf, RepTab rep) {
  extern __shared__ float4 smem4[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  FastLds L = carve<NA, NZ>(reinterpret_cast<float*>(smem4) + wave * fast_wave_floats_row(cap, true, rowf, p.S), cap, true, rowf);
  float wv[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // this lane's share of the wave's virial
  float er = 0.f;                                                // ... and of its repulsion energy
#ifdef ABLB_STAMPS
  unsigned long long stamp_prev, stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
  const int nw = gridDim.x * kWavesB;
  for (int k = blockIdx.x * kWavesB + wave; k < a.kcount; k += nw) {
    const int row = a.row_list ? __builtin_amdgcn_readfirstlane(a.row_list[a.k0 + k]) : a.k0 + k;
    const hdr_t hc = load_header(a, row);
    BWD_STAMP(0);   // header (waits for whatever this wave still has in flight: the last centre's atomics)
    if (hc[0] < 0) continue;
    Loaded<NCH, true, GR> cur;
    load_lists(p, a, row, hdr_nrad(hc), hdr_nang(hc), lane, cur);
#ifdef ABLB_STAMPS_VM
    BWD_STAMP_VM(1);   // list + dE/dAEV row
#else
    BWD_STAMP(1);
#endif
    backward_centre<NA, NZ, NCH, GR, VIR>(p, a, L, row, hc, cur, lane, wv, rep, er, [] {}, stamp_prev, stamp_acc);
    stamp_acc[6] += 1;
  }
  if (wave == 0 && lane == 0) {
    for (int q = 0; q < 6; q++) atomicAdd(&g_bwd_stamps[q], stamp_acc[q]);
    atomicAdd(&g_bwd_stamps[8], stamp_acc[6]);
  }
#else
  // Persistent loop of the backward pass (rows by ticket, a group's rows interleaved with the other groups': ANI_PERSISTENT_LOOP).
  // A centre ends with its force atomics, and the memory queue is in order: loads issued behind them return behind them, so a
  // wave that asks for its next centre's lists AFTER the scatter waits a whole atomic round trip per centre (without the global
  // atomics the kernel takes 0.205 instead of 0.254 ms).  So the next centre's header (a scalar load: its own counter) and lists
  // are requested in front of the final scatter, when only the registers of the centre's tail are live.
  ANI_TICKETS_BEGIN(kWavesB)
  const int kstop = tk_ctr ? tk_base + tk_share : kend;
  auto rank_of = [&](int kx) -> int { return tk_ctr ? (kx - tk_base) * tk_ngrp + tk_grp : kx; };
  auto row_of = [&](int kk) -> int {
    const int kc = kk < a.kcount ? kk : a.kcount - 1;
    return a.row_list ? __builtin_amdgcn_readfirstlane(a.row_list[a.k0 + kc]) : a.k0 + kc;
  };
  bool have = false;   // hc / cur already hold the centre of ticket k (requested in front of the previous centre's final scatter)
  hdr_t hc;
  Loaded<NCH, true, GR> cur;
  while (k < kstop) {
    ANI_TICKET_DRAW(tk_tn)
    const int kk = rank_of(k);
    const int row = row_of(kk);
    const bool in_range = kk < a.kcount;
    if (!have) {
      hc = load_header_scalar(a, row);
      if (hc[0] >= 0 && in_range) load_lists(p, a, row, hdr_nrad(hc), hdr_nang(hc), lane, cur);
    }
    have = false;
    ANI_TICKET_TAKE(tk_k2)
    if (hc[0] >= 0 && in_range) {
      // header and lists go in by value: the hook overwrites hc / cur with the next centre's while this one finishes
      backward_centre<NA, NZ, NCH, GR, VIR>(p, a, L, row, hc, cur, lane, wv, rep, er, [&]() {
        const int kk1 = rank_of(k1);
        if (k1 < kstop && kk1 < a.kcount) {   // wave-uniform
          const int row1 = row_of(kk1);
          hc = load_header_scalar(a, row1);
          if (hc[0] >= 0) load_lists(p, a, row1, hdr_nrad(hc), hdr_nang(hc), lane, cur);
          have = true;
        }
      });
    }
    ANI_TICKET_NEXT(tk_tn, tk_k2)
  }
  ANI_TICKETS_END(kWavesB)
#endif
  if (rep.on) {
    double se = (double)er;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) se += __shfl_xor(se, off);
    if (lane == 0) atomicAdd(&rep.erep[(blockIdx.x * kWavesB + wave) & (kVirialSlots - 1)], se);
  }
  if constexpr (VIR) {
    if (a.virial) {
#pragma unroll
      for (int k = 0; k < 9; k++) {
        double sv = (double)wv[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sv += __shfl_xor(sv, off);
        if (lane == 0) atomicAdd(&a.virial[9 * ((blockIdx.x * kWavesB + wave) & (kVirialSlots - 1)) + k], -sv);
      }
    }
  }
}

// =====================================================================================================
// generic path (any model shape): LDS float atomics, precise libm transcendentals
// =====================================================================================================
struct WaveLds {
  float dx[kMaxRad], dy[kMaxRad], dz[kMaxRad], r[kMaxRad], fc[kMaxRad];
  int sp[kMaxRad], j[kMaxRad];
  int ang[kMaxAng];
  float fca[kMaxAng];
  float row[kAevMax];
  float gd[3 * kMaxRad];
};

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

__global__ __launch_bounds__(64 * kWaves) void aev_forward_generic(AevParams p, AevArgs a) {
  __shared__ WaveLds lds[kWaves];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * kWaves + wave;
  if (row >= a.nrows) return;
  const int ii = a.centre_of_row[row];
  if (ii < 0) return;
  WaveLds& L = lds[wave];

  for (int e = lane; e < p.aev_stride; e += 64) L.row[e] = 0.f;
  int nrad, nang;
  bool over;
  compact_neighbours(p, a, ii, lane, L, nrad, nang, over);
  if (over && lane == 0) atomicOr(a.err_flag, 1);
  wave_sync();

  const int nR = p.nR;
  for (int t = lane; t < nrad * nR; t += 64) {
    const int q = t / nR, k = t - q * nR;
    const float dr = L.r[q] - p.ShfR[k];
    const float v = 0.25f * expf(-p.EtaR * dr * dr) * L.fc[q];
    atomicAdd(&L.row[L.sp[q] * nR + k], v);
  }
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

__global__ __launch_bounds__(64 * kWaves) void aev_backward_generic(AevParams p, AevArgs a) {
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
    for (int z = 0; z < p.nZ; z++) {
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

  float fx = 0.f, fy = 0.f, fz = 0.f;
  float v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int q = lane; q < nrad; q += 64) {
    const float gx = L.gd[3 * q], gy = L.gd[3 * q + 1], gz = L.gd[3 * q + 2];
    const int j = L.j[q];
    atomicAdd(&a.fbuf[4 * j + 0], -gx);
    atomicAdd(&a.fbuf[4 * j + 1], -gy);
    atomicAdd(&a.fbuf[4 * j + 2], -gz);
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
    atomicAdd(&a.fbuf[4 * i + 0], fx);
    atomicAdd(&a.fbuf[4 * i + 1], fy);
    atomicAdd(&a.fbuf[4 * i + 2], fz);
  }
  if (a.virial) {
#pragma unroll
    for (int k = 0; k < 9; k++) {
      float sv = v[k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) sv += __shfl_xor(sv, off);
      if (lane == 0) atomicAdd(&a.virial[k], -(double)sv);
    }
  }
}

int aev_read_stamps(unsigned long long* out32, int reset) {
#ifdef ABLB_STAMPS
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_bwd_stamps), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_stamps), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 2;
#else
  (void)out32; (void)reset;
  return 0;
#endif
}

// =====================================================================================================
// launchers
// =====================================================================================================
static int fast_kind(const AevParams& p) {
  if (p.nR != 16 || p.S > 8 || (p.aev_stride & 3) || !p.equi) return 0;
  {
    // the backward kernel's radial Gaussians come from a recurrence around the midpoints of the two halves of the shifts
    // (backward_centre): its ratio H^2 = exp(2 EtaR dc Delta) must stay inside fp32 for every r in (0, Rcr], 1 / C_3.5 =
    // exp(EtaR (3.5 Delta)^2) too, and where E = exp(-EtaR dc^2) underflows the half's largest term must be negligible.
    // Parameter sets beyond that take the generic kernels.
    const float d = fabsf(p.dShfR), c0 = p.ShfR0 + 3.5f * p.dShfR, c1 = p.ShfR0 + 11.5f * p.dShfR;
    const float dmax = fmaxf(fmaxf(fabsf(p.Rcr - c0), fabsf(c0)), fmaxf(fabsf(p.Rcr - c1), fabsf(c1)));
    const float q = sqrtf(87.f / p.EtaR) - 3.5f * d;
    if (2.f * p.EtaR * dmax * d > 80.f || p.EtaR * 12.25f * d * d > 80.f || q <= 0.f || p.EtaR * q * q < 20.f) return 0;
  }
  if (p.nA == 8 && p.nZ == 4) return 1;
  if (p.nA == 4 && p.nZ == 8) return 2;
  return 0;
}

// Capacity of the per-centre radial list in LDS.  pyaev semantics keep every list entry, so the longest list decides.
// With the radial screen at Rcr only the entries inside Rcr are kept -- for the reference's settings (5.1 A inside a
// 7.1 A list) 37 % of a uniform neighbourhood -- so 3/4 of the longest list (at least 128 slots) is reserved instead of
// all of it; the smaller LDS slice is what lets more waves share a CU.  A centre that still overflows raises the
// capacity error like any other overflow (never a silent truncation), and AevParams::full_cap restores the full size.
#ifndef ANI_CAP_GRAIN
#define ANI_CAP_GRAIN 64
#endif
static int radial_cap(const AevParams& p, int max_numneigh) {
  int full = (max_numneigh + 63) / 64 * 64;
  if (full < 64) full = 64;
  if (p.compat || p.full_cap) return full;
  int est = (3 * max_numneigh + 3) / 4;
  if (est < 128) est = 128;
  est = (est + ANI_CAP_GRAIN - 1) / ANI_CAP_GRAIN * ANI_CAP_GRAIN;   // 32 (a fifth forward workgroup per CU) measured slower: 0.246 against 0.230 ms
  return est < full ? est : full;
}

bool aev_fast_path(const AevParams& p, int max_numneigh) {
  if (!fast_kind(p)) return false;
  // a backward workgroup (kWavesB slices) and a forward workgroup (kWaves slices) must each fit in 160 KB
  const int c = radial_cap(p, max_numneigh);
  return (size_t)fast_wave_floats(c, true) * 4 * kWavesB <= 160 * 1024 && (size_t)fast_wave_floats(c, false) * 4 * kWaves <= 160 * 1024;
}

static int num_cus() {
  static const int n = [] {
    int dev = 0, v = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
    return v > 0 ? v : 256;
  }();
  return n;
}
static const char* waves_env() { return "ANI_AEV_WAVES_PER_CU"; }  // experiment knob: cap on resident waves per CU
// persistent grid of the fast path: as many workgroups as fit on the chip by LDS (at most 8 per CU)
template <typename K>
static int persistent_blocks(K kernel, int nrows, int waves_per_block, size_t lds_bytes) {
  // resident workgroups per CU by registers AND LDS; a grid larger than what is resident would run a second,
  // mostly empty round of persistent workgroups
  // the occupancy query costs microseconds of host time per launch: asked once per (kernel, LDS size)
  static std::map<std::pair<const void*, size_t>, int> cache;
  static std::mutex mtx;
  int per_cu = 0;
  {
    std::lock_guard<std::mutex> lock(mtx);
    const auto key = std::make_pair((const void*)kernel, lds_bytes);
    const auto it = cache.find(key);
    if (it != cache.end()) per_cu = it->second;
    else {
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 64 * waves_per_block, lds_bytes) != hipSuccess || per_cu < 1)
        per_cu = (int)((160 * 1024) / (lds_bytes ? lds_bytes : 1));
      cache[key] = per_cu;
    }
  }
  if (per_cu > 8) per_cu = 8;
  static const int forced = [] { const char* e = getenv(waves_env()); return e ? atoi(e) : 0; }();
  if (forced > 0 && forced * waves_per_block / 2 >= 1) per_cu = std::min(per_cu, std::max(1, forced / waves_per_block));
  if (per_cu < 1) per_cu = 1;
  const int need = (nrows + waves_per_block - 1) / waves_per_block;
  const int fit = num_cus() * per_cu;
  return need < fit ? need : fit;
}

template <typename K, typename... Extra>
static void launch_fast(K kernel, const AevParams& p, const AevArgs& a, int waves, size_t lds, int cap, int rowf, hipStream_t st,
                        Extra... extra) {
  // raising the dynamic-LDS limit is per kernel: once per instantiation
  static std::set<std::pair<int, const void*>> raised;   // per device: a process may drive several
  static std::mutex mtx;
  {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mtx);
    if (raised.insert(std::make_pair(dev, (const void*)kernel)).second)
      note_launch_error(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  hipLaunchKernelGGL(kernel, dim3(persistent_blocks(kernel, a.kcount, waves, lds)), dim3(64 * waves), lds, st, p, a, cap, rowf,
                     extra...);
}

int aev_compact_stride(const AevParams& p, int max_numneigh) {
  return aev_fast_path(p, max_numneigh) ? kMaxAng + radial_cap(p, max_numneigh) : 0;
}

void launch_nbr_compact(const AevParams& p, const AevArgs& a, int max_numneigh, hipStream_t st) {
  if (a.kcount <= 0 || !aev_fast_path(p, max_numneigh)) return;
  const int cap = radial_cap(p, max_numneigh);
  auto go = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, dim3(persistent_blocks(kernel, a.kcount, kWavesC, 0)), dim3(64 * kWavesC), 0, st, p, a, cap);
  };
  if (max_numneigh <= 128) go(nbr_compact_kernel<2>);
  else if (max_numneigh <= 192) go(nbr_compact_kernel<3>);
  else go(nbr_compact_kernel<4>);
}

void launch_aev_forward(const AevParams& p, const AevArgs& a, int max_numneigh, hipStream_t st) {
  if (a.nrows <= 0) return;
  const dim3 grid((a.nrows + kWaves - 1) / kWaves), block(64 * kWaves);
  if (aev_fast_path(p, max_numneigh)) {
    if (a.kcount <= 0) return;   // an empty range of rows
    const int cap = radial_cap(p, max_numneigh);
    const int rowf = (p.aev_stride + 63) / 64 * 64;
    const size_t lds = (size_t)fast_wave_floats_row(cap, false, rowf, p.S) * 4 * kWaves;
    // 64-entry chunks of the radial-only stream prefetched per centre: 1 with the radial screen on (more than 64
    // neighbours between Rca and Rcr are loaded in place), 3 in pyaev mode where every candidate stays
    const bool k1 = fast_kind(p) == 1, n2 = !p.compat;
    if (k1 && n2) launch_fast(aev_forward_fast<8, 4, 1>, p, a, kWaves, lds, cap, rowf, st);
    else if (k1) launch_fast(aev_forward_fast<8, 4, 3>, p, a, kWaves, lds, cap, rowf, st);
    else if (n2) launch_fast(aev_forward_fast<4, 8, 1>, p, a, kWaves, lds, cap, rowf, st);
    else launch_fast(aev_forward_fast<4, 8, 3>, p, a, kWaves, lds, cap, rowf, st);
  } else {
    hipLaunchKernelGGL(aev_forward_generic, grid, block, 0, st, p, a);
  }
}

// compaction + forward in one launch; false: not applicable (the caller launches the two kernels)
bool launch_aev_forward_fused(const AevParams& p, const AevArgs& a, int max_numneigh, hipStream_t st) {
  if (a.nrows <= 0 || a.kcount <= 0 || !aev_fast_path(p, max_numneigh) || max_numneigh > 256) return false;
  const int cap = radial_cap(p, max_numneigh);
  const int rowf = (p.aev_stride + 63) / 64 * 64;
  const size_t lds = (size_t)fast_wave_floats_row(cap, false, rowf, p.S) * 4 * kWaves;
  const bool k1 = fast_kind(p) == 1;
  const int nch = max_numneigh <= 128 ? 2 : (max_numneigh <= 192 ? 3 : 4);
  if (k1) {
    if (nch == 2) launch_fast(aev_forward_fused<8, 4, 2>, p, a, kWaves, lds, cap, rowf, st);
    else if (nch == 3) launch_fast(aev_forward_fused<8, 4, 3>, p, a, kWaves, lds, cap, rowf, st);
    else launch_fast(aev_forward_fused<8, 4, 4>, p, a, kWaves, lds, cap, rowf, st);
  } else {
    if (nch == 2) launch_fast(aev_forward_fused<4, 8, 2>, p, a, kWaves, lds, cap, rowf, st);
    else if (nch == 3) launch_fast(aev_forward_fused<4, 8, 3>, p, a, kWaves, lds, cap, rowf, st);
    else launch_fast(aev_forward_fused<4, 8, 4>, p, a, kWaves, lds, cap, rowf, st);
  }
  return true;
}

bool launch_aev_backward(const AevParams& p, const AevArgs& a, int max_numneigh, hipStream_t st, const RepTab* rep) {
  if (a.nrows <= 0) return true;
  RepTab rt{};
  if (rep) rt = *rep;
  const dim3 grid((a.nrows + kWaves - 1) / kWaves), block(64 * kWaves);
  if (aev_fast_path(p, max_numneigh)) {
    if (a.kcount <= 0) return true;
    const int cap = radial_cap(p, max_numneigh);
    const int rowf = (p.aev_stride + 63) / 64 * 64;
    const size_t lds = (size_t)fast_wave_floats_row(cap, true, rowf, p.S) * 4 * kWavesB;
    const bool k1 = fast_kind(p) == 1, n2 = !p.compat, g1 = p.aev_stride <= 256;
#define ANI_BWD_CASE(NA, NZ, NCH, GR)                                                                   \
  do {                                                                                                  \
    if (a.virial) launch_fast(aev_backward_fast<NA, NZ, NCH, GR, true>, p, a, kWavesB, lds, cap, rowf, st, rt);  \
    else launch_fast(aev_backward_fast<NA, NZ, NCH, GR, false>, p, a, kWavesB, lds, cap, rowf, st, rt);          \
  } while (0)
    if (k1) {
      if (n2 && g1) ANI_BWD_CASE(8, 4, 1, 1);
      else if (n2) ANI_BWD_CASE(8, 4, 1, 4);
      else if (g1) ANI_BWD_CASE(8, 4, 3, 1);
      else ANI_BWD_CASE(8, 4, 3, 4);
    } else {
      if (n2 && g1) ANI_BWD_CASE(4, 8, 1, 1);
      else if (n2) ANI_BWD_CASE(4, 8, 1, 4);
      else if (g1) ANI_BWD_CASE(4, 8, 3, 1);
      else ANI_BWD_CASE(4, 8, 3, 4);
    }
#undef ANI_BWD_CASE
    return true;
  }
  hipLaunchKernelGGL(aev_backward_generic, grid, block, 0, st, p, a);
  return false;
}

}  // namespace ani
