// ani_kernels.h — launch wrappers of the HIP kernels (ani_kernels_*.hip).  Internal to libani_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "ani_model.h"

namespace ani {

constexpr int kRowTile = 128;   // AEV rows per GEMM block; species buckets are padded to this
constexpr int kMaxRad = 256;    // per-centre capacity of the radial neighbour list held in LDS
constexpr int kMaxAng = 96;     // per-centre capacity of the angular neighbour list held in LDS
constexpr int kBucketInfoInts = 2 * kMaxSpecies + 5;
constexpr int kVirialSlots = 1024;  // rows of 9 doubles the waves of the AEV backward spread their virial sums over

enum Epilogue { EPI_PLAIN = 0, EPI_CELU = 1, EPI_LAST = 2, EPI_BWD = 3 };

// The launch wrappers below return void; a failing HIP call inside one (a memset, a function attribute) is noted here and
// the entry point that issued the work reports it (take_launch_error() returns the first failure since the last take and
// clears it).  One collector per host thread: a handle is driven from one thread at a time (include/ani_hip.h).
void note_launch_error(hipError_t e);
hipError_t take_launch_error();

// C[rows][N] = epi( A[rows][K] * Bt[N][K]^T ), fp32 MFMA.  Batched over blockIdx.y (ensemble members).
struct GemmArgs {
  const float* A;
  const float* Amask;      // optional, same shape/strides as A: the stored activation H of the layer A was differentiated
                           // through; the kernel then multiplies A by celu'(z) = (H > 0 ? 1 : H/alpha + 1) while staging it
  const float* Bt;
  float* C;
  const float* bias;       // [N] per batch            (CELU, LAST)
  const float* aux;        // BWD: stored activation H[rows][N] of the layer being differentiated through
                           // LAST: output-layer weight vector w[N] per batch
  const float* bias_last;  // LAST: output-layer bias, one per batch
  float* e_out;            // LAST: per-row energy contribution of this batch, [rows] per batch
  const int* centre_of_row;  // >= 0 for real rows, -1 for bucket padding
  long long sA, sB, sC, sBias, sAux, sE;  // per-batch strides in elements
  int lda, ldb, ldc, ldaux;
  int row0, rows;          // rows is a multiple of kRowTile
  int N, K;                // K multiple of 4; operands zero-padded accordingly
  int batch;
  float scale;             // LAST: 1/num_models
  float alpha, inv_alpha;  // CELU
  // split paths: Bt as 16-bit planes -- three bf16 (x = hi + mid + lo exactly) or two fp16 (x 2^s = h + l to 2^-22) --
  // blocked [kbp][N][planes][16] with kbp = ceil(K/16) 16-k blocks per row (zero padded); sBp = per-batch stride in
  // 16-bit elements.  NULL: only the fp32 path can run.
  const unsigned short* Btp;
  long long sBp;
  int kbp;
  // two-term fp16 path: A is multiplied by a_scale (a power of two) while it is staged, the accumulators by
  // inv_scale = 1 / (a_scale * scale of the weight planes) before the epilogue
  float a_scale, inv_scale;
};
// arithmetic of the MLP products (ani_set_option "mlp_arith")
enum MlpArith {
  MLP_FP32 = 0,     // v_mfma_f32_32x32x2_f32
  MLP_BF16X3 = 1,   // six v_mfma_f32_32x32x16_bf16 products of the exact three-way bf16 splits of both operands
  MLP_F16X2 = 2     // three v_mfma_f32_32x32x16_f16 products of two-term fp16 splits of the scaled operands
};
inline int mlp_planes(MlpArith a) { return a == MLP_F16X2 ? 2 : 3; }
void launch_gemm(const GemmArgs& g, Epilogue epi, hipStream_t st);
// one launch for several problems of the same epilogue (all species buckets of one layer).
// arith: see MlpArith; the problems' Btp / a_scale / inv_scale must have been set up for it
void launch_gemm_group(const GemmArgs* probs, int nprob, Epilogue epi, hipStream_t st, MlpArith arith);
// Small systems (about one 64-row tile per CU or less): the six products of the MLP as ONE launch.  A workgroup takes a
// row tile through all layers in order — every layer of a tile reads only what the same workgroup wrote for the layer
// before (activations go through global memory, L2-hot) — which removes five launch/drain gaps and the per-launch
// prologues that dominate when a launch is ~20 us.  layers[l * nprob + p]: problem p (species bucket) of layer l, all
// with the same `rows`; epi[l]: its epilogue.  batch must be 1 (one ensemble member).  `plan` caches the device copy.
struct ChainPlan {
  void* d_desc = nullptr;      // device: GemmArgs[nlayers * nprob], then int epi[nlayers], int tile_start[nprob + 1]
  size_t bytes = 0;
  int* d_done = nullptr;       // pipeline launch: one completion flag per (layer, tile)
  size_t done_n = 0;
  std::vector<unsigned char> host;
};
// returns the HIP status of the descriptor upload (a failed allocation must not let the step run on stale activations)
// pipeline = true (large systems, two-term arithmetic, layers no wider than 256): persistent workgroups walk (layer, tile)
// items in layer-major order with a completion flag per item instead of a workgroup per tile (mlp_pipeline_x2); err_flag
// gets bit 2 if a wait ever runs into its bound
hipError_t launch_mlp_chain(const GemmArgs* layers, const int* epi, int nlayers, int nprob, ChainPlan* plan, hipStream_t st, MlpArith arith,
                            bool pipeline = false, int* err_flag = nullptr);
void free_chain_plan(ChainPlan& p);
int mlp_chain_slots();   // workgroups the chain kernel can keep resident (2 per CU)

// ---- the whole MLP of a row tile in one workgroup (ani_kernels_mlpf.hip) ----------------------------------------------
// Networks with three hidden layers whose widths fit one of the compiled shapes (fused_shape_for >= 0), AEV width a multiple
// of 16, split arithmetic.  Per species bucket: the weight stream (1 KB pieces in consumption order, members back to back:
// fused_pieces_per_member each) and the constants block (fused_consts_floats floats per member: b0 | b1 | b2 | w3 padded to
// the shape's tiles, then {b3, 1/scale of the six products, 0}).
constexpr int kMaxProblems = 16;   // species buckets per launch
struct FusedProb {
  const float* aev;             // [rows][aev_stride] rows of the bucket
  float* gaev;                  // [rows][aev_stride] dE/dAEV out
  long long gaev_row0;          // first row of the problem in the step's dE/dAEV array (FusedArgs::member_items)
  float* e_rows;                // member m's row energies at e_rows + m * sE
  const int* centre_of_row;     // >= 0 for real rows
  const unsigned char* stream;
  const float* consts;
  long long sE;
  int tiles;                    // 128-row tiles
  int shape;                    // index of the compiled shape
  int ks0, nt0, acols, aev_stride;   // AEV: 16-column k-steps, 32-column tiles, columns in use, row stride
  int pieces_per_member, consts_per_member;
};
struct FusedArgs {
  FusedProb p[kMaxProblems];
  int tile_start[kMaxProblems + 1];   // prefix of tiles per problem, costliest problems first
  int nprob, M;
  int member_items;                   // 1: a work item is (tile, member) and member m writes its dE/dAEV rows to
                                      //    gaev_parts + m * part_stride (summed afterwards: launch_sum_parts); 0: a work item is a
                                      //    tile whose members run one after the other, accumulating in gaev
  float* gaev_parts;
  long long part_stride;              // floats per member in gaev_parts (= rows of the whole step x aev_stride)
  float alpha, inv_alpha, scale;      // CELU; scale = 1 / M
  int* counter;                       // device word the workgroups draw tiles from (zeroed by the launcher); unused with a schedule
  const int* sched_items;             // static schedule (fused_schedule): work items of workgroup b = sched_items[sched_off[b] ..
  const int* sched_off;               //   sched_off[b + 1]); null: items are drawn from `counter`
  int sched_blocks;                   // workgroups the schedule was made for (= the grid)
  int* err_flag;                      // device error word: bit 4 = the weight ring's schedule broke (cannot happen: tests/ring_sim.cpp)
};
int fused_shape_for(int d1, int d2, int d3);          // -1: no compiled shape holds these widths
void fused_shape_tiles(int shape, int nt[3]);         // 32-feature tiles of the three hidden layers
int fused_consts_floats(int shape);
long long fused_pieces_per_member(int shape, int acols, int P);
// one product's share of a stream: src[row][k] (ld) -> NT x KS blocks of P pieces at dst (see ani_kernels_mlpf.hip);
// chunk: 0, -1 (tile-major: the hidden backward products) or 4 for the dE/dAEV product (the kernel walks four output tiles
// at a time through all k-steps)
void launch_build_stream(const float* src, int ld, int rows_valid, int k_valid, int NT, int KS, int chunk, int P, float scale,
                         unsigned short* dst, hipStream_t st);
hipError_t launch_mlp_fused(const FusedArgs& G, MlpArith arith, hipStream_t st);
// The sixteen-rows-per-wave form of the same kernel (ani_kernels_mlpg.hip): waves = 8 takes the 128-row tiles of FusedArgs as
// they are (two waves per SIMD), waves = 4 reads FusedProb::tiles / tile_start in 64-row tiles (one wave per SIMD, for launches
// that would leave CUs idle).  Its weight stream has its own order (launch_build_stream16, fused16_pieces_per_member); the
// constants block is the 32-row kernel's.  FusedProb::ks0 / nt0 are not used (derived from acols).
hipError_t launch_mlp_fused16(const FusedArgs& G, MlpArith arith, int waves, hipStream_t st);
long long fused16_pieces_per_member(int shape, int acols, int P);
int fused16_b1_chunks(int nt0);                 // dE/dAEV chunks of nt0 16-column tiles ...
int fused16_b1_chunk_tiles(int nt0, int ci);    // ... and the tiles of chunk ci (16, then 8, then the rest)
// NT x KS blocks (16-row output tiles from tile nt_off on, 32-deep k-steps) of src[row][k]; order 0: k-step-major, 1: tile-major;
// identity: k-slots in column order (the AEV operand of the first product) instead of the accumulator order
void launch_build_stream16(const float* src, int ld, int rows_valid, int k_valid, int NT, int KS, int order, int nt_off, int identity,
                           int P, float scale, unsigned short* dst, hipStream_t st);
int fused_num_cus();
// Static schedule of a launch: `nitem_types` kinds of work items (type j: count[j] items of relative cost[j], items numbered
// type after type), `bins` workgroups.  Multifit: the smallest makespan T for which first-fit-decreasing packs every item into
// the bins.  items_out[sum count]: item numbers, workgroup after workgroup; off_out[bins + 1].  Returns the makespan.
// (Drawing items from a counter, costliest first, is list scheduling: at 100 002 water atoms -- 521 + 261 tiles of cost 1 and
// 0.67 on 256 CUs -- its last 14 tiles start when most CUs have finished, makespan 3.35; first-fit finds 3.0.)
double fused_schedule(int nitem_types, const int* count, const double* cost, int bins, int* items_out, int* off_out);
// the same with half items for the sixteen-row kernel (ids total + 2 i + h); see ani_kernels_mlpf.hip
double fused_schedule_halves(int nitem_types, const int* count, const double* cost, double half_ratio, int bins, int split_mode,
                             int* split_out, int* items_out, int* off_out, int* n_items_out, double min_gain = 0.08);
// dst[i] = sum over m < M of parts[m * part_stride + i], i < n (n a multiple of 4, 16-byte aligned pointers)
void launch_sum_parts(const float* parts, long long part_stride, int M, float* dst, long long n, hipStream_t st);
// diagnostic builds (-DABLF_STAMPS) only: cycles per phase summed over tiles; returns 0 in the shipped build
int aev_read_stamps(unsigned long long* out32, int reset);
int fused_read_stamps(unsigned long long* out16, int reset);

// dst[kb][N][planes][16] 16-bit planes of src[N][ld] (first K columns), kb = ceil(K/16); batch matrices src + i*s_src ->
// dst + i*split_elems(N, K, arith).  scale: power of two applied to the weights of the fp16 path before they are split
void launch_split_planes(const float* src, int batch, long long s_src, int N, int K, int ld, MlpArith arith, float scale,
                         unsigned short* dst, hipStream_t st);
inline size_t split_elems(int N, int K, MlpArith arith) { return (size_t)N * ((K + 15) / 16) * 16 * mlp_planes(arith); }

// x (double [ntotal*3]) + species -> float4 {x,y,z,bits of cmap[species]}
struct SpeciesMap { int m[kMaxSpecies]; };
// also clears this step's accumulators: fbuf[4*ntotal], virial_acc[kVirialSlots*9] and the 10 doubles of ev_zero (the
// energy/virial output the finish kernel adds to; may be NULL) -- no separate memsets
// positions are stored relative to d_origin[3] (device; launch_origin sets it to the midpoint of the atoms' bounding box)
void launch_origin(const double* d_x, int ntotal, double* d_origin, hipStream_t st);
// atoms [i0, i1); virial_acc / ev_zero may be NULL (a second launch of a step for the ghost atoms clears neither)
// ghosts that are images of this rank's own atoms (ani_set_ghost_fold): ghost g (atom nlocal + g) = atom owner[g] displaced by shift[g]
struct GhostFold {
  const long long* owner = nullptr;
  const double* shift = nullptr;
  int nlocal = 0;
};
void launch_ghost_chain(const long long* d_owner, int nghost, int nlocal, int* d_head, int* d_next, int* d_bad, hipStream_t st);
void launch_pack(const double* d_x, const int* d_species, int i0, int i1, const SpeciesMap& cmap, float4* xyzs, float* fbuf,
                 double* virial_acc, double* ev_zero, const double* d_origin, hipStream_t st, const GhostFold* gf = nullptr);
// rebuild time: list[0 .. count[0]) = the rows with a ghost atom (index >= nlocal) among their candidates, ascending; then
// the other rows, ascending.  flag: scratch, [nrows]
void launch_row_classes(const int4* row_info, const int* jlist, int nrows, int nlocal, int* flag, int* list, int* count, hipStream_t st);

// {energy, virial[9], error word, stamp} of a step -> host_out[12] (page-locked, device-visible host memory)
void launch_step_tail(const double* d_ev, const int* d_flag, double* host_out, double stamp, hipStream_t st);
// d_f[n] -> host[n] (page-locked) in nchunks chunks, host_flags[c] = epoch once chunk c has landed; d_ctr[nchunks] zeroed once
void launch_copy_out(const double* d_f, double* host, long long n, int nchunks, unsigned long long* d_ctr, unsigned* host_flags,
                     unsigned epoch, int nblocks, hipStream_t st);

// rebuild-time preparation: neighbour offsets, species buckets.  All outputs device arrays.
struct PrepOut {
  int* nbr_off;         // [nlocal+1] exclusive scan of numneigh (row_stride = 0) or ii * row_stride
  int* row_of_centre;   // [prepare_scratch_ints(nlocal)]: [nlocal] rows, then scratch
  int* centre_of_row;   // [nrows_cap] (-1 = padding)
  int4* row_info;       // [nrows_cap] per AEV row: {i, list begin, list length, ii}; i = -1 for padding rows
  int* bucket_info;     // [kBucketInfoInts]: count[s], row_start[s], nrows, bad_species flag, max numneigh, species-present mask, pairs
  int* row_of_atom;     // [ntotal] or NULL: the AEV row of an atom that is a centre, -1 for every other atom (ghosts)
};
size_t prepare_scratch_ints(int nlocal);   // size of PrepOut::row_of_centre (rows + scratch of the two kernels)
void launch_prepare(const int* d_species, const int* d_ilist, const int* d_numneigh, int nlocal, int ntotal, int S, int nrows_cap,
                    const PrepOut& o, hipStream_t st, int row_stride = 0);
// 1 in *d_out unless every entry i -> j between two centres of the list has its mirror j -> i (ani_kernels_misc.hip); d_acc: [ntotal] scratch
void launch_list_symmetry(const int* d_ilist, const int* d_nbr_off, const int* d_numneigh, const int* d_jraw, const int* d_row_of_atom,
                          int nlocal, int ntotal, unsigned* d_acc, int* d_out, hipStream_t st);

#ifndef ANI_TK_GROUPS
#define ANI_TK_GROUPS 64
#endif
constexpr int kTicketGroups = ANI_TK_GROUPS;   // row-ticket counters of a fused forward launch (AevArgs::row_counter) ...
constexpr int kTicketMaxGroups = 256;            // what the counter block of a row range has room for
static_assert(kTicketGroups <= kTicketMaxGroups, "ANI_TK_GROUPS");
constexpr int kTicketStride = 64;   // ... one per 256 bytes: {next ticket, waves done, -...}
struct AevArgs {
  const float4* xyzs;
  const int* ilist;
  const int* numneigh;
  const int* nbr_off;
  const int* jlist;
  const int* centre_of_row;
  const int4* row_info;  // [nrows] {centre atom i (-1: padding row), first list slot, list length, centre position ii}
  int nrows;
  float* aev;        // [nrows][aev_stride]
  const float* gaev; // backward only
  float* fbuf;       // backward: [ntotal*4] float accumulators {fx,fy,fz,-} (Hartree/Angstrom), atomically added
  double* virial;    // backward: [kVirialSlots][9] (Hartree) partial sums, atomically added (fast path: slot = wave id
                     // mod kVirialSlots; other kernels use slot 0); may be NULL
  int* err_flag;     // set to 1 on LDS capacity overflow
  // per-step compact neighbour lists of the fast path (nbr_compact_kernel writes, forward / backward read)
  int4* cl_hdr;      // [2*nrows] {i (-1: skip), nrad | nang<<16 | centre species<<24, 8 x u8 angular counts} | 8 x u16 radial-only counts
  float4* cl_xyz;    // [nrows*cl_stride] {dx, dy, dz, r}: kMaxAng slots for the neighbours inside Rca, then the radial-only ones
  int* cl_j;         // [nrows*cl_stride] neighbour atom index (low 28 bits) | compact species << 28
  int cl_stride;     // entries reserved per row (kMaxAng + the kernels' radial LDS capacity)
  // the range of rows a fast-path launch walks: row_list[k0 + k], k in [0, kcount); without a list the rows k0 + k
  // (every row: k0 = 0, kcount = nrows).  The generic kernels always take every row.
  const int* row_list;
  int k0, kcount;
  // backward, fast path: AEV row of an atom (-1: not a centre here) -- with it a centre collects BOTH radial terms of a pair with a
  // local neighbour (its own and, read from the neighbour's dE/dAEV row, the neighbour's) and scatters nothing to that neighbour;
  // NULL: every radial gradient is scattered to the neighbour
  const int* row_of_atom;
  // forward launch with the compaction inside: kTicketGroups x {next ticket, waves done} at a stride of kTicketStride ints -- its
  // waves draw their rows from their group's counter instead of a fixed stride (a group's last wave out resets the pair for the
  // next launch); NULL: fixed stride
  int* row_counter;
};
// entries per row the compact lists need (0: the model shape takes the generic kernels, which keep no lists)
int aev_compact_stride(const AevParams& p, int max_numneigh);
// fast path only: screen every centre's candidate list into cl_hdr / cl_xyz / cl_j (once per step, before the forward pass)
void launch_nbr_compact(const AevParams& p, const AevArgs& a, int max_numneigh, hipStream_t st);
// max_numneigh (known at rebuild) sizes the per-centre LDS neighbour lists of the fast path
void launch_aev_forward(const AevParams& p, const AevArgs& a, int max_numneigh, hipStream_t st);
// compaction + forward as ONE launch (fast path, candidate lists of at most 256 entries); false: not applicable
bool launch_aev_forward_fused(const AevParams& p, const AevArgs& a, int max_numneigh, hipStream_t st);
// optional pairwise repulsion folded into the radial stage of the fast backward kernel (tables indexed by the compact
// species of the run, like the AEV layout); `on` = 0: none.  Returns true if the kernel that ran applied it (the fast
// path); false: the caller adds it with launch_repulsion.
struct RepTab {
  int on;
  float cutoff;
  float y[64], sa[64], k[64];   // [8][8], atomic units
  double* erep;                 // [kVirialSlots] energy partial sums (Hartree)
  const double* x64;            // [ntotal*3] the caller's fp64 positions: pair distances of this steep term are taken
                                // from them (a float position in a 100 A box is 4e-6 A off, worth 4e-3 kcal/mol/A
                                // of force on a bonded O-H pair)
};
bool launch_aev_backward(const AevParams& p, const AevArgs& a, int max_numneigh, hipStream_t st, const RepTab* rep = nullptr);
bool aev_fast_path(const AevParams& p, int max_numneigh);
// rebuild time: stable sort of every centre's neighbour segment by neighbour species (jin -> jout)
void launch_sort_jlist(const int* d_species, const int* d_nbr_off, const int* d_numneigh, const int* d_jin, int* d_jout,
                       int nlocal, int S, int present_mask, hipStream_t st, int in_stride = 0);

// device-side neighbour list (ani_kernels_nbr.hip): cells of edge >= cutneigh over [lo, hi)
struct NbrGrid {
  double lo[3], inv[3];  // inv = nc / (hi - lo)
  int nc[3], ncell;
  int reach = 1;         // cells to search either side: 1 for cells no smaller than the cutoff, 2 for half that edge
};
struct NbrScratch {
  int* cell_id;     // [ntotal]
  int* cell_count;  // [ncell+1]
  int* cell_start;  // [ncell+1]
  int* cursor;      // [ncell]
  int* order;       // [ntotal] atoms sorted by cell
  double* xs;       // [ntotal*3] positions in that order
  float4* xq = nullptr;  // [ntotal] fp32 position relative to the grid origin + (atom | species << 28), same order (optional)
};
void launch_nbr_bin(const double* d_x, int ntotal, const NbrGrid& g, const NbrScratch& s, hipStream_t st, const int* d_species = nullptr);
// numneigh[nlocal] and its exclusive scan nbr_off[nlocal+1]
void launch_nbr_count(int nlocal, int ntotal, const NbrGrid& g, const NbrScratch& s, double cutneigh, int* d_numneigh,
                      int* d_nbr_off, hipStream_t st);
// jlist (flattened in atom order) and the identity ilist
void launch_nbr_fill(int nlocal, int ntotal, const NbrGrid& g, const NbrScratch& s, double cutneigh, const int* d_nbr_off,
                     int* d_jlist, int* d_ilist, hipStream_t st);
// one pass instead of launch_nbr_count + launch_nbr_fill: rows of `cap` entries, true counts, their scan, an overflow word
void launch_nbr_onepass(int nlocal, int ntotal, const NbrGrid& g, const NbrScratch& s, double cutneigh, int cap, int* d_numneigh,
                        int* d_nbr_off, int* d_jrows, int* d_ilist, int* d_ovf, hipStream_t st);
// search + species grouping in one kernel into rows of `cap` entries (bins from launch_nbr_bin with species)
bool nbr_sorted_rows_supported(int ntotal, int S, int cap);
void launch_nbr_sorted_rows(int nlocal, int ntotal, const NbrGrid& g, const NbrScratch& s, double cutneigh, int S, int cap,
                            int* d_numneigh, int* d_ilist, int* d_jrows, int* d_ovf, hipStream_t st);

// optional pairwise repulsion (ani_kernels_rep.hip)
struct RepArgs {
  const int4* row_info;   // [nrows] {centre atom, list begin, list length, ii}
  int nrows;
  const int* jlist;
  const int* species;     // [ntotal] model species
  const double* pos;      // [ntotal*3] positions as handed in by the caller (fp64 in both precisions)
  void* fbuf;             // fp32 path: float4[ntotal] accumulators; fp64 path: double[ntotal*3]
  double* virial;         // [vslots][9] or NULL
  double* erep;           // [nslots] energy partial sums (Hartree), zeroed by the caller
  const double* tables;   // device: y_ab[S*S], sqrt_alpha_ab[S*S], k_rep_ab[S*S]
  int S, nslots, vslots;  // slot counts are powers of two
  double cutoff;          // Angstrom
};
void launch_repulsion(const RepArgs& a, bool fp64, hipStream_t st);
void launch_repulsion_energy(const double* erep, int nslots, double* d_ev, hipStream_t st);

// energy reduction (+ self energies), per-centre energies, force conversion
struct FinishArgs {
  const float* e_rows;   // [M][nrows_ld] per-member row energies (already scaled by 1/M)
  int M, nrows, nrows_ld;
  const int* centre_of_row;
  const int* ilist;
  const int* species;    // [ntotal]
  double sae[kMaxSpecies];
  const float* fbuf;     // [ntotal*4] {fx,fy,fz,-}
  int atom0, atom1;      // forces of the atoms [atom0, atom1) are written
  int energy;            // 1: also reduce energy (+ virial); 0: forces only (the ghost atoms' part of a split step)
  const double* virial_acc;  // [kVirialSlots][9] Hartree (unsymmetrised partial sums), or NULL
  double* f_out;         // [ntotal*3] kcal/mol/A
  int f_accumulate;      // 1: +=, 0: overwrite
  double* ev_out;        // [10]; [0] is ADDED to (zeroed by launch_pack), [1..9] written when virial_acc is given
  double* eatom_out;     // [nlocal] indexed by centre, or NULL
  const int* err_flag;   // capacity overflow flag: energy becomes NaN so device-resident callers notice
  const int* fold_head = nullptr;   // ghost fold: chain of the images of every owned atom (launch_ghost_chain), or NULL
  const int* fold_next = nullptr;
  int fold_nlocal = 0;
};
void launch_finish(const FinishArgs& a, hipStream_t st);

// ---- double precision path (ani_kernels_f64.hip) ------------------------------------------------------------
struct Aev64Params {
  int S, nR, nA, nZ, radial_len, aev_len, aev_stride, compat;
  double Rcr, Rca, EtaR, EtaA, Zeta;
  double ShfR[kMaxShfR], ShfA[kMaxShfA], cosZ[kMaxShfZ], sinZ[kMaxShfZ];
};
struct Aev64Args {
  const double* x;         // [ntotal*3]
  const int* species;      // [ntotal] model species
  SpeciesMap cmap;         // species -> index among the species present
  const int* jlist;
  const int4* row_info;
  int nrows;
  double* aev;
  const double* gaev;
  double* fbuf;            // [ntotal*3] Hartree/Angstrom
  double* virial;          // [9] or NULL
  int* err_flag;
};
struct Gemm64Args {
  const double* A; const double* Bt; double* C; const double* bias; const double* aux; const double* bias_last; double* e_out;
  const int* centre_of_row;
  long long sA, sB, sC, sBias, sAux, sE;
  int lda, ldb, ldc, ldaux, rows, N, K, batch;
  double scale, alpha;
};
struct Sae64 { double v[kMaxSpecies]; };
void launch_cvt_f32_f64(const float* src, double* dst, size_t n, hipStream_t st);
void launch_aev64_forward(const Aev64Params& p, const Aev64Args& a, hipStream_t st);
void launch_aev64_backward(const Aev64Params& p, const Aev64Args& a, hipStream_t st);
void launch_gemm64(const Gemm64Args& g, Epilogue epi, hipStream_t st);
void launch_finish64(const double* e_rows, int M, int nrows, const int* centre_of_row, const int* ilist, const int* species,
                     const Sae64& sae, const double* fbuf, int ntotal, const double* vir, double* f_out, int accumulate, double* ev,
                     double* eatom, const int* err_flag, hipStream_t st);

}  // namespace ani
