// ani_kernels.h — launch wrappers of the HIP kernels (ani_kernels_*.hip).  Internal to libani_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ani_model.h"

namespace ani {

constexpr int kRowTile = 128;   // AEV rows per GEMM block; species buckets are padded to this
constexpr int kMaxRad = 256;    // per-centre capacity of the radial neighbour list held in LDS
constexpr int kMaxAng = 96;     // per-centre capacity of the angular neighbour list held in LDS
constexpr int kBucketInfoInts = 2 * kMaxSpecies + 4;

enum Epilogue { EPI_PLAIN = 0, EPI_CELU = 1, EPI_LAST = 2, EPI_BWD = 3 };

// C[rows][N] = epi( A[rows][K] * Bt[N][K]^T ), fp32 MFMA.  Batched over blockIdx.y (ensemble members).
struct GemmArgs {
  const float* A;
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
};
void launch_gemm(const GemmArgs& g, Epilogue epi, hipStream_t st);
// one launch for several problems of the same epilogue (all species buckets of one layer)
void launch_gemm_group(const GemmArgs* probs, int nprob, Epilogue epi, hipStream_t st);

// x (double [ntotal*3]) + species -> float4 {x,y,z,bits of cmap[species]}
struct SpeciesMap { int m[kMaxSpecies]; };
void launch_pack(const double* d_x, const int* d_species, int ntotal, const SpeciesMap& cmap, float4* xyzs, hipStream_t st);

// rebuild-time preparation: neighbour offsets, species buckets.  All outputs device arrays.
struct PrepOut {
  int* nbr_off;         // [nlocal+1] exclusive scan of numneigh
  int* row_of_centre;   // [nlocal]
  int* centre_of_row;   // [nrows_cap] (-1 = padding)
  int4* row_info;       // [nrows_cap] per AEV row: {i, list begin, list length, ii}; i = -1 for padding rows
  int* bucket_info;     // [kBucketInfoInts]: count[s], row_start[s], nrows, bad_species flag, max numneigh, species-present mask
};
void launch_prepare(const int* d_species, const int* d_ilist, const int* d_numneigh, int nlocal, int ntotal, int S, int nrows_cap,
                    const PrepOut& o, hipStream_t st);

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
  float* fbuf;       // backward: [ntotal*3] float accumulators (Hartree/Angstrom), atomically added
  double* virial;    // backward: [9] (Hartree), atomically added; may be NULL
  int* err_flag;     // set to 1 on LDS capacity overflow
};
// max_numneigh (known at rebuild) sizes the per-centre LDS neighbour lists of the fast path
void launch_aev_forward(const AevParams& p, const AevArgs& a, int max_numneigh, hipStream_t st);
void launch_aev_backward(const AevParams& p, const AevArgs& a, int max_numneigh, hipStream_t st);
bool aev_fast_path(const AevParams& p, int max_numneigh);
// rebuild time: stable sort of every centre's neighbour segment by neighbour species (jin -> jout)
void launch_sort_jlist(const int* d_species, const int* d_nbr_off, const int* d_numneigh, const int* d_jin, int* d_jout,
                       int nlocal, int S, hipStream_t st);

// energy reduction (+ self energies), per-centre energies, force conversion
struct FinishArgs {
  const float* e_rows;   // [M][nrows_ld] per-member row energies (already scaled by 1/M)
  int M, nrows, nrows_ld;
  const int* centre_of_row;
  const int* ilist;
  const int* species;    // [ntotal]
  double sae[kMaxSpecies];
  const float* fbuf;     // [ntotal*3]
  int ntotal;
  const double* virial_acc;  // [9] Hartree (unsymmetrised), or NULL
  double* f_out;         // [ntotal*3] kcal/mol/A
  int f_accumulate;      // 1: +=, 0: overwrite
  double* ev_out;        // [10]
  double* eatom_out;     // [nlocal] indexed by centre, or NULL
  double* partial;       // [256] scratch
  const int* err_flag;   // capacity overflow flag: energy becomes NaN so device-resident callers notice
};
void launch_finish(const FinishArgs& a, hipStream_t st);

}  // namespace ani
