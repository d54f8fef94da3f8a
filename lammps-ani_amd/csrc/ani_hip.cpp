// ani_hip.cpp — C ABI of libani_hip.so (include/ani_hip.h): model upload, device buffer management and the
// per-step pipeline.  Replaces class ANI of the reference (src/ani_csrc/ani.cpp) without libtorch.
//
// Per step (all on one HIP stream, no per-step allocation once buffers have grown):
//   [ago==0 only]  upload / copy the neighbour list, scan numneigh, bucket the centres by species (rows padded to
//                  128 per species so that every GEMM tile is species-pure)            (ani.cpp:213-229)
//   pack           double positions -> float4 {x,y,z,species}                          (ani.cpp:206-209)
//   AEV forward    -> aev[nrows][aev_stride], rows in species-bucket order             (lammps_ani.py:174)
//   MLP            forward + input-gradient backward on MFMA (fp32 products as six bf16 terms, or fp32-input MFMA) -> e_rows, gaev      (lammps_ani.py:182-184,197)
//   AEV backward   gaev -> forces on local+ghost atoms, virial                         (lammps_ani.py:195-216)
//   finish         fp64 energy sum + self energies, kcal/mol conversion                (ani.cpp:246-262)
#include "../../include/ani_hip.h"
#include "../../include/ani_comm.h"

#include <hip/hip_runtime.h>
#ifndef ANI_NO_ROCTX
#include <rocprofiler-sdk-roctx/roctx.h>
#else   // built without the profiler SDK: the markers (only profiling runs look at them) become no-ops
static inline void roctxRangePushA(const char*) {}
static inline void roctxRangePop() {}
static inline void roctxMarkA(const char*) {}
#endif

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ani_kernels.h"
#include "ani_model.h"

using namespace ani;

namespace ani {
static thread_local hipError_t g_launch_error = hipSuccess;
void note_launch_error(hipError_t e) { if (e != hipSuccess && g_launch_error == hipSuccess) g_launch_error = e; }
hipError_t take_launch_error() { const hipError_t e = g_launch_error; g_launch_error = hipSuccess; return e; }
}  // namespace ani

namespace {

thread_local std::string g_create_error;

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  // grow-only, 1.5x policy like the reference's jlist (src/pair_ani.cpp:119-127); contents are NOT preserved
  hipError_t reserve(size_t n, bool zero = false) {
    if (n <= cap && p) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    size_t want = std::max<size_t>(n + n / 2, 64);
    hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
    if (e != hipSuccess) { cap = 0; return e; }
    cap = want;
    // hipMemset on device memory runs on the null stream and may return before it has run; the handle's work is on
    // non-blocking streams, which do not wait for the null stream: wait here (allocation time only)
    if (zero) {
      e = hipMemset(p, 0, want * sizeof(T));
      if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    }
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// roctx range for the profiler's timeline (the reference marks its phases with NVTX: src/pair_ani.cpp:198-200,
// src/ani_csrc/ani.cpp:128,215); costs nothing measurable when no tool is attached
struct TraceRange {
  explicit TraceRange(const char* name) { roctxRangePushA(name); }
  ~TraceRange() { roctxRangePop(); }
  TraceRange(const TraceRange&) = delete;
  TraceRange& operator=(const TraceRange&) = delete;
};

struct SpeciesNet {
  // device weights; layouts described in compute_mlp()
  std::vector<float*> W;    // [L-1]   W[k]: [M][d[k+1]][kpad(k)]        forward, k = 0..L-2
  std::vector<float*> b;    // [L-1]   b[k]: [M][d[k+1]]
  std::vector<float*> WT;   // [L-1]   WT[k], k = 1..L-2: [M][d[k]][w(k+1)] ; WT[0]: [aev_len][M*w(1)]
  float* W0c = nullptr;     // first layer restricted to the AEV columns of the species present: [M][d[1]][aev_stride']
  float* WT0c = nullptr;    // and its transposed copy [aev_len'][M*w(1)]   (null when every species is present)
  float* w_out = nullptr;   // [M][w(L-1)]
  float* b_out = nullptr;   // [M]
  std::vector<int> w;       // w[k] = round_up(d[k],4), k = 0..L-1 (w[0] = aev_stride)
  // double-precision mirrors of the above (precision 'double' only)
  std::vector<double*> W64, b64, WT64;
  double *W0c64 = nullptr, *WT0c64 = nullptr, *w_out64 = nullptr, *b_out64 = nullptr;
  // 16-bit planes of W / WT / W0c / WT0c for the split MFMA paths (blocked [ceil(K/16)][N][planes][16]):
  // sp[0] three bf16 planes (MLP_BF16X3), sp[1] two fp16 planes of wscale[k] * W (MLP_F16X2)
  struct Planes {
    std::vector<unsigned short*> W, WT;
    unsigned short *W0c = nullptr, *WT0c = nullptr;
  } sp[2];
  std::vector<float> wscale;   // [L-1] power of two that brings the largest |weight| of layer k (all members) below 2^13
  // the fused one-workgroup-per-tile MLP (ani_kernels_mlpf.hip): weight stream and constants per split arithmetic
  // ([0] three bf16 planes, [1] two fp16 planes), built for the AEV layout of the epoch (fused_mask)
  struct Fused {
    unsigned char* stream = nullptr;
    float* consts = nullptr;
    long long ppm = 0;   // pieces per member
    int cpm = 0;         // constants (floats) per member
  } fu[4];   // [arithmetic + 2 * generation]: generation 0 = the 32-rows-per-wave kernel's stream order, 1 = the 16-row kernel's
  int fused_shape = -1;
};
inline MlpArith planes_arith(int i) { return i == 0 ? MLP_BF16X3 : MLP_F16X2; }

}  // namespace

struct ani_handle {
  HostModel model;
  AevParams ap{};       // the model's full AEV layout
  AevParams ap_run{};   // the layout the kernels run with this epoch: ap restricted to the species present (set by specialize();
                        // value-initialised: the entry points look at ap_run.full_cap before the first epoch exists)
  SpeciesMap cmap{};  // species -> index among the species present
  int active_mask = -1;
  bool prune = true;  // ani_set_option("prune_absent_species")
  int mlp_pipeline = 0;   // ani_set_option("mlp_pipeline"): 1 = large systems run all MLP layers as one launch of persistent workgroups
                          // (two-term arithmetic, shapes off the fused kernel).  Off since round 4: one run of the test suite in
                          // three returned a stale tile at 60 000 atoms (tests/test_hip_properties.py) -- an experiment, not a default
                          // with per-tile dependencies, 2 = at any size (measurement knob), 0 = one launch per layer
  int aev_fused = 1;         // ani_set_option("aev_fused"): 1 = neighbour compaction inside the forward AEV launch, 0 = its own kernel
  int aev_sym_radial = 1;        // ani_set_option("aev_symmetric_radial"): see AevArgs::row_of_atom
  int aev_tickets_min = 40000;   // ani_set_option("aev_tickets_min"): launches of fewer rows keep the fixed stride
  int mlp_fused_sched = 1;   // ani_set_option("mlp_fused_schedule"): 1 = static first-fit schedule of the fused launch, 0 = a counter
  int mlp_fused_halves = 1;  // ani_set_option("mlp_fused_halves"): 1 = the schedule may cut straggler items in two (sixteen-row kernel),
                             // 0 = whole items only, 2 = every item as two halves (tests, measurements)
  int sched_nitems = 0;      // entries of the schedule in fused_sched (whole items + halves)
  const char* last_mlp_kernel = "";   // ani_last_mlp_kernel: the kernel that ran the MLP of the last step
  int mlp_fused = 1;   // ani_set_option("mlp_fused"): 1 = networks of three hidden layers run as one launch, a 128-row tile per
                       // workgroup with the activations in registers (ani_kernels_mlpf.hip); 0 = the per-layer kernels
  int fused_mask[4] = {-2, -2, -2, -2};   // active_mask the fused streams of each (arithmetic, generation) were built for
  int mlp_fused_gen = 1;    // ani_set_option("mlp_fused_gen"): 1 = sixteen rows per wave (ani_kernels_mlpg.hip), 0 = thirty-two (mlpf)
  int mlp_fused_rows = 0;   // ani_set_option("mlp_fused_rows"): rows per workgroup of the 16-row kernel: 128, 64, 0 = by size
  DevBuf<int> fused_counter;
  DevBuf<int> fused_sched;    // static schedule of the fused launch: items, then offsets (fused_schedule); remade per list epoch
  int sched_key[4] = {-1, -1, -1, -1};   // what it was made for: total tiles, items per tile, problems, bins
  DevBuf<float> gaev_parts;   // fused MLP with (tile, member) work items: every member's own dE/dAEV rows
  int mlp_chain = 1;   // ani_set_option("mlp_chain"): 1 = one chained launch for the MLP of small systems, 2 = at any size, 0 = never
  ChainPlan chain_plan;
  bool profiling = false;  // ani_set_option("profiling"): every entry point synchronises its stream before returning
  bool dev_overwrite = false;  // ani_set_option("device_overwrite_forces"): ani_compute_full_device writes d_f instead of adding
  MlpArith mlp_arith = MLP_BF16X3;  // ani_set_option("mlp_arith"): how the MLP evaluates its fp32 products (ani_kernels.h); the exact
                                   // split is the default, the reduced-precision fp16 split an opt-in -- the mapping of the reference's
                                   // TF32 switch (off unless LAMMPS_ANI_ALLOW_TF32, src/ani_csrc/ani.cpp:41-43)
  std::vector<int> colmap;  // ap_run column -> ap column
  int device = 0;
  bool use_cuaev = true, use_fullnbr = true, use_single = true;
  hipStream_t stream = nullptr;
  std::string err;
  std::vector<SpeciesNet> nets;

  // list epoch state (valid while ago > 0)
  bool have_list = false;
  // split step (ani_step_*): rows with a ghost among their candidates first in row_list, the others after
  DevBuf<int> row_flag, row_list, row_count;
  int n_boundary = 0;
  bool classes_valid = false;
  int split_phase = 0;          // 0: no split step open, 1: begun, 2: ghosts done
  // the backward pass of the rows without ghosts runs on a low-priority side stream beside the one of the rows with
  // ghosts (both only wait for the MLP): the ghost forces are still done first, and the two kernels share one tail
  hipStream_t side = nullptr;
  hipEvent_t ev_mlp = nullptr, ev_side = nullptr;
  struct { const double* d_x; int eflag_atom, vflag; double *d_f, *d_ev, *d_eatom; } split{};
  int nlocal = 0, ntotal = 0, nrows = 0;
  long long npairs = 0;
  int count[kMaxSpecies] = {0}, row_start[kMaxSpecies] = {0};

  ani_comm* comm = nullptr;   // ani_attach_comm: ghost forces go home on the device (host-pointer entry points)
  int sticky_flags = 0;   // every bit the device error word has ever shown the host (bit 1: LDS capacity, bit 2: MLP wait timeout)
  int max_numneigh = 0;
  DevBuf<int> species, ilist, numneigh, jlist, jraw, nbr_off, row_of_centre, centre_of_row, bucket_info, err_flag;
  DevBuf<int> row_of_atom;   // [ntotal] AEV row of an atom, -1 for atoms that are no centres (AevArgs::row_of_atom)
  DevBuf<unsigned> sym_acc;  // [ntotal] scratch of the list symmetry check (launch_list_symmetry)
  // one-pass list build (launch_nbr_onepass): rows of jraw_stride entries in jraw (0: dense segments), sized from the longest
  // list of the build before
  int jraw_stride = 0, nbr_cap_hint = 0, nbr_onepass = 1;
  // the build in one kernel (launch_nbr_sorted_rows): jlist itself holds rows of jlist_row_stride entries, already grouped by
  // species; the overflow word of that kernel is read behind rebuild()'s own synchronisation (0: dense segments, as ever)
  int jlist_row_stride = 0, nbr_sorted_rows = 1, nbr_half_cells = 0;
  double* pinned_ev = nullptr;  // page-locked {energy, virial[9], error word} of a host-pointer step (launch_step_tail)
  int* pinned_ints = nullptr;   // page-locked host words for small read-backs that ride on a later synchronisation
  DevBuf<int> nb_ovf;
  DevBuf<float4> nb_xq;
  // ghost fold handed over BEFORE the list it belongs to (ani_stage_ghost_fold): installed by the next list build, its check
  // behind that build's synchronisation
  const int64_t* stage_owner = nullptr;
  const double* stage_shift = nullptr;
  int stage_nghost = -1;
  bool origin_from_box = false;   // ani_build_list*: the caller's bounding box gives the epoch's origin, no kernel
  // ghost fold of the epoch (ani_set_ghost_fold): maps of the caller + the chains of images made from them
  GhostFold fold;
  int fold_nghost = -1;      // -1: none installed (every list build clears it)
  DevBuf<int> fold_head, fold_next, fold_bad;
  bool warned_asymmetric = false;
  bool list_is_ours = false; // the installed list was built by ani_build_list*: symmetric by construction
  bool list_symmetric = true;  // this epoch's list passed the check (or is ours): the symmetric radial collection may run
  DevBuf<int> row_ctr;   // {ticket, waves done} pairs of the fused forward launch, one pair per row range (AevArgs::row_counter)
  DevBuf<float4> xyzs, cl_xyz;
  DevBuf<int4> row_info, cl_hdr;
  DevBuf<int> cl_j;
  int cl_stride = 0;
  DevBuf<double> x64, f64, ev, eatom, virial_acc;
  DevBuf<double> origin;       // [3] the fp32 positions of an epoch are relative to this point (set at its first step)
  bool need_origin = true;
  DevBuf<double> rep_tables, erep;  // optional pairwise repulsion: tables of the model file, energy partial sums
  DevBuf<float> aev, gaev, act, e_rows, fbuf;
  DevBuf<double> aev64, gaev64, act64, e_rows64, fbuf64;  // precision 'double'
  DevBuf<int> nb_cell_id, nb_cell_count, nb_cell_start, nb_cursor, nb_order;  // device-side list build (row f1)
  DevBuf<double> nb_xs;
  std::vector<std::vector<double*>> Hbuf64;
  std::vector<std::vector<float*>> Hbuf, Gbuf;  // [S][k] pointers into act: stored activations H_k; raw (unmasked) dE/dh_k
  bool arena_valid = false;                     // ... laid out for this epoch's row counts (ensure_arena)
  // host staging for the host-pointer entry points
  std::vector<int> h_species32;
  int reuse_upload = 0;               // ani_set_option("reuse_build_list_upload")
  // ani_set_option("out_force_accumulate"): the host entry points ADD the forces into out_force, chunk by chunk through a
  // page-locked buffer of the library, the additions of one chunk beside the copy of the next
  int out_force_accumulate = 0;
  double* stage_force = nullptr;
  size_t stage_force_cap = 0;
  static constexpr int kForceChunks = 16;
  unsigned* stage_flags = nullptr;          // page-locked: chunk c of the step has landed when stage_flags[c] == copy_epoch
  DevBuf<unsigned long long> copy_ctr;      // workgroups through with chunk c, over all steps (launch_copy_out)
  unsigned copy_epoch = 0;
  unsigned long long host_step = 0;         // stamp of the host-pointer steps (launch_step_tail)
  const double* x64_from = nullptr;   // ani_build_list has just uploaded these coordinates into x64: the step that follows reuses them
  std::vector<int> h_half_num, h_half_j;

  // phase timing
  bool timing = false;
  // one set of 6 events per timed step, recorded on the compute stream without synchronising; elapsed times are
  // resolved lazily in ani_phase_times() so the timed region of a benchmark is not perturbed
  std::vector<hipEvent_t> evt_pool;
  size_t evt_used = 0;
  hipEvent_t* evt = nullptr;
  double phase_ms[5] = {0, 0, 0, 0, 0};   // aev_fwd, mlp, aev_bwd, other, nbr_compact
  int phase_calls = 0;
};

namespace {

#define HIP_TRY(h, expr)                                                                              \
  do {                                                                                                \
    hipError_t _e = (expr);                                                                           \
    if (_e != hipSuccess) {                                                                           \
      (h)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                                   \
      return ANI_ERR_DEVICE;                                                                          \
    }                                                                                                 \
  } while (0)

int upload(ani_handle* h, float** dst, const std::vector<float>& src) {
  HIP_TRY(h, hipMalloc((void**)dst, std::max<size_t>(src.size(), 1) * sizeof(float)));
  HIP_TRY(h, hipMemcpy(*dst, src.data(), src.size() * sizeof(float), hipMemcpyHostToDevice));
  return ANI_OK;
}

// precision 'double': the model file stores fp32 weights (exactly representable), so the fp64 copies are made on the
// device from the fp32 uploads
int mirror64(ani_handle* h, double** dst, const float* src, size_t n) {
  if (h->use_single) return ANI_OK;
  if (*dst) (void)hipFree(*dst);
  HIP_TRY(h, hipMalloc((void**)dst, std::max<size_t>(n, 1) * sizeof(double)));
  launch_cvt_f32_f64(src, *dst, n, h->stream);
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return ANI_OK;
}

// device copies of `batch` matrices [N][ld] (first K columns) as blocked 16-bit planes, one per split arithmetic
int make_split(ani_handle* h, unsigned short** dst3, unsigned short** dst2, const float* src, int batch, long long s_src, int N, int K,
               int ld, float wscale) {
  unsigned short** dst[2] = {dst3, dst2};
  for (int i = 0; i < 2; i++) {
    const MlpArith ar = planes_arith(i);
    const size_t elems = split_elems(N, K, ar) * (size_t)batch;
    HIP_TRY(h, hipMalloc((void**)dst[i], std::max<size_t>(elems, 1) * sizeof(unsigned short)));
    launch_split_planes(src, batch, s_src, N, K, ld, ar, wscale, *dst[i], nullptr);
  }
  HIP_TRY(h, hipDeviceSynchronize());
  return ANI_OK;
}

int upload_model(ani_handle* h) {
  const HostModel& m = h->model;
  const int L = m.L, M = m.M;
  const int aev_stride = round_up(m.aev_len, 4);
  h->nets.resize(m.S);
  for (int s = 0; s < m.S; s++) {
    SpeciesNet& n = h->nets[s];
    const std::vector<int>& d = m.dims[s];
    n.w.resize(L);
    n.w[0] = aev_stride;
    for (int k = 1; k < L; k++) n.w[k] = round_up(d[k], 4);
    if (d[L - 1] > 256) { h->err = "last hidden layer wider than 256 is not supported"; return ANI_ERR_MODEL; }
    n.W.assign(L - 1, nullptr); n.b.assign(L - 1, nullptr); n.WT.assign(L - 1, nullptr);
    n.W64.assign(L - 1, nullptr); n.b64.assign(L - 1, nullptr); n.WT64.assign(L - 1, nullptr);
    for (auto& sp : n.sp) { sp.W.assign(L - 1, nullptr); sp.WT.assign(L - 1, nullptr); }
    n.wscale.assign(L - 1, 1.f);
    for (int k = 0; k < L - 1; k++) {
      const int out = d[k + 1], in = d[k], kp = n.w[k];
      float wmax = 0.f;
      for (int a = 0; a < M; a++)
        for (float v : m.W[a][s][k]) wmax = std::max(wmax, std::fabs(v));
      if (!std::isfinite(wmax)) { h->err = "non-finite weight in the model file"; return ANI_ERR_MODEL; }
      int e = 0;
      if (wmax > 0.f) (void)std::frexp(wmax, &e);   // wmax = f * 2^e, f in [0.5, 1)
      n.wscale[k] = std::ldexp(1.f, std::max(-24, std::min(24, 13 - e)));
      std::vector<float> W((size_t)M * out * kp, 0.f), b((size_t)M * out);
      for (int a = 0; a < M; a++) {
        for (int o = 0; o < out; o++) {
          memcpy(&W[((size_t)a * out + o) * kp], &m.W[a][s][k][(size_t)o * in], sizeof(float) * in);
          b[(size_t)a * out + o] = m.b[a][s][k][o];
        }
      }
      int rc = upload(h, &n.W[k], W); if (rc) return rc;
      rc = upload(h, &n.b[k], b); if (rc) return rc;
      rc = mirror64(h, &n.W64[k], n.W[k], W.size()); if (rc) return rc;
      rc = mirror64(h, &n.b64[k], n.b[k], b.size()); if (rc) return rc;
      rc = make_split(h, &n.sp[0].W[k], &n.sp[1].W[k], n.W[k], M, (long long)out * kp, out, kp, kp, n.wscale[k]); if (rc) return rc;
      // transposed copies for the backward products
      if (k == 0) {
        const int w1 = n.w[1];
        std::vector<float> T((size_t)m.aev_len * M * w1, 0.f);   // [aev_len][M*w1]
        for (int a = 0; a < M; a++)
          for (int o = 0; o < out; o++)
            for (int i = 0; i < in; i++) T[(size_t)i * M * w1 + (size_t)a * w1 + o] = m.W[a][s][0][(size_t)o * in + i];
        rc = upload(h, &n.WT[0], T); if (rc) return rc;
        rc = mirror64(h, &n.WT64[0], n.WT[0], T.size()); if (rc) return rc;
        rc = make_split(h, &n.sp[0].WT[0], &n.sp[1].WT[0], n.WT[0], 1, 0, m.aev_len, M * w1, M * w1, n.wscale[0]); if (rc) return rc;
      } else {
        const int wk1 = n.w[k + 1 < L ? k + 1 : k];  // K of the backward product through layer k = padded d[k+1]
        std::vector<float> T((size_t)M * in * wk1, 0.f);         // [M][d[k]][w(k+1)]
        for (int a = 0; a < M; a++)
          for (int o = 0; o < out; o++)
            for (int i = 0; i < in; i++) T[((size_t)a * in + i) * wk1 + o] = m.W[a][s][k][(size_t)o * in + i];
        rc = upload(h, &n.WT[k], T); if (rc) return rc;
        rc = mirror64(h, &n.WT64[k], n.WT[k], T.size()); if (rc) return rc;
        rc = make_split(h, &n.sp[0].WT[k], &n.sp[1].WT[k], n.WT[k], M, (long long)in * wk1, in, wk1, wk1, n.wscale[k]); if (rc) return rc;
      }
    }
    {  // output layer
      const int in = d[L - 1], kp = n.w[L - 1];
      std::vector<float> w((size_t)M * kp, 0.f), b(M);
      for (int a = 0; a < M; a++) {
        memcpy(&w[(size_t)a * kp], m.W[a][s][L - 1].data(), sizeof(float) * in);
        b[a] = m.b[a][s][L - 1][0];
      }
      int rc = upload(h, &n.w_out, w); if (rc) return rc;
      rc = upload(h, &n.b_out, b); if (rc) return rc;
      rc = mirror64(h, &n.w_out64, n.w_out, w.size()); if (rc) return rc;
      rc = mirror64(h, &n.b_out64, n.b_out, b.size()); if (rc) return rc;
    }
  }
  if (m.has_rep) {
    HIP_TRY(h, h->rep_tables.reserve(m.rep_tables.size()));
    HIP_TRY(h, hipMemcpy(h->rep_tables.p, m.rep_tables.data(), sizeof(double) * m.rep_tables.size(), hipMemcpyHostToDevice));
  }
  AevParams& p = h->ap;
  memset(&p, 0, sizeof(p));
  p.S = m.S; p.nR = m.nR; p.nA = m.nA; p.nZ = m.nZ; p.nAZ = m.nA * m.nZ;
  p.radial_len = m.radial_len; p.aev_len = m.aev_len; p.aev_stride = aev_stride;
  p.compat = h->use_cuaev ? 0 : 1;
  p.full_cap = 0;
  p.Rcr = (float)m.Rcr; p.Rca = (float)m.Rca; p.EtaR = (float)m.EtaR; p.EtaA = (float)m.EtaA; p.Zeta = (float)m.Zeta;
  p.pi_over_Rcr = (float)(M_PI / m.Rcr); p.pi_over_Rca = (float)(M_PI / m.Rca);
  for (int k = 0; k < m.nR; k++) p.ShfR[k] = (float)m.ShfR[k];
  for (int k = 0; k < m.nA; k++) p.ShfA[k] = (float)m.ShfA[k];
  for (int k = 0; k < m.nZ; k++) { p.cosZ[k] = (float)cos(m.ShfZ[k]); p.sinZ[k] = (float)sin(m.ShfZ[k]); }
  {
    const double dR = m.nR > 1 ? (m.ShfR[m.nR - 1] - m.ShfR[0]) / (m.nR - 1) : 0.0;
    const double dA = m.nA > 1 ? (m.ShfA[m.nA - 1] - m.ShfA[0]) / (m.nA - 1) : 0.0;
    bool equi = true;
    for (int k = 0; k < m.nR; k++) equi = equi && std::fabs(m.ShfR[0] + k * dR - m.ShfR[k]) < 1e-9;
    for (int k = 0; k < m.nA; k++) equi = equi && std::fabs(m.ShfA[0] + k * dA - m.ShfA[k]) < 1e-9;
    p.ShfR0 = (float)m.ShfR[0]; p.dShfR = (float)dR; p.ShfA0 = (float)m.ShfA[0]; p.dShfA = (float)dA;
    p.equi = equi ? 1 : 0;
  }
  if (aev_stride > 1024) { h->err = "AEV longer than 1024 is not supported"; return ANI_ERR_MODEL; }
  return ANI_OK;
}


// AEV columns of species that do not occur in the system (centres or neighbours) are identically zero, so the
// first-layer products only need the columns of the species present: exact same sums with the zero terms left out.
// The AEV kernels then run with a compact layout (S' species, A' = 16 S' + 32 S'(S'+1)/2 columns) and the first layer /
// dE/dAEV products use column-gathered weights.  Water with ANI-2x: 1008 -> 128 columns; C,H,N,O systems: 384.
int specialize(ani_handle* h, int mask) {
  const HostModel& m = h->model;
  if (!h->prune) mask = (1 << m.S) - 1;
  if (mask == h->active_mask) return ANI_OK;
  for (auto& n : h->nets) {
    if (n.W0c) (void)hipFree(n.W0c);
    if (n.WT0c) (void)hipFree(n.WT0c);
    if (n.W0c64) (void)hipFree(n.W0c64);
    if (n.WT0c64) (void)hipFree(n.WT0c64);
    for (auto& sp : n.sp) {
      if (sp.W0c) (void)hipFree(sp.W0c);
      if (sp.WT0c) (void)hipFree(sp.WT0c);
      sp.W0c = sp.WT0c = nullptr;
    }
    n.W0c = n.WT0c = nullptr;
    n.W0c64 = n.WT0c64 = nullptr;
  }
  std::vector<int> act;
  for (int s = 0; s < m.S; s++) {
    h->cmap.m[s] = (int)act.size();
    if (mask & (1 << s)) act.push_back(s);
  }
  for (int s = m.S; s < kMaxSpecies; s++) h->cmap.m[s] = 0;
  const int na = (int)act.size();
  h->ap_run = h->ap;
  h->colmap.resize(m.aev_len);
  for (int c = 0; c < m.aev_len; c++) h->colmap[c] = c;
  h->active_mask = mask;
  if (na == m.S || na == 0) {
    for (int s = 0; s < m.S; s++) h->cmap.m[s] = s;
    return ANI_OK;
  }
  const int nAZ = m.nA * m.nZ;
  const int alen = na * m.nR + na * (na + 1) / 2 * nAZ;
  const int astride = round_up(alen, 4);
  AevParams& p = h->ap_run;
  p.S = na; p.radial_len = na * m.nR; p.aev_len = alen; p.aev_stride = astride;
  h->colmap.assign(alen, 0);
  for (int cs = 0; cs < na; cs++)
    for (int k = 0; k < m.nR; k++) h->colmap[cs * m.nR + k] = act[cs] * m.nR + k;
  {
    int b = 0;
    for (int c1 = 0; c1 < na; c1++)
      for (int c2 = c1; c2 < na; c2++, b++) {
        const int lo = act[c1], hi = act[c2];
        const int dense_b = lo * m.S - lo * (lo - 1) / 2 + (hi - lo);
        for (int t = 0; t < nAZ; t++) h->colmap[na * m.nR + b * nAZ + t] = m.radial_len + dense_b * nAZ + t;
      }
  }
  const int M = m.M;
  for (int s = 0; s < m.S; s++) {
    SpeciesNet& n = h->nets[s];
    const int out = m.dims[s][1], in = m.dims[s][0], w1 = n.w[1];
    std::vector<float> W((size_t)M * out * astride, 0.f), T((size_t)alen * M * w1, 0.f);
    for (int a = 0; a < M; a++)
      for (int o = 0; o < out; o++)
        for (int c = 0; c < alen; c++) {
          const float v = m.W[a][s][0][(size_t)o * in + h->colmap[c]];
          W[((size_t)a * out + o) * astride + c] = v;
          T[(size_t)c * M * w1 + (size_t)a * w1 + o] = v;
        }
    int rc = upload(h, &n.W0c, W); if (rc) return rc;
    rc = upload(h, &n.WT0c, T); if (rc) return rc;
    rc = mirror64(h, &n.W0c64, n.W0c, W.size()); if (rc) return rc;
    rc = mirror64(h, &n.WT0c64, n.WT0c, T.size()); if (rc) return rc;
    rc = make_split(h, &n.sp[0].W0c, &n.sp[1].W0c, n.W0c, M, (long long)out * astride, out, astride, astride, n.wscale[0]); if (rc) return rc;
    rc = make_split(h, &n.sp[0].WT0c, &n.sp[1].WT0c, n.WT0c, 1, 0, alen, M * w1, M * w1, n.wscale[0]); if (rc) return rc;
  }
  return ANI_OK;
}

// (re)build everything that depends on the neighbour list: offsets, species buckets, activation arena.
// d_species/d_ilist/d_numneigh/d_jlist already hold this epoch's list in the handle's own buffers.
constexpr double kFusedHalfCost = 0.75;   // cost of a half item of the sixteen-row fused MLP relative to its whole item (measured 0.64 - 0.76)
constexpr int kCopyOutBlocks = 32;   // workgroups of launch_copy_out: enough stores in flight for the link, few enough to finish chunks in order
constexpr int kRebuildRowsOverflow = -1000;   // internal: the rows of launch_nbr_sorted_rows were too short
int rebuild(ani_handle* h, hipStream_t st) {
  const HostModel& m = h->model;
  roctxMarkA("neighbor list rebuilt");   // src/ani_csrc/ani.cpp:128,215
  TraceRange tr("ani: list epoch set-up (offsets, species buckets, segment sort)");
  h->need_origin = !h->origin_from_box;
  h->fold_nghost = -1;       // the maps belonged to the list before
  h->classes_valid = false;
  h->split_phase = 0;
  const int nlocal = h->nlocal;
  const int nrows_cap = round_up(nlocal, kRowTile) + m.S * kRowTile;
  HIP_TRY(h, h->nbr_off.reserve((size_t)nlocal + 1));
  HIP_TRY(h, h->row_of_centre.reserve(prepare_scratch_ints(nlocal)));
  HIP_TRY(h, h->centre_of_row.reserve(nrows_cap));
  HIP_TRY(h, h->row_info.reserve(nrows_cap));
  HIP_TRY(h, h->bucket_info.reserve(kBucketInfoInts));
  // a capacity overflow of an earlier epoch must not poison this one (the flag turns the device path's energy into NaN)
  HIP_TRY(h, h->err_flag.reserve(1, true));
  // ... but what it said is kept: the device path only turns the energy into NaN, and a caller that looks at the energy
  // every few steps must still be able to tell a capacity overflow from a stalled kernel (ani_debug_get: error_flags).  The
  // word travels to a page-locked host word without a synchronisation of its own: it is read behind the one the bucket
  // counts below need anyway.
  if (!h->pinned_ints) HIP_TRY(h, hipHostMalloc((void**)&h->pinned_ints, sizeof(int) * 32, hipHostMallocDefault));
  h->pinned_ints[0] = h->pinned_ints[1] = h->pinned_ints[2] = 0;
  HIP_TRY(h, hipMemcpyAsync(&h->pinned_ints[0], h->err_flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(h, hipMemsetAsync(h->err_flag.p, 0, sizeof(int), st));
  HIP_TRY(h, h->row_of_atom.reserve((size_t)std::max(h->ntotal, 1)));
  PrepOut o{h->nbr_off.p, h->row_of_centre.p, h->centre_of_row.p, h->row_info.p, h->bucket_info.p, h->row_of_atom.p};
  launch_prepare(h->species.p, h->ilist.p, h->numneigh.p, nlocal, h->ntotal, m.S, nrows_cap, o, st, h->jlist_row_stride);
  if (h->jlist_row_stride) HIP_TRY(h, hipMemcpyAsync(&h->pinned_ints[1], h->nb_ovf.p, sizeof(int), hipMemcpyDeviceToHost, st));
  // a staged ghost fold: its chains are made here, its check rides on the synchronisation below
  const bool fold_staged = h->stage_nghost >= 0 && h->stage_owner && h->use_single && h->stage_nghost == h->ntotal - nlocal;
  if (fold_staged) {
    HIP_TRY(h, h->fold_head.reserve((size_t)std::max(nlocal, 1)));
    HIP_TRY(h, h->fold_next.reserve((size_t)std::max(h->stage_nghost, 1)));
    HIP_TRY(h, h->fold_bad.reserve(1));
    launch_ghost_chain(reinterpret_cast<const long long*>(h->stage_owner), h->stage_nghost, nlocal, h->fold_head.p, h->fold_next.p, h->fold_bad.p, st);
    HIP_TRY(h, hipMemcpyAsync(&h->pinned_ints[2], h->fold_bad.p, sizeof(int), hipMemcpyDeviceToHost, st));
  }
  int info[kBucketInfoInts];
  HIP_TRY(h, hipMemcpyAsync(info, h->bucket_info.p, sizeof(info), hipMemcpyDeviceToHost, st));
  HIP_TRY(h, hipStreamSynchronize(st));
  h->sticky_flags |= h->pinned_ints[0];   // the error word of the epoch before (copied above)
  HIP_TRY(h, take_launch_error());
  if (h->jlist_row_stride && h->pinned_ints[1]) return kRebuildRowsOverflow;   // build_list fills dense segments instead
  if (h->jlist_row_stride) h->npairs = info[2 * kMaxSpecies + 4];
  if (fold_staged) {
    if (h->pinned_ints[2]) { h->stage_nghost = -1; h->err = "ani_stage_ghost_fold: an owner index lies outside [0, nlocal)"; return ANI_ERR_ARG; }
    h->fold.owner = reinterpret_cast<const long long*>(h->stage_owner); h->fold.shift = h->stage_shift; h->fold.nlocal = nlocal;
    h->fold_nghost = h->stage_nghost;
  }
  h->stage_nghost = -1;
  if (info[2 * kMaxSpecies + 1]) { h->err = "an atom has a species outside the model's species list (or ilist holds an index outside [0, ntotal))"; return ANI_ERR_ARG; }
  for (int s = 0; s < m.S; s++) { h->count[s] = info[s]; h->row_start[s] = info[kMaxSpecies + s]; }
  h->nrows = info[2 * kMaxSpecies];
  h->max_numneigh = info[2 * kMaxSpecies + 2];
  {
    const int rcs = specialize(h, info[2 * kMaxSpecies + 3]);
    if (rcs) return rcs;
  }
  // A caller's list is checked for symmetry once per epoch (the symmetric radial collection relies on it); one that fails
  // runs with the scatter of every term, which is right for any list.  Said once.
  h->list_symmetric = true;
  if (!h->list_is_ours && h->use_single && h->aev_sym_radial && nlocal > 0) {
    HIP_TRY(h, h->sym_acc.reserve((size_t)std::max(h->ntotal, 1)));
    launch_list_symmetry(h->ilist.p, h->nbr_off.p, h->numneigh.p, h->jraw.p, h->row_of_atom.p, nlocal, h->ntotal, h->sym_acc.p,
                         h->bucket_info.p, st);   // bucket_info has been read: its first word carries the verdict
    int bad = 0;
    HIP_TRY(h, hipMemcpyAsync(&bad, h->bucket_info.p, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    HIP_TRY(h, take_launch_error());
    if (bad) {
      h->list_symmetric = false;
      h->sticky_flags |= 8;
      if (!h->warned_asymmetric) {
        h->warned_asymmetric = true;
        fprintf(stderr, "libani_hip: the neighbour list is not symmetric between owned atoms (j in i's list without i in j's): "
                        "aev_symmetric_radial is off for such epochs, every radial term is scattered\n");
      }
    }
  }
  // neighbour segments grouped by species: what lets the AEV kernels accumulate without atomics (launch_nbr_sorted_rows
  // leaves its rows that way)
  if (!h->jlist_row_stride) launch_sort_jlist(h->species.p, h->nbr_off.p, h->numneigh.p, h->jraw.p, h->jlist.p, nlocal, m.S, info[2 * kMaxSpecies + 3], st, h->jraw_stride);

  const size_t stride = h->ap_run.aev_stride;
  if (!h->use_single) {
    HIP_TRY(h, h->aev64.reserve((size_t)std::max(h->nrows, 1) * stride));
    HIP_TRY(h, h->gaev64.reserve((size_t)std::max(h->nrows, 1) * stride));
    HIP_TRY(h, hipMemsetAsync(h->aev64.p, 0, (size_t)h->nrows * stride * sizeof(double), st));
    HIP_TRY(h, h->e_rows64.reserve((size_t)m.M * std::max(h->nrows, 1)));
    size_t need64 = 0;
    for (int s = 0; s < m.S; s++)
      for (int k = 1; k < m.L; k++) need64 += (size_t)round_up(h->count[s], kRowTile) * m.M * h->nets[s].w[k];
    HIP_TRY(h, h->act64.reserve(std::max<size_t>(need64, 1)));
    HIP_TRY(h, hipMemsetAsync(h->act64.p, 0, need64 * sizeof(double), st));
    h->Hbuf64.assign(m.S, std::vector<double*>(m.L, nullptr));
    size_t off64 = 0;
    for (int s = 0; s < m.S; s++)
      for (int k = 1; k < m.L; k++) {
        h->Hbuf64[s][k] = h->act64.p + off64;
        off64 += (size_t)round_up(h->count[s], kRowTile) * m.M * h->nets[s].w[k];
      }
    return ANI_OK;
  }
  HIP_TRY(h, h->aev.reserve((size_t)std::max(h->nrows, 1) * stride));
  HIP_TRY(h, h->gaev.reserve((size_t)std::max(h->nrows, 1) * stride));
  HIP_TRY(h, hipMemsetAsync(h->aev.p, 0, (size_t)h->nrows * stride * sizeof(float), st));  // padding rows stay zero
  HIP_TRY(h, h->e_rows.reserve((size_t)m.M * std::max(h->nrows, 1)));
  // per-step compact neighbour lists of the fast AEV path
  h->cl_stride = aev_compact_stride(h->ap_run, h->max_numneigh);
  if (h->cl_stride > 0) {
    HIP_TRY(h, h->cl_hdr.reserve((size_t)2 * std::max(h->nrows, 1)));
    HIP_TRY(h, h->cl_xyz.reserve((size_t)std::max(h->nrows, 1) * h->cl_stride));
    HIP_TRY(h, h->cl_j.reserve((size_t)std::max(h->nrows, 1) * h->cl_stride));
  }

  h->arena_valid = false;   // the per-layer kernels' activation arena is laid out when they first run in this epoch
  return ANI_OK;
}

// activation arena of the per-layer MLP kernels: per species, H_k (k = 1..L-1; H_{L-1} is overwritten by dE/dz_{L-1}) and
// the raw gradients dE/dh_k (k = 1..L-2), whose celu' factor is applied by the product that consumes them.  The fused
// kernel keeps all of this in registers: an epoch that only ever runs it never allocates or clears the arena.
int ensure_arena(ani_handle* h, hipStream_t st) {
  if (h->arena_valid) return ANI_OK;
  const HostModel& m = h->model;
  size_t need = 0;
  for (int s = 0; s < m.S; s++) {
    const size_t rows = round_up(h->count[s], kRowTile);
    for (int k = 1; k < m.L; k++) need += rows * (size_t)m.M * h->nets[s].w[k];
    for (int k = 1; k < m.L - 1; k++) need += rows * (size_t)m.M * h->nets[s].w[k];
  }
  HIP_TRY(h, h->act.reserve(std::max<size_t>(need, 1)));
  HIP_TRY(h, hipMemsetAsync(h->act.p, 0, need * sizeof(float), st));  // zero K-padding columns
  h->Hbuf.assign(m.S, std::vector<float*>(m.L, nullptr));
  h->Gbuf.assign(m.S, std::vector<float*>(m.L, nullptr));
  size_t off = 0;
  for (int s = 0; s < m.S; s++) {
    const size_t rows = round_up(h->count[s], kRowTile);
    for (int k = 1; k < m.L; k++) {
      h->Hbuf[s][k] = h->act.p + off;
      off += rows * (size_t)m.M * h->nets[s].w[k];
    }
    for (int k = 1; k < m.L - 1; k++) {
      h->Gbuf[s][k] = h->act.p + off;
      off += rows * (size_t)m.M * h->nets[s].w[k];
    }
  }
  h->arena_valid = true;
  return ANI_OK;
}

// ---- fused MLP: eligibility, stream construction, launch ---------------------------------------------------------
bool fused_eligible(const ani_handle* h) {
  const HostModel& m = h->model;
  if (!h->mlp_fused || h->mlp_arith == MLP_FP32 || m.L != 4) return false;
  // Small systems: a fused tile takes ~0.1 ms whatever else happens, so with fewer tiles than CUs the kernel costs that much
  // however few rows there are, while the chained per-layer launch of small systems scales down with them (MLP, exact
  // arithmetic: 12 501 atoms 0.09 fused against 0.08 chained; 25 002 atoms 0.09 against 0.12): fused from ~16 000 atoms on.
  if (h->mlp_fused < 2 && m.M == 1 && !h->mlp_fused_gen) {   // (several members: the fused kernel's (tile, member) work items win at every size measured)
    int tiles = 0;
    for (int s = 0; s < m.S; s++) tiles += round_up(h->count[s], kRowTile) / kRowTile;
    if (tiles < 125) return false;
  }
  const int acols = h->ap_run.aev_len;
  if (acols < 16 || (acols & 15) || (h->ap_run.aev_stride & 3)) return false;
  for (int s = 0; s < m.S; s++)
    if (fused_shape_for(m.dims[s][1], m.dims[s][2], m.dims[s][3]) < 0 || m.dims[s][4] != 1) return false;
  return true;
}

void free_fused(ani_handle* h, int fi) {
  for (auto& n : h->nets) {
    if (n.fu[fi].stream) (void)hipFree(n.fu[fi].stream);
    if (n.fu[fi].consts) (void)hipFree(n.fu[fi].consts);
    n.fu[fi] = SpeciesNet::Fused{};
  }
  h->fused_mask[fi] = -2;
}

// weight streams + constants of every species for arithmetic `arith`, in the AEV layout of this epoch
int ensure_fused(ani_handle* h, MlpArith arith, int gen, hipStream_t st) {
  const int pi = arith == MLP_F16X2 ? 1 : 0, P = mlp_planes(arith), fi = pi + 2 * gen;
  if (h->fused_mask[fi] == h->active_mask) return ANI_OK;
  free_fused(h, fi);
  const HostModel& m = h->model;
  const int M = m.M, acols = h->ap_run.aev_len, ka = h->ap_run.aev_stride;
  const int ks0 = acols / 16, nt0 = (acols + 31) / 32;
  for (int s = 0; s < m.S; s++) {
    SpeciesNet& n = h->nets[s];
    const std::vector<int>& d = m.dims[s];
    const int shape = fused_shape_for(d[1], d[2], d[3]);
    n.fused_shape = shape;
    int nt[3];
    fused_shape_tiles(shape, nt);
    SpeciesNet::Fused& f = n.fu[fi];
    f.ppm = gen ? fused16_pieces_per_member(shape, acols, P) : fused_pieces_per_member(shape, acols, P);
    f.cpm = fused_consts_floats(shape);
    HIP_TRY(h, hipMalloc((void**)&f.stream, (size_t)f.ppm * M * 1024));
    HIP_TRY(h, hipMalloc((void**)&f.consts, sizeof(float) * (size_t)f.cpm * M));
    std::vector<float> cst((size_t)f.cpm * M, 0.f);
    for (int a = 0; a < M; a++) {
      unsigned short* dst = reinterpret_cast<unsigned short*>(f.stream + (size_t)a * f.ppm * 1024);
      const float ws0 = n.wscale[0], ws1 = n.wscale[1], ws2 = n.wscale[2];
      if (!gen) {
        auto emit = [&](const float* src, int ld, int rows, int kval, int NT, int KS, int chunk, float scale) {
          launch_build_stream(src, ld, rows, kval, NT, KS, chunk, P, scale, dst, st);
          dst += (size_t)NT * KS * P * 512;
        };
        emit((n.W0c ? n.W0c : n.W[0]) + (size_t)a * d[1] * ka, ka, d[1], acols, nt[0], ks0, 0, ws0);                 // F1
        emit(n.W[1] + (size_t)a * d[2] * n.w[1], n.w[1], d[2], d[1], nt[1], 2 * nt[0], 0, ws1);                       // F2
        emit(n.W[2] + (size_t)a * d[3] * n.w[2], n.w[2], d[3], d[2], nt[2], 2 * nt[1], 0, ws2);                       // F3
        emit(n.WT[2] + (size_t)a * d[2] * n.w[3], n.w[3], d[2], d[3], nt[1], 2 * nt[2], -1, ws2);                     // B3
        emit(n.WT[1] + (size_t)a * d[1] * n.w[2], n.w[2], d[1], d[2], nt[0], 2 * nt[1], -1, ws1);                     // B2
        emit((n.WT0c ? n.WT0c : n.WT[0]) + (size_t)a * n.w[1], M * n.w[1], acols, d[1], nt0, 2 * nt[0], 4, ws0);      // B1 (chunks of kChunk tiles)
      } else {
        // the 16-row kernel's order (ani_kernels_mlpg.hip): 16-feature output tiles, 32-deep k-steps
        const int n1 = 2 * nt[0], n2 = 2 * nt[1], n3 = 2 * nt[2], ks1 = (acols + 31) / 32, nt16 = (acols + 15) / 16;
        auto emit = [&](const float* src, int ld, int rows, int kval, int NT, int KS, int order, int nt_off, int identity, float scale) {
          launch_build_stream16(src, ld, rows, kval, NT, KS, order, nt_off, identity, P, scale, dst, st);
          dst += (size_t)NT * KS * P * 512;
        };
        emit((n.W0c ? n.W0c : n.W[0]) + (size_t)a * d[1] * ka, ka, d[1], acols, n1, ks1, 0, 0, 1, ws0);               // F1: a slab per k-step
        emit(n.W[1] + (size_t)a * d[2] * n.w[1], n.w[1], d[2], d[1], n2, n1 / 2, 0, 0, 0, ws1);                       // F2
        emit(n.W[2] + (size_t)a * d[3] * n.w[2], n.w[2], d[3], d[2], n3, n2 / 2, 0, 0, 0, ws2);                       // F3
        emit(n.WT[2] + (size_t)a * d[2] * n.w[3], n.w[3], d[2], d[3], n2, n3 / 2, 1, 0, 0, ws2);                      // B3: tile-major
        emit(n.WT[1] + (size_t)a * d[1] * n.w[2], n.w[2], d[1], d[2], n1, n2 / 2, 1, 0, 0, ws1);                      // B2
        for (int ci = 0, c0 = 0; ci < fused16_b1_chunks(nt16); ci++) {                                                // B1: chunks of tiles
          const int ntc = fused16_b1_chunk_tiles(nt16, ci);
          emit((n.WT0c ? n.WT0c : n.WT[0]) + (size_t)a * n.w[1], M * n.w[1], acols, d[1], ntc, n1 / 2, 0, c0, 0, ws0);
          c0 += ntc;
        }
      }
      if ((size_t)(dst - reinterpret_cast<unsigned short*>(f.stream + (size_t)a * f.ppm * 1024)) != (size_t)f.ppm * 512) {
        h->err = "internal: fused MLP stream size mismatch";
        return ANI_ERR_MODEL;
      }
      float* c = cst.data() + (size_t)a * f.cpm;
      int off = 0;
      for (int o = 0; o < d[1]; o++) c[off + o] = m.b[a][s][0][o];
      off += 32 * nt[0];
      for (int o = 0; o < d[2]; o++) c[off + o] = m.b[a][s][1][o];
      off += 32 * nt[1];
      for (int o = 0; o < d[3]; o++) c[off + o] = m.b[a][s][2][o];
      off += 32 * nt[2];
      for (int o = 0; o < d[3]; o++) c[off + o] = m.W[a][s][3][o];
      off += 32 * nt[2];
      c[off] = m.b[a][s][3][0];
      const bool f16 = arith == MLP_F16X2;
      c[off + 1] = f16 ? 1.f / (16.f * ws0) : 1.f;
      c[off + 2] = f16 ? 1.f / (16.f * ws1) : 1.f;
      c[off + 3] = f16 ? 1.f / (16.f * ws2) : 1.f;
      c[off + 4] = f16 ? 1.f / (4096.f * ws2) : 1.f;
      c[off + 5] = f16 ? 1.f / (4096.f * ws1) : 1.f;
      c[off + 6] = f16 ? 1.f / (4096.f * ws0) : 1.f;
    }
    HIP_TRY(h, hipMemcpyAsync(f.consts, cst.data(), sizeof(float) * cst.size(), hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipStreamSynchronize(st));   // cst is a local
  }
  HIP_TRY(h, hipGetLastError());
  h->fused_mask[fi] = h->active_mask;
  return ANI_OK;
}

int compute_mlp_fused(ani_handle* h, hipStream_t st) {
  const HostModel& m = h->model;
  const MlpArith arith = h->mlp_arith;
  const int gen = h->mlp_fused_gen ? 1 : 0, pi = (arith == MLP_F16X2 ? 1 : 0) + 2 * gen;
  int rc = ensure_fused(h, arith, gen, st);
  if (rc) return rc;
  // The 16-row kernel: 128-row tiles on eight waves (two per SIMD) where that fills the CUs, else 64-row tiles on four -- twice
  // the workgroups, about half the latency of a tile (a decomposed box's per-GPU share: 12 501 water atoms are 99 tiles of 128).
  int sub = 1;   // 64-row tiles per 128-row tile
  if (gen) {
    int tiles128 = 0;
    for (int s = 0; s < m.S; s++) tiles128 += round_up(h->count[s], kRowTile) / kRowTile;
    const int items128 = tiles128 * (m.M > 1 && h->mlp_fused != 3 ? m.M : 1);
    // 64-row tiles only where they still fit one round (a tile of either form costs about the same time per row)
    const bool small = h->mlp_fused_rows == 64 || (h->mlp_fused_rows == 0 && 2 * items128 <= fused_num_cus() + fused_num_cus() / 8);
    sub = small ? 2 : 1;
  }
  HIP_TRY(h, h->fused_counter.reserve(1));
  FusedArgs G{};
  G.M = m.M; G.alpha = (float)m.alpha; G.inv_alpha = (float)(1.0 / m.alpha); G.scale = 1.f / (float)m.M;
  G.counter = h->fused_counter.p;
  G.err_flag = h->err_flag.p;
  const int acols = h->ap_run.aev_len, ka = h->ap_run.aev_stride;
  int np = 0, total = 0;
  for (int shape = 0; shape < 3; shape++)        // costliest species first
    for (int s = 0; s < m.S; s++) {
      const SpeciesNet& n = h->nets[s];
      if (h->count[s] == 0 || n.fused_shape != shape) continue;
      if (np == kMaxProblems) { h->err = "too many species buckets for one fused MLP launch"; return ANI_ERR_ARG; }
      FusedProb& p = G.p[np];
      p.aev = h->aev.p + (size_t)h->row_start[s] * ka;
      p.gaev = h->gaev.p + (size_t)h->row_start[s] * ka;
      p.gaev_row0 = h->row_start[s];
      p.e_rows = h->e_rows.p + h->row_start[s];
      p.centre_of_row = h->centre_of_row.p + h->row_start[s];
      p.stream = n.fu[pi].stream; p.consts = n.fu[pi].consts;
      p.sE = h->nrows;
      p.tiles = sub * (round_up(h->count[s], kRowTile) / kRowTile);
      p.shape = shape;
      p.ks0 = acols / 16; p.nt0 = (acols + 31) / 32; p.acols = acols; p.aev_stride = ka;
      p.pieces_per_member = (int)n.fu[pi].ppm; p.consts_per_member = n.fu[pi].cpm;
      G.tile_start[np] = total;
      total += p.tiles;
      np++;
    }
  G.tile_start[np] = total;
  G.nprob = np;
  // Several members: a work item is (tile, member) -- M times the parallelism, and an even finish -- each member writing its
  // own dE/dAEV rows, summed by one small kernel (MLP ms, exact arithmetic, 8 members, against the grouped per-layer launches:
  // 10 002 water atoms 0.250 / 0.453, 25 002: 0.543 / 0.818, 50 001: 0.995 / 1.578, CH4/O2 100 008 (ANI-1x): 1.65 / 2.03; a
  // tile's members one after the other in its workgroup: 0.608 at 10 002 atoms).  The copies cost M times the dE/dAEV array:
  // beyond 8 GB the members run in sequence instead.
  const size_t parts_bytes = (size_t)std::max(h->nrows, 1) * ka * sizeof(float) * m.M;
  G.member_items = (m.M > 1 && parts_bytes <= ((size_t)8 << 30) && h->mlp_fused != 3) ? 1 : 0;
  if (G.member_items) {
    const size_t per = (size_t)std::max(h->nrows, 1) * ka;
    HIP_TRY(h, h->gaev_parts.reserve(per * m.M));
    G.gaev_parts = h->gaev_parts.p;
    G.part_stride = (long long)per;
  }
  // which workgroup runs which items: a static schedule, remade when the tile counts change (re-neighbouring)
  if (h->mlp_fused_sched) {
    const int per_tile = G.member_items ? m.M : 1, bins = fused_num_cus();
    const int nitems = total * per_tile;
    int mix = np;   // the tile counts and shapes of the problems, folded into one word
    for (int q = 0; q < np; q++) mix = mix * 1000003 + G.p[q].tiles * 4 + G.p[q].shape;
    mix = mix * 31 + sub + 2 * gen + 4 * h->mlp_fused_halves;
    if (h->sched_key[0] != total || h->sched_key[1] != per_tile || h->sched_key[2] != mix || h->sched_key[3] != bins || !h->fused_sched.p) {
      // item types: one per problem; a (tile, member) item costs what its tile's member costs.  Item t of the kernel's numbering
      // is tile t / per_tile: the items of a problem are contiguous.
      int cnt[kMaxProblems];
      double cost[kMaxProblems];
      for (int q = 0; q < np; q++) {
        int nt[3];
        fused_shape_tiles(G.p[q].shape, nt);
        cnt[q] = G.p[q].tiles * per_tile;
        cost[q] = (double)G.p[q].ks0 * nt[0] + 4.0 * nt[0] * nt[1] + 4.0 * nt[1] * nt[2] + 2.0 * nt[0] * G.p[q].nt0 +
                  12.0;   // MFMA blocks of a member + a little for what a tile costs whatever its size
        if (!G.member_items) cost[q] *= m.M;
      }
      // The sixteen-row kernel takes half items: the schedule may cut the items of a last, mostly idle round in two.  A half costs
      // about kFusedHalfCost of its item (64 of 128 rows with all the weights streamed: profiles/r04_mlp_halves.log), and the
      // cost model is good to about 5 %: a split the model likes by at least 3 % is TIMED against the whole items on this very
      // step's rows (three launches each, the first not counted; the launch is idempotent) and kept only if it is faster --
      // once per set of tile counts (ANI_FUSED_AUTOTUNE=0: no timing, a split needs 8 % by the model).
      std::vector<int> items((size_t)2 * std::max(nitems, 1)), off(bins + 1);
      int nsched = nitems;
      const int split_mode = !gen ? 0 : (h->mlp_fused_halves == 2 ? 2 : (sub != 2 ? h->mlp_fused_halves : 0));   // searched for 128-row tiles only
      static const double half_cost = [] { const char* e = getenv("ANI_FUSED_HALF_COST"); return e ? atof(e) : kFusedHalfCost; }();
      static const bool autotune = [] { const char* e = getenv("ANI_FUSED_AUTOTUNE"); return !(e && atoi(e) == 0); }();
      auto upload = [&](const std::vector<int>& it, int n) -> int {
        HIP_TRY(h, h->fused_sched.reserve((size_t)2 * std::max(nitems, 1) + bins + 1));
        HIP_TRY(h, hipMemcpyAsync(h->fused_sched.p, it.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, st));
        HIP_TRY(h, hipMemcpyAsync(h->fused_sched.p + n, off.data(), sizeof(int) * (size_t)(bins + 1), hipMemcpyHostToDevice, st));
        HIP_TRY(h, hipStreamSynchronize(st));   // items / off are locals; once per re-neighbouring
        h->sched_nitems = n;
        return ANI_OK;
      };
      auto timed = [&](float* ms) -> int {   // the fused launch with the schedule just uploaded
        G.sched_items = h->fused_sched.p; G.sched_off = h->fused_sched.p + h->sched_nitems; G.sched_blocks = bins;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        HIP_TRY(h, hipEventCreate(&e0));
        HIP_TRY(h, hipEventCreate(&e1));
        hipError_t err = launch_mlp_fused16(G, arith, 8, st);
        if (err == hipSuccess) err = hipEventRecord(e0, st);
        for (int k = 0; k < 2 && err == hipSuccess; k++) err = launch_mlp_fused16(G, arith, 8, st);
        if (err == hipSuccess) err = hipEventRecord(e1, st);
        if (err == hipSuccess) err = hipEventSynchronize(e1);
        if (err == hipSuccess) err = hipEventElapsedTime(ms, e0, e1);
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        HIP_TRY(h, err);
        return ANI_OK;
      };
      std::vector<int> split(np, 0);
      (void)fused_schedule_halves(np, cnt, cost, half_cost, bins, split_mode, split.data(), items.data(), off.data(), &nsched,
                                  autotune && split_mode == 1 ? 0.03 : 0.08);
      int any_split = 0;
      for (int q = 0; q < np; q++) any_split += split[q];
      {
        const int rcu = upload(items, nsched);
        if (rcu) return rcu;
      }
      if (autotune && split_mode == 1 && any_split && !getenv("ANI_FUSED_SPLIT")) {
        float t_split = 0.f, t_whole = 0.f;
        int rct = timed(&t_split);
        if (rct) return rct;
        std::vector<int> items_w((size_t)std::max(nitems, 1)), off_split = off;
        int n_w = nitems;
        (void)fused_schedule_halves(np, cnt, cost, half_cost, bins, 0, nullptr, items_w.data(), off.data(), &n_w);
        rct = upload(items_w, n_w);
        if (!rct) rct = timed(&t_whole);
        if (rct) return rct;
        if (t_split < t_whole) {   // the split stays: back it goes
          off = off_split;
          rct = upload(items, nsched);
          if (rct) return rct;
        }
        if (getenv("ANI_FUSED_AUTOTUNE_VERBOSE"))
          fprintf(stderr, "libani_hip: fused MLP schedule, %d items: whole %.4f ms, with half items %.4f ms per launch -> %s\n", nitems,
                  t_whole / 2, t_split / 2, t_split < t_whole ? "halves" : "whole");
      }
      h->sched_key[0] = total; h->sched_key[1] = per_tile; h->sched_key[2] = mix; h->sched_key[3] = bins;
    }
    G.sched_items = h->fused_sched.p;
    G.sched_off = h->fused_sched.p + h->sched_nitems;
    G.sched_blocks = bins;
  }
  if (gen) {
    HIP_TRY(h, launch_mlp_fused16(G, arith, sub == 2 ? 4 : 8, st));
    h->last_mlp_kernel = arith == MLP_F16X2 ? (sub == 2 ? "mlp_fused16<2, 4>" : "mlp_fused16<2, 8>") : (sub == 2 ? "mlp_fused16<3, 4>" : "mlp_fused16<3, 8>");
  } else {
    HIP_TRY(h, launch_mlp_fused(G, arith, st));
    h->last_mlp_kernel = arith == MLP_F16X2 ? "mlp_fused<2>" : "mlp_fused<3>";
  }
  if (G.member_items) launch_sum_parts(h->gaev_parts.p, G.part_stride, m.M, h->gaev.p, (long long)h->nrows * ka, st);
  return ANI_OK;
}

// The MLP ensemble for every species bucket, one grouped launch per layer (all species and members together).
// Buffers of species s are local to the bucket (row 0 = row_start[s]).
//   W[k]  : [M][d[k+1]][w[k]]   Bt of forward layer k (K = w[k], zero padded)
//   WT[k] : [M][d[k]][w[k+1]]   Bt of the backward product through layer k (k >= 1);  WT[0]: [aev_len][M*w[1]]
//   H_k   : [rows][M*w[k]]      activations after layer k-1 (member a at column a*w[k]); overwritten by G_k = dE/dz_k
int compute_mlp(ani_handle* h, hipStream_t st) {
  if (fused_eligible(h)) return compute_mlp_fused(h, st);
  {
    const int rca = ensure_arena(h, st);
    if (rca) return rca;
  }
  const HostModel& m = h->model;
  const int L = m.L, M = m.M;
  const float alpha = (float)m.alpha, inv_alpha = (float)(1.0 / m.alpha);
  std::vector<GemmArgs> probs;
  probs.reserve(m.S);
  // every layer's problems and epilogue, in launch order: either six grouped launches or one chained launch
  std::vector<std::vector<GemmArgs>> layer_probs;
  std::vector<int> layer_epi;
  const MlpArith arith = h->mlp_arith;
  const int spi = arith == MLP_F16X2 ? 1 : 0;
  // two-term fp16 path: power-of-two scales of the A operands (ani_kernels_mlp.hip): activations 2^4, gradients 2^12
  const float a_fwd = arith == MLP_F16X2 ? 16.f : 1.f, a_bwd = arith == MLP_F16X2 ? 4096.f : 1.f;
  auto set_planes = [&](GemmArgs& g, const unsigned short* planes, long long per_batch_matrices, float a_scale, float wscale) {
    g.Btp = planes; g.kbp = (g.K + 15) / 16;
    g.sBp = per_batch_matrices ? (long long)split_elems(g.N, g.K, arith) : 0;
    g.a_scale = a_scale;
    g.inv_scale = arith == MLP_F16X2 ? 1.f / (a_scale * wscale) : 1.f;
  };
  auto base_args = [&](int s) {
    GemmArgs g{};
    g.rows = round_up(h->count[s], kRowTile);
    g.row0 = 0;
    g.batch = M;
    g.alpha = alpha;
    g.inv_alpha = inv_alpha;
    g.scale = 1.f / (float)M;
    g.centre_of_row = h->centre_of_row.p + h->row_start[s];
    return g;
  };
  // forward
  for (int k = 0; k <= L - 2; k++) {
    probs.clear();
    for (int s = 0; s < m.S; s++) {
      if (h->count[s] == 0) continue;
      const SpeciesNet& n = h->nets[s];
      const std::vector<int>& d = m.dims[s];
      GemmArgs g = base_args(s);
      if (k == 0) {
        const int ka = h->ap_run.aev_stride;  // first layer over the AEV columns of the species present
        g.A = h->aev.p + (size_t)h->row_start[s] * ka; g.lda = ka; g.sA = 0;
        g.K = ka; g.Bt = n.W0c ? n.W0c : n.W[0]; g.ldb = ka; g.sB = (long long)d[1] * ka;
      } else {
        g.A = h->Hbuf[s][k]; g.lda = M * n.w[k]; g.sA = n.w[k];
        g.K = n.w[k]; g.Bt = n.W[k]; g.ldb = n.w[k]; g.sB = (long long)d[k + 1] * n.w[k];
      }
      g.N = d[k + 1];
      set_planes(g, (k == 0 && n.W0c) ? n.sp[spi].W0c : n.sp[spi].W[k], 1, a_fwd, n.wscale[k]);
      g.bias = n.b[k]; g.sBias = d[k + 1];
      g.C = h->Hbuf[s][k + 1]; g.ldc = M * n.w[k + 1]; g.sC = n.w[k + 1];
      if (k == L - 2) {
        g.aux = n.w_out; g.sAux = n.w[L - 1];
        g.bias_last = n.b_out;
        g.e_out = h->e_rows.p + h->row_start[s]; g.sE = h->nrows;
      }
      probs.push_back(g);
    }
    layer_probs.push_back(probs); layer_epi.push_back(k == L - 2 ? EPI_LAST : EPI_CELU);
  }
  // backward: dE/dh_{k-1} = G_k W[k-1] with G_k = dE/dh_k * celu'(z_k).  The celu' factor of a layer is applied when
  // its raw gradient is staged as the A operand of the next product (Amask = the stored activation H_k), not in the
  // epilogue of the product that made it: the H loads then ride the prefetch pipeline instead of stalling the epilogue.
  // G_{L-1} comes fully formed out of the fused last forward layer.
  for (int k = L - 1; k >= 2; k--) {
    probs.clear();
    for (int s = 0; s < m.S; s++) {
      if (h->count[s] == 0) continue;
      const SpeciesNet& n = h->nets[s];
      const std::vector<int>& d = m.dims[s];
      GemmArgs g = base_args(s);
      g.A = (k == L - 1) ? h->Hbuf[s][k] : h->Gbuf[s][k];
      g.Amask = (k == L - 1) ? nullptr : h->Hbuf[s][k];
      g.lda = M * n.w[k]; g.sA = n.w[k]; g.K = n.w[k];
      g.Bt = n.WT[k - 1]; g.ldb = n.w[k]; g.sB = (long long)d[k - 1] * n.w[k];
      g.N = d[k - 1];
      set_planes(g, n.sp[spi].WT[k - 1], 1, a_bwd, n.wscale[k - 1]);
      g.C = h->Gbuf[s][k - 1]; g.ldc = M * n.w[k - 1]; g.sC = n.w[k - 1];
      probs.push_back(g);
    }
    layer_probs.push_back(probs); layer_epi.push_back(EPI_PLAIN);
  }
  // dE/dAEV = sum over members of G_1 W[0]  (members concatenated along K)
  probs.clear();
  for (int s = 0; s < m.S; s++) {
    if (h->count[s] == 0) continue;
    const SpeciesNet& n = h->nets[s];
    GemmArgs g = base_args(s);
    g.batch = 1;
    g.A = h->Gbuf[s][1]; g.Amask = h->Hbuf[s][1]; g.lda = M * n.w[1]; g.K = M * n.w[1];
    g.Bt = n.WT0c ? n.WT0c : n.WT[0]; g.ldb = M * n.w[1];
    g.N = h->ap_run.aev_len;
    set_planes(g, n.WT0c ? n.sp[spi].WT0c : n.sp[spi].WT[0], 0, a_bwd, n.wscale[0]);
    g.C = h->gaev.p + (size_t)h->row_start[s] * h->ap_run.aev_stride; g.ldc = h->ap_run.aev_stride;
    probs.push_back(g);
  }
  layer_probs.push_back(probs); layer_epi.push_back(EPI_PLAIN);

  // One ensemble member, split arithmetic -- three ways to launch the six products:
  //  * up to one round of chained workgroups (2 per CU; <= 512 64-row tiles = 32 000 water atoms): ONE chained launch, a
  //    workgroup takes its tile through all layers (mlp_chain_x3);
  //  * beyond (two-term arithmetic, layers no wider than 256): ONE launch of persistent workgroups over (layer, tile) items
  //    with a completion flag per item (mlp_pipeline_x2) -- measured against the alternatives, MLP ms: 50 001 atoms 0.149
  //    (chained 0.203), 75 000 0.206 (grouped 0.240), 100 002 0.281 (grouped 0.320); 25 002 atoms 0.113 against 0.095
  //    chained;
  //  * otherwise one grouped launch per layer (several members, the exact bf16 split with many tiles, wide layers).
  const int np = (int)layer_probs[0].size();
  int tiles = 0;
  for (const GemmArgs& g : layer_probs[0]) tiles += g.rows / 64;
  const int cslots = mlp_chain_slots(), lslots = cslots + cslots / 2;
  bool pipeline = h->mlp_pipeline && arith == MLP_F16X2 && M == 1 && np > 0 && (tiles > cslots || h->mlp_pipeline > 1) &&
                  h->mlp_chain <= 1;
  for (const auto& lp : layer_probs) {
    pipeline = pipeline && (int)lp.size() == np;
    for (const GemmArgs& g : lp)
      pipeline = pipeline && g.N <= 256 && g.batch == 1 && (g.ldc & 31) == 0 && (reinterpret_cast<uintptr_t>(g.C) & 127) == 0;
  }
  // without the pipeline the choice between chain and grouped launches is a matter of rounds (0.106 ms per chained round,
  // 0.137 ms per round of six grouped launches at the benchmark shapes)
  const bool chain_wins = tiles <= cslots || (tiles <= 2 * cslots &&
                                              0.106 * ((tiles + cslots - 1) / cslots) < 0.137 * ((tiles + lslots - 1) / lslots));
  bool chain = !pipeline && h->mlp_chain && arith != MLP_FP32 && M == 1 && np > 0 && (h->mlp_chain > 1 || chain_wins);
  for (const auto& lp : layer_probs) chain = chain && (int)lp.size() == np;
  if (chain || pipeline) {
    std::vector<GemmArgs> flat;
    for (const auto& lp : layer_probs) flat.insert(flat.end(), lp.begin(), lp.end());
    HIP_TRY(h, launch_mlp_chain(flat.data(), layer_epi.data(), (int)layer_probs.size(), np, &h->chain_plan, st, arith, pipeline,
                                h->err_flag.p));
    h->last_mlp_kernel = pipeline ? "mlp_pipeline" : "mlp_chain";
  } else {
    h->last_mlp_kernel = "gemm_grouped (one launch per layer)";
    for (size_t l = 0; l < layer_probs.size(); l++)
      launch_gemm_group(layer_probs[l].data(), (int)layer_probs[l].size(), (Epilogue)layer_epi[l], st, arith);
  }
  return ANI_OK;
}


// precision 'double': same pipeline on the fp64 kernels (ani_kernels_f64.hip)
int run_step64(ani_handle* h, const double* d_x, int eflag_atom, int vflag, double* d_f, int f_accumulate, double* d_ev,
               double* d_eatom, hipStream_t st) {
  const HostModel& m = h->model;
  const int L = m.L, M = m.M;
  HIP_TRY(h, h->fbuf64.reserve((size_t)std::max(h->ntotal, 1) * 3));
  HIP_TRY(h, h->virial_acc.reserve(9 * kVirialSlots));
  HIP_TRY(h, h->err_flag.reserve(1, true));
  HIP_TRY(h, hipMemsetAsync(h->fbuf64.p, 0, sizeof(double) * 3 * (size_t)h->ntotal, st));
  HIP_TRY(h, hipMemsetAsync(h->virial_acc.p, 0, sizeof(double) * 9, st));
  HIP_TRY(h, hipMemsetAsync(h->e_rows64.p, 0, sizeof(double) * (size_t)M * std::max(h->nrows, 1), st));
  Aev64Params p{};
  const AevParams& r = h->ap_run;
  p.S = r.S; p.nR = r.nR; p.nA = r.nA; p.nZ = r.nZ; p.radial_len = r.radial_len; p.aev_len = r.aev_len; p.aev_stride = r.aev_stride;
  p.compat = r.compat;
  p.Rcr = m.Rcr; p.Rca = m.Rca; p.EtaR = m.EtaR; p.EtaA = m.EtaA; p.Zeta = m.Zeta;
  for (int k = 0; k < m.nR; k++) p.ShfR[k] = m.ShfR[k];
  for (int k = 0; k < m.nA; k++) p.ShfA[k] = m.ShfA[k];
  for (int k = 0; k < m.nZ; k++) { p.cosZ[k] = cos(m.ShfZ[k]); p.sinZ[k] = sin(m.ShfZ[k]); }
  Aev64Args a{};
  a.x = d_x; a.species = h->species.p; a.cmap = h->cmap; a.jlist = h->jlist.p; a.row_info = h->row_info.p; a.nrows = h->nrows;
  a.aev = h->aev64.p; a.gaev = h->gaev64.p; a.fbuf = h->fbuf64.p; a.virial = vflag ? h->virial_acc.p : nullptr;
  a.err_flag = h->err_flag.p;
  launch_aev64_forward(p, a, st);
  const int ka = r.aev_stride;
  for (int s = 0; s < m.S; s++) {
    if (h->count[s] == 0) continue;
    const SpeciesNet& n = h->nets[s];
    const std::vector<int>& d = m.dims[s];
    const int rows = round_up(h->count[s], kRowTile), r0 = h->row_start[s];
    auto base = [&]() {
      Gemm64Args g{};
      g.rows = rows; g.batch = M; g.alpha = m.alpha; g.scale = 1.0 / M; g.centre_of_row = h->centre_of_row.p + r0;
      return g;
    };
    for (int k = 0; k <= L - 2; k++) {
      Gemm64Args g = base();
      if (k == 0) {
        g.A = h->aev64.p + (size_t)r0 * ka; g.lda = ka; g.sA = 0; g.K = ka;
        g.Bt = n.W0c64 ? n.W0c64 : n.W64[0]; g.ldb = ka; g.sB = (long long)d[1] * ka;
      } else {
        g.A = h->Hbuf64[s][k]; g.lda = M * n.w[k]; g.sA = n.w[k]; g.K = n.w[k];
        g.Bt = n.W64[k]; g.ldb = n.w[k]; g.sB = (long long)d[k + 1] * n.w[k];
      }
      g.N = d[k + 1]; g.bias = n.b64[k]; g.sBias = d[k + 1];
      g.C = h->Hbuf64[s][k + 1]; g.ldc = M * n.w[k + 1]; g.sC = n.w[k + 1];
      if (k == L - 2) {
        g.aux = n.w_out64; g.sAux = n.w[L - 1]; g.bias_last = n.b_out64;
        g.e_out = h->e_rows64.p + r0; g.sE = h->nrows;
      }
      launch_gemm64(g, k == L - 2 ? EPI_LAST : EPI_CELU, st);
    }
    for (int k = L - 1; k >= 2; k--) {
      Gemm64Args g = base();
      g.A = h->Hbuf64[s][k]; g.lda = M * n.w[k]; g.sA = n.w[k]; g.K = n.w[k];
      g.Bt = n.WT64[k - 1]; g.ldb = n.w[k]; g.sB = (long long)d[k - 1] * n.w[k];
      g.N = d[k - 1];
      g.aux = h->Hbuf64[s][k - 1]; g.ldaux = M * n.w[k - 1]; g.sAux = n.w[k - 1];
      g.C = h->Hbuf64[s][k - 1]; g.ldc = M * n.w[k - 1]; g.sC = n.w[k - 1];
      launch_gemm64(g, EPI_BWD, st);
    }
    {
      Gemm64Args g = base();
      g.batch = 1;
      g.A = h->Hbuf64[s][1]; g.lda = M * n.w[1]; g.K = M * n.w[1];
      g.Bt = n.WT0c64 ? n.WT0c64 : n.WT64[0]; g.ldb = M * n.w[1];
      g.N = r.aev_len;
      g.C = h->gaev64.p + (size_t)r0 * ka; g.ldc = ka;
      launch_gemm64(g, EPI_PLAIN, st);
    }
  }
  launch_aev64_backward(p, a, st);
  if (m.has_rep) {
    HIP_TRY(h, h->erep.reserve(kVirialSlots));
    HIP_TRY(h, hipMemsetAsync(h->erep.p, 0, sizeof(double) * kVirialSlots, st));
    RepArgs ra{};
    ra.row_info = h->row_info.p; ra.nrows = h->nrows; ra.jlist = h->jlist.p; ra.species = h->species.p;
    ra.pos = d_x; ra.fbuf = h->fbuf64.p; ra.virial = vflag ? h->virial_acc.p : nullptr; ra.erep = h->erep.p;
    ra.tables = h->rep_tables.p; ra.S = m.S; ra.nslots = kVirialSlots; ra.vslots = 1; ra.cutoff = m.rep_cut;
    launch_repulsion(ra, true, st);
  }
  Sae64 sae{};
  for (int s = 0; s < m.S; s++) sae.v[s] = m.sae[s];
  launch_finish64(h->e_rows64.p, M, h->nrows, h->centre_of_row.p, h->ilist.p, h->species.p, sae, h->fbuf64.p, h->ntotal,
                  vflag ? h->virial_acc.p : nullptr, d_f, f_accumulate, d_ev, eflag_atom ? d_eatom : nullptr, h->err_flag.p, st);
  if (m.has_rep) launch_repulsion_energy(h->erep.p, kVirialSlots, d_ev, st);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, take_launch_error());
  return ANI_OK;
}

// the per-step pipeline on device-resident inputs; the list of this epoch is already in the handle's buffers
// ---- one step of the fp32 pipeline, in pieces: a whole step (run_step) strings them together on one stream; the split
// entry points (ani_step_begin / ani_step_ghosts_ready / ani_step_finish) cut it where the ghost exchange happens ----
struct StepCtx {
  const double* d_x = nullptr;
  int eflag_atom = 0, vflag = 0, f_accumulate = 0;
  double *d_f = nullptr, *d_ev = nullptr, *d_eatom = nullptr;
  bool fold = false;   // whole-step device call: an installed ghost fold applies (pack gathers the images, finish folds their forces)
};

int step_prologue(ani_handle* h, const StepCtx& c, bool timed, hipStream_t st) {
  if (h->ntotal >= (1 << 28)) {   // the compact lists keep an atom index in 28 bits (AevArgs::cl_j)
    h->err = "more than 2^28 atoms (owned + ghost) on one rank";
    return ANI_ERR_ARG;
  }
  HIP_TRY(h, h->xyzs.reserve(h->ntotal));
  HIP_TRY(h, h->fbuf.reserve((size_t)h->ntotal * 4));  // one float4 per atom
  HIP_TRY(h, h->virial_acc.reserve(9 * kVirialSlots));
  HIP_TRY(h, h->err_flag.reserve(1, true));
  HIP_TRY(h, h->row_ctr.reserve((size_t)3 * kTicketMaxGroups * kTicketStride, true));
  h->evt = nullptr;
  if (timed && h->timing) {
    if (h->evt_used + 6 > h->evt_pool.size()) {
      const size_t old = h->evt_pool.size();
      h->evt_pool.resize(old + 6 * 64, nullptr);
      for (size_t i = old; i < h->evt_pool.size(); i++) HIP_TRY(h, hipEventCreate(&h->evt_pool[i]));
    }
    h->evt = &h->evt_pool[h->evt_used];
    h->evt_used += 6;
  }
  if (h->evt) HIP_TRY(h, hipEventRecord(h->evt[0], st));
  HIP_TRY(h, h->origin.reserve(3));
  if (h->need_origin) {
    launch_origin(c.d_x, h->ntotal, h->origin.p, st);
    h->need_origin = false;
  }
  if (h->model.has_rep) {
    HIP_TRY(h, h->erep.reserve(kVirialSlots));
    HIP_TRY(h, hipMemsetAsync(h->erep.p, 0, sizeof(double) * kVirialSlots, st));
  }
  return ANI_OK;
}

// atoms [i0, i1) -> fp32 positions, force accumulators cleared; `first`: also the step's virial / energy accumulators
void step_pack(ani_handle* h, const StepCtx& c, int i0, int i1, bool first, hipStream_t st) {
  const bool fold = h->fold_nghost >= 0 && c.fold;
  launch_pack(c.d_x, h->species.p, i0, i1, h->cmap, h->xyzs.p, h->fbuf.p, first ? h->virial_acc.p : nullptr,
              first ? c.d_ev : nullptr, h->origin.p, st, fold ? &h->fold : nullptr);
}

// rows: 0 = every row, 1 = the rows with a ghost among their candidates, 2 = the others (fast path only)
AevArgs step_aev_args(ani_handle* h, const StepCtx& c, int rows) {
  AevArgs a{};
  a.xyzs = h->xyzs.p; a.ilist = h->ilist.p; a.numneigh = h->numneigh.p; a.nbr_off = h->nbr_off.p; a.jlist = h->jlist.p;
  a.centre_of_row = h->centre_of_row.p; a.row_info = h->row_info.p; a.nrows = h->nrows; a.aev = h->aev.p; a.gaev = h->gaev.p; a.fbuf = h->fbuf.p;
  a.virial = c.vflag ? h->virial_acc.p : nullptr;
  a.err_flag = h->err_flag.p;
  a.cl_hdr = h->cl_hdr.p; a.cl_xyz = h->cl_xyz.p; a.cl_j = h->cl_j.p; a.cl_stride = h->cl_stride;
  a.row_list = nullptr; a.k0 = 0; a.kcount = h->nrows;
  a.row_counter = h->row_ctr.p ? h->row_ctr.p + rows * kTicketMaxGroups * kTicketStride : nullptr;
  if (rows == 1) { a.row_list = h->row_list.p; a.k0 = 0; a.kcount = h->n_boundary; }
  if (rows == 2) { a.row_list = h->row_list.p; a.k0 = h->n_boundary; a.kcount = h->nrows - h->n_boundary; }
  // rows by ticket pay from a few rows per wave on (the ~5000 resident waves of a launch each draw three tickets before their
  // first centre): below that the fixed stride is as good and starts at once (option "aev_tickets_min")
  if (a.kcount < h->aev_tickets_min) a.row_counter = nullptr;
  // a centre may read its neighbours' dE/dAEV rows only when every row of the step has been through the MLP: not in a split step
  a.row_of_atom = (rows == 0 && h->aev_sym_radial && h->list_symmetric) ? h->row_of_atom.p : nullptr;
  return a;
}

int step_compact_forward(ani_handle* h, const AevArgs& a, hipStream_t st) {
  if (h->aev_fused) {
    TraceRange tr("ani: neighbour compaction + AEV forward");
    if (h->evt) HIP_TRY(h, hipEventRecord(h->evt[5], st));   // the compaction phase is inside the forward launch
    if (launch_aev_forward_fused(h->ap_run, a, h->max_numneigh, st)) {
      if (h->evt) HIP_TRY(h, hipEventRecord(h->evt[1], st));
      return ANI_OK;
    }
  }
  {
    TraceRange tr("ani: neighbour compaction");
    launch_nbr_compact(h->ap_run, a, h->max_numneigh, st);
  }
  if (h->evt) HIP_TRY(h, hipEventRecord(h->evt[5], st));
  {
    TraceRange tr("ani: AEV forward");
    launch_aev_forward(h->ap_run, a, h->max_numneigh, st);
  }
  if (h->evt) HIP_TRY(h, hipEventRecord(h->evt[1], st));
  return ANI_OK;
}

int step_backward(ani_handle* h, const StepCtx& c, const AevArgs& a, hipStream_t st) {
  const HostModel& m = h->model;
  // the fast backward kernel applies the repulsion in its radial stage (tables by compact species, like the AEV layout);
  // the generic kernel does not, and then the stand-alone kernel adds it
  RepTab rt{};
  if (m.has_rep && h->ap_run.S <= 8) {
    rt.on = 1; rt.cutoff = (float)m.rep_cut; rt.erep = h->erep.p; rt.x64 = c.d_x;
    std::vector<int> act;
    for (int s = 0; s < m.S; s++) if (h->active_mask & (1 << s)) act.push_back(s);
    if ((int)act.size() != h->ap_run.S) { act.clear(); for (int s = 0; s < m.S; s++) act.push_back(s); }
    const size_t n2 = (size_t)m.S * m.S;
    for (size_t ca = 0; ca < act.size(); ca++)
      for (size_t cb = 0; cb < act.size(); cb++) {
        const size_t src = (size_t)act[ca] * m.S + act[cb];
        rt.y[8 * ca + cb] = (float)m.rep_tables[src];
        rt.sa[8 * ca + cb] = (float)m.rep_tables[n2 + src];
        rt.k[8 * ca + cb] = (float)m.rep_tables[2 * n2 + src];
      }
  }
  const bool rep_done = launch_aev_backward(h->ap_run, a, h->max_numneigh, st, rt.on ? &rt : nullptr) && rt.on;
  if (m.has_rep && !rep_done) {
    RepArgs ra{};
    ra.row_info = h->row_info.p; ra.nrows = h->nrows; ra.jlist = h->jlist.p; ra.species = h->species.p;
    ra.pos = c.d_x; ra.fbuf = h->fbuf.p; ra.virial = c.vflag ? h->virial_acc.p : nullptr; ra.erep = h->erep.p;
    ra.tables = h->rep_tables.p; ra.S = m.S; ra.nslots = kVirialSlots; ra.vslots = kVirialSlots; ra.cutoff = m.rep_cut;
    launch_repulsion(ra, false, st);
  }
  return ANI_OK;
}

// forces of the atoms [atom0, atom1) leave the accumulators; `energy`: also the step's energy / virial / repulsion energy
void step_finish(ani_handle* h, const StepCtx& c, int atom0, int atom1, bool energy, hipStream_t st) {
  const HostModel& m = h->model;
  FinishArgs fa{};
  fa.e_rows = h->e_rows.p; fa.M = m.M; fa.nrows = h->nrows; fa.nrows_ld = h->nrows;
  fa.centre_of_row = h->centre_of_row.p; fa.ilist = h->ilist.p; fa.species = h->species.p;
  for (int s = 0; s < m.S; s++) fa.sae[s] = m.sae[s];
  fa.fbuf = h->fbuf.p; fa.atom0 = atom0; fa.atom1 = atom1; fa.energy = energy ? 1 : 0;
  fa.virial_acc = c.vflag ? h->virial_acc.p : nullptr;
  fa.f_out = c.d_f; fa.f_accumulate = c.f_accumulate; fa.ev_out = c.d_ev;
  fa.eatom_out = c.eflag_atom ? c.d_eatom : nullptr;
  fa.err_flag = h->err_flag.p;
  if (h->fold_nghost >= 0 && c.fold) {
    fa.fold_head = h->fold_head.p; fa.fold_next = h->fold_next.p; fa.fold_nlocal = h->nlocal;
    fa.atom1 = std::min(fa.atom1, h->nlocal);   // the ghost rows of d_f are not written: their forces went home
  }
  launch_finish(fa, st);
  if (energy && m.has_rep) launch_repulsion_energy(h->erep.p, kVirialSlots, c.d_ev, st);
}

int run_step(ani_handle* h, const double* d_x, int eflag_atom, int vflag, double* d_f, int f_accumulate, double* d_ev,
             double* d_eatom, hipStream_t st, bool fold = false) {
  if (!h->use_single) return run_step64(h, d_x, eflag_atom, vflag, d_f, f_accumulate, d_ev, d_eatom, st);
  StepCtx c;
  c.fold = fold;
  c.d_x = d_x; c.eflag_atom = eflag_atom; c.vflag = vflag; c.f_accumulate = f_accumulate; c.d_f = d_f; c.d_ev = d_ev; c.d_eatom = d_eatom;
  TraceRange tr_step("ani: step");
  int rc = step_prologue(h, c, true, st);
  if (rc) return rc;
  step_pack(h, c, 0, h->ntotal, true, st);   // also clears fbuf / virial_acc / the energy word
  const AevArgs a = step_aev_args(h, c, 0);
  rc = step_compact_forward(h, a, st);
  if (rc) return rc;
  {
    TraceRange tr("ani: MLP forward + backward");
    rc = compute_mlp(h, st);
    if (rc) return rc;
  }
  if (h->evt) HIP_TRY(h, hipEventRecord(h->evt[2], st));
  TraceRange tr_bwd("ani: AEV backward + finish");
  rc = step_backward(h, c, a, st);
  if (rc) return rc;
  if (h->evt) HIP_TRY(h, hipEventRecord(h->evt[3], st));
  step_finish(h, c, 0, h->ntotal, true, st);
  if (h->evt) HIP_TRY(h, hipEventRecord(h->evt[4], st));
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, take_launch_error());
  return ANI_OK;
}

// Split step (include/ani_hip.h, ani_step_*).  The rows are classed once per list epoch, on first use.
int ensure_row_classes(ani_handle* h, hipStream_t st) {
  if (h->classes_valid) return ANI_OK;
  HIP_TRY(h, h->row_flag.reserve(std::max(h->nrows, 1)));
  HIP_TRY(h, h->row_list.reserve(std::max(h->nrows, 1)));
  HIP_TRY(h, h->row_count.reserve(1));
  launch_row_classes(h->row_info.p, h->jlist.p, h->nrows, h->nlocal, h->row_flag.p, h->row_list.p, h->row_count.p, st);
  int nb = 0;
  HIP_TRY(h, hipMemcpyAsync(&nb, h->row_count.p, sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(h, hipStreamSynchronize(st));   // once per re-neighbouring
  h->n_boundary = nb;
  h->classes_valid = true;
  return ANI_OK;
}

bool split_supported(const ani_handle* h) { return h->use_single && h->cl_stride > 0 && !(h->model.has_rep && h->ap_run.S > 8); }

int check_args(ani_handle* h, int ntotal, int nlocal, long long npairs, int ago) {
  if (!h) return ANI_ERR_ARG;
  if (ntotal < 0 || nlocal < 0 || nlocal > ntotal || npairs < 0) { h->err = "inconsistent ntotal/nlocal/npairs"; return ANI_ERR_ARG; }
  if (npairs >= (1LL << 31)) { h->err = "more than 2^31 neighbour pairs per rank is not supported"; return ANI_ERR_ARG; }
  if (ago != 0 && (!h->have_list || ntotal != h->ntotal || nlocal != h->nlocal)) {
    h->err = "ago != 0 but no neighbour list of matching size is cached (the first call, and every call after a rebuild, must pass ago = 0)";
    return ANI_ERR_ARG;
  }
  return ANI_OK;
}

// spin on a word the device writes into page-locked host memory; false after 5 s (a stalled device)
template <typename T>
bool wait_host_word(const volatile T* w, T want) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    for (int k = 0; k < 512; k++) {
      if (*w == want) {
        std::atomic_thread_fence(std::memory_order_acquire);
        return true;
      }
      __builtin_ia32_pause();
    }
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) return false;
  }
}

int finish_host(ani_handle* h, int ntotal, int nlocal, int eflag_atom, int vflag, double* out_energy, double* out_force,
                double* out_atomic_energies, double* out_virial) {
  hipStream_t st = h->stream;
  if (!h->pinned_ev) {
    // written by kernels and read by a host that may be polling: fine-grained (coherent) whatever HIP_HOST_COHERENT says
    HIP_TRY(h, hipHostMalloc((void**)&h->pinned_ev, sizeof(double) * 16, hipHostMallocCoherent));
    memset(h->pinned_ev, 0, sizeof(double) * 16);
  }
  const double* ev = h->pinned_ev;
  const double stamp = (double)(++h->host_step);
  launch_step_tail(h->ev.p, h->err_flag.p, h->pinned_ev, stamp, st);
  const bool accumulate = h->out_force_accumulate && out_force;
  const size_t rows = h->comm ? (size_t)nlocal : (size_t)ntotal;   // with a communicator the ghost rows have gone home on the device
  if (h->comm) {
    // the pair style's reverse communication, on the device: ghost rows -> their owners' rows (here or on a peer)
    if (ani_comm_reverse(h->comm, h->f64.p, nlocal, st) != ANI_OK) {
      h->err = std::string("ani_comm_reverse: ") + ani_comm_last_error(h->comm);
      return ANI_ERR_DEVICE;
    }
  }
  // Accumulating: the forces come over by a kernel that writes page-locked host memory chunk after chunk and announces every
  // chunk with a word (launch_copy_out); the additions of chunk c run while chunk c + 1 is on its way, and nothing waits for a
  // copy-engine command (each costs ~15 us of gap to its neighbours on this platform).  ANI_FORCE_DMA=1: one copy, then add.
  constexpr int kChunks = ani_handle::kForceChunks;
  const bool polled = accumulate && rows >= 4096 && !getenv("ANI_FORCE_DMA");
  if (accumulate) {
    if (h->stage_force_cap < 3 * rows) {
      if (h->stage_force) HIP_TRY(h, hipHostFree(h->stage_force));
      h->stage_force = nullptr;
      h->stage_force_cap = 3 * rows + 3 * rows / 2 + 64;
      HIP_TRY(h, hipHostMalloc((void**)&h->stage_force, h->stage_force_cap * sizeof(double), hipHostMallocCoherent));
    }
    if (polled) {
      if (!h->stage_flags) {
        HIP_TRY(h, hipHostMalloc((void**)&h->stage_flags, sizeof(unsigned) * kChunks, hipHostMallocCoherent));
        memset(h->stage_flags, 0, sizeof(unsigned) * kChunks);
        HIP_TRY(h, h->copy_ctr.reserve(kChunks, true));
      }
      h->copy_epoch++;
      launch_copy_out(h->f64.p, h->stage_force, (long long)(3 * rows), kChunks, h->copy_ctr.p, h->stage_flags, h->copy_epoch,
                      kCopyOutBlocks, st);
    } else {
      HIP_TRY(h, hipMemcpyAsync(h->stage_force, h->f64.p, sizeof(double) * 3 * rows, hipMemcpyDeviceToHost, st));
    }
  } else if (out_force) {
    HIP_TRY(h, hipMemcpyAsync(out_force, h->f64.p, sizeof(double) * 3 * rows, hipMemcpyDeviceToHost, st));
    if (rows < (size_t)ntotal) memset(out_force + 3 * rows, 0, sizeof(double) * 3 * ((size_t)ntotal - rows));
  }
  if (eflag_atom && out_atomic_energies)
    HIP_TRY(h, hipMemcpyAsync(out_atomic_energies, h->eatom.p, sizeof(double) * (size_t)nlocal, hipMemcpyDeviceToHost, st));
  HIP_TRY(h, hipGetLastError());
  if (polled) {
    // the error word first: nothing may be added on a step that is going to be repeated or reported
    if (!wait_host_word(&h->pinned_ev[11], stamp)) { h->err = "the device did not finish the step within 5 s"; return ANI_ERR_DEVICE; }
  } else {
    HIP_TRY(h, hipStreamSynchronize(st));
  }
  const int flag = (int)ev[10];
  if (flag) {
    HIP_TRY(h, hipStreamSynchronize(st));   // a copy-out in flight lands in the staging buffer only
    HIP_TRY(h, hipMemsetAsync(h->err_flag.p, 0, sizeof(int), st));
    h->sticky_flags |= flag;
    if (flag & 2) {
      // a wait between workgroups of the one-launch MLP ran into its bound: a forward-progress problem on the device, not
      // a capacity problem -- no retry with larger lists
      h->err = "a wait inside the one-launch MLP kernel timed out (device-side stall); results of this step are invalid";
      return ANI_ERR_DEVICE;
    }
    h->err = "an atom has more neighbours inside the radial/angular cutoff than the kernels' LDS capacity (radial: the "
             "longest list, or 3/4 of it unless option full_radial_capacity is set; angular: " + std::to_string(kMaxAng) + ")";
    return ANI_ERR_CAPACITY;
  }
  if (accumulate) {
    const double* __restrict__ src = h->stage_force;
    double* __restrict__ dst = out_force;
    const long long n = (long long)(3 * rows);
    if (polled) {
      for (int c = 0; c < kChunks; c++) {
        if (!wait_host_word(&h->stage_flags[c], h->copy_epoch)) { h->err = "the device did not deliver the forces within 5 s"; return ANI_ERR_DEVICE; }
        const long long a = (n * c / kChunks) & ~1LL, b = c + 1 == kChunks ? n : ((n * (c + 1) / kChunks) & ~1LL);   // as copy_out_kernel
        for (long long k = a; k < b; k++) dst[k] += src[k];
      }
      HIP_TRY(h, hipStreamSynchronize(st));   // the kernel has delivered everything: this returns at once (per-atom energies, if any)
    } else {
      for (long long k = 0; k < n; k++) dst[k] += src[k];
    }
  }
  if (out_energy) *out_energy = ev[0];
  if (vflag && out_virial) memcpy(out_virial, ev + 1, sizeof(double) * 9);
  return ANI_OK;
}

}  // namespace

extern "C" {

int ani_create(const char* model_file, int local_rank, int use_num_models, int use_cuaev, int use_fullnbr, int use_single,
               ani_handle** out) {
  if (out) *out = nullptr;
  if (!model_file || !out) { g_create_error = "null argument"; return ANI_ERR_ARG; }
  if (local_rank < 0) {
    g_create_error = "device 'cpu' (local_rank = -1) is not available: libani_hip is the HIP/gfx950 path only and has no host fallback";
    return ANI_ERR_ARG;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_create_error = "no HIP device visible"; return ANI_ERR_DEVICE; }
  ani_handle* h = new ani_handle;
  h->device = local_rank % ndev;  // src/pair_ani.cpp:269-272
  h->use_cuaev = use_cuaev != 0; h->use_fullnbr = use_fullnbr != 0; h->use_single = use_single != 0;
  std::string e = load_model(model_file, use_num_models, h->model);
  if (!e.empty()) { g_create_error = e; delete h; return ANI_ERR_MODEL; }
  if (hipSetDevice(h->device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
    g_create_error = "cannot initialise HIP device " + std::to_string(h->device);
    delete h;
    return ANI_ERR_DEVICE;
  }
  // the reference's opt-in to reduced-precision products (LAMMPS_ANI_ALLOW_TF32, src/ani_csrc/ani.cpp:41-43): gfx950 has no
  // TF32; the counterpart here is the two-term fp16 split (option "mlp_arith" 2), off unless asked for
  if (const char* tf = getenv("LAMMPS_ANI_ALLOW_TF32")) {
    if (tf[0] && strcmp(tf, "0") != 0) h->mlp_arith = MLP_F16X2;
  }
  if (const char* e = getenv("ANI_AEV_FUSED")) h->aev_fused = atoi(e) != 0;   // experiment knobs: the defaults of the options
  if (const char* e = getenv("ANI_AEV_TICKETS_MIN")) h->aev_tickets_min = atoi(e);
  int rc = upload_model(h);
  if (rc != ANI_OK) { g_create_error = h->err; ani_destroy(h); return rc; }
  // banner, same fields as src/ani_csrc/ani.cpp:88-92
  printf("Successfully loaded the model \nfile: '%s' \ndevice: hip:%d \ndtype: %s \nnbrlist: %s \nani_aev: %s \nuse_num_models: %d/%d\n\n",
         model_file, h->device, h->use_single ? "float (FP32)" : "double (FP64)", h->use_fullnbr ? "full" : "half", h->use_cuaev ? "cuaev" : "pyaev", h->model.M, h->model.M_file);
  fflush(stdout);
  *out = h;
  return ANI_OK;
}

void ani_destroy(ani_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->pinned_ints) { (void)hipHostFree(h->pinned_ints); h->pinned_ints = nullptr; }
  if (h->stage_force) { (void)hipHostFree(h->stage_force); h->stage_force = nullptr; }
  if (h->pinned_ev) { (void)hipHostFree(h->pinned_ev); h->pinned_ev = nullptr; }
  if (h->stage_flags) { (void)hipHostFree(h->stage_flags); h->stage_flags = nullptr; }
  h->copy_ctr.release();
  if (h->side) { (void)hipStreamSynchronize(h->side); (void)hipStreamDestroy(h->side); }
  if (h->ev_mlp) (void)hipEventDestroy(h->ev_mlp);
  if (h->ev_side) (void)hipEventDestroy(h->ev_side);
  for (auto& n : h->nets) {
    for (float* p : n.W) if (p) (void)hipFree(p);
    for (float* p : n.b) if (p) (void)hipFree(p);
    for (float* p : n.WT) if (p) (void)hipFree(p);
    if (n.W0c) (void)hipFree(n.W0c);
    if (n.WT0c) (void)hipFree(n.WT0c);
    if (n.w_out) (void)hipFree(n.w_out);
    if (n.b_out) (void)hipFree(n.b_out);
    for (auto& sp : n.sp) {
      for (unsigned short* p : sp.W) if (p) (void)hipFree(p);
      for (unsigned short* p : sp.WT) if (p) (void)hipFree(p);
      if (sp.W0c) (void)hipFree(sp.W0c);
      if (sp.WT0c) (void)hipFree(sp.WT0c);
    }
    for (double* p : n.W64) if (p) (void)hipFree(p);
    for (double* p : n.b64) if (p) (void)hipFree(p);
    for (double* p : n.WT64) if (p) (void)hipFree(p);
    if (n.W0c64) (void)hipFree(n.W0c64);
    if (n.WT0c64) (void)hipFree(n.WT0c64);
    if (n.w_out64) (void)hipFree(n.w_out64);
    if (n.b_out64) (void)hipFree(n.b_out64);
  }
  h->species.release(); h->ilist.release(); h->numneigh.release(); h->jlist.release(); h->jraw.release(); h->nbr_off.release();
  h->fold_head.release(); h->fold_next.release(); h->fold_bad.release(); h->nb_ovf.release(); h->nb_xq.release(); h->row_of_centre.release(); h->centre_of_row.release(); h->bucket_info.release(); h->err_flag.release(); h->row_ctr.release(); h->row_of_atom.release();
  h->xyzs.release(); h->cl_xyz.release(); h->cl_hdr.release(); h->cl_j.release(); h->row_info.release(); h->row_flag.release(); h->row_list.release(); h->row_count.release(); h->x64.release(); h->f64.release(); h->ev.release(); h->eatom.release(); h->origin.release();
  h->rep_tables.release(); h->erep.release();
  h->nb_cell_id.release(); h->nb_cell_count.release(); h->nb_cell_start.release(); h->nb_cursor.release(); h->nb_order.release(); h->nb_xs.release();
  h->virial_acc.release(); h->aev.release(); h->gaev.release(); h->act.release(); h->aev64.release(); h->gaev64.release(); h->act64.release(); h->e_rows64.release(); h->fbuf64.release(); h->e_rows.release(); h->fbuf.release();
  for (int fi = 0; fi < 4; fi++) free_fused(h, fi);
  h->fused_counter.release(); h->gaev_parts.release(); h->fused_sched.release();
  free_chain_plan(h->chain_plan);
  for (auto& e : h->evt_pool) if (e) (void)hipEventDestroy(e);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

const char* ani_last_error(const ani_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }
int ani_num_models(const ani_handle* h) { return h ? h->model.M_file : 0; }
int ani_use_num_models(const ani_handle* h) { return h ? h->model.M : 0; }
int ani_num_species(const ani_handle* h) { return h ? h->model.S : 0; }
int ani_aev_length(const ani_handle* h) { return h ? h->model.aev_len : 0; }
double ani_cutoff_radial(const ani_handle* h) { return h ? h->model.Rcr : 0; }
double ani_cutoff_angular(const ani_handle* h) { return h ? h->model.Rca : 0; }

int ani_compute_full_device(ani_handle* h, int ntotal, int nlocal, const int* d_species, const double* d_x, int64_t npairs,
                            const int* d_ilist, const int* d_jlist, const int* d_numneigh, int ago, int eflag_atom, int vflag,
                            double* d_f, double* d_ev, double* d_eatom, void* stream) {
  int rc = check_args(h, ntotal, nlocal, npairs, ago);
  if (rc) return rc;
  if (!d_x || !d_ev) { h->err = "null device pointer"; return ANI_ERR_ARG; }
  HIP_TRY(h, hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;  // NULL = the HIP default stream (what torch's default stream is)
  if (ago == 0) {
    if (!d_species || !d_ilist || !d_jlist || !d_numneigh) { h->err = "null list pointer with ago == 0"; return ANI_ERR_ARG; }
    h->ntotal = ntotal; h->nlocal = nlocal; h->npairs = npairs;
    HIP_TRY(h, h->species.reserve(ntotal));
    HIP_TRY(h, h->ilist.reserve(nlocal));
    HIP_TRY(h, h->numneigh.reserve(nlocal));
    HIP_TRY(h, h->jlist.reserve(npairs));
    HIP_TRY(h, h->jraw.reserve(npairs));
    HIP_TRY(h, hipMemcpyAsync(h->species.p, d_species, sizeof(int) * (size_t)ntotal, hipMemcpyDeviceToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(h->ilist.p, d_ilist, sizeof(int) * (size_t)nlocal, hipMemcpyDeviceToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(h->numneigh.p, d_numneigh, sizeof(int) * (size_t)nlocal, hipMemcpyDeviceToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(h->jraw.p, d_jlist, sizeof(int) * (size_t)npairs, hipMemcpyDeviceToDevice, st));
    h->have_list = false;
    // A device-resident caller cannot be handed ANI_ERR_CAPACITY and retried (nothing synchronises; the energy turns NaN
    // and a loop notices steps later): the radial lists get the capacity of the longest candidate list from the start.
    // (The 3/4 estimate of the host entry points saves LDS only; they repeat the step when it was too small.)
    h->ap.full_cap = h->ap_run.full_cap = 1;
    h->list_is_ours = false;
    h->jraw_stride = h->jlist_row_stride = 0;
    h->origin_from_box = false;
    rc = rebuild(h, st);
    if (rc) return rc;
    h->have_list = true;
  }
  rc = run_step(h, d_x, eflag_atom, vflag, d_f, /*accumulate=*/h->dev_overwrite ? 0 : 1, d_ev, d_eatom, st, /*fold=*/true);
  // LAMMPS_ANI_PROFILING (src/pair_ani_kokkos.cpp:68-70,210-212): the host's timers see the device work of this call
  if (rc == ANI_OK && h->profiling) HIP_TRY(h, hipStreamSynchronize(st));
  return rc;
}

// ---- split step: see include/ani_hip.h ----------------------------------------------------------------------
static StepCtx split_ctx(const ani_handle* h) {
  StepCtx c;
  c.d_x = h->split.d_x; c.eflag_atom = h->split.eflag_atom; c.vflag = h->split.vflag; c.f_accumulate = h->dev_overwrite ? 0 : 1;
  c.d_f = h->split.d_f; c.d_ev = h->split.d_ev; c.d_eatom = h->split.d_eatom;
  return c;
}

int ani_step_begin(ani_handle* h, int ntotal, int nlocal, const double* d_x, int eflag_atom, int vflag, double* d_f, double* d_ev,
                   double* d_eatom, void* stream) {
  int rc = check_args(h, ntotal, nlocal, 0, /*ago=*/1);
  if (rc) return rc;
  if (!d_x || !d_ev) { h->err = "null device pointer"; return ANI_ERR_ARG; }
  if (h->split_phase != 0) { h->err = "ani_step_begin: the previous split step was not finished"; return ANI_ERR_ARG; }
  HIP_TRY(h, hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  h->split.d_x = d_x; h->split.eflag_atom = eflag_atom; h->split.vflag = vflag; h->split.d_f = d_f; h->split.d_ev = d_ev; h->split.d_eatom = d_eatom;
  h->split_phase = 1;
  if (!split_supported(h)) return ANI_OK;   // the whole step runs in ani_step_ghosts_ready
  rc = ensure_row_classes(h, st);
  if (rc) { h->split_phase = 0; return rc; }
  const StepCtx c = split_ctx(h);
  TraceRange tr("ani: step, rows without ghosts: pack + compaction + AEV forward");
  rc = step_prologue(h, c, false, st);
  if (rc) { h->split_phase = 0; return rc; }
  step_pack(h, c, 0, h->nlocal, true, st);
  rc = step_compact_forward(h, step_aev_args(h, c, 2), st);
  if (rc) { h->split_phase = 0; return rc; }
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, take_launch_error());
  return ANI_OK;
}

int ani_step_ghosts_ready(ani_handle* h, void* stream) {
  if (!h) return ANI_ERR_ARG;
  if (h->split_phase != 1) { h->err = "ani_step_ghosts_ready without ani_step_begin"; return ANI_ERR_ARG; }
  HIP_TRY(h, hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  const StepCtx c = split_ctx(h);
  h->split_phase = 2;
  if (!split_supported(h)) {
    const int rc = run_step(h, c.d_x, c.eflag_atom, c.vflag, c.d_f, c.f_accumulate, c.d_ev, c.d_eatom, st);
    if (rc) h->split_phase = 0;
    return rc;
  }
  TraceRange tr("ani: step, rows with ghosts + MLP + their backward pass");
  step_pack(h, c, h->nlocal, h->ntotal, false, st);
  const AevArgs a = step_aev_args(h, c, 1);
  int rc = step_compact_forward(h, a, st);
  if (!rc) rc = compute_mlp(h, st);
  if (rc) { h->split_phase = 0; return rc; }
  if (!h->side) {
    int lo = 0, hi = 0;
    HIP_TRY(h, hipDeviceGetStreamPriorityRange(&lo, &hi));   // lo = the numerically largest value = the least urgent
    HIP_TRY(h, hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, lo));
    HIP_TRY(h, hipEventCreateWithFlags(&h->ev_mlp, hipEventDisableTiming));
    HIP_TRY(h, hipEventCreateWithFlags(&h->ev_side, hipEventDisableTiming));
  }
  HIP_TRY(h, hipEventRecord(h->ev_mlp, st));
  rc = step_backward(h, c, a, st);
  if (rc) { h->split_phase = 0; return rc; }
  step_finish(h, c, h->nlocal, h->ntotal, false, st);   // the ghost atoms' forces are complete: no other row touches them
  // the other rows' backward pass, beside the above (launched second, on the less urgent stream)
  HIP_TRY(h, hipStreamWaitEvent(h->side, h->ev_mlp, 0));
  rc = step_backward(h, c, step_aev_args(h, c, 2), h->side);
  if (rc) { h->split_phase = 0; return rc; }
  HIP_TRY(h, hipEventRecord(h->ev_side, h->side));
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, take_launch_error());
  return ANI_OK;
}

int ani_step_finish(ani_handle* h, void* stream) {
  if (!h) return ANI_ERR_ARG;
  if (h->split_phase != 2) { h->err = "ani_step_finish without ani_step_ghosts_ready"; return ANI_ERR_ARG; }
  HIP_TRY(h, hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  h->split_phase = 0;
  if (!split_supported(h)) return ANI_OK;
  const StepCtx c = split_ctx(h);
  TraceRange tr("ani: step, finish");
  HIP_TRY(h, hipStreamWaitEvent(st, h->ev_side, 0));   // the backward pass launched by ani_step_ghosts_ready on the side stream
  step_finish(h, c, 0, h->nlocal, true, st);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, take_launch_error());
  if (h->profiling) HIP_TRY(h, hipStreamSynchronize(st));
  return ANI_OK;
}

void ani_trace_push(const char* name) { if (name) roctxRangePushA(name); }
void ani_trace_pop(void) { roctxRangePop(); }
void ani_trace_mark(const char* name) { if (name) roctxMarkA(name); }

}  // extern "C"

namespace {
// the list build proper; d_species may be h->species.p itself (already in place) or a caller's device array
int build_list(ani_handle* h, int ntotal, int nlocal, const int* d_species, const double* d_x, double cutneigh, const double* lo,
               const double* hi, int64_t* out_npairs, hipStream_t st) {
  int rc = 0;
  NbrGrid g;
  long long ncell = 1;
  for (int k = 0; k < 3; k++) {
    const double len = hi[k] - lo[k];
    if (!(len > 0)) { h->err = "empty bounding box"; return ANI_ERR_ARG; }
    int nc = (int)(len / (h->nbr_half_cells ? 0.5 * cutneigh : cutneigh));
    nc = std::max(1, std::min(nc, 1024));
    g.lo[k] = lo[k]; g.inv[k] = nc / len; g.nc[k] = nc;
    ncell *= nc;
  }
  if (ncell > (1LL << 26)) { h->err = "bounding box too large for the cell grid"; return ANI_ERR_ARG; }
  g.ncell = (int)ncell;
  g.reach = h->nbr_half_cells ? 2 : 1;
  h->have_list = false;
  h->ntotal = ntotal; h->nlocal = nlocal;
  HIP_TRY(h, h->nb_cell_id.reserve(ntotal));
  HIP_TRY(h, h->nb_cell_count.reserve((size_t)g.ncell + 1));
  HIP_TRY(h, h->nb_cell_start.reserve((size_t)g.ncell + 1));
  HIP_TRY(h, h->nb_cursor.reserve(g.ncell));
  HIP_TRY(h, h->nb_order.reserve(ntotal));
  HIP_TRY(h, h->nb_xs.reserve((size_t)3 * ntotal));
  HIP_TRY(h, h->species.reserve(ntotal));
  HIP_TRY(h, h->ilist.reserve(nlocal));
  HIP_TRY(h, h->numneigh.reserve(nlocal));
  HIP_TRY(h, h->nbr_off.reserve((size_t)nlocal + 1));
  NbrScratch s{h->nb_cell_id.p, h->nb_cell_count.p, h->nb_cell_start.p, h->nb_cursor.p, h->nb_order.p, h->nb_xs.p};
  if (d_species != h->species.p)
    HIP_TRY(h, hipMemcpyAsync(h->species.p, d_species, sizeof(int) * (size_t)ntotal, hipMemcpyDeviceToDevice, st));
  // the epoch's origin (the fp32 positions of the steps are relative to it): the middle of the caller's box, no kernel
  if (!h->pinned_ints) HIP_TRY(h, hipHostMalloc((void**)&h->pinned_ints, sizeof(int) * 32, hipHostMallocDefault));
  {
    HIP_TRY(h, h->origin.reserve(3));
    double* mid = reinterpret_cast<double*>(h->pinned_ints + 8);
    for (int k = 0; k < 3; k++) mid[k] = 0.5 * (lo[k] + hi[k]);
    HIP_TRY(h, hipMemcpyAsync(h->origin.p, mid, 3 * sizeof(double), hipMemcpyHostToDevice, st));
    h->origin_from_box = true;
  }
  h->list_is_ours = true;
  h->jraw_stride = h->jlist_row_stride = 0;
  // Rows of a fixed capacity where the longest list of the build before says how long a row can get; the first build of a
  // handle, and any build that overflows its rows, counts first and fills dense segments (count and fill walk the same cells
  // with the same distance tests: 0.125 + 0.154 ms at 100 002 atoms).
  const int cap = h->nbr_cap_hint > 0 ? round_up(h->nbr_cap_hint + h->nbr_cap_hint / 8 + 8, 8) : 0;
  const bool rows_fit = h->nbr_onepass && cap > 0 && (long long)nlocal * cap < (1LL << 31);
  const bool sorted_rows = rows_fit && h->nbr_sorted_rows && nbr_sorted_rows_supported(ntotal, h->model.S, cap);
  if (sorted_rows) {
    // search and species grouping in one kernel straight into jlist; pair count, overflow word, bucket counts and the check of
    // a staged ghost fold all come back behind the ONE synchronisation of rebuild()
    HIP_TRY(h, h->nb_xq.reserve((size_t)std::max(ntotal, 1)));
    s.xq = h->nb_xq.p;
    launch_nbr_bin(d_x, ntotal, g, s, st, h->species.p);
    HIP_TRY(h, h->nb_ovf.reserve(1));
    HIP_TRY(h, h->jlist.reserve((size_t)std::max(nlocal, 1) * cap));
    launch_nbr_sorted_rows(nlocal, ntotal, g, s, cutneigh, h->model.S, cap, h->numneigh.p, h->ilist.p, h->jlist.p, h->nb_ovf.p, st);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, take_launch_error());
    h->jlist_row_stride = cap;
    rc = rebuild(h, st);
    if (rc != kRebuildRowsOverflow) {
      if (rc) return rc;
      h->nbr_cap_hint = h->max_numneigh;
      h->have_list = true;
      if (out_npairs) *out_npairs = h->npairs;
      return ANI_OK;
    }
    h->jlist_row_stride = 0;   // a row was too short: the bins stand, count and fill as below
  } else {
    launch_nbr_bin(d_x, ntotal, g, s, st);
  }
  int total = 0, ovf = 0;
  const bool onepass = rows_fit && !sorted_rows;
  if (onepass) {
    HIP_TRY(h, h->nb_ovf.reserve(1));
    HIP_TRY(h, h->jraw.reserve((size_t)std::max(nlocal, 1) * cap));
    launch_nbr_onepass(nlocal, ntotal, g, s, cutneigh, cap, h->numneigh.p, h->nbr_off.p, h->jraw.p, h->ilist.p, h->nb_ovf.p, st);
    HIP_TRY(h, hipMemcpyAsync(&ovf, h->nb_ovf.p, sizeof(int), hipMemcpyDeviceToHost, st));
  } else {
    launch_nbr_count(nlocal, ntotal, g, s, cutneigh, h->numneigh.p, h->nbr_off.p, st);
  }
  HIP_TRY(h, hipMemcpyAsync(&total, h->nbr_off.p + nlocal, sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(h, hipStreamSynchronize(st));  // the pair count sizes the list buffers (rebuild steps only)
  if (total < 0) { h->err = "more than 2^31 neighbour pairs per rank is not supported"; return ANI_ERR_ARG; }
  h->npairs = total;
  HIP_TRY(h, h->jlist.reserve(total));
  if (onepass && !ovf) {
    h->jraw_stride = cap;
  } else {
    HIP_TRY(h, h->jraw.reserve(total));
    launch_nbr_fill(nlocal, ntotal, g, s, cutneigh, h->nbr_off.p, h->jraw.p, h->ilist.p, st);
  }
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, take_launch_error());
  rc = rebuild(h, st);
  if (rc) return rc;
  h->nbr_cap_hint = h->max_numneigh;
  h->have_list = true;
  if (out_npairs) *out_npairs = total;
  return ANI_OK;
}

int build_list_args(ani_handle* h, int ntotal, int nlocal, const void* species, const void* x, double cutneigh, const double* lo,
                    const double* hi, const char* who) {
  int rc = check_args(h, ntotal, nlocal, 0, 0);
  if (rc) return rc;
  if (!species || !x || !lo || !hi) { h->err = "null pointer"; return ANI_ERR_ARG; }
  if (!(cutneigh >= h->model.Rcr)) { h->err = "cutneigh must be at least the model's radial cutoff"; return ANI_ERR_ARG; }
  if (!h->use_fullnbr) { h->err = std::string(who) + " builds a full list; the handle was created for half lists"; return ANI_ERR_ARG; }
  return ANI_OK;
}
}  // namespace

extern "C" {

int ani_build_list_device(ani_handle* h, int ntotal, int nlocal, const int* d_species, const double* d_x, double cutneigh,
                          const double* lo, const double* hi, int64_t* out_npairs, void* stream) {
  const int rc = build_list_args(h, ntotal, nlocal, d_species, d_x, cutneigh, lo, hi, "ani_build_list_device");
  if (rc) return rc;
  HIP_TRY(h, hipSetDevice(h->device));
  h->ap.full_cap = h->ap_run.full_cap = 1;   // device-resident callers: see ani_compute_full_device
  return build_list(h, ntotal, nlocal, d_species, d_x, cutneigh, lo, hi, out_npairs, (hipStream_t)stream);
}

int ani_build_list(ani_handle* h, int ntotal, int nlocal, const int64_t* species, const double* coordinates, double cutneigh,
                   const double* lo, const double* hi, int64_t* out_npairs) {
  const int rc = build_list_args(h, ntotal, nlocal, species, coordinates, cutneigh, lo, hi, "ani_build_list");
  if (rc) return rc;
  HIP_TRY(h, hipSetDevice(h->device));
  hipStream_t st = h->stream;
  h->h_species32.resize(ntotal);
  for (int i = 0; i < ntotal; i++) h->h_species32[i] = (int)species[i];
  HIP_TRY(h, h->species.reserve(ntotal));
  HIP_TRY(h, h->x64.reserve((size_t)ntotal * 3));
  HIP_TRY(h, hipMemcpyAsync(h->species.p, h->h_species32.data(), sizeof(int) * (size_t)ntotal, hipMemcpyHostToDevice, st));
  HIP_TRY(h, hipMemcpyAsync(h->x64.p, coordinates, sizeof(double) * 3 * (size_t)ntotal, hipMemcpyHostToDevice, st));
  const int rcb = build_list(h, ntotal, nlocal, h->species.p, h->x64.p, cutneigh, lo, hi, out_npairs, st);
  // the step that follows in the same timestep passes the same array with the same contents (include/ani_hip.h): no second upload
  h->x64_from = (rcb == ANI_OK && h->reuse_upload) ? coordinates : nullptr;
  return rcb;
}

int ani_compute_full(ani_handle* h, int ntotal, int nlocal, const int64_t* species, const double* coordinates, int64_t npairs,
                     const int* ilist_unique, const int* jlist, const int* numneigh, int ago, int eflag_atom, int vflag,
                     double* out_energy, double* out_force, double* out_atomic_energies, double* out_virial) {
  int rc = check_args(h, ntotal, nlocal, npairs, ago);
  if (rc) return rc;
  if (!coordinates || !out_force || !out_energy) { h->err = "null pointer argument"; return ANI_ERR_ARG; }
  HIP_TRY(h, hipSetDevice(h->device));
  hipStream_t st = h->stream;
  if (h->comm && h->use_cuaev && !(h->ap.full_cap && h->ap_run.full_cap)) {
    // With a communicator attached the step ends in a matched send / receive with every peer (finish_host), posted before
    // the capacity word is read.  The retry below would post a SECOND exchange on this rank alone: its peers have added the
    // overflowed step's ghost forces already and pair the extra message with their next step (forces one step stale, or a
    // hang at destroy).  So no retry can happen here: the lists are sized for the worst case from the start, as for the
    // device entry points.
    h->ap.full_cap = h->ap_run.full_cap = 1;
    if (ago != 0) {
      rc = rebuild(h, st);
      if (rc) return rc;
    }
  }
  if (ago == 0) {
    if (!species || !ilist_unique || !numneigh || (npairs > 0 && !jlist)) { h->err = "null list pointer with ago == 0"; return ANI_ERR_ARG; }
    h->ntotal = ntotal; h->nlocal = nlocal; h->npairs = npairs;
    h->h_species32.resize(ntotal);
    for (int i = 0; i < ntotal; i++) h->h_species32[i] = (int)species[i];
    HIP_TRY(h, h->species.reserve(ntotal));
    HIP_TRY(h, h->ilist.reserve(nlocal));
    HIP_TRY(h, h->numneigh.reserve(nlocal));
    HIP_TRY(h, h->jlist.reserve(npairs));
    HIP_TRY(h, h->jraw.reserve(npairs));
    HIP_TRY(h, hipMemcpyAsync(h->species.p, h->h_species32.data(), sizeof(int) * (size_t)ntotal, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(h->ilist.p, ilist_unique, sizeof(int) * (size_t)nlocal, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(h->numneigh.p, numneigh, sizeof(int) * (size_t)nlocal, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(h->jraw.p, jlist, sizeof(int) * (size_t)npairs, hipMemcpyHostToDevice, st));
    h->have_list = false;
    h->list_is_ours = false;
    h->jraw_stride = h->jlist_row_stride = 0;
    h->origin_from_box = false;
    rc = rebuild(h, st);
    if (rc) return rc;
    h->have_list = true;
  }
  HIP_TRY(h, h->x64.reserve((size_t)ntotal * 3));
  HIP_TRY(h, h->f64.reserve((size_t)ntotal * 3));
  HIP_TRY(h, h->ev.reserve(10));
  HIP_TRY(h, h->eatom.reserve(std::max(nlocal, 1)));
  if (h->x64_from != coordinates)   // (a re-neighbouring step of the `devlist` mode: ani_build_list uploaded them a moment ago)
    HIP_TRY(h, hipMemcpyAsync(h->x64.p, coordinates, sizeof(double) * 3 * (size_t)ntotal, hipMemcpyHostToDevice, st));
  h->x64_from = nullptr;
  rc = run_step(h, h->x64.p, eflag_atom, vflag, h->f64.p, /*accumulate=*/0, h->ev.p, h->eatom.p, st);
  if (rc) return rc;
  rc = finish_host(h, ntotal, nlocal, eflag_atom, vflag, out_energy, out_force, out_atomic_energies, out_virial);
  if (rc == ANI_ERR_CAPACITY && h->use_cuaev && !h->ap_run.full_cap && !h->comm) {
    // The screened radial lists are sized for 3/4 of the longest candidate list; a system denser than that inside Rcr
    // (small skin, compressed fluid) is input the reference handles, so the step is repeated once with the capacity of
    // the full list and the setting kept for the rest of the run.  An overflow that survives (more than kMaxAng
    // neighbours inside Rca) is still reported.
    h->ap.full_cap = h->ap_run.full_cap = 1;
    fprintf(stderr, "libani_hip: radial neighbour capacity exceeded, continuing with full_radial_capacity = 1\n");
    rc = rebuild(h, st);
    if (rc) return rc;
    rc = run_step(h, h->x64.p, eflag_atom, vflag, h->f64.p, /*accumulate=*/0, h->ev.p, h->eatom.p, st);
    if (rc) return rc;
    rc = finish_host(h, ntotal, nlocal, eflag_atom, vflag, out_energy, out_force, out_atomic_energies, out_virial);
  }
  return rc;
}

int ani_compute_half(ani_handle* h, int ntotal, int nlocal, const int64_t* species, const double* coordinates, int64_t npairs_half,
                     const int64_t* atom_index12, int ago, int eflag_atom, int vflag, double* out_energy, double* out_force,
                     double* out_atomic_energies, double* out_virial) {
  if (!h) return ANI_ERR_ARG;
  // A half pair feeds both of its local ends (src/ani_csrc/ani.cpp:100-180: every atom < nlocal is a centre), so the
  // half list is expanded once per rebuild into the per-centre form the kernels consume.
  if (ago == 0) {
    if (!atom_index12 && npairs_half > 0) { h->err = "null atom_index12 with ago == 0"; return ANI_ERR_ARG; }
    std::vector<int>& num = h->h_half_num;
    std::vector<int>& jl = h->h_half_j;
    num.assign(nlocal, 0);
    for (int64_t p = 0; p < npairs_half; p++) {
      const int64_t a = atom_index12[p], b = atom_index12[npairs_half + p];
      if (a < 0 || a >= ntotal || b < 0 || b >= ntotal) { h->err = "atom_index12 entry out of range"; return ANI_ERR_ARG; }
      if (a < nlocal) num[a]++;
      if (b < nlocal) num[b]++;
    }
    std::vector<int64_t> off(nlocal + 1, 0);
    for (int i = 0; i < nlocal; i++) off[i + 1] = off[i] + num[i];
    jl.resize(off[nlocal]);
    std::vector<int64_t> fill(off.begin(), off.end() - 1);
    for (int64_t p = 0; p < npairs_half; p++) {
      const int64_t a = atom_index12[p], b = atom_index12[npairs_half + p];
      if (a < nlocal) jl[fill[a]++] = (int)b;
      if (b < nlocal) jl[fill[b]++] = (int)a;
    }
    std::vector<int> il(nlocal);
    for (int i = 0; i < nlocal; i++) il[i] = i;
    return ani_compute_full(h, ntotal, nlocal, species, coordinates, (int64_t)jl.size(), il.data(), jl.data(), num.data(), 0,
                            eflag_atom, vflag, out_energy, out_force, out_atomic_energies, out_virial);
  }
  return ani_compute_full(h, ntotal, nlocal, species, coordinates, h->npairs, nullptr, nullptr, nullptr, ago, eflag_atom, vflag,
                          out_energy, out_force, out_atomic_energies, out_virial);
}

// Page-locking of caller arrays for the host entry points: explicit, because a registration belongs to an address range and
// only the owner of the memory knows when that range stops meaning the same pages.
int ani_host_register(const void* p, size_t bytes) {
  if (!p || bytes == 0) return ANI_ERR_ARG;
  if (hipHostRegister(const_cast<void*>(p), bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return ANI_ERR_DEVICE; }
  return ANI_OK;
}
int ani_host_unregister(const void* p) {
  if (!p) return ANI_ERR_ARG;
  if (hipHostUnregister(const_cast<void*>(p)) != hipSuccess) { (void)hipGetLastError(); return ANI_ERR_DEVICE; }
  return ANI_OK;
}

int ani_set_ghost_fold(ani_handle* h, const int64_t* d_owner, const double* d_shift, int nghost, void* stream) {
  if (!h) return ANI_ERR_ARG;
  if (!d_owner || nghost < 0) { h->fold_nghost = -1; return ANI_OK; }   // cleared
  if (!h->use_single) { h->err = "ani_set_ghost_fold: the fp64 kernels do not fold ghosts (use the exchange kernels)"; return ANI_ERR_ARG; }
  if (!h->have_list || nghost != h->ntotal - h->nlocal) { h->err = "ani_set_ghost_fold: no list installed, or nghost differs from the list's"; return ANI_ERR_ARG; }
  if (!d_shift && nghost > 0) { h->err = "ani_set_ghost_fold: null shift"; return ANI_ERR_ARG; }
  HIP_TRY(h, hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(h, h->fold_head.reserve((size_t)std::max(h->nlocal, 1)));
  HIP_TRY(h, h->fold_next.reserve((size_t)std::max(nghost, 1)));
  HIP_TRY(h, h->fold_bad.reserve(1));
  static_assert(sizeof(long long) == sizeof(int64_t), "owner indices are 64-bit");
  launch_ghost_chain(reinterpret_cast<const long long*>(d_owner), nghost, h->nlocal, h->fold_head.p, h->fold_next.p, h->fold_bad.p, st);
  int bad = 0;
  HIP_TRY(h, hipMemcpyAsync(&bad, h->fold_bad.p, sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(h, hipStreamSynchronize(st));   // once per re-neighbouring
  HIP_TRY(h, take_launch_error());
  if (bad) { h->fold_nghost = -1; h->err = "ani_set_ghost_fold: an owner index lies outside [0, nlocal)"; return ANI_ERR_ARG; }
  h->fold.owner = reinterpret_cast<const long long*>(d_owner); h->fold.shift = d_shift; h->fold.nlocal = h->nlocal;
  h->fold_nghost = nghost;
  return ANI_OK;
}

int ani_stage_ghost_fold(ani_handle* h, const int64_t* d_owner, const double* d_shift, int nghost) {
  if (!h) return ANI_ERR_ARG;
  h->stage_owner = nullptr; h->stage_shift = nullptr; h->stage_nghost = -1;
  if (!d_owner || nghost < 0) return ANI_OK;   // cleared
  if (!h->use_single) { h->err = "ani_stage_ghost_fold: the fp64 kernels do not fold ghosts (use the exchange kernels)"; return ANI_ERR_ARG; }
  if (!d_shift && nghost > 0) { h->err = "ani_stage_ghost_fold: null shift"; return ANI_ERR_ARG; }
  h->stage_owner = d_owner; h->stage_shift = d_shift; h->stage_nghost = nghost;
  return ANI_OK;
}

const char* ani_last_mlp_kernel(const ani_handle* h) { return h ? h->last_mlp_kernel : ""; }

int ani_attach_comm(ani_handle* h, void* comm) {
  if (!h) return ANI_ERR_ARG;
  h->comm = static_cast<ani_comm*>(comm);
  return ANI_OK;
}

int ani_debug_get(ani_handle* h, ani_debug_view* out) {
  if (!h || !out) return ANI_ERR_ARG;
  memset(out, 0, sizeof(*out));
  out->nlocal = h->nlocal; out->ntotal = h->ntotal; out->nrows = h->nrows; out->npairs = h->npairs;
  out->d_aev = h->aev.p; out->d_gaev = h->gaev.p; out->d_row_of_centre = h->row_of_centre.p;
  out->aev_stride = h->ap_run.aev_stride; out->aev_active_length = h->ap_run.aev_len;
  if (h->err_flag.p) {   // a diagnostics call: wait for whatever is queued on the device, then read the word as it stands
    int flag = 0;
    if (hipSetDevice(h->device) == hipSuccess && hipDeviceSynchronize() == hipSuccess &&
        hipMemcpy(&flag, h->err_flag.p, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess)
      h->sticky_flags |= flag;
  }
  out->error_flags = h->sticky_flags;
  for (int s = 0; s < kMaxSpecies && s < 16; s++) out->species_count[s] = h->count[s];
  return ANI_OK;
}

int ani_debug_list(ani_handle* h, const int** d_numneigh, const int** d_nbr_off, const int** d_jlist) {
  if (!h || !h->have_list) return ANI_ERR_ARG;
  if (d_numneigh) *d_numneigh = h->numneigh.p;
  if (d_nbr_off) *d_nbr_off = h->nbr_off.p;
  if (d_jlist) *d_jlist = h->jlist.p;   // the installed list: every centre's segment grouped by neighbour species
  return ANI_OK;
}

int ani_debug_colmap(ani_handle* h, int* out) {
  if (!h || !out) return ANI_ERR_ARG;
  for (int c = 0; c < h->ap_run.aev_len && c < (int)h->colmap.size(); c++) out[c] = h->colmap[c];
  return ANI_OK;
}

int ani_set_option(ani_handle* h, const char* name, int value) {
  if (!h || !name) return ANI_ERR_ARG;
  if (strcmp(name, "prune_absent_species") == 0) {
    h->prune = value != 0;
    h->have_list = false;  // takes effect at the next rebuild (ago = 0), which the caller must issue
    h->active_mask = -1;
    return ANI_OK;
  }
  if (strcmp(name, "mlp_chain") == 0) {
    h->mlp_chain = value;
    return ANI_OK;
  }
  if (strcmp(name, "nbr_onepass") == 0) {
    h->nbr_onepass = value != 0;
    return ANI_OK;
  }
  if (strcmp(name, "out_force_accumulate") == 0) {
    h->out_force_accumulate = value != 0;
    return ANI_OK;
  }
  if (strcmp(name, "nbr_sorted_rows") == 0) {
    h->nbr_sorted_rows = value != 0;
    return ANI_OK;
  }
  if (strcmp(name, "nbr_half_cells") == 0) {
    h->nbr_half_cells = value != 0;
    return ANI_OK;
  }
  if (strcmp(name, "reuse_build_list_upload") == 0) {
    h->reuse_upload = value != 0;
    h->x64_from = nullptr;
    return ANI_OK;
  }
  if (strcmp(name, "mlp_fused") == 0) {
    if (value < 0 || value > 3) { h->err = "mlp_fused must be 0, 1, 2 or 3"; return ANI_ERR_ARG; }
    h->mlp_fused = value;
    return ANI_OK;
  }
  if (strcmp(name, "aev_symmetric_radial") == 0) {
    h->aev_sym_radial = value != 0;
    return ANI_OK;
  }
  if (strcmp(name, "aev_tickets_min") == 0) {
    if (value < 0) { h->err = "aev_tickets_min must be >= 0"; return ANI_ERR_ARG; }
    h->aev_tickets_min = value;
    return ANI_OK;
  }
  if (strcmp(name, "aev_fused") == 0) {
    h->aev_fused = value != 0;
    return ANI_OK;
  }
  if (strcmp(name, "mlp_fused_gen") == 0) {
    if (value < 0 || value > 1) { h->err = "mlp_fused_gen must be 0 or 1"; return ANI_ERR_ARG; }
    h->mlp_fused_gen = value;
    h->sched_key[0] = -1;
    return ANI_OK;
  }
  if (strcmp(name, "mlp_fused_rows") == 0) {
    if (value != 0 && value != 64 && value != 128) { h->err = "mlp_fused_rows must be 0, 64 or 128"; return ANI_ERR_ARG; }
    h->mlp_fused_rows = value;
    h->sched_key[0] = -1;
    return ANI_OK;
  }
  if (strcmp(name, "mlp_fused_halves") == 0) {
    if (value < 0 || value > 2) { h->err = "mlp_fused_halves must be 0, 1 or 2"; return ANI_ERR_ARG; }
    h->mlp_fused_halves = value;
    h->sched_key[0] = -1;
    return ANI_OK;
  }
  if (strcmp(name, "mlp_fused_schedule") == 0) {
    h->mlp_fused_sched = value != 0;
    h->sched_key[0] = -1;
    return ANI_OK;
  }
  if (strcmp(name, "mlp_pipeline") == 0) {
    if (value < 0 || value > 2) { h->err = "mlp_pipeline must be 0, 1 or 2"; return ANI_ERR_ARG; }
    h->mlp_pipeline = value;
    return ANI_OK;
  }
  if (strcmp(name, "mlp_arith") == 0) {
    if (value < 0 || value > 2) { h->err = "mlp_arith must be 0 (fp32-input MFMA), 1 (bf16 x 3, exact) or 2 (fp16 x 2)"; return ANI_ERR_ARG; }
    h->mlp_arith = (MlpArith)value;
    return ANI_OK;
  }
  if (strcmp(name, "mlp_split_bf16") == 0) {   // earlier name: 1 = the exact bf16 split, 0 = fp32-input MFMA
    h->mlp_arith = value != 0 ? MLP_BF16X3 : MLP_FP32;
    return ANI_OK;
  }
  if (strcmp(name, "device_overwrite_forces") == 0) {
    h->dev_overwrite = value != 0;
    return ANI_OK;
  }
  if (strcmp(name, "profiling") == 0) {
    h->profiling = value != 0;
    return ANI_OK;
  }
  if (strcmp(name, "full_radial_capacity") == 0) {
    h->ap.full_cap = h->ap_run.full_cap = value != 0;
    return ANI_OK;
  }
  h->err = std::string("unknown option '") + name + "'";
  return ANI_ERR_ARG;
}

// development probe (tools/mlpf_stamps.py): phase cycle counters of a -DABLF_STAMPS build of the fused MLP; 0 otherwise
int ani_debug_fused_stamps(unsigned long long* out32, int reset) {   // 1: fused MLP stamps, 2: AEV backward stamps, 0: neither built in
  const int r = fused_read_stamps(out32, reset);
  return r ? r : aev_read_stamps(out32, reset);
}

int ani_debug_fused_schedule_halves(int ntypes, const int* count, const double* cost, double half_ratio, int bins, int split_mode,
                                    int* split_out, int* items_out, int* off_out, int* n_items_out, double* makespan_out) {
  if (ntypes < 1 || ntypes > kMaxProblems || !count || !cost || bins < 1 || !items_out || !off_out || split_mode < 0 || split_mode > 2 ||
      !(half_ratio > 0.0))
    return ANI_ERR_ARG;
  for (int j = 0; j < ntypes; j++)
    if (count[j] < 0 || !(cost[j] > 0.0)) return ANI_ERR_ARG;
  const double T = fused_schedule_halves(ntypes, count, cost, half_ratio, bins, split_mode, split_out, items_out, off_out, n_items_out);
  if (makespan_out) *makespan_out = T;
  return ANI_OK;
}

int ani_debug_fused_schedule(int ntypes, const int* count, const double* cost, int bins, int* items_out, int* off_out,
                             double* makespan_out) {
  if (ntypes < 1 || ntypes > kMaxProblems || !count || !cost || bins < 1 || !items_out || !off_out) return ANI_ERR_ARG;
  for (int j = 0; j < ntypes; j++)
    if (count[j] < 0 || !(cost[j] > 0.0)) return ANI_ERR_ARG;
  (void)fused_schedule(ntypes, count, cost, bins, items_out, off_out);
  if (makespan_out) {
    std::vector<int> type_of;
    for (int j = 0; j < ntypes; j++) type_of.insert(type_of.end(), count[j], j);
    double worst = 0.0;
    for (int b = 0; b < bins; b++) {
      double load = 0.0;
      for (int i = off_out[b]; i < off_out[b + 1]; i++) load += cost[type_of[items_out[i]]];
      worst = std::max(worst, load);
    }
    *makespan_out = worst;
  }
  return ANI_OK;
}

int ani_debug_read(ani_handle* h, const void* d_src, void* host_dst, uint64_t bytes) {
  if (!h || !d_src || !host_dst) return ANI_ERR_ARG;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipMemcpy(host_dst, d_src, bytes, hipMemcpyDeviceToHost));
  return ANI_OK;
}

int ani_phase_timing(ani_handle* h, int enable) {
  if (!h) return ANI_ERR_ARG;
  h->timing = enable != 0;
  if (enable != 0 && h->evt_pool.empty()) {
    // the first batch of events is made here, not inside the first step that records (a timed step of the caller)
    (void)hipSetDevice(h->device);
    h->evt_pool.resize(6 * 64, nullptr);
    for (hipEvent_t& e : h->evt_pool) HIP_TRY(h, hipEventCreate(&e));
  }
  if (enable == 1) {  // fresh accumulation; 0 (stop) and 2 (resume) keep what has been recorded
    for (double& v : h->phase_ms) v = 0;
    h->phase_calls = 0;
    h->evt_used = 0;
  }
  return ANI_OK;
}

int ani_phase_times(ani_handle* h, double* ms5, int* ncalls) {
  if (!h || !ms5 || !ncalls) return ANI_ERR_ARG;
  // resolve the recorded-but-unread steps (the caller has synchronised, or we wait here for the last event)
  // events of a step: 0 start, 5 after pack + compaction, 1 after AEV forward, 2 after the MLP, 3 after AEV backward, 4 end
  for (size_t b = 0; b + 6 <= h->evt_used; b += 6) {
    HIP_TRY(h, hipEventSynchronize(h->evt_pool[b + 4]));
    float t;
    HIP_TRY(h, hipEventElapsedTime(&t, h->evt_pool[b + 0], h->evt_pool[b + 5])); h->phase_ms[4] += t;
    HIP_TRY(h, hipEventElapsedTime(&t, h->evt_pool[b + 5], h->evt_pool[b + 1])); h->phase_ms[0] += t;
    for (int i = 1; i < 4; i++) {
      HIP_TRY(h, hipEventElapsedTime(&t, h->evt_pool[b + i], h->evt_pool[b + i + 1]));
      h->phase_ms[i] += t;
    }
    h->phase_calls++;
  }
  h->evt_used = 0;
  for (int i = 0; i < 5; i++) ms5[i] = h->phase_ms[i];
  *ncalls = h->phase_calls;
  return ANI_OK;
}

}  // extern "C"
