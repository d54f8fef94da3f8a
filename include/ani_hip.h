/*
 * ani_hip.h — C ABI of libani_hip.so, the MI355X (gfx950) implementation of the lammps-ani hot path.
 *
 * This is the drop-in boundary: it replaces the reference's libtorch shim `class ANI`
 * (src/ani_csrc/ani.h:11-85, src/ani_csrc/ani.cpp) and everything beneath it (TorchScript LammpsANI wrapper,
 * torchani cuaev/pyaev, BmmEnsemble).  Plain pointers and sizes only; no C++ or torch types cross it; no
 * exceptions cross it (nonzero return + ani_last_error()).  The LAMMPS-side adapter
 * (lammps-ani_amd/csrc/pair_ani.cpp) and the python binding (lammps-ani_amd/ani_hip.py) both sit on this header.
 *
 * Units at the boundary are the reference's (src/ani_csrc/ani.cpp:246-262): energies kcal/mol, forces
 * kcal/mol/Angstrom, virial kcal/mol, coordinates Angstrom.
 *
 * There is NO cpu device in this library: local_rank < 0 (the reference's `device cpu`) is refused with an
 * error, and nothing here falls back to host arithmetic.
 */
#ifndef ANI_HIP_H
#define ANI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ani_handle ani_handle;

/* src/ani_csrc/ani.h:9 */
#define ANI_HARTREE2KCALMOL 627.5094738898777

/* return codes */
#define ANI_OK 0
#define ANI_ERR_ARG 1      /* bad argument / unsupported option */
#define ANI_ERR_MODEL 2    /* model file unreadable or malformed */
#define ANI_ERR_DEVICE 3   /* HIP runtime error, no device */
#define ANI_ERR_CAPACITY 4 /* a per-atom neighbour count exceeded the kernels' LDS capacity */

/*
 * Replaces ANI::ANI(model_file, local_rank, use_num_models, use_cuaev, use_fullnbr, use_single)
 * (src/ani_csrc/ani.h:31-36, src/ani_csrc/ani.cpp:35-97).
 *   model_file      flat model file (lammps-ani_amd/model_file.py), not a TorchScript archive; an optional trailing block
 *                   carries the tables of the pairwise repulsion the reference attaches to reactive models
 *                   (models/ani_models.py:50-53), which is then added to energy, forces and virial
 *   local_rank      HIP device ordinal (the reference maps node-local rank % device count, src/pair_ani.cpp:255-283);
 *                   -1 = cpu is refused
 *   use_num_models  first n ensemble members, -1 = all (models/lammps_ani.py:332-343)
 *   use_cuaev       1: radial terms screened at Rcr (cuaev behaviour); 0: "pyaev" behaviour, every list pair
 *                   contributes with the cosine cutoff evaluated past Rcr.
 *                   Unlike the reference (models/lammps_ani.py:151-153,211) the virial is computed in both modes.
 *   use_fullnbr     1: ani_compute_full will be used, 0: ani_compute_half
 *   use_single      1: fp32 arithmetic on device (energy/virial sums in fp64) — the fast path (MFMA, tuned AEV kernels);
 *                   0: fp64 throughout (the reference's `double`), a plain correctness path without MFMA
 * On failure *out is NULL and ani_last_error(NULL) holds the message.
 */
int ani_create(const char* model_file, int local_rank, int use_num_models, int use_cuaev, int use_fullnbr,
               int use_single, ani_handle** out);

void ani_destroy(ani_handle* h);

/* message of the last failure on this handle (or, with h == NULL, of the last failed ani_create) */
const char* ani_last_error(const ani_handle* h);

/* model.attr("num_models") / model.attr("use_num_models") (src/ani_csrc/ani.cpp:66,90) and model constants */
int ani_num_models(const ani_handle* h);
int ani_use_num_models(const ani_handle* h);
int ani_num_species(const ani_handle* h);
int ani_aev_length(const ani_handle* h);
double ani_cutoff_radial(const ani_handle* h);
double ani_cutoff_angular(const ani_handle* h);

/*
 * Replaces ANI::compute, full-neighbour-list overload (src/ani_csrc/ani.h:54-67, src/ani_csrc/ani.cpp:183-265).
 *   species[ntotal]        0-based species (LAMMPS type-1, src/pair_ani.cpp:110); read only when ago == 0
 *   coordinates[ntotal*3]  every call
 *   ilist_unique[nlocal], numneigh[nlocal], jlist[npairs]
 *                          the LAMMPS full list flattened in ilist order, neighbour indices already masked with
 *                          NEIGHMASK (src/pair_ani.cpp:129-150); read only when ago == 0 and cached on the device
 *                          otherwise (src/ani_csrc/ani.cpp:213-229).  numneigh[ii] belongs to centre ilist_unique[ii].
 *                          A LAMMPS full list is symmetric between owned atoms (j in i's list exactly when i in j's); a list
 *                          that is not is accepted and costs the symmetric radial collection (option aev_symmetric_radial).
 *   ago                    neighbor->ago: 0 = the list was rebuilt this step
 *   eflag_atom, vflag      whether out_atomic_energies / out_virial are wanted
 *   out_energy             total energy of the nlocal centres, self energies included
 *   out_force[ntotal*3]    overwritten with -dE/dx for local AND ghost atoms (the caller reverse-communicates ghosts)
 *   out_atomic_energies[nlocal]  per-centre energies in ilist order, or NULL
 *   out_virial[9]          row-major 3x3, -sym(sum dE/d(diff) x diff) (models/lammps_ani.py:199-200,215), or NULL
 */
int ani_compute_full(ani_handle* h, int ntotal, int nlocal, const int64_t* species, const double* coordinates,
                     int64_t npairs, const int* ilist_unique, const int* jlist, const int* numneigh, int ago,
                     int eflag_atom, int vflag, double* out_energy, double* out_force, double* out_atomic_energies,
                     double* out_virial);

/*
 * Replaces ANI::compute, half-neighbour-list overload (src/ani_csrc/ani.h:39-51, src/ani_csrc/ani.cpp:100-180).
 *   atom_index12[2*npairs_half]  [0:n] = i, [n:2n] = j (src/pair_ani.cpp:144-145); every atom < nlocal is a centre.
 */
int ani_compute_half(ani_handle* h, int ntotal, int nlocal, const int64_t* species, const double* coordinates,
                     int64_t npairs_half, const int64_t* atom_index12, int ago, int eflag_atom, int vflag,
                     double* out_energy, double* out_force, double* out_atomic_energies, double* out_virial);

/*
 * Device-resident variant; replaces the Kokkos tensor overload (src/ani_csrc/ani.h:70-84,
 * src/ani_csrc/ani.cpp:268-316, caller src/pair_ani_kokkos.cpp:143-191).  Every d_* pointer is device memory on the
 * handle's device; nothing is copied to the host and nothing synchronises.
 *   d_species[ntotal]   int32, read when ago == 0
 *   d_x[ntotal*3]       double positions (LAMMPS atomKK->k_x layout)
 *   d_ilist, d_numneigh, d_jlist   as above, int32, read when ago == 0 (cached / re-bucketed on the device)
 *   d_f[ntotal*3]       double; forces are ADDED in place (src/pair_ani_kokkos.cpp:190-191)
 *   d_ev[10]            double; overwritten with {energy, virial[9]} (virial zero unless vflag)
 *   d_eatom[nlocal]     double per-centre energies (overwritten) or NULL
 *   stream              hipStream_t to enqueue on; NULL is the HIP default (null) stream, i.e. the caller's
 *                       own ordering domain (Kokkos' / torch's default stream), NOT a private stream
 */
int ani_compute_full_device(ani_handle* h, int ntotal, int nlocal, const int* d_species, const double* d_x,
                            int64_t npairs, const int* d_ilist, const int* d_jlist, const int* d_numneigh, int ago,
                            int eflag_atom, int vflag, double* d_f, double* d_ev, double* d_eatom, void* stream);

/*
 * Ghosts that are images of the rank's OWN atoms (one rank with periodic boundaries; the self-images of a rank whose brick spans
 * the box in some direction): their two exchanges of a step folded into the step's own first and last kernel.  With a fold
 * installed, ani_compute_full_device
 *   - computes the position of ghost g (atom nlocal + g) as d_x[d_owner[g]] + d_shift[g] and writes it to d_x as well (the
 *     forward communication: d_x is written although the prototype says const), and
 *   - adds the force rows of an atom's images into the atom's own row of d_f; the ghost rows of d_f are NOT written (the reverse
 *     communication, comm->reverse_comm(this) of src/pair_ani.cpp:197-201 for a one-rank run).
 * Every ghost must be such an image (d_owner[g] in [0, nlocal)): a mixed ghost shell keeps its exchange kernels.  The maps belong
 * to a list epoch: installed after the list (ago == 0 call or ani_build_list_device), cleared by the next one; device pointers,
 * kept, not copied.  d_owner == NULL clears.  fp32 handles only.  One host synchronisation (rebuild steps only).
 */
int ani_set_ghost_fold(ani_handle* h, const int64_t* d_owner, const double* d_shift, int nghost, void* stream);
/*
 * The same fold handed over BEFORE the list it belongs to: the next ani_build_list_device / ani_build_list installs it with the
 * list it builds (nghost must equal that list's ntotal - nlocal, otherwise it is dropped), and its check -- every owner inside
 * [0, nlocal) -- comes back behind the synchronisation the build has anyway instead of one of its own; a failed check makes the
 * build return ANI_ERR_ARG.  Nothing is enqueued by this call.  d_owner == NULL clears what was staged.
 */
int ani_stage_ghost_fold(ani_handle* h, const int64_t* d_owner, const double* d_shift, int nghost);

/*
 * The same device-resident step in three calls, cut where a domain-decomposed caller exchanges ghost data, so that
 * both exchanges can run on another stream beside the work that does not need them (the reference has no
 * counterpart: its forward / reverse communication are LAMMPS' blocking MPI swaps around PairANI::compute,
 * src/pair_ani.cpp:197-201).  At the first use after a re-neighbouring the library classes the centres: those with a
 * ghost atom (index >= nlocal) anywhere in their list, and the rest (one host synchronisation per epoch).
 *   ani_step_begin         needs the positions of the nlocal OWNED atoms only: packs them, screens the neighbour lists
 *                          and computes the AEVs of the centres without ghosts.
 *   ani_step_ghosts_ready  call when d_x[nlocal..ntotal) holds this step's ghost positions (stream-ordered after the
 *                          forward exchange): the centres with ghosts, the MLP of every centre, then the backward pass
 *                          of the centres with ghosts; when its work is done the GHOST rows of d_f are final (no other
 *                          centre touches a ghost atom) and the reverse exchange can start.
 *                          (The backward pass of the centres WITHOUT ghosts is started here as well, on a less urgent
 *                          stream of the library's own, beside the one of the centres with ghosts: both only wait for the
 *                          MLP, the ghost forces still come first, and two half-size kernels side by side end sooner
 *                          than one after the other.)
 *   ani_step_finish        waits for that pass; rows [0, nlocal) of d_f, energy, virial, per-atom energies.
 * Arguments as ani_compute_full_device with ago != 0 (the list of the epoch must be in place: ani_build_list_device or a
 * call with ago == 0).  The three calls may use different streams if the caller orders them with events; results equal
 * those of the one-call step up to the order of the fp32 force atomics.  Models or precisions without the fast kernels
 * (generic AEV shapes, `double`) run the whole step inside ani_step_ghosts_ready.
 */
int ani_step_begin(ani_handle* h, int ntotal, int nlocal, const double* d_x, int eflag_atom, int vflag, double* d_f, double* d_ev,
                   double* d_eatom, void* stream);
int ani_step_ghosts_ready(ani_handle* h, void* stream);
int ani_step_finish(ani_handle* h, void* stream);

/*
 * Device-side neighbour list: does on the device what LAMMPS core does on the host for
 * `neighbor <skin> bin` + the full-list request of src/pair_ani.cpp:219-223, and installs the result in the handle
 * as this epoch's list.  After it, call ani_compute_full_device with ago != 0 and NULL list pointers until the next
 * rebuild.
 *   d_species[ntotal], d_x[ntotal*3]   owned atoms first, then ghosts (ghosts carry the periodic images)
 *   cutneigh                           force cutoff + skin (7.1 A for the reference's inputs)
 *   lo[3], hi[3]                       host doubles: a box around the atoms (sub-domain widened by the ghost cutoff);
 *                                      atoms outside it are still handled correctly (clamped into the edge cells)
 *   out_npairs                         pairs in the list (may be NULL)
 * Centres are the atoms 0..nlocal-1 in order (ilist = identity).  Synchronises `stream` once (the pair count sizes
 * the list); list order is deterministic (by cell, then atom index).
 */
int ani_build_list_device(ani_handle* h, int ntotal, int nlocal, const int* d_species, const double* d_x, double cutneigh,
                          const double* lo, const double* hi, int64_t* out_npairs, void* stream);

/*
 * The same list build for callers that hold species and positions on the HOST (the role of LAMMPS' own neighbour
 * build for a non-Kokkos run: `pair_style ani ... devlist` of lammps-ani_amd/csrc/pair_ani.cpp requests only an
 * occasional list from LAMMPS, never builds it, and calls this at re-neighbouring steps instead of flattening and
 * uploading LAMMPS' host list).  Uploads species[ntotal] (int64, as ani_compute_full takes them) and
 * coordinates[ntotal*3], builds and installs the list.  Follow with ani_compute_full(..., ago != 0, NULL list
 * pointers).  Pairs are selected by distance only: no special-bond exclusions (ANI has no topology).
 */
int ani_build_list(ani_handle* h, int ntotal, int nlocal, const int64_t* species, const double* coordinates, double cutneigh,
                   const double* lo, const double* hi, int64_t* out_npairs);
/* device pointers of the installed list (tests): numneigh[nlocal], its exclusive scan [nlocal+1], jlist[npairs] */
int ani_debug_list(ani_handle* h, const int** d_numneigh, const int** d_nbr_off, const int** d_jlist);

/*
 * Ghost forces over RCCL instead of the caller's host MPI (the reference: comm->reverse_comm(this) after the D2H copy,
 * src/pair_ani.cpp:197-201,461-484).  With a communicator of include/ani_comm.h attached (maps of the epoch installed by the
 * caller: ani_comm_set_epoch, and ani_comm_set_ghost_order when the ghost block is in LAMMPS' swap order), the host-pointer
 * entry points ani_compute_full / ani_compute_half sum every ghost's force into its owner ON THE DEVICE before the forces
 * leave it: out_force[0 .. 3 nlocal) is then complete, the ghost rows come back as zeros, and the caller skips its reverse
 * communication.  `comm` is an `ani_comm*` (NULL detaches); not owned by the handle.
 */
int ani_attach_comm(ani_handle* h, void* comm);

/* last-step diagnostics for tests / roofline accounting (device pointers valid until the next compute or destroy) */
typedef struct {
  int nlocal, ntotal, nrows;       /* nrows: species-bucketed AEV rows incl. padding */
  int64_t npairs;                  /* list pairs cached on the device */
  const float* d_aev;              /* [nrows][aev_length] */
  const float* d_gaev;             /* [nrows][aev_length]  dE/dAEV (Hartree) */
  const int* d_row_of_centre;      /* [nlocal] row of centre ii */
  int species_count[16];           /* centres per species */
  int aev_stride;                  /* floats per row of d_aev / d_gaev */
  int aev_active_length;           /* columns in use: the AEV entries of the species present in the system */
  int error_flags;                 /* every bit the device error word has shown so far (latched at host reads and at each
                                      re-neighbouring): 1 = LDS neighbour capacity exceeded, 2 = a wait inside the
                                      one-launch MLP kernel timed out, 8 = a caller's list was not symmetric between owned
                                      atoms (informational: option aev_symmetric_radial was off for that epoch).  The device entry points cannot return these (nothing
                                      synchronises; the energy becomes NaN): a loop that finds a NaN energy reads this */
} ani_debug_view;
/*
 * Page-lock / release a caller's host array (hipHostRegister / hipHostUnregister), so that the per-step copies of the host
 * entry points from and to it are direct DMA transfers instead of the runtime's staged pageable copies (the reference copies
 * pageable every step, src/ani_csrc/ani.cpp:206-209,250-251).  Explicit, not automatic: a registration belongs to an ADDRESS
 * RANGE, and a range that is freed while registered and handed out again by the allocator would be transferred from stale
 * pages -- release before the memory is freed or reallocated.  The LAMMPS adapter registers its own out_force buffer and, per
 * neighbour-list epoch, the atom->x block (csrc/pair_ani.cpp).  Return ANI_ERR_DEVICE when the runtime refuses (overlap with
 * another registration, limits): the copies then stay pageable, nothing else changes.
 */
int ani_host_register(const void* p, size_t bytes);
int ani_host_unregister(const void* p);

/* name of the kernel that ran the MLP of the last step ("mlp_fused<3>", "mlp_chain", "mlp_pipeline", "gemm_grouped ..."): what a
 * profile's kernel statistics list it under (bench.py's roofline block names it) */
const char* ani_last_mlp_kernel(const ani_handle* h);
int ani_debug_get(ani_handle* h, ani_debug_view* out);
/* out[c], c < aev_active_length: column of the model's full AEV that column c of d_aev holds */
int ani_debug_colmap(ani_handle* h, int* out);

/*
 * Options (take effect at the next call with ago == 0):
 *   "prune_absent_species" (default 1): AEV entries of species that occur nowhere in the system (neither as centre
 *       nor as neighbour) are identically zero; with 1 the kernels work on the remaining columns only (ANI-2x water:
 *       128 of 1008) and the first-layer products use the matching weight columns — the same sums without the zero
 *       terms.  0 forces the full 1008-column layout.
 *   "mlp_fused" (default 1): networks of three hidden layers (every ANI-1x / ANI-2x member) run as ONE launch in which a
 *       workgroup takes a 128-row tile through all six products of every member with the activations in registers
 *       (ani_kernels_mlpf.hip); HBM sees the AEV rows and the dE/dAEV rows only.  0 = the per-layer kernels below.  Needs a
 *       split arithmetic ("mlp_arith" 1 or 2).  With several ensemble members a work item is (tile, member), each
 *       member writing its own dE/dAEV rows, summed afterwards.  One member and fewer than ~16 000 atoms: the chained per-layer
 *       launch is faster and is used.  2 = the fused kernel whatever the size, 3 = the same with a tile's members one after
 *       the other in its workgroup (tests, measurements).  Takes effect at the next call.
 *   "aev_fused" (default 1): the per-step screening of the candidate lists (neighbour compaction) runs inside the forward AEV
 *     launch, each wave on the centre it is about to featurise, instead of as a kernel of its own in front of it; 0 = two
 *     kernels.  Same results; candidate lists longer than 256 entries and AEV shapes off the fast path take the two kernels
 *     whatever the option says.
 *   "out_force_accumulate" (default 0; the LAMMPS adapter turns it on where it adds out_force to atom->f as it is): ani_compute_full
 *     / ani_compute_half ADD the forces into out_force instead of overwriting it -- rows [0, nlocal) with a communicator attached
 *     (the ghost rows have gone home on the device), all ntotal rows otherwise -- through a page-locked buffer of the library, in
 *     chunks, the additions of one chunk running beside the copy of the next.  Nothing is added on a call that returns an error.
 *   "nbr_sorted_rows" (default 1): ani_build_list* searches the cells and groups every centre's neighbours by species in ONE kernel,
 *     into rows of a capacity taken from the longest list of the build before (a build that overflows its rows, and the first one
 *     of a handle, counts and fills dense segments instead); ani_debug_list then returns row offsets i * capacity.  0 = separate
 *     search and grouping kernels.  "nbr_onepass" 0 = always count first, then fill.  "nbr_half_cells" (default 0): cells of half
 *     the cutoff, 5 x 5 x 5 of them searched.  The same neighbour sets in every case.
 *   "mlp_fused_halves" (default 1; sixteen-row kernel): the static schedule of the fused launch may run an item as two HALF items (the
 *     lower half of a workgroup's waves takes 64 of the 128 rows, the upper half only its share of the loads; ~0.7 of the item's
 *     time each) where that shortens the launch -- the items beyond the last full round of workgroups otherwise make a round of
 *     their own with most of the chip idle.  A split the cost model likes is timed against the whole items (three launches each,
 *     once per set of tile counts, on the step that builds the schedule) and kept only if faster; results do not depend on the
 *     choice.  0 = whole items only, 2 = every item as two halves (tests, measurements).
 *   "reuse_build_list_upload" (default 0; the LAMMPS adapter turns it on): the ani_compute_full call that follows an
 *     ani_build_list with the SAME coordinates pointer uses the positions that call uploaded instead of uploading them again.
 *     For callers whose array has not changed in between (the same timestep); a caller that might hand over another array at a
 *     recycled address leaves it off.
 *   "aev_symmetric_radial" (default 1): in the backward pass a centre takes both radial terms of a pair with a neighbour that is
 *     a centre of the same call on itself (the neighbour's term read from the neighbour's dE/dAEV row) and scatters no radial
 *     gradient to it -- two thirds of the pass's global atomics; ghost neighbours keep the scatter.  0 = scatter every radial
 *     gradient.  Same results to the order of the fp32 sums.  Relies on what a LAMMPS full list guarantees (and the built-in
 *     list builder keeps): j is in i's list exactly when i is in j's.  A list handed in by the caller (ani_compute_full,
 *     ani_compute_half, ani_compute_full_device at ago == 0) is CHECKED for this once per epoch (a hash sum over the entries
 *     between owned atoms, ~0.1 ms at 100 000 atoms); an epoch whose list fails runs with the scatter of every term, sets bit 8
 *     of error_flags and says so once on stderr.  Not used by the split step (ani_step_begin ...).
 *   "aev_tickets_min" (default 40000): AEV launches over at least this many rows hand their rows to the waves by ticket (64 groups
 *     of workgroups, a counter each) instead of at a fixed stride; smaller launches keep the stride.  Same results.
 *   "mlp_fused_schedule" (default 1): which workgroup of the fused launch runs which tiles is decided on the host, once per
 *       re-neighbouring, by first-fit-decreasing on the tiles' costs (the smallest makespan that packs them into the CUs);
 *       0 = the workgroups draw tiles from a counter, costliest first.  Takes effect at the next call.
 *   "mlp_arith" (default 1): how the MLP evaluates its fp32 products; fp32 accumulation in every case, takes effect at
 *       the next call.  1, the exact split, is what the reference's "fp32 with TF32 off" means on this chip; 2 is the
 *       counterpart of its opt-in LAMMPS_ANI_ALLOW_TF32 (src/ani_csrc/ani.cpp:41-43), also selected by that variable.
 *         2 = three v_mfma_f32_32x32x16_f16 products of two-term fp16 splits: every operand, scaled by a power of two
 *             (weights so that a layer's largest sits below 2^13, activations by 2^4, gradients by 2^12; undone exactly on
 *             the accumulators), is h + l with two fp16 numbers rounded to nearest -- good to 2^-22 relative (fp32: 2^-24)
 *             for magnitudes within 2^16 of the tensor's range, 2^-25 / scale absolute below -- and the dropped l*l term is
 *             below 2^-22 of the product.  Against the fp64 oracle the forces are as close as with 1 or 0 (the error
 *             of the step is the AEV's fp32 arithmetic; bench.py "parity", tests/test_hip_properties.py).  An activation
 *             beyond 4094 or a gradient beyond 16 Hartree per unit overflows fp16: inf, NaN energy, reported -- not wrong.
 *         1 = six v_mfma_f32_32x32x16_bf16 products of the EXACT three-term bf16 splits (8+8+8 mantissa bits) of both
 *             operands; the three dropped terms are below 2^-23 of the product: one fp32 rounding.
 *         0 = the fp32-input instruction v_mfma_f32_32x32x2_f32 (1/16 of the 16-bit matrix rate on gfx950).
 *   "mlp_split_bf16": earlier name; 1 selects "mlp_arith" 1, 0 selects "mlp_arith" 0.
 *   "mlp_pipeline" (default 0: an experiment; per-layer kernels only): 1 = with one ensemble member, arithmetic 2, layers no wider than
 *       256 and more 64-row tiles than one round of chained workgroups (about 32 000 water atoms) the six MLP products run as ONE
 *       launch of persistent workgroups that walk (layer, tile) items in order, each waiting for the flag of the same rows' item
 *       of the layer before; 2 = at any size, 0 = never.  Not a default for anything since round 4: one run of the test suite in
 *       three returned a stale tile at 60 000 atoms (a read across XCDs inside a launch).  Takes effect at the next call.
 *   "mlp_chain" (default 1; per-layer kernels only): with one ensemble member and few row tiles (small systems) the six MLP products run as one
 *       chained launch instead of six grouped ones; 2 = at any size (measurement knob), 0 = never.  Takes effect at the
 *       next call.
 *   "device_overwrite_forces" (default 0): ani_compute_full_device ADDS forces into d_f like the reference's Kokkos
 *       overload (src/pair_ani_kokkos.cpp:190-191); 1 makes it overwrite d_f[0 .. 3*ntotal) instead, for callers that
 *       would otherwise clear the array first.  Takes effect at the next call.
 *   "profiling" (default 0): the reference's LAMMPS_ANI_PROFILING (src/pair_ani.cpp:49-50, src/pair_ani_kokkos.cpp:68-70,
 *       210-212): ani_compute_full_device synchronises its stream before it returns, so the caller's host timers
 *       (LAMMPS' timing breakdown) charge the device work to the pair style.  The host-pointer entry points always
 *       synchronise.  Takes effect at the next call.
 *   "full_radial_capacity" (default 0): with the radial screen at Rcr (use_cuaev = 1) the kernels reserve LDS for 3/4
 *       of the longest neighbour list (>= 128 entries) per centre -- a uniform 7.1 A list holds 37 % of its entries
 *       inside 5.1 A -- instead of all of it; a centre that needs more raises ANI_ERR_CAPACITY.  1 reserves the full
 *       list length (takes effect at the next call).
 */
int ani_set_option(ani_handle* h, const char* name, int value);
/*
 * Profiler timeline markers (rocprofv3 --marker-trace; roctx underneath).  The library brackets its own phases ("ani:
 * neighbour compaction", "ani: AEV forward", "ani: MLP forward + backward", "ani: AEV backward + finish") and marks
 * "neighbor list rebuilt" like the reference (NVTX: src/ani_csrc/ani.cpp:128,215); the adapter uses these three to
 * bracket its own work, e.g. "reverse_comm" (src/pair_ani.cpp:198-200), without a tracing dependency of its own.
 */
void ani_trace_push(const char* name);
void ani_trace_pop(void);
void ani_trace_mark(const char* name);

/* development probe: per-phase cycle counters of a diagnostic (-DABLF_STAMPS) build of the fused MLP kernel
 * (tools/mlpf_stamps.py); returns 0 and writes nothing in the shipped build */
int ani_debug_fused_stamps(unsigned long long* out16, int reset);

/* development probe (host arithmetic only, no device): the static schedule the fused MLP launch uses ("mlp_fused_schedule") for
 * ntypes kinds of work items -- count[j] items of relative cost[j], numbered type after type -- on `bins` workgroups:
 * items_out[sum count] = item numbers, workgroup after workgroup; off_out[bins + 1]; *makespan_out = the largest workgroup load */
int ani_debug_fused_schedule(int ntypes, const int* count, const double* cost, int bins, int* items_out, int* off_out,
                             double* makespan_out);
/* the same with HALF items ("mlp_fused_halves"; the sixteen-row kernel runs item sum(count) + 2 i + h as half h of item i, at half_ratio
 * of its cost): split_mode 0 none, 1 searched, 2 every item; split_out[ntypes] = how many of each type's LAST items were cut,
 * items_out[2 sum(count)] (capacity), *n_items_out = entries written, *makespan_out = the bound the packing was made for */
int ani_debug_fused_schedule_halves(int ntypes, const int* count, const double* cost, double half_ratio, int bins, int split_mode,
                                    int* split_out, int* items_out, int* off_out, int* n_items_out, double* makespan_out);

/* copy `bytes` from a device pointer of the view to host memory (synchronises the handle's stream first) */
int ani_debug_read(ani_handle* h, const void* d_src, void* host_dst, uint64_t bytes);

/* kernel timing hooks for bench.py: brackets the phases of the NEXT computes with hipEvents on the compute stream;
 * ani_phase_times returns accumulated milliseconds {aev_fwd, mlp, aev_bwd, other (finish), pack + neighbour compaction}
 * and the call count.  enable: 1 = start a fresh accumulation, 0 = stop recording (what was recorded stays readable),
 * 2 = resume. */
int ani_phase_timing(ani_handle* h, int enable);
int ani_phase_times(ani_handle* h, double* ms5, int* ncalls);

#ifdef __cplusplus
}
#endif
#endif /* ANI_HIP_H */
