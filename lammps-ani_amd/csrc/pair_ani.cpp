/* ----------------------------------------------------------------------
   pair_style ani — LAMMPS host adapter over libani_hip.so (see pair_ani.h).

   What this file has to reproduce from the reference adapter (src/pair_ani.cpp), and where it deliberately differs:
     - units real only, comm_reverse(_off) = 3, single_enable = 0                      (ref :36-51)
     - settings grammar and defaults                                                    (ref :285-341)
     - pair_coeff * * only                                                              (ref :347-368)
     - newton_pair must be off for both list kinds; REQ_FULL or default half request    (ref :374-390)
     - per step: x, type-1 and the neighbour list go to the model; forces for local AND ghost atoms come back into
       out_force, ghosts are summed to their owners with reverse_comm(this) when newton is off, then f += out_force
       for all ntotal atoms; eng_vdwl, eatom, virial[6] = xx yy zz xy xz yz            (ref :66-233)
     - restart record layout                                                            (ref :408-455)
   Optional ninth keyword `rcclcomm` (default `mpicomm`): the ghost-force reverse communication (ref :197-201,461-484) runs on
   the device over RCCL before the forces are copied to the host -- one message per peer instead of LAMMPS' six dependent
   host swaps; the owner of every ghost is found once per re-neighbouring with comm->forward_comm(this).
   Differences: atom->x is handed over in place (it is one contiguous ntotal*3 block) instead of being copied;
   numneigh is gathered THROUGH ilist (the reference sums numneigh[ii] but walks firstneigh[ilist[ii]], which
   only agrees when ilist is the identity, SURVEY.md section 7 "numneigh indexing wart"); eatom is scattered through
   ilist; device "cpu" is refused (no host arithmetic in this build); "hip" is accepted next to "cuda".
------------------------------------------------------------------------- */

#include "pair_ani.h"

#include <mpi.h>

#include <cstdlib>
#include <cstring>

#include "atom.h"
#include "comm.h"
#include "domain.h"
#include "error.h"
#include "force.h"
#include "memory.h"
#include "neigh_list.h"
#include "neighbor.h"
#include "update.h"

using namespace LAMMPS_NS;

PairANI::PairANI(LAMMPS* lmp) : Pair(lmp) {
  writedata = 0;
  single_enable = 0;
  restartinfo = 1;
  if (strcmp(update->unit_style, "real") != 0) error->all(FLERR, "Pair ani requires real units");
  comm_reverse = 3;
  comm_reverse_off = 3;
  // src/pair_ani.cpp:49-50: with LAMMPS_ANI_PROFILING set the device work of a step is finished when compute() returns,
  // so LAMMPS' timing breakdown charges it to Pair (the host-pointer entry points used here synchronise in any case; the
  // option also covers the device-resident one)
  const char* prof = getenv("LAMMPS_ANI_PROFILING");
  profiling = prof && *prof && strcmp(prof, "0") != 0;
  printf("LAMMPS_ANI_PROFILING mode: %d\n", profiling ? 1 : 0);
}

PairANI::~PairANI() {
  if (allocated) {
    memory->destroy(setflag);
    memory->destroy(cutsq);
  }
  release_pins();
  free(out_force);
  if (ani) ani_destroy(ani);
  if (acomm) ani_comm_destroy(acomm);
}

// host arrays page-locked for the per-step copies (include/ani_hip.h ani_host_register): released before the memory behind them
// can change hands
void PairANI::release_pins() {
  if (x_registered) { ani_host_unregister(x_registered); x_registered = nullptr; }
  if (out_force_pinned) { ani_host_unregister(out_force); out_force_pinned = false; }
}

void PairANI::grow_out_force(size_t doubles) {
  if (doubles <= out_force_cap && out_force) return;
  if (out_force_pinned) { ani_host_unregister(out_force); out_force_pinned = false; }
  free(out_force);
  out_force_cap = doubles + doubles / 2 + 64;   // grown 1.5x like the reference's arrays (src/pair_ani.cpp:119-127)
  out_force = static_cast<double*>(malloc(out_force_cap * sizeof(double)));
  if (!out_force) error->one(FLERR, "Pair ani: out of memory");
  if (pin_host) out_force_pinned = ani_host_register(out_force, out_force_cap * sizeof(double)) == ANI_OK;
}

void PairANI::allocate() {
  allocated = 1;
  const int n = atom->ntypes;
  memory->create(setflag, n + 1, n + 1, "pair:setflag");
  for (int i = 1; i <= n; i++)
    for (int j = i; j <= n; j++) setflag[i][j] = 0;
  memory->create(cutsq, n + 1, n + 1, "pair:cutsq");
}

/* node-local rank -> device ordinal (the library takes it modulo the visible device count) */
int PairANI::node_local_rank() {
  int rank = 0, local = 0;
  MPI_Comm_rank(world, &rank);
  MPI_Comm node;
  MPI_Comm_split_type(world, MPI_COMM_TYPE_SHARED, rank, MPI_INFO_NULL, &node);
  MPI_Comm_rank(node, &local);
  MPI_Comm_free(&node);
  return local;
}

void PairANI::create_model() {
  if (device_str == "cpu")
    error->all(FLERR, "Pair ani: device 'cpu' is not available in the MI355X build (use 'hip' or 'cuda')");
  if (device_str != "hip" && device_str != "cuda") error->all(FLERR, "Pair ani: device must be hip (or cuda)");
  if (ani) {
    ani_destroy(ani);
    ani = nullptr;
  }
  const int rc = ani_create(model_file.c_str(), node_local_rank(), use_num_models, use_cuaev ? 1 : 0, use_fullnbr ? 1 : 0,
                            use_single ? 1 : 0, &ani);
  if (rc != ANI_OK) error->one(FLERR, std::string("Pair ani: ") + ani_last_error(nullptr));
  if (profiling) ani_set_option(ani, "profiling", 1);
  direct_add = false;
  ani_set_option(ani, "reuse_build_list_upload", 1);   // devlist: ani_build_list and the step that follows see the same atom->x
  if (use_rccl) {
    // one RCCL communicator over the ranks of `world`: rank 0 makes the id, MPI carries it (the only MPI traffic of this path
    // besides the per-rebuild map exchange)
    if (acomm) { ani_comm_destroy(acomm); acomm = nullptr; }
    int me = 0, nprocs = 1;
    MPI_Comm_rank(world, &me);
    MPI_Comm_size(world, &nprocs);
    char id[ANI_COMM_ID_BYTES];
    memset(id, 0, sizeof(id));
    if (me == 0 && ani_comm_get_unique_id(id) != ANI_OK) error->one(FLERR, std::string("Pair ani: ") + ani_comm_last_error(nullptr));
    MPI_Bcast(id, ANI_COMM_ID_BYTES, MPI_BYTE, 0, world);
    if (ani_comm_create(nprocs, me, id, node_local_rank(), &acomm) != ANI_OK)
      error->one(FLERR, std::string("Pair ani: ") + ani_comm_last_error(nullptr));
    ani_attach_comm(ani, acomm);
    comm_forward = 2;
  }
  // One rank (mpicomm or not): every ghost is a periodic image of an owned atom and comm->reverse_comm would only copy inside
  // this process -- the library sums the image rows into their owners on the device instead, through a communicator that never
  // loads RCCL, and 1.1 MB less comes back per step (100 002 atoms).  LAMMPS_ANI_NO_SELF_FOLD=1 keeps the host path.
  use_self_fold = false;
  if (!use_rccl) {
    int nprocs = 1;
    MPI_Comm_size(world, &nprocs);
    const char* off = getenv("LAMMPS_ANI_NO_SELF_FOLD");
    if (nprocs == 1 && !(off && off[0] && strcmp(off, "0") != 0)) {
      if (acomm) { ani_comm_destroy(acomm); acomm = nullptr; }
      if (ani_comm_create_local(node_local_rank(), &acomm) != ANI_OK)
        error->one(FLERR, std::string("Pair ani: ") + ani_comm_last_error(nullptr));
      ani_attach_comm(ani, acomm);
      comm_forward = 2;
      use_self_fold = true;
    }
  }
}

/* pair_style ani cutoff model_file device [num_models] [cuaev|pyaev] [full|half] [single|double] [hostlist|devlist]
   [mpicomm|rcclcomm] */
void PairANI::settings(int narg, char** arg) {
  if (narg < 3) error->all(FLERR, "Illegal pair_style command");
  cutoff = utils::numeric(FLERR, arg[0], false, lmp);
  model_file = arg[1];
  device_str = arg[2];
  use_num_models = narg > 3 ? utils::inumeric(FLERR, arg[3], false, lmp) : -1;
  use_cuaev = true;
  use_fullnbr = true;
  use_single = true;
  if (narg > 4) {
    if (strcmp(arg[4], "cuaev") == 0) use_cuaev = true;
    else if (strcmp(arg[4], "pyaev") == 0) use_cuaev = false;
    else error->all(FLERR, "ani_aev should be cuaev or pyaev");
  }
  if (narg > 5) {
    if (strcmp(arg[5], "full") == 0) use_fullnbr = true;
    else if (strcmp(arg[5], "half") == 0) use_fullnbr = false;
    else error->all(FLERR, "ani_neighbor should be full or half");
  }
  if (narg > 6) {
    if (strcmp(arg[6], "single") == 0) use_single = true;
    else if (strcmp(arg[6], "double") == 0) use_single = false;
    else error->all(FLERR, "precision should be single or double");
  }
  use_devlist = false;
  if (narg > 7) {
    if (strcmp(arg[7], "devlist") == 0) use_devlist = true;
    else if (strcmp(arg[7], "hostlist") != 0) error->all(FLERR, "neighbor list source should be hostlist or devlist");
    if (use_devlist && !use_fullnbr) error->all(FLERR, "devlist builds a full neighbor list: use it with 'full'");
  }
  use_rccl = false;
  if (narg > 8) {
    if (strcmp(arg[8], "rcclcomm") == 0) use_rccl = true;
    else if (strcmp(arg[8], "mpicomm") != 0) error->all(FLERR, "ghost-force communication should be mpicomm or rcclcomm");
  }
  create_model();
}

void PairANI::coeff(int narg, char** arg) {
  if (!allocated) allocate();
  if (narg != 2) error->all(FLERR, "Incorrect args for pair coefficients, it should be set as: pair_coeff * *");
  int ilo, ihi, jlo, jhi;
  const int n = atom->ntypes;
  utils::bounds(FLERR, arg[0], 1, n, ilo, ihi, error);
  utils::bounds(FLERR, arg[1], 1, n, jlo, jhi, error);
  if (ilo != 1 || jlo != 1 || ihi != n || jhi != n)
    error->all(FLERR, "Incorrect args for pair coefficients, it should be set as: pair_coeff * *");
  for (int i = ilo; i <= ihi; i++)
    for (int j = (jlo > i ? jlo : i); j <= jhi; j++) setflag[i][j] = 1;
}

void PairANI::init_style() {
  if (!ani) error->all(FLERR, "Pair ani: no model loaded");
  release_pins();   // a new run: whatever was page-locked for the last one may have been reallocated since
  if (const char* e = getenv("LAMMPS_ANI_NO_PIN")) pin_host = !(e[0] && strcmp(e, "0") != 0);
  if (!pin_host) { free(out_force); out_force = nullptr; out_force_cap = 0; }
  if (force->newton_pair == 1) {
    if (use_fullnbr) error->all(FLERR, "Pair style ANI requires newton pair off when using full neighbor list");
    error->all(FLERR, "Pair style ANI requires newton pair off when using half neighbor list");
  }
  if (use_devlist)
    // the list is built on the device (ani_build_list); LAMMPS keeps its re-neighbouring schedule (neighbor->ago, ghost
    // borders) but an occasional list is only built on demand, and we never ask
    neighbor->add_request(this, NeighConst::REQ_FULL | NeighConst::REQ_OCCASIONAL);
  else if (use_fullnbr)
    neighbor->add_request(this, NeighConst::REQ_FULL);
  else
    neighbor->add_request(this);
}

double PairANI::init_one(int, int) { return cutoff; }

void* PairANI::extract(const char*, int&) { return nullptr; }

void PairANI::compute(int eflag, int vflag) {
  ev_init(eflag, vflag);

  double** x = atom->x;
  double** f = atom->f;
  const int* type = atom->type;
  const int nlocal = atom->nlocal;
  const int ntotal = nlocal + atom->nghost;
  const int ago = neighbor->ago;
  const int inum = use_devlist ? nlocal : list->inum;
  // LAMMPS may have reallocated atom->x while re-neighbouring (memory->grow: the block moves, or grows in place and nmax with
  // it): a registration that no longer describes the block goes before anything else happens; one that still does is kept
  // (registering 3.5 MB costs as much as it saves over an epoch)
  if (ago == 0 && x_registered && (x_registered != (ntotal > 0 ? &x[0][0] : nullptr) || x_registered_nmax != atom->nmax)) {
    ani_host_unregister(x_registered);
    x_registered = nullptr;
  }

  if (use_devlist) {
    if (ago == 0) {
      species.resize(ntotal);
      for (int i = 0; i < ntotal; i++) species[i] = type[i] - 1;
      // a box around owned atoms and ghosts for the cell grid (any box does: atoms outside are clamped into edge cells):
      // the rank's sub-box widened by the ghost cutoff, which LAMMPS has at hand; a triclinic box keeps both in lamda
      // coordinates, so there the positions are scanned
      double lo[3] = {0, 0, 0}, hi[3] = {1, 1, 1};
      for (int k = 0; k < 3; k++) {
        if (!domain->triclinic) {
          lo[k] = domain->sublo[k] - comm->cutghost[k] - 0.25; hi[k] = domain->subhi[k] + comm->cutghost[k] + 0.25;
        } else {
          double a = ntotal > 0 ? x[0][k] : 0.0, b = a;
          for (int i = 1; i < ntotal; i++) { a = x[i][k] < a ? x[i][k] : a; b = x[i][k] > b ? x[i][k] : b; }
          lo[k] = a - 0.25; hi[k] = b + 0.25;
        }
      }
      if (ntotal > 0 &&
          ani_build_list(ani, ntotal, nlocal, species.data(), &x[0][0], cutoff + neighbor->skin, lo, hi, &npairs) != ANI_OK)
        error->one(FLERR, std::string("Pair ani: ") + ani_last_error(ani));
      flat_ilist.resize(nlocal);
      for (int i = 0; i < nlocal; i++) flat_ilist[i] = i;
    }
  } else if (ago == 0) {
    species.resize(ntotal);
    for (int i = 0; i < ntotal; i++) species[i] = type[i] - 1;
    const int* ilist = list->ilist;
    const int* numneigh = list->numneigh;
    int** firstneigh = list->firstneigh;
    npairs = 0;
    for (int ii = 0; ii < inum; ii++) npairs += numneigh[ilist[ii]];
    if (use_fullnbr) {
      if ((int64_t)flat_jlist.capacity() < npairs) flat_jlist.reserve((size_t)(npairs * 1.5));
      flat_jlist.resize(npairs);
      flat_ilist.resize(inum);
      flat_numneigh.resize(inum);
      int64_t p = 0;
      for (int ii = 0; ii < inum; ii++) {
        const int i = ilist[ii];
        const int jnum = numneigh[i];
        const int* jl = firstneigh[i];
        flat_ilist[ii] = i;
        flat_numneigh[ii] = jnum;
        for (int jj = 0; jj < jnum; jj++) flat_jlist[p++] = jl[jj] & NEIGHMASK;
      }
    } else {
      if ((int64_t)atom_index12.capacity() < 2 * npairs) atom_index12.reserve((size_t)(2 * npairs * 1.5));
      atom_index12.resize(2 * npairs);
      int64_t p = 0;
      for (int ii = 0; ii < inum; ii++) {
        const int i = ilist[ii];
        const int jnum = numneigh[i];
        const int* jl = firstneigh[i];
        for (int jj = 0; jj < jnum; jj++) {
          atom_index12[p] = i;
          atom_index12[npairs + p] = jl[jj] & NEIGHMASK;
          p++;
        }
      }
    }
  }

  const bool dev_reverse = use_rccl || use_self_fold;   // the ghost rows go home on the device
  if (dev_reverse && ago == 0) build_rccl_maps(nlocal, atom->nghost);

  // Where nothing reads the raw result before it is added to atom->f (rcclcomm: the ghost rows have gone home on the device;
  // newton on: LAMMPS reverse-communicates f itself) the library adds it into f directly, chunk by chunk beside its own copies
  // (option out_force_accumulate).  atom->f is one block in LAMMPS (memory->create); anything else keeps the buffer.
  const bool direct = (dev_reverse || force->newton) && ntotal > 0 && f[ntotal - 1] == f[0] + 3 * (size_t)(ntotal - 1);
  if (direct != direct_add) {
    ani_set_option(ani, "out_force_accumulate", direct ? 1 : 0);
    direct_add = direct;
  }
  if (!direct) grow_out_force((size_t)ntotal * 3);
  double* const fout = direct ? &f[0][0] : out_force;
  if (ago == 0 && pin_host && !x_registered && ntotal > 0 && atom->nmax >= ntotal &&
      ani_host_register(&x[0][0], sizeof(double) * 3 * (size_t)atom->nmax) == ANI_OK) {
    x_registered = &x[0][0];
    x_registered_nmax = atom->nmax;
  }
  if (eflag_atom) out_eatom.resize(use_fullnbr ? inum : nlocal);
  double out_energy = 0.0;
  double out_virial[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  const double* coords = ntotal > 0 ? &x[0][0] : nullptr;
  static const double dummy[3] = {0, 0, 0};
  if (!coords) coords = dummy;

  int rc;
  if (use_devlist) {
    rc = ntotal > 0 ? ani_compute_full(ani, ntotal, nlocal, nullptr, coords, npairs, nullptr, nullptr, nullptr, /*ago=*/1,
                                       eflag_atom ? 1 : 0, vflag_either ? 1 : 0, &out_energy, fout,
                                       eflag_atom ? out_eatom.data() : nullptr, out_virial)
                    : ANI_OK;
  } else if (use_fullnbr) {
    if (inum != nlocal) error->one(FLERR, "Pair ani: full neighbor list does not cover every local atom");
    rc = ani_compute_full(ani, ntotal, nlocal, species.data(), coords, npairs, flat_ilist.data(), flat_jlist.data(),
                          flat_numneigh.data(), ago, eflag_atom ? 1 : 0, vflag_either ? 1 : 0, &out_energy, fout,
                          eflag_atom ? out_eatom.data() : nullptr, out_virial);
  } else {
    rc = ani_compute_half(ani, ntotal, nlocal, species.data(), coords, npairs, atom_index12.data(), ago, eflag_atom ? 1 : 0,
                          vflag_either ? 1 : 0, &out_energy, fout, eflag_atom ? out_eatom.data() : nullptr, out_virial);
  }
  if (rc != ANI_OK) error->one(FLERR, std::string("Pair ani: ") + ani_last_error(ani));

  // ghost forces go home from out_force (f's ghost entries are not cleared between steps when newton is off)
  if (!force->newton && !dev_reverse) {   // rcclcomm / one rank: the library has summed them on the device (ani_attach_comm)
    ani_trace_push("reverse_comm");   // src/pair_ani.cpp:198-200
    comm->reverse_comm(this);
    ani_trace_pop();
  }

  // with newton off nobody reads the ghost rows of f (the reference adds them all the same and says so,
  // src/pair_ani.cpp:203-210); with newton on LAMMPS reverse-communicates f itself and needs them
  const int nadd = force->newton ? ntotal : nlocal;
  if (direct) {
    // added by the library
  } else if (nadd > 0 && f[nadd - 1] == f[0] + 3 * (size_t)(nadd - 1)) {
    double* __restrict__ ff = &f[0][0];
    const double* __restrict__ src = out_force;
    for (size_t k = 0; k < 3 * (size_t)nadd; k++) ff[k] += src[k];
  } else {
    for (int i = 0; i < nadd; i++) {
      f[i][0] += out_force[3 * i + 0];
      f[i][1] += out_force[3 * i + 1];
      f[i][2] += out_force[3 * i + 2];
    }
  }
  if (eflag_global) eng_vdwl += out_energy;
  if (eflag_atom) {
    if (use_fullnbr)
      for (int ii = 0; ii < inum; ii++) eatom[flat_ilist[ii]] += out_eatom[ii];
    else
      for (int i = 0; i < nlocal; i++) eatom[i] += out_eatom[i];
  }
  if (vflag_either) {
    virial[0] += out_virial[0];
    virial[1] += out_virial[4];
    virial[2] += out_virial[8];
    virial[3] += out_virial[1];
    virial[4] += out_virial[2];
    virial[5] += out_virial[5];
  }
}

/* restart record, byte-compatible with the reference (src/pair_ani.cpp:435-455):
   double cutoff; int use_num_models; bool use_cuaev, use_fullnbr, use_single; int len_model, len_device; chars */
void PairANI::write_restart(FILE* fp) {
  fwrite(&cutoff, sizeof(double), 1, fp);
  fwrite(&use_num_models, sizeof(int), 1, fp);
  fwrite(&use_cuaev, sizeof(bool), 1, fp);
  fwrite(&use_fullnbr, sizeof(bool), 1, fp);
  fwrite(&use_single, sizeof(bool), 1, fp);
  const int n1 = (int)model_file.size(), n2 = (int)device_str.size();
  fwrite(&n1, sizeof(int), 1, fp);
  fwrite(&n2, sizeof(int), 1, fp);
  fwrite(model_file.c_str(), sizeof(char), n1, fp);
  fwrite(device_str.c_str(), sizeof(char), n2, fp);
}

void PairANI::read_restart(FILE* fp) {
  int n1 = 0, n2 = 0;
  utils::sfread(FLERR, &cutoff, sizeof(double), 1, fp, nullptr, error);
  utils::sfread(FLERR, &use_num_models, sizeof(int), 1, fp, nullptr, error);
  utils::sfread(FLERR, &use_cuaev, sizeof(bool), 1, fp, nullptr, error);
  utils::sfread(FLERR, &use_fullnbr, sizeof(bool), 1, fp, nullptr, error);
  utils::sfread(FLERR, &use_single, sizeof(bool), 1, fp, nullptr, error);
  utils::sfread(FLERR, &n1, sizeof(int), 1, fp, nullptr, error);
  utils::sfread(FLERR, &n2, sizeof(int), 1, fp, nullptr, error);
  model_file.resize(n1);
  device_str.resize(n2);
  utils::sfread(FLERR, &model_file[0], sizeof(char), n1, fp, nullptr, error);
  utils::sfread(FLERR, &device_str[0], sizeof(char), n2, fp, nullptr, error);
  create_model();
}

int PairANI::pack_reverse_comm(int n, int first, double* buf) {
  int m = 0;
  for (int i = first; i < first + n; i++) {
    buf[m++] = out_force[3 * i + 0];
    buf[m++] = out_force[3 * i + 1];
    buf[m++] = out_force[3 * i + 2];
  }
  return m;
}

void PairANI::unpack_reverse_comm(int n, int* list, double* buf) {
  int m = 0;
  for (int i = 0; i < n; i++) {
    const int j = list[i];
    out_force[3 * j + 0] += buf[m++];
    out_force[3 * j + 1] += buf[m++];
    out_force[3 * j + 2] += buf[m++];
  }
}

/* ---- rcclcomm: who owns each ghost, and what each peer holds of ours -------------------------------------------------
   forward_comm(this) carries {owning rank, index there} from every owned atom to all of its ghost images (through ghosts of
   ghosts as LAMMPS' swaps do); the ghosts are then grouped by owner, the counts cross with MPI_Alltoall, the index lists with
   MPI_Alltoallv, and the library keeps the maps on the device until the next re-neighbouring. */
int PairANI::pack_forward_comm(int n, int* list, double* buf, int, int*) {
  int m = 0;
  for (int i = 0; i < n; i++) {
    buf[m++] = owner_info[2 * (size_t)list[i]];
    buf[m++] = owner_info[2 * (size_t)list[i] + 1];
  }
  return m;
}

void PairANI::unpack_forward_comm(int n, int first, double* buf) {
  int m = 0;
  for (int i = first; i < first + n; i++) {
    owner_info[2 * (size_t)i] = buf[m++];
    owner_info[2 * (size_t)i + 1] = buf[m++];
  }
}

void PairANI::build_rccl_maps(int nlocal, int nghost) {
  int me = 0, nprocs = 1;
  MPI_Comm_rank(world, &me);
  MPI_Comm_size(world, &nprocs);
  owner_info.assign(2 * (size_t)(nlocal + nghost), -1.0);
  for (int i = 0; i < nlocal; i++) { owner_info[2 * (size_t)i] = me; owner_info[2 * (size_t)i + 1] = i; }
  comm->forward_comm(this);
  std::vector<int> recv_n(nprocs, 0), send_n(nprocs, 0), rdisp(nprocs + 1, 0), sdisp(nprocs + 1, 0);
  for (int g = 0; g < nghost; g++) {
    const int p = (int)owner_info[2 * (size_t)(nlocal + g)];
    if (p < 0 || p >= nprocs) error->one(FLERR, "Pair ani rcclcomm: a ghost atom has no owner (forward_comm did not reach it)");
    recv_n[p]++;
  }
  for (int p = 0; p < nprocs; p++) rdisp[p + 1] = rdisp[p] + recv_n[p];
  std::vector<int64_t> ghost_of(nghost);
  std::vector<int> lidx(nghost), fill(rdisp.begin(), rdisp.end() - 1);
  for (int g = 0; g < nghost; g++) {   // stable: ghosts of one owner keep their order
    const int p = (int)owner_info[2 * (size_t)(nlocal + g)];
    ghost_of[fill[p]] = g;
    lidx[fill[p]++] = (int)owner_info[2 * (size_t)(nlocal + g) + 1];
  }
  MPI_Alltoall(recv_n.data(), 1, MPI_INT, send_n.data(), 1, MPI_INT, world);
  for (int p = 0; p < nprocs; p++) sdisp[p + 1] = sdisp[p] + send_n[p];
  std::vector<int> mine(sdisp[nprocs]);
  MPI_Alltoallv(lidx.data(), recv_n.data(), rdisp.data(), MPI_INT, mine.data(), send_n.data(), sdisp.data(), MPI_INT, world);
  std::vector<int64_t> send_idx(mine.begin(), mine.end()), sc(send_n.begin(), send_n.end()), rc(recv_n.begin(), recv_n.end());
  for (int64_t v : send_idx)
    if (v < 0 || v >= nlocal) error->one(FLERR, "Pair ani rcclcomm: a peer asked for an atom this rank does not own");
  if (ani_comm_set_epoch_host(acomm, sc.data(), rc.data(), send_idx.data(), nullptr, ghost_of.data()) != ANI_OK)
    error->one(FLERR, std::string("Pair ani: ") + ani_comm_last_error(acomm));
}
