// ani_comm.cpp — the ghost exchange of include/ani_comm.h: grouped ncclSend / ncclRecv on the caller's stream, device
// buffers on both ends.  RCCL is bound at first use (dlopen), the library does not link it.
//
// Reference counterpart: comm->reverse_comm(this) + pack/unpack_reverse_comm (src/pair_ani.cpp:197-201,461-484) and the
// Verlet loop's comm->forward_comm(); both host MPI there.
#include "../../include/ani_comm.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ani_hip.h"
#include "../../include/ani_md.h"

namespace {

struct Rccl {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};

thread_local std::string g_comm_create_error;

// librccl.so.1 as the process already has it (a host program that links RCCL, torch's copy in the python loop), else from
// ROCM_PATH
Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // ANI_COMM_DISABLE_RCCL=1: behave as if librccl were absent (callers then take their other transport: LAMMPS' own reverse
    // communication in the adapter, torch.distributed in the python loop) -- also how the fallback is rehearsed on one card
    if (const char* off = getenv("ANI_COMM_DISABLE_RCCL")) {
      if (off[0] && strcmp(off, "0") != 0) { r.err = "RCCL disabled by ANI_COMM_DISABLE_RCCL"; return; }
    }
    const char* rocm = getenv("ROCM_PATH");
    const std::string fallbacks[] = {"librccl.so.1", std::string(rocm ? rocm : "/opt/rocm") + "/lib/librccl.so.1", "librccl.so"};
    for (const std::string& name : fallbacks) {
      r.so = dlopen(name.c_str(), RTLD_NOW | RTLD_GLOBAL);
      if (r.so) break;
    }
    if (!r.so) { r.err = std::string("cannot load librccl.so.1: ") + dlerror(); return; }
    auto bind = [&](const char* sym) {
      void* p = dlsym(r.so, sym);
      if (!p && r.err.empty()) r.err = std::string("librccl has no symbol ") + sym;
      return p;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId))bind("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))bind("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))bind("ncclCommDestroy");
    r.GroupStart = (decltype(r.GroupStart))bind("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))bind("ncclGroupEnd");
    r.Send = (decltype(r.Send))bind("ncclSend");
    r.Recv = (decltype(r.Recv))bind("ncclRecv");
    r.AllReduce = (decltype(r.AllReduce))bind("ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))bind("ncclGetErrorString");
  });
  return &r;
}

void plan(int n, const int64_t* sc, const int64_t* rc, int64_t* so, int64_t* ro, int64_t* ns, int64_t* nr) {
  int64_t a = 0, b = 0;
  for (int p = 0; p < n; p++) {
    if (so) so[p] = a;
    if (ro) ro[p] = b;
    a += sc[p];
    b += rc[p];
  }
  if (ns) *ns = a;
  if (nr) *nr = b;
}

}  // namespace

struct ani_comm {
  int nranks = 1, rank = 0, device = 0;
  ncclComm_t comm = nullptr;
  bool self_rccl = false;
  bool broken = false;   // an RCCL call failed inside an exchange: peers may be mid-way through it, nothing more is sent
  long long n_forward = 0, n_reverse = 0, n_a2a = 0;   // exchanges posted (ani_comm_get_stat)
  std::string err, err_first;
  // epoch
  std::vector<int64_t> sc, rc, so, ro;
  int64_t nsend = 0, nrecv = 0;
  const int64_t* d_send_idx = nullptr;
  const double* d_send_shift = nullptr;
  // staging: positions packed for sending / forces received for unpacking ([nsend][3] doubles), counts for exchange_counts
  double* stage = nullptr;
  size_t stage_cap = 0;
  const int64_t* d_ghost_of = nullptr;   // message order -> ghost, or NULL (ani_comm_set_ghost_order)
  double* gstage = nullptr;              // [nrecv][3]: the ghost block in message order
  size_t gstage_cap = 0;
  int64_t* d_counts = nullptr;
  // maps handed in as host arrays (ani_comm_set_epoch_host): device copies owned here
  int64_t *own_idx = nullptr, *own_ghost_of = nullptr;
  double* own_shift = nullptr;
  size_t own_idx_cap = 0, own_ghost_cap = 0, own_shift_cap = 0;
};

#define COMM_HIP(c, expr)                                                                         \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess) { (c)->err = std::string(#expr) + ": " + hipGetErrorString(_e); return ANI_ERR_DEVICE; } \
  } while (0)
#define COMM_NCCL(c, expr)                                                                        \
  do {                                                                                            \
    ncclResult_t _r = (expr);                                                                     \
    if (_r != ncclSuccess) { (c)->err = std::string(#expr) + ": " + rccl()->GetErrorString(_r); return ANI_ERR_DEVICE; } \
  } while (0)

namespace {

int reserve_stage(ani_comm* c, size_t doubles) {
  if (doubles <= c->stage_cap && c->stage) return ANI_OK;
  if (c->stage) (void)hipFree(c->stage);
  c->stage = nullptr;
  c->stage_cap = 0;
  const size_t want = doubles + doubles / 2 + 64;
  COMM_HIP(c, hipMalloc((void**)&c->stage, want * sizeof(double)));
  c->stage_cap = want;
  return ANI_OK;
}

template <typename T>
int grow(ani_comm* c, T** p, size_t* cap, size_t n) {
  if (n <= *cap && *p) return ANI_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr; *cap = 0;
  const size_t want = n + n / 2 + 64;
  COMM_HIP(c, hipMalloc((void**)p, want * sizeof(T)));
  *cap = want;
  return ANI_OK;
}

// one all-to-all of byte chunks: chunk p of `send` (offset so[p], sc[p] items) goes to rank p, chunk p of `recv` comes from it
int a2a_bytes(ani_comm* c, const char* send, const int64_t* sc, const int64_t* so, char* recv, const int64_t* rc, const int64_t* ro,
              size_t item, hipStream_t st) {
  Rccl* r = c->comm ? rccl() : nullptr;   // a local communicator (ani_comm_create_local) never gets as far as using it
  if (c->broken) { c->err = "the communicator is unusable after an earlier RCCL failure: " + c->err_first; return ANI_ERR_DEVICE; }
  c->n_a2a++;
  bool any = false;
  for (int p = 0; p < c->nranks; p++) any = any || ((p != c->rank || c->self_rccl) && (sc[p] > 0 || rc[p] > 0));
  if (any) {
    COMM_NCCL(c, r->GroupStart());
    // a failed Send / Recv must not return from inside the bracket: the group would stay open on this thread and every later
    // RCCL call (the next exchange, an all-reduce) would be queued into it and never launched -- a silent hang.  The first
    // error is kept, the group is closed, the communicator is marked unusable.
    ncclResult_t bad = ncclSuccess;
    const char* what = "";
    for (int p = 0; p < c->nranks && bad == ncclSuccess; p++) {
      if (p == c->rank && !c->self_rccl) continue;
      if (sc[p] > 0) { bad = r->Send(send + (size_t)so[p] * item, (size_t)sc[p] * item, ncclChar, p, c->comm, st); what = "ncclSend"; }
      if (bad == ncclSuccess && rc[p] > 0) { bad = r->Recv(recv + (size_t)ro[p] * item, (size_t)rc[p] * item, ncclChar, p, c->comm, st); what = "ncclRecv"; }
    }
    const ncclResult_t end = r->GroupEnd();
    if (bad == ncclSuccess && end != ncclSuccess) { bad = end; what = "ncclGroupEnd"; }
    if (bad != ncclSuccess) {
      c->broken = true;
      c->err_first = std::string(what) + ": " + r->GetErrorString(bad);
      c->err = c->err_first;
      return ANI_ERR_DEVICE;
    }
  }
  if (!c->self_rccl) {
    const int me = c->rank;
    if (sc[me] != rc[me]) { c->err = "the rank's own chunk differs between the send and the receive side"; return ANI_ERR_ARG; }
    if (sc[me] > 0)
      COMM_HIP(c, hipMemcpyAsync(recv + (size_t)ro[me] * item, send + (size_t)so[me] * item, (size_t)sc[me] * item,
                                 hipMemcpyDeviceToDevice, st));
  }
  return ANI_OK;
}

}  // namespace

extern "C" {

int ani_comm_get_unique_id(void* id128) {
  Rccl* r = rccl();
  if (!r->err.empty()) { g_comm_create_error = r->err; return ANI_ERR_DEVICE; }
  if (!id128) { g_comm_create_error = "null argument"; return ANI_ERR_ARG; }
  static_assert(sizeof(ncclUniqueId) == ANI_COMM_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  const ncclResult_t rc = r->GetUniqueId(&id);
  if (rc != ncclSuccess) { g_comm_create_error = std::string("ncclGetUniqueId: ") + r->GetErrorString(rc); return ANI_ERR_DEVICE; }
  memcpy(id128, &id, sizeof(id));
  return ANI_OK;
}

int ani_comm_create(int nranks, int rank, const void* id128, int device, ani_comm** out) {
  if (out) *out = nullptr;
  if (!out || !id128 || nranks < 1 || rank < 0 || rank >= nranks) { g_comm_create_error = "bad argument"; return ANI_ERR_ARG; }
  Rccl* r = rccl();
  if (!r->err.empty()) { g_comm_create_error = r->err; return ANI_ERR_DEVICE; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_comm_create_error = "no HIP device visible"; return ANI_ERR_DEVICE; }
  ani_comm* c = new ani_comm;
  c->nranks = nranks; c->rank = rank; c->device = device % ndev;
  if (hipSetDevice(c->device) != hipSuccess) { g_comm_create_error = "cannot select the HIP device"; delete c; return ANI_ERR_DEVICE; }
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  const ncclResult_t rc = r->CommInitRank(&c->comm, nranks, id, rank);
  if (rc != ncclSuccess) {
    g_comm_create_error = std::string("ncclCommInitRank: ") + r->GetErrorString(rc);
    delete c;
    return ANI_ERR_DEVICE;
  }
  if (hipMalloc((void**)&c->d_counts, sizeof(int64_t) * 2 * (size_t)nranks) != hipSuccess) {
    g_comm_create_error = "hipMalloc failed";
    (void)r->CommDestroy(c->comm);
    delete c;
    return ANI_ERR_DEVICE;
  }
  c->sc.assign(nranks, 0); c->rc.assign(nranks, 0); c->so.assign(nranks, 0); c->ro.assign(nranks, 0);
  *out = c;
  return ANI_OK;
}

int ani_comm_create_local(int device, ani_comm** out) {
  if (out) *out = nullptr;
  if (!out) { g_comm_create_error = "bad argument"; return ANI_ERR_ARG; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_comm_create_error = "no HIP device visible"; return ANI_ERR_DEVICE; }
  ani_comm* c = new ani_comm;
  c->nranks = 1; c->rank = 0; c->device = device % ndev;
  if (hipSetDevice(c->device) != hipSuccess) { g_comm_create_error = "cannot select the HIP device"; delete c; return ANI_ERR_DEVICE; }
  if (hipMalloc((void**)&c->d_counts, sizeof(int64_t) * 2) != hipSuccess) { g_comm_create_error = "hipMalloc failed"; delete c; return ANI_ERR_DEVICE; }
  c->sc.assign(1, 0); c->rc.assign(1, 0); c->so.assign(1, 0); c->ro.assign(1, 0);
  *out = c;   // c->comm stays NULL: every exchange of a one-rank communicator is a device copy or one kernel
  return ANI_OK;
}

void ani_comm_destroy(ani_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  if (c->comm) (void)rccl()->CommDestroy(c->comm);
  if (c->stage) (void)hipFree(c->stage);
  if (c->gstage) (void)hipFree(c->gstage);
  if (c->d_counts) (void)hipFree(c->d_counts);
  if (c->own_idx) (void)hipFree(c->own_idx);
  if (c->own_ghost_of) (void)hipFree(c->own_ghost_of);
  if (c->own_shift) (void)hipFree(c->own_shift);
  delete c;
}

const char* ani_comm_last_error(const ani_comm* c) { return c ? c->err.c_str() : g_comm_create_error.c_str(); }
int ani_comm_rank(const ani_comm* c) { return c ? c->rank : -1; }
int ani_comm_size(const ani_comm* c) { return c ? c->nranks : 0; }

int ani_comm_plan(int nranks, const int64_t* send_counts, const int64_t* recv_counts, int64_t* send_off, int64_t* recv_off,
                  int64_t* nsend, int64_t* nrecv) {
  if (nranks < 1 || !send_counts || !recv_counts) return ANI_ERR_ARG;
  for (int p = 0; p < nranks; p++)
    if (send_counts[p] < 0 || recv_counts[p] < 0) return ANI_ERR_ARG;
  plan(nranks, send_counts, recv_counts, send_off, recv_off, nsend, nrecv);
  return ANI_OK;
}

long long ani_comm_get_stat(const ani_comm* c, const char* name) {
  if (!c || !name) return -1;
  if (strcmp(name, "forward_exchanges") == 0) return c->n_forward;
  if (strcmp(name, "reverse_exchanges") == 0) return c->n_reverse;
  if (strcmp(name, "alltoalls") == 0) return c->n_a2a;
  if (strcmp(name, "broken") == 0) return c->broken ? 1 : 0;
  return -1;
}

int ani_comm_set_option(ani_comm* c, const char* name, int value) {
  if (!c || !name) return ANI_ERR_ARG;
  if (strcmp(name, "self_through_rccl") == 0) {
    if (value && !c->comm) { c->err = "a local communicator (ani_comm_create_local) has no RCCL side"; return ANI_ERR_ARG; }
    c->self_rccl = value != 0;
    return ANI_OK;
  }
  c->err = std::string("unknown option '") + name + "'";
  return ANI_ERR_ARG;
}

int ani_comm_exchange_counts(ani_comm* c, const int64_t* send_counts, int64_t* recv_counts, void* stream) {
  if (!c || !send_counts || !recv_counts) return ANI_ERR_ARG;
  COMM_HIP(c, hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  const int n = c->nranks;
  if (n == 1 && !c->self_rccl) { recv_counts[0] = send_counts[0]; return ANI_OK; }
  COMM_HIP(c, hipMemcpyAsync(c->d_counts, send_counts, sizeof(int64_t) * n, hipMemcpyHostToDevice, st));
  std::vector<int64_t> one(n, 1), off(n);
  for (int p = 0; p < n; p++) off[p] = p;
  const int rc = a2a_bytes(c, reinterpret_cast<const char*>(c->d_counts), one.data(), off.data(), reinterpret_cast<char*>(c->d_counts + n),
                           one.data(), off.data(), sizeof(int64_t), st);
  if (rc) return rc;
  COMM_HIP(c, hipMemcpyAsync(recv_counts, c->d_counts + n, sizeof(int64_t) * n, hipMemcpyDeviceToHost, st));
  COMM_HIP(c, hipStreamSynchronize(st));
  return ANI_OK;
}

int ani_comm_alltoallv(ani_comm* c, const void* d_send, const int64_t* send_counts, void* d_recv, const int64_t* recv_counts,
                       int item_bytes, void* stream) {
  if (!c || !send_counts || !recv_counts || item_bytes <= 0) return ANI_ERR_ARG;
  COMM_HIP(c, hipSetDevice(c->device));
  std::vector<int64_t> so(c->nranks), ro(c->nranks);
  int64_t ns = 0, nr = 0;
  if (ani_comm_plan(c->nranks, send_counts, recv_counts, so.data(), ro.data(), &ns, &nr) != ANI_OK) { c->err = "negative count"; return ANI_ERR_ARG; }
  if ((ns > 0 && !d_send) || (nr > 0 && !d_recv)) { c->err = "null buffer"; return ANI_ERR_ARG; }
  return a2a_bytes(c, static_cast<const char*>(d_send), send_counts, so.data(), static_cast<char*>(d_recv), recv_counts, ro.data(),
                   (size_t)item_bytes, (hipStream_t)stream);
}

int ani_comm_set_epoch(ani_comm* c, const int64_t* send_counts, const int64_t* recv_counts, const int64_t* d_send_idx,
                       const double* d_send_shift) {
  if (!c || !send_counts || !recv_counts) return ANI_ERR_ARG;
  COMM_HIP(c, hipSetDevice(c->device));
  if (ani_comm_plan(c->nranks, send_counts, recv_counts, c->so.data(), c->ro.data(), &c->nsend, &c->nrecv) != ANI_OK) {
    c->err = "negative count";
    return ANI_ERR_ARG;
  }
  if (c->nsend > 0 && (!d_send_idx || !d_send_shift)) { c->err = "null send map"; return ANI_ERR_ARG; }
  memcpy(c->sc.data(), send_counts, sizeof(int64_t) * c->nranks);
  memcpy(c->rc.data(), recv_counts, sizeof(int64_t) * c->nranks);
  c->d_send_idx = d_send_idx;
  c->d_send_shift = d_send_shift;
  c->d_ghost_of = nullptr;
  return reserve_stage(c, (size_t)c->nsend * 3);
}

int ani_comm_set_epoch_host(ani_comm* c, const int64_t* send_counts, const int64_t* recv_counts, const int64_t* send_idx,
                            const double* send_shift, const int64_t* ghost_of) {
  if (!c || !send_counts || !recv_counts) return ANI_ERR_ARG;
  COMM_HIP(c, hipSetDevice(c->device));
  int64_t ns = 0, nr = 0;
  if (ani_comm_plan(c->nranks, send_counts, recv_counts, nullptr, nullptr, &ns, &nr) != ANI_OK) { c->err = "negative count"; return ANI_ERR_ARG; }
  if (ns > 0 && !send_idx) { c->err = "null send map"; return ANI_ERR_ARG; }
  int rc = grow(c, &c->own_idx, &c->own_idx_cap, (size_t)ns);
  if (!rc) rc = grow(c, &c->own_shift, &c->own_shift_cap, (size_t)ns * 3);
  if (!rc && ghost_of) rc = grow(c, &c->own_ghost_of, &c->own_ghost_cap, (size_t)nr);
  if (rc) return rc;
  // plain (synchronous) copies: the caller's arrays may go away when this returns; rebuild steps only
  if (ns > 0) COMM_HIP(c, hipMemcpy(c->own_idx, send_idx, sizeof(int64_t) * (size_t)ns, hipMemcpyHostToDevice));
  if (ns > 0) {
    if (send_shift) COMM_HIP(c, hipMemcpy(c->own_shift, send_shift, sizeof(double) * 3 * (size_t)ns, hipMemcpyHostToDevice));
    else {
      COMM_HIP(c, hipMemset(c->own_shift, 0, sizeof(double) * 3 * (size_t)ns));
      COMM_HIP(c, hipStreamSynchronize(nullptr));   // the null-stream fill is not ordered with the caller's non-blocking stream
    }
  }
  rc = ani_comm_set_epoch(c, send_counts, recv_counts, c->own_idx, c->own_shift);
  if (rc || !ghost_of) return rc;
  if (nr > 0) COMM_HIP(c, hipMemcpy(c->own_ghost_of, ghost_of, sizeof(int64_t) * (size_t)nr, hipMemcpyHostToDevice));
  return ani_comm_set_ghost_order(c, c->own_ghost_of);
}

int ani_comm_set_ghost_order(ani_comm* c, const int64_t* d_ghost_of) {
  if (!c) return ANI_ERR_ARG;
  c->d_ghost_of = d_ghost_of;
  if (!d_ghost_of) return ANI_OK;
  COMM_HIP(c, hipSetDevice(c->device));
  const size_t need = (size_t)c->nrecv * 3;
  if (need > c->gstage_cap || !c->gstage) {
    if (c->gstage) (void)hipFree(c->gstage);
    c->gstage = nullptr;
    c->gstage_cap = 0;
    const size_t want = need + need / 2 + 64;
    COMM_HIP(c, hipMalloc((void**)&c->gstage, want * sizeof(double)));
    c->gstage_cap = want;
  }
  return ANI_OK;
}

int ani_comm_forward(ani_comm* c, double* d_x, int nlocal, void* stream) {
  if (!c || !d_x || nlocal < 0) return ANI_ERR_ARG;
  if (c->nsend == 0 && c->nrecv == 0) return ANI_OK;
  COMM_HIP(c, hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  c->n_forward++;
  const int rc = ani_md_pack_ghosts(d_x, c->d_send_idx, c->d_send_shift, (int)c->nsend, c->stage, st);
  if (rc) { c->err = std::string("pack kernel: ") + hipGetErrorString((hipError_t)rc); return ANI_ERR_DEVICE; }
  if (!c->d_ghost_of)
    return a2a_bytes(c, reinterpret_cast<const char*>(c->stage), c->sc.data(), c->so.data(), reinterpret_cast<char*>(d_x + 3 * (size_t)nlocal),
                     c->rc.data(), c->ro.data(), 3 * sizeof(double), st);
  const int rc2 = a2a_bytes(c, reinterpret_cast<const char*>(c->stage), c->sc.data(), c->so.data(), reinterpret_cast<char*>(c->gstage),
                            c->rc.data(), c->ro.data(), 3 * sizeof(double), st);
  if (rc2) return rc2;
  const int rc3 = ani_md_scatter_rows(d_x + 3 * (size_t)nlocal, c->d_ghost_of, (int)c->nrecv, c->gstage, st);
  if (rc3) { c->err = std::string("scatter kernel: ") + hipGetErrorString((hipError_t)rc3); return ANI_ERR_DEVICE; }
  return ANI_OK;
}

int ani_comm_reverse_send(ani_comm* c, const double* d_f, int nlocal, void* stream) {
  if (!c || !d_f || nlocal < 0) return ANI_ERR_ARG;
  if (c->nsend == 0 && c->nrecv == 0) return ANI_OK;
  COMM_HIP(c, hipSetDevice(c->device));
  c->n_reverse++;
  // the roles swap: what came in as ghosts goes back to where it came from
  const double* src = d_f + 3 * (size_t)nlocal;
  if (c->d_ghost_of) {
    const int rc = ani_md_gather_rows(src, c->d_ghost_of, (int)c->nrecv, c->gstage, stream);
    if (rc) { c->err = std::string("gather kernel: ") + hipGetErrorString((hipError_t)rc); return ANI_ERR_DEVICE; }
    src = c->gstage;
  }
  return a2a_bytes(c, reinterpret_cast<const char*>(src), c->rc.data(), c->ro.data(), reinterpret_cast<char*>(c->stage),
                   c->sc.data(), c->so.data(), 3 * sizeof(double), (hipStream_t)stream);
}

int ani_comm_reverse_unpack(ani_comm* c, double* d_f, void* stream) {
  if (!c || !d_f) return ANI_ERR_ARG;
  if (c->nsend == 0) return ANI_OK;
  COMM_HIP(c, hipSetDevice(c->device));
  const int rc = ani_md_unpack_reverse(d_f, c->d_send_idx, (int)c->nsend, c->stage, stream);
  if (rc) { c->err = std::string("unpack kernel: ") + hipGetErrorString((hipError_t)rc); return ANI_ERR_DEVICE; }
  return ANI_OK;
}

int ani_comm_reverse(ani_comm* c, double* d_f, int nlocal, void* stream) {
  if (c && d_f && nlocal >= 0 && c->nranks == 1 && !c->broken) {
    // one rank: every ghost is an image of an atom of this rank -- no message, ONE kernel (gather, self-copy and unpack as three
    // commands cost 60 us of gaps between a copy engine command and the kernels either side, 100 002 atoms)
    if (c->nsend == 0 && c->nrecv == 0) return ANI_OK;
    COMM_HIP(c, hipSetDevice(c->device));
    c->n_reverse++;
    const int rc1 = ani_md_reverse_ghosts_ordered(d_f, c->d_send_idx, c->d_ghost_of, nlocal, (int)c->nsend, stream);
    if (rc1) { c->err = std::string("reverse kernel: ") + hipGetErrorString((hipError_t)rc1); return ANI_ERR_DEVICE; }
    return ANI_OK;
  }
  const int rc = ani_comm_reverse_send(c, d_f, nlocal, stream);
  return rc ? rc : ani_comm_reverse_unpack(c, d_f, stream);
}

int ani_comm_allreduce_f64(ani_comm* c, double* d_buf, int n, int op, void* stream) {
  if (!c || !d_buf || n < 0 || (op != 0 && op != 1)) return ANI_ERR_ARG;
  if (n == 0) return ANI_OK;
  if (c->broken) { c->err = "the communicator is unusable after an earlier RCCL failure: " + c->err_first; return ANI_ERR_DEVICE; }
  if (!c->comm) return ANI_OK;   // a local communicator: one rank, the buffer is its own reduction
  COMM_HIP(c, hipSetDevice(c->device));
  COMM_NCCL(c, rccl()->AllReduce(d_buf, d_buf, (size_t)n, ncclDouble, op == 0 ? ncclSum : ncclMax, c->comm, (hipStream_t)stream));
  return ANI_OK;
}

}  // extern "C"
