/*
 * ani_comm.h — ghost exchange between the ranks of a domain-decomposed run, on the device, over RCCL (xGMI), part of
 * libani_hip.so.  No torch, no MPI: plain pointers and sizes; RCCL itself is loaded on first use (dlopen of
 * librccl.so.1), so single-GPU users of the library do not need it installed; environment ANI_COMM_DISABLE_RCCL=1 makes every
 * entry point that needs RCCL fail as if the library were absent.
 *
 * What it replaces in the reference: the pair style sums ghost forces into their owners with LAMMPS' host-side
 * `comm->reverse_comm(this)` and its pack/unpack callbacks (src/pair_ani.cpp:197-201,461-484), after the forces have been
 * copied to the host; ghost positions come from LAMMPS' `comm->forward_comm()` of the Verlet loop.  Both are six dependent
 * face swaps over MPI.  Here every rank sends every peer that holds images of its atoms ONE message per direction, device
 * buffer to device buffer (grouped ncclSend / ncclRecv on the caller's stream; nothing synchronises), and a rank's own
 * periodic images are a device copy.
 *
 * Layout the exchange assumes (what LAMMPS' Comm::borders leaves, and what lammps-ani_amd/comm.py:DomainComm builds):
 *   x[0 .. nlocal)            owned atoms
 *   x[nlocal .. nlocal+nrecv) ghosts, grouped by the rank that owns them, in rank order; recv_counts[p] of them from rank p
 * and, on the owner's side, per epoch (= between two re-neighbourings):
 *   send_idx[0 .. nsend)      owned atoms whose images some rank holds as ghosts, grouped by that rank, in rank order;
 *                             send_counts[p] entries for rank p, in the order rank p stores them
 *   send_shift[nsend][3]      periodic image displacement added to the position of each
 */
#ifndef ANI_COMM_H
#define ANI_COMM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ani_comm ani_comm;

#define ANI_COMM_ID_BYTES 128 /* sizeof(ncclUniqueId) */

/* rank 0 makes the id (ncclGetUniqueId); the caller hands the 128 bytes to every rank by whatever it has (MPI_Bcast in
 * LAMMPS, torch.distributed in the python loop, a file) */
int ani_comm_get_unique_id(void* id128);

/* every rank, collectively (ncclCommInitRank); `device` = HIP device ordinal of this rank.  On failure *out is NULL and
 * ani_comm_last_error(NULL) holds the message. */
int ani_comm_create(int nranks, int rank, const void* id128, int device, ani_comm** out);
/* a communicator of ONE rank without RCCL (nothing is loaded, no id, not collective): every exchange is a device copy or one
 * kernel.  For a caller whose run has a single rank -- all its ghosts are periodic images of its own atoms -- and that wants the
 * ghost forces summed on the device (ani_attach_comm) without bringing RCCL up. */
int ani_comm_create_local(int device, ani_comm** out);
void ani_comm_destroy(ani_comm* c);
const char* ani_comm_last_error(const ani_comm* c);
int ani_comm_rank(const ani_comm* c);
int ani_comm_size(const ani_comm* c);

/*
 * Message layout of one all-to-all with per-peer counts: offsets (in items) of every peer's chunk in the send and the
 * receive buffer, and the totals.  Pure host arithmetic (no device, no RCCL): exported so that the layout the device
 * exchange uses can be tested against another transport (tests/test_comm_plan_gloo.py).
 */
int ani_comm_plan(int nranks, const int64_t* send_counts, const int64_t* recv_counts, int64_t* send_off, int64_t* recv_off,
                  int64_t* nsend, int64_t* nrecv);

/* per-peer message sizes of the next exchange: recv_counts[p] = what rank p's send_counts[me] says (host arrays of nranks
 * entries; synchronises `stream`; rebuild steps only) */
int ani_comm_exchange_counts(ani_comm* c, const int64_t* send_counts, int64_t* recv_counts, void* stream);

/* one all-to-all of device buffers with per-peer counts (items of item_bytes bytes; chunks in rank order, see ani_comm_plan):
 * what Comm::exchange / Comm::borders need at a re-neighbouring (migrating atoms, new ghost shells) */
int ani_comm_alltoallv(ani_comm* c, const void* d_send, const int64_t* send_counts, void* d_recv, const int64_t* recv_counts,
                       int item_bytes, void* stream);

/* installs the maps of the epoch (device pointers are kept, not copied: they must stay valid until the next call) */
int ani_comm_set_epoch(ani_comm* c, const int64_t* send_counts, const int64_t* recv_counts, const int64_t* d_send_idx,
                       const double* d_send_shift);

/* The same for callers that hold the maps on the HOST (the LAMMPS adapter): the arrays are copied into device buffers the
 * communicator owns.  send_shift may be NULL (no image shifts: a caller that only uses the reverse exchange, or whose ghost
 * positions come from elsewhere; ani_comm_forward then adds nothing); ghost_of may be NULL (ghost block already grouped by
 * owner), see ani_comm_set_ghost_order. */
int ani_comm_set_epoch_host(ani_comm* c, const int64_t* send_counts, const int64_t* recv_counts, const int64_t* send_idx,
                            const double* send_shift, const int64_t* ghost_of);

/* Callers whose ghost block is NOT grouped by owning rank (LAMMPS orders ghosts by the swap that brought them): d_ghost_of[k],
 * k < nrecv, is the ghost (0-based within the ghost block) that entry k of the rank-grouped message order stands for; the
 * exchanges then go through a second staging buffer (one gather / scatter kernel more).  NULL = the ghost block is already in
 * message order (the default).  Kept, not copied; reset by ani_comm_set_epoch. */
int ani_comm_set_ghost_order(ani_comm* c, const int64_t* d_ghost_of);

/* forward: d_x[nlocal ..) <- owners' current positions + image shift (LAMMPS Comm::forward_comm) */
int ani_comm_forward(ani_comm* c, double* d_x, int nlocal, void* stream);
/* reverse: ghost rows d_f[nlocal ..) are added into their owners' rows, on whichever rank those are
 * (comm->reverse_comm(this), src/pair_ani.cpp:197-201, with unpack_reverse_comm :474-484 as a device kernel) */
int ani_comm_reverse(ani_comm* c, double* d_f, int nlocal, void* stream);
/* the two halves of ani_comm_reverse for callers that overlap the transfer with other work: _send moves the ghost rows into
 * the staging buffer of their owners (all the communication), _unpack adds the staged rows into d_f */
int ani_comm_reverse_send(ani_comm* c, const double* d_f, int nlocal, void* stream);
int ani_comm_reverse_unpack(ani_comm* c, double* d_f, void* stream);

/* in-place all-reduce of n doubles on the device; op 0 = sum, 1 = max (thermo output, the displacement check of
 * Neighbor::decide) */
int ani_comm_allreduce_f64(ani_comm* c, double* d_buf, int n, int op, void* stream);

/* counters of this communicator: "forward_exchanges", "reverse_exchanges" (calls of ani_comm_forward / ani_comm_reverse_send that
 * had something to move), "alltoalls" (every grouped exchange posted, the rebuild-time ones included), "broken" (1 after an RCCL
 * call failed inside an exchange: every later call fails fast instead of queueing into a half-posted group); -1 = unknown name */
long long ani_comm_get_stat(const ani_comm* c, const char* name);

/* options: "self_through_rccl" (default 0): 1 sends the rank's own chunk through ncclSend / ncclRecv as well instead of a
 * device copy -- a test knob that drives the RCCL point-to-point path on a single GPU */
int ani_comm_set_option(ani_comm* c, const char* name, int value);

#ifdef __cplusplus
}
#endif
#endif /* ANI_COMM_H */
