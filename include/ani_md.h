/*
 * ani_md.h — device kernels of the LAMMPS-free timestep loop (lammps-ani_amd/md.py, bench.py, tests).
 *
 * NOT part of the drop-in boundary (that is ani_hip.h).  Under LAMMPS these steps are LAMMPS' own: `fix nve`
 * (initial_integrate / final_integrate), `fix langevin` (post_force), Neighbor::check_distance, and the forward / reverse
 * ghost communication of Comm.  The stand-in loop that produces bench.py's MD rate runs them on the device; fused here so
 * that a step is five small launches around the hot path instead of a dozen tensor operations.
 * Units: LAMMPS `real` (A, fs, g/mol, kcal/mol).  Every pointer is device memory; `stream` is a hipStream_t (NULL: the
 * default stream); nothing synchronises.  Return 0 on success, else a hipError_t value.
 */
#ifndef ANI_MD_H
#define ANI_MD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* fix nve, initial_integrate:  v += dtfm[i] f ;  x += dt v  for the nlocal owned atoms (dtfm[i] = dt/2 * ftm2v / mass_i),
 * and Neighbor::check_distance folded in:  *d2max = max(*d2max, |x_i - x_built_i|^2)  (the caller zeroes *d2max when it
 * has read it).  x, v, f: [n][3]. */
int ani_md_initial_integrate(double* x, double* v, const double* f, const double* dtfm, double dt, int nlocal,
                             const double* x_built, double* d2max, void* stream);

/* fix langevin (post_force) + fix nve final_integrate:  f += g1[i] v + g2[i] r  with r uniform in [-0.5, 0.5) per
 * component (a counter-based generator keyed by seed, step and the atom's global tag: the same atom draws the same
 * numbers whatever rank owns it), skipped when langevin == 0;  then  v += dtfm[i] f. */
int ani_md_final_integrate(double* v, double* f, const double* dtfm, int nlocal, int langevin, const double* g1,
                           const double* g2, const int64_t* tag, uint64_t seed, uint64_t step, void* stream);

/* ani_md_final_integrate of the step that ends (its `step` number) followed by ani_md_initial_integrate of the step that begins, in
 * one pass: for stretches of a run in which nothing looks at the full-step velocities between two steps.  Same arithmetic, same
 * rounding as the two calls. */
int ani_md_final_initial_integrate(double* x, double* v, double* f, const double* dtfm, double dt, int nlocal, int langevin,
                                   const double* g1, const double* g2, const int64_t* tag, uint64_t seed, uint64_t step,
                                   const double* x_built, double* d2max, void* stream);

/* Comm::forward_comm on one rank (periodic self-images):  x[nlocal + g] = x[owner[g]] + shift[g] */
int ani_md_forward_ghosts(double* x, const int64_t* owner, const double* shift, int nlocal, int nghost, void* stream);

/* the pair style's reverse communication on one rank:  f[owner[g]] += f[nlocal + g] */
int ani_md_reverse_ghosts(double* f, const int64_t* owner, int nlocal, int nghost, void* stream);

/* Several ranks: the two halves of the same exchanges around the all-to-all.
 * pack:    out[s] = x[owner[s]] + shift[s]   for the nsend atoms this rank sends (its message buffer, [nsend][3])
 * unpack:  f[owner[s]] += in[s]              for the nsend ghost forces that came back ([nsend][3]) */
int ani_md_pack_ghosts(const double* x, const int64_t* owner, const double* shift, int nsend, double* out, void* stream);
int ani_md_unpack_reverse(double* f, const int64_t* owner, int nsend, const double* in, void* stream);

/* rows of three doubles through an index list:  out[k] = src[idx[k]]  /  dst[idx[k]] = in[k]  (the ghost block of a caller
 * whose ghosts are not grouped by owner, brought into / out of message order: include/ani_comm.h ani_comm_set_ghost_order) */
int ani_md_gather_rows(const double* src, const int64_t* idx, int n, double* out, void* stream);
int ani_md_scatter_rows(double* dst, const int64_t* idx, int n, const double* in, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ANI_MD_H */
