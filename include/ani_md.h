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
/* the same for a ghost block that is not in message order:  f[owner[g]] += f[nlocal + ghost_of[g]]  (ghost_of NULL: as above) */
int ani_md_reverse_ghosts_ordered(double* f, const int64_t* owner, const int64_t* ghost_of, int nlocal, int nghost, void* stream);

/* Several ranks: the two halves of the same exchanges around the all-to-all.
 * pack:    out[s] = x[owner[s]] + shift[s]   for the nsend atoms this rank sends (its message buffer, [nsend][3])
 * unpack:  f[owner[s]] += in[s]              for the nsend ghost forces that came back ([nsend][3]) */
int ani_md_pack_ghosts(const double* x, const int64_t* owner, const double* shift, int nsend, double* out, void* stream);
int ani_md_unpack_reverse(double* f, const int64_t* owner, int nsend, const double* in, void* stream);

/* rows of three doubles through an index list:  out[k] = src[idx[k]]  /  dst[idx[k]] = in[k]  (the ghost block of a caller
 * whose ghosts are not grouped by owner, brought into / out of message order: include/ani_comm.h ani_comm_set_ghost_order) */
int ani_md_gather_rows(const double* src, const int64_t* idx, int n, double* out, void* stream);
int ani_md_scatter_rows(double* dst, const int64_t* idx, int n, const double* in, void* stream);

/*
 * Re-neighbouring of the stand-in loop, natively (the reference's runs leave this to LAMMPS: Domain::pbc, Comm::borders,
 * Neighbor::decide).
 *   ani_md_wrap_positions     x[i][k] -> [lo[k], lo[k] + len[k]) for the periodic dimensions (bit k of periodic_mask)
 *   ani_md_ghost_shell_count  which owned atoms are ghosts of which (brick, image) combination: atom a belongs to combination c
 *                             when clo[c][k] <= x[a][k] < chi[c][k] for k = 0..2.  Counts per (combination, block of 256 atoms) into
 *                             blk_cnt[ncombo * nblk], their exclusive scan into blk_off (nblk = ceil(n / 256), at least 1), and
 *                             out_counts[0] = all hits, out_counts[1 + c] = hits of combination c (device; the caller reads them)
 *   ani_md_ghost_shell_fill   the hits themselves, combination-major, atoms ascending: send_idx[p] = atom, send_shift[p] = cshift[c]
 *   ani_md_append_ghosts      one rank: x[nlocal + g] = x[owner[g]] + shift[g], species[nlocal + g] = species[owner[g]]
 *   ani_md_check              *out = *d2max (then *d2max = 0), or +inf when ev[0] is not finite
 */
int ani_md_wrap_positions(double* x, int n, const double* lo3, const double* len3, int periodic_mask, void* stream);
int ani_md_ghost_shell_count(const double* x, int n, const double* clo, const double* chi, int ncombo, int* blk_cnt, int* blk_off,
                             int* out_counts, void* stream);
int ani_md_ghost_shell_fill(const double* x, int n, const double* clo, const double* chi, const double* cshift, int ncombo,
                            const int* blk_off, int64_t* send_idx, double* send_shift, void* stream);
int ani_md_append_ghosts(double* x, int* species, int nlocal, const int64_t* owner, const double* shift, int nghost, void* stream);
int ani_md_check(double* d2max, const double* ev, double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ANI_MD_H */
