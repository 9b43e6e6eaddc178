/*
 * ek_hip_debug.h -- tuning, profiling and test hooks of libek_hip.so.  NOT part of the drop-in
 * boundary (include/ek_hip.h): nothing a host of the reference's shape needs is declared here, and
 * these entries may change between rounds.  Used by bench.py (roofline instrumentation), tools/ and
 * the GPU test-suite (stage-level checks of the two-stage tridiagonalisation).
 */
#ifndef EK_HIP_DEBUG_H
#define EK_HIP_DEBUG_H

#ifdef __cplusplus
extern "C" {
#endif

/* Instrumentation for the roofline line of bench.py: enable = k > 0 brackets the launch of the
 * HBM-bound symv kernel of every k-th Householder column (one-stage path; k = 1: every launch; a
 * uniform sample over the trailing orders otherwise) by HIP events on its own stream; 0 switches it
 * off.  _get returns the accumulated device seconds, the number of timed launches and their
 * algorithmic bytes (8 B x lower triangle of the active matrix per launch). */
int ek_hip_profile_symv(int enable);
int ek_hip_profile_symv_get(double *seconds, long long *launches, double *algorithmic_bytes);

/* The same for the kernels of the two-stage path: seconds[i], launches[i] with i = 0 q2_apply_kernel
 * (application of the bulge-chasing reflectors, the largest kernel of a full-spectrum solve),
 * 1 chase_kernel, 2 symm_lower_kernel (every 8th panel), 3 unused. */
int ek_hip_profile_kernels(int enable);
int ek_hip_profile_kernels_get(double *seconds /* 4 */, long long *launches /* 4 */);

/* One-stage tridiagonalisation: tridiagonalise a device-generated synthetic matrix (order n, leading
 * dimension ld >= n rounded up to 128) `reps` times; *seconds = stage time per repetition. */
int ek_hip_debug_sytrd(int n, int ld, int reps, double *seconds);
unsigned long long ek_hip_debug_sytrd_work_bytes(int n);
int ek_hip_debug_sytrd_at(int n, int max_cols, int reps, double *dA, void *work, double *vecs, double *seconds);
int ek_hip_debug_gemm_at(int transa, int transb, int m, int n, int k, const double *dA, int lda, const double *dB,
                         int ldb, double beta, double *dC, int ldc, int lower_only, int reps, double *seconds);
int ek_hip_debug_sytrd_split(void *alt, int mask);  /* placement experiments: sub-buffers of the scratch from alt */
int ek_hip_debug_set_sytrd_maxcols(int max_cols);   /* the hooks stop after max_cols columns (-1: all) */
int ek_hip_debug_sytrd_team(int n, int nteam, int reps, double *seconds);
int ek_hip_debug_reduce_team(int n, int nteam, int reps, double *seconds /* [2]: potrf, sygst */);

/* Two-stage tridiagonalisation, piece by piece on host arrays.
 *   _sy2sb : A (n x n, lda, symmetric, lower referenced) -> band (half bandwidth 64) left in the lower
 *            band of A, explicit reflectors V (n x n, ldv: column j = v_j, unit entry at row j + 64),
 *            tau (n); *flag: low byte 0 (a panel CholeskyQR2 cannot factor -- rank deficient, cond > 1e7 -- is factored
 *            by Householder reflections inside the stage), bits 8.. = the number of panels that took that rescue.
 *   _sb2st : the lower band of A -> d (n), e (n-1) by bulge chasing; Z (n x ncols, ldz; may be NULL
 *            with ncols = 0) <- Q2 Z; *flag bit 2 = the persistent kernel was abandoned.
 *   _two_stage_timing : seconds[0..3] = dense->band, band->tridiagonal, Q2 applied to ncols columns,
 *            Q1 applied, on a device-generated synthetic matrix.
 *   _set_two_stage : order from which the whole-path calls use two stages (-1 default, 0 never). */
int ek_hip_debug_sy2sb(int n, double *A, int lda, double *V, int ldv, double *tau, int *flag);
int ek_hip_debug_sb2st(int n, const double *A, int lda, double *d, double *e, double *Z, int ldz, int ncols, int *flag);
int ek_hip_debug_two_stage_timing(int n, int ncols, int reps, double *seconds, int *flag);
int ek_hip_debug_set_two_stage(int min_order);
/*   _sy2sb_team : the first stage over a 1 x P team (128-wide column strips, strip S on rank S mod P; per panel one
 *            broadcast of [V | T | tau] and one all-reduce of Y): nteam >= 1 rehearses the whole team inside this
 *            process (every member with its own copy of A -- NaN outside its strips when EK_HIP_TEAM_POISON=1), nteam = 0
 *            makes this process one rank of the attached communicator.  Out: the gathered band in the lower band of A
 *            (zero elsewhere), V and tau of member 0; *mismatch = entries in which the members' bands, V or tau differ. */
int ek_hip_debug_sy2sb_team(int n, double *A, int lda, double *V, int ldv, double *tau, int nteam, int *flag,
                            long long *mismatch);
int ek_hip_debug_sy2sb_team_timing(int n, int nteam, int reps, double *seconds);   /* device-generated matrix; whole team */
/* the same with the look-ahead threshold of the team form set (0: none, -1: default) and, optionally, HIP-event sums of
   the panel chains and of the "rest of the update" sections: parts[0..3] = stage, chains, rest-updates, and
   first chain + sum over panels of max(chain p + 1, update p / P) (seconds; meaningful with lookahead_min = 0) */
int ek_hip_debug_sy2sb_team_profile(int n, int nteam, int reps, int lookahead_min, double *seconds, double *parts);

/* Counters of this process's last whole-path solve: out[0] = flops the merge products of the divide & conquer
 * executed (2 M N K over both GEMMs of every merge, with the dimensions deflation and the column selection left
 * on the device: what bench.py prices that stage with, the nominal 4 n^3 / 3 being an upper bound), out[1] = 1
 * if the tridiagonalisation ran in two stages, out[2] = panels of the dense -> band stage that CholeskyQR2 could not
 * factor and the Householder rescue did, out[3] = 1 if the matrix was already a band of half width 64 on entry and
 * the dense -> band stage (and Q1) were skipped. */
int ek_hip_debug_last_solve_stats(double *out, int count);

/* workspace one whole-path call asks for (host arithmetic, no GPU): one GPU (nranks <= 1) or rank 0 of a 1 x nranks team;
   parts[0..5] (optional): one padded matrix, the persistent operators (L, Q1's reflectors), the eigenvector columns,
   X0 (the matrix, then the bulge chasing's reflectors), X1 (scratch of the reduction -> D&C bases -> Q2's records -> Q1's
   scratch), the rest.  ek_solve.hip plan_path. */
unsigned long long ek_hip_debug_workspace_bytes(int problem, int n, int n_vec, int nranks, unsigned long long *parts);

/* test aid: the next `times` bulge chasings of whole-path calls count as abandoned (exercises the repetition from the
   saved band and the -992 exit of ek_solve.hip) */
int ek_hip_debug_fail_next_chase(int times);

/* Rehearsal of the divide & conquer's team form on one GPU (ek_stedc.hip, StedcTeam): while nranks >= 2, a grid cell solved
   WITHOUT a communicator (ek_hip_solve_device_grid / ek_hip_solve_replicated) forms the heights below the top merge that a
   team of nranks shards by strips rank after rank in this process (no exchange; same bits as any other form).  levels:
   sharded heights (-1: the library's default for the order and team); profile != 0: HIP events around the call and every
   rank's sections.  ek_hip_debug_stedc_team(0, -1, 0) switches it off.
   _get: seconds[0] the D&C, [1] all ranks' sections, [2] the longest rank's section summed over the heights -- a rank of
   a real team computes for [0] - [1] + [2] seconds (the all-gathers are not in it: one GPU). */
int ek_hip_debug_stedc_team(int nranks, int levels, int profile);
int ek_hip_debug_stedc_team_get(double *seconds);

/* Team Cholesky (ek_chol.hip potrf_lower_dist): look-ahead 0 off / 1 on / -1 default; profile != 0: HIP events around every
   owner's chain and every rest-of-update section of the next rehearsals (meaningful with the look-ahead OFF).  _get (after the
   rehearsal): parts[0] all chains, [1] all update sections, [2] first chain + sum over the strips of max(next chain, update / nteam):
   what the two cost a rank of a real team with the look-ahead. */
int ek_hip_debug_potrf_team_profile(int lookahead, int profile);
int ek_hip_debug_potrf_team_profile_get(int nteam, double *parts);

/* what the staging pipeline of the last ek_hip_solve on host arrays did: out[0] bytes in, [1] span of the input transfers
   (s), [2] busy seconds of the input workers, [3..5] the same on the way out, [6] seconds the main thread waited for
   inputs, [7] for the drain at the end, [8] 100 x workers on the way in + workers on the way out, [9] 0 (round 4: directions through a pinned ring, removed), [10] seconds from the start of the pipeline to its end, [11] seconds before the first input transfer */
int ek_hip_debug_last_pipe_stats(double *out, int count);

#ifdef __cplusplus
}
#endif
#endif /* EK_HIP_DEBUG_H */
