/*
 * mgrit_hip.h -- C ABI of libmgrit_hip.so, the MI355X (gfx950) MGRIT relaxation engine.
 *
 * The reference (PyMGRIT, pure Python) has no FFI; this ABI is the boundary a maintainer would bind with ctypes
 * to replace the per-time-point Python loops of src/pymgrit/core/mgrit.py. Every entry point
 * names the reference code it replaces (file:line under /root/reference).
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success and a negative code on error
 * (message via mgrit_hip_last_error()); no exceptions cross the ABI; one host thread per engine; all device work
 * is enqueued on the hipStream_t given at creation; state slabs are DEVICE pointers owned by the caller:
 * float64 [n_local_points][ld] with ld = mgrit_hip_row_stride(n); inside a row the n spatial values are stored in the
 * engine's lane-blocked order (natural index j lives at mgrit_hip_row_position(n, j); the other positions are padding
 * and must stay zero). Host arrays are borrowed for the duration of the call only.
 */
#ifndef MGRIT_HIP_H
#define MGRIT_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MGRIT_HIP_ABI_VERSION 3
#define MGRIT_HIP_E 16            /* elements per lane (arithmetic spec) */
#define MGRIT_HIP_MAX_N 16384     /* max DOFs per time point for the register-resident steppers */
#define MGRIT_HIP_MAX_N_WIDE 65536 /* Heat1D levels above MGRIT_HIP_MAX_N: the same Phi as three launches over rows in HBM
                                      (csrc/mgrit_hip_wide.inc): relaxations, residual / jump, the unfused FAS right-hand side,
                                      transfers, exchange; no fused passes, no AT-MGRIT */
#define MGRIT_HIP_MAX_N_2PTS 4096 /* two-point steppers: values per half of a pair that ONE workgroup holds (two coefficient sets in
                                      LDS); wider pairs, up to MGRIT_HIP_MAX_N_WIDE per half, take every half-solve as three
                                      launches over rows in HBM like wide Heat1D states (round 4; csrc/mgrit_hip_wide.inc, wide2_*) */
#define MGRIT_HIP_BLOCK_K 16      /* time-parallel forward solve: steps per block */
#define MGRIT_HIP_BLOCK_RMAX 256  /* ... and the most sine modes its recurrence over the blocks may need */

typedef struct mgrit_hip_engine mgrit_hip_engine;

enum { MGRIT_HIP_OK = 0, MGRIT_HIP_EINVAL = -1, MGRIT_HIP_EHIP = -2, MGRIT_HIP_ENODEV = -3, MGRIT_HIP_EUNSUPPORTED = -4 };
enum { MGRIT_HIP_STEPPER_HEAT1D = 1, MGRIT_HIP_STEPPER_ADVECTION1D = 2, MGRIT_HIP_STEPPER_HEAT2D = 3,
       MGRIT_HIP_STEPPER_HEAT1D_2PTS = 4 };
enum { MGRIT_HIP_TRANSFER_COPY = 0, MGRIT_HIP_TRANSFER_HEAT1D = 1, MGRIT_HIP_TRANSFER_PERIODIC1D = 2,
       /* the caller applies restriction / interpolation itself (a user's GridTransfer, reference core/grid_transfer.py:31-55:
          any Python code): mgrit_hip_restrict_u / _fas_rhs / _error_correction / _interpolate refuse the level pair, the FAS
          right-hand side is taken in two halves around the caller's restriction (mgrit_hip_fas_fine_rows / _fas_coarse) */
       MGRIT_HIP_TRANSFER_CALLER = 3 };
enum { MGRIT_HIP_RELAX_F = 0, MGRIT_HIP_RELAX_C = 1, MGRIT_HIP_RELAX_CHAIN = 2,
       /* f_relax + c_relax (mgrit.py:270-275) of a level > 0 in one pass, valid right after the finer level's FAS sweep has
          filled the level (u == v): a run = the F-points of an interval and the C-point closing it, started from v of the
          C-point in front of it; only the closing C-point is stored (1-D one-point steppers, weight_c = 1) */
       MGRIT_HIP_RELAX_FC = 3 };

int mgrit_hip_abi_version(void);
const char *mgrit_hip_last_error(void);
/* Row geometry of a slab (pure functions, no device needed): stride in doubles for n DOFs per time point, and the
 * position of natural spatial index j inside a row (-1 when out of range). */
int mgrit_hip_row_stride(int n);
int mgrit_hip_row_position(int n, int j);
/* number of visible HIP devices (0 when none); does not create a context */
int mgrit_hip_device_count(void);

/* Engine = per-rank solver state of Mgrit.__init__ (mgrit.py:145-222) minus the slabs. stream: hipStream_t. */
int mgrit_hip_create(mgrit_hip_engine **out, int n_levels, void *stream);
int mgrit_hip_destroy(mgrit_hip_engine *e);
int mgrit_hip_sync(mgrit_hip_engine *e);

/*
 * Level description. t_local: the rank-local time grid Mgrit.t[lvl] (mgrit.py:792, ghost point first when present).
 * Replaces Heat1D.__init__/compute_matrix/step (heat/heat_1d.py:138-217): Phi = (I + dt*L)^-1 (u + dt*b(x,t_stop)),
 * L = fac*tridiag(-1,2,-1), fac = a/dx^2; forcing b(x,t_i) = sum_k s[k][x]*tau[k][i] (K = 0: zero forcing).
 */
int mgrit_hip_level_heat1d(mgrit_hip_engine *e, int lvl, int n_pts_local, const double *t_local, int n, int ld,
                           double fac, int K, const double *s, const double *tau);
/* General (non-separable) forcing of a Heat1D level described with K = 0: rows = DEVICE slab [n_pts_local][ld] (row storage
 * order, caller-owned like the state slabs) with rows[i] = rhs(x, t_i) * (t_i - t_{i-1}), the product Heat1D.step adds to
 * u_start (heat/heat_1d.py:213: u_start + rhs(x, t_stop) * (t_stop - t_start)); Phi then computes (I + dt L)^-1 (u + rows[i]).
 * +8 B per DOF and Phi of HBM traffic. NULL switches it off again. */
int mgrit_hip_level_forcing_rows(mgrit_hip_engine *e, int lvl, const double *rows);
/* Replaces Heat1DBDF1.step / Heat1DBDF2.step (heat/heat_1d_2pts_bdf1.py:84-117, heat/heat_1d_2pts_bdf2.py:82-138): a state
 * is the pair (u(t), u(t + dtau)) of VectorHeat1D2Pts (heat/vector_heat_1d_2pts.py:9-140) stored as one slab row
 * [first | second], each half laid out like a Heat1D row of n values: ld = 2 * mgrit_hip_row_stride(n). One Phi = two
 * tridiagonal Toeplitz solves: order 1: backward Euler t_{i-1}+dtau -> t_i -> t_i+dtau; order 2: variable-step BDF2.
 * Forcing b(x,t) = sum_k s[k][n] * tau_k(t) with tau[k][i] = tau_k(t_i) and tau2[k][i] = tau_k(t_i + dtau). */
int mgrit_hip_level_heat1d_2pts(mgrit_hip_engine *e, int lvl, int n_pts_local, const double *t_local, int n, int ld,
                                double fac, double dtau, int order, int K, const double *s, const double *tau,
                                const double *tau2);
/* Replaces Advection1D.compute_matrix/step (advection/advection_1d.py:101-143): (I + dt*fac*(I - S_periodic)) x = u */
int mgrit_hip_level_advection1d(mgrit_hip_engine *e, int lvl, int n_pts_local, const double *t_local, int n, int ld,
                                double fac);
/* Replaces Heat2D.__init__/compute_matrix/compute_rhs/step (heat/heat_2d.py:147-366): theta-scheme (theta = 1 BE, 0.5 CN,
 * 0 FE) on the full nx x ny grid (row-major, x slow); rows of the slabs hold the grid in natural order, ld >= nx*ny a
 * multiple of 16. bc: nx*ny boundary values (zero inside, corner order of heat_2d.py:244-247); forcing on the interior
 * b(x,y,t_i) = sum_k S[k][(nx-2)*(ny-2)] * tau[k][i]. The implicit solve is a fast diagonalisation: four batched half-size
 * sine transforms (even/odd split of the sine matrix) on the FP64 matrix cores per step. */
int mgrit_hip_level_heat2d(mgrit_hip_engine *e, int lvl, int n_pts_local, const double *t_local, int nx, int ny, int ld,
                           double fx, double fy, double theta, const double *bc, int K, const double *S, const double *tau);
/* General forcing rhs(x, y, t) of a Heat2D level described with K = 0 (the reference takes any callable, heat/heat_2d.py:148,
 * 289-320, 346-356): rows = DEVICE slab [n_pts_local][Mi][Mj] (Mi, Mj = the interior nx-2, ny-2 padded as mgrit_hip_heat2d_padded
 * reports; pads zero), rows[i] = rhs(x, y, t_i) -- not multiplied by a step size: the theta-scheme weighs both ends of a step,
 * theta*dt*rhs(t_i) + (1-theta)*dt*rhs(t_{i-1}). Caller-owned like the state slabs; NULL switches it off. +8 B per DOF and Phi. */
int mgrit_hip_heat2d_padded(mgrit_hip_engine *e, int lvl, int *Mi, int *Mj);
int mgrit_hip_level_heat2d_forcing_rows(mgrit_hip_engine *e, int lvl, const double *rows);
/* Device slabs u, v, g of Mgrit.create_u_v_g (mgrit.py:840-858); v and g may be NULL on level 0. */
int mgrit_hip_level_bind(mgrit_hip_engine *e, int lvl, double *u, double *v, double *g);
/* Hand-over state of forward_solve (mgrit.py:459-486) between the owners of a level (op 5, mgrit.py:468-485): a level whose
 * forward solve runs in the overlapped form continues on the next rank bit for bit only if that rank gets,
 * besides the last point itself, the carry-free part of it and the carries of its step. *len_out = number of doubles of
 * that state (0: the level hands over the last point alone). The caller owns the buffer (mgrit_hip_chain_bind), sends it
 * behind the last point, and calls mgrit_hip_chain_resume(…, 1) after it has received one: the next CHAIN relax on the level
 * then starts from the state instead of from the ghost point alone (the flag clears itself). */
/* mgrit_hip_chain_enable(…, 0) keeps the level on the plain per-step forward solve: the engine sees only this rank's points,
 * but every owner of a level has to take the same form (and exchange the same hand-over), so the caller decides from the
 * level's GLOBAL time grid (one step size everywhere) and tells every rank the same. Default: enabled. */
int mgrit_hip_chain_enable(mgrit_hip_engine *e, int lvl, int on);
int mgrit_hip_chain_state_len(mgrit_hip_engine *e, int lvl, int *len_out);
int mgrit_hip_chain_bind(mgrit_hip_engine *e, int lvl, double *state);
int mgrit_hip_chain_resume(mgrit_hip_engine *e, int lvl, int on);
/* Time-parallel form of Mgrit.forward_solve (mgrit.py:459-486) on a Heat1D level > 0 (csrc/mgrit_hip_blk.inc): the
 * level's steps in blocks of MGRIT_HIP_BLOCK_K, every block stepped from a zero state, the block ends put right by a recurrence
 * over the blocks in the r lowest sine modes (the steps share their eigenvectors; over one block all other modes decay below
 * 2^-60), the block interiors stepped again from the corrected block starts. The same solve up to rounding; 2 Phi per step, all
 * blocks at once, instead of 1 Phi per step one after the other.
 * Advection1D levels (periodic upwind: circulant steps, Fourier modes, nothing decays) take the same scheme on ALL n modes, 64 <= n
 * <= 8192: through a radix-2 FFT of the block ends where n is a power of two, through the transforms as ordered sums on the matrix
 * cores for any other n; r = n, and the hand-over buffers hold n complex amplitudes (2 n doubles).
 * Heat2D levels (one rank): the full sine spectrum of the interior, backward Euler and Crank-Nicolson (the latter checks per solve
 * that the errors at the block ends have zero rims and steps through the level otherwise).
 *   mgrit_hip_block_solve_rank    the rule (pure host arithmetic) for a stepper kind MGRIT_HIP_STEPPER_HEAT1D / _ADVECTION1D:
 *                                 *r_out = modes needed for the time grid t[0..nt-1] (the level's
 *                                 GLOBAL grid when it is sharded: every owner must take the same form), 0 = the level is solved
 *                                 step by step (fewer than 4 blocks; Heat1D: more than MGRIT_HIP_BLOCK_RMAX modes; Advection1D: n
 *                                 outside [64, 8192]).
 *   mgrit_hip_block_solve_config  r > 0: the level's CHAIN relax over all its steps takes this form with r modes; r = 0: never;
 *                                 r = -1: the engine applies the rule to its local grid (one rank). first_real = 0: the rank has
 *                                 a predecessor -- its first block starts from zero too, and the recurrence starts from the
 *                                 amplitudes at the ghost point in uh_in; has_successor: the amplitudes at the last local point
 *                                 go to uh_out. Both are device buffers of MGRIT_HIP_BLOCK_RMAX doubles (Advection1D: 2 n) owned by the caller
 *                                 (the op-5 message of a sharded run carries them behind the point). A rank's share must be
 *                                 whole blocks: its first local slot a multiple of MGRIT_HIP_BLOCK_K steps from the start.
 *   mgrit_hip_block_solve         phases (bit mask) of the solve for callers that interleave them with the exchange:
 *                                 1 = first pass (needs nothing from the predecessor), 2 = recurrence over the blocks (needs
 *                                 uh_in) and, with a successor, the corrected last point, 4 = corrections + second pass (needs
 *                                 the ghost point). mgrit_hip_relax(CHAIN) over all steps = the three phases in a row.
 *   mgrit_hip_block_solve_state   *r_out = modes in effect on the level (0: step by step).
 *   mgrit_hip_block_solve_form    *form_out = MGRIT_HIP_BLOCK_FORM_STEPS (step by step), _PHASES (a launch or more per phase) or
 *                                 _ONE_LAUNCH: a small Heat1D level on one rank -- one group of values (n <= 1024), at most 128
 *                                 blocks and 64 modes -- whose three phases run as ONE launch with device-wide barriers between them
 *                                 (the launch gaps are most of the solve at that size; same bits; MGRIT_HIP_BLK_ONE=0 in the
 *                                 environment keeps the per-phase launches). */
#define MGRIT_HIP_BLOCK_FORM_STEPS 0
#define MGRIT_HIP_BLOCK_FORM_PHASES 1
#define MGRIT_HIP_BLOCK_FORM_ONE_LAUNCH 2
int mgrit_hip_block_solve_rank(int stepper, int n, double fac, int nt, const double *t, int *r_out);
int mgrit_hip_block_solve_config(mgrit_hip_engine *e, int lvl, int r, int first_real, int has_successor, double *uh_in,
                                 double *uh_out);
int mgrit_hip_block_solve(mgrit_hip_engine *e, int lvl, int phases);
int mgrit_hip_block_solve_state(mgrit_hip_engine *e, int lvl, int *r_out);
int mgrit_hip_block_solve_form(mgrit_hip_engine *e, int lvl, int *form_out);
/* Spatial transfer between lvl and lvl+1: GridTransferCopy (core/grid_transfer_copy.py:23-47) or the full-weighting
 * / linear-interpolation pair of examples/example_spatial_coarsening.py:33-82 (fine n = 2*coarse n + 1), or its periodic
 * analogue for Advection1D grids (fine n = 2*coarse n; no reference class exists, BASELINE config 5). */
int mgrit_hip_level_transfer(mgrit_hip_engine *e, int lvl, int kind);

/*
 * Run list: run r covers the consecutive local points start[r] .. start[r]+len[r]-1, each computed from its
 * predecessor (start[r]-1 must be a valid local index). F-intervals (mgrit.py:313-331), C-points (mgrit.py:355-368,
 * len 1 unless C-points are adjacent) and the coarsest-level chain (mgrit.py:472-481) are all run lists.
 */
int mgrit_hip_runs_create(mgrit_hip_engine *e, int lvl, int n_runs, const int32_t *start, const int32_t *len, int *id_out);
/* Pair list (fine local index on lvl, coarse local index on lvl+1) for the grid-transfer sweeps. */
int mgrit_hip_pairs_create(mgrit_hip_engine *e, int lvl, int n_pairs, const int32_t *fine_idx, const int32_t *coarse_idx,
                           int *id_out);

/* Mgrit.f_relax (mgrit.py:292-333), Mgrit.c_relax (mgrit.py:335-370), Mgrit.forward_solve (mgrit.py:459-486):
 * mode F / CHAIN:  u_i = [g_i +] Phi(u_{i-1})  (CHAIN = the sequential coarsest-level solve, same arithmetic);
 * mode C: u_i = ([g_i +] Phi(u_{i-1}))*w + u_i*(1-w). g is used on lvl > 0. */
int mgrit_hip_relax(mgrit_hip_engine *e, int lvl, int runs_id, int mode, double weight_c);
/* Mgrit.compute_residual (mgrit.py:387-413): per run r (len 1) sumsq_out[r] = ||Phi(u_{i-1}) - u_i||_2^2 (device ptr). */
int mgrit_hip_residual(mgrit_hip_engine *e, int lvl, int runs_id, double *sumsq_out);
/* Mgrit.compute_jump (mgrit.py:372-385): sumsq_out[r] = ||u_i - prev_i||_2^2, prev a device slab shaped like u. */
int mgrit_hip_jump(mgrit_hip_engine *e, int lvl, int runs_id, const double *prev, double *sumsq_out);

/* Mgrit.fas_residual (mgrit.py:488-549), split at its communication points:
 *   restrict_u: u^{l+1}_j = R(u^l_i) for every pair                      (mgrit.py:498-500)
 *   copy_u_to_v: v^{l+1} = clone(u^{l+1}) for the whole local slab         (mgrit.py:520)
 *   fas_rhs:    g^{l+1}_j = R(Phi_l(u_{i-1}) - u_i) + v_j - Phi_{l+1}(v_{j-1})            (lvl 0, mgrit.py:528-536)
 *               g^{l+1}_j = R(g_i - u_i + Phi_l(u_{i-1})) + v_j - Phi_{l+1}(v_{j-1})      (lvl>0, mgrit.py:538-547) */
int mgrit_hip_restrict_u(mgrit_hip_engine *e, int lvl, int pairs_id);
int mgrit_hip_copy_u_to_v(mgrit_hip_engine *e, int lvl_coarse);
int mgrit_hip_fas_rhs(mgrit_hip_engine *e, int lvl, int pairs_id);
/* The two halves of mgrit_hip_fas_rhs for a transfer the caller applies (MGRIT_HIP_TRANSFER_CALLER; 1-D steppers):
 *   mgrit_hip_fas_fine_rows: rows[p] (device, row stride ld_rows >= the level's row stride, engine row order) =
 *       Phi_l(u_{i-1}) - u_i  (lvl 0)  or  (g_i - u_i) + Phi_l(u_{i-1})  for the p-th pair's fine point i   (mgrit.py:528-532, 538-543)
 *   the caller restricts every row and stores the result in g^{l+1}_j of the pair's coarse point
 *   mgrit_hip_fas_coarse:    g^{l+1}_j = (g^{l+1}_j + v^{l+1}_j) - Phi_{l+1}(v^{l+1}_{j-1})                        (mgrit.py:533-536, 544-547) */
int mgrit_hip_fas_fine_rows(mgrit_hip_engine *e, int lvl, int pairs_id, double *rows, int ld_rows);
int mgrit_hip_fas_coarse(mgrit_hip_engine *e, int lvl, int pairs_id);
/* The same sweep fused into one pass for the identity transfer (GridTransferCopy) and like steppers on both levels:
 * triple = (fine slot i of a C-point, fine slot of the PREVIOUS C-point (must be local), coarse slot j). For every triple:
 * u^{l+1}_j = v^{l+1}_j = u^l_i and g^{l+1}_j as above with v_{j-1} read as u^l at the previous C-point. Pairs whose
 * predecessor lives on another rank go through restrict_u / copy_pairs_u_to_v / fas_rhs after the exchange. */
int mgrit_hip_triples_create(mgrit_hip_engine *e, int lvl, int n, const int32_t *fine_idx, const int32_t *prev_fine_idx,
                             const int32_t *coarse_idx, int *id_out);
int mgrit_hip_fas_fused(mgrit_hip_engine *e, int lvl, int triples_id);
/* mgrit_hip_fas_fused with options (Heat1D on both levels; 0 = mgrit_hip_fas_fused):
 *   MGRIT_HIP_FAS_WITH_F_RELAX: the F-relaxation that precedes the sweep in Mgrit.iteration (mgrit.py:275, or :271 when
 *       cf_iter = 0) is part of it: the F-points between the previous C-point and i are stepped through inside the pass,
 *       u_k = g_k + Phi(u_{k-1}), and NOT stored -- nothing reads a level's F-points between this sweep and the error
 *       correction + F-relaxation of the way up, which rewrites them. All points between the two C-points must be F-points.
 *   MGRIT_HIP_FAS_SKIP_COARSE_U: u^{l+1}_j is not stored (a coarsest level whose forward_solve, mgrit.py:459-486, overwrites
 *       every point but the first before anything reads it). */
enum { MGRIT_HIP_FAS_WITH_F_RELAX = 1, MGRIT_HIP_FAS_SKIP_COARSE_U = 2 };
int mgrit_hip_fas_fused_opts(mgrit_hip_engine *e, int lvl, int triples_id, int opts);
/* v^{l+1}_j = u^{l+1}_j for the coarse slots of a pair list (row-wise part of mgrit.py:520) */
int mgrit_hip_copy_pairs_u_to_v(mgrit_hip_engine *e, int lvl, int pairs_id);

/* Mgrit.error_correction (mgrit.py:715-726): u^l_i = u^l_i + P(u^{l+1}_j - v^{l+1}_j) */
int mgrit_hip_error_correction(mgrit_hip_engine *e, int lvl, int pairs_id);
/* Mgrit.nested_iteration interpolation (mgrit.py:559-563): u^l_i = P(u^{l+1}_j) */
int mgrit_hip_interpolate(mgrit_hip_engine *e, int lvl, int pairs_id);
/* AtMgrit.forward_solve on one rank (core/at_mgrit.py:79-87): every point p >= 1 of coarse level lvl becomes the value
 * obtained from the OLD u at point max(0, p-k+1) by the steps up to p, u_i = g_i + Phi(u_{i-1}); points are independent. Every
 * stepper: register-resident 1-D states and two-point pairs in one launch, Heat2D and the wide states as one batch of their Phi
 * launches per step distance (round 4). */
int mgrit_hip_at_solve(mgrit_hip_engine *e, int lvl, int k);
/* Mgrit.error_correction followed by Mgrit.f_relax (mgrit.py:715-726, then 292-333, as Mgrit.iteration calls them,
 * mgrit.py:283-284) in one pass, for 1-D steppers and the identity transfer: a run list whose run r additionally names the
 * coarse slot coarse_idx[r] of its predecessor C-point (-1: predecessor is a ghost or is not corrected). The predecessor
 * receives u += u^{l+1}_j - v^{l+1}_j and is written back, then the run's points follow as in mgrit_hip_relax mode F. */
int mgrit_hip_ec_runs_create(mgrit_hip_engine *e, int lvl, int n_runs, const int32_t *start, const int32_t *len,
                             const int32_t *coarse_idx, int *id_out);
int mgrit_hip_ec_relax(mgrit_hip_engine *e, int lvl, int ec_runs_id);

/* Whole-level sweeps in one pass for Heat1D with separable forcing on lvl and lvl+1, the identity transfer and lvl = 0.
 * Interval list: interval i runs from the C-point at fine slot cstart[i] to the next C-point at cend[i] (>= 1 F-point between
 * them); cend_coarse[i] / cstart_coarse[i] = their coarse slots, cstart_coarse[i] = -1 when the starting C-point takes no part
 * in the sweep (the first point of the time grid: neither relaxed nor corrected); res_pos[i] = position of the closing
 * C-point in the residual output of res_len values (a list may hold only some of the
 * level's intervals: one block of a planned cycle). Workgroups walk chunks of at most `chunk` consecutive intervals in time order.
 *   mgrit_hip_cf_fas:        Mgrit.c_relax, then Mgrit.f_relax, then Mgrit.fas_residual (mgrit.py:335-370, 292-333, 488-549 in
 *                            the order of Mgrit.iteration, mgrit.py:277-281; weight_c = 1) for the C-points the intervals end
 *                            on: they receive u_i = Phi(u_{i-1}), and u, v, g of lvl+1 the injected value and the FAS
 *                            right-hand side. The F-points of lvl are NOT written (nothing reads them before the next
 *                            error correction + F-relaxation rewrites them).
 *   mgrit_hip_ec_relax_res:  Mgrit.error_correction, then Mgrit.f_relax (mgrit.py:715-726, 292-333 as in mgrit.py:283-284), then
 *                            Mgrit.compute_residual (mgrit.py:387-413): ||Phi(u_{i-1}) - u_i||^2 of every closing C-point is
 *                            kept in pinned host memory; mgrit_hip_residual_fetch(e, n, out) waits for the sweep and copies
 *                            the n values (order res_pos) out. store_all_f = 0 writes, of every interval's F-points, only
 *                            the last one (which the next C-relaxation reads): nothing else of an MGRIT cycle reads an
 *                            F-point before the next F-relaxation rewrites it, and mgrit_hip_relax(mode F) over the level's
 *                            F-runs restores all of them bit for bit when the caller wants to look at the solution
 *                            (C-point storage, as in XBraid's default storage mode). store_all_f = 2: as 0, but the row
 *                            of that last F-point receives Phi(u_{i-1}) -- the value the residual needs anyway -- instead of
 *                            u_{i-1}: it is exactly what the C-relaxation of the next cycle assigns to the C-point
 *                            (mgrit.py:365 with weight 1), so mgrit_hip_cf_fas(..., pre_relaxed = 1) reads it and skips that
 *                            Phi. Any other reader of the level needs the F-relaxation first (it restores the row).
 * keep[i] (null: 3 everywhere) names the rows of lvl+1 that the closing C-point of interval i must receive from mgrit_hip_cf_fas:
 * bit 0 = u^{l+1} (not needed where the first sweep of lvl+1 overwrites it unread: its F-points when the level starts with an
 * F-relaxation, mgrit.py:270-271; every point but the first of a coarsest level solved by forward_solve, mgrit.py:459-486),
 * bit 1 = v^{l+1} (not needed when mgrit_hip_ec_relax_res performs the error correction: it takes the same bits from the fine
 * C-point; the library adds the bit where a chunk ends). Chunks are cut where res_pos is a multiple of `chunk`, so all lists
 * of a level are cut alike (chunk = 0: 1, 2 or 4 by res_len and the level's state width -- 1 while the level has fewer intervals
 * than the chip holds workgroups; chunk = MGRIT_HIP_CHUNK_LONG: the same rule up to 16 where a workgroup still gets four chunks,
 * for the lists of mgrit_hip_cf_fas / mgrit_hip_ec_relax_res -- a chunk's start costs the way down a row, the way up two);
 * the lists given to mgrit_hip_cf_fas and mgrit_hip_ec_relax_res within one cycle must be the same. */
#define MGRIT_HIP_CHUNK_LONG (-1)
int mgrit_hip_intervals_create(mgrit_hip_engine *e, int lvl, int n, const int32_t *cstart, const int32_t *cend,
                               const int32_t *cstart_coarse, const int32_t *cend_coarse, const int32_t *res_pos, int res_len,
                               int chunk, const int32_t *keep, int *id_out);
int mgrit_hip_cf_fas(mgrit_hip_engine *e, int lvl, int intervals_id, int pre_relaxed);
int mgrit_hip_ec_relax_res(mgrit_hip_engine *e, int lvl, int intervals_id, int store_all_f);
/* the same pass with the per-point sums of squares written to sumsq_out[res_pos] (device-accessible memory of the caller, e.g.
 * pinned host memory) instead of the engine's buffer: a rank of a sharded run fills the values of its complete intervals this
 * way and the first local C-point's through mgrit_hip_residual */
int mgrit_hip_ec_relax_res_to(mgrit_hip_engine *e, int lvl, int intervals_id, int store_all_f, double *sumsq_out);
int mgrit_hip_residual_fetch(mgrit_hip_engine *e, int n, double *sumsq_host);

/* The same two whole-level passes for ANY pair of register-resident 1-D steppers of one kind (Heat1D with any forcing,
 * Advection1D) joined by ANY of the library's transfers (copy, MGRIT_HIP_TRANSFER_HEAT1D, MGRIT_HIP_TRANSFER_PERIODIC1D;
 * examples/example_spatial_coarsening.py:33-82), on every level with a coarser one (BASELINE config 5: Advection1D, F-cycle,
 * spatial coarsening):
 *   mgrit_hip_gen_down:  Mgrit.c_relax, Mgrit.f_relax, Mgrit.fas_residual (mgrit.py:335-370, 292-333, 488-549 as in
 *                        mgrit.py:277-281; weight_c = 1) for the C-points the intervals end on, as two launches: the sweeps
 *                        of the fine level (F-points not stored) with the restriction of the C-point and of the defect row
 *                        r_i = Phi(u_{i-1}) - u_i [+ g_i] taken from the registers -- u^{l+1}_j = v^{l+1}_j = R(u_i),
 *                        g^{l+1}_j = R(r_i) + v^{l+1}_j --, and the coarse half g_j -= Phi_c(v_{j-1}). keep[] of the list is not
 *                        used: every row of lvl+1 is written.
 *   mgrit_hip_gen_up:    Mgrit.error_correction with the interpolation P(u^{l+1}_j - v^{l+1}_j) evaluated in registers
 *                        (mgrit.py:715-726), Mgrit.f_relax (every F-point stored) and, with_residual != 0 (level 0 only),
 *                        Mgrit.compute_residual (mgrit.py:387-413) into the engine's buffer (mgrit_hip_residual_fetch) or
 *                        sumsq_out when given: one launch. Needs the mgrit_hip_gen_down of the same list earlier in the cycle (it
 *                        takes the uncorrected value of the C-point a chunk starts from out of that pass's side slab).
 * Arithmetic: exactly that of the separate sweeps (same expressions, same order). */
int mgrit_hip_gen_down(mgrit_hip_engine *e, int lvl, int intervals_id);
/* the two halves apart (parts: 1 = the fine level's pass with the restriction, 2 = the coarse half, 3 = both): a rank of a sharded
 * run receives v^{l+1} of its ghost point -- the last point of the rank before -- between the two (op 4 of Mgrit.fas_residual,
 * mgrit.py:511-520) */
int mgrit_hip_gen_down_part(mgrit_hip_engine *e, int lvl, int intervals_id, int parts);
int mgrit_hip_gen_up(mgrit_hip_engine *e, int lvl, int intervals_id, int with_residual, double *sumsq_out);

/* Same two reductions with the result delivered to HOST memory (sumsq_host[r], r < n_runs) when the call returns: the
 * device->host leg of Mgrit.convergence_criterion (mgrit.py:425-432). */
int mgrit_hip_residual_host(mgrit_hip_engine *e, int lvl, int runs_id, double *sumsq_host);
int mgrit_hip_jump_host(mgrit_hip_engine *e, int lvl, int runs_id, const double *prev, double *sumsq_host);

/* Device time of the sweeps, measured with HIP events on the engine's stream around EVERY sweep entry point while timing is
 * enabled (mgrit_hip_set_timing(e, 1)): what the reference reports per sweep through logging.debug (mgrit.py:301-302,333,
 * 344,370,486,549). mgrit_hip_last_kernel_ms: the most recent timed call. mgrit_hip_timing_drain: all timed calls since the
 * last drain, in call order, as (MGRIT_HIP_T_* kind, level, milliseconds); waits for them to finish; at most max_records
 * are returned (the rest is dropped), *n_out = number returned. */
enum { MGRIT_HIP_T_RELAX_F = 0, MGRIT_HIP_T_RELAX_C = 1, MGRIT_HIP_T_CHAIN = 2, MGRIT_HIP_T_RESIDUAL = 3, MGRIT_HIP_T_JUMP = 4,
       MGRIT_HIP_T_RESTRICT = 5, MGRIT_HIP_T_COPY = 6, MGRIT_HIP_T_FAS_RHS = 7, MGRIT_HIP_T_FAS_FUSED = 8,
       MGRIT_HIP_T_ERROR_CORRECTION = 9, MGRIT_HIP_T_INTERPOLATE = 10, MGRIT_HIP_T_EC_RELAX = 11, MGRIT_HIP_T_AT = 12,
       MGRIT_HIP_T_CF_FAS = 13, MGRIT_HIP_T_EC_RELAX_RES = 14, MGRIT_HIP_T_RELAX_FC = 15, MGRIT_HIP_T_F_FAS = 16,
       MGRIT_HIP_T_EXCHANGE = 17, MGRIT_HIP_T_GEN_DOWN = 18, MGRIT_HIP_T_GEN_UP = 19, MGRIT_HIP_T_KINDS = 20 };
int mgrit_hip_set_timing(mgrit_hip_engine *e, int enabled);
int mgrit_hip_last_kernel_ms(mgrit_hip_engine *e, float *ms);
int mgrit_hip_timing_drain(mgrit_hip_engine *e, int max_records, int *kind, int *lvl, float *ms, int *n_out);

/* All device work of later calls goes to this hipStream_t (the stream given at creation until then). The caller orders
 * streams against each other (events): used to run the coarsest-level chain of one block of time points beside the
 * bandwidth-bound sweeps of other blocks (pymgrit_amd/core/cycle_plan.py). */
int mgrit_hip_set_stream(mgrit_hip_engine *e, void *stream);
/* Diagnostics of the most recent overlapped chain launch that has finished: shader clock (MHz, cycle counter against the
 * constant 100 MHz counter) and microseconds per step as seen by worker 0. Zeros when there is none. */
int mgrit_hip_chain_clock(mgrit_hip_engine *e, double *mhz, double *us_per_step);
/* n_cus > 0: the sweeps of later calls leave n_cus CUs of XCD 0 free (their workgroups draw items from a queue; of those that
 * land on XCD 0 only 32 - n_cus per CU-slot stay) and the chain launches take their workers from XCD 0: the two run side by
 * side on two streams. 0 (default): every sweep fills the chip, the chain's workers are the blocks with blockIdx % 8 == 0. */
int mgrit_hip_set_reserve(mgrit_hip_engine *e, int n_cus);

/*
 * Ghost exchange between the owners of neighbouring time points: Mgrit.send / Mgrit.receive (mgrit.py:693-713, pickled mpi4py
 * isend + blocking recv matched by tag) and their call sites -- f_relax ops 0 / 1 (mgrit.py:306,310,317,331), c_relax op 2
 * (mgrit.py:348,352), compute_residual op 7 (mgrit.py:399,403), forward_solve op 5 (mgrit.py:469,484), fas_residual ops 3 / 4
 * (mgrit.py:504-517). Here a message is a whole slab row moved by a STREAM operation of the engine (on the stream of
 * mgrit_hip_set_stream, like every sweep): ordered against the kernels that produce and consume the row by the stream
 * alone, one host call, capturable into a hipGraph with the sweeps around it. Messages of one link are matched by their
 * order, which is the order of the exchange points of Mgrit.iteration -- the same on both owners.
 *
 * A link (handle 0 .. MGRIT_HIP_MAX_LINKS-1, chosen by the caller) is ONE direction of one pair of ranks:
 *   mgrit_hip_link_attach   an RCCL communicator (ncclSend / ncclRecv over xGMI on an MI355X node); peer = the other rank in
 *                           it. Either any ncclComm_t the caller already has, or one made here: mgrit_hip_comm_unique_id on
 *                           one rank (128 bytes, passed to the others by the caller), mgrit_hip_comm_init_rank on every rank
 *                           of it (= ncclCommInitRank, collective), mgrit_hip_comm_destroy at the end (abort != 0:
 *                           ncclCommAbort). pymgrit_amd makes a two-rank communicator per link and channel, so that a
 *                           communicator is only ever used from one stream at a time and a send waiting for its receiver never
 *                           holds back a receive from another rank. The caller keeps ownership of attached communicators.
 *   mgrit_hip_link_mailbox  all ranks in ONE process on ONE GPU (tests, bench.py --emulate-rank): both ends name the same
 *                           mailbox (mgrit_hip_mailbox_create: n_slots rows of slot_doubles doubles in device memory); a send
 *                           copies the row into slot `slot`, a receive copies it out -- who may touch which slot when is the
 *                           caller's hand-shake (pymgrit_amd/core/comm.py, LoopbackComm). RCCL links ignore `slot`.
 * librccl is loaded on first use (no link-time dependency); without it the RCCL entry points return MGRIT_HIP_EUNSUPPORTED.
 *
 * mgrit_hip_exchange: ONE exchange point of operation op (0-5, 7) on level lvl: send row send_idx of u^lvl over send_link
 * (send_link < 0: nothing to send), then receive row recv_idx over recv_link (< 0: nothing). Operation 5 (the hand-over of
 * forward_solve) with handover = mgrit_hip_chain_state_len(lvl) > 0 moves the chain's running state behind the point and
 * arms mgrit_hip_chain_resume on the receiving side. mgrit_hip_send / mgrit_hip_recv: the same for count doubles at any
 * device address (rows the caller has staged, e.g. by mgrit_hip_error_correction_to).
 * mgrit_hip_sync_bounded: mgrit_hip_sync that gives up after timeout_s seconds -- a peer that never sends or never receives
 * must end in an error of this rank, not in a silent stall: the links are aborted (ncclCommAbort) and an error is returned.
 */
#define MGRIT_HIP_MAX_LINKS 16
int mgrit_hip_comm_unique_id(void *id_out_128_bytes);
int mgrit_hip_comm_init_rank(void **nccl_comm_out, const void *unique_id_128_bytes, int nranks, int rank);
int mgrit_hip_comm_destroy(void *nccl_comm, int abort);
int mgrit_hip_link_attach(mgrit_hip_engine *e, int link, void *nccl_comm, int peer);
int mgrit_hip_mailbox_create(void **mailbox_out, int n_slots, int slot_doubles);
int mgrit_hip_mailbox_destroy(void *mailbox);
int mgrit_hip_link_mailbox(mgrit_hip_engine *e, int link, void *mailbox);
int mgrit_hip_link_stats(mgrit_hip_engine *e, int link, uint64_t *messages_sent, uint64_t *bytes_sent, uint64_t *messages_received);
int mgrit_hip_links_close(mgrit_hip_engine *e, int abort);
int mgrit_hip_exchange(mgrit_hip_engine *e, int lvl, int op, int send_link, int send_idx, int send_slot, int recv_link,
                       int recv_idx, int recv_slot, int handover);
int mgrit_hip_send(mgrit_hip_engine *e, int link, int slot, const double *rows, int count);
int mgrit_hip_recv(mgrit_hip_engine *e, int link, int slot, double *rows, int count);
int mgrit_hip_sync_bounded(mgrit_hip_engine *e, double timeout_s);
/* Mgrit.error_correction (mgrit.py:715-726) of the pairs with the corrected rows written to rows_out[p] (device, row stride
 * ld_out >= the level's) instead of back into u^l: a rank sends the corrected value of its LAST C-point to the next owner
 * (op 0 of the F-relaxation that follows, mgrit.py:306) before the whole-level pass that corrects it in place has run. */
int mgrit_hip_error_correction_to(mgrit_hip_engine *e, int lvl, int pairs_id, double *rows_out, int ld_out);
/* copies the n per-point sums of squares of the last mgrit_hip_ec_relax_res pass (the engine's pinned buffer) to dst
 * (device-accessible, e.g. pinned host memory) on the stream: a replayed cycle always writes the engine's buffer, a solver
 * that looks at its stopping values some cycles late (Mgrit._solve_pipelined) keeps each cycle's values in a slot of its own */
int mgrit_hip_residual_stash(mgrit_hip_engine *e, int n, double *dst);
/* A HIP stream whose kernels run on CUs [first_cu, first_cu + n_cus) only (hipExtStreamCreateWithCUMask); for mgrit_hip_set_stream.
 * A planned cycle (pymgrit_amd/core/cycle_plan.py) of Heat2D levels gives the sequential coarsest-level solve -- six small
 * launches per step -- a few CUs of its own and the batched sweeps the rest: side by side the sweeps' thousands of workgroups
 * would otherwise sit in front of every one of its launches. No reference counterpart. */
int mgrit_hip_stream_create_masked(void **stream_out, int first_cu, int n_cus);
int mgrit_hip_stream_destroy(void *stream);
/* C-point mirror: from the next level-0 mgrit_hip_ec_relax_res pass on (stream order), every corrected C-point u^0_i (mgrit.py:
 * 724-726) is ALSO stored to row row0 + res_pos of the device slab `rows` ([>= row0 + res_len][ld]); NULL switches it off. A
 * solver that examines its stopping value some cycles late keeps the level-0 C-points of every cycle it may have to return to
 * (Mgrit._solve_pipelined): the pass that writes them writes the copy too, instead of a gather over the level afterwards. The
 * slab is named through a word in device memory that a one-thread kernel on the stream updates, so a captured cycle follows
 * it; row0 is fixed with the first call. */
int mgrit_hip_cpoint_mirror(mgrit_hip_engine *e, double *rows, int row0);

#ifdef __cplusplus
}
#endif
#endif
