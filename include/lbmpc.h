/*
 * lbmpc.h -- C ABI of liblbmpc_hip.so: the MI355X (gfx950) data-parallel core of learning-based
 * rigid-tube MPC, a drop-in for the MATLAB entry points of gabormarko/learning-based-mpc.
 *
 * The reference has no FFI of its own (it is MATLAB calling MPT3 / YALMIP / CasADi+IPOPT); the
 * boundary kept is the MATLAB signatures.  Each entry point below cites the
 * reference code it replaces (paths relative to the reference root).  A MEX gateway
 * (learning-based-mpc_amd/matlab/lbmpc_mex.cpp) and the ctypes mirror
 * (learning-based-mpc_amd/lbmpc/) bind exactly these symbols.
 *
 * Conventions
 *   - all reals are IEEE fp64; all matrices are COLUMN-MAJOR (MATLAB layout), leading dim = rows.
 *   - "host" entry points take host pointers and copy in/out; "_dev" entry points take device
 *     pointers for the bulk operands (samples, directions, outputs) and a hipStream_t (as void*),
 *     enqueue asynchronously and do not synchronise the device.  Small constant operands (A, W_A,
 *     ...) are always host pointers.  Two of them may ALLOCATE device memory (hipMalloc, which is not
 *     legal under stream capture): lbmpc_support_batch_dev the first time it sees a parameter set
 *     (A, V_W, s, transpose) on a device (kept in a small per-process cache afterwards), and
 *     lbmpc_mc_run_dev when a call needs more scratch than any earlier call on the same context
 *     (grow-only).  lbmpc_halfspace_max_noalloc is the entry point that never allocates.
 *   - return value: 0 ok; <0 usage error (LBMPC_ERR_*); >0 numerical status (LBMPC_ST_*).
 *     No exceptions cross the ABI.  The reference has no status at all (IPOPT fails silently,
 *     Functions_general/UQRobustMPC.m:156-158).
 *   - every entry point needs a gfx950 device; there is no CPU fallback.
 */
#ifndef LBMPC_H
#define LBMPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LBMPC_VERSION 111 /* 0.1.1 */

/* usage errors */
#define LBMPC_ERR_ARG (-1)         /* null pointer / negative size / inconsistent dims      */
#define LBMPC_ERR_DIM (-2)         /* dimension beyond the compiled limits below            */
#define LBMPC_ERR_HIP (-3)         /* a HIP runtime call failed (see lbmpc_last_error)      */
#define LBMPC_ERR_UNSUPPORTED (-4) /* e.g. an option the build or the kernel shape lacks      */
#define LBMPC_ERR_NODEVICE (-5)    /* no gfx950 device visible                              */

/* numerical statuses (also the per-trial / per-problem status words) */
#define LBMPC_ST_OK 0
#define LBMPC_ST_QP_INFEASIBLE 1 /* tube QP has no feasible point                          */
#define LBMPC_ST_LP_INFEASIBLE 2 /* a sample lies outside the prior W (beta < 0 needed)    */
#define LBMPC_ST_ITER_CAP 3      /* active-set iteration cap reached                       */
#define LBMPC_ST_SAMPLER_CAP 4   /* rejection sampler hit max attempts                     */
#define LBMPC_ST_LP_UNBOUNDED 5  /* support LP unbounded in the asked direction            */

/* compiled limits */
#define LBMPC_MAX_NX 4
#define LBMPC_MAX_NZ 16 /* nx + N*nu */
#define LBMPC_MAX_NC 16
#define LBMPC_MAX_NW 16
#define LBMPC_MAX_NXI 16 /* rows of Xi_true_EV / Xi_true_LV H-rep */
#define LBMPC_MAX_NU 4  /* inputs (subsystem 3) */
#define LBMPC_MAX_NV 1024 /* vertices of the V-rep in the support sweep; also s*nV <= 2048 */
#define LBMPC_MAX_S 64   /* terms of the Minkowski sum            */

int lbmpc_version(void);
/* Validate a device (it must be gfx950) before anything else runs on it.  device >= 0: select it for the calling
 * thread (hipSetDevice) and validate it; LBMPC_DEVICE_CURRENT: validate the calling thread's current device without
 * changing the selection (one process per GPU under a launcher that has already bound the device).  Call it once per
 * device used.  Entry points without a context run on the calling thread's current device; entry points that take
 * a context run on the device the context was created on, whatever the caller has selected (the caller's selection
 * is restored on return). */
#define LBMPC_DEVICE_CURRENT (-1)
int lbmpc_init(int device);
/* Text of the last HIP/usage error on this thread (static storage). */
const char* lbmpc_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Subsystem 1: support functions for the mRPI set / tube tightening
 * ---------------------------------------------------------------------------------------- */

/* h_out[i] = (1-alpha)^-1 * sum_{t=0}^{s-1} h_W(P_t d_i),  h_W(p) = max_v Vw[v,:].p,
 * P_t = (A^t)' if transpose_flag else A^t.  D is nx x nd (one direction per column).
 * s=1, alpha=0 gives plain h_W(d).
 * Replaces: MPT3 `W.support(dirs)` as used by Functions_general/MRPISet.m:29,45 and the
 * support LPs of Functions_general/UQRobustMPC.m:109-122 (Com_hs) /
 * Functions_general/ComputeFeasibleRegion.m:447-503, for S = (1-alpha)^-1 (+)_{t<s} A^t W. */
int lbmpc_support_batch(const double* A, int nx, const double* Vw, int nV, int s, double alpha,
                        const double* D, int64_t nd, int transpose_flag, double* h_out);
int lbmpc_support_batch_dev(const double* A, int nx, const double* Vw, int nV, int s, double alpha,
                            const double* D_dev, int64_t nd, int transpose_flag, double* h_out_dev,
                            void* stream);

/* The (s, alpha) search of Functions_general/MRPISet.m:27-63 (mode 0: Ms = max over t=1..s and
 * d in Dirs of h_W(A^t d)), of Functions/MRPISet.m:8-17 (mode 1: Ms = max_d sum_{t=1..s}
 * h_W(A^t d)), or the exact Rakovic form (mode 2: alpha = max_j h_W((A^s)' f_j)/g_j,
 * Ms = max_d sum_{t=0}^{s-1} h_W((A^t)' d)).  Dirs is nx x ndirs (column = direction), WA is
 * nW x nx, Wb nW.  Loop: s = 0; while alpha > eps/(eps+Ms): s++ ... (both start at 1000). */
#define LBMPC_MS_MAX_LITERAL 0
#define LBMPC_MS_SUM_LITERAL 1
#define LBMPC_MS_RAKOVIC 2
int lbmpc_mrpi_s_alpha(const double* A, int nx, const double* Vw, int nV, const double* WA,
                       const double* Wb, int nW, const double* Dirs, int ndirs, double eps, int mode,
                       int max_s, int* s_out, double* alpha_out, double* Ms_out);

/* ------------------------------------------------------------------------------------------
 * Subsystem 2: homothetic disturbance-set learning over the information set I_0^w
 * ---------------------------------------------------------------------------------------- */

/* m_out[j + nW*b] = max_i WA[j,:] . samples[:, i] over the N samples of information set b.
 * samples: `batch` blocks, block b starts at samples + b*batch_stride (in doubles), each block
 * is nx x N column-major (one sample per column).  N >= 1.
 * Replaces: the nW*N `W_A*(w_i - y) <= (1-beta) W_b` constraint rows of
 * Functions_general/InitialSetComputation.m:77-79 / :93-95 (they enter the LP only through
 * this maximum) and the sample row of Functions_general/UQRobustMPC.m:135. */
int lbmpc_halfspace_max(const double* WA, int nW, int nx, const double* samples, int64_t N,
                        int64_t batch, int64_t batch_stride, double* m_out);
int lbmpc_halfspace_max_dev(const double* WA, int nW, int nx, const double* samples_dev, int64_t N,
                            int64_t batch, int64_t batch_stride, double* m_out_dev, void* stream);

/* The same reduction for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc): device pointers and a stream as lbmpc_halfspace_max_dev, but the partial maxima of sets
 * that are split over several waves live in a static device array of LBMPC_NOALLOC_MAX_PARTIALS doubles (2 MiB,
 * held by every GPU the library runs on until the process ends) instead of a hipMallocAsync buffer.  No counterpart
 * in the reference (MATLAB has no such constraint).  Calls no hipMalloc / hipFree / allocating routine.
 * The library splits a set over `chunks` waves when batch < 16384 (chunks = min(ceil(16384 / batch),
 * ceil(N / (256 * spl))), spl = 2 for nx in {1, 3} else 1); a call with chunks > 1 and batch * chunks * nW >
 * LBMPC_NOALLOC_MAX_PARTIALS returns LBMPC_ERR_DIM (split the batch).  Results are bit-identical to
 * lbmpc_halfspace_max_dev.  All calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams,
 * two threads) race; serialise them. */
#define LBMPC_NOALLOC_MAX_PARTIALS 262144
int lbmpc_halfspace_max_noalloc(const double* WA, int nW, int nx, const double* samples_dev, int64_t N,
                                int64_t batch, int64_t batch_stride, double* m_out_dev, void* stream);

/* Sampler description: w = xi_EV - xi_LV - B*u_LV with xi_* uniform on {x: Xi_A x <= Xi_b} by
 * rejection from the bounding box [lo, hi] of the vertices, u_LV ~ U[min_u, max_u].
 * Optional oriented proposal box: with EV_U / LV_U (nx x nx, column-major, invertible) the candidate is x = U t,
 * t uniform in [lo, hi] (lo, hi then bound the vertices in the coordinates t = U^-1 x); NULL means U = I, the
 * reference's axis-aligned box.  Either way the accepted samples are uniform on the polytope; a box that fits a
 * rotated polytope only needs fewer attempts.
 * Replaces Functions_general/InitialSetComputation.m:50-53,101-130 and
 * Functions_general/ModelingCar.m:34-91.  The reference uses MATLAB's unseeded global `rand`;
 * this library defines its own counter-based stream (Philox4x32-10, DESIGN.md "RNG"). */
typedef struct lbmpc_sampler {
    int nx, nu;
    int nEV, nLV;              /* rows of the two H-reps                                  */
    const double* EV_A;        /* nEV x nx */
    const double* EV_b;        /* nEV      */
    const double* EV_lo;       /* nx  bounding box of Xi_true_EV.V                        */
    const double* EV_hi;       /* nx                                                      */
    const double* LV_A;        /* nLV x nx */
    const double* LV_b;
    const double* LV_lo;
    const double* LV_hi;
    const double* B;           /* nx x nu  */
    const double* min_u_LV;    /* nu       */
    const double* max_u_LV;    /* nu       */
    const double* EV_U;        /* nx x nx or NULL: proposal basis of Xi_EV (see above)    */
    const double* LV_U;        /* nx x nx or NULL                                         */
} lbmpc_sampler;

/* Draw `batch` information sets of N samples each (set b uses RNG stream (seed, set0 + b)),
 * write them to samples_out (nx x N per set, sets contiguous; may be NULL to skip the store)
 * and their halfspace maxima to m_out (nW x batch).  status_out (batch ints, may be NULL) gets
 * LBMPC_ST_SAMPLER_CAP where a rejection loop hit `max_attempts`.
 * Replaces the sampling loop + constraint build of InitialSetComputation.m:48-54. */
int lbmpc_sample_information_sets(const lbmpc_sampler* smp, const double* WA, int nW, int64_t N,
                                  int64_t batch, uint64_t seed, uint64_t set0, int max_attempts,
                                  double* samples_out, double* m_out, int* status_out);
int lbmpc_sample_information_sets_dev(const lbmpc_sampler* smp, const double* WA, int nW, int64_t N,
                                      int64_t batch, uint64_t seed, uint64_t set0, int max_attempts,
                                      double* samples_out_dev, double* m_out_dev, int* status_out_dev,
                                      void* stream);

/* n points uniform on {x : XA x <= Xb} (XA: nr x nx) by rejection from the box [lo, hi]: point i is draw
 * (seed, stream, index0 + i, purpose) of the library's Philox stream, `purpose` in 0..15 selecting an independent
 * sub-stream.  out: nx x n.  status_out (n ints, may be NULL): LBMPC_ST_SAMPLER_CAP where max_attempts was hit.
 * Replaces polytope_sample_EV/_LV of Functions_general/ModelingCar.m:47-91 and
 * InitialSetComputation.m:101-130 when the caller drives the plant itself (Case_3...:58,75). */
int lbmpc_sample_polytope(const double* XA, const double* Xb, int nr, int nx, const double* lo, const double* hi,
                          int64_t n, uint64_t seed, uint64_t stream, uint32_t index0, int purpose, int max_attempts,
                          double* out, int* status_out);

/* The homothety LP  max beta  s.t. 0<=beta<=1, WA y <= beta Wb, beta Wb - WA y <= Wb - m
 * (alpha = 1-beta, v = y/beta), one problem per column of m (nW x batch).
 * Replaces Functions_general/InitialSetComputation.m:69-83,:85-99 (LP, LP_opt) and
 * Functions_general/UQRobustMPC.m:124-139 (LPforUQ, with m = max(b_hat_prev, WA w_new)).
 * alpha is unique; y is not (SURVEY.md hard part 1): tiebreak LBMPC_TIE_CENTER returns, for an axis-aligned box W
 * (every reference configuration), the midpoint of each coordinate's optimal interval (closed form), and for any other
 * bounded polytope W (nx+1 <= nW <= 16) the mean of the feasible basic solutions of the optimal face (vertex
 * enumeration, one lane per LP: C(2nW+2, nx+1) small solves; the closed loop uses the same routine for the steps in
 * which the new disturbance falls outside the current set, see lbmpc_uqmpc_create); the two coincide on a box.
 * active_mask (batch x uint32, may be NULL): box W: bit k set if axis k attains alpha (k < nx); general W: bit j set if
 * row j of W binds on either side.
 * status (batch ints, may be NULL): LBMPC_ST_LP_INFEASIBLE where the LP has no feasible point (some m_j > Wb_j). */
#define LBMPC_TIE_CENTER 1
int lbmpc_homothety_lp(const double* WA, const double* Wb, int nW, int nx, const double* m,
                       int64_t batch, int tiebreak, double* alpha, double* v, uint32_t* active_mask,
                       int* status);
int lbmpc_homothety_lp_dev(const double* WA, const double* Wb, int nW, int nx, const double* m_dev,
                           int64_t batch, int tiebreak, double* alpha_dev, double* v_dev,
                           uint32_t* active_mask_dev, int* status_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * Subsystem 3: the per-step tightened tube-MPC QP and the batched closed loop
 * ---------------------------------------------------------------------------------------- */

/* Fields mirror the `parameters` struct consumed by Functions_general/UQRobustMPC.m:39-62 and
 * NominalRobustMPC.m:28-45 (Polyhedra unpacked to A,b).  All host pointers; copied at create. */
typedef struct lbmpc_params {
    int nx, nu, N, nc;
    int Nu;               /* nu_s: constraint horizon, rows i = 0..Nu                     */
    int nS, nW;
    const double* A;      /* nx x nx */
    const double* B;      /* nx x nu */
    const double* K;      /* nu x nx */
    const double* Phi;    /* nx x nx  (= A + B K) */
    const double* Px;     /* nx x nx */
    const double* Pc;     /* (N nu) x (N nu) */
    const double* F;      /* nc x nx */
    const double* G;      /* nc x nu */
    const double* F_bar;  /* nc x (nx + N nu) */
    const double* Psi;    /* (nx + N nu)^2 */
    const double* S_A;    /* nS x nx */
    const double* S_b;    /* nS */
    const double* W_A;    /* nW x nx */
    const double* W_b;    /* nW */
    const double* hs;     /* nc : h_S((F+GK)_i) */
} lbmpc_params;

typedef struct lbmpc_ctx lbmpc_ctx;

/* Build the device-resident constant block (constraint matrix [-S_A 0; F_bar Psi^i], (2P)^-1,
 * T = (I-Phi)^-1, F_Com T, S_A T ...).  Replaces the constructors UQRobustMPC.m:39-66 /
 * NominalRobustMPC.m:28-45 (which build CasADi Opti objects). *ctx_out = NULL on error.
 * The prior W may be any bounded polytope (UQRobustMPC.m:124-139 takes any W): an axis-aligned box (nW = 2 nx, every
 * reference configuration) gets the closed-form set update; any other W with nx+1 <= nW <= 16 rows gets the
 * vertex-enumeration LP of lbmpc_homothety_lp, W_hat rows in the caller's order (LBMPC_ERR_DIM outside that range).
 * Such contexts run lbmpc_uqmpc_step, lbmpc_mc_run and lbmpc_multi_* like box contexts, one trial per lane and
 * without the steady-state kernel (written for a box). */
int lbmpc_uqmpc_create(const lbmpc_params* p, lbmpc_ctx** ctx_out);
void lbmpc_destroy(lbmpc_ctx* ctx);

/* One UQ-RMPC step for `batch` independent problems (column b of every array is problem b).
 * Replaces Functions_general/UQRobustMPC.m:68-91 (`solve`): online set update (LPforUQ), set
 * algebra :75-78, h_k (:80-84, closed form alpha*hs + (1-alpha) F_Com T v), tube QP (:87),
 * u_k = K x_k + c_0 (:90).
 *   in : x (nx x batch), alpha_prev (batch), v_prev (nx x batch), w_new (nx x batch)
 *   out: s0 (nx x batch), c (N nu x batch), u (nu x batch), alpha (batch), v (nx x batch),
 *        hk (nc x batch), Sb_hat (nS x batch), Wb_hat (nW x batch), status (batch),
 *        n_active (batch), active_set (LBMPC_MAX_NZ x batch, ascending row ids of the QP rows
 *        [S rows 0..nS-1, then row nS + i*nc + r for F_bar Psi^i row r], -1 padded),
 *        qp_iters (batch).  Any out pointer may be NULL.
 * uq_update = 0 skips the set update (alpha = alpha_prev, v = v_prev): with alpha_prev = 1,
 * v_prev = 0 this is Functions_general/NominalRobustMPC.m:47-52. */
int lbmpc_uqmpc_step(lbmpc_ctx* ctx, int64_t batch, int uq_update, const double* x,
                     const double* alpha_prev, const double* v_prev, const double* w_new, double* s0,
                     double* c, double* u, double* alpha, double* v, double* hk, double* Sb_hat,
                     double* Wb_hat, int* status, int* n_active, int* active_set, int* qp_iters);

/* Batched closed-loop Monte-Carlo (the loops of Cases_general/Case_3...:57-88, Case_4...:32-66,
 * Case_5...:55-98): per trial, learn (alpha0, v0) from card_I0 sampled disturbances
 * (InitialSetComputation.solve), then `steps` UQ-RMPC steps against the sampled plant
 * (ModelingCar.m:34-45) in relative coordinates x+ = A x + B u + xi_EV - xi_LV - B u_LV. */
typedef struct lbmpc_mc_cfg {
    int64_t trials;       /* trials run by this call                                       */
    int64_t trial0;       /* global id of the first one (RNG stream key; multi-GPU shards) */
    int32_t steps;        /* K_N                                                           */
    int32_t card_I0;      /* N_pre_sam                                                     */
    uint64_t seed;
    const double* x0;     /* nx: initial relative state x_RM_0 (same for all trials)       */
    double acc_bound;     /* |u| bound of the violation test, Case_5...:90-91              */
    double distance_error;/* state bound of the violation test, Case_5...:92-93            */
    int32_t traj_trials;  /* record full trajectories for the first traj_trials trials     */
    int32_t sampler_max_attempts; /* 1000 in ModelingCar.m:54                              */
} lbmpc_mc_cfg;

typedef struct lbmpc_mc_out { /* device or host pointers depending on the entry point; any may be NULL */
    int32_t* infeasible;  /* trials : Infeasible_Index of Case_5...:90-97                  */
    int32_t* status;      /* trials : max status word seen                                 */
    int32_t* qp_iters;    /* trials : total active-set iterations                          */
    double* alpha_final;  /* trials                                                        */
    double* v_final;      /* nx x trials                                                   */
    double* alpha0;       /* trials : after the initial set learning                       */
    double* x_final;      /* nx x trials                                                   */
    double* traj_x;       /* nx x (steps+1) x traj_trials  (State_RM)                      */
    double* traj_u;       /* nu x steps x traj_trials      (Control_EV)                    */
    double* traj_alpha;   /* (steps+1) x traj_trials                                       */
    double* traj_s0;      /* nx x steps x traj_trials      (Nominal_RM_State)              */
} lbmpc_mc_out;

int lbmpc_mc_run(lbmpc_ctx* ctx, const lbmpc_sampler* smp, const lbmpc_mc_cfg* cfg, lbmpc_mc_out* out);
int lbmpc_mc_run_dev(lbmpc_ctx* ctx, const lbmpc_sampler* smp, const lbmpc_mc_cfg* cfg,
                     lbmpc_mc_out* out_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * Scenario MPC, the comparison baseline of Case_6
 * ---------------------------------------------------------------------------------------- */

/* Fields of the struct Functions_general/ScenarioMPC.m:30-48 consumes (Cases_general/Case_6...:22-40). */
typedef struct lbmpc_scmpc_params {
    int nx, nu, N, nc;
    int N_SC;             /* scenarios per step (Case_6...:25: N_Samples / N = 50)        */
    const double* A;      /* nx x nx */
    const double* B;      /* nx x nu */
    const double* Q;      /* nx x nx */
    const double* R;      /* nu x nu */
    const double* F;      /* nc x nx */
    const double* G;      /* nc x nu */
} lbmpc_scmpc_params;

typedef struct lbmpc_scmpc_out { /* host pointers; any may be NULL */
    int32_t* infeasible;  /* trials : bound violation flag, as lbmpc_mc_out                 */
    int32_t* status;      /* trials : max status (LBMPC_ST_QP_INFEASIBLE: scenario QP empty) */
    int32_t* qp_iters;    /* trials                                                         */
    double* x_final;      /* nx x trials                                                    */
    double* traj_x;       /* nx x (steps+1) x traj_trials (State_RM_SCMPC)                  */
    double* traj_u;       /* nu x steps x traj_trials     (Control_EV_SCMPC)                */
} lbmpc_scmpc_out;

/* Batched closed loop under ScenarioMPC.solve (ScenarioMPC.m:52-101): per step and trial draw N_SC x N disturbance
 * scenarios w = xi_EV - xi_LV - B u_LV (:60-68), solve the scenario QP in the N*nu inputs (states eliminated; the
 * N_SC*N*nc rows F x + G u <= 1 share nc*N normals, so only the tightest scenario per normal is kept: a max-reduction
 * over the scenarios), apply u_0, step the plant with the SAME step disturbances lbmpc_mc_run uses for these trial ids
 * (Case_6 runs both controllers against one realisation).  cfg->card_I0 is ignored.  One trial per wave (the scenarios
 * of a step are spread over the lanes). */
int lbmpc_scmpc_mc_run(const lbmpc_scmpc_params* p, const lbmpc_sampler* smp, const lbmpc_mc_cfg* cfg, lbmpc_scmpc_out* out);

/* ------------------------------------------------------------------------------------------
 * Feasible-region geometry
 * ---------------------------------------------------------------------------------------- */

/* h_out[i + nd*s] = max { C[:,i]' z : A z <= b[:,s] }: the support function of a polytope in H-rep by an exact LP, for
 * nd directions (C: n x nd) and nb right-hand sides (b: m x nb, every entry > 0: z = 0 strictly feasible); A: m x n.
 * z_out (n x nd x nb, may be NULL): a maximiser; status_out (nd x nb, may be NULL): 0, LBMPC_ST_LP_UNBOUNDED or
 * LBMPC_ST_ITER_CAP.  One LP per lane (primal active-set method with Bland's rule).
 * Replaces the IPOPT LPs of Functions_general/ComputeFeasibleRegion.m: Com_hs / Com_hs_ini (:447-503, A = S_A),
 * Com_z (:505-519, the nu_s search of :66-116), and the sampled projections of :153-171, :395-413 (with A the rows
 * F_bar Psi^i, i = 0..nu, and C = [d; 0] it returns the exact support of the nominal feasible region F_Ns, so that
 * h_{F_N}(d) = h_{F_Ns}(d) + h_{S_hat}(d) needs no hull of 5000 projected samples). */
int lbmpc_polytope_support(const double* A, int m, int n, const double* b, int64_t nb, const double* C, int64_t nd,
                           double* h_out, double* z_out, int* status_out);

/* Row redundancy of {z : A z <= b} (A: m x n column-major, b > 0): LP j maximises the unit normal of row j over all
 * the OTHER rows, one LP per lane.  slack_out[j] = b_j/|a_j| - max: >= 0 means the other rows imply row j (redundant),
 * < 0 that row j cuts; -inf with status LBMPC_ST_LP_UNBOUNDED when only row j bounds that direction.
 * These are the LPs behind MPT3's minHRep, which the reference calls on every set it builds
 * (offline_parameters_computation_nxn.m:221-222, :399, :431, :457; Functions_general/ComputeFeasibleRegion.m:112);
 * duplicate merging and the final keep-list are host logic (lbmpc.api.minHRep). */
int lbmpc_polytope_redundant(const double* A, int m, int n, const double* b, double* slack_out, int* status_out);

/* ------------------------------------------------------------------------------------------
 * One node, several GPUs, inside one process (the MATLAB drop-in is a single process)
 * ---------------------------------------------------------------------------------------- */

/* One controller context per device, built from the same parameters (devices: ngpu indices, NULL = 0..ngpu-1;
 * every listed device is validated as lbmpc_init does).  No counterpart in the reference, whose only parallel
 * construct is `parfor` (Functions_general/MRPISet.m:110); SURVEY.md 8e. */
typedef struct lbmpc_multi lbmpc_multi;
int lbmpc_multi_create(const lbmpc_params* p, const int* devices, int ngpu, lbmpc_multi** out);
void lbmpc_multi_destroy(lbmpc_multi* m);
int lbmpc_multi_ngpu(const lbmpc_multi* m);
/* The contiguous block partition used by every sharded entry point (and by the one-process-per-GPU driver,
 * lbmpc/dist.py): shard `rank` of `world` gets [first, first + count), sizes differ by at most one.  Needs no device. */
int lbmpc_block_shard(int64_t total, int rank, int world, int64_t* first, int64_t* count);

/* lbmpc_mc_run over the devices of `m`: contiguous block shards of the trials (device g of G runs trials
 * [g*T/G, (g+1)*T/G) of cfg, keyed by their global ids cfg->trial0 + ...), one host thread per device, no exchange
 * during the closed loop; every device copies its shard into the caller's HOST arrays.  Results are bitwise those of
 * lbmpc_mc_run on one device.  Replaces the serial Monte-Carlo loops of Cases_general/Case_4...:32-74,
 * Case_5...:55-98. */
int lbmpc_multi_mc_run(lbmpc_multi* m, const lbmpc_sampler* smp, const lbmpc_mc_cfg* cfg, lbmpc_mc_out* out);

/* lbmpc_halfspace_max with every information set sharded BY SAMPLE over ngpu devices: local max-reductions, then
 * ONE all-reduce(max) of batch*nW doubles over RCCL (ncclCommInitAll in-process; librccl is loaded on first use),
 * bit-identical to a single device.  Host pointers.  Replaces the sample rows of
 * Functions_general/InitialSetComputation.m:77-79 for information sets too large for, or already spread over,
 * several GPUs. */
int lbmpc_halfspace_max_multi(const double* WA, int nW, int nx, const double* samples, int64_t N, int64_t batch,
                              int64_t batch_stride, double* m_out, const int* devices, int ngpu);

/* Work counters of the last lbmpc_mc_run / lbmpc_mc_run_dev on this context (synchronises the device).  No
 * counterpart in the reference (its loops are timed with tic/toc, Case_3...:57,69); used by bench.py to state the
 * arithmetic that was actually executed.  out[0] wave-steps whose rows were scanned by the fp32 filter (64 trials
 * each), out[1] wave-steps in total, out[2] QP solves, then the 16-lane-row solver: out[3] rounds (up to 4 QPs each),
 * out[4] loop trips, out[5] full pivot scans (4 QPs each), out[6] passes over the H rows inside them (2 QPs each);
 * the one-QP-per-lane solver of the S-row regime: out[7] loop trips (up to 64 QPs each), out[8] scan passes over
 * the S rows (up to 4 QPs each); out[9] QPs that went through the 16-lane-row solver; out[10] QPs served by the scan passes of out[8];
 * out[11] reserved. */
int lbmpc_mc_counters(lbmpc_ctx* ctx, uint64_t* out12);

/* The same counters and what SURVEY 8(d) asks of subsystem 3 beside them (no counterpart in the reference either): the
 * first min(n, 160) slots of: out[0..10] as lbmpc_mc_counters; the one-QP-per-lane solver with rows of F_bar Psi^i in
 * play (qp_lane_gen.hpp): out[11] loop trips (up to 64 QPs each), out[12] scan passes over the S rows (up to 4 QPs
 * each), out[13] passes over the H rows (2 QPs each), out[14] QPs served by the scan passes, out[15] trips that rebuilt
 * factor rows after a drop, out[16] QPs that entered this solver, out[17] chunks of 64 H rows its scans evaluated (the others were certified); out[18..31] reserved;
 * out[32 + i], i < 64: QP solves that took i active-set iterations (i = 63: 63 or more);
 * out[96 + k], k < 64: QP solves that ended with k active rows. */
int lbmpc_mc_stats(lbmpc_ctx* ctx, uint64_t* out, int n);

/* Launch shape and kernel resources lbmpc_mc_run would use for `trials` x `steps` on this context (occupancy
 * and LDS bytes/CU of subsystem 3; no counterpart in the reference).  The first min(n, 11) of: out[0] trials per wave,
 * [1] trials per lane, [2] waves per block, [3] dynamic LDS bytes per block (one block per CU), [4] lanes of the
 * general lane solver's LDS region per wave (0: that tier is not available), [5] steps run by the transient kernel,
 * [6] 1 if the steady-state kernel runs the rest, [7] registers per lane of the transient kernel, [8] its scratch bytes
 * per lane, [9] resident waves per SIMD (the smaller of the register and LDS limits), [10] blocks of the transient launch. */
int lbmpc_mc_plan(lbmpc_ctx* ctx, int64_t trials, int steps, int32_t* out, int n);

#ifdef __cplusplus
}
#endif
#endif /* LBMPC_H */
