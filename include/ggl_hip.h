/*
 * ggl_hip.h -- C ABI of libggl_hip.so: the MI355X (gfx950) ADMM inner loop for General Graphical
 * Lasso.  This is the drop-in boundary for the hot path of fabian-sp/GGLasso; every entry point
 * names the reference interface it replaces (paths relative to /root/reference/src/gglasso/).
 *
 * Conventions
 *   - All matrices are float64, C-contiguous, row-major; stacks are (K,p,p) with k slowest.
 *   - "host" pointers are ordinary CPU memory owned by the caller (NumPy arrays); the library
 *     owns every device buffer inside the opaque ggl_ctx and never keeps a host pointer.
 *   - Every function returns 0 on success, <0 on error (GGL_E_*); ggl_last_error() returns a
 *     thread-local message.  No C++ exception crosses the ABI.
 *   - A ctx is bound to one device and one HIP stream and is not thread-safe.  Multi-GPU = one
 *     process (rank) per GPU, each with its own ctx (see gglasso_amd/dist.py).
 *   - Eigenvector matrices cross the ABI in NumPy's convention: Q (K,p,p) with eigenvectors in
 *     COLUMNS, eigenvalues ascending (numpy.linalg.eigh, called at solver/admm_solver.py:181,199).
 */
#ifndef GGL_HIP_H
#define GGL_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version: major * 100 + minor.  Bumped whenever an entry point changes its argument layout or an existing buffer its
 * format -- 300 (round 6): ggl_pipeline_stats writes out[10] (was out[5]), GGL_BUF_GROUPSQ is the packed upper triangle with
 * the flag at p (p + 1) / 2 (was (p,p) + 1), the int8 entry points live in the development library only, ggl_debug_poison
 * takes a byte pattern.  A binding built against another major version must refuse to load (gglasso_amd/_lib.py does). */
#define GGL_VERSION 300

/* error codes */
#define GGL_OK 0
#define GGL_E_ARG (-1)      /* bad argument (the reference would raise AssertionError) */
#define GGL_E_HIP (-2)      /* HIP runtime error */
#define GGL_E_SOLVER (-3)   /* rocSOLVER / eigensolver did not converge */
#define GGL_E_ALLOC (-4)
#define GGL_E_COMM (-5)     /* RCCL error, or librccl could not be loaded */

/* penalty selector: reg argument of ADMM_MGL (solver/admm_solver.py:13-31) / ADMM_SGL */
#define GGL_REG_SGL 0       /* K independent single problems: prox_od_1norm (ggl_helper.py:16-27) */
#define GGL_REG_GGL 1       /* prox_phi_ggl (ggl_helper.py:68-71) */
#define GGL_REG_FGL 2       /* prox_phi_fgl (ggl_helper.py:131-134) */

/* eigensolver selector (ctx flags, low byte) */
#define GGL_EIG_AUTO 0      /* p <= GGL_NS_MIN_P: LDS Jacobi for everything.  Larger p: Newton-Schulz matrix functions for the
                             * Omega-step and the L-step; where eigenvalues themselves are needed (exit checks, KKT, objective,
                             * selection statistics, the final L, fallbacks) LDS Jacobi up to GGL_JACOBI_MAX_P, rocSOLVER above. */
#define GGL_EIG_JACOBI 1    /* hand-written one-workgroup-per-matrix Jacobi (p <= GGL_JACOBI_MAX_P) */
#define GGL_EIG_ROCSOLVER 2 /* rocsolver_dsyevd_strided_batched */
#define GGL_EIG_NEWTON_SCHULZ 3 /* Omega-step without an eigendecomposition: Omega = (W + sqrt(W^2+4 beta I))/2
                                 * by scaled Newton-Schulz products on the FP64 matrix cores (any p).
                                 * Steps that need eigenvalues themselves (L-step, exit checks) use AUTO. */
#define GGL_JACOBI_MAX_P 128
#define GGL_NS_MIN_P 8        /* GGL_EIG_AUTO: Omega- / L-step by Newton-Schulz matrix functions for p > this (measured crossover) */
/* Newton-Schulz controls in the bits above the selector (ctx flags and the eig_method argument of the stateless
 * ggl_phiplus_matrix / ggl_rank_matrix): product mode 0 auto (by condition number), 1 all-symmetric products,
 * 2 stable unsymmetrised products; highest step degree 3, 5 or 9 (0 = 9).  Used by the parity tests to reach every
 * schedule; the solvers pass neither. */
#define GGL_EIG_NS_MODE(m) (((m) & 0x3) << 8)
#define GGL_EIG_NS_DEGREES(d) (((d) & 0xf) << 12)
/* ctx flag: the `stream` argument of ggl_ctx_create is the caller's stream even when the handle is NULL (the legacy
 * default stream -- what torch.cuda.current_stream().cuda_stream is for torch's default stream).  Without the bit a
 * NULL handle means "create a private non-blocking stream", which has NO implicit ordering with the NULL stream. */
#define GGL_CTX_STREAM_GIVEN (1 << 16)

/* which-buffer selector of ggl_device_ptr */
#define GGL_BUF_S 0
#define GGL_BUF_OMEGA 1
#define GGL_BUF_THETA 2
#define GGL_BUF_L 3
#define GGL_BUF_X 4
#define GGL_BUF_GROUPSQ 5   /* partial sum_k u^2 of the GGL Theta-step (K-sharded runs): PACKED upper triangle, row-major with the
                             * diagonal, element (i,j), i <= j, at i p - i (i - 1) / 2 + (j - i): p (p + 1) / 2 doubles, + 1 flag double */
#define GGL_BUF_NORMS 6     /* (K,5) per-instance squared norms of the stopping test */
#define GGL_BUF_OMEGA_PREV 7

typedef struct ggl_ctx ggl_ctx;

int ggl_version(void);
const char *ggl_last_error(void);
/* number of visible HIP devices, or <0 (used by the loader to fail loudly on a GPU-less box) */
int ggl_device_count(void);

/* ---- context ---------------------------------------------------------------------------------
 * Replaces the NumPy temporaries ADMM_MGL allocates per call (admm_solver.py:142-154).
 * stream: an existing hipStream_t or NULL to create one (see GGL_CTX_STREAM_GIVEN for the NULL stream itself). */
int ggl_ctx_create(int device, int K, int p, int flags, void *stream, ggl_ctx **out);
int ggl_ctx_destroy(ggl_ctx *ctx);          /* (NULL: a no-op that succeeds, like free) */
int ggl_ctx_sync(ggl_ctx *ctx);
void *ggl_device_ptr(ggl_ctx *ctx, int which);

/* ctx options.  The shipped library reads NO environment variables: every dispatch decision is a function of the
 * problem shape and of these options (defaults in brackets).  None of them changes what is computed, only how; the
 * parity tests run the iteration under each setting.  (A GGL_DEV build of the library -- libggl_hip_dev.so, used by
 * tools/ -- additionally maps the environment variables GGL_SPECULATE, GGL_SPEC_FACTOR, GGL_NS_MODE, GGL_NS_DEGREES,
 * GGL_THETA_FLAT, GGL_RANK_EIG, GGL_TWO_STREAM, GGL_PARTS_MAX_TILES, GGL_SYMM_VARIANT, GGL_SPIN_WAIT,
 * GGL_FUSED_BOUNDS, GGL_PIPELINE, GGL_FUSED_START, GGL_PARTS_SMALL onto them.) */
/* Options marked DEV are the measured-and-rejected alternatives of rounds 2-5 (each built, bit-identical or parity-tested, and
 * slower or within noise: DESIGN.md 8.1, 9.7, 9.10, 10.7).  Since round 6 they exist in the development library only
 * (libggl_hip_dev.so); the product library accepts the value 0 for them and refuses anything else with GGL_E_ARG. */
#define GGL_OPT_SPECULATE 1        /* [1] speculative Omega-step (schedule from the previous iteration's bounds)     */
#define GGL_OPT_SPEC_FACTOR 2      /* [1.02] inflation of the previous bounds; < 1 forces validation misses (tests)  */
#define GGL_OPT_NS_MODE 3          /* [0] as GGL_EIG_NS_MODE                                                          */
#define GGL_OPT_NS_DEGREES 4       /* [9] as GGL_EIG_NS_DEGREES                                                       */
#define GGL_OPT_THETA_FLAT 5       /* [2] GGL Theta-step for exactly symmetric states, K <= 32: 0 tile-pair kernels, 1 per-element
                                    * kernel, 2 per-element kernel with the K-column split over the four waves of a workgroup */
#define GGL_OPT_RANK_EIG 6         /* [0] L-step by eigendecomposition instead of the sign iteration                 */
#define GGL_OPT_PARTS 7            /* [2] parts of the batch that run their launch sequences concurrently (1..4)     */
#define GGL_OPT_PARTS_MAX_TILES 8  /* [2048] concurrent parts only up to this many 64x64 tile pairs in the batch     */
#define GGL_OPT_SYMM_VARIANT 9     /* [-1 = by size] product-kernel instance, see csrc/gemm_sym.hip                   */
#define GGL_OPT_SPIN_WAIT 10       /* [1] wait for the end of an iteration by polling a pinned sequence number       */
#define GGL_OPT_FUSED_BOUNDS 11    /* [1] spectral-bound partials from the epilogue of the product launch (no norm pass) */
#define GGL_OPT_PIPELINE 12        /* [1] ggl_admm_step launches the next iteration's Omega-step chain before it returns     */
#define GGL_OPT_FUSED_START 13     /* [1] speculative step: the first step's start matrix is the B' launch's second output   */
#define GGL_OPT_PARTS_SMALL 14     /* [8] smallest batch below 16 (p >= 384) that still runs as two concurrent parts; 0 = none */
#define GGL_OPT_NS_TOL 15          /* [2e-12] relative spectral accuracy the Omega-step's matrix square root is iterated to:
                                    * |Omega - phiplus(W)|_2 <= tol * |sqrt(W^2 + 4 beta I)|_2 / 2 (+ rounding).  0 (or anything
                                    * below 4e-16) iterates to fp64 resolution: results then agree with eigh to ~1e-13 at one
                                    * more product per iteration around condition number 2.  Measured at GGL (32,500), solved
                                    * to 1e-10: |Theta - reference|_F = 1.3e-10 (default) / 2.8e-13 (exact) on a stack of norm
                                    * 128; the reference comparison asks for 1e-8.  The stateless ggl_phiplus_matrix is exact. */
#define GGL_OPT_CW_WARM 16         /* [1] the Collatz-Wielandt weight vector of the spectral bound is carried from one iteration to
                                    * the next (a power iteration at no extra pass; the bound stays rigorous and tightens
                                    * towards the Perron root of |B'|); 0: the row sums of |B'| every time                  */
#define GGL_OPT_CHAIN 17           /* DEV [0] 1: speculative Omega-steps of batches that cover the XCDs (K >= 8, even p, >= 1100 64x64 tile
                                    * pairs) run their whole product chain as ONE persistent launch with per-instance
                                    * dependencies (k_omega_chain, csrc/gemm_sym.hip) instead of one launch per product in
                                    * concurrent parts; 2: wherever it can run (any K >= 8, even p).  Same products, same bits
                                    * (tests/test_gpu_chain.py).  Off by default: measured SLOWER on MI355X (headline Omega
                                    * phase 0.89 vs 0.72 ms; DESIGN.md section 8.1, profiles/r3_omega_chain_*.txt)           */
#define GGL_OPT_RANK_DEFLATE 20    /* [1] L-step (sign iteration): after a first pass at GGL_OPT_RANK_L0_DEFLATE the eigenvalues next to the
                                      threshold are deflated (range of I - X^2, exact small problem) instead of iterated down */
#define GGL_OPT_RANK_L0_DEFLATE 21 /* [2e-3] resolution of that first pass */
#define GGL_OPT_FUSED_CW 22        /* DEV [0] the bound validation of a speculative Omega-step (row sums + Collatz-Wielandt pass) as ONE launch:
                                      built in round 4 for the small slabs, measured equal there and slower at the headline */
#define GGL_OPT_OMEGA_LDS 23       /* [1] p <= 64: the whole Omega-step as ONE launch, one workgroup per instance, the Newton-Schulz chain
                                      resident in LDS, bound and per-instance schedule chosen on the device (omega_lds.hip); an instance
                                      outside its range (condition number of W^2 + 4 beta I above 300) sends the step to the launch chain.
                                      1: waves per workgroup by size (8, i.e. two per SIMD, for 32 < p <= 64; round 5), 4 / 8: that many */
#define GGL_OPT_EARLY_PART 24      /* [1] with GGL_OPT_PIPELINE, ggl_admm_step: the first part of the NEXT iteration's Omega-step chain (tables,
                                      W, A', B': scratch only) goes into the stream before the host waits for this iteration's residuals,
                                      on the prediction that the rho rule keeps rho; its schedule is built from the bounds validated one
                                      iteration earlier, so iterates agree with GGL_OPT_EARLY_PART = 0 to the Omega-step's tolerance */
#define GGL_OPT_FUSED_W 26         /* [1] with GGL_OPT_EARLY_PART, GGL, exactly symmetric state and S: the Theta kernel that precedes an early
                                      first part also writes that part's W = Theta - X - beta S (admm_solver.py:180) from the values it
                                      holds -- one pass over three stacks and one launch per part less (round 5) */
#define GGL_OPT_RANK_CW 27         /* DEV [0] L-step (sign iteration): the norm bound |C|_2 from a Collatz-Wielandt pass over C C with a vector
                                      carried across ADMM iterations (as the Omega-step's bound) instead of sqrt(min(|C C|_inf, |C C|_F)).
                                      Built and measured in round 5 (FGL K=50, p=500, latent): the schedule is a step function of the
                                      resolution -- 22.0 products with either bound at GGL_OPT_RANK_L0_DEFLATE = 2e-3, 21.4 / 20.8 at
                                      4e-3 where 3-4x as many instances need the continuation -- and the pass costs 25 us: 211 it/s
                                      without, 202 with.  Off. */
#define GGL_OPT_BOUND_SIDE 28      /* DEV [0] speculative Omega-step: the two kernels that validate the assumed bound run on a side stream beside the
                                      chain's first products, joined before B' is overwritten (round 5).  0 off, 1 on, 2 only for two
                                      concurrent parts of a large batch.  Bitwise the in-chain order; measured: the small launch-bound
                                      sequences lose 2-5 % to the cross-stream waits, the headline gains 0.8 % over nine A/B pairs
                                      whose scatter is +-2 % (profiles/r5_bound_side.txt): off */
#define GGL_OPT_LDS_PINNED 29      /* [1] p <= 64: the LDS-resident kernels (and the Theta kernel of a batched grid behind them) read their per-step
                                      parameters from the pinned host mirror instead of waiting for a copy kernel in front of them */
#define GGL_OPT_JOIN_FLAG 30       /* [1] concurrent parts of a speculative Omega-step are joined through flag words in device memory (a one-wave
                                      kernel on the main stream polls what the parts' streams set behind their last launch) instead of a
                                      cross-queue event wait, which idles the waiting queue ~25 us after the event has fired */
#define GGL_OPT_CW_RIDER 31        /* [1] speculative Omega-step: the validation of the assumed spectral bound (row sums, Collatz-Wielandt ratio,
                                      Frobenius norm of B' = A'^2) runs as extra workgroups of the first product launch after B' instead of two
                                      dependent launches in front of it (the schedule does not wait for it; small launch sequences leave most
                                      CUs idle).  Same arithmetic, same bound.  2: the rider as a launch of its own (what a chain without such a
                                      product launch gets; for tests). */
#define GGL_OPT_COPY_RIDER 32      /* [1] the small host-to-device table transfers in front of an Omega-step chain (schedule coefficients, assumed bounds,
                                      flag zeroing) ride in the A' = W^2 + 4 beta I launch as extra workgroups when nothing else is pending and the
                                      device's coefficient rows of A' are already those of this beta: one dependent launch less per iteration.  1 = where the
                                      chain is ONE launch sequence (two concurrent parts measured slower with it), 2 = always, 0 = never */
#define GGL_OPT_REDUCE_RIDER 33    /* [2] the norm reduction behind a Theta-step rides in the A' launch of the next chain's early first part (one extra
                                      workgroup; same sums in the same order) when that launch follows it in the stream anyway: one dependent
                                      launch less per iteration.  1 = single launch sequences, 2 = always, 0 = never */
#define GGL_OPT_PARTS_BIAS 34      /* DEV [0] two concurrent parts of an Omega-step take K/2 + bias and K/2 - bias instances (the second part starts and
                                      ends ~40 us after the first) */
#define GGL_OPT_PARTS_ORDER 35     /* DEV [0] two concurrent parts: 1 = the part on the ctx's main stream is queued after the other one */
#define GGL_OPT_DOWNLOAD_THREADS 36 /* [8] ggl_get_state / ggl_get_snapshots of more than 32 MB: host threads that touch the pages of the caller's
                                      (typically freshly allocated) arrays before the copy -- the first touch, not the transfer, is what a
                                      download into new memory waits for; 1 = none */
#define GGL_OPT_GROUP_SCHED 37     /* [1] a batch whose instances need different product counts for their matrix square root (a grid of
                                      independent problems: the reference solves every point with its own eigh, helper/model_selection.py:
                                      619-633) runs as up to 3 contiguous groups with their own Newton-Schulz schedules -- the concurrent
                                      parts of the engine -- where the size rule would run it as one launch sequence on the worst
                                      instance's schedule; taken when a deterministic time model gains >= 6 %.  0 off, 2 / 3 = at most
                                      that many groups; 12 / 13 (tests) = at most 2 / 3 groups wherever the product counts differ */
#define GGL_OPT_PART_PRIORITY 25   /* DEV [0] streams of the concurrent parts of an Omega-step: 0 = created like any stream, 1 = with the highest,
                                      2 = with the lowest stream priority (streams of another priority never share a hardware queue with
                                      the ctx's main stream) */
#define GGL_OPT_ISOLATE 19         /* [0] batches of independent problems: an instance whose data turn non-finite or whose eigensolver
                                      does not converge is marked (ggl_failed_instances) instead of failing the call */
#define GGL_OPT_RANK_L0_COARSE 18  /* [8e-5] two-tier L-step (sign iteration, p > GGL_JACOBI_MAX_P): the first pass over the whole batch
                                    * resolves eigenvalues of C down to this distance from the threshold (relative to |C - mu I|);
                                    * the instances whose residual check says that was not enough are continued, from the iterate
                                    * they have, as a compact sub-batch down to the resolution of the one-tier run (1e-6).  The
                                    * schedule's length is set by the WORST instance of a batch and its degree sequence is common
                                    * to a launch: at K = 50, p = 500 the batch needs 1e-6, nine instances in ten 1e-4
                                    * (28 instead of 37 products).  Measured there (with the norm bound from C^2): 154 -> 175 it/s,
                                    * flat for 5e-5 .. 1e-4.
                                    * 0: one tier.                                                                            */
int ggl_ctx_set_option(ggl_ctx *ctx, int option, double value);
int ggl_ctx_get_option(ggl_ctx *ctx, int option, double *value);

/* ---- state upload / download ----------------------------------------------------------------
 * S: admm_solver.py:13 (argument S).  set_state: Omega_0/Theta_0/X_0 copies, admm_solver.py:142-150
 * (NULL leaves the buffer as is; L NULL zeroes it like admm_solver.py:149).  get_state: the `sol`
 * dict of admm_solver.py:303 (NULL = skip). */
int ggl_set_S(ggl_ctx *ctx, const double *S_host);
int ggl_set_state(ggl_ctx *ctx, const double *Omega, const double *Theta, const double *L,
                  const double *X);
/* The same with arrays SHARED by the instances of a batch: `period` / periods[0..3] (Omega, Theta, L, X; NULL = all 0) is the
 * number of instances the host array holds -- 0: all K; P > 0 (a divisor of K, anything else is GGL_E_ARG): P, and instance k of the
 * ctx is its instance k % P.  P = 1: one (p,p) matrix for all (the grid of single_grid_search starts every instance from the same S, Omega_0, X_0);
 * P = K': the stack of ONE multiple-graph problem for each of the G grid points of grid_search.  Uploaded once, replicated on
 * the device. */
int ggl_set_S_ex(ggl_ctx *ctx, const double *S_host, int period);
int ggl_set_state_ex(ggl_ctx *ctx, const double *Omega, const double *Theta, const double *L, const double *X,
                     const int *periods);
int ggl_get_state(ggl_ctx *ctx, double *Omega, double *Theta, double *L, double *X);
/* restore == 0: keep a device copy of the iterate; != 0: make it the iterate again (ggl_set_state with the arrays of that
 * moment, without the host round trip; everything carried from earlier iterations is forgotten, as in ggl_set_state). */
int ggl_state_snapshot(ggl_ctx *ctx, int restore);
/* lambda1 * lambda1_mask as a (p,p) array (single_admm_solver.py:114); NULL clears it. */
int ggl_set_lambda1_mask(ggl_ctx *ctx, const double *lam_pp_host);

/* ---- one ADMM iteration ---------------------------------------------------------------------
 * Body of the hot loop, admm_solver.py:179-224 / single_admm_solver.py:163-193:
 *   W = Theta - L - X - (nk/rho) S ; (D,Q) = eigh(W) ; Omega = phiplus(nk/rho, D, Q)
 *   Theta = prox_p(Omega + L + X, lambda1/rho, lambda2/rho, reg)      (prox_od_1norm for SGL)
 *   latent: L = prox_rank_norm(Theta - X - Omega, mu1/rho)
 *   X += Omega - Theta + L
 * out_norms[5] = squared Frobenius norms over the whole stack that ADMM_stopping_criterion
 * (admm_solver.py:316-331) needs: |Omega|^2, |Theta-L|^2, |X|^2, |Omega-Theta+L|^2,
 * |Omega-Omega_prev|^2.  rho update and the stopping decision stay on the host (admm_solver.py:227-246).
 * mu1: host (K,) or NULL; nk: host (K,) or NULL (= ones).  rho, lambda1 are per-iteration scalars.
 * For p > 128 the Omega- and L-steps are matrix-function iterations on the FP64 matrix cores; with
 * the same rho as in the previous call the Omega-step runs on a schedule built from that call's spectral bounds
 * and is validated on the device -- a rejected step leaves the iterate untouched and is repeated inside this call
 * (GGL_OPT_SPECULATE 0 switches that off).  One host synchronisation per call in the common case.  With GGL_OPT_PIPELINE
 * the call also launches the NEXT iteration's Omega-step chain for the same rho before it returns whenever the
 * reference's rho rule (admm_solver.py:227-233) would keep rho for the residuals just computed; the next call takes the
 * chain over if its rho (and nk) are the same and drops it otherwise, and every other entry point drops it first. */
int ggl_admm_step(ggl_ctx *ctx, double rho, double lambda1, double lambda2, int reg, int latent,
                  const double *mu1, const double *nk, double out_norms[5]);
/* The caller's loop ends after the next ggl_admm_step (max_iter reached, admm_solver.py:172): that call does not
 * pre-launch a chain nobody would take over.  One-shot; without it the unused chain is merely dropped by the next
 * entry point. */
int ggl_hint_last_step(ggl_ctx *ctx);

/* The same iteration split at the one point where a K-sharded GGL run has to exchange data
 * (sum_k u^2 couples the shards, ggl_helper.py:38-43):
 *   ggl_step_omega          Omega-step on the local K-slab
 *   ggl_step_group_partial  u = soft(Omega+L+X, l1/rho); GROUPSQ = packed upper triangle of sum_{local k} u^2  (GGL only)
 *   -- caller all-reduces GGL_BUF_GROUPSQ over ranks (RCCL) --
 *   ggl_step_finish         Theta from the reduced GROUPSQ, L-step, X update, local norms       */
int ggl_step_omega(ggl_ctx *ctx, double rho, int latent, const double *nk);
int ggl_step_group_partial(ggl_ctx *ctx, double rho, double lambda1);
/* ggl_step_omega that may run speculatively (schedule from the previous iteration's bounds, no host sync).  The
 * validation flag of this rank is appended to GROUPSQ as element p(p+1)/2 by ggl_step_group_partial, so the caller's
 * all-reduce must cover p(p+1)/2 + 1 doubles; ggl_step_finish reads the reduced flag back and, if ANY rank missed, leaves
 * the iterate alone; ggl_norms_read (or ggl_step_finish without deferred norms) then returns 1 on every rank:
 * repeat the iteration with ggl_step_omega. */
int ggl_step_omega_spec(ggl_ctx *ctx, double rho, int latent, const double *nk);
int ggl_step_finish(ggl_ctx *ctx, double rho, double lambda1, double lambda2, int reg, int latent,
                    const double *mu1, int groupsq_ready, double out_norms[5]);
/* groupsq_ready bit 1 (value 2): leave the five local sums in the NORMS buffer instead of returning them, so that
 * a K-sharded run can all-reduce them on the device (no host round trip); ggl_norms_read then copies them out
 * (one stream sync), exactly what ggl_step_finish does itself without the bit. */
int ggl_norms_read(ggl_ctx *ctx, double out_norms[5]);

/* ---- RCCL behind the ABI: the K-sharded GGL iteration as one call -------------------------------
 * One process per GPU, each with a ctx over its K-slab.  librccl.so.1 is resolved at run time
 * (dlopen), libggl_hip.so does not link it.
 *   ggl_comm_unique_id  ncclGetUniqueId on one rank; the host program distributes the 128 bytes by its own means
 *                       (MPI_Bcast, a file, torch.distributed's store -- gglasso_amd/dist.py uses broadcast_object_list)
 *   ggl_comm_init       ncclCommInitRank for this ctx's device; the collectives run on the ctx stream
 *   ggl_admm_step_sharded  the whole iteration of admm_solver.py:179-224 on this rank's slab with its two exchanges:
 *                       Omega-step (speculative) | local sum_k u^2 -> ncclAllReduce of GROUPSQ (p(p+1)/2 + 1 doubles: the
 *                       packed upper triangle, the validation flag rides along) | Theta-step, dual update, five local sums -> ncclAllReduce (5
 *                       doubles) | one host synchronisation.  out_norms are the GLOBAL sums: every rank takes the same
 *                       rho / stopping decision.  A rejected speculative step is repeated inside the call on all ranks.
 *   ggl_allreduce_groupsq / ggl_allreduce_norms  the two collectives alone, for callers that use the split entry points. */
int ggl_comm_unique_id(char id_out[128]);
int ggl_comm_init(ggl_ctx *ctx, int rank, int nranks, const char id[128]);
int ggl_comm_destroy(ggl_ctx *ctx);
/* ncclCommCount of the ctx's communicator: the number of ranks RCCL itself sees (bench.py reports it as n_ranks_seen) */
int ggl_comm_count(ggl_ctx *ctx, int *nranks_out);
int ggl_allreduce_groupsq(ggl_ctx *ctx);
int ggl_allreduce_norms(ggl_ctx *ctx);
int ggl_admm_step_sharded(ggl_ctx *ctx, double rho, double lambda1, double lambda2, const double *nk,
                          double out_norms[5]);
/* the same with latent variables (admm_solver.py:197-208): the L-step and the dual update are per instance, so they run on
 * the local slab between the Theta-step (reduced group sums) and the all-reduce of the five sums; mu1 (K_local,).  A latent
 * step neither speculates nor pre-launches (the L-step has its own verification synchronisation). */
int ggl_admm_step_sharded_latent(ggl_ctx *ctx, double rho, double lambda1, double lambda2, int latent,
                                 const double *mu1, const double *nk, double out_norms[5]);

/* ---- K independent single problems (batched lambda path) ---------------------------------------
 * The ctx stack is used as K separate ADMM_SGL problems (single_admm_solver.py:157-214), each with its
 * own rho_k and lambda1_k -- what the outer loop of single_grid_search (helper/model_selection.py:619-630)
 * solves one after the other.  out_norms is (K,5): the five squared norms per instance.
 * ggl_scale_X_batch: X_k <- factor_k X_k.  ggl_get_state_k: one instance of the state (snapshot at the
 * iteration it converged). */
int ggl_sgl_batch_step(ggl_ctx *ctx, const double *rho, const double *lambda1, int latent,
                       const double *mu1, double *out_norms);
/* The batch as problems of DIFFERENT dimension -- the connected components block_SGL solves one after the other
 * (single_admm_solver.py:422-459): instance k is the leading (pk[k],pk[k]) block of its (p,p) slot, the caller pads the rest of
 * S, Omega, Theta with an identity block and X with zeros (a decoupled fixed point of the iteration, as for ggl_ext_*);
 * ggl_sgl_batch_step's five sums then run over the block only, so every instance stops exactly where its own ADMM_SGL
 * would.  NULL: all instances have dimension p again.  Not with latent variables.
 * ggl_set_lambda1_mask_k: one (p,p) threshold array lambda1 * lambda1_mask PER INSTANCE, (K,p,p) (block_SGL hands every
 * component its own slice of the mask, :447); NULL clears it (the shared mask of ggl_set_lambda1_mask applies again). */
int ggl_set_instance_dims(ggl_ctx *ctx, const int *pk);
int ggl_set_lambda1_mask_k(ggl_ctx *ctx, const double *lam_Kpp_host);
int ggl_scale_X_batch(ggl_ctx *ctx, const double *factor);
int ggl_get_state_k(ggl_ctx *ctx, int k, double *Omega, double *Theta, double *L, double *X);

/* ---- G independent multiple-graph problems on the same S (batched lambda1 x lambda2 grid) -----------
 * What the MAIN LOOP of grid_search (helper/model_selection.py:208-224) solves one (lambda1, lambda2) point after the
 * other: here the ctx stack holds G problems of K/G instances each, problem g in the slots g*K/G .., every problem with
 * its own rho_g, lambda1_g, lambda2_g; one call = one ADMM_MGL iteration (admm_solver.py:179-224) of all of them
 * (batched Omega-/L-step over all G*K/G matrices, one Theta-step launch with a grid-point dimension).  mu1: (K total)
 * per instance; nk: (K/G) per instance of a problem or NULL; out_norms (G,5).  The caller keeps per-problem rho updates
 * and stopping decisions (ggl_scale_X_batch takes one factor per instance; ggl_get_state_k / ggl_snapshot_k per instance).
 * GGL: K/G <= 32; state exactly symmetric. */
int ggl_mgl_batch_step(ggl_ctx *ctx, int G, const double *rho, const double *lambda1, const double *lambda2,
                       int reg, int latent, const double *mu1, const double *nk, double *out_norms);

/* ---- n iterations of a batch per call (round 5): the HOST LOOP of the grid walks in C -----------------------------------
 * The reference's grid walks (helper/model_selection.py:208-224, :619-633) run, per point and iteration, the stopping test
 * (ADMM_stopping_criterion, solver/admm_solver.py:316-331 / single_admm_solver.py:277-291), the residual-balancing rho rule
 * and the dual rescale (admm_solver.py:227-237 / single_admm_solver.py:196-206).  gglasso_amd.batch did that in NumPy between
 * two C calls (~100 us per batch iteration where the device needs 50 us at p <= 64).
 * ggl_batch_decide (host only, no GPU): one iteration's decisions for n points, bit for bit gglasso_amd.batch._decide --
 *   sq (n,5) squared norms; live (n) 0/1; marked (n) 0/1 or NULL (ggl_failed_instances); rho (n) in/out; dims (n);
 *   last (n,4) {r_t, s_t, e_pri, e_dual} (rows of live finite points rewritten); fac (n) out = rho / rho_new;
 *   status (n) out: 0 goes on, 1 converged now, 2 failed (non-finite sums or marked).  Returns #{status != 0}.
 * ggl_sgl_batch_run / ggl_mgl_batch_run: up to n_iters iterations of ggl_sgl_batch_step / ggl_mgl_batch_step with those
 *   decisions and the X rescale in between.  status (n) in/out: 0 live, 1 converged, 2 failed (points that are not 0 on
 *   entry are dragged along without decisions); last (n,4) in/out; fin_iter (n) in/out: it_base + the iteration of this call
 *   a point finished in (1-based).  snap_ctx NULL: return after the first iteration in which a live point converges or
 *   fails.  snap_ctx given (ctx itself, or the ORIGINAL ctx of a compacted batch; snap_index: destination slot of every
 *   instance slot): a finishing point's whole solution is snapshotted there on the device (ggl_snapshot_state_from), a
 *   failed point is then parked (ggl_reset_instance), and the loop goes on until every point is finished, at least
 *   stop_after points are (> 0: the caller may compact), or n_iters.  Return value: iterations run, < 0 on error.
 * ggl_snapshot_state_from: ggl_snapshot_from + Omega and X; ggl_get_snapshots: the snapshot stacks (K,p,p), any NULL --
 *   ONE download per stack at the end of a batch instead of three or four small ones per point. */
int ggl_batch_decide(int n, const double *sq, const unsigned char *live, const unsigned char *marked, double *rho,
                     const double *dims, double tol, double rtol, int update_rho, double *last, double *fac, int *status);
int ggl_sgl_batch_run(ggl_ctx *ctx, int n_iters, double *rho, const double *lambda1, int latent, const double *mu1,
                      const double *dims, double tol, double rtol, int update_rho, double *last, int *status,
                      int *fin_iter, int it_base, ggl_ctx *snap_ctx, const int *snap_index, int stop_after);
int ggl_mgl_batch_run(ggl_ctx *ctx, int G, int n_iters, double *rho, const double *lambda1, const double *lambda2,
                      int reg, int latent, const double *mu1, const double *nk, const double *dims, double tol, double rtol,
                      int update_rho, double *last, int *status, int *fin_iter, int it_base, ggl_ctx *snap_ctx,
                      const int *snap_index, int stop_after);
int ggl_snapshot_state_from(ggl_ctx *ctx, int k_dst, ggl_ctx *src, int k_src);
int ggl_get_snapshots(ggl_ctx *ctx, double *Omega, double *Theta, double *L, double *X);

/* host only: out = {largest K/G of the batched GGL grid (per-element Theta kernel), largest K of the FGL Theta-step (the
 * K-vectors of a tile of pairs live in LDS)}; callers choose between the batched grid and the sequential walk with it */
int ggl_theta_limits(int out[2]);

/* X <- factor * X : dual rescale after a rho update (admm_solver.py:236). */
int ggl_scale_X(ggl_ctx *ctx, double factor);

/* Exit checks of admm_solver.py:284-301: out = {max|Omega-Omega^T|, max|Theta-Theta^T|,
 * max|L-L^T|, min eig(Theta-L), min eig(L)}. */
int ggl_exit_checks(ggl_ctx *ctx, int latent, double out[5]);

/* The same five numbers per instance, out (K,5) (what the per-instance messages of solver/ext_admm_solver.py:290-311 need). */
int ggl_exit_checks_k(ggl_ctx *ctx, int latent, double *out);
/* The same checks as the DECISIONS the reference takes, without the eigenvalues: out[k*5..] = { the three asymmetries,
 * 1 if Theta_k - L_k - shift_tl I is positive definite else 0, 1 if L_k + shift_l I is positive definite else 0 (1 when not
 * latent) } -- two batched Cholesky factorisations instead of two eigendecompositions.  The reference warns when
 * min eig(Theta - L) <= shift_tl (0 for ADMM_MGL / ADMM_SGL, 1e-5 for ext_ADMM_MGL) resp. min eig(L) < -shift_l (1e-5 / 1e-8):
 * the instances whose flag is 0; ggl_exit_checks_k gives the eigenvalues for the message of a warning that has to be printed. */
int ggl_exit_checks_fast_k(ggl_ctx *ctx, int latent, double shift_tl, double shift_l, double *out);

/* ---- ext_ADMM_MGL: Group Graphical Lasso over instances of DIFFERENT dimension -------------------
 * solver/ext_admm_solver.py:18-323 (loop body :196-231, prox_2norm_G / prox_G_inner :394-453, stopping criterion :325-345,
 * KKT residual :347-392); bookkeeping array G as built by helper/ext_admm_helper.py:104-144, checked like check_G :82-102.
 * Layout: the ctx is created with p = max_k p_k; instance k is the leading (p_k,p_k) block of its (p,p) slot and the
 * caller pads the rest of every stack with an identity block for S, Omega, Theta, Lambda and zeros for the duals (a
 * decoupled fixed point of the iteration; the stopping-test sums skip it).  Omega/Theta/L/X0 travel through
 * ggl_set_state / ggl_get_state (X is X0), Lambda and X1 through ggl_ext_set_state / ggl_ext_get_state.
 *   ggl_ext_setup      pk (K) instance dimensions; G (2,L,K) int32 row-major, -1 = the instance does not hold the pair.
 *                      Every (instance, i, j) may be listed once (GGL_E_ARG otherwise: the groups run in parallel).
 *   ggl_ext_admm_step  one iteration; lambda1 (K) per instance (:133-134), out_norms = the five sums of :325-345:
 *                      |Omega|^2+|Lambda|^2, |Theta-L|^2+|Theta|^2, |X0|^2+|X1|^2, |Omega-Theta+L|^2+|Lambda-Theta|^2,
 *                      |Omega-Omega_prev|^2+|Lambda-Lambda_prev|^2 over all instances (this solver has no rho update).
 *   ggl_ext_kkt_residual  the opt-in stopping_criterion='kkt' of the same solver. */
int ggl_ext_setup(ggl_ctx *ctx, const int *pk, const int *G, int L);
int ggl_ext_set_state(ggl_ctx *ctx, const double *Lambda, const double *X1);
int ggl_ext_get_state(ggl_ctx *ctx, double *Lambda, double *X1);
int ggl_ext_admm_step(ggl_ctx *ctx, double rho, const double *lambda1K, double lambda2, int latent,
                      const double *mu1, double out_norms[5]);
int ggl_ext_kkt_residual(ggl_ctx *ctx, double rho, const double *lambda1K, double lambda2, int latent,
                         const double *mu1, double *out);
/* nprob independent ext_ADMM_MGL problems with the SAME instance dimensions and bookkeeping array in one ctx -- the
 * (lambda1, lambda2) points the MAIN LOOP of grid_search (helper/model_selection.py:208-224) hands to ext_ADMM_MGL one after
 * the other: the ctx stack holds nprob * K instances, problem g in the slots g*K .. (the caller replicates the padded S).
 *   ggl_ext_setup_batch  pk (K) and G (2,L,K) of ONE problem
 *   ggl_ext_batch_step   one iteration of all problems: lambda1 (nprob*K) per instance slot, lambda2 (nprob) per problem, one
 *                        rho (this solver has no rho update), mu1 (nprob*K); out_norms (nprob,5): the five sums per problem.
 * The caller keeps the stopping decision per problem (ggl_get_state_k / ggl_ext_get_state per instance slot). */
int ggl_ext_setup_batch(ggl_ctx *ctx, int nprob, const int *pk, const int *G, int L);
int ggl_ext_batch_step(ggl_ctx *ctx, double rho, const double *lambda1K, const double *lambda2G, int latent,
                       const double *mu1, double *out_norms);

/* Model selection over a batch of independent problems (reference: single_grid_search,
 * helper/model_selection.py:505-692; criteria :812-856; robust_logdet :884-894).
 * ggl_snapshot_k keeps a device copy of instance k's Theta at the moment the host loop declares it converged;
 * ggl_selection_stats then returns, for every instance's snapshot,
 * out[k*4..] = { <S_k,Theta_k>, log det Theta_k (-inf if lambda_min <= 1e-12), count_nonzero(Theta_k),
 * lambda_min(Theta_k) }.  After a latent step ggl_snapshot_k keeps L_k as well.
 *
 * ggl_threshold_scan: tune_threshold (helper/model_selection.py:707-737, thresholding :698-705) for every snapshot at
 * once: out[(k*ntau + j)*4..] = the same four numbers for T = Theta_k with the off-diagonal entries |t| <= tau[j]
 * zeroed.  Thresholds that zero the same entries of an instance give the same matrix; only the distinct ones go
 * through the eigenvalue kernel (*n_eig = how many, may be NULL).
 *
 * ggl_selection_rank: numpy.linalg.matrix_rank of every L_k snapshot (model_selection.py:256, :638):
 * out[k*4..] = { #{|lambda_i| > rel_tol * max|lambda|}, max|lambda|, largest |lambda| not counted, smallest |lambda|
 * counted }; rel_tol <= 0: numpy's p*eps.  The two neighbours of the cut let the caller see how clear the decision
 * was: the null space of an L that came out of the sign-iteration L-step (p > GGL_JACOBI_MAX_P) carries that
 * iteration's residual (~1e-12 |L|), far above numpy's tolerance but far below any eigenvalue the prox keeps. */
int ggl_snapshot_k(ggl_ctx *ctx, int k);
/* the same from another ctx of the same dimension (a compacted batch, ggl_ctx_create_subset): instance ks of src into slot kd */
int ggl_snapshot_from(ggl_ctx *ctx, int kd, ggl_ctx *src, int ks);
int ggl_selection_stats(ggl_ctx *ctx, double *out);
int ggl_threshold_scan(ggl_ctx *ctx, const double *tau, int ntau, double *out, int *n_eig);
int ggl_selection_rank(ggl_ctx *ctx, double rel_tol, double *out);
/* The latent component a solve RETURNS (reference: prox_rank_norm, solver/ggl_helper.py:29-36, as the last L-step of
 * admm_solver.py:197-205 / single_admm_solver.py:172-175 leaves it: Q diag(max(d - mu1/rho, 0)) Q^T, null space exact to
 * rounding -- the reference's callers apply numpy.linalg.matrix_rank to it, helper/model_selection.py:254, :638).
 * Above GGL_JACOBI_MAX_P the per-iteration L-step is the sign iteration, whose L is entrywise right to ~1e-13 |L| but whose
 * null space carries that residual.  ggl_finalize_L rebuilds L from ONE eigendecomposition of the last L-step's input
 * C = Theta - X - Omega (kept on the device by the step itself / by ggl_snapshot_k) -- the reference's own L-step, run
 * once per solve instead of once per iteration.  X is left as the last dual update wrote it.
 *   which 0: the live iterate's L (call before ggl_get_state / ggl_exit_checks);  1: the L snapshots of ggl_snapshot_k
 *   rank_out (K ints, may be NULL): #{ eigenvalues of C_k above mu1_k/rho } for every rebuilt instance, -1 otherwise
 * Returns the number of instances rebuilt -- 0 when every L already is an eigendecomposition's (p <= GGL_JACOBI_MAX_P,
 * GGL_OPT_RANK_EIG, a fallback in the last step, an uploaded L) and nothing was done -- or an error code < 0.
 * ggl_get_snapshot_k: Theta and L (either may be NULL) of instance k's snapshot. */
int ggl_finalize_L(ggl_ctx *ctx, int which, int *rank_out);
/* Fault isolation and compaction of a batch of independent problems (the reference's grid walk, helper/model_selection.py:
 * 208-224, solves its points one by one: a point that fails costs that point, a point that has converged costs nothing more).
 *   GGL_OPT_ISOLATE       see above; ggl_failed_instances: out[k] = 1 for every instance marked so far (out may be NULL),
 *                         returns their number
 *   ggl_reset_instance    parks instance k on the identity problem (S = Omega = Theta = I, L = X = 0): finite, cheapest schedules
 *   ggl_ctx_create_subset a new ctx with the m instances idx[] of src (S, iterate, masks, dimensions, options; device to
 *                         device), for the points still iterating once a good part of the batch is done; src stays valid */
/* Debugging aid, process-wide: after ggl_debug_poison(1) every ctx created fills its device and pinned buffers with 0xFF bytes
 * (NaN doubles, -1 ints) instead of zeros before it initialises what it documents as initialised -- a read of memory the library
 * never wrote then fails at once instead of depending on what an earlier allocation left behind.  2..255: that byte instead
 * (0x7F: 1.4e306, 0x47: 1.5e35 -- finite garbage, which a max / min reduction or a comparison keeps where it drops a NaN).
 * tests/conftest.py switches it on for a whole session when GGL_DEBUG_POISON=<n> is set in the environment of the TEST process. */
int ggl_debug_poison(int on);
/* Process-wide switch: odd p on the direct-to-LDS product kernel (1, default since round 6) or on the register-staged kernel as
 * in rounds 1-5 (0); returns the previous setting.  Same products either way (parity test of the two routes, A/B runs). */
int ggl_set_odd_dl(int on);
int ggl_failed_instances(ggl_ctx *ctx, int *out);
/* Why instance k was marked (the first mark stays): out[0] = 1 a spectral or norm bound that was not finite or not positive,
 * 2 an eigensolver that did not converge, 3 a non-finite residual or trace in the L-step's sign iteration, 4 marked inside a
 * fused batch iteration, 0 not marked; out[1] = the offending value (the bound, the solver's info, the residual).  What the
 * batch drivers put into the warning that goes with a 'solver error' point. */
int ggl_failed_reason(ggl_ctx *ctx, int k, double out[2]);
int ggl_reset_instance(ggl_ctx *ctx, int k);
int ggl_ctx_create_subset(ggl_ctx *src, const int *idx, int m, ggl_ctx **out);
int ggl_get_snapshot_k(ggl_ctx *ctx, int k, double *Theta, double *L);
/* Omega and X of instance k's snapshot (ggl_snapshot_state_from), either may be NULL: what a grid walk fetches for the ONE point it
 * selects after it has taken Theta (and L) of all points with ggl_get_snapshots(NULL, Theta, L, NULL) -- the reference's
 * single_grid_search returns the whole sol of the best point only (helper/model_selection.py:652-660). */
int ggl_get_snapshot_state_k(ggl_ctx *ctx, int k, double *Omega, double *X);
/* Objective pieces for measure=True (admm_solver.py:213): out = {sum_k -logdet Omega_k,
 * <Omega,S>, P_val(Theta)} (ggl_helper.py:266-270,162-176). */
int ggl_objective(ggl_ctx *ctx, double lambda1, double lambda2, int reg, double out[3]);

/* KKT residual of admm_solver.py:333-371 / single_admm_solver.py:293-320 on the ctx state
 * (opt-in stopping_criterion='kkt').  The dual passed by the reference is rho*X. */
int ggl_kkt_residual(ggl_ctx *ctx, double rho, double lambda1, double lambda2, int reg, int latent,
                     const double *mu1, const double *nk, double *out);

/* ---- per-phase device timing (HIP events on the ctx stream; what bench.py's roofline uses) ------
 * Phases of one iteration; ms[] accumulates elapsed milliseconds, count[] the number of launches. */
#define GGL_PH_FORM_W 0       /* W = Theta - L - X - (nk/rho) S                                  */
#define GGL_PH_EIG_OMEGA 1    /* eigensolver of the Omega-step (LDS Jacobi: includes phiplus)     */
#define GGL_PH_RECON_OMEGA 2  /* Q phip(D) Q^T reconstruction (FP64 MFMA)                          */
#define GGL_PH_THETA 3        /* Theta-step (+ fused dual update and norms when not latent)       */
#define GGL_PH_EIG_L 4        /* eigensolver of the L-step                                        */
#define GGL_PH_RECON_L 5      /* Q max(D-mu,0) Q^T                                                 */
#define GGL_PH_DUAL 6         /* X += Omega - Theta + L and norms (latent path)                   */
#define GGL_PH_REDUCE 7       /* partial-sum reduction of the norms                                */
#define GGL_PH_EIG_OMEGA2 8   /* second part of the Newton-Schulz Omega-step (after the spectral-bound sync) */
#define GGL_PH_BOUND 9        /* (not recorded any more: the bound kernels run inside the Omega-step chains) */
#define GGL_PH_ALLREDUCE_GROUPSQ 10 /* K-sharded run: ncclAllReduce of the p(p+1)/2 + 1 packed group sums (ggl_admm_step_sharded)  */
#define GGL_PH_ALLREDUCE_NORMS 11   /* K-sharded run: ncclAllReduce of the five sums                                    */
#define GGL_NPHASE 12
/* on: 0 off, 1 every phase, 2 only GGL_PH_EIG_OMEGA / _OMEGA2 / _EIG_L and the two all-reduces (4-10 event records per iteration) */
int ggl_profile_enable(ggl_ctx *ctx, int on);
int ggl_profile_read(ggl_ctx *ctx, double ms[GGL_NPHASE], long long count[GGL_NPHASE], int reset);
/* Newton-Schulz statistics since ctx creation: Omega-step {calls, steps, calls that took the stable
 * schedule, algorithmic work in units of K*p^3 flop, kernel launches}; L-step {calls, retries at the
 * finer resolution, fallbacks to the eigendecomposition, kernel launches in units of K*p^3 flop (a launch on a compact
 * sub-batch of m instances counts m/K)}; speculative
 * Omega-steps {taken, failed validation and repeated}; [11] end-of-iteration polls that timed out and fell back to a
 * stream synchronisation; what the LAST matrix-function step dispatched: [12] concurrent parts, [13] product-kernel
 * variant (csrc/gemm_sym.hip); [14] Omega-steps that fell back to the eigendecomposition; [15] pre-launched Omega-step
 * chains (GGL_OPT_PIPELINE) that were dropped unused. */
int ggl_ns_stats(ggl_ctx *ctx, long long out[16]);
/* The LDS-resident Omega-step (GGL_OPT_OMEGA_LDS): out = { launches, launches repeated on the launch chain because an instance
 * fell outside the kernel's range, products summed over all instances of all launches, Newton-Schulz steps likewise }. */
int ggl_lds_stats(ggl_ctx *ctx, long long out[4]);
/* GGL_OPT_GROUP_SCHED: out = { Omega-steps that ran as groups with their own schedules, groups of the last step (1 = whole batch),
 * lengths of its groups [4], product units (A', B' included) of their schedules [4], grouped steps whose split differed from
 * the previous grouped step's }; units_sum (4 doubles, may be NULL): the units of every group slot summed over the grouped steps. */
int ggl_group_stats(ggl_ctx *ctx, long long out[11], double *units_sum);
/* c_k >= lambda_max(W_k^2 + 4 beta_k I) and beta_k (K doubles each) of the last validated matrix-function Omega-step; returns 1,
 * or 0 when there is none yet. */
int ggl_spectral_bounds(ggl_ctx *ctx, double *c_out, double *beta_out);
/* Pipelining across iterations (GGL_OPT_PIPELINE, ggl_admm_step): out = { whole Omega-step chains launched ahead of the caller's
 * next step (after an iteration was validated, while the caller looks at its residuals), of those forgotten because rho changed,
 * early first parts (tables, W, A', B' of the NEXT iteration's chain, put into the stream before the host waits for this
 * iteration's residuals), of those continued, fresh streams the concurrency probe had to try before the first two-part
 * Omega-step until one ran BESIDE the ctx's main stream (HIP streams share a small pool of hardware queues, and two streams on
 * one queue serialise; 0: the part stream was fine, -1: not probed yet), Theta-steps that also wrote the next Omega-step's W
 * (GGL_OPT_FUSED_W), of those used by the early first part that followed, bound validations that rode in a product launch
 * (GGL_OPT_CW_RIDER; counted per part), table transfers that rode in an A' launch (GGL_OPT_COPY_RIDER; per part), norm
 * reductions that did (GGL_OPT_REDUCE_RIDER) }. */
int ggl_pipeline_stats(ggl_ctx *ctx, long long out[10]);
/* L-step (sign iteration): out = { calls, calls whose first pass was continued on a compact sub-batch, instances continued in
 * total, calls that fell back to the eigendecomposition } */
int ggl_rank_stats(ggl_ctx *ctx, long long out[4]);
/* out = { L-step calls whose first pass was followed by the deflation, instances that had directions to deflate } */
int ggl_deflate_stats(ggl_ctx *ctx, long long out[2]);
/* ggl_finalize_L checks its eigendecomposition (the eigenvalues of every rebuilt instance add up to the trace of its L-step input,
 * taken before the eigensolver overwrites it) and repeats it once on a kept copy if they do not: out = { eigendecompositions
 * run, of those repeated }. */
int ggl_finalize_stats(ggl_ctx *ctx, long long out[2]);
/* What ran last: out = { concurrent parts and product-kernel variant of the last matrix-function step (as in ggl_ns_stats),
 * code of the Theta kernel of the process's last Theta-step (0 GGL tile pairs; 100+KMAX per-element kernel with the K-column
 * in one thread; 100*KQ+NW per-element kernel with the K-column over NW waves: 404, 408, 808, 816, 1616; 2000+tile FGL
 * Condat tiles), eigendecompositions ggl_finalize_L ran on this ctx }.  The parity tests assert the dispatch with it. */
int ggl_last_dispatch(ggl_ctx *ctx, long long out[4]);
/* Event timeline of ggl_admm_step's iterations without a profiler (round 5; tools/event_timeline.py): after ggl_trace_start every
 * launch of the iteration is followed by an event on its stream and the host notes its own marks; ggl_trace_read stops the
 * recording and returns rows {kind 0 device / 1 host, lane (0 main stream, 1.. part streams; -1 host), tag, microseconds since the
 * start}.  Device rows give the COMPLETION time of the launch they follow.  Device tags: 1 parameter copy, 2 form_W, 3 bound_rows,
 * 4 cw_final, 10 product, 11 pair of products, 20 Theta-step, 21 norm reduction; host tags: 100 step entered, 101 Theta-step and
 * reduction queued, 102 early first part queued, 103 residuals seen, 104 rest of the next chain queued (the step returns). */
int ggl_trace_start(ggl_ctx *ctx, int max_events);
int ggl_trace_read(ggl_ctx *ctx, double *out, int cap);
/* per-instance status of the last eigensolver launch a step fetched: sweeps of the LDS Jacobi kernel (-1: not converged) or
 * rocSOLVER's info */
int ggl_eig_info(ggl_ctx *ctx, int *out);

/* ---- kernel-level test / measurement entry points (not used by the solvers) --------------------
 * ggl_dev_symm: one launch of the symmetric-product kernel on host data (kernel unit test; variant < 0 = by size).
 * ggl_dev_symm_bench: average milliseconds of `iters` launches on random device data (HIP events).
 * Both return GGL_E_ARG for a variant that is not in this build. */
int ggl_dev_symm(int K, int p, const double *A, const double *B, const double *E, const double *coef5K,
                 double *C, double *C2, int variant);
int ggl_dev_symm_bench(int K, int p, int variant, int iters, double *ms_out);
/* ggl_dev_omega_lds: the Omega-step of small matrices (p <= 64) as ONE launch, one workgroup per instance, the Newton-Schulz
 * chain resident in LDS, bound and schedule chosen on the device (omega_lds.hip; kernel unit test and timing).
 * Omega = phiplus(Theta - L - X - beta S, beta), L may be NULL.  cbound (K) or NULL receives the bound used.
 * out[14] = {ms per launch, fallback flag, products summed over the instances, entries of the schedule table, [4..12] phase
 * stamps of instance 0 in us (start, W, A', B', bound, first step, steps, W again, Omega), [13] products of instance 0}.
 * degrees + 1000 * w: w = 4 or 8 waves per workgroup (0: by size, as the solver runs it). */
int ggl_dev_omega_lds(int K, int p, const double *Theta, const double *L, const double *X, const double *S, const double *beta,
                      double tol, int degrees, double *Omega, double *cbound, int iters, double *out);
/* ggl_dev_symm_bounds: C = A B on the direct-to-LDS product kernel with the bound partials of its epilogue, reduced to
 * the row sums of |C| (K,p), |C|_F^2 (K) and the spectral bound sqrt(min(|C|_inf, Collatz-Wielandt ratio, |C|_F)) (K)
 * that the Omega-step takes from B' = (W^2 + 4 beta I)^2 (kernel unit test; even p, variants 16 / 17 / 20). */
int ggl_dev_symm_bounds(int K, int p, const double *A, const double *B, int variant, double *C,
                        double *rowsum_out, double *fro2_out, double *bound_out);
/* host only (no GPU needed): the Newton-Schulz step schedule the Omega-step would run for a spectrum in [l,1]
 * (x = sqrt(eig(Z Y))).  degrees 3 = cubic steps, 5 = cubic/quintic mix, 9 = cubic/quintic/degree-nine mix.
 * deg_out[max_steps] receives 3, 5 or 9 per step, coef_out[max_steps*6] = {t0..t4,l_after} of
 * x -> x (t0 + t1 x^2 + ... + t4 x^8), *units_out the number of
 * symmetric products of the stack (incl. A' and B').  degrees + 100: the schedule of the L-step's sign iteration
 * instead (every step costs X^2, [t], X t: 2 / 3 / 4 products; *units_out without the first and the last product).
 * Returns the number of steps or GGL_E_ARG. */
int ggl_dev_ns_schedule(double l, int degrees, int max_steps, int *deg_out, double *coef_out, int *units_out);
/* the same for the Omega-step at a stopping tolerance (GGL_OPT_NS_TOL; 0 = fp64 resolution, what the call above plans for) */
int ggl_dev_ns_schedule_tol(double l, int degrees, double tol, int max_steps, int *deg_out, double *coef_out,
                            int *units_out);
/* host only: the grouping rule of GGL_OPT_GROUP_SCHED -- units[k] = product count of instance k's schedule; returns the number of
 * contiguous groups (1 = whole) and their lengths in len_out[3] -- and the product count (A', B' included) of the all-symmetric
 * schedule for a spectrum in [l, 1] at the stopping tolerance tol (-1: condition number above 300, the stable schedule's range) */
int ggl_dev_group_partition(const int *units, int K, int p, int max_groups, int *len_out);
int ggl_dev_ns_units(double l, int degrees, double tol);
#ifdef GGL_DEV
/* C = A B on the INT8 matrix cores from S signed-digit slices per operand (error-free split; gemm_i8.hip), slice pairs
 * t + u <= dmax; |A| <= scaleA, |B| <= scaleB entrywise.  ms_out = {slicing both operands, one product launch, overflow flag}. */
int ggl_dev_i8_stages(int n);   /* LDS stages of the int8 product kernel: 1 (default) or 2 */
int ggl_dev_symm_i8(int K, int p, int S, int dmax, const double *A, const double *B, double scaleA, double scaleB, double *C,
                    int iters, double *ms_out);
/* The whole Omega-step phiplus(W) (solver/ggl_helper.py:272-303) on the int8 matrix cores, stand-alone: W (K,p,p), beta (K),
 * cbound (K) >= lambda_max(W^2 + 4 beta I); cfg5 = {slices of the full products, of F F, of G F^2, of Y E, diagonal cut of Y E}
 * or NULL; tol: the schedule's stopping tolerance.  ms_out = {ms per step, product launches, overflow flag, fp64 products the
 * schedule stands for}. */
int ggl_dev_omega_i8(int K, int p, const double *W, const double *beta, const double *cbound, const int *cfg5, double tol,
                     double *Omega, int iters, double *ms_out);
/* libggl_hip_dev.so only (python -m gglasso_amd.build --dev): measured FP64 matrix-core ceiling of this GPU in TFLOP/s
 * (MFMA-only probe kernel); per-workgroup timestamps {start, loop begin, loop end, end, XCC id} of one launch of the
 * 64x64 kernel */
int ggl_dev_mfma_f64_peak(double *tflops_out);
/* FP64 VALU || FP64 MFMA co-issue probe (csrc/probes_dev.hip), TF/s: {MFMA only, v_fma_f64 only, waves split per SIMD:
 * MFMA, DFMA, one wave with 4 / 8 / 16 / 32 v_fma_f64 behind every MFMA: MFMA, DFMA each} */
int ggl_dev_coissue_probe(double *out12);
/* MFMA fed from LDS (the product kernel's inner loop alone), TF/s for wave tiles {2x2 @3 WG/CU, 2x2 @5, 2x4 @3, 4x4 @2, 4x4 @3, 1x1 @5} */
int ggl_dev_mfma_lds_probe(double *out6);
/* k_omega_chain on a synthetic chain of nprod dependent products X <- I - 1.5 X^2: out = {ms as nprod launches, ms as one
 * persistent launch, persistent workgroups, max |difference| of the results, completion flag, done counters [K]} */
int ggl_dev_chain_run(int K, int p, int nprod, int iters, double *out);
/* What a change of kernel between dependent launches costs (tools/kernel_switch_cost.py): mode 0 `iters` product launches, 1 each
 * followed by a one-thread kernel, 2 the one-thread kernel alone, 3 each followed by an elementwise kernel over its output;
 * ms per repetition. */
int ggl_dev_switch_bench(int K, int p, int variant, int iters, int mode, double *ms_out);
/* A measured ceiling for the product kernel: the vendor's FP64 GEMM at the same shapes (tools/bench_vendor.py).  mode 0
 * rocblas_dgemm_strided_batched (N,N), 1 (T,N), 2 rocblas_dsyrk_strided_batched (one triangle), 3 rocblas_dsyrkx_strided_batched
 * (one triangle of A B^T: the library's form of a symmetric product), 4 the product kernel itself.  ms per call. */
int ggl_dev_vendor_bench(int K, int p, int mode, int iters, double *ms_out);
/* Probe for reading (A) of round 5's intermittent RANK table: hipMalloc + hipMemsetAsync of a (K,p,p) stack and
 * hipMemcpyAsync device-to-device copies of `slices` instances into it on one stream, `reps` times; out = { repetitions, slices
 * that came out zero }. */
int ggl_dev_fill_copy_probe(int reps, int K, int p, int slices, int gap_us, long long out[2]);
int ggl_dev_symm_timeline(int K, int p, long long *out, int max_blocks, int *nblocks_out);
/* persistent-chain probe: nprod dependent products X <- X X of a K-batch as nprod launches (out[0], ms) and as ONE cooperative
 * launch with grid-wide barriers between the products (out[1], ms); out[2] grid of the latter, out[3] max |difference| of
 * the two chains' results (must be 0), out[4] barrier flag (1 time-out, 2 a workgroup not on XCD blockIdx % 8).  variant: 16, 17
 * (64x64 tiles), 20 (32x32).  two_level: one L2 write-back per XCD and barrier instead of one per workgroup. */
int ggl_dev_chain_probe(int K, int p, int variant, int nprod, int iters, int two_level, double *out);
#endif

/* ---- stateless operator entry points (host buffers; used for operator-level parity) ---------- */
/* numpy.linalg.eigh on a stack (lower triangle read); D (K,p) ascending, Q (K,p,p) columns. */
int ggl_eigh_batched(int K, int p, const double *A, double *D, double *Q, int eig_method);
/* phiplus(beta,D,Q), ggl_helper.py:280-303, for K matrices; beta (K,). */
int ggl_phiplus(int K, int p, const double *beta, const double *D, const double *Q, double *out);
/* prox_rank_norm(A,beta,D,Q), ggl_helper.py:29-36. */
int ggl_prox_rank_norm(int K, int p, const double *beta, const double *D, const double *Q, double *out);
/* eigh + phiplus / rank shrink fused, straight from the matrix (what the ADMM step runs). */
int ggl_phiplus_matrix(int K, int p, const double *beta, const double *W, double *out, int eig_method);
int ggl_rank_matrix(int K, int p, const double *beta, const double *C, double *out, int eig_method);
/* the same with the two-tier control of the sign-iteration L-step exposed: l0_coarse as GGL_OPT_RANK_L0_COARSE (< 0: the default),
 * stats (may be NULL) = { calls, calls continued on a compact sub-batch, instances continued, eigendecomposition fallbacks,
 * retries of the whole batch, product launches } */
int ggl_rank_matrix_ex(int K, int p, const double *beta, const double *C, double *out, int eig_method, double l0_coarse,
                       long long stats[6]);
/* the same with the DEFLATING L-step (a first pass at l0_deflate, <= 0: the default GGL_OPT_RANK_L0_DEFLATE, then the
 * eigenvalues next to the threshold as the range of I - X^2; csrc/deflate.hip); ggl_rank_matrix_ex with l0_coarse >= 0 runs
 * the two-tier iteration without it.  stats[8] = the six above + { calls followed by the deflation, instances deflated } */
int ggl_rank_matrix_deflate(int K, int p, const double *beta, const double *C, double *out, int eig_method, double l0_deflate,
                            long long stats[8]);
/* prox_od_1norm(A,l), ggl_helper.py:16-27; lam_pp NULL => scalar lam. */
int ggl_prox_od_1norm(int p, const double *A, double lam, const double *lam_pp, double *out);
/* prox_p(X,l1,l2,reg), ggl_helper.py:190-207 (reg = GGL_REG_GGL | GGL_REG_FGL). */
int ggl_prox_p(int K, int p, const double *X, double l1, double l2, int reg, double *out);
/* n independent K-vectors, Y (n,K) row-major: prox_tv = condat_method (fgl_helper.py:11-68),
 * prox_2norm (ggl_helper.py:38-43), prox_phi_ggl / prox_phi_fgl. */
int ggl_prox_tv(int n, int K, const double *Y, double lam, double *out);
int ggl_prox_2norm(int n, int K, const double *Y, double lam, double *out);
int ggl_prox_phi(int n, int K, const double *Y, double l1, double l2, int reg, double *out);

#ifdef __cplusplus
}
#endif
#endif
