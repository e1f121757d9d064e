// Internals shared by the translation units of the C ABI of libggl_hip.so (include/ggl_hip.h): the ctx, the error / profiling
// macros and the host-side helpers that more than one of capi_*.hip calls.  Round 6 split the former ggl_capi.hip (5 700
// lines) by subject:
//   capi_ctx.hip        ctx, arenas / stream pool, options, state upload / download
//   capi_omega.hip      the Omega-step pipeline (speculation, parts, groups, riders), Theta-step, ggl_admm_step
//   capi_lstep.hip      the L-step (sign iteration, deflation, continuation) and ggl_finalize_L
//   capi_batch.hip      batches of independent problems: steps, the loop in C, isolation, compaction
//   capi_snapshots.hip  per-instance snapshots of finished points
//   capi_checks.hip     exit checks, objective, KKT, selection statistics, thresholds, rank
//   capi_stats.hip      counters, profiling, event timeline
//   capi_comm.hip       RCCL behind the C ABI (K-sharded step)
//   capi_ext.hip        ext_ADMM_MGL (instances of different dimension)
//   capi_ops.hip        stateless operator entry points, development probes
#pragma once
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <atomic>
#include <thread>
#include <mutex>
#include <chrono>
#include <vector>

#include "../../include/ggl_hip.h"
#include "kernels.hpp"
#include "ggl_comm.hpp"

using namespace ggl;


#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            (void)hipGetLastError();   /* the runtime keeps a failed call's code as the thread's "last error": taken here, or  \
                                          the next launch check of an unrelated, valid call reports it (tools/abi_misuse.py) */ \
            return fail(GGL_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
        }                                                                                              \
    } while (0)

#define ARGCHK(cond, msg)                                       \
    do {                                                        \
        if (!(cond)) return fail(GGL_E_ARG, "bad argument: %s", msg); \
    } while (0)

struct ggl_ctx {
    int device = 0, K = 0, p = 0, flags = 0, eig = 0;
    size_t n = 0;   // K*p*p
    hipStream_t stream = nullptr;
    bool own_stream = false;
    size_t arena_tot[3] = {0, 0, 0};                 // sizes of the three arenas (reuse across ctxs: pool_take_arenas)
    rocblas_handle blas = nullptr;
    double *S = nullptr, *Om[2] = {nullptr, nullptr}, *Theta = nullptr, *L = nullptr, *X = nullptr, *W = nullptr;
    int cur = 0;              // Om[cur] is Omega_t, Om[cur^1] is Omega_{t-1}
    double *DvO = nullptr, *DvL = nullptr, *scale = nullptr, *E = nullptr;   // (K,p), (K,p), (2,K,p), (K,p)
    int* info = nullptr;      // (K)
    double* par = nullptr;    // device: beta[K] | l1[K] | mu[K] | nk[K] | 1/rho[K] | X scale[K] | l2[K] | spare
    double* par_h = nullptr;  // pinned mirror
    double *mask = nullptr, *groupsq = nullptr;   // (p,p)
    double* snap[4] = {nullptr, nullptr, nullptr, nullptr};   // device copy of a start point (ggl_state_snapshot), lazy
    bool snap_symmetric = true;
    double* maskK = nullptr;                      // (K,p,p) per-instance thresholds (ggl_set_lambda1_mask_k), lazy
    bool has_maskK = false;
    int* inst_pk = nullptr;                       // (K) instance dimensions of a padded batch of single problems, lazy
    bool has_dims = false;
    double* sqwork = nullptr;                     // (ggl_chunks, p, p) per-chunk sums of squares
    bool has_mask = false;
    double* partials = nullptr;
    double* partials_own = nullptr;               // a partials buffer grown beyond the arena's (ensure_partials)
    size_t partials_len = 0;
    double *norms = nullptr, *norms_h = nullptr;  // (K,8) device / pinned
    int* info_h = nullptr;                        // pinned (K)
    double* gflag_h = nullptr;                    // pinned: the all-reduced speculation flag of a K-sharded step
    unsigned* arrive = nullptr;                   // device: arrival counter of a multi-row norm reduction that publishes seq
    // event timeline of iterations without a profiler (ggl_trace_start / ggl_trace_read): an event behind every launch of the
    // step on its stream, host marks beside them
    struct Trace {
        bool on = false;
        int cap = 0, n = 0, nhost = 0;
        hipEvent_t base = nullptr;
        std::vector<hipEvent_t> ev;
        std::vector<int> tag, lane;              // lane: 0 main stream, 1.. part / side streams
        std::vector<double> host_us;             // host marks: microseconds since the base event completed
        std::vector<int> host_tag;
        std::chrono::steady_clock::time_point t0;
    } trace;
    // K independent single problems at p <= 64 (ggl_sgl_batch_step): the LDS-resident Omega-step goes on with the Theta-step
    // (omega_lds.hip, LdsSgl) -- sgl_req: what the caller asks omega_step for; sgl_done: the fused form was launched
    const LdsSgl* sgl_req = nullptr;
    bool sgl_done = false;
    bool pending_beta_only = false;               // the caller's pending parameter transfer holds beta (slot 0) and nothing else
    bool pending_pinned_ok = false;               // ... or more, and every kernel that reads it this step can take the pinned mirror
    bool lds_pinned = true;                       // GGL_OPT_LDS_PINNED: the LDS kernels read their parameters from the pinned mirror
    bool par0_stale = false;                      // the device copy of beta was skipped (the LDS kernel read the pinned mirror)
    int* sgl_fail_h = nullptr;                    // pinned (K): instances the fused kernel could not serve
    long long sgl_fused_calls = 0, sgl_fallback_instances = 0;
    bool nk_valid = false;
    // Newton-Schulz Omega-step (newton_schulz.hip)
    bool omega_ns = false;
    bool dvo_valid = false;                    // DvO holds the eigenvalues of the last Omega-step
    int symm_variant = -1;
    double *nsYP[2] = {nullptr, nullptr}, *nsT = nullptr;   // [Y|Z] scratch pairs (2 stacks each), T
    // the Omega-step's product chain as ONE persistent launch with per-instance dependencies (k_omega_chain, gemm_sym.hip)
    // Measured-and-rejected alternatives (each built, bit-identical or parity-tested, and slower or
    // within noise) are options of the DEVELOPMENT library only (round 6): in the product build the switches are constants,
    // the compiler drops their branches, and ggl_ctx_set_option refuses them.  GGL_DEV_OPT(type, name): a field there, 0 here.
#ifdef GGL_DEV
#define GGL_DEV_OPT(type, name) type name = 0
#else
#define GGL_DEV_OPT(type, name) static constexpr type name = 0
#endif
    GGL_DEV_OPT(int, chain_mode);              // GGL_OPT_CHAIN: 0 never, 1 where chain_tile() says so, 2 wherever it can run
    double* nsNX = nullptr;                    // third [Y|Z] pair: the chain leaves A', B' intact for the bound kernels (lazy)
    unsigned* chain_cnt = nullptr;             // per-instance completion / ticket words, one 128-byte line each (lazy)
    long long chain_calls = 0;
    bool flags_dirty = false;
    int step_latent = 0;                       // latent flag of the last Omega-step (the split entry points that follow it)                  // a validation flag was raised: clear ALL device slots before the next step
    int ns_force = 0;                          // 0 auto, 1 symmetric products, 2 stable products
    bool use_syevj = false;
    static constexpr int MAX_PARTS = 4;
    hipStream_t streamx[MAX_PARTS - 1] = {};   // extra streams: the parts of the batch run their
    hipEvent_t ev_fork = nullptr, ev_join[MAX_PARTS - 1] = {};   // Newton-Schulz launch sequences concurrently
    // GGL_OPT_BOUND_SIDE: the bound kernels that validate a speculative step's assumed bound run on a side stream beside the
    // chain's first products (per part: fork after B', join before the first launch that overwrites B')
    GGL_DEV_OPT(int, bound_side);              // 0 off, 1 on, 2 by regime (two concurrent parts of a large batch)
    // GGL_OPT_JOIN_FLAG: the parts of a speculative chain are joined through flag words in device memory (k_set_flag /
    // k_wait_flags) instead of a cross-queue event wait
    bool join_flag = true;
    int red_rider = 2;                         // GGL_OPT_REDUCE_RIDER: 0 off, 1 single launch sequences only, 2 always
    long long red_rides = 0;
    RedRider red_pending;                      // a Theta-step's norm reduction waiting for the early part's A' launch
    int copy_rider = 1;                        // GGL_OPT_COPY_RIDER: 0 off, 1 single launch sequences only, 2 always
    long long copy_rides = 0;
    std::vector<double> pre0_beta;             // [K] the beta the DEVICE's coefficient rows of A' were last written for (NaN: none)
    int cw_rider = 1;                          // GGL_OPT_CW_RIDER: 0 two kernels, 1 rides in the next product launch, 2 the rider's own launch
    long long cw_rides = 0;
    bool parts_serial = false;                 // probe_part_streams found no part stream that runs beside the main one
    unsigned long long* join_words = nullptr;   // device [MAX_PARTS]
    unsigned long long join_seq = 0;
    hipEvent_t ev_bfork[MAX_PARTS] = {}, ev_bjoin[MAX_PARTS] = {};
    // Pipelining across iterations: right after an iteration has been validated, ggl_admm_step launches the NEXT iteration's
    // (speculative) Omega-step chain for the same beta before it returns, so the GPU works through the host's round trip
    // (norms -> rho rule -> next call).  The chain only writes scratch and Omega[cur^1]; it is consumed by the next call if
    // beta is unchanged and dropped otherwise (every other entry point drops it first).
    bool pipeline = true, pre_valid = false, pre_spec_pending = false;
    bool last_step_hint = false;               // ggl_hint_last_step: the next ggl_admm_step is the caller's last one
    double* pre_beta = nullptr;                // host: beta the pre-launched chain was built for (K)
    long long pre_launched = 0, pre_dropped = 0;
    int download_threads = 8;                  // GGL_OPT_DOWNLOAD_THREADS: host threads that touch a download's destination pages first
    GGL_DEV_OPT(int, parts_order);             // GGL_OPT_PARTS_ORDER
    GGL_DEV_OPT(int, parts_bias);              // GGL_OPT_PARTS_BIAS: two concurrent parts take K/2 + bias and K/2 - bias instances
    // GGL_OPT_GROUP_SCHED: a batch whose instances need different product counts (a grid of independent problems) runs as up
    // to three contiguous groups with their own schedules (ns_group_partition, newton_schulz.hip) where the size rule would
    // run it as one launch sequence
    int group_sched = 1;
    long long group_steps = 0;                 // Omega-steps that ran as such groups
    long long group_changes = 0;               // ... whose split differed from the previous grouped step's
    int last_groups = 1, last_group_len[MAX_PARTS] = {}, last_group_units[MAX_PARTS] = {};
    double group_units_sum[MAX_PARTS] = {};    // per group slot: product units summed over the grouped steps
    int parts_small = 8;                       // smallest K (< 16, p >= 384) that is split into two concurrent parts; 0 = never
                                               // (measured at p = 500: K = 8 +7.6 % iterations/s as 4 + 4, K = 4 -2.4 % as 2 + 2)
    bool fused_start = true;                   // speculative step: first step's start matrix as 2nd output of the B' launch
    bool fused_bounds = true;                  // spectral-bound partials from the epilogue of the B' launch (GGL_OPT_FUSED_BOUNDS)
    // small matrices: the whole Omega-step as ONE launch with the chain resident in LDS (omega_lds.hip, GGL_OPT_OMEGA_LDS)
    bool lds_omega = true;
    int lds_waves = 0;                          // waves per workgroup of k_omega_lds: 0 by size, 4, 8 (GGL_OPT_OMEGA_LDS = 4 / 8)
    double* lds_tab = nullptr;                 // device: schedule table + the two counters behind it (lazy)
    int lds_ntab = 0, lds_tab_deg = -1;
    double lds_lnq = 0.0, lds_tab_tol = -1.0;
    long long lds_calls = 0, lds_misses = 0;
    int lds_cool = 0, lds_cool_next = 4;       // launches to sit out after an instance fell outside the kernel's range (doubles per miss)
    bool lds_last = false;                     // the Omega-step launched last was the LDS kernel
    GGL_DEV_OPT(bool, fused_cw);               // k_bound_rows + k_cw_final as ONE launch (GGL_OPT_FUSED_CW): measured, no gain
                                               // (K=4: 4654 / 4892 vs 4602 / 4774 it/s; headline -6 %): opt-in, DESIGN 9.7
    int theta_flat = 2;                        // GGL Theta-step for symmetric states: 0 tile pairs, 1 per-element kernel, 2 per-element with the K-column over four waves
    bool state_symmetric = true;               // X and L exactly symmetric (checked when the state is set)
    bool S_symmetric = false;                  // S exactly symmetric (checked by ggl_set_S)
    // W of the next Omega-step written by this iteration's Theta kernel (GGL_OPT_FUSED_W, launch_theta_pair's WNext): valid for
    // the early first part that follows in the same ggl_admm_step, built for wf_beta
    bool fused_w = true, wf_ready = false;
    double* wf_beta = nullptr;                 // host (K)
    long long wf_written = 0, wf_used = 0;
    // speculative Omega-step: the schedule is built from the PREVIOUS iteration's spectral bounds (inflated) and the
    // products are launched without waiting for this iteration's bounds; a device-side check sets spec_flag when a
    // bound was exceeded, the state-changing kernels of the step then do nothing and the host repeats the step
    bool spec_enable = true, spec_have = false, spec_pending = false;
    double *spec_c = nullptr, *spec_beta = nullptr;   // host: bounds / beta of the last validated step (K each)
    double *cuse = nullptr, *cuse_h = nullptr;        // bounds the running schedule assumes (device / pinned)
    // The pinned plan tables (coef_h, cuse_h) exist twice and every Omega-step that writes a plan takes the other copy: the
    // copy kernel of an EARLY phase A (below) may still be waiting in the stream when the host builds the next plan.
    double *coef_hh[2] = {nullptr, nullptr}, *cuse_hh[2] = {nullptr, nullptr};
    int plan_par = 0;
    // Early phase A (GGL_OPT_PIPELINE, ggl_admm_step): the first part of the NEXT iteration's speculative chain -- parameter
    // tables, W, A', B' (+ the first step's start): scratch only, no validation flags, nothing the repeat of a rejected step
    // needs -- is put into the stream BEHIND this iteration's Theta-step and BEFORE the host waits for its residuals, so the
    // device has ~0.2 ms of work queued across the host's round trip.  The rest of the chain (bound kernels, products,
    // Omega) follows from the SAME plan once the iteration is validated and the rho rule leaves rho alone; otherwise the
    // early part is forgotten (it wrote W and the A'/B' scratch pair only).
    struct EarlyA {
        bool valid = false;
        NsPlan plans[4];
        double* fused[4] = {nullptr, nullptr, nullptr, nullptr};
        double* beta = nullptr;                       // (K) beta the part was built for
        int nh = 0, Kh[4] = {}, k0h[4] = {};          // the split the part was launched with (the rest must use the same)
    } early;
    bool early_part = true;                           // GGL_OPT_EARLY_PART
    GGL_DEV_OPT(int, part_priority);                  // GGL_OPT_PART_PRIORITY
    bool parts_probed = false;                        // streamx[0] has been checked to run concurrently with the main stream
    int parts_replaced = 0;                           // candidates tried by that check (0: the stream was fine)
    bool early_caller = false;                        // set by ggl_admm_step around its Theta-step: the early part may be launched
    bool early_request = false;                       // omega_step: launch phase A only
    bool ratio_calm = false;                          // last validated iteration: residual ratio well inside the rho rule's band
    long long early_launched = 0, early_used = 0;
    int *spec_flag = nullptr, *spec_flag_h = nullptr; // MAX_PARTS validation flags (device / pinned)
    long long spec_calls = 0, spec_misses = 0;
    double spec_factor = 1.02;                 // inflation of the previous bounds (GGL_SPEC_FACTOR; < 1 forces misses)
    unsigned long long* seq_h = nullptr;       // pinned: sequence number published by the last kernel of a step
    unsigned long long seq_next = 0, seq_wait = 0;   // seq_wait != 0: finish_norms may poll instead of synchronising
    unsigned long long stamp_want = 0;         // a single-row reduction carries its sequence number in slot GGL_NNORM of its row as well
    bool spin_wait = true;
    long long spin_timeouts = 0;               // polls that hit GGL_SPIN_LIMIT_MS and fell back to a stream sync
    bool sharded_check = false;                // this step's Theta kernels ran under the all-reduced validation flag
    bool info_dirty = true;                    // an eigensolver wrote `info` since it was last fetched
    bool norms_host = false;                   // the last norm reduction wrote straight into norms_h
    int spec_cool = 0;                         // iterations without speculation left after a failed one
    int ns_degrees = 9;                        // highest Newton-Schulz step degree: 3, 5 or 9
    double ns_tol = NS_TOL_DEFAULT;            // Omega-step schedules end with the spectrum inside [1 - ns_tol, 1]
    long parts_max_tiles = 2048;               // concurrent parts only up to this many 64x64 tile pairs in the batch
    int ns_parts = 1;                          // concurrent launch sequences (parts of the batch) wanted
    int* sweeps = nullptr;
    long long ns_stable_calls = 0;
    int last_parts = 0, last_variant = -1;     // concurrent parts / product-kernel variant of the last matrix-function step
    double *coef = nullptr, *coef_h = nullptr; // [NS_MAX_LAUNCHES][2K][NS_NCOEF]
    double* bounds_h = nullptr;                // pinned: spectral / norm bound per instance, written by k_bound_final

    // K-sharded runs: RCCL communicator of this rank (ggl_comm_init), collectives go on `stream`
    void* comm = nullptr;
    int comm_rank = 0, comm_nranks = 1;
    // ext_ADMM_MGL (instances of different dimension, ggl_ext_*): lazy
    double *Lam[2] = {nullptr, nullptr}, *X1 = nullptr;   // Lambda ping-pong (Lam[lcur] current) and the second dual
    int lcur = 0;
    int *ext_pk = nullptr, *ext_Gt = nullptr, *ext_gsize = nullptr;   // p_k [K]; G transposed [2][K][L]; group sizes [L]
    int ext_L = -1;                            // -1: ggl_ext_setup not called
    int ext_nprob = 1;                         // independent problems in the stack (ggl_ext_setup_batch), K / ext_nprob instances each
    double* snapT = nullptr;                   // per-instance snapshots of Theta (model selection), lazy
    double* snapL = nullptr;                   // ... and of L once a latent step has run
    // ggl_finalize_L: the L a solve RETURNS is rebuilt from one eigendecomposition of the last L-step's input C where that
    // L-step was the sign iteration (whose null space carries the iteration's residual, ~1e-13 |L|, instead of 1e-16 |L|).
    // rank_step keeps that C by swapping W with Ckeep (no copy); ggl_snapshot_k keeps the instance's C beside its L.
    void *arena_dev = nullptr, *arena_pin = nullptr, *arena_pin_coh = nullptr;   // ctx_alloc: everything allocated at creation
    double* Ckeep_alloc = nullptr;             // what hipMalloc returned for Ckeep (Ckeep and W swap NAMES: rank_step)
    double* Ckeep = nullptr;                   // (K,p,p) C = Theta - X - Omega of the last sign-iteration L-step, lazy
    double* Ckeep_beta = nullptr;              // host (K): mu1_k / rho of that step
    bool l_ns = false;                         // L is the sign iteration's (Ckeep valid); false once rebuilt / set / eigh route
    double* snapC = nullptr;                   // (K,p,p) snapshots of C, lazy
    double* snapOm = nullptr;                  // (K,p,p) snapshots of Omega and X (ggl_snapshot_state_from), lazy
    double* snapX = nullptr;
    double* snap_beta = nullptr;               // host (K)
    unsigned char* snap_ns = nullptr;          // host (K): snapshot k's L is a sign-iteration L (snapC_k, snap_beta[k] valid)
    long long finalize_calls = 0;              // eigendecompositions ggl_finalize_L ran
    long long finalize_retries = 0;            // ... that had to be repeated (eigenvalues that did not add up to the trace)
    // GGL_OPT_ISOLATE (batches of independent problems): an instance whose data turn non-finite (a NaN in its S, a diverged
    // iterate) or whose eigensolver does not converge is MARKED instead of failing the call for the whole batch
    // (helper/model_selection.py:208-224 walks the grid point by point and never loses it to one point); the host reads the
    // marks (ggl_failed_instances), reports the point and parks its slots on the identity problem (ggl_reset_instance)
    bool isolate = false;
    unsigned char* failed = nullptr;           // host (K), lazy
    int* fail_why = nullptr;                   // host (K): why the instance was marked first (mark_failed), with
    double* fail_value = nullptr;              // host (K): the offending value
    double* nbrow = nullptr;                   // [K][p] row abs-sums of B' (Collatz-Wielandt weight vector)
    // the Collatz-Wielandt vector carried across iterations (k_cw_final): [cw_cur] was left behind by the last ACCEPTED
    // bound pass, the other one is what the pass in flight writes; cw_have: there is an accepted one
    double* cwvec[2] = {nullptr, nullptr};
    int cw_cur = 0;
    bool cw_have = false, cw_warm = true, cw_pending = false, pre_cw_pending = false;
    // the same for the L-step's norm bound (round 5, GGL_OPT_RANK_CW): |C|_2^2 = lambda_max(C C) <= the Collatz-Wielandt ratio of
    // |C C| for a vector carried across ADMM iterations (lazy buffers, one pair)
    double* cwvecL[2] = {nullptr, nullptr};
    int cwL_cur = 0;
    bool cwL_have = false;
    GGL_DEV_OPT(bool, rank_cw);                  // GGL_OPT_RANK_CW (measured at C4: no gain, see include/ggl_hip.h)
    // bound partials written by the epilogue of the B' product launch (no norm pass over B'): row sums per tile column,
    // Frobenius shares per tile, block maxima of the row sums; merge cells of the Collatz-Wielandt kernel
    double *rowpart = nullptr, *fropart = nullptr, *infpart = nullptr;
    unsigned long long* cwmax = nullptr;
    unsigned* cwcnt = nullptr;
    double* nbpart = nullptr;                  // [K][blocks][2] + [K][blocks]: norm / Collatz-Wielandt partials
    double *maxdev = nullptr, *maxdev_h = nullptr;   // [K] residual of the sign iteration
    bool rank_ns = false;                            // L-step by sign Newton-Schulz (else eigendecomposition)
    bool rank_eig = false;                           // GGL_OPT_RANK_EIG: force the eigendecomposition route
    double rank_l0 = 1e-6;                           // resolution of the scaling schedule
    double rank_l0_coarse = 8e-5;                    // two-tier L-step: resolution of the first pass over the whole batch (0: one tier)
    // deflation of the eigenvalues next to the threshold after a coarse pass (deflate.hip; GGL_OPT_RANK_DEFLATE)
    bool rank_deflate = true;
    double rank_l0_deflate = 2e-3;                   // resolution of the coarse pass the deflation follows
    double *defl_G = nullptr, *defl_work = nullptr, *defl_meta = nullptr, *defl_meta_h = nullptr;   // lazy
    long long rank_deflated_calls = 0, rank_deflated_instances = 0;
    int* rank_idx = nullptr;                         // [K] instances of the compact continuation batch (device), lazy
    int* rank_idx_h = nullptr;                       // ... pinned mirror
    long long rank_continued = 0, rank_cont_instances = 0;
    double rank_units = 0.0;                         // product launches in units of the WHOLE batch (a compact launch of m counts m / K)
    int rank_hold = 0;                               // iterations to stay at the fine resolution
    long long rank_calls = 0, rank_retries = 0, rank_fallbacks = 0, rank_launches = 0;
    long long ns_steps_total = 0, ns_calls = 0, ns_units_total = 0, ns_launches_total = 0, ns_eigh_fallbacks = 0;
    double ns_units_frac = 0.0, ns_steps_frac = 0.0;
    // per-phase HIP-event timing
    int prof_on = 0;          // 0 off, 1 every phase, 2 only the eigen/matrix-function phases (fewer event records)
    hipEvent_t ev[GGL_NPHASE][2] = {};
    bool ev_used[GGL_NPHASE] = {};
    // an EARLY first part of the Omega-step is recorded one iteration ahead of the collection that belongs to it, and the next
    // one goes into the stream before that collection: two event pairs, collected whenever their end has been reached
    hipEvent_t ev_early[2][2] = {};
    bool ev_early_used[2] = {false, false};
    int ev_early_par = 0;
    double ph_ms[GGL_NPHASE] = {};
    long long ph_cnt[GGL_NPHASE] = {};
};

#define PROF_HOT(ph) ((ph) == GGL_PH_EIG_OMEGA || (ph) == GGL_PH_EIG_OMEGA2 || (ph) == GGL_PH_EIG_L || \
                      (ph) == GGL_PH_ALLREDUCE_GROUPSQ || (ph) == GGL_PH_ALLREDUCE_NORMS)
#define PROF_ACTIVE(c, ph) ((c)->prof_on == 1 || ((c)->prof_on == 2 && PROF_HOT(ph)))
#define PB(c, ph) do { if (PROF_ACTIVE(c, ph)) (void)hipEventRecord((c)->ev[ph][0], (c)->stream); } while (0)
#define PE(c, ph) do { if (PROF_ACTIVE(c, ph)) { (void)hipEventRecord((c)->ev[ph][1], (c)->stream); (c)->ev_used[ph] = true; } } while (0)




// ---- shared host-side helpers: internal to the library (hidden visibility: only the extern "C" entry points of
// include/ggl_hip.h are the library's interface) ---------------------------------------------------------------------------
#pragma GCC visibility push(hidden)
static constexpr size_t STACK_SLACK = 64;      // bytes behind every stack that can be a product operand (odd p, see ctx_alloc)
static constexpr int GGL_SPIN_LIMIT_MS = 2000;
static constexpr int GGL_SPEC_RETRY = 1;     // internal: a speculative step failed validation, repeat it
static constexpr int GGL_NOT_LAUNCHED = 2;   // internal: omega_step(only_spec) found no speculative schedule and launched nothing
struct Xfer { void* dst; const void* src; size_t bytes; };
int poison_fill();
template <class T> inline hipError_t malloc_filled(T** p, size_t bytes, hipStream_t st)
{
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) return e;
    return hipMemsetAsync(*p, poison_fill(), bytes, st);
}
int fail(int code, const char* fmt, ...);
void prof_collect(ggl_ctx* c);   // call after a stream sync
int blas_handle(ggl_ctx* c, rocblas_handle* out);
bool use_jacobi(const ggl_ctx* c);
bool use_ns(int eig, int p);
int check_info(ggl_ctx* c, const char* what);
int download_stacks(ggl_ctx* c, const std::vector<Xfer>& xs);
int drop_prelaunch(ggl_ctx* c);
bool early_wanted(const ggl_ctx* c);
int eig_recon(ggl_ctx* c, double* A, double* out, double* Dv, int map, const double* betaK, int ph_eig = -1,
                     int ph_recon = -1);
int eigvals_only(ggl_ctx* c, double* A, double* Dv);
int ensure_partials(ggl_ctx* c, size_t need);
int finish_norms(ggl_ctx* c, int rows, double* out_norms, int group = 0);
int ggl_step_finish_impl(ggl_ctx* c, double rho, double lambda1, double lambda2, int reg, int latent,
                                const double* mu1, int groupsq_ready, double out_norms[5]);
int host_reduce(ggl_ctx* c, int rows, int nv, double* out /*nv*/, bool take_max);
void lds_missed(ggl_ctx* c);
int lds_table(ggl_ctx* c);
void mark_failed(ggl_ctx* c, int k, int why = 0, double value = 0.0);
int maybe_early(ggl_ctx* c);
int maybe_prelaunch(ggl_ctx* c, double rho, const double out_norms[5]);
int mgl_batch_step_impl(ggl_ctx* c, int G, const double* rho, const double* lambda1, const double* lambda2,
                               int reg, int latent, const double* mu1, const double* nk, double* out_norms);
int omega_step(ggl_ctx* c, int latent, CopySegs* pending = nullptr, bool allow_spec = false, bool only_spec = false);
int probe_part_streams(ggl_ctx* c);
int rank_step(ggl_ctx* c);
int rank_step_impl(ggl_ctx* c);
void sanitize_bounds(ggl_ctx* c, double* b, const double* repl, double repl_scale, double repl_scalar = 1.0);
int sgl_batch_step_impl(ggl_ctx* c, const double* rho, const double* lambda1, int latent, const double* mu1,
                               double* out_norms);
int snapshot_many(ggl_ctx* c, const int* kd, ggl_ctx* src, const int* ks, int n, bool with_state);
bool take_prelaunched(ggl_ctx* c, int latent);
void trace_host(ggl_ctx* c, int tag);
int trace_lane(const ggl_ctx* c, hipStream_t st);
void trace_mark(ggl_ctx* c, hipStream_t st, int tag);
void trace_symm_hook(hipStream_t st, int kind, void* arg);
int upload_par(ggl_ctx* c, int slot, const double* vals, double scalar, double div, CopySegs* pending = nullptr);
int validate_spec(ggl_ctx* c);
#pragma GCC visibility pop
#define DROP_PRE(c) do { int rc_ = drop_prelaunch(c); if (rc_) return rc_; } while (0)

// ---------------------------------------------------------------------------------------------
// stateless operators (host buffers in, host buffers out)
// ---------------------------------------------------------------------------------------------
namespace {
struct DevBuf {
    double* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(double) + STACK_SLACK); }
};
}  // namespace

#define UP(dst, src, n) HIPCHK(hipMemcpy(dst, src, (size_t)(n) * sizeof(double), hipMemcpyHostToDevice))
#define DOWN(dst, src, n) HIPCHK(hipMemcpy(dst, src, (size_t)(n) * sizeof(double), hipMemcpyDeviceToHost))

