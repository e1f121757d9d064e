// C ABI of libggl_hip.so (include/ggl_hip.h): context, state movement, the ADMM iteration and the
// stateless operator entry points.  Everything here is host code that sequences the gfx950 kernels
// of elementwise.hip / theta_pair.hip / eig_jacobi.hip / recon_gemm.hip on one HIP stream.
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

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(GGL_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
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

static void prof_collect(ggl_ctx* c)   // call after a stream sync
{
    if (!c->prof_on) return;
    for (int ph = 0; ph < GGL_NPHASE; ++ph) {
        if (!c->ev_used[ph]) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->ev[ph][0], c->ev[ph][1]) == hipSuccess) {
            c->ph_ms[ph] += ms;
            c->ph_cnt[ph] += 1;
        }
        c->ev_used[ph] = false;
    }
    for (int e = 0; e < 2; ++e) {
        float ms = 0.f;
        if (c->ev_early_used[e] && hipEventQuery(c->ev_early[e][1]) == hipSuccess) {
            if (hipEventElapsedTime(&ms, c->ev_early[e][0], c->ev_early[e][1]) == hipSuccess) {
                c->ph_ms[GGL_PH_EIG_OMEGA2] += ms;
                c->ph_cnt[GGL_PH_EIG_OMEGA2] += 1;
            }
            c->ev_early_used[e] = false;
        }
    }
}

// rocSOLVER needs a rocBLAS handle; creating one costs ~0.1-0.3 s (library initialisation), and the eigendecomposition
// route is only taken off the per-iteration path (exit checks, KKT, objective, fallbacks).  One handle per device for
// the whole process, created on first use and re-pointed at the calling ctx's stream (a ctx is used by one host thread
// at a time; the mutex only guards creation).
static int blas_handle(ggl_ctx* c, rocblas_handle* out)
{
    static std::mutex mu;
    static rocblas_handle handles[64] = {};
    if (c->device < 0 || c->device >= 64) return fail(GGL_E_ARG, "bad argument: device index");
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!handles[c->device] && rocblas_create_handle(&handles[c->device]) != rocblas_status_success) {
            handles[c->device] = nullptr;
            return fail(GGL_E_SOLVER, "rocblas_create_handle failed");
        }
    }
    if (rocblas_set_stream(handles[c->device], c->stream) != rocblas_status_success)
        return fail(GGL_E_SOLVER, "rocblas_set_stream failed");
    *out = handles[c->device];
    return GGL_OK;
}

static bool use_jacobi(const ggl_ctx* c)
{
    if (c->eig == GGL_EIG_JACOBI) return true;
    if (c->eig == GGL_EIG_ROCSOLVER) return false;
    return jacobi_fits(c->p);   // AUTO and NEWTON_SCHULZ: eigenvalue consumers use Jacobi when it fits
}

static bool use_ns(int eig, int p)
{
    // GGL_EIG_AUTO: the matrix-function (Newton-Schulz) Omega- / L-step from p = GGL_NS_MIN_P + 1 on.  Measured in round 4
    // (profiles/r4_jacobi_kernel_measured.txt, tools/bench_jacobi.py): the one-workgroup-per-matrix LDS Jacobi kernel takes
    // ~2.4 us per round-robin STEP whatever K (shuffle-reduction latency + a 1024-thread barrier; ~10 sweeps of p - 1 steps:
    // 2.5 ms at p = 100, 3.6 ms at p = 128), while the 7-8 FP64-MFMA products of the matrix-function route take 45-115 us from
    // p = 16 to p = 128 -- 3x faster at p = 16, 20x at p = 64, 35-45x at p = 128; the two meet at p = 8 (60 us), and Jacobi wins
    // below (22 us at p = 4).  Rounds 1-3 sent every p <= GGL_JACOBI_MAX_P to Jacobi, unmeasured.
    if (eig == GGL_EIG_NEWTON_SCHULZ) return true;
    return eig == GGL_EIG_AUTO && p > GGL_NS_MIN_P;
}

extern "C" int ggl_version(void) { return GGL_VERSION; }
extern "C" const char* ggl_last_error(void) { return g_err; }

extern "C" int ggl_theta_limits(int out[2])
{
    ARGCHK(out, "out");
    out[0] = GGL_FLAT_MAX_K;      // batched GGL grid: instances per problem the per-element Theta kernel takes
    out[1] = fgl_max_K();         // FGL: K-vectors that fit the LDS scan buffer of the Condat tile kernel
    return GGL_OK;
}

extern "C" int ggl_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(GGL_E_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

// ---------------------------------------------------------------------------------------------
// Reuse across ctxs: a solve of a small problem spent more time creating and destroying its ctx than iterating (K = 20,
// p = 50: create 0.53 ms, destroy 1.74 ms -- three frees that each wait for the device and unmap, four stream
// destructions -- against 2.0 ms for 30 iterations; tools/time_ctx.py).  Destroyed ctxs therefore leave their three arenas
// (up to POOL_MAX_BYTES of device memory, two sets) and their streams behind for the next ctx on the same device whose
// arenas have exactly the same sizes -- the usual case: a grid walked point by point, a compaction, repeated solves.  A
// reused arena is cleared completely (a fresh one is not guaranteed to be, but in practice is: the same state either way).
// Whatever is still pooled when the process ends is left to the driver.
// ---------------------------------------------------------------------------------------------
namespace {
constexpr size_t POOL_MAX_BYTES = (size_t)256 << 20;
constexpr int POOL_SETS = 2, POOL_STREAMS = 8;
struct ArenaSet { bool used = false; int device = 0; size_t tot[3] = {0, 0, 0}; void* ptr[3] = {nullptr, nullptr, nullptr}; unsigned long long age = 0; };
struct PoolStream { hipStream_t s = nullptr; int device = 0; };
std::mutex g_pool_mu;
ArenaSet g_arenas[POOL_SETS];
PoolStream g_streams[POOL_STREAMS];
int g_nstreams = 0;
unsigned long long g_pool_clock = 0;

bool pool_take_arenas(int device, const size_t tot[3], void* out[3])
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (ArenaSet& a : g_arenas)
        if (a.used && a.device == device && a.tot[0] == tot[0] && a.tot[1] == tot[1] && a.tot[2] == tot[2]) {
            for (int i = 0; i < 3; ++i) out[i] = a.ptr[i];
            a.used = false;
            return true;
        }
    return false;
}

// returns false when the set was not taken (the caller frees it)
bool pool_put_arenas(int device, const size_t tot[3], void* const ptr[3])
{
    if (tot[0] > POOL_MAX_BYTES) return false;
    ArenaSet victim;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        ArenaSet* slot = nullptr;
        for (ArenaSet& a : g_arenas)
            if (!a.used) { slot = &a; break; }
        if (!slot) {
            slot = &g_arenas[0];
            for (ArenaSet& a : g_arenas)
                if (a.age < slot->age) slot = &a;
            victim = *slot;
        }
        slot->used = true;
        slot->device = device;
        slot->age = ++g_pool_clock;
        for (int i = 0; i < 3; ++i) { slot->tot[i] = tot[i]; slot->ptr[i] = ptr[i]; }
    }
    if (victim.used) {
        (void)hipSetDevice(victim.device);
        (void)hipFree(victim.ptr[0]);
        (void)hipHostFree(victim.ptr[1]);
        (void)hipHostFree(victim.ptr[2]);
        (void)hipSetDevice(device);
    }
    return true;
}

hipError_t pool_stream_create(int device, hipStream_t* out)
{
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (int i = 0; i < g_nstreams; ++i)
            if (g_streams[i].device == device) {
                *out = g_streams[i].s;
                g_streams[i] = g_streams[--g_nstreams];
                return hipSuccess;
            }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

void pool_stream_release(int device, hipStream_t s, bool poolable)
{
    if (!s) return;
    if (poolable) {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (g_nstreams < POOL_STREAMS) {
            g_streams[g_nstreams].s = s;
            g_streams[g_nstreams].device = device;
            ++g_nstreams;
            return;
        }
    }
    (void)hipStreamDestroy(s);
}
}  // namespace

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
// All buffers a ctx owns from its creation come out of THREE allocations -- one device arena, one pinned arena, one
// fine-grained (coherent) pinned arena -- carved at 256-byte boundaries: a ctx used to take ~45 hipMalloc / hipHostMalloc calls
// and, worse, as many hipFree calls (each a device synchronisation: ggl_ctx_destroy cost 5 ms, half of a whole ADMM_MGL call at
// (20,200); tools/time_ctx.py).  Buffers that only some uses need (snapshots, ext state, deflation work, ...) stay lazy and own.
// Lazily allocated device buffers of a ctx start from zeros as its arenas do (0xFF bytes after ggl_debug_poison(1), see ctx_alloc)
static int g_poison = 0;
static int poison_fill() { return g_poison; }
// process-wide: odd p on the direct-to-LDS product kernel (default 1) or on the register-staged one as in rounds 1-5 (0) -- for
// A/B runs and the parity test of the two routes; returns the previous setting
extern "C" int ggl_set_odd_dl(int on)
{
    const int was = symm_dl_serves(3) ? 1 : 0;
    symm_set_odd_dl(on != 0);
    return was;
}

extern "C" int ggl_debug_poison(int on)
{
    // 0: zeros (default); 1: 0xFF bytes (NaN doubles, -1 ints); 2..255: that byte -- 0x7F gives 1.4e306 doubles and 0x47
    // gives 1.5e35: FINITE garbage, which a max / min reduction keeps where it drops a NaN
    g_poison = on == 1 ? 0xFF : (on & 0xFF);
    return GGL_OK;
}
template <class T> static hipError_t malloc_filled(T** p, size_t bytes, hipStream_t st)
{
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) return e;
    return hipMemsetAsync(*p, poison_fill(), bytes, st);
}

static constexpr size_t STACK_SLACK = 64;      // bytes behind every stack that can be a product operand (odd p, see ctx_alloc)

static int ctx_alloc(ggl_ctx* c)
{
    const size_t nb = c->n * sizeof(double);
    const size_t kp = (size_t)c->K * c->p;
    struct Req { void** pp; size_t bytes; int kind; };
    std::vector<Req> reqs;
#define DEV(ptr, bytes) reqs.push_back({(void**)&(ptr), (size_t)(bytes), 0})
#define PIN(ptr, bytes, kind) reqs.push_back({(void**)&(ptr), (size_t)(bytes), (kind)})
    // (+ STACK_SLACK: for odd p the product kernel's DMA reads the last element of a stack as the first half of a 16-byte
    // pair, gemm_sym.hip symm_dl_serves -- every buffer that can be a product operand has a few bytes behind it)
    DEV(c->S, nb + STACK_SLACK);
    DEV(c->Om[0], nb + STACK_SLACK);
    DEV(c->Om[1], nb + STACK_SLACK);
    DEV(c->Theta, nb + STACK_SLACK);
    DEV(c->L, nb + STACK_SLACK);
    DEV(c->X, nb + STACK_SLACK);
    DEV(c->W, nb + STACK_SLACK);
    DEV(c->DvO, kp * sizeof(double));
    DEV(c->DvL, kp * sizeof(double));
    DEV(c->scale, 2 * kp * sizeof(double));
    DEV(c->E, kp * sizeof(double));
    DEV(c->info, c->K * sizeof(int));
    DEV(c->sweeps, c->K * sizeof(int));
    DEV(c->par, 8 * (size_t)c->K * sizeof(double));
    PIN(c->par_h, 8 * (size_t)c->K * sizeof(double), 1);
    DEV(c->mask, (size_t)c->p * c->p * sizeof(double));
    // (p,p) + one trailing double: the speculation flag of K-sharded runs rides on the same all-reduce
    DEV(c->groupsq, ((size_t)c->p * c->p + 8) * sizeof(double));
    DEV(c->sqwork, (size_t)ggl_chunks(c->K, c->p) * c->p * c->p * sizeof(double));
    size_t pl = (size_t)c->K * elementwise_blocks(c->p) * GGL_NNORM;
    pl = std::max(pl, (size_t)pair_blocks(c->p, GGL_REG_GGL, c->K) * GGL_NNORM);
    pl = std::max(pl, (size_t)pair_blocks(c->p, GGL_REG_FGL, c->K) * GGL_NNORM);
    pl = std::max(pl, (size_t)theta_partial_blocks(c->p, GGL_REG_GGL, c->K, 1) * GGL_NNORM);
    pl = std::max(pl, (size_t)theta_partial_blocks(c->p, GGL_REG_GGL, c->K, 2) * GGL_NNORM);
    pl = std::max(pl, (size_t)theta_partial_blocks(c->p, GGL_REG_FGL, c->K, 2) * GGL_NNORM);
    c->partials_len = pl;
    DEV(c->partials, pl * sizeof(double));
    // (K,8) rows, and 2 * nprob * GGL_NNORM doubles for a batch of ext problems with ONE instance each (nprob = K)
    const size_t nl = (size_t)c->K * std::max(8, 2 * GGL_NNORM);
    DEV(c->norms, nl * sizeof(double));
    PIN(c->norms_h, nl * sizeof(double), 2);
    PIN(c->info_h, (size_t)c->K * sizeof(int), 1);
    PIN(c->gflag_h, sizeof(double), 2);
    PIN(c->sgl_fail_h, (size_t)c->K * sizeof(int), 2);
    DEV(c->arrive, 256);
    DEV(c->join_words, 256);
    if (c->omega_ns) {
        for (int i = 0; i < 2; ++i) { DEV(c->nsYP[i], 2 * nb + STACK_SLACK); }
        DEV(c->nsT, nb + STACK_SLACK);
        const size_t cl = (size_t)NS_MAX_LAUNCHES * NS_SLOT(c->K) * sizeof(double);   // last 3 slots: start / pre tables
        DEV(c->coef, cl);
        PIN(c->coef_hh[0], cl, 1);
        PIN(c->coef_hh[1], cl, 1);
        const size_t bl = 2 * (size_t)c->K * sizeof(double);
        PIN(c->bounds_h, bl, 2);
        const size_t nbl = 3 * (size_t)c->K * norm_bounds_blocks(c->p) * sizeof(double);   // + Collatz-Wielandt maxima
        DEV(c->nbrow, (size_t)c->K * c->p * sizeof(double));
        DEV(c->cwvec[0], (size_t)c->K * c->p * sizeof(double));
        DEV(c->cwvec[1], (size_t)c->K * c->p * sizeof(double));
        DEV(c->nbpart, nbl);
        const size_t t32 = (c->p + 31) / 32;
        DEV(c->rowpart, (size_t)c->K * t32 * c->p * sizeof(double));
        DEV(c->fropart, (size_t)c->K * (t32 * (t32 + 1) / 2) * sizeof(double));
        DEV(c->infpart, (size_t)c->K * bound_rows_blocks(c->p) * sizeof(double));
        DEV(c->cwmax, c->K * sizeof(unsigned long long));
        DEV(c->cwcnt, c->K * sizeof(unsigned));
        DEV(c->cuse, c->K * sizeof(double));
        PIN(c->cuse_hh[0], c->K * sizeof(double), 1);
        PIN(c->cuse_hh[1], c->K * sizeof(double), 1);
        // the words the host polls / reads right after the poll: explicitly coherent (fine-grained) pinned memory, so a
        // device store is visible to the host without a stream synchronisation whatever HIP_HOST_COHERENT says
        PIN(c->seq_h, sizeof(unsigned long long), 2);
        DEV(c->spec_flag, ggl_ctx::MAX_PARTS * sizeof(int));
        PIN(c->spec_flag_h, ggl_ctx::MAX_PARTS * sizeof(int), 2);
        c->spec_c = (double*)malloc(c->K * sizeof(double));
        c->spec_beta = (double*)malloc(c->K * sizeof(double));
        c->pre_beta = (double*)malloc(c->K * sizeof(double));
        c->early.beta = (double*)malloc(c->K * sizeof(double));
        c->wf_beta = (double*)malloc(c->K * sizeof(double));
        c->pre0_beta.assign(c->K, std::nan(""));
        DEV(c->maxdev, 2 * c->K * sizeof(double));          // [K] residuals | [K] traces of the sign iterate
        PIN(c->maxdev_h, 2 * c->K * sizeof(double), 1);
        HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        for (int i = 0; i < ggl_ctx::MAX_PARTS - 1; ++i) {
            HIPCHK(pool_stream_create(c->device, &c->streamx[i]));
            HIPCHK(hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming));
        }
        c->rank_ns = !c->rank_eig;
    }
#undef DEV
#undef PIN
    size_t tot[3] = {0, 0, 0};
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    for (const Req& r : reqs) tot[r.kind] += up(std::max<size_t>(r.bytes, 8));
    for (int i = 0; i < 3; ++i) c->arena_tot[i] = std::max<size_t>(tot[i], 256);
    void* reused[3];
    if (pool_take_arenas(c->device, c->arena_tot, reused)) {
        c->arena_dev = reused[0];
        c->arena_pin = reused[1];
        c->arena_pin_coh = reused[2];
    } else {
        HIPCHK(hipMalloc(&c->arena_dev, c->arena_tot[0]));
        HIPCHK(hipHostMalloc(&c->arena_pin, c->arena_tot[1]));
        HIPCHK(hipHostMalloc(&c->arena_pin_coh, c->arena_tot[2], hipHostMallocCoherent));
    }
    // Every arena starts from zeros, fresh or reused: hipMalloc hands back whatever an earlier allocation of the process left
    // there (a test of the full GPU suite failed once in eight runs and never alone -- behind the 20 GB ctxs of the C5 tests).
    // ggl_debug_poison(1) (process-wide; the tests call it when GGL_DEBUG_POISON=1 is in THEIR environment -- the library reads
    // none) fills them with 0xFF bytes instead -- NaN doubles, -1 ints -- so that a buffer which is read before it is written
    // shows up at once instead of once in a while.
    {
        const int fill = poison_fill();
        HIPCHK(hipMemsetAsync(c->arena_dev, fill, c->arena_tot[0], c->stream));
        memset(c->arena_pin, fill, c->arena_tot[1]);
        memset(c->arena_pin_coh, fill, c->arena_tot[2]);
    }
    size_t off[3] = {0, 0, 0};
    char* base[3] = {(char*)c->arena_dev, (char*)c->arena_pin, (char*)c->arena_pin_coh};
    for (const Req& r : reqs) {
        *r.pp = base[r.kind] + off[r.kind];
        off[r.kind] += up(std::max<size_t>(r.bytes, 8));
    }
    c->coef_h = c->coef_hh[0];
    c->cuse_h = c->cuse_hh[0];
    // initial contents
    HIPCHK(hipMemsetAsync(c->groupsq, 0, ((size_t)c->p * c->p + 8) * sizeof(double), c->stream));
    HIPCHK(hipMemsetAsync(c->arrive, 0, 256, c->stream));
    HIPCHK(hipMemsetAsync(c->join_words, 0, 256, c->stream));
    HIPCHK(hipMemsetAsync(c->L, 0, nb, c->stream));
    HIPCHK(hipMemsetAsync(c->X, 0, nb, c->stream));
    HIPCHK(hipMemsetAsync(c->Om[1], 0, nb, c->stream));
    if (c->omega_ns) {
        HIPCHK(hipMemsetAsync(c->cwmax, 0, c->K * sizeof(unsigned long long), c->stream));
        HIPCHK(hipMemsetAsync(c->cwcnt, 0, c->K * sizeof(unsigned), c->stream));
        *c->seq_h = 0;
        HIPCHK(hipMemsetAsync(c->spec_flag, 0, ggl_ctx::MAX_PARTS * sizeof(int), c->stream));
        memset(c->spec_flag_h, 0, ggl_ctx::MAX_PARTS * sizeof(int));
    }
    // whatever route the ctx takes: its first user may write these buffers from ANOTHER stream (ggl_ctx_create_subset copies
    // on the source's stream), and a memset still queued here would land on top of that
    HIPCHK(hipStreamSynchronize(c->stream));
    return GGL_OK;
}


static int drop_prelaunch(ggl_ctx* c);

static int set_option(ggl_ctx* c, int opt, double v)
{
    int rcd = drop_prelaunch(c);
    if (rcd) return rcd;
    switch (opt) {
        case GGL_OPT_SPECULATE: c->spec_enable = v != 0.0; break;
        case GGL_OPT_SPEC_FACTOR:
            if (!(v > 0.0)) return fail(GGL_E_ARG, "bad argument: GGL_OPT_SPEC_FACTOR must be positive");
            c->spec_factor = v;
            break;
        case GGL_OPT_NS_MODE:
            if (v != 0.0 && v != 1.0 && v != 2.0) return fail(GGL_E_ARG, "bad argument: GGL_OPT_NS_MODE is 0, 1 or 2");
            c->ns_force = (int)v;
            break;
        case GGL_OPT_NS_DEGREES: c->ns_degrees = v >= 9 ? 9 : (v >= 5 ? 5 : 3); break;
        case GGL_OPT_THETA_FLAT: c->theta_flat = (v == 2.0) ? 2 : (v != 0.0 ? 1 : 0); break;
        case GGL_OPT_RANK_EIG: c->rank_eig = v != 0.0; c->rank_ns = c->omega_ns && !c->rank_eig; break;
        case GGL_OPT_PARTS: c->ns_parts = std::min(std::max((int)v, 1), (int)ggl_ctx::MAX_PARTS); break;
        case GGL_OPT_PARTS_MAX_TILES: c->parts_max_tiles = (long)v; break;
        case GGL_OPT_SYMM_VARIANT:
            if (v >= 0 && !symm_variant_built((int)v))
                return fail(GGL_E_ARG, "bad argument: product-kernel variant not in this build");
            c->symm_variant = (int)v;
            break;
        case GGL_OPT_SPIN_WAIT: c->spin_wait = v != 0.0; break;
        case GGL_OPT_FUSED_BOUNDS: c->fused_bounds = v != 0.0; break;
        case GGL_OPT_PIPELINE: c->pipeline = v != 0.0; break;
        case GGL_OPT_FUSED_START: c->fused_start = v != 0.0; break;
        case GGL_OPT_PARTS_SMALL: c->parts_small = (int)v; break;
        case GGL_OPT_GROUP_SCHED:
            if (v != 0.0 && v != 1.0 && v != 2.0 && v != 3.0 && v != 12.0 && v != 13.0)
                return fail(GGL_E_ARG, "bad argument: GGL_OPT_GROUP_SCHED is 0, 1, 2, 3, 12 or 13");
            c->group_sched = (int)v;
            break;
#ifdef GGL_DEV
        case GGL_OPT_PARTS_BIAS: c->parts_bias = (int)v; break;
        case GGL_OPT_PARTS_ORDER: c->parts_order = (int)v; break;
        case GGL_OPT_CHAIN: c->chain_mode = (v == 2.0) ? 2 : (v != 0.0 ? 1 : 0); break;
        case GGL_OPT_FUSED_CW: c->fused_cw = v != 0.0; break;
        case GGL_OPT_RANK_CW: c->rank_cw = v != 0.0; break;
        case GGL_OPT_BOUND_SIDE: c->bound_side = (int)v; break;
#else
        case GGL_OPT_PARTS_BIAS: case GGL_OPT_PARTS_ORDER: case GGL_OPT_CHAIN: case GGL_OPT_FUSED_CW: case GGL_OPT_RANK_CW:
        case GGL_OPT_BOUND_SIDE: case GGL_OPT_PART_PRIORITY:
            if (v == 0.0) break;                 // (the default, which is what the product library runs)
            return fail(GGL_E_ARG, "bad argument: option %d is a measured-and-rejected alternative that only the development "
                        "library (libggl_hip_dev.so, python -m gglasso_amd.build --dev) carries", opt);
#endif
        case GGL_OPT_DOWNLOAD_THREADS: c->download_threads = std::min(std::max((int)v, 1), 64); break;
        case GGL_OPT_CW_WARM: c->cw_warm = v != 0.0; break;
        case GGL_OPT_ISOLATE: c->isolate = v != 0.0; break;
        case GGL_OPT_OMEGA_LDS: c->lds_omega = v != 0.0; c->lds_waves = (v == 4.0 || v == 8.0) ? (int)v : 0; break;
        case GGL_OPT_EARLY_PART: c->early_part = v != 0.0; break;
        case GGL_OPT_FUSED_W: c->fused_w = v != 0.0; break;
        case GGL_OPT_LDS_PINNED: c->lds_pinned = v != 0.0; break;
        case GGL_OPT_JOIN_FLAG: c->join_flag = v != 0.0; break;
        case GGL_OPT_CW_RIDER: c->cw_rider = (int)v; break;
        case GGL_OPT_COPY_RIDER: c->copy_rider = (int)v; break;
        case GGL_OPT_REDUCE_RIDER: c->red_rider = (int)v; break;
#ifdef GGL_DEV
        case GGL_OPT_PART_PRIORITY: {
            if (v != 0.0 && v != 1.0 && v != 2.0) return fail(GGL_E_ARG, "bad argument: GGL_OPT_PART_PRIORITY is 0, 1 or 2");
            if (!c->omega_ns || (int)v == c->part_priority) break;
            int lo = 0, hi = 0;                       // (numerically: hi <= 0 <= lo)
            HIPCHK(hipSetDevice(c->device));
            HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
            HIPCHK(hipStreamSynchronize(c->stream));
            for (int i = 0; i < ggl_ctx::MAX_PARTS - 1; ++i) {
                HIPCHK(hipStreamSynchronize(c->streamx[i]));
                HIPCHK(hipStreamDestroy(c->streamx[i]));
                c->streamx[i] = nullptr;
                if (v == 0.0) HIPCHK(hipStreamCreateWithFlags(&c->streamx[i], hipStreamNonBlocking));
                else HIPCHK(hipStreamCreateWithPriority(&c->streamx[i], hipStreamNonBlocking, v == 1.0 ? hi : lo));
            }
            c->part_priority = (int)v;
            c->parts_probed = false;
            break;
        }
#endif
        case GGL_OPT_RANK_DEFLATE: c->rank_deflate = v != 0.0; break;
        case GGL_OPT_RANK_L0_DEFLATE:
            if (!(v > 0.0) || v > 0.1) return fail(GGL_E_ARG, "bad argument: GGL_OPT_RANK_L0_DEFLATE is in (0, 0.1]");
            c->rank_l0_deflate = v;
            break;
        case GGL_OPT_RANK_L0_COARSE:
            if (!(v >= 0.0) || v > 0.1) return fail(GGL_E_ARG, "bad argument: GGL_OPT_RANK_L0_COARSE is in [0, 0.1]");
            c->rank_l0_coarse = v;
            break;
        case GGL_OPT_NS_TOL:
            if (!(v >= 0.0) || v > 1e-6) return fail(GGL_E_ARG, "bad argument: GGL_OPT_NS_TOL is in [0, 1e-6]");
            c->ns_tol = std::max(v, NS_TOL_EXACT);
            break;
        default: return fail(GGL_E_ARG, "bad argument: unknown ctx option %d", opt);
    }
    c->spec_have = false;      // a schedule built under other settings is not reused
    c->cw_have = false;
    c->cwL_have = false;
    return GGL_OK;
}

extern "C" int ggl_ctx_set_option(ggl_ctx* c, int opt, double value)
{
    ARGCHK(c, "ctx");
    return set_option(c, opt, value);
}

extern "C" int ggl_ctx_get_option(ggl_ctx* c, int opt, double* value)
{
    ARGCHK(c && value, "ctx, value");
    switch (opt) {
        case GGL_OPT_SPECULATE: *value = c->spec_enable; break;
        case GGL_OPT_SPEC_FACTOR: *value = c->spec_factor; break;
        case GGL_OPT_NS_MODE: *value = c->ns_force; break;
        case GGL_OPT_NS_DEGREES: *value = c->ns_degrees; break;
        case GGL_OPT_THETA_FLAT: *value = c->theta_flat; break;
        case GGL_OPT_RANK_EIG: *value = c->rank_eig; break;
        case GGL_OPT_PARTS: *value = c->ns_parts; break;
        case GGL_OPT_PARTS_MAX_TILES: *value = (double)c->parts_max_tiles; break;
        case GGL_OPT_SYMM_VARIANT: *value = c->symm_variant; break;
        case GGL_OPT_SPIN_WAIT: *value = c->spin_wait; break;
        case GGL_OPT_FUSED_BOUNDS: *value = c->fused_bounds; break;
        case GGL_OPT_PIPELINE: *value = c->pipeline; break;
        case GGL_OPT_FUSED_START: *value = c->fused_start; break;
        case GGL_OPT_PARTS_SMALL: *value = c->parts_small; break;
        case GGL_OPT_GROUP_SCHED: *value = c->group_sched; break;
        case GGL_OPT_PARTS_BIAS: *value = c->parts_bias; break;
        case GGL_OPT_PARTS_ORDER: *value = c->parts_order; break;
        case GGL_OPT_DOWNLOAD_THREADS: *value = c->download_threads; break;
        case GGL_OPT_NS_TOL: *value = c->ns_tol; break;
        case GGL_OPT_CW_WARM: *value = c->cw_warm; break;
        case GGL_OPT_CHAIN: *value = c->chain_mode; break;
        case GGL_OPT_RANK_L0_COARSE: *value = c->rank_l0_coarse; break;
        case GGL_OPT_ISOLATE: *value = c->isolate; break;
        case GGL_OPT_FUSED_CW: *value = c->fused_cw; break;
        case GGL_OPT_OMEGA_LDS: *value = c->lds_omega ? (c->lds_waves ? c->lds_waves : 1) : 0; break;
        case GGL_OPT_EARLY_PART: *value = c->early_part; break;
        case GGL_OPT_FUSED_W: *value = c->fused_w; break;
        case GGL_OPT_RANK_CW: *value = c->rank_cw; break;
        case GGL_OPT_BOUND_SIDE: *value = c->bound_side; break;
        case GGL_OPT_LDS_PINNED: *value = c->lds_pinned; break;
        case GGL_OPT_JOIN_FLAG: *value = c->join_flag; break;
        case GGL_OPT_CW_RIDER: *value = c->cw_rider; break;
        case GGL_OPT_COPY_RIDER: *value = c->copy_rider; break;
        case GGL_OPT_REDUCE_RIDER: *value = c->red_rider; break;
        case GGL_OPT_PART_PRIORITY: *value = c->part_priority; break;
        case GGL_OPT_RANK_DEFLATE: *value = c->rank_deflate; break;
        case GGL_OPT_RANK_L0_DEFLATE: *value = c->rank_l0_deflate; break;
        default: return fail(GGL_E_ARG, "bad argument: unknown ctx option %d", opt);
    }
    return GGL_OK;
}

#ifdef GGL_DEV
// development builds only (libggl_hip_dev.so): experiment knobs from the environment, applied on top of the defaults
static void dev_env_options(ggl_ctx* c)
{
    static const struct { const char* name; int opt; } tab[] = {
        {"GGL_SPECULATE", GGL_OPT_SPECULATE}, {"GGL_SPEC_FACTOR", GGL_OPT_SPEC_FACTOR}, {"GGL_NS_MODE", GGL_OPT_NS_MODE},
        {"GGL_NS_DEGREES", GGL_OPT_NS_DEGREES}, {"GGL_THETA_FLAT", GGL_OPT_THETA_FLAT}, {"GGL_RANK_EIG", GGL_OPT_RANK_EIG},
        {"GGL_TWO_STREAM", GGL_OPT_PARTS}, {"GGL_PARTS_MAX_TILES", GGL_OPT_PARTS_MAX_TILES},
        {"GGL_SYMM_VARIANT", GGL_OPT_SYMM_VARIANT}, {"GGL_SPIN_WAIT", GGL_OPT_SPIN_WAIT},
        {"GGL_FUSED_BOUNDS", GGL_OPT_FUSED_BOUNDS}, {"GGL_PIPELINE", GGL_OPT_PIPELINE},
        {"GGL_FUSED_START", GGL_OPT_FUSED_START}, {"GGL_PARTS_SMALL", GGL_OPT_PARTS_SMALL}, {"GGL_CHAIN", GGL_OPT_CHAIN}};
    for (const auto& t : tab)
        if (const char* v = getenv(t.name)) (void)set_option(c, t.opt, atof(v));
    if (const char* v = getenv("GGL_ROCSOLVER_SYEVJ")) c->use_syevj = atoi(v) != 0;
}
#endif

extern "C" int ggl_ctx_create(int device, int K, int p, int flags, void* stream, ggl_ctx** out)
{
    ARGCHK(out != nullptr, "out");
    ARGCHK(K >= 1 && p >= 1, "K, p must be positive");
    const int eig = flags & 0xff;
    ARGCHK(eig == GGL_EIG_AUTO || eig == GGL_EIG_JACOBI || eig == GGL_EIG_ROCSOLVER || eig == GGL_EIG_NEWTON_SCHULZ,
           "eigensolver selector");
    ARGCHK(eig != GGL_EIG_JACOBI || jacobi_fits(p), "GGL_EIG_JACOBI needs p <= GGL_JACOBI_MAX_P");
    const int nsm = (flags >> 8) & 0x3, nsd = (flags >> 12) & 0xf;
    ARGCHK(nsm <= 2, "GGL_EIG_NS_MODE is 0, 1 or 2");
    ARGCHK(nsd == 0 || nsd == 3 || nsd == 5 || nsd == 9, "GGL_EIG_NS_DEGREES is 3, 5 or 9");
    HIPCHK(hipSetDevice(device));
    ggl_ctx* c = new ggl_ctx();
    c->device = device;
    c->K = K;
    c->p = p;
    c->flags = flags;
    c->eig = eig;
    c->omega_ns = use_ns(eig, p);
    c->ns_force = nsm;
    if (nsd) c->ns_degrees = nsd;
    c->ns_parts = 2;
    c->n = (size_t)K * p * p;
    if (stream || (flags & GGL_CTX_STREAM_GIVEN)) {
        // GGL_CTX_STREAM_GIVEN: `stream` is the caller's stream even when the handle is NULL (the legacy default
        // stream, e.g. torch's default stream); without the bit a NULL handle means "create one"
        c->stream = (hipStream_t)stream;
    } else {
        hipError_t e = pool_stream_create(c->device, &c->stream);
        if (e != hipSuccess) { delete c; return fail(GGL_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
        c->own_stream = true;
    }
#ifdef GGL_DEV
    dev_env_options(c);
#endif
    int rc = ctx_alloc(c);
    if (rc != GGL_OK) { ggl_ctx_destroy(c); return rc; }
    *out = c;
    return GGL_OK;
}

extern "C" int ggl_ctx_destroy(ggl_ctx* c)
{
    if (!c) return GGL_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);      // also valid for the NULL (legacy default) stream
    // the part streams too, BEFORE anything is freed or handed to the pool: an early first part (maybe_early) with several
    // parts returns without joining them, so they may still be writing W / the Newton-Schulz scratch
    for (int i = 0; i < ggl_ctx::MAX_PARTS - 1; ++i)
        if (c->streamx[i]) (void)hipStreamSynchronize(c->streamx[i]);
    if (c->comm) {
        if (const RcclApi* api = rccl_api(nullptr)) (void)api->CommDestroy(c->comm);
        c->comm = nullptr;
    }
    // (the rocBLAS handle is the process-wide one of blas_handle(): never destroyed here)
    // lazily allocated buffers, each its own allocation
    double* lazy[] = {c->partials_own, c->nsNX, c->lds_tab, c->snapT, c->snapL, c->Lam[0], c->Lam[1], c->X1, c->Ckeep_alloc, c->snapC, c->snapOm, c->snapX, c->cwvecL[0], c->cwvecL[1], c->defl_G,
                      c->defl_work, c->defl_meta, c->maskK};
    for (double* b : lazy)
        if (b) (void)hipFree(b);
    if (c->defl_meta_h) (void)hipHostFree(c->defl_meta_h);
    free(c->Ckeep_beta);
    free(c->failed);
    free(c->fail_why);
    free(c->fail_value);
    free(c->snap_beta);
    free(c->snap_ns);
    for (int* b : {c->ext_pk, c->ext_Gt, c->ext_gsize, c->inst_pk, c->rank_idx})
        if (b) (void)hipFree(b);
    if (c->rank_idx_h) (void)hipHostFree(c->rank_idx_h);
    for (double* b : c->snap)
        if (b) (void)hipFree(b);
    if (c->chain_cnt) (void)hipFree(c->chain_cnt);
    free(c->spec_c);
    free(c->spec_beta);
    free(c->pre_beta);
    free(c->early.beta);
    free(c->wf_beta);
    // everything ctx_alloc handed out: three allocations
    {
        void* ptr[3] = {c->arena_dev, c->arena_pin, c->arena_pin_coh};
        if (!(c->arena_dev && c->arena_pin && c->arena_pin_coh && pool_put_arenas(c->device, c->arena_tot, ptr))) {
            if (c->arena_dev) (void)hipFree(c->arena_dev);
            if (c->arena_pin) (void)hipHostFree(c->arena_pin);
            if (c->arena_pin_coh) (void)hipHostFree(c->arena_pin_coh);
        }
    }
    for (int ph = 0; ph < GGL_NPHASE; ++ph)
        for (int e = 0; e < 2; ++e)
            if (c->ev[ph][e]) (void)hipEventDestroy(c->ev[ph][e]);
    for (int q = 0; q < 2; ++q)
        for (int e = 0; e < 2; ++e)
            if (c->ev_early[q][e]) (void)hipEventDestroy(c->ev_early[q][e]);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->trace.on) symm_set_launch_hook(nullptr, nullptr);
    for (hipEvent_t e : c->trace.ev) if (e) (void)hipEventDestroy(e);
    if (c->trace.base) (void)hipEventDestroy(c->trace.base);
    for (int i = 0; i < ggl_ctx::MAX_PARTS; ++i) {
        if (c->ev_bfork[i]) (void)hipEventDestroy(c->ev_bfork[i]);
        if (c->ev_bjoin[i]) (void)hipEventDestroy(c->ev_bjoin[i]);
    }
    for (int i = 0; i < ggl_ctx::MAX_PARTS - 1; ++i) {
        if (c->streamx[i]) { (void)hipStreamSynchronize(c->streamx[i]); pool_stream_release(c->device, c->streamx[i], c->part_priority == 0); }
        if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]);
    }
    if (c->own_stream && c->stream) pool_stream_release(c->device, c->stream, true);
    delete c;
    return GGL_OK;
}

extern "C" int ggl_ctx_sync(ggl_ctx* c)
{
    ARGCHK(c, "ctx");
    HIPCHK(hipStreamSynchronize(c->stream));
    return GGL_OK;
}

// A pre-launched Omega-step chain (see ggl_ctx::pipeline) uses W, the Newton-Schulz scratch and Omega[cur^1]; whoever
// touches the state or that scratch outside ggl_admm_step waits for it and forgets it.
static int drop_prelaunch(ggl_ctx* c)
{
    c->early.valid = false;          // (an early phase A wrote scratch only: nothing to undo, nothing to wait for)
    if (!c->pre_valid) return GGL_OK;
    c->pre_valid = false;
    c->pre_dropped += 1;
    // (spec_c still holds the bounds of the last VALIDATED chain: the replacement chain is built from them exactly as
    // the dropped one was, so dropping changes no iterate)
    HIPCHK(hipStreamSynchronize(c->stream));      // the chain's parts were joined into the main stream when it was launched
    // the dropped chain may have failed its validation: clear BOTH copies of every flag slot.  (omega_step re-zeroes only
    // the slots of the parts it launches; a part count changed after the drop would otherwise leave a stale 1 on the
    // device that every later speculative step's Theta / dual kernels read as "skip" -- ADVICE r2.)
    for (int h = 0; h < ggl_ctx::MAX_PARTS; ++h) c->spec_flag_h[h] = 0;
    if (c->spec_flag) HIPCHK(hipMemsetAsync(c->spec_flag, 0, ggl_ctx::MAX_PARTS * sizeof(int), c->stream));
    c->pre_cw_pending = false;      // its Collatz-Wielandt vector is never flipped in: the replacement rewrites it
    return GGL_OK;
}
#define DROP_PRE(c) do { int rc_ = drop_prelaunch(c); if (rc_) return rc_; } while (0)

extern "C" void* ggl_device_ptr(ggl_ctx* c, int which)
{
    if (!c) return nullptr;
    switch (which) {
        case GGL_BUF_S: return c->S;
        case GGL_BUF_OMEGA: return c->Om[c->cur];
        case GGL_BUF_OMEGA_PREV: return c->Om[c->cur ^ 1];
        case GGL_BUF_THETA: return c->Theta;
        case GGL_BUF_L: return c->L;
        case GGL_BUF_X: return c->X;
        case GGL_BUF_GROUPSQ: return c->groupsq;
        case GGL_BUF_NORMS: return c->norms;
        default: return nullptr;
    }
}

// ---------------------------------------------------------------------------------------------
// state
// ---------------------------------------------------------------------------------------------
// host array -> device stack; an array SHARED by the instances -- one (p,p) matrix for all of them (SGL grids: same S, Omega_0,
// X_0), or the (K',p,p) stack of one problem for each of the G grid points of a multiple-graph grid -- is uploaded once and
// replicated on the device by doubling copies instead of travelling K times over PCIe
static int upload_stack(ggl_ctx* c, double* dst, const double* src, int period)
{
    // period 0: the host array holds all K instances; P > 0: it holds P, and instance k is its instance k % P
    const size_t pp = (size_t)c->p * c->p;
    if (period <= 0 || period >= c->K) {
        HIPCHK(hipMemcpyAsync(dst, src, c->n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        return GGL_OK;
    }
    ARGCHK(c->K % period == 0, "the period of a shared array must divide K");
    HIPCHK(hipMemcpyAsync(dst, src, (size_t)period * pp * sizeof(double), hipMemcpyHostToDevice, c->stream));
    for (size_t have = (size_t)period; have < (size_t)c->K; have *= 2) {
        const size_t take = std::min(have, (size_t)c->K - have);
        HIPCHK(hipMemcpyAsync(dst + have * pp, dst, take * pp * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    }
    return GGL_OK;
}

static int host_reduce(ggl_ctx* c, int rows, int nv, double* out /*nv*/, bool take_max);
extern "C" int ggl_set_S_ex(ggl_ctx* c, const double* S, int period)
{
    ARGCHK(c && S, "ctx, S");
    c->spec_have = false;
    c->cw_have = false;
    c->cwL_have = false;
    HIPCHK(hipSetDevice(c->device));
    DROP_PRE(c);
    int rc = upload_stack(c, c->S, S, period);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    // exact symmetry of S decides whether a Theta kernel may form the next W per element (GGL_OPT_FUSED_W)
    launch_asym_max(c->stream, c->S, c->K, c->p, c->norms);
    HIPCHK(hipGetLastError());
    double asym = 1.0;
    rc = host_reduce(c, c->K, 1, &asym, true);
    if (rc) return rc;
    c->S_symmetric = (asym == 0.0);
    c->wf_ready = false;
    return GGL_OK;
}

extern "C" int ggl_set_S(ggl_ctx* c, const double* S) { return ggl_set_S_ex(c, S, 0); }

static int host_reduce(ggl_ctx* c, int rows, int nv, double* out /*nv*/, bool take_max);
extern "C" int ggl_set_state_ex(ggl_ctx* c, const double* Omega, const double* Theta, const double* L, const double* X,
                                const int* periods);

extern "C" int ggl_set_state(ggl_ctx* c, const double* Omega, const double* Theta, const double* L, const double* X)
{
    return ggl_set_state_ex(c, Omega, Theta, L, X, nullptr);
}

extern "C" int ggl_set_state_ex(ggl_ctx* c, const double* Omega, const double* Theta, const double* L, const double* X,
                                const int* periods)
{
    // periods (may be null = all 0): how many instances the host array of Omega / Theta / L / X holds (0: all K)
    ARGCHK(c, "ctx");
    const int pr[4] = {periods ? periods[0] : 0, periods ? periods[1] : 0, periods ? periods[2] : 0, periods ? periods[3] : 0};
    c->spec_have = false;      // bounds of another iterate say nothing about this one
    c->cw_have = false;
    c->cwL_have = false;        // (any positive vector would do, but every solve shall start the same way)
    HIPCHK(hipSetDevice(c->device));
    DROP_PRE(c);
    const size_t nb = c->n * sizeof(double);
    int rc = GGL_OK;
    if (Omega) rc = upload_stack(c, c->Om[c->cur], Omega, pr[0]);
    if (!rc && Theta) rc = upload_stack(c, c->Theta, Theta, pr[1]);
    if (!rc && L) rc = upload_stack(c, c->L, L, pr[2]);
    if (rc) return rc;
    if (!L) HIPCHK(hipMemsetAsync(c->L, 0, nb, c->stream));
    c->step_latent = (L != nullptr);          // a snapshot taken before any step keeps an uploaded L as well
    c->l_ns = false;                          // (an uploaded L is the caller's: ggl_finalize_L leaves it alone)
    if (X) { rc = upload_stack(c, c->X, X, pr[3]); if (rc) return rc; }
    HIPCHK(hipStreamSynchronize(c->stream));
    // exact symmetry of the dual and latent stacks decides whether the per-element Theta-step may be used
    c->state_symmetric = true;
    const double* chk[2] = {X ? c->X : nullptr, L ? c->L : nullptr};
    for (int i = 0; i < 2; ++i) {
        if (!chk[i]) continue;
        launch_asym_max(c->stream, chk[i], c->K, c->p, c->norms);
        HIPCHK(hipGetLastError());
        double asym = 0.0;
        int rc = host_reduce(c, c->K, 1, &asym, true);
        if (rc) return rc;
        if (!(asym == 0.0)) c->state_symmetric = false;
    }
    return GGL_OK;
}

extern "C" int ggl_state_snapshot(ggl_ctx* c, int restore)
{
    // restore == 0: keep a device copy of the iterate (Omega, Theta, L, X); != 0: make that copy the iterate again -- what
    // ggl_set_state does with the host arrays it was given, without the trip over PCIe (repeated solves from one start point:
    // benchmark regions, restarts).  Like ggl_set_state it forgets everything carried from earlier iterations.
    ARGCHK(c, "ctx");
    HIPCHK(hipSetDevice(c->device));
    DROP_PRE(c);
    const size_t nb = c->n * sizeof(double);
    double* cur[4] = {c->Om[c->cur], c->Theta, c->L, c->X};
    if (!restore) {
        for (int i = 0; i < 4; ++i) {
            if (!c->snap[i]) HIPCHK(malloc_filled(&c->snap[i], nb, c->stream));
            HIPCHK(hipMemcpyAsync(c->snap[i], cur[i], nb, hipMemcpyDeviceToDevice, c->stream));
        }
        c->snap_symmetric = c->state_symmetric;
        HIPCHK(hipStreamSynchronize(c->stream));
        return GGL_OK;
    }
    ARGCHK(c->snap[0], "no snapshot taken");
    for (int i = 0; i < 4; ++i) HIPCHK(hipMemcpyAsync(cur[i], c->snap[i], nb, hipMemcpyDeviceToDevice, c->stream));
    c->state_symmetric = c->snap_symmetric;
    c->spec_have = false;
    c->cw_have = false;
    c->cwL_have = false;
    c->l_ns = false;
    return GGL_OK;
}

// Whole stacks to the caller's (pageable) arrays.  MEASURED (tools/time_download.py, profiles/r5_download.txt): into arrays whose
// pages exist the copy runs at 55 GB/s (256 MB of a headline solve: 4.6 ms); into the FRESH arrays a solve returns it runs at
// 10 GB/s (26 ms) -- the time goes into the first touch of the destination's pages (a fault and a zeroed page per 4 KB, all in
// the one thread that copies out of the runtime's staging buffer), not into the transfer; more copy threads on more streams
// change nothing (tried: 2 threads +-10 %, 3-4 slower).  So the pages are touched first, by several host threads at once (one
// byte per page of memory that is about to be overwritten anyway), then ONE copy per stack: 25 -> 20 ms at the headline, 63 ->
// 46 ms for 640 MB, 40 -> 29 ms for C4's 400 MB (four threads do what sixteen do; what is left is the caller's allocator).
struct Xfer { void* dst; const void* src; size_t bytes; };
static int download_stacks(ggl_ctx* c, const std::vector<Xfer>& xs)
{
    size_t total = 0;
    for (const Xfer& x : xs) total += x.bytes;
    const int nthr = std::min(c->download_threads, (int)std::max(1u, std::thread::hardware_concurrency()));
    if (total >= ((size_t)32 << 20) && nthr > 1) {
        const size_t block = (size_t)2 << 20, page = 4096;
        // (MADV_HUGEPAGE on the destination first, on a box with transparent huge pages on request: no difference, measured)
        std::vector<Xfer> work;
        for (const Xfer& x : xs)
            for (size_t o = 0; o < x.bytes; o += block) work.push_back({(char*)x.dst + o, nullptr, std::min(block, x.bytes - o)});
        std::atomic<int> next{0};
        std::vector<std::thread> th;
        for (int t = 0; t < nthr; ++t)
            th.emplace_back([&]() {
                for (int i = next++; i < (int)work.size(); i = next++) {
                    volatile char* d = (volatile char*)work[i].dst;
                    for (size_t o = 0; o < work[i].bytes; o += page) d[o] = 0;
                    d[work[i].bytes - 1] = 0;
                }
            });
        for (std::thread& t : th) t.join();
    }
    for (const Xfer& x : xs) HIPCHK(hipMemcpyAsync(x.dst, x.src, x.bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return GGL_OK;
}

extern "C" int ggl_get_state(ggl_ctx* c, double* Omega, double* Theta, double* L, double* X)
{
    ARGCHK(c, "ctx");
    HIPCHK(hipSetDevice(c->device));
    DROP_PRE(c);
    const size_t nb = c->n * sizeof(double);
    std::vector<Xfer> xs;
    if (Omega) xs.push_back({Omega, c->Om[c->cur], nb});
    if (Theta) xs.push_back({Theta, c->Theta, nb});
    if (L) xs.push_back({L, c->L, nb});
    if (X) xs.push_back({X, c->X, nb});
    return download_stacks(c, xs);
}

extern "C" int ggl_set_lambda1_mask(ggl_ctx* c, const double* lam)
{
    ARGCHK(c, "ctx");
    HIPCHK(hipSetDevice(c->device));
    c->has_mask = (lam != nullptr);
    if (lam) {
        HIPCHK(hipMemcpyAsync(c->mask, lam, (size_t)c->p * c->p * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return GGL_OK;
}

extern "C" int ggl_set_lambda1_mask_k(ggl_ctx* c, const double* lam)
{
    ARGCHK(c, "ctx");
    HIPCHK(hipSetDevice(c->device));
    c->has_maskK = (lam != nullptr);
    if (lam) {
        if (!c->maskK) HIPCHK(malloc_filled(&c->maskK, c->n * sizeof(double), c->stream));
        HIPCHK(hipMemcpyAsync(c->maskK, lam, c->n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return GGL_OK;
}

extern "C" int ggl_set_instance_dims(ggl_ctx* c, const int* pk)
{
    ARGCHK(c, "ctx");
    HIPCHK(hipSetDevice(c->device));
    c->has_dims = (pk != nullptr);
    if (pk) {
        for (int k = 0; k < c->K; ++k) ARGCHK(pk[k] >= 1 && pk[k] <= c->p, "1 <= p_k <= p (the padded dimension of the ctx)");
        if (!c->inst_pk) HIPCHK(malloc_filled(&c->inst_pk, c->K * sizeof(int), c->stream));
        HIPCHK(hipMemcpyAsync(c->inst_pk, pk, c->K * sizeof(int), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return GGL_OK;
}

// ---------------------------------------------------------------------------------------------
// eigen-decomposition + eigenvalue map + reconstruction of a device stack (in: A, destroyed when
// the rocSOLVER path is taken; out may alias nothing).  Dv receives the eigenvalues.
// ---------------------------------------------------------------------------------------------
static int eig_recon(ggl_ctx* c, double* A, double* out, double* Dv, int map, const double* betaK, int ph_eig = -1,
                     int ph_recon = -1)
{
    c->info_dirty = true;
    if (use_jacobi(c)) {
        if (ph_eig >= 0) PB(c, ph_eig);
        HIPCHK(launch_jacobi(c->stream, A, Dv, nullptr, out, map, betaK, c->info, c->K, c->p));
        if (ph_eig >= 0) PE(c, ph_eig);
        return GGL_OK;
    }
    if (ph_eig >= 0) PB(c, ph_eig);
    {
        int rcb = blas_handle(c, &c->blas);
        if (rcb) return rcb;
    }
    if (c->use_syevj) {
        // experiment: rocSOLVER's Jacobi driver instead of syevd (GGL_ROCSOLVER_SYEVJ=1)
        rocblas_status sj = rocsolver_dsyevj_strided_batched(c->blas, rocblas_esort_none, rocblas_evect_original,
                                                             rocblas_fill_upper, c->p, A, c->p,
                                                             (rocblas_stride)c->p * c->p, 0.0, c->E, 100,
                                                             (rocblas_int*)c->sweeps, Dv, c->p, c->info, c->K);
        if (sj != rocblas_status_success) return fail(GGL_E_SOLVER, "rocsolver_dsyevj_strided_batched: status %d", (int)sj);
        if (ph_eig >= 0) PE(c, ph_eig);
        if (ph_recon >= 0) PB(c, ph_recon);
        launch_recon(c->stream, out, A, Dv, betaK, map, c->K, c->p, c->scale);
        if (ph_recon >= 0) PE(c, ph_recon);
        HIPCHK(hipGetLastError());
        return GGL_OK;
    }
    // row-major symmetric == column-major symmetric; the row-major LOWER triangle (what numpy's
    // eigh reads) is the column-major UPPER one.  Eigenvectors come back in column-major columns
    // == row-major ROWS, the layout launch_recon wants.
    rocblas_status st = rocsolver_dsyevd_strided_batched(c->blas, rocblas_evect_original, rocblas_fill_upper, c->p, A,
                                                         c->p, (rocblas_stride)c->p * c->p, Dv, c->p, c->E, c->p,
                                                         c->info, c->K);
    if (st != rocblas_status_success) return fail(GGL_E_SOLVER, "rocsolver_dsyevd_strided_batched: status %d", (int)st);
    if (ph_eig >= 0) PE(c, ph_eig);
    if (ph_recon >= 0) PB(c, ph_recon);
    launch_recon(c->stream, out, A, Dv, betaK, map, c->K, c->p, c->scale);
    if (ph_recon >= 0) PE(c, ph_recon);
    HIPCHK(hipGetLastError());
    return GGL_OK;
}

static int eigvals_only(ggl_ctx* c, double* A, double* Dv)
{
    c->info_dirty = true;
    if (use_jacobi(c)) {
        HIPCHK(launch_jacobi(c->stream, A, Dv, nullptr, nullptr, MAP_IDENT, nullptr, c->info, c->K, c->p));
        return GGL_OK;
    }
    {
        int rcb = blas_handle(c, &c->blas);
        if (rcb) return rcb;
    }
    rocblas_status st = rocsolver_dsyevd_strided_batched(c->blas, rocblas_evect_none, rocblas_fill_upper, c->p, A, c->p,
                                                         (rocblas_stride)c->p * c->p, Dv, c->p, c->E, c->p, c->info,
                                                         c->K);
    if (st != rocblas_status_success) return fail(GGL_E_SOLVER, "rocsolver_dsyevd (values): status %d", (int)st);
    return GGL_OK;
}

// why: 1 a spectral / norm bound that is not finite or not positive (value = the bound), 2 an eigensolver that did not converge
// (value = its info), 3 a non-finite residual or trace of the L-step's sign iteration (value = it), 4 marked in a subset ctx
// (fused batch iteration) -- kept for ggl_failed_reason: the FIRST mark of an instance stays
static void mark_failed(ggl_ctx* c, int k, int why = 0, double value = 0.0)
{
    if (!c->failed) {
        c->failed = (unsigned char*)calloc(c->K, 1);
        c->fail_why = (int*)calloc(c->K, sizeof(int));
        c->fail_value = (double*)calloc(c->K, sizeof(double));
    }
    if (!c->failed[k]) { c->fail_why[k] = why; c->fail_value[k] = value; }
    c->failed[k] = 1;
}

// GGL_OPT_ISOLATE: a non-finite (or non-positive) bound of instance k marks the instance and is replaced by `repl[k]` (or
// repl_scalar), so that the batch's schedule is planned for the healthy instances; without the option nothing is touched and
// the planner reports the non-finite input as it always did.
static void sanitize_bounds(ggl_ctx* c, double* b, const double* repl, double repl_scale, double repl_scalar = 1.0)
{
    if (!c->isolate) return;
    for (int k = 0; k < c->K; ++k)
        if (!std::isfinite(b[k]) || !(b[k] > 0.0)) {
            mark_failed(c, k, 1, b[k]);
            b[k] = repl ? repl_scale * repl[k] : repl_scalar;
        }
}

static int check_info(ggl_ctx* c, const char* what)
{
    const bool jac = use_jacobi(c);
    for (int k = 0; k < c->K; ++k) {
        const int v = c->info_h[k];
        if (jac ? (v < 0) : (v != 0)) {
            if (c->isolate) { mark_failed(c, k, 2, (double)v); continue; }
            return fail(GGL_E_SOLVER, "%s: eigensolver did not converge for instance %d (info=%d)", what, k, v);
        }
    }
    return GGL_OK;
}

// Small host<->device transfers of the iteration go through launch_copy_small (pinned host memory is
// device-visible): an ordinary kernel in the stream instead of a blit with its queue barriers.
static int upload_par(ggl_ctx* c, int slot, const double* vals, double scalar, double div, CopySegs* pending = nullptr)
{
    // par[slot][k] = (vals ? vals[k] : scalar) / div;  pending: append to a transfer the caller launches
    double* h = c->par_h + (size_t)slot * c->K;
    for (int k = 0; k < c->K; ++k) h[k] = (vals ? vals[k] : scalar) / div;
    if (pending) {
        pending->add(c->par + (size_t)slot * c->K, h, c->K * sizeof(double));
        return GGL_OK;
    }
    CopySegs sg;
    sg.add(c->par + (size_t)slot * c->K, h, c->K * sizeof(double));
    launch_copy_small(c->stream, sg);
    HIPCHK(hipGetLastError());
    return GGL_OK;
}

// ---------------------------------------------------------------------------------------------
// the iteration
// ---------------------------------------------------------------------------------------------
static constexpr int GGL_SPIN_LIMIT_MS = 2000;
static constexpr int GGL_SPEC_RETRY = 1;     // internal: a speculative step failed validation, repeat it
static constexpr int GGL_NOT_LAUNCHED = 2;   // internal: omega_step(only_spec) found no speculative schedule and launched nothing
static int omega_step(ggl_ctx* c, int latent, CopySegs* pending = nullptr, bool allow_spec = false, bool only_spec = false);
static int maybe_early(ggl_ctx* c);
static bool early_wanted(const ggl_ctx* c);

extern "C" int ggl_step_omega(ggl_ctx* c, double rho, int latent, const double* nk)
{
    ARGCHK(c, "ctx");
    ARGCHK(rho > 0, "rho must be positive");
    HIPCHK(hipSetDevice(c->device));
    DROP_PRE(c);
    // the previous step's pinned parameters are consumed: every step ends with a stream sync
    CopySegs sg;
    int rc = upload_par(c, 0, nk, 1.0, rho, &sg);   // beta_k = nk/rho    (admm_solver.py:180,184)
    if (rc) return rc;
    return omega_step(c, latent, &sg);
}

extern "C" int ggl_step_omega_spec(ggl_ctx* c, double rho, int latent, const double* nk)
{
    ARGCHK(c, "ctx");
    ARGCHK(rho > 0, "rho must be positive");
    HIPCHK(hipSetDevice(c->device));
    DROP_PRE(c);
    CopySegs sg;
    int rc = upload_par(c, 0, nk, 1.0, rho, &sg);
    if (rc) return rc;
    // with more than MAX_PARTS - 1 parts there is no slot left for the all-reduced flag
    return omega_step(c, latent, &sg, c->ns_parts < ggl_ctx::MAX_PARTS);
}

// HIP hands its streams a small pool of hardware queues, and two streams on the SAME queue run one after the other: the
// concurrent parts of an Omega-step then serialise without any error (seen with RCCL in the process: every kernel of both
// parts on one queue, K = 8 slabs 2300 instead of 3190 it/s).  Which queue a stream got cannot be asked, so it is measured,
// once per ctx before the first two-part step: an idle wave of 150 us on the main stream and on the part stream at the same
// time -- together they take ~150 us on different queues and ~300 us on one.  A part stream that serialises is replaced by
// the first of up to eight fresh streams that does not (stream priorities would force another queue, but starve one part:
// headline 1310 -> 940 / 864 it/s with a high / low priority part stream, GGL_OPT_PART_PRIORITY).
static int probe_part_streams(ggl_ctx* c)
{
    if (c->parts_probed) return GGL_OK;
    c->parts_probed = true;
    hipEvent_t e0, e1, e2;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventCreate(&e2));
    auto serial = [&](hipStream_t cand, bool* out) -> int {
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipStreamSynchronize(cand));
        HIPCHK(hipEventRecord(e0, c->stream));
        launch_spin_us(c->stream, 150);
        launch_spin_us(cand, 150);
        HIPCHK(hipEventRecord(e1, c->stream));
        HIPCHK(hipEventRecord(e2, cand));
        HIPCHK(hipEventSynchronize(e1));
        HIPCHK(hipEventSynchronize(e2));
        float a = 0.f, b = 0.f;
        HIPCHK(hipEventElapsedTime(&a, e0, e1));
        HIPCHK(hipEventElapsedTime(&b, e0, e2));
        *out = std::max(a, b) > 0.24f;
        return GGL_OK;
    };
    int rc = GGL_OK;
    bool ser = false;
    rc = serial(c->streamx[0], &ser);
    hipStream_t spare[8];
    int ns = 0;
    while (rc == GGL_OK && ser && ns < 8) {
        hipStream_t cand = nullptr;
        if (hipStreamCreateWithFlags(&cand, hipStreamNonBlocking) != hipSuccess) break;
        spare[ns++] = cand;
        rc = serial(cand, &ser);
        if (rc == GGL_OK && !ser) {
            std::swap(c->streamx[0], spare[ns - 1]);      // the old part stream joins the ones to destroy
            c->parts_replaced = ns;
        }
    }
    c->parts_serial = ser;
    for (int i = 0; i < ns; ++i) (void)hipStreamDestroy(spare[i]);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipEventDestroy(e2);
    return rc;
}

// the LDS-resident Omega-step's schedule table for the ctx's stopping tolerance / degree set (rebuilt when they change)
static int lds_table(ggl_ctx* c)
{
    if (c->lds_tab && c->lds_tab_tol == c->ns_tol && c->lds_tab_deg == c->ns_degrees) return GGL_OK;
    const size_t nt = (size_t)OMEGA_LDS_MAXTAB * OMEGA_LDS_ENT;
    if (!c->lds_tab) {
        HIPCHK(hipMalloc(&c->lds_tab, (nt + 2) * sizeof(double)));
        HIPCHK(hipMemsetAsync(c->lds_tab + nt, 0, 2 * sizeof(double), c->stream));
    }
    std::vector<double> tab(nt, 0.0);
    c->lds_ntab = omega_lds_build_table(c->ns_tol, c->ns_degrees, tab.data(), OMEGA_LDS_MAXTAB, &c->lds_lnq);
    if (c->lds_ntab < 1) return fail(GGL_E_SOLVER, "LDS Omega-step: empty schedule table");
    HIPCHK(hipStreamSynchronize(c->stream));           // (a launch still reading the old table)
    HIPCHK(hipMemcpy(c->lds_tab, tab.data(), nt * sizeof(double), hipMemcpyHostToDevice));
    c->lds_tab_tol = c->ns_tol;
    c->lds_tab_deg = c->ns_degrees;
    return GGL_OK;
}

static void lds_missed(ggl_ctx* c)
{
    c->lds_misses += 1;
    c->lds_cool = c->lds_cool_next;
    c->lds_cool_next = std::min(2 * c->lds_cool_next, 64);
    c->lds_last = false;
}

// ---- event timeline ------------------------------------------------------------------------------------------------------
// tags: 1 parameter copy, 2 form_W, 3 bound_rows, 4 cw_final, 10 product, 11 pair of products, 20 Theta, 21 norm reduction,
// 22 group sums (K-sharded), 23 all-reduce; host marks: 100 step entered, 101 Theta + reduction queued, 102 early part
// queued, 103 residuals seen, 104 rest of the next chain queued (step returns)
static int trace_lane(const ggl_ctx* c, hipStream_t st)
{
    if (st == c->stream) return 0;
    for (int i = 0; i < ggl_ctx::MAX_PARTS - 1; ++i)
        if (st == c->streamx[i]) return i + 1;
    return -1;
}
static void trace_mark(ggl_ctx* c, hipStream_t st, int tag)
{
    ggl_ctx::Trace& t = c->trace;
    if (!t.on || t.n >= t.cap) return;
    if (hipEventRecord(t.ev[t.n], st) != hipSuccess) return;
    t.tag[t.n] = tag;
    t.lane[t.n] = trace_lane(c, st);
    t.n += 1;
}
static void trace_host(ggl_ctx* c, int tag)
{
    ggl_ctx::Trace& t = c->trace;
    if (!t.on || t.nhost >= t.cap) return;
    t.host_us[t.nhost] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t.t0).count();
    t.host_tag[t.nhost] = tag;
    t.nhost += 1;
}
static void trace_symm_hook(hipStream_t st, int kind, void* arg) { trace_mark((ggl_ctx*)arg, st, 10 + kind); }

// GGL_OPT_GROUP_SCHED: contiguous groups of a batch whose instances need different product counts (ns_group_partition).
// cb[k] >= lambda_max(A'_k), beta_k = nk/rho.  Returns the number of groups (1: the batch stays whole, Kh / k0h untouched).
static int omega_groups(const ggl_ctx* c, const double* cb, const double* beta_h, int K, int* Kh, int* k0h, int* gunits)
{
    if (!c->group_sched || K < 2 || c->ns_force == 2 || c->chain_mode || c->comm) return 1;
    std::vector<int> u(K);
    for (int k = 0; k < K; ++k) {
        double ck = cb[k] * (1.0 + 1e-10);
        if (!(ck > 0.0) || !std::isfinite(ck) || !(beta_h[k] > 0.0)) return 1;
        if (ck < 4.0 * beta_h[k]) ck = 4.0 * beta_h[k];
        u[k] = ns_units_query(std::sqrt(4.0 * beta_h[k] / ck), c->ns_degrees, c->ns_tol);
        if (u[k] < 0) return 1;                      // (an instance for the stable schedule: the whole batch as one sequence)
    }
    int len[ggl_ctx::MAX_PARTS];
    // 12 / 13 (tests): up to 2 / 3 groups wherever the product counts differ -- the time model as if the launches were large
    const bool force = c->group_sched >= 10;
    const int gmax = force ? c->group_sched - 10 : (c->group_sched >= 2 ? c->group_sched : 3);
    const int G = ns_group_partition(u.data(), K, force ? 20000 : c->p, std::min(gmax, (int)ggl_ctx::MAX_PARTS - 1), len);
    if (G <= 1) return 1;
    for (int g = 0, k0 = 0; g < G; ++g) {
        Kh[g] = len[g];
        k0h[g] = k0;
        gunits[g] = 0;
        for (int k = k0; k < k0 + len[g]; ++k) gunits[g] = std::max(gunits[g], u[k]);
        k0 += len[g];
    }
    return G;
}

static void note_groups(ggl_ctx* c, int G, const int* Kh, const NsPlan* plans)
{
    const int G_prev = c->last_groups;
    c->last_groups = G;
    if (G <= 1) return;
    bool same = (G_prev == G);
    for (int g = 0; same && g < G; ++g) same = (c->last_group_len[g] == Kh[g]);
    if (c->group_steps > 0 && !same) c->group_changes += 1;
    c->group_steps += 1;
    for (int g = 0; g < ggl_ctx::MAX_PARTS; ++g) {
        c->last_group_len[g] = g < G ? Kh[g] : 0;
        c->last_group_units[g] = g < G ? plans[g].units : 0;
        if (g < G) c->group_units_sum[g] += plans[g].units;
    }
}

// Omega-step with beta_k in parameter slot 0 (already on the device, or part of the pending transfer)
static int omega_step(ggl_ctx* c, int latent, CopySegs* pending, bool allow_spec, bool only_spec)
{
    // only_spec: launch the chain only if it can run speculatively (no host synchronisation inside); else do nothing
    int rc;
    const double* beta = c->par;
    const int nxt = c->cur ^ 1;
    c->step_latent = latent;
    if (latent && !c->pre0_beta.empty()) std::fill(c->pre0_beta.begin(), c->pre0_beta.end(), std::nan(""));   // (the L-step's tables share the buffer)
    CopySegs first;
    if (pending) first = *pending;
    // a step whose kernels read their parameters from the pinned mirror never uploaded them: whoever reads the DEVICE copy
    // next (every other route below does, through `first`) gets it now
    else if (c->par0_stale) first.add(c->par, c->par_h, 8 * (size_t)c->K * sizeof(double));      // (all eight slots: a few KB)
    c->par0_stale = false;
    if (c->omega_ns) {
        const int K = c->K;
        // early phase A (ggl_ctx::EarlyA): `want_A` launches the first part of a speculative chain only; `resume` finds that
        // part in the stream, built for this beta, and adds the rest from the same plan
        const bool want_A = c->early_request;
        c->early_request = false;
        // W already in place?  (written by the Theta kernel that precedes this early first part in the stream, for this beta)
        bool w_ready = c->wf_ready && want_A && !latent;
        for (int k = 0; w_ready && k < K; ++k) w_ready = (c->par_h[k] == c->wf_beta[k]);
        c->wf_ready = false;
        bool resume = c->early.valid && allow_spec && !latent && !want_A && c->spec_enable;
        for (int k = 0; resume && k < K; ++k) resume = (c->par_h[k] == c->early.beta[k]);
        c->early.valid = false;
        if (!resume) {
            // a new plan goes into the OTHER copy of the pinned tables (a forgotten early part's copy kernel may not have run yet)
            c->plan_par ^= 1;
            c->coef_h = c->coef_hh[c->plan_par];
            c->cuse_h = c->cuse_hh[c->plan_par];
        }
        // phase A: A' = W^2 + 4 beta I, B' = A'^2 (both needed anyway), then the bound from B'
        double* pre = c->coef_h + (size_t)(NS_MAX_LAUNCHES - 2) * NS_SLOT(K);
        for (int k = 0; !resume && k < K; ++k) {
            double* o0 = pre + (size_t)k * NS_NCOEF;
            double* o1 = pre + NS_SLOT(K) + (size_t)k * NS_NCOEF;
            o0[0] = 4.0 * c->par_h[k]; o0[1] = 1.0; o0[2] = o0[3] = o0[4] = o0[5] = 0.0;
            o1[0] = 0.0; o1[1] = 1.0; o1[2] = o1[3] = o1[4] = o1[5] = 0.0;
        }
        double* pre_d = c->coef + (size_t)(NS_MAX_LAUNCHES - 2) * NS_SLOT(K);
        // Parts of the batch on concurrent streams: while one part's product drains its output and the next
        // launch ramps up, the other part keeps the matrix cores busy (a single launch sequence leaves them idle
        // for ~20 % of every product at p = 500).  Each part gets its own schedule.
        // (measured: +9 % at K=32,p=500; -2 % at p=1000 where several rounds of tiles already overlap;
        //  -12 % at K=20,p=200 where the launches are too small to split)
        // Every part runs its WHOLE chain (parameters, W, A', B', bound | schedule, products) on its own stream:
        // cross-stream event waits cost ~15 us of queue idle time each (rocprofv3 kernel trace), so there is one
        // fork at the very start, when the streams are idle anyway, and one join at the end.
        const long t64 = (c->p + 63) / 64;
        const long ntile = t64 * (t64 + 1) / 2 * K;
        int nh = (K >= 16 && ntile >= 600 && ntile <= c->parts_max_tiles) ? std::min(c->ns_parts, K / 8) : 1;
        // small batches of large matrices (the per-GPU slabs of a K-sharded run): one launch keeps the matrix cores ~40 %
        // busy whatever the tile shape, two concurrent launch sequences of K/2 instances each overlap their bubbles
        // (K = 8, p = 500: +2 %; K = 16 on the 32x32 kernel: -13 %, K = 4: -16 % -- so only the narrow band below 16)
        if (nh == 1 && c->parts_small && K >= c->parts_small && K < 16 && c->p >= 384 && c->ns_parts >= 2) nh = 2;
        nh = std::max(nh, 1);
        // Speculation: same beta as the last validated step => its bounds, inflated by 2 %, are very likely still
        // bounds (W moves little between ADMM iterations and the spectrum usually shrinks); the schedule is built
        // from them NOW and the products follow the bound kernels without the host round trip.
        bool spec = allow_spec && c->spec_enable && c->spec_have && !latent && c->spec_cool == 0;
        if (allow_spec && !only_spec && c->spec_cool > 0) c->spec_cool -= 1;      // one tick per iteration, not per attempt
        for (int k = 0; spec && k < K; ++k) spec = (c->par_h[k] == c->spec_beta[k]);
        if (resume) spec = true;
        int Kh[ggl_ctx::MAX_PARTS], k0h[ggl_ctx::MAX_PARTS];
        bool grouped = false;
        int gunits[ggl_ctx::MAX_PARTS] = {};
        if (resume) {
            nh = c->early.nh;
            for (int h = 0; h < nh; ++h) { Kh[h] = c->early.Kh[h]; k0h[h] = c->early.k0h[h]; }
        } else {
            for (int h = 0, k0 = 0; h < nh; ++h) {
                Kh[h] = K / nh + (h < K % nh ? 1 : 0);
                if (nh == 2 && c->parts_bias && std::abs(c->parts_bias) < K / 2) Kh[h] += h == 0 ? c->parts_bias : -c->parts_bias;
                k0h[h] = k0;
                k0 += Kh[h];
            }
            if (spec && nh == 1) {
                // instances that need different product counts: contiguous groups with their own schedules
                std::vector<double> cb(K);
                for (int k = 0; k < K; ++k) cb[k] = c->spec_c[k] * c->spec_factor;
                const int G = omega_groups(c, cb.data(), c->par_h, K, Kh, k0h, gunits);
                if (G > 1) { nh = G; grouped = true; }
            }
        }
        const size_t pp = (size_t)c->p * c->p;
        const int nbb = norm_bounds_blocks(c->p);
        // concurrent parts of a large batch: the 3-stage 64x64 DMA kernel; parts of a small batch: the size rule
        // (groups of different sizes: ONE kernel instance for all of them -- the bound partials of the parts are laid out
        // by the tile size, and the size rule could pick 32x32 tiles for a small group next to 64x64 for a large one)
        const int var_parts = grouped ? (c->symm_variant >= 0 ? c->symm_variant : (K >= 16 ? 17 : symm_auto_variant(K, c->p)))
                                      : ((c->symm_variant < 0 && nh > 1 && K >= 16) ? 17 : c->symm_variant);
        c->last_parts = nh;
        c->last_variant = symm_effective_variant(var_parts >= 0 ? var_parts : symm_auto_variant(Kh[0], c->p), c->p);
        const size_t region = (size_t)(NS_MAX_LAUNCHES - 4) / nh * NS_SLOT(K);      // coefficient slots per part
        NsPlan plans[ggl_ctx::MAX_PARTS];
        double* start_base_h = c->coef_h + (size_t)(NS_MAX_LAUNCHES - 3) * NS_SLOT(K);
        double* start_base_d = c->coef + (size_t)(NS_MAX_LAUNCHES - 3) * NS_SLOT(K);
        double* fused[ggl_ctx::MAX_PARTS] = {};      // speculative step: the first step's start as 2nd output of the B' launch
        bool cw_written = false;                     // this step's bound pass left a Collatz-Wielandt vector behind
        if (c->flags_dirty) {
            // a step was rejected since the flags were last cleared wholesale: whatever slot carried the 1 (a part that does
            // not exist in this step's split, the chain's completion check) must not outlive it.  On the main stream BEFORE
            // the fork, so it is ordered ahead of every part's own zeroing and kernels.
            HIPCHK(hipMemsetAsync(c->spec_flag, 0, ggl_ctx::MAX_PARTS * sizeof(int), c->stream));
            for (int h = 0; h < ggl_ctx::MAX_PARTS; ++h) c->spec_flag_h[h] = 0;
            c->flags_dirty = false;
        }
        // ---- small matrices: the whole step as ONE launch, one workgroup per instance, the chain resident in LDS ----------
        // The kernel finds bound and schedule itself, so it needs no host round trip: where the caller can repeat a step
        // (allow_spec) it runs like a speculative chain -- an instance outside its range (kappa > 300, non-finite data)
        // raises validation flag 0, the Theta-step leaves the iterate alone and the step is repeated on the launch chain --
        // elsewhere the flag is read back after a stream synchronisation.
        c->lds_last = false;
        const LdsSgl* sgl_req = c->sgl_req;         // (consumed here, whichever route the step takes)
        c->sgl_req = nullptr;
        c->sgl_done = false;
        if (c->lds_omega && c->p <= omega_lds_max_p() && c->ns_force == 0 && c->symm_variant < 0 && !c->chain_mode && !want_A && !resume) {
            const bool as_spec = allow_spec && c->spec_enable && !latent;
            if (c->lds_cool > 0) {
                if (!only_spec) c->lds_cool -= 1;
            } else if (as_spec || !only_spec) {
                rc = lds_table(c);
                if (rc) return rc;
                // K independent single problems: the same workgroup goes on with the Theta-step and the stopping-test sums
                LdsSgl sgl;
                const bool fused = sgl_req && as_spec && c->seq_h && c->spin_wait && !c->prof_on;
                // ... and takes its three parameters per instance (beta, lambda1 / rho, 1 / rho) straight from the pinned
                // mirror the caller has just filled: no parameter copy in front of it, the iteration is ONE launch (the
                // device flag stays zero in this form -- a miss clears it itself, sgl_fused_finish)
                // The plain form does the same when the pending transfer is nothing but beta (ggl_admm_step): the validation
                // flags it used to zero with that copy ARE zero unless a step was rejected (flags_dirty, handled above).
                const bool no_copy = c->lds_pinned && !c->info_dirty && ((fused && pending != nullptr) || (!sgl_req && (pending == nullptr || c->pending_beta_only || c->pending_pinned_ok)));
                CopySegs sg = first;
                sg.add(c->spec_flag, nullptr, sizeof(int));
                sg.add(c->spec_flag + ggl_ctx::MAX_PARTS - 1, nullptr, sizeof(int));
                c->spec_flag_h[0] = c->spec_flag_h[ggl_ctx::MAX_PARTS - 1] = 0;
                if (c->info_dirty) sg.add(c->info, nullptr, K * sizeof(int));
                if (!no_copy) launch_copy_small(c->stream, sg);
                PB(c, GGL_PH_EIG_OMEGA);
                unsigned long long* cnt = (unsigned long long*)(c->lds_tab + (size_t)OMEGA_LDS_MAXTAB * OMEGA_LDS_ENT);
                if (fused) {
                    sgl = *sgl_req;
                    sgl.Theta = c->Theta; sgl.X = c->X; sgl.OmegaPrev = c->Om[c->cur];
                    sgl.norms = c->norms_h; sgl.fail = c->sgl_fail_h;
                    sgl.seq = c->seq_h; sgl.seq_val = c->seq_wait = ++c->seq_next; sgl.arrive = c->arrive;
                    memset(c->sgl_fail_h, 0, K * sizeof(int));
                    if (no_copy) {
                        sgl.l1K = c->par_h + K;
                        sgl.invrhoK = c->par_h + 4 * (size_t)K;
                    }
                }
                if (no_copy && first.n > 0) { beta = c->par_h; c->par0_stale = true; }
                if (!launch_omega_lds(c->stream, c->Theta, latent ? c->L : nullptr, c->X, c->S, beta, c->Om[nxt], c->lds_tab,
                                      c->lds_ntab, c->lds_lnq, K, c->p, c->spec_flag, c->spec_flag_h, 0, cnt, c->bounds_h, nullptr,
                                      c->lds_waves, fused ? &sgl : nullptr))
                    return fail(GGL_E_HIP, "k_omega_lds: p = %d outside the kernel's range, or the LDS attribute was refused", c->p);
                PE(c, GGL_PH_EIG_OMEGA);
                c->sgl_done = fused;
                HIPCHK(hipGetLastError());
                c->last_parts = 1;
                c->last_variant = 41;
                c->lds_calls += 1;
                c->ns_calls += 1;
                c->ns_launches_total += 1;
                c->lds_last = true;
                if (c->info_dirty) { memset(c->info_h, 0, K * sizeof(int)); c->info_dirty = false; }
                if (as_spec) {
                    // validated by the caller after its stream sync (validate_spec) -- the fused SGL form is not speculative in
                    // that sense: an instance outside the range is redone ALONE by the caller (sgl_fused_finish)
                    c->spec_pending = !fused;
                    c->cw_pending = false;
                    c->dvo_valid = false;
                    c->cur = nxt;
                    return GGL_OK;
                }
                HIPCHK(hipStreamSynchronize(c->stream));
                if (c->spec_flag_h[0] == 0) {
                    sanitize_bounds(c, c->bounds_h, c->par_h, 4.0);
                    for (int k = 0; k < K; ++k) { c->spec_c[k] = c->bounds_h[k]; c->spec_beta[k] = c->par_h[k]; }
                    c->spec_have = true;
                    c->lds_cool_next = 4;
                    c->dvo_valid = false;
                    c->cur = nxt;
                    return GGL_OK;
                }
                // outside the kernel's range: this step (and the next few) on the launch chain
                c->par0_stale = false;                 // (whose parameter copy carries `first`)
                lds_missed(c);
                HIPCHK(hipMemsetAsync(c->spec_flag, 0, ggl_ctx::MAX_PARTS * sizeof(int), c->stream));
                for (int h = 0; h < ggl_ctx::MAX_PARTS; ++h) c->spec_flag_h[h] = 0;
            }
        }
#ifdef GGL_DEV
        // ---- the whole product chain as ONE persistent launch with per-instance dependencies (k_omega_chain) ----------
        if (spec && !want_A && !resume && c->chain_mode && c->fused_start && c->fused_bounds && (c->symm_variant < 0 || c->symm_variant == 17) &&
            chain_tile(K, c->p, c->chain_mode == 2) == 64) {
            if (!c->nsNX) HIPCHK(malloc_filled(&c->nsNX, 2 * c->n * sizeof(double) + STACK_SLACK, c->stream));
            if (!c->chain_cnt) HIPCHK(hipMalloc(&c->chain_cnt, (size_t)K * CHAIN_CNT_STRIDE * sizeof(unsigned)));
            for (int k = 0; k < K; ++k) c->cuse_h[k] = c->spec_c[k] * c->spec_factor;
            NsPlan& pl = plans[0];
            SymmOp ops[CHAIN_MAX_OPS];
            int nops = 0;
            const int bT = (c->p + 63) / 64;
            if (ns_plan(c->cuse_h, c->par_h, K, c->coef_h, start_base_h, &pl, c->ns_force, c->ns_degrees, c->ns_tol) == 0 &&
                !pl.stable) {
                double* f0 = nullptr;
                for (int k = 0; k < K; ++k)
                    f0 = ns_fused_start(pl, start_base_h + 5 * (size_t)k, c->nsYP[1], c->nsT, (size_t)K * pp,
                                        pre + NS_SLOT(K) + (size_t)k * NS_NCOEF + 3);
                if (f0)
                    nops = ns_chain_ops(pl, pre_d, pre_d + NS_SLOT(K), c->coef, c->W, c->nsYP[0], c->nsYP[1], c->nsNX, c->nsT,
                                        c->Om[nxt], K, c->p, 0, f0, c->rowpart, c->fropart, ops, CHAIN_MAX_OPS);
            }
            if (nops > 0) {
                CopySegs sg = first;
                sg.add(pre_d, pre, (size_t)K * NS_NCOEF * sizeof(double));
                sg.add(pre_d + NS_SLOT(K), pre + NS_SLOT(K), (size_t)K * NS_NCOEF * sizeof(double));
                // validation flags of this step: slot 0 the bound check, slot 1 the chain's completion check, last slot the
                // all-reduced flag of K-sharded runs
                sg.add(c->spec_flag, nullptr, 2 * sizeof(int));
                sg.add(c->spec_flag + ggl_ctx::MAX_PARTS - 1, nullptr, sizeof(int));
                c->spec_flag_h[0] = c->spec_flag_h[1] = c->spec_flag_h[ggl_ctx::MAX_PARTS - 1] = 0;
                const int nb_launch = pl.products - 2;
                if (nb_launch > 0) sg.add(c->coef, c->coef_h, (size_t)nb_launch * NS_SLOT(K) * sizeof(double));
                sg.add(c->cuse, c->cuse_h, (size_t)K * sizeof(double));
                if (c->info_dirty) sg.add(c->info, nullptr, K * sizeof(int));
                sg.add(c->chain_cnt, nullptr, (size_t)K * CHAIN_CNT_STRIDE * sizeof(unsigned));
                launch_copy_small(c->stream, sg);
                PB(c, GGL_PH_FORM_W);
                launch_form_W_sym(c->stream, c->W, c->Theta, nullptr, c->X, c->S, beta, K, c->p);
                PE(c, GGL_PH_FORM_W);
                PB(c, GGL_PH_EIG_OMEGA);
                ChainProg P;
                P.nops = nops; P.K = K; P.p = c->p; P.ntiles = bT * (bT + 1) / 2;
                P.begin[0] = 0;
                for (int i = 0; i < nops; ++i) { P.op[i] = ops[i]; P.begin[i + 1] = P.begin[i] + P.ntiles * (ops[i].pair ? 2 : 1); }
                if (launch_omega_chain(c->stream, P, c->chain_cnt, c->spec_flag + 1, c->spec_flag_h + 1) < 0)
                    return fail(GGL_E_HIP, "k_omega_chain: launch failed (%s)", hipGetErrorString(hipGetLastError()));
                // the bound of THIS iteration's A' (validation of the assumed one; next iteration's schedule): B' is intact
                launch_bound_rows(c->stream, c->rowpart, bT, K, c->p, c->nbrow, c->infpart);
                launch_cw_final(c->stream, c->nsYP[0] + c->n, c->nbrow, K, c->p, c->infpart, c->fropart, bT * (bT + 1) / 2,
                                c->cwmax, c->cwcnt, c->bounds_h, c->cuse, c->spec_flag, c->spec_flag_h, 0,
                                (c->cw_warm && c->cw_have) ? c->cwvec[c->cw_cur] : nullptr,
                                c->cw_warm ? c->cwvec[c->cw_cur ^ 1] : nullptr);
                PE(c, GGL_PH_EIG_OMEGA);
                HIPCHK(hipGetLastError());
                c->last_parts = 1;
                c->last_variant = 40;
                c->chain_calls += 1;
                c->ns_launches_total += pl.products;
                c->ns_units_frac += pl.units;
                c->ns_steps_frac += pl.steps;
                c->ns_units_total = (long long)(c->ns_units_frac + 0.5);
                c->ns_steps_total = (long long)(c->ns_steps_frac + 0.5);
                c->ns_calls += 1;
                c->spec_calls += 1;
                c->spec_pending = true;
                c->cw_pending = c->cw_warm;
                if (c->info_dirty) { memset(c->info_h, 0, K * sizeof(int)); c->info_dirty = false; }
                c->dvo_valid = false;
                c->cur = nxt;
                return GGL_OK;
            }
        }
#endif   // GGL_DEV (GGL_OPT_CHAIN)
        if (resume) {
            for (int h = 0; h < nh; ++h) { plans[h] = c->early.plans[h]; fused[h] = c->early.fused[h]; }
            c->early_used += 1;
        } else if (spec) {
            for (int k = 0; k < K; ++k) c->cuse_h[k] = c->spec_c[k] * c->spec_factor;
            sanitize_bounds(c, c->cuse_h, c->par_h, 4.0);
            for (int h = 0; spec && h < nh; ++h) {
                const int k0 = k0h[h];
                const int prc = ns_plan(c->cuse_h + k0, c->par_h + k0, Kh[h], c->coef_h + h * region,
                                        start_base_h + 5 * k0, &plans[h], c->ns_force, c->ns_degrees, c->ns_tol);
                spec = (prc == 0) && !plans[h].stable;
                for (int k = k0; spec && c->fused_start && k < k0 + Kh[h]; ++k) {
                    // the bound is assumed known, so the start is a fixed combination of A' and B': {dI, dC, dE} of B' launch
                    fused[h] = ns_fused_start(plans[h], start_base_h + 5 * (size_t)k, c->nsYP[1] + k0 * pp, c->nsT + k0 * pp,
                                              nh > 1 ? c->n : (size_t)K * pp, pre + NS_SLOT(K) + (size_t)k * NS_NCOEF + 3);
                }
            }
        }
        if (only_spec && !spec) return GGL_NOT_LAUNCHED;
        if (nh > 1 && !c->parts_probed) {
            rc = probe_part_streams(c);
            if (rc) return rc;
        }
        if (nh > 1 && !resume) {
            HIPCHK(hipEventRecord(c->ev_fork, c->stream));
            for (int h = 1; h < nh; ++h) HIPCHK(hipStreamWaitEvent(c->streamx[h - 1], c->ev_fork, 0));
        }
        // (The parts' launches are issued part after part.  Issuing them round-robin, so that the parts start together
        // instead of ~100 us apart, was measured 3 % SLOWER at (32,500): the stagger is what keeps the parts' prologues
        // and epilogues from coinciding.)
        for (int hh = 0; hh < nh; ++hh) {
            // GGL_OPT_PARTS_ORDER: the part on the main stream is queued LAST, so that it is the one that ends last and the
            // Theta kernel behind it finds the other part's flag set already
            const int h = (c->parts_order && nh == 2 && c->prof_on == 0) ? nh - 1 - hh : hh;
            hipStream_t sh = h == 0 ? c->stream : c->streamx[h - 1];
            const int k0 = k0h[h];
            // The host's mirrors of the validation flags are cleared when the REST of the chain is launched: the mirrors of an
            // early part's iteration are still to be read when the part goes into the stream (the device words, cleared by
            // the part's copy kernel, have been read by then -- the Theta-step that takes them is ahead in the stream).
            if (!want_A) {
                c->spec_flag_h[h] = 0;
                if (h == 0 && nh < ggl_ctx::MAX_PARTS) c->spec_flag_h[ggl_ctx::MAX_PARTS - 1] = 0;
            }
            double* Ap = c->nsYP[0] + k0 * pp;
            double* Bp = c->nsYP[0] + c->n + k0 * pp;
            const int btile = c->fused_bounds ? symm_bounds_tile(Kh[h], c->p, var_parts) : 0;
            const int bT = btile ? (c->p + btile - 1) / btile : 0;
            double* rowp = btile ? c->rowpart + (size_t)k0 * bT * c->p : nullptr;
            double* frop = btile ? c->fropart + (size_t)k0 * (bT * (bT + 1) / 2) : nullptr;
            if (!resume) {
            // ---- first part: parameter tables, W, A', B' (scratch only) ----
            // the pending parameter transfers are repeated on every part's stream (identical values, a few KB)
            // GGL_OPT_COPY_RIDER: nothing pending and the device's coefficient rows of A' = W^2 + 4 beta I already those of this
            // beta (they only change with rho): no launch of its own reads the rest before B', so the tables ride in the A'
            // launch (symm_set_copy_rider) -- one dependent launch less between the norm reduction and A'
            // MEASURED (profiles/r5_copy_rider_ab.txt, three interleaved pairs per workload in one box): single launch sequences
            // K = 4 slab +4 %, (20,200) +4 %, K = 16 +1.4 %, (64,100) +4 %, (32,128) +5 %; TWO concurrent parts lose -- headline
            // -2.3 %, K = 8 slab -5 %, three of three pairs each (both A' launches end ~7 us earlier in the event timeline and
            // the iteration is no shorter: the parts are bound by what they share, not by their first launch) -- so: 1 = only
            // where the chain is one sequence.
            bool ride_copy = (c->copy_rider == 2 || (c->copy_rider == 1 && nh == 1)) && first.n == 0 && !latent && btile != 0 &&
                             !c->chain_mode && c->prof_on != 1;
            for (int k = k0; ride_copy && k < k0 + Kh[h]; ++k) ride_copy = (c->pre0_beta[k] == c->par_h[k]);
            CopySegs sg = first;
            if (!ride_copy) {
                sg.add(pre_d + NS_NCOEF * (size_t)k0, pre + NS_NCOEF * (size_t)k0, (size_t)Kh[h] * NS_NCOEF * sizeof(double));
                for (int k = k0; k < k0 + Kh[h]; ++k) c->pre0_beta[k] = c->par_h[k];
            }
            sg.add(pre_d + NS_SLOT(K) + NS_NCOEF * (size_t)k0, pre + NS_SLOT(K) + NS_NCOEF * (size_t)k0,
                   (size_t)Kh[h] * NS_NCOEF * sizeof(double));
            // validation flags of this step: this part's slot, and (part 0) the slot of the all-reduced flag of K-sharded
            // runs, where a rank must skip and repeat the step when ANY rank's speculation failed -- also a rank that
            // did not speculate itself
            sg.add(c->spec_flag + h, nullptr, sizeof(int));
            if (h == 0 && nh < ggl_ctx::MAX_PARTS) sg.add(c->spec_flag + ggl_ctx::MAX_PARTS - 1, nullptr, sizeof(int));
            if (spec) {
                sg.add(start_base_d + 5 * (size_t)k0, start_base_h + 5 * (size_t)k0, (size_t)Kh[h] * 5 * sizeof(double));
                const int nb_launch = plans[h].products - 2;
                if (nb_launch > 0)
                    sg.add(c->coef + h * region, c->coef_h + h * region, (size_t)nb_launch * NS_SLOT(Kh[h]) * sizeof(double));
                sg.add(c->cuse + k0, c->cuse_h + k0, (size_t)Kh[h] * sizeof(double));
                if (h == 0 && c->info_dirty) sg.add(c->info, nullptr, K * sizeof(int));
            }
            if (h == 0 && c->red_pending.nblk > 0) {
                // (with anything else in front of A' the reduction goes first, as its own launch)
                if (w_ready && (c->red_rider == 2 || nh == 1)) {
                    symm_set_reduce_rider(c->red_pending);
                    c->red_rides += 1;
                } else {
                    launch_reduce_partials(sh, c->red_pending.partials, 1, c->red_pending.nblk, c->red_pending.nv, c->red_pending.out,
                                           c->red_pending.seq, c->red_pending.seq_val);
                    trace_mark(c, sh, 21);
                }
                c->red_pending = RedRider{};
            }
            if (ride_copy) {
                symm_set_copy_rider(sg);
                c->copy_rides += 1;
            } else {
                launch_copy_small(sh, sg);
                trace_mark(c, sh, 1);
            }
            if (h == 0) PB(c, GGL_PH_FORM_W);
            if (!w_ready) {
                launch_form_W_sym(sh, c->W + k0 * pp, c->Theta + k0 * pp, latent ? c->L + k0 * pp : nullptr, c->X + k0 * pp,
                                  c->S + k0 * pp, beta + k0, Kh[h], c->p);
                trace_mark(c, sh, 2);
            } else if (h == 0) c->wf_used += 1;
            const bool early_ev = want_A && h == 0 && c->prof_on == 2;
            if (early_ev) {
                c->ev_early_par ^= 1;
                (void)hipEventRecord(c->ev_early[c->ev_early_par][0], c->stream);
            } else if (h == 0 && !want_A) { PE(c, GGL_PH_FORM_W); PB(c, GGL_PH_EIG_OMEGA); }
            // lambda_max(A')^2 = lambda_max(B') <= min(|B'|_inf, |B'|_F, Collatz-Wielandt ratio), reduced on the
            // device; only the K_part bounds travel to the (pinned, device-visible) host array.  Where the B' launch is
            // the direct-to-LDS kernel, its epilogue leaves the row sums and Frobenius shares of B' behind (no norm pass
            // over B'), and the Collatz-Wielandt pass finishes the bound itself.
            ns_prepare(sh, pre_d + NS_NCOEF * (size_t)k0, pre_d + NS_SLOT(K) + NS_NCOEF * (size_t)k0, c->W + k0 * pp, Ap, Bp, Kh[h], c->p,
                       var_parts, spec ? fused[h] : nullptr, rowp, frop);
            symm_flush_rider(sh);
            if (early_ev) {
                (void)hipEventRecord(c->ev_early[c->ev_early_par][1], c->stream);
                c->ev_early_used[c->ev_early_par] = true;
            }
            }
            if (want_A) continue;
            if (resume && h == 0) PB(c, GGL_PH_EIG_OMEGA);
            // ---- the rest: bound of this iteration's A' (validation of the assumed one), products, Omega ----
            // speculative chain of a small launch sequence: the two bound kernels only VALIDATE (the schedule was built from
            // the previous iteration's bound), so they need not sit in the chain's dependent sequence -- side stream, beside
            // the first products, joined before B' is overwritten (ns_run) -- where the chip has room (one or two parts of few
            // tiles; at the headline both parts are bound by throughput and round 3 measured this slower)
            hipStream_t sb = sh;
            hipEvent_t bfree = nullptr;
            const int side_slot = nh + h - 1;                      // part streams 0 .. nh-2 are taken by the parts
            // measured (profiles/r5_bound_side.txt): three interleaved pairs per workload in one box -- headline (two parts of
            // 16) +3.0 / +1.5 / +2.2 %; K = 16 and K = 4 within noise; K = 8 (two parts of 4) -1.5 %, C3 -5.5 %, (64,100) -5 %,
            // (32,128) -4.6 %: a cross-stream wait costs more than the small launches hide -- and six more headline pairs in a
            // second box: -2.4 / +0.8 / -1.6 % with 50-step regions, -0.1 / +0.8 / +3.1 % with the driver's 20-step regions.
            // Nine pairs, +0.8 % on average with a run-to-run scatter of +-2 %: not a result.  Off by default.
            const bool side_on = c->bound_side == 1 || (c->bound_side == 2 && nh > 1 && K >= 16);
            if (side_on && spec && btile && !c->fused_cw && side_slot < ggl_ctx::MAX_PARTS - 1 && c->streamx[side_slot]) {
                if (!c->ev_bfork[h]) {
                    HIPCHK(hipEventCreateWithFlags(&c->ev_bfork[h], hipEventDisableTiming));
                    HIPCHK(hipEventCreateWithFlags(&c->ev_bjoin[h], hipEventDisableTiming));
                }
                sb = c->streamx[side_slot];
                HIPCHK(hipEventRecord(c->ev_bfork[h], sh));
                HIPCHK(hipStreamWaitEvent(sb, c->ev_bfork[h], 0));
                bfree = c->ev_bjoin[h];
            }
            // GGL_OPT_CW_RIDER: the validation rides in the first product launch of ns_run (CwRider, kernels.hpp) -- needs the
            // Collatz-Wielandt vector of the previous iteration
            const bool ride = c->cw_rider && spec && btile && !c->fused_cw && sb == sh && c->cw_warm && c->cw_have;
            if (ride) {
                CwRider r;
                r.B = Bp; r.rowpart = rowp; r.fropart = frop;
                r.dprev = c->cwvec[c->cw_cur] + (size_t)k0 * c->p;
                r.dnext = c->cwvec[c->cw_cur ^ 1] + (size_t)k0 * c->p;
                r.d_out = c->nbrow + (size_t)k0 * c->p;
                r.cwmax = c->cwmax + k0; r.cnt = c->cwcnt + k0; r.out = c->bounds_h + k0; r.cuse = c->cuse + k0;
                r.flag = c->spec_flag; r.flag_host = c->spec_flag_h; r.flag_slot = h;
                r.T = bT; r.ntile = bT * (bT + 1) / 2; r.p = c->p; r.K = Kh[h]; r.nbx = (c->p + 15) / 16;
                symm_set_rider(r);
                if (c->cw_rider == 2) symm_flush_rider(sh);
                c->cw_rides += 1;
                cw_written = true;
            } else if (btile) {
                if (c->fused_cw) {
                    launch_bound_cw(sh, Bp, rowp, bT, Kh[h], c->p, c->nbrow + (size_t)k0 * c->p, frop, bT * (bT + 1) / 2,
                                    c->cwmax + k0, c->cwcnt + k0, c->bounds_h + k0, spec ? c->cuse + k0 : nullptr,
                                    spec ? c->spec_flag : nullptr, spec ? c->spec_flag_h : nullptr, h,
                                    (c->cw_warm && c->cw_have) ? c->cwvec[c->cw_cur] + (size_t)k0 * c->p : nullptr,
                                    c->cw_warm ? c->cwvec[c->cw_cur ^ 1] + (size_t)k0 * c->p : nullptr);
                } else {
                const int nib = bound_rows_blocks(c->p);
                launch_bound_rows(sb, rowp, bT, Kh[h], c->p, c->nbrow + (size_t)k0 * c->p, c->infpart + (size_t)k0 * nib);
                trace_mark(c, sb, 3);
                launch_cw_final(sb, Bp, c->nbrow + (size_t)k0 * c->p, Kh[h], c->p, c->infpart + (size_t)k0 * nib, frop,
                                bT * (bT + 1) / 2, c->cwmax + k0, c->cwcnt + k0, c->bounds_h + k0, spec ? c->cuse + k0 : nullptr,
                                spec ? c->spec_flag : nullptr, spec ? c->spec_flag_h : nullptr, h,
                                (c->cw_warm && c->cw_have) ? c->cwvec[c->cw_cur] + (size_t)k0 * c->p : nullptr,
                                c->cw_warm ? c->cwvec[c->cw_cur ^ 1] + (size_t)k0 * c->p : nullptr);
                trace_mark(c, sb, 4);
                }
                cw_written = c->cw_warm;
            } else {
                double* nb2 = c->nbpart + 2 * (size_t)k0 * nbb;
                double* nbc = c->nbpart + 2 * (size_t)K * nbb + (size_t)k0 * nbb;
                launch_norm_bounds(sh, Bp, Kh[h], c->p, nb2, c->nbrow + (size_t)k0 * c->p);
                launch_cw_bounds(sh, Bp, c->nbrow + (size_t)k0 * c->p, Kh[h], c->p, nbc);
                launch_bound_final(sh, nb2, nbc, nbb, Kh[h], c->bounds_h + k0, 0, spec ? c->cuse + k0 : nullptr,
                                   spec ? c->spec_flag + h : nullptr, spec ? c->spec_flag_h + h : nullptr);
            }
            if (bfree) HIPCHK(hipEventRecord(bfree, sb));
            if (spec) {
                ns_run(sh, plans[h], c->coef + h * region, start_base_d + 5 * k0, c->W + k0 * pp, c->nsYP[0] + k0 * pp,
                       c->nsYP[1] + k0 * pp, c->nsT + k0 * pp, c->Om[nxt] + k0 * pp, Kh[h], c->p,
                       var_parts, nh > 1 ? c->n : 0, fused[h] != nullptr, bfree);
                symm_flush_rider(sh);                     // (a chain without a direct-to-LDS product launch: its own launch)
                c->ns_launches_total += plans[h].products;
                const double frac = (double)Kh[h] / K;
                c->ns_units_frac += frac * plans[h].units;
                c->ns_steps_frac += frac * plans[h].steps;
            }
            if (h == 0 && !spec) PE(c, GGL_PH_EIG_OMEGA);
        }
        HIPCHK(hipGetLastError());
        if (want_A) {
            for (int h = 0; h < nh; ++h) { c->early.plans[h] = plans[h]; c->early.fused[h] = fused[h]; c->early.Kh[h] = Kh[h]; c->early.k0h[h] = k0h[h]; }
            c->early.nh = nh;
            memcpy(c->early.beta, c->par_h, K * sizeof(double));
            c->early.valid = true;
            c->early_launched += 1;
            return GGL_OK;
        }
        if (spec) {
            // (parts that share a hardware queue keep the event join: a polling wave in front of the kernel it waits for would
            // sit out its time limit -- the host queues the set before the wait, so this is belt and braces)
            if (nh > 1 && c->join_flag && !c->parts_serial) {
                // (see k_wait_flags: the waiting queue idles ~25 us behind a cross-queue event that has fired)
                c->join_seq += 1;
                for (int h = 1; h < nh; ++h) launch_set_flag(c->streamx[h - 1], c->join_words + h, c->join_seq);
                launch_wait_flags(c->stream, c->join_words + 1, nh - 1, c->join_seq, c->spec_flag, c->spec_flag_h, 0, 200.0);
                HIPCHK(hipGetLastError());
            } else {
                for (int h = 1; h < nh; ++h) {
                    HIPCHK(hipEventRecord(c->ev_join[h - 1], c->streamx[h - 1]));
                    HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join[h - 1], 0));
                }
            }
            PE(c, GGL_PH_EIG_OMEGA);
            c->ns_units_total = (long long)(c->ns_units_frac + 0.5);
            c->ns_steps_total = (long long)(c->ns_steps_frac + 0.5);
            c->ns_calls += 1;
            c->spec_calls += 1;
            note_groups(c, grouped ? nh : 1, Kh, plans);
            c->spec_pending = true;        // validated by the caller after its stream sync (finish_norms)
            c->cw_pending = cw_written;
            if (c->info_dirty) { memset(c->info_h, 0, K * sizeof(int)); c->info_dirty = false; }
            c->dvo_valid = false;
            c->cur = nxt;
            return GGL_OK;
        }
        for (int h = 0; h < nh; ++h) HIPCHK(hipStreamSynchronize(h == 0 ? c->stream : c->streamx[h - 1]));
        sanitize_bounds(c, c->bounds_h, c->par_h, 4.0);          // (GGL_OPT_ISOLATE: lambda_min(A') = 4 beta stands in)
        // validated bounds: the next step may speculate on them
        for (int k = 0; k < K; ++k) { c->spec_c[k] = c->bounds_h[k]; c->spec_beta[k] = c->par_h[k]; }
        c->spec_have = true;
        if (cw_written) { c->cw_cur ^= 1; c->cw_have = true; }
        bool any_stable = false;
        size_t region_b = region;
        int var_b = var_parts;
        if (nh == 1) {
            // phase A ran as one launch sequence; the products may still run as groups with their own schedules
            const int G = omega_groups(c, c->bounds_h, c->par_h, K, Kh, k0h, gunits);
            if (G > 1) {
                nh = G;
                grouped = true;
                region_b = (size_t)(NS_MAX_LAUNCHES - 4) / nh * NS_SLOT(K);
                var_b = c->symm_variant >= 0 ? c->symm_variant : (K >= 16 ? 17 : symm_auto_variant(K, c->p));
                if (!c->parts_probed) {
                    rc = probe_part_streams(c);
                    if (rc) return rc;
                }
            }
        }
        for (int h = 0; h < nh; ++h) {
            const int k0 = k0h[h];
            const int prc = ns_plan(c->bounds_h + k0, c->par_h + k0, Kh[h], c->coef_h + h * region_b, start_base_h + 5 * k0,
                                    &plans[h], c->ns_force, c->ns_degrees, c->ns_tol);
            if (prc == -1) return fail(GGL_E_SOLVER, "Newton-Schulz Omega-step: non-finite W (diverged iterate?)");
            if (prc == -2) {
                // pathological scaling (|W|^2 rho / nk > 1e12): eigendecomposition of the (still intact) W
                c->ns_eigh_fallbacks += 1;
                rc = eig_recon(c, c->W, c->Om[nxt], c->DvO, MAP_PHIPLUS, beta, -1, GGL_PH_RECON_OMEGA);
                if (rc) return rc;
                c->dvo_valid = true;
                c->cur = nxt;
                return GGL_OK;
            }
            any_stable = any_stable || plans[h].stable;
        }
        if (nh > 1 && any_stable) {
            // the stable schedule multiplies a contiguous [Y|P] pair: run the whole batch as one sequence
            const int prc = ns_plan(c->bounds_h, c->par_h, K, c->coef_h, start_base_h, &plans[0], c->ns_force, c->ns_degrees, c->ns_tol);
            if (prc != 0) return fail(GGL_E_SOLVER, "Newton-Schulz Omega-step: plan failed (%d)", prc);
        }
        const int nrun = (nh > 1 && !any_stable) ? nh : 1;
        if (nrun == 1) { Kh[0] = K; k0h[0] = 0; }
        c->last_parts = nrun;
        c->last_variant = symm_effective_variant((nrun > 1 && var_b >= 0) ? var_b
                          : (c->symm_variant >= 0 ? c->symm_variant : symm_auto_variant(Kh[0], c->p)), c->p);
        note_groups(c, (grouped && nrun > 1) ? nrun : 1, Kh, plans);
        PB(c, GGL_PH_EIG_OMEGA2);
        for (int h = 0; h < nrun; ++h) {
            const int Kr = Kh[h], k0 = k0h[h];
            hipStream_t sh = h == 0 ? c->stream : c->streamx[h - 1];
            CopySegs up;
            up.add(start_base_d + 5 * (size_t)k0, start_base_h + 5 * (size_t)k0, (size_t)Kr * 5 * sizeof(double));
            const int nb_launch = plans[h].products - 2;     // launches of phase B
            if (nb_launch > 0)
                up.add(c->coef + h * region_b, c->coef_h + h * region_b, (size_t)nb_launch * NS_SLOT(Kr) * sizeof(double));
            if (h == 0 && c->info_dirty) up.add(c->info, nullptr, K * sizeof(int));   // no eigensolver ran: info = 0
            launch_copy_small(sh, up);
            ns_run(sh, plans[h], c->coef + h * region_b, start_base_d + 5 * k0,
                   c->W + k0 * pp, c->nsYP[0] + k0 * pp, c->nsYP[1] + k0 * pp, c->nsT + k0 * pp, c->Om[nxt] + k0 * pp, Kr,
                   c->p,
                   // tile choice by the work of the WHOLE batch: the other parts share the chip (measured +6.7 %);
                   // with parts, the 3-stage DMA pipeline is 2.8 % ahead of the double buffer (4 % behind without)
                   nrun > 1 ? var_b : c->symm_variant, nrun > 1 ? c->n : 0);
            c->ns_stable_calls += plans[h].stable ? 1 : 0;
            c->ns_launches_total += plans[h].products;
            // algorithmic work in units of (whole-stack) K p^3 flop
            const double frac = (double)Kr / K;
            c->ns_units_frac += frac * plans[h].units;
            c->ns_steps_frac += frac * plans[h].steps;
        }
        for (int h = 1; h < nrun; ++h) {
            HIPCHK(hipEventRecord(c->ev_join[h - 1], c->streamx[h - 1]));
            HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join[h - 1], 0));
        }
        PE(c, GGL_PH_EIG_OMEGA2);
        HIPCHK(hipGetLastError());
        c->ns_units_total = (long long)(c->ns_units_frac + 0.5);
        c->ns_steps_total = (long long)(c->ns_steps_frac + 0.5);
        c->ns_calls += 1;
        c->dvo_valid = false;
        if (c->info_dirty) { memset(c->info_h, 0, K * sizeof(int)); c->info_dirty = false; }
        c->cur = nxt;
        return GGL_OK;
    }
    if (only_spec) return GGL_NOT_LAUNCHED;
    launch_copy_small(c->stream, first);
    PB(c, GGL_PH_FORM_W);
    launch_form_W(c->stream, c->W, c->Theta, latent ? c->L : nullptr, c->X, c->S, beta, c->K, c->p);
    PE(c, GGL_PH_FORM_W);
    HIPCHK(hipGetLastError());
    rc = eig_recon(c, c->W, c->Om[nxt], c->DvO, MAP_PHIPLUS, beta, GGL_PH_EIG_OMEGA, GGL_PH_RECON_OMEGA);
    if (rc) return rc;
    c->dvo_valid = true;
    c->cur = nxt;
    return GGL_OK;
}

extern "C" int ggl_step_group_partial(ggl_ctx* c, double rho, double lambda1)
{
    ARGCHK(c, "ctx");
    ARGCHK(rho > 0, "rho must be positive");
    HIPCHK(hipSetDevice(c->device));
    // u = soft(Omega + L + X, l1/rho) (admm_solver.py:190-191): L only takes part in the latent model (it is zero otherwise)
    launch_group_sums_packed(c->stream, c->groupsq, c->sqwork, c->Om[c->cur], c->step_latent ? c->L : nullptr, c->X,
                             (1.0 / rho) * lambda1, c->K, c->p, c->spec_pending ? c->spec_flag : nullptr);
    HIPCHK(hipGetLastError());
    return GGL_OK;
}

// After the stream sync that ends an iteration: a speculative Omega-step (here, or on another rank of a K-sharded run) is
// accepted or -- GGL_SPEC_RETRY -- undone, the caller then repeats the step without speculation.
static int validate_spec(ggl_ctx* c)
{
    if (c->spec_pending || c->sharded_check) {
        // speculative Omega-step (here, or on another rank of a K-sharded run): were the assumed bounds still bounds?
        const bool mine = c->spec_pending, sharded = c->sharded_check;
        c->spec_pending = false;
        c->sharded_check = false;
        bool bad = false;
        for (int h = 0; h < ggl_ctx::MAX_PARTS; ++h) bad = bad || (c->spec_flag_h[h] != 0);
        if (sharded) bad = bad || (*c->gflag_h > 0.5);        // some rank (possibly this one) missed: all repeat
        if (bad) {
            // no: the Theta-step kernels saw the flag and left the iterate alone; un-flip Omega and tell the caller
            if (mine && c->lds_last) lds_missed(c);
            else if (mine) c->spec_misses += 1;
            c->flags_dirty = true;
            c->cw_pending = false;
            c->spec_have = false;
            c->spec_cool = 4;
            c->cur ^= 1;
            return GGL_SPEC_RETRY;
        }
        if (mine) {
            if (c->lds_last) { c->lds_cool_next = 4; c->spec_have = true; }
            sanitize_bounds(c, c->bounds_h, c->par_h, 4.0);
            for (int k = 0; k < c->K; ++k) { c->spec_c[k] = c->bounds_h[k]; c->spec_beta[k] = c->par_h[k]; }
            if (c->cw_pending) { c->cw_cur ^= 1; c->cw_have = true; }
        }
        c->cw_pending = false;
    }
    return GGL_OK;
}

static int finish_norms(ggl_ctx* c, int rows, double* out_norms, int group = 0)
{
    // out_norms: 5 sums over all rows; group > 0: (rows/group, 5) -- one row of sums per `group` consecutive rows
    // the reduction usually wrote the sums straight into pinned host memory and no eigensolver touched `info`:
    // then there is nothing to copy, only the stream to wait for
    CopySegs dn;
    if (!c->norms_host) dn.add(c->norms_h, c->norms, (size_t)rows * GGL_NNORM * sizeof(double));
    if (c->info_dirty) dn.add(c->info_h, c->info, c->K * sizeof(int));
    if (c->sharded_check) dn.add(c->gflag_h, c->groupsq + ggl::tri_len(c->p), sizeof(double));   // the all-reduced speculation flag
    // a few words to fetch (the all-reduced sums and flag of a K-sharded step) and a host that may poll: the copy publishes
    // the sequence number itself, behind its copies
    size_t dn_words = 0;
    for (int i = 0; i < dn.n; ++i) dn_words += dn.words[i];
    const bool dn_seq = dn.n > 0 && dn_words <= 4096 && c->seq_h && c->spin_wait && !c->prof_on;
    if (dn_seq) {
        c->seq_wait = ++c->seq_next;
        launch_copy_small_seq(c->stream, dn, c->seq_h, c->seq_wait);
    } else {
        launch_copy_small(c->stream, dn);
    }
    HIPCHK(hipGetLastError());
    bool waited = false;
    const unsigned long long want = c->seq_wait;
    if (want != 0 && (dn.n == 0 || dn_seq)) {
        // everything this step produced for the host is in (coherent) pinned memory and the reduction publishes a
        // sequence number after it: poll that word (the stream is in order, so all earlier work is complete as well).
        // Bounded: after GGL_SPIN_LIMIT_MS the wait falls back to a stream synchronisation, and a sequence number that
        // is still missing after THAT is an error, not a silent pass.
        const volatile unsigned long long* sq = c->seq_h;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spin = 1;; ++spin) {
            if (*sq == want) { waited = true; break; }
            __builtin_ia32_pause();
            if ((spin & 0xfff) == 0 &&
                std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(GGL_SPIN_LIMIT_MS)) {
                c->spin_timeouts += 1;
                break;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (waited && dn.n == 0 && c->stamp_want == want) {
            // the row itself carries the sequence number behind its sums (k_reduce_partials, red_rider_body): seen only with them
            const volatile double* stamp = c->norms_h + GGL_NNORM;
            for (unsigned spin = 1; *stamp != (double)want; ++spin) {
                __builtin_ia32_pause();
                if ((spin & 0xfff) == 0 &&
                    std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(GGL_SPIN_LIMIT_MS)) {
                    c->spin_timeouts += 1;
                    waited = false;
                    break;
                }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        }
    }
    c->stamp_want = 0;
    c->seq_wait = 0;
    if (!waited || c->prof_on) HIPCHK(hipStreamSynchronize(c->stream));
    if (want != 0 && (dn.n == 0 || dn_seq) && !waited && *(const volatile unsigned long long*)c->seq_h != want)
        return fail(GGL_E_HIP, "end of iteration: the norm reduction did not publish sequence number %llu (found %llu) "
                    "although the stream is idle", want, *(const volatile unsigned long long*)c->seq_h);
    c->norms_host = false;
    prof_collect(c);
    {
        const int vrc = validate_spec(c);
        if (vrc) return vrc;
    }
    int rc = check_info(c, "ADMM step");
    if (rc) return rc;
    const int gsz = group > 0 ? group : rows;
    for (int r0 = 0, o = 0; r0 < rows; r0 += gsz, ++o) {
        for (int v = 0; v < GGL_NNORM; ++v) {
            double s = 0.0;
            for (int r = r0; r < r0 + gsz; ++r) s += c->norms_h[(size_t)r * GGL_NNORM + v];
            out_norms[(size_t)o * GGL_NNORM + v] = s;
        }
    }
    return GGL_OK;
}

// L = (C - mu I)_+ with C in c->W and mu_k/rho in parameter slot 2 (pinned mirror par_h + 2K).
// Sign Newton-Schulz with a-posteriori verification; retries at a finer resolution, then falls back to the
// eigendecomposition, so the result always meets the eigh route's accuracy.
static int rank_step_impl(ggl_ctx* c);

static int rank_step(ggl_ctx* c)
{
    const long long fallbacks = c->rank_fallbacks;
    int rc = rank_step_impl(c);
    if (rc) return rc;
    c->l_ns = c->rank_ns && c->rank_fallbacks == fallbacks;
    if (c->l_ns) {
        // keep C for ggl_finalize_L: W is scratch that every step forms anew, so the two stacks swap names (the stream was
        // synchronised by the step's checks; a latent step neither speculates nor pre-launches, nothing in flight holds W)
        if (!c->Ckeep) {
            HIPCHK(malloc_filled(&c->Ckeep_alloc, c->n * sizeof(double) + STACK_SLACK, c->stream));
            c->Ckeep = c->Ckeep_alloc;
            c->Ckeep_beta = (double*)malloc(c->K * sizeof(double));
        }
        std::swap(c->W, c->Ckeep);
        memcpy(c->Ckeep_beta, c->par_h + 2 * (size_t)c->K, c->K * sizeof(double));
    }
    return GGL_OK;
}

static int rank_step_impl(ggl_ctx* c)
{
    const int K = c->K;
    const double* mu_h = c->par_h + 2 * (size_t)K;
    if (!c->rank_ns) return eig_recon(c, c->W, c->L, c->DvL, MAP_RANK, c->par + 2 * (size_t)K, GGL_PH_EIG_L, GGL_PH_RECON_L);
    PB(c, GGL_PH_EIG_L);
    // |C|_2 bound.  From P = C C where the product kernel leaves bound partials (newton_schulz.hip, k_bound_sqrt_inf_fro: 2.4x
    // the spectral radius instead of the 10x of min(|C|_inf, |C|_F) -- about three products of the schedule); P is the first
    // product of the iteration anyway and stays in nsT for the first pass (t0_ready).
    const int btile = c->fused_bounds ? symm_bounds_tile(K, c->p, c->symm_variant) : 0;
    bool have_P = false;
    if (btile) {
        for (int k = 0; k < K; ++k) {
            double* o = c->coef_h + (size_t)k * NS_NCOEF;
            o[0] = 0.0; o[1] = 1.0; o[2] = o[3] = o[4] = o[5] = 0.0;
        }
        CopySegs upP;
        upP.add(c->coef, c->coef_h, (size_t)K * NS_NCOEF * sizeof(double));
        launch_copy_small(c->stream, upP);
        const int bT = (c->p + btile - 1) / btile;
        launch_symm(c->stream, c->W, c->W, c->nsT, nullptr, nullptr, c->coef, K, c->p, c->symm_variant, nullptr, c->rowpart,
                    c->fropart);
        launch_bound_rows(c->stream, c->rowpart, bT, K, c->p, c->nbrow, c->infpart);
        if (c->rank_cw) {
            // one pass over P for the Collatz-Wielandt ratio max_i (|P| v)_i / v_i >= rho(|P|) >= |C|_2^2, v carried across ADMM
            // iterations (any positive v keeps it a bound; the kernel leaves |P| v / |P|_inf for the next call): the row-sum
            // bound is ~2.4x the spectral radius on an ADMM run's C, this one settles near 1.1x -- and every factor 2.6 of
            // slack is a cubic step of the sign iteration (the Omega-step's bound of B' has done this since round 2)
            if (!c->cwvecL[0])
                for (double*& b : c->cwvecL) HIPCHK(malloc_filled(&b, (size_t)K * c->p * sizeof(double), c->stream));
            launch_cw_final(c->stream, c->nsT, c->nbrow, K, c->p, c->infpart, c->fropart, bT * (bT + 1) / 2, c->cwmax, c->cwcnt,
                            c->bounds_h, nullptr, nullptr, nullptr, 0, c->cwL_have ? c->cwvecL[c->cwL_cur] : nullptr,
                            c->cwvecL[c->cwL_cur ^ 1]);
            c->cwL_cur ^= 1;
            c->cwL_have = true;
        } else {
            launch_bound_sqrt_inf_fro(c->stream, c->infpart, bound_rows_blocks(c->p), c->fropart, bT * (bT + 1) / 2, K, c->bounds_h);
        }
        c->rank_units += 1.0;
        have_P = true;
    } else {
        const int nbb = norm_bounds_blocks(c->p);
        launch_norm_bounds(c->stream, c->W, K, c->p, c->nbpart);
        launch_bound_final(c->stream, c->nbpart, nullptr, nbb, K, c->bounds_h, 1);     // min(|C|_inf, |C|_F)
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    sanitize_bounds(c, c->bounds_h, nullptr, 0.0, 1.0);
    std::vector<double> cn(c->bounds_h, c->bounds_h + K);
    c->rank_calls += 1;
    double l0 = (c->rank_hold > 0) ? 1e-10 : c->rank_l0;
    if (c->rank_hold > 0) c->rank_hold -= 1;
    // Two tiers.  The schedule's length is set by the smallest gap between an eigenvalue of C and the threshold over ALL
    // instances of the batch (C4, K = 50: 2e-6 .. 4e-5 of |C - mu I| for the worst instance, 1.5e-4 for the 10 % quantile, 1.3e-3
    // for the median: profiles/r3_c4_lstep_threshold_gaps.txt), and the degree sequence is common to a launch.  So the first
    // pass plans for rank_l0_coarse (32 products at 2e-5 instead of 37), the checks say per instance whether that was
    // enough, and the few instances it was not enough for go on as a compact sub-batch -- from the iterate they have, with
    // the schedule for where an eigenvalue at the FINE resolution would be by now (rank_ns_image).  Same guarantee as the
    // one-tier run (eigenvalues at least rank_l0 |B| away from the threshold are resolved, the check catches the others).
    const double l_fine = l0;
    // Deflation (deflate.hip): the first pass only has to resolve the eigenvalues farther than rank_l0_deflate |B| from the
    // threshold (22 products at 1e-3 instead of 28-32); the one or two per instance that are closer are found as the range
    // of I - X^2 and corrected exactly.  An instance whose residual has more than DEFL_Q0 - 1 directions, whose probes do not
    // vanish or whose trace does not come out an integer goes on with the others that need it as the compact continuation.
    const bool deflate = c->rank_deflate && c->rank_l0_deflate > l_fine && c->p <= deflate_max_p() && c->rank_hold == 0;
    const bool two_tier = deflate || (c->rank_l0_coarse > l_fine && K >= 4);
    // resolutions of the full-batch passes, in order: [coarse (+ continuation of the instances it left),] fine, 1e-10
    double stages[3];
    int nstage = 0;
    if (two_tier) stages[nstage++] = deflate ? c->rank_l0_deflate : c->rank_l0_coarse;
    stages[nstage++] = l_fine;
    if (l_fine > 1e-10) stages[nstage++] = 1e-10;
    const size_t pp = (size_t)c->p * c->p;
    for (int stage = 0; stage < nstage; ++stage) {
        l0 = stages[stage];
        const bool coarse = two_tier && stage == 0;
        NsPlan plan;
        if (rank_ns_plan(cn.data(), mu_h, K, l0, c->coef_h, &plan, c->ns_degrees) != 0)
            return fail(GGL_E_SOLVER, "L-step: non-finite C (diverged iterate?)");
        CopySegs up;
        up.add(c->coef, c->coef_h, (size_t)plan.products * NS_SLOT(K) * sizeof(double));
        up.add(c->maxdev, nullptr, K * sizeof(double));
        launch_copy_small(c->stream, up);
        // the parts of the batch run their launch sequences concurrently, as in the Omega-step
        const long t64 = (c->p + 63) / 64;
        const long ntile = t64 * (t64 + 1) / 2 * K;
        int nh = (K >= 16 && ntile >= 600 && ntile <= c->parts_max_tiles) ? std::min(c->ns_parts, K / 8) : 1;
        nh = std::max(nh, 1);
        c->last_parts = nh;
        c->last_variant = symm_effective_variant((c->symm_variant >= 0) ? c->symm_variant : (nh > 1 ? 17 : symm_auto_variant(K, c->p)), c->p);
        if (nh > 1) {
            HIPCHK(hipEventRecord(c->ev_fork, c->stream));
            for (int h = 1; h < nh; ++h) HIPCHK(hipStreamWaitEvent(c->streamx[h - 1], c->ev_fork, 0));
        }
        const bool t0_ready = have_P && stage == 0;       // nsT still holds P = C C of the bound: T0 in place, no first product
        for (int h = 0, k0 = 0; h < nh; ++h) {
            const int Kr = K / nh + (h < K % nh ? 1 : 0);
            hipStream_t sh = h == 0 ? c->stream : c->streamx[h - 1];
            if (t0_ready) launch_rank_t0(sh, c->nsT + k0 * pp, c->W + k0 * pp, c->coef + NS_NCOEF * (size_t)k0, Kr, c->p);
            // scratch: Xa = nsYP[0], Xb = nsYP[0] + n, P2 = nsYP[1], T = nsT
            rank_ns_run(sh, plan, c->coef + NS_NCOEF * (size_t)k0, c->W + k0 * pp,
                        c->nsYP[0] + k0 * pp, c->nsYP[0] + c->n + k0 * pp, c->nsT + k0 * pp, c->nsYP[1] + k0 * pp,
                        c->L + k0 * pp, c->maxdev + k0, Kr, c->p, (c->symm_variant < 0 && nh > 1) ? 17 : c->symm_variant,
                        NS_SLOT(K), t0_ready);
            k0 += Kr;
        }
        for (int h = 1; h < nh; ++h) {
            HIPCHK(hipEventRecord(c->ev_join[h - 1], c->streamx[h - 1]));
            HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join[h - 1], 0));
        }
        HIPCHK(hipGetLastError());
        c->rank_units += plan.products - (t0_ready ? 1 : 0);      // (the C C product was counted with the bound)
        c->rank_launches = (long long)(c->rank_units + 0.5);
        // the two checks of the result (newton_schulz.hip: rank_check, rank_trace_tolerance): the entrywise residual of the
        // last step, and the distance of trace(X_last) = trace(P2) - p from an integer
        launch_trace(c->stream, c->nsYP[1], K, c->p, (double)c->p, c->maxdev + K);
        const bool defl_stage = coarse && deflate;
        if (defl_stage) {
            if (!c->defl_G) {
                // fixed Gaussian test matrix [DEFL_Q][p] (a deterministic stream: the same solve gives the same bits)
                std::vector<double> g((size_t)DEFL_Q * c->p);
                unsigned long long s = 0x9E3779B97F4A7C15ull;
                auto u01 = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return ((double)(s >> 11) + 0.5) / 9007199254740992.0; };
                for (size_t i = 0; i < g.size(); i += 2) {
                    const double r = std::sqrt(-2.0 * std::log(u01())), a = 6.283185307179586 * u01();
                    g[i] = r * std::cos(a);
                    if (i + 1 < g.size()) g[i + 1] = r * std::sin(a);
                }
                HIPCHK(hipMalloc(&c->defl_G, g.size() * sizeof(double)));
                HIPCHK(hipMemcpy(c->defl_G, g.data(), g.size() * sizeof(double), hipMemcpyHostToDevice));
                HIPCHK(malloc_filled(&c->defl_work, 4 * (size_t)K * DEFL_Q * c->p * sizeof(double), c->stream));
                HIPCHK(malloc_filled(&c->defl_meta, 4 * (size_t)K * sizeof(double), c->stream));
                HIPCHK(hipHostMalloc(&c->defl_meta_h, 4 * (size_t)K * sizeof(double)));
            }
            const double* Xl = ((plan.steps - 1) & 1) ? c->nsYP[0] + c->n : c->nsYP[0];
            launch_deflate(c->stream, Xl, c->W, c->par + 2 * (size_t)K, c->L, c->defl_G, c->defl_work, c->defl_meta, K, c->p, 1e-11, 1e-10);
        }
        CopySegs dn;
        dn.add(c->maxdev_h, c->maxdev, 2 * K * sizeof(double));
        if (defl_stage) dn.add(c->defl_meta_h, c->defl_meta, 4 * (size_t)K * sizeof(double));
        launch_copy_small(c->stream, dn);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(c->stream));
        if (defl_stage) {
            // after the deflation the entrywise residual of the coarse pass says nothing (it is what was deflated); an instance
            // is resolved when its residual had at most DEFL_Q0 - 1 directions, the probes found nothing outside them (noise
            // is ~1e-13) and trace(X) + trace(D) is an integer.  Folded into the two numbers the generic check reads.
            int nd = 0;
            for (int k = 0; k < K; ++k) {
                const double* m = c->defl_meta_h + 4 * (size_t)k;
                const bool ok = m[0] < DEFL_Q0 && m[1] <= 1e-10 && std::isfinite(m[2]);
                c->maxdev_h[k] = ok ? 0.0 : (std::isfinite(c->maxdev_h[k]) ? 1.0 : c->maxdev_h[k]);
                c->maxdev_h[K + k] += m[2];
                nd += m[0] > 0 ? 1 : 0;
            }
            c->rank_deflated_calls += 1;
            c->rank_deflated_instances += nd;
        }
        const double ttol = defl_stage ? 1e-10 : rank_trace_tolerance(l0, c->p);
        auto unresolved = [&](int k, int ktr, double check, double tol) {
            const double t = c->maxdev_h[ktr];
            return !(c->maxdev_h[k] <= check) || !(std::fabs(t - std::nearbyint(t)) <= tol);
        };
        if (c->isolate && c->failed) {
            // an instance that is ALREADY marked (its norm bound was not finite: sanitize_bounds above, or an earlier step)
            // counts as resolved -- its L is garbage in its own slot only, the host parks the slot on the identity problem --
            // so that it does not drag the batch through the retries and the eigh fallback.  A non-finite residual of an
            // instance that is NOT marked is not a verdict on the instance (round 5 marked it here, at whatever stage, and the
            // point was lost although the finer pass or the eigendecomposition would have served it -- ADVICE r5): it counts
            // as unresolved like any other failed check and goes the next stage's way; if the eigendecomposition at the end
            // cannot serve it either, check_info / the non-finite sums of the stopping test report it.
            for (int k = 0; k < K; ++k)
                if (c->failed[k] && (!std::isfinite(c->maxdev_h[k]) || !std::isfinite(c->maxdev_h[K + k]))) {
                    c->maxdev_h[k] = 0.0;
                    c->maxdev_h[K + k] = 0.0;
                }
        }
        bool finite = true, all_ok = true;
        for (int k = 0; k < K; ++k) {
            finite = finite && std::isfinite(c->maxdev_h[k]) && std::isfinite(c->maxdev_h[K + k]);
            all_ok = all_ok && !unresolved(k, K + k, plan.check, ttol);
        }
        if (finite && all_ok) {
            PE(c, GGL_PH_EIG_L);
            return GGL_OK;
        }
        if (coarse) {
            if (!finite) continue;                                  // (next stage reports a non-finite C)
            // the instances the first pass did not resolve go on as a compact sub-batch
            std::vector<int> bad;
            for (int k = 0; k < K; ++k)
                if (unresolved(k, K + k, plan.check, ttol)) bad.push_back(k);
            const int m = (int)bad.size();
            const double lp = rank_ns_image(l0, c->ns_degrees, l_fine) * (1.0 - 1e-9);
            if (2 * m > K || !(lp > 0.0) || !(lp < 0.999)) continue;   // too many for a compact batch: the whole batch at l_fine
            if (!c->rank_idx) {
                HIPCHK(malloc_filled(&c->rank_idx, K * sizeof(int), c->stream));
                HIPCHK(hipHostMalloc(&c->rank_idx_h, K * sizeof(int)));
            }
            std::vector<double> mu2(m);
            for (int i = 0; i < m; ++i) { c->rank_idx_h[i] = bad[i]; mu2[i] = mu_h[bad[i]]; }
            NsPlan plan2;
            if (rank_ns_plan_continue(mu2.data(), m, lp, c->coef_h, &plan2, c->ns_degrees, NS_SLOT(K)) != 0) continue;
            const size_t mp = (size_t)m * pp;
            // free after the first pass: nsT, both halves of nsYP[1]; the X buffer the last step wrote holds the iterate
            double* Cc = c->nsT;
            double* Xc = c->nsT + mp;
            double* Xnc = c->nsYP[1];
            double* Tbc = c->nsYP[1] + mp;
            double* P2c = c->nsYP[1] + 2 * mp;
            double* outc = c->nsYP[1] + 3 * mp;
            const double* Xlast = ((plan.steps - 1) & 1) ? c->nsYP[0] + c->n : c->nsYP[0];
            HIPCHK(hipMemcpyAsync(c->rank_idx, c->rank_idx_h, m * sizeof(int), hipMemcpyHostToDevice, c->stream));
            CopySegs up2;
            up2.add(c->coef, c->coef_h, (size_t)plan2.products * NS_SLOT(K) * sizeof(double));
            up2.add(c->maxdev, nullptr, m * sizeof(double));
            launch_copy_small(c->stream, up2);
            launch_copy_instances(c->stream, Cc, c->W, c->rank_idx, m, pp, false);
            launch_copy_instances(c->stream, Xc, Xlast, c->rank_idx, m, pp, false);
            rank_ns_steps(c->stream, plan2, c->coef, Cc, Xc, Xnc, Tbc, P2c, outc, c->maxdev, m, c->p, c->symm_variant,
                          NS_SLOT(K));
            launch_copy_instances(c->stream, c->L, outc, c->rank_idx, m, pp, true);
            launch_trace(c->stream, P2c, m, c->p, (double)c->p, c->maxdev + K);
            CopySegs dn2;
            dn2.add(c->maxdev_h, c->maxdev, m * sizeof(double));
            dn2.add(c->maxdev_h + K, c->maxdev + K, m * sizeof(double));
            launch_copy_small(c->stream, dn2);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(c->stream));
            c->rank_units += (double)plan2.products * m / K;
            c->rank_launches = (long long)(c->rank_units + 0.5);
            c->rank_continued += 1;
            c->rank_cont_instances += m;
            bool ok2 = true;
            const double ttol2 = rank_trace_tolerance(l_fine, c->p);
            for (int i = 0; i < m; ++i)
                ok2 = ok2 && std::isfinite(c->maxdev_h[i]) && std::isfinite(c->maxdev_h[K + i]) &&
                      !unresolved(i, K + i, plan2.check, ttol2);
            if (ok2) {
                PE(c, GGL_PH_EIG_L);
                return GGL_OK;
            }
            // an eigenvalue within l_fine |B| of the threshold (what a failed fine pass means in the one-tier run, which then
            // repeats the WHOLE batch at 1e-10): the compact batch again, from scratch, at 1e-10
            c->rank_retries += 1;
            c->rank_hold = 8;
            if (l_fine > 1e-10) {
                std::vector<double> cn2(m);
                for (int i = 0; i < m; ++i) cn2[i] = cn[bad[i]];
                NsPlan plan3;
                if (rank_ns_plan(cn2.data(), mu2.data(), m, 1e-10, c->coef_h, &plan3, c->ns_degrees) != 0)
                    return fail(GGL_E_SOLVER, "L-step: non-finite C (diverged iterate?)");
                CopySegs up3;
                up3.add(c->coef, c->coef_h, (size_t)plan3.products * NS_SLOT(m) * sizeof(double));
                up3.add(c->maxdev, nullptr, m * sizeof(double));
                launch_copy_small(c->stream, up3);
                rank_ns_run(c->stream, plan3, c->coef, Cc, Xc, Xnc, Tbc, P2c, outc, c->maxdev, m, c->p, c->symm_variant,
                            NS_SLOT(m));
                launch_copy_instances(c->stream, c->L, outc, c->rank_idx, m, pp, true);
                launch_trace(c->stream, P2c, m, c->p, (double)c->p, c->maxdev + K);
                launch_copy_small(c->stream, dn2);
                HIPCHK(hipGetLastError());
                HIPCHK(hipStreamSynchronize(c->stream));
                c->rank_units += (double)plan3.products * m / K;
                c->rank_launches = (long long)(c->rank_units + 0.5);
                bool ok3 = true;
                const double ttol3 = rank_trace_tolerance(1e-10, c->p);
                for (int i = 0; i < m; ++i)
                    ok3 = ok3 && std::isfinite(c->maxdev_h[i]) && std::isfinite(c->maxdev_h[K + i]) &&
                          !unresolved(i, K + i, plan3.check, ttol3);
                if (ok3) {
                    PE(c, GGL_PH_EIG_L);
                    return GGL_OK;
                }
            }
            break;                              // -> the eigendecomposition
        }
        c->rank_retries += 1;
        c->rank_hold = 8;       // an eigenvalue sits within l0*|B| of the threshold: stay fine for a while
    }
    PE(c, GGL_PH_EIG_L);
    c->rank_fallbacks += 1;
    return eig_recon(c, c->W, c->L, c->DvL, MAP_RANK, c->par + 2 * (size_t)K, -1, GGL_PH_RECON_L);
}

static int ggl_step_finish_impl(ggl_ctx* c, double rho, double lambda1, double lambda2, int reg, int latent,
                                const double* mu1, int groupsq_ready, double out_norms[5]);

extern "C" int ggl_step_finish(ggl_ctx* c, double rho, double lambda1, double lambda2, int reg, int latent,
                               const double* mu1, int groupsq_ready, double out_norms[5])
{
    ARGCHK(c, "ctx");
    DROP_PRE(c);
    return ggl_step_finish_impl(c, rho, lambda1, lambda2, reg, latent, mu1, groupsq_ready, out_norms);
}

static int ggl_step_finish_impl(ggl_ctx* c, double rho, double lambda1, double lambda2, int reg, int latent,
                                const double* mu1, int groupsq_ready, double out_norms[5])
{
    ARGCHK(c && out_norms, "ctx, out_norms");
    ARGCHK(rho > 0, "rho must be positive");
    ARGCHK(reg == GGL_REG_SGL || reg == GGL_REG_GGL || reg == GGL_REG_FGL, "reg");
    ARGCHK(!latent || mu1, "latent needs mu1");
    HIPCHK(hipSetDevice(c->device));
    const bool defer_norms = (groupsq_ready & 2) != 0;
    groupsq_ready &= 1;
    // the five sums go straight to pinned host memory, unless a K-sharded run wants them on the device first
    double* norms_dst = defer_norms ? c->norms : c->norms_h;
    c->norms_host = !defer_norms;
    const double inv_rho = 1.0 / rho;
    const double l1 = inv_rho * lambda1, l2 = inv_rho * lambda2;   // admm_solver.py:191-192
    double* Om = c->Om[c->cur];
    double* OmPrev = c->Om[c->cur ^ 1];
    int rows = 1;
    if (reg == GGL_REG_SGL) {
        int rc = upload_par(c, 1, nullptr, l1, 1.0);
        if (rc) return rc;
        rc = upload_par(c, 4, nullptr, inv_rho, 1.0);
        if (rc) return rc;
        PB(c, GGL_PH_THETA);
        launch_theta_sgl(c->stream, c->Theta, c->X, c->W, Om, OmPrev, latent ? c->L : nullptr, c->par + c->K,
                         c->has_mask ? c->mask : nullptr, c->par + 4 * (size_t)c->K, latent, c->partials, c->K, c->p,
                         c->spec_pending ? c->spec_flag : nullptr);
        PE(c, GGL_PH_THETA);
        HIPCHK(hipGetLastError());
        if (!latent) {
            PB(c, GGL_PH_REDUCE);
            launch_reduce_partials(c->stream, c->partials, c->K, elementwise_blocks(c->p), GGL_NNORM, norms_dst);
            PE(c, GGL_PH_REDUCE);
            rows = c->K;
        }
    } else {
        ARGCHK(lambda1 > 0 && lambda2 > 0, "lambda1, lambda2 must be positive");
        if (reg == GGL_REG_FGL && c->K > fgl_max_K())
            return fail(GGL_E_ARG, "FGL Theta-step: K = %d exceeds the %d instances whose K-vectors fit the LDS scan buffer "
                        "of one workgroup (solver/fgl_helper.py:11-68 is a serial scan along K)", c->K, fgl_max_K());
        PB(c, GGL_PH_THETA);
        // K-sharded: the reduced flag (behind the packed sums) decides for every rank, whether it speculated itself or not --
        // the Theta kernels read it there, the host gets it with the norms (finish_norms)
        if (groupsq_ready && c->omega_ns) c->sharded_check = true;
        // the flat GGL kernel computes every (i,j) from its own inputs: only for an exactly symmetric state
        const int flat = (c->theta_flat && c->state_symmetric) ? c->theta_flat : 0;
        // the early first part of the next chain will follow (same conditions as maybe_early): let the kernel write that
        // chain's W = Theta - X - beta S itself -- beta is this iteration's, which is what the early part assumes
        WNext wn;
        int wn_done = 0;
        if (c->fused_w && !latent && !groupsq_ready && reg == GGL_REG_GGL && flat && c->S_symmetric && early_wanted(c)) {
            wn.S = c->S;
            wn.beta = c->par;
        }
        HIPCHK(launch_theta_pair(c->stream, reg, c->Theta, c->X, c->W, Om, OmPrev, latent ? c->L : nullptr, l1, l2,
                                 groupsq_ready ? c->groupsq : nullptr, c->sqwork, latent ? 0 : 1, c->partials, c->K,
                                 c->p, flat, (c->spec_pending || c->sharded_check) ? c->spec_flag : nullptr, wn, &wn_done));
        c->wf_ready = wn_done != 0;
        if (c->wf_ready) { memcpy(c->wf_beta, c->par_h, c->K * sizeof(double)); c->wf_written += 1; }
        trace_mark(c, c->stream, 20);
        PE(c, GGL_PH_THETA);
        if (!latent) {
            PB(c, GGL_PH_REDUCE);
            if (!defer_norms && c->seq_h && c->spin_wait) c->stamp_want = c->seq_wait = ++c->seq_next;
            // GGL_OPT_REDUCE_RIDER: the early first part of the next chain follows and its first launch is A' (W written by the
            // Theta kernel above): the reduction rides in that launch (RedRider) -- maybe_early below hands it over, and
            // launches it after all if no A' came
            c->red_pending = RedRider{};
            // MEASURED (profiles/r5_reduce_rider_ab.txt, three interleaved rounds per workload, always / off): K = 4 slab +3.9 %,
            // K = 8 +1.8 %, (20,200) +2.5 %, K = 16 +1.7 %, headline +0.6 %, (64,100) +3 %, (32,128) +2.8 %.
            if ((c->red_rider == 2 || (c->red_rider == 1 && c->last_parts == 1)) && wn_done && c->seq_wait && !defer_norms &&
                c->prof_on != 1) {
                c->red_pending.partials = c->partials;
                c->red_pending.nblk = theta_partial_blocks(c->p, reg, c->K, flat);
                c->red_pending.nv = GGL_NNORM;
                c->red_pending.out = norms_dst;
                c->red_pending.seq = c->seq_h;
                c->red_pending.seq_val = c->seq_wait;
            } else {
                launch_reduce_partials(c->stream, c->partials, 1, theta_partial_blocks(c->p, reg, c->K, flat), GGL_NNORM,
                                       norms_dst, c->seq_wait ? c->seq_h : nullptr, c->seq_wait);
                trace_mark(c, c->stream, 21);
            }
            trace_host(c, 101);
            PE(c, GGL_PH_REDUCE);
            rows = 1;
        }
    }
    if (latent) {
        int rc = upload_par(c, 2, mu1, 0.0, rho);   // mu1_k / rho   (admm_solver.py:202)
        if (rc) return rc;
        rc = rank_step(c);
        if (rc) return rc;
        PB(c, GGL_PH_DUAL);
        launch_dual_update(c->stream, c->X, Om, OmPrev, c->Theta, c->L, c->partials, c->K, c->p);
        PE(c, GGL_PH_DUAL);
        PB(c, GGL_PH_REDUCE);
        if (defer_norms) {
            // K-sharded latent run: ONE row of sums over the whole local slab, so that the all-reduce over ranks covers 5
            // doubles as in the non-latent case (the unsharded path keeps per-instance rows and adds them on the host)
            launch_reduce_partials(c->stream, c->partials, 1, c->K * elementwise_blocks(c->p), GGL_NNORM, norms_dst);
            rows = 1;
        } else {
            launch_reduce_partials(c->stream, c->partials, c->K, elementwise_blocks(c->p), GGL_NNORM, norms_dst);
            rows = c->K;
        }
        PE(c, GGL_PH_REDUCE);
    }
    HIPCHK(hipGetLastError());
    if (defer_norms) {
        // K-sharded run: the caller all-reduces the five sums in NORMS on the device, then ggl_norms_read
        if (rows != 1) return fail(GGL_E_ARG, "deferred norms need a single row of sums (non-latent GGL/FGL)");
        return GGL_OK;
    }
    if (!latent) {
        const int rce = maybe_early(c);
        if (c->red_pending.nblk > 0) {
            // no early part after all (or an error on the way): the reduction as its own launch
            launch_reduce_partials(c->stream, c->red_pending.partials, 1, c->red_pending.nblk, c->red_pending.nv, c->red_pending.out,
                                   c->red_pending.seq, c->red_pending.seq_val);
            trace_mark(c, c->stream, 21);
            c->red_pending = RedRider{};
        }
        if (rce) return rce;
    }
    return finish_norms(c, rows, out_norms);
}

extern "C" int ggl_norms_read(ggl_ctx* c, double out_norms[5])
{
    ARGCHK(c && out_norms, "ctx, out_norms");
    HIPCHK(hipSetDevice(c->device));
    return finish_norms(c, 1, out_norms);      // 1 = a speculative step failed validation on some rank: repeat it
}

// Early first part of the NEXT iteration's chain (ggl_ctx::EarlyA), called with this iteration's Theta-step and reduction in the
// stream and the host about to wait for them.  Launched on a prediction -- the last iteration's residual ratio was calm, so
// the rho rule will very likely keep rho -- and forgotten if the prediction fails (cost: ~0.2 ms of device time).
static bool early_wanted(const ggl_ctx* c)
{
    if (!c->early_caller || !c->early_part || !c->pipeline || !c->omega_ns || !c->spec_enable || c->prof_on == 1 || c->last_step_hint || !c->ratio_calm ||
        c->chain_mode || c->pre_valid)
        return false;
    if (c->lds_omega && c->p <= omega_lds_max_p()) return false;       // (one kernel writes Omega there: nothing to split)
    return true;
}

static int maybe_early(ggl_ctx* c)
{
    if (!early_wanted(c)) return GGL_OK;
    c->early_request = true;
    const int rc = omega_step(c, 0, nullptr, /*allow_spec=*/true, /*only_spec=*/true);
    c->early_request = false;
    trace_host(c, 102);
    return rc == GGL_NOT_LAUNCHED ? GGL_OK : rc;
}

// Pipelining across iterations (ggl_ctx::pipeline).  take_prelaunched: beta_k = nk/rho of the step about to run is in
// par_h slot 0; if the chain launched at the end of the previous call was built for exactly this beta it becomes this
// iteration's Omega-step, otherwise it is forgotten (the chain the caller launches next follows it on the same streams,
// zeroes its validation flags again and overwrites everything it wrote).
static bool take_prelaunched(ggl_ctx* c, int latent)
{
    if (!c->pre_valid) return false;
    bool have = !latent;
    for (int k = 0; have && k < c->K; ++k) have = (c->par_h[k] == c->pre_beta[k]);
    if (!have) {
        // wait for the forgotten chain before the replacement rewrites the pinned coefficient / bound / flag tables its
        // copy kernels and k_cw_final may still be reading or writing (timing-safe is not safe); rare -- a
        // rho change the rho rule did not predict -- so the synchronisation costs nothing measurable
        (void)drop_prelaunch(c);
        return false;
    }
    c->pre_valid = false;
    c->cur ^= 1;
    c->spec_pending = c->pre_spec_pending;
    c->cw_pending = c->pre_cw_pending;
    return true;
}

// After a validated iteration: keep the GPU busy through the host's round trip.  If the reference's rho rule
// (admm_solver.py:227-233) leaves rho alone for these residuals, the next call will ask for the same beta -- launch its
// Omega-step chain now (beta is in parameter slot 0 already).  out_norms are the sums the caller is about to see.
static int maybe_prelaunch(ggl_ctx* c, double rho, const double out_norms[5])
{
    const bool last = c->last_step_hint;
    c->last_step_hint = false;
    c->ratio_calm = false;
    if (!c->pipeline || last || !c->omega_ns || c->prof_on == 1) { c->early.valid = false; return GGL_OK; }
    const double r_t = std::sqrt(out_norms[3]), s_t = rho * std::sqrt(out_norms[4]);
    if (r_t >= 10.0 * s_t || s_t >= 10.0 * r_t) { c->early.valid = false; return GGL_OK; }
    const int cur0 = c->cur;
    int rc = omega_step(c, 0, nullptr, /*allow_spec=*/true, /*only_spec=*/true);      // (continues an early first part, if there is one)
    if (rc == GGL_NOT_LAUNCHED) return GGL_OK;
    if (rc) return rc;
    // residuals well inside the band in which the rho rule keeps rho: the next iteration may put the first part of ITS
    // successor's chain into the stream before it waits for its own residuals (maybe_early)
    // (the rule acts at a ratio of 10 and the ratio moves by a few per cent per iteration: inside a factor 8 the prediction
    // "rho stays" fails about once per rho change, and a failed prediction costs one forgotten early part.  Round 4 used a
    // factor 4, which tools/event_timeline.py showed to switch the early part OFF for good once a solve's residual ratio
    // settles between 4 and 10 -- C3 from iteration ~40 on.)
    c->ratio_calm = (r_t < 8.0 * s_t && s_t < 8.0 * r_t);
    c->cur = cur0;                               // Omega_t stays the current iterate until the chain is taken over
    c->pre_spec_pending = c->spec_pending;
    c->spec_pending = false;
    c->pre_cw_pending = c->cw_pending;
    c->cw_pending = false;
    c->pre_valid = true;
    c->pre_launched += 1;
    memcpy(c->pre_beta, c->par_h, c->K * sizeof(double));
    return GGL_OK;
}

extern "C" int ggl_hint_last_step(ggl_ctx* c)
{
    ARGCHK(c, "ctx");
    c->last_step_hint = true;
    return GGL_OK;
}

extern "C" int ggl_admm_step(ggl_ctx* c, double rho, double lambda1, double lambda2, int reg, int latent,
                             const double* mu1, const double* nk, double out_norms[5])
{
    ARGCHK(c, "ctx");
    ARGCHK(rho > 0, "rho must be positive");
    HIPCHK(hipSetDevice(c->device));
    trace_host(c, 100);
    CopySegs sg;
    int rc = upload_par(c, 0, nk, 1.0, rho, &sg);   // beta_k = nk/rho    (admm_solver.py:180,184)
    if (rc) return rc;
    if (!take_prelaunched(c, latent)) {
        c->pending_beta_only = true;
        rc = omega_step(c, latent, &sg, /*allow_spec=*/true);
        c->pending_beta_only = false;
        if (rc) return rc;
    }
    c->early_caller = true;
    rc = ggl_step_finish_impl(c, rho, lambda1, lambda2, reg, latent, mu1, 0, out_norms);
    c->early_caller = false;
    if (rc == GGL_SPEC_RETRY) {
        // the speculative schedule did not cover this iteration's spectrum: same step again, bounds first
        rc = omega_step(c, latent, nullptr, false);
        if (rc) return rc;
        rc = ggl_step_finish_impl(c, rho, lambda1, lambda2, reg, latent, mu1, 0, out_norms);
    }
    if (rc != GGL_OK || latent) return rc;
    trace_host(c, 103);
    rc = maybe_prelaunch(c, rho, out_norms);
    trace_host(c, 104);
    return rc;
}

// ---- K independent single problems with their own rho / lambda1 (batched lambda path) ----------
static int sgl_batch_step_impl(ggl_ctx* c, const double* rho, const double* lambda1, int latent, const double* mu1,
                               double* out_norms);
static int sgl_fused_finish(ggl_ctx* c, const double* rho, const double* lambda1, double* out_norms);

extern "C" int ggl_sgl_batch_step(ggl_ctx* c, const double* rho, const double* lambda1, int latent, const double* mu1,
                                  double* out_norms)
{
    ARGCHK(c && rho && lambda1 && out_norms, "ctx, rho, lambda1, out_norms");
    ARGCHK(!latent || mu1, "latent needs mu1");
    HIPCHK(hipSetDevice(c->device));
    DROP_PRE(c);
    return sgl_batch_step_impl(c, rho, lambda1, latent, mu1, out_norms);
}

// After the fused launch of a batch of single problems (k_omega_lds<.., SGL>): wait for the sequence number its last workgroup
// publishes; the rows of sums are in pinned memory.  Instances the kernel could not serve (condition number of W^2 + 4 beta I
// above 300, non-finite data: their iterate is untouched) are redone ALONE on the launch chain -- a compact ctx of just those
// instances (ggl_ctx_create_subset) runs the ordinary step and its Omega, Theta, X and sums are scattered back -- and the
// kernel sits out the next few steps as after any miss.
static int sgl_fused_finish(ggl_ctx* c, const double* rho, const double* lambda1, double* out_norms)
{
    const int K = c->K;
    c->sgl_done = false;
    c->sgl_fused_calls += 1;
    c->norms_host = true;
    double* rows = c->norms_h;
    {
        // (finish_norms' wait, without its validation of a speculative step: nothing here is speculative)
        const unsigned long long want = c->seq_wait;
        const volatile unsigned long long* sq = c->seq_h;
        const auto t0 = std::chrono::steady_clock::now();
        bool waited = false;
        for (unsigned spin = 1;; ++spin) {
            if (*sq == want) { waited = true; break; }
            __builtin_ia32_pause();
            if ((spin & 0xfff) == 0 &&
                std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(GGL_SPIN_LIMIT_MS)) {
                c->spin_timeouts += 1;
                break;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        c->seq_wait = 0;
        if (!waited) {
            HIPCHK(hipStreamSynchronize(c->stream));
            if (*sq != want) return fail(GGL_E_HIP, "fused SGL step: sequence number %llu not published (found %llu)", want, (unsigned long long)*sq);
        }
    }
    c->norms_host = false;
    memcpy(out_norms, rows, (size_t)K * GGL_NNORM * sizeof(double));
    if (c->spec_flag_h[0] == 0) {
        // every instance served: the bounds the kernel used are validated ones (what validate_spec keeps for the launch chain)
        c->lds_cool_next = 4;
        c->spec_have = true;
        sanitize_bounds(c, c->bounds_h, c->par_h, 4.0);
        for (int k = 0; k < K; ++k) { c->spec_c[k] = c->bounds_h[k]; c->spec_beta[k] = c->par_h[k]; }
        return GGL_OK;
    }
    // ---- some instances fell outside the kernel's range ----
    lds_missed(c);
    c->spec_have = false;
    HIPCHK(hipMemsetAsync(c->spec_flag, 0, ggl_ctx::MAX_PARTS * sizeof(int), c->stream));
    for (int h = 0; h < ggl_ctx::MAX_PARTS; ++h) c->spec_flag_h[h] = 0;
    std::vector<int> idx;
    for (int k = 0; k < K; ++k) if (c->sgl_fail_h[k]) idx.push_back(k);
    const int m = (int)idx.size();
    if (m == 0) return fail(GGL_E_SOLVER, "fused SGL step: the flag is raised but no instance is marked");
    c->sgl_fallback_instances += m;
    // the compact ctx takes Omega_t as its current iterate: the fused launch has flipped `cur` already
    c->cur ^= 1;
    ggl_ctx* sub = nullptr;
    int rc = ggl_ctx_create_subset(c, idx.data(), m, &sub);
    c->cur ^= 1;
    if (rc) return rc;
    sub->lds_omega = false;
    std::vector<double> r(m), l(m), on((size_t)m * GGL_NNORM);
    for (int i = 0; i < m; ++i) { r[i] = rho[idx[i]]; l[i] = lambda1[idx[i]]; }
    rc = sgl_batch_step_impl(sub, r.data(), l.data(), 0, nullptr, on.data());
    int* didx = nullptr;
    if (!rc && hipMalloc(&didx, m * sizeof(int)) != hipSuccess) rc = fail(GGL_E_HIP, "fused SGL step: allocation failed");
    if (!rc) {
        const size_t pp = (size_t)c->p * c->p;
        hipError_t e = hipMemcpyAsync(didx, idx.data(), m * sizeof(int), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            launch_copy_instances(c->stream, c->Om[c->cur], sub->Om[sub->cur], didx, m, pp, true);
            launch_copy_instances(c->stream, c->Theta, sub->Theta, didx, m, pp, true);
            launch_copy_instances(c->stream, c->X, sub->X, didx, m, pp, true);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(GGL_E_HIP, "fused SGL step: %s", hipGetErrorString(e));
        for (int i = 0; i < m && !rc; ++i) {
            memcpy(out_norms + (size_t)idx[i] * GGL_NNORM, on.data() + (size_t)i * GGL_NNORM, GGL_NNORM * sizeof(double));
            if (sub->failed && sub->failed[i]) mark_failed(c, idx[i], sub->fail_why ? sub->fail_why[i] : 4, sub->fail_value ? sub->fail_value[i] : 0.0);
        }
    }
    if (didx) (void)hipFree(didx);
    (void)ggl_ctx_destroy(sub);
    return rc;
}

static int sgl_batch_step_impl(ggl_ctx* c, const double* rho, const double* lambda1, int latent, const double* mu1,
                               double* out_norms)
{
    const int K = c->K;
    for (int k = 0; k < K; ++k) ARGCHK(rho[k] > 0, "rho must be positive");
    double* h = c->par_h;
    for (int k = 0; k < K; ++k) {
        const double ir = 1.0 / rho[k];
        h[k] = ir;                         // beta_k = 1/rho_k           (single_admm_solver.py:163,166)
        h[K + k] = ir * lambda1[k];        // (1/rho) * lambda1          (:169)
        h[2 * K + k] = latent ? mu1[k] / rho[k] : 0.0;   // mu1/rho      (:175)
        h[4 * K + k] = ir;
    }
    CopySegs sg;
    sg.add(c->par, h, 5 * (size_t)K * sizeof(double));
    ARGCHK(!(c->has_dims && latent), "padded instances of different dimension: not with latent variables");
    int rc = GGL_OK;
    for (int attempt = 0; attempt < 2; ++attempt) {
        // (first attempt: the LDS-resident Omega-step may run unvalidated; when an instance falls outside its range the
        // Theta-step has left the iterate alone and the step is repeated on the launch chain)
        LdsSgl req;
        if (attempt == 0 && !latent) {
            // p <= 64: ask for the fused form -- Omega-step, Theta-step, dual update and sums in ONE launch (omega_lds.hip)
            req.l1K = c->par + K;
            req.mask = c->has_maskK ? c->maskK : (c->has_mask ? c->mask : nullptr);
            req.mask_stride = c->has_maskK ? (size_t)c->p * c->p : 0;
            req.invrhoK = c->par + 4 * (size_t)K;
            req.pk = c->has_dims ? c->inst_pk : nullptr;
            c->sgl_req = &req;
        }
        rc = attempt == 0 ? omega_step(c, latent, &sg, /*allow_spec=*/!latent) : omega_step(c, latent, nullptr, false);
        c->sgl_req = nullptr;
        if (rc) return rc;
        if (c->sgl_done) return sgl_fused_finish(c, rho, lambda1, out_norms);
        double* Om = c->Om[c->cur];
        double* OmPrev = c->Om[c->cur ^ 1];
        PB(c, GGL_PH_THETA);
        launch_theta_sgl(c->stream, c->Theta, c->X, c->W, Om, OmPrev, latent ? c->L : nullptr, c->par + K,
                         c->has_maskK ? c->maskK : (c->has_mask ? c->mask : nullptr), c->par + 4 * (size_t)K, latent, c->partials,
                         K, c->p, c->spec_pending ? c->spec_flag : nullptr, c->has_dims ? c->inst_pk : nullptr,
                         c->has_maskK ? (size_t)c->p * c->p : 0);
        PE(c, GGL_PH_THETA);
        HIPCHK(hipGetLastError());
        if (latent) {
            rc = rank_step(c);
            if (rc) return rc;
            PB(c, GGL_PH_DUAL);
            launch_dual_update(c->stream, c->X, Om, OmPrev, c->Theta, c->L, c->partials, K, c->p);
            PE(c, GGL_PH_DUAL);
        }
        PB(c, GGL_PH_REDUCE);
        // the K rows of sums go to pinned memory, the last workgroup publishes a sequence number: the host polls that word
        // instead of synchronising the stream (~10 us of a 70-us batch iteration at p <= 64)
        if (c->seq_h && c->spin_wait) c->seq_wait = ++c->seq_next;
        launch_reduce_partials(c->stream, c->partials, K, elementwise_blocks(c->p), GGL_NNORM, c->norms_h,
                               c->seq_wait ? c->seq_h : nullptr, c->seq_wait, c->arrive);
        PE(c, GGL_PH_REDUCE);
        HIPCHK(hipGetLastError());
        c->norms_host = true;
        rc = finish_norms(c, K, out_norms, 1);          // (waits, validates a speculative step, checks the eigensolver's status)
        if (rc != GGL_SPEC_RETRY) break;
    }
    return rc;
}

static int ensure_partials(ggl_ctx* c, size_t need)
{
    if (need <= c->partials_len) return GGL_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    // (the first partials buffer is part of the ctx's device arena: only a grown one is an allocation of its own)
    if (c->partials_own) (void)hipFree(c->partials_own);
    c->partials_own = nullptr;
    c->partials = nullptr;
    c->partials_len = 0;
    HIPCHK(malloc_filled(&c->partials_own, need * sizeof(double), c->stream));
    c->partials = c->partials_own;
    c->partials_len = need;
    return GGL_OK;
}

// ---- G independent multiple-graph problems of K instances each (batched model-selection grid) -------------------
static int mgl_batch_finish(ggl_ctx* c, int G, int Kp, int reg, int latent, double* out_norms)
{
    const int K = c->K;
    double* Om = c->Om[c->cur];
    double* OmPrev = c->Om[c->cur ^ 1];
    const int* skip = c->spec_pending ? c->spec_flag : nullptr;
    // (par0_stale: this step's LDS kernel read its beta from the pinned mirror and the parameter copy was skipped -- the
    // Theta kernel takes its two thresholds per problem from there as well)
    const double* parb = c->par0_stale ? c->par_h : c->par;
    PB(c, GGL_PH_THETA);
    HIPCHK(launch_theta_batch(c->stream, reg, c->Theta, c->X, c->W, Om, OmPrev, latent ? c->L : nullptr, parb + K,
                              parb + 6 * (size_t)K, latent ? 0 : 1, c->partials, G, Kp, c->p, skip));
    PE(c, GGL_PH_THETA);
    int rows, group;
    c->norms_host = true;
    if (!latent) {
        PB(c, GGL_PH_REDUCE);
        if (c->seq_h && c->spin_wait) c->seq_wait = ++c->seq_next;
        launch_reduce_partials(c->stream, c->partials, G, theta_partial_blocks(c->p, reg, Kp, 2, G), GGL_NNORM, c->norms_h,
                               c->seq_wait ? c->seq_h : nullptr, c->seq_wait, c->arrive);
        PE(c, GGL_PH_REDUCE);
        rows = G;
        group = 1;
    } else {
        int rc = rank_step(c);
        if (rc) return rc;
        PB(c, GGL_PH_DUAL);
        launch_dual_update(c->stream, c->X, Om, OmPrev, c->Theta, c->L, c->partials, K, c->p);
        PE(c, GGL_PH_DUAL);
        PB(c, GGL_PH_REDUCE);
        if (c->seq_h && c->spin_wait) c->seq_wait = ++c->seq_next;
        launch_reduce_partials(c->stream, c->partials, K, elementwise_blocks(c->p), GGL_NNORM, c->norms_h,
                               c->seq_wait ? c->seq_h : nullptr, c->seq_wait, c->arrive);
        PE(c, GGL_PH_REDUCE);
        rows = K;
        group = Kp;
    }
    HIPCHK(hipGetLastError());
    return finish_norms(c, rows, out_norms, group);
}

static int mgl_batch_step_impl(ggl_ctx* c, int G, const double* rho, const double* lambda1, const double* lambda2,
                               int reg, int latent, const double* mu1, const double* nk, double* out_norms);

extern "C" int ggl_mgl_batch_step(ggl_ctx* c, int G, const double* rho, const double* lambda1, const double* lambda2,
                                  int reg, int latent, const double* mu1, const double* nk, double* out_norms)
{
    ARGCHK(c && rho && lambda1 && lambda2 && out_norms, "ctx, rho, lambda1, lambda2, out_norms");
    ARGCHK(reg == GGL_REG_GGL || reg == GGL_REG_FGL, "reg");
    ARGCHK(G >= 1 && c->K % G == 0, "the ctx holds G problems of K/G instances each");
    ARGCHK(!latent || mu1, "latent needs mu1");
    ARGCHK(c->state_symmetric, "the batched Theta-step needs exactly symmetric dual / latent start points");
    HIPCHK(hipSetDevice(c->device));
    DROP_PRE(c);
    return mgl_batch_step_impl(c, G, rho, lambda1, lambda2, reg, latent, mu1, nk, out_norms);
}

static int mgl_batch_step_impl(ggl_ctx* c, int G, const double* rho, const double* lambda1, const double* lambda2,
                               int reg, int latent, const double* mu1, const double* nk, double* out_norms)
{
    const int K = c->K, Kp = K / G;
    if (reg == GGL_REG_GGL && Kp > GGL_FLAT_MAX_K)
        return fail(GGL_E_ARG, "batched GGL grid: %d instances per problem exceed the %d of the per-element Theta kernel", Kp,
                    GGL_FLAT_MAX_K);
    if (reg == GGL_REG_FGL && Kp > fgl_max_K())
        return fail(GGL_E_ARG, "batched FGL grid: %d instances per problem exceed the %d of the Condat tile kernel", Kp,
                    fgl_max_K());
    int rc = ensure_partials(c, (size_t)G * theta_partial_blocks(c->p, reg, Kp, 2, G) * GGL_NNORM);
    if (rc) return rc;
    double* h = c->par_h;
    for (int g = 0; g < G; ++g) {
        ARGCHK(rho[g] > 0 && lambda1[g] > 0 && lambda2[g] > 0, "rho, lambda1, lambda2 must be positive");
        const double ir = 1.0 / rho[g];
        for (int k = 0; k < Kp; ++k) {
            const int i = g * Kp + k;
            h[i] = (nk ? nk[k] : 1.0) / rho[g];           // beta = nk / rho              (admm_solver.py:180,184)
            h[K + i] = ir * lambda1[g];                   // (1/rho) lambda1              (:191)
            h[2 * (size_t)K + i] = latent ? mu1[i] / rho[g] : 0.0;   // mu1_k / rho     (:202)
            h[6 * (size_t)K + i] = ir * lambda2[g];       // (1/rho) lambda2              (:192)
        }
    }
    CopySegs sg;
    sg.add(c->par, h, 3 * (size_t)K * sizeof(double));
    sg.add(c->par + 6 * (size_t)K, h + 6 * (size_t)K, (size_t)K * sizeof(double));
    c->pending_pinned_ok = !latent;
    rc = omega_step(c, latent, &sg, /*allow_spec=*/true);
    c->pending_pinned_ok = false;
    if (rc) return rc;
    rc = mgl_batch_finish(c, G, Kp, reg, latent, out_norms);
    if (rc != GGL_SPEC_RETRY) return rc;
    rc = omega_step(c, latent, nullptr, false);
    if (rc) return rc;
    return mgl_batch_finish(c, G, Kp, reg, latent, out_norms);
}

extern "C" int ggl_scale_X_batch(ggl_ctx* c, const double* factor)
{
    ARGCHK(c && factor, "ctx, factor");
    HIPCHK(hipSetDevice(c->device));
    DROP_PRE(c);
    double* h = c->par_h + 5 * (size_t)c->K;
    memcpy(h, factor, c->K * sizeof(double));
    CopySegs sg;
    sg.add(c->par + 5 * (size_t)c->K, h, c->K * sizeof(double));
    launch_copy_small(c->stream, sg);
    launch_scale_batch(c->stream, c->X, c->par + 5 * (size_t)c->K, c->K, c->p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));   // the pinned slot is reused by the next call
    return GGL_OK;
}

// ---- n iterations of a batch per call: the host side of the grid walks in C ------------------------------------------------
// One iteration's decisions for all points of a batch: ADMM_stopping_criterion (solver/admm_solver.py:316-331,
// single_admm_solver.py:277-291) and the residual-balancing rule (:227-233 / :196-206), per point, in the reference's order
// of operations -- bit for bit what gglasso_amd.batch._decide computes with NumPy (tests/test_cpu_batch_decisions.py runs the
// two against each other).  Host only.
//   sq (n,5) squared norms {|Omega|, |Theta - L|, |X|, |Omega - Theta + L|, |Omega - Omega_prev|};  live (n) 0/1;  marked (n) 0/1 or
//   NULL (points the library marked: they end like points with non-finite sums);  rho (n) in/out;  dims (n);
//   last (n,4) = {r_t, s_t, e_pri, e_dual}, rows of the live finite points are rewritten;  fac (n) out: rho / rho_new (1 for
//   the others);  status (n) out: 0 goes on (or not live), 1 converged in this iteration, 2 failed (non-finite sums / marked).
// Returns the number of points with status != 0.
extern "C" int ggl_batch_decide(int n, const double* sq, const unsigned char* live, const unsigned char* marked, double* rho,
                                const double* dims, double tol, double rtol, int update_rho, double* last, double* fac,
                                int* status)
{
    ARGCHK(n >= 0 && sq && live && rho && dims && last && fac && status, "batch_decide: arguments");
    int events = 0;
    for (int i = 0; i < n; ++i) {
        fac[i] = 1.0;
        status[i] = 0;
        if (!live[i]) continue;
        const double* q = sq + (size_t)i * GGL_NNORM;
        bool finite = !(marked && marked[i]);
        for (int j = 0; j < GGL_NNORM; ++j) finite = finite && std::isfinite(q[j]);
        if (!finite) { status[i] = 2; events += 1; continue; }
        const double n_om = std::sqrt(q[0]), n_thl = std::sqrt(q[1]), n_x = std::sqrt(q[2]), n_r = std::sqrt(q[3]),
                     n_s = std::sqrt(q[4]);
        const double r = rho[i];
        const double r_t = n_r, s_t = r * n_s;
        const double e_pri = dims[i] * tol + rtol * std::fmax(n_om, n_thl);
        const double e_dual = dims[i] * tol + (rtol * r) * n_x;
        if (update_rho) {
            const double rn = (r_t >= 10 * s_t) ? 2 * r : ((s_t >= 10 * r_t) ? 0.5 * r : 1. * r);
            fac[i] = r / rn;
            rho[i] = rn;
        }
        double* l = last + (size_t)i * 4;
        l[0] = r_t; l[1] = s_t; l[2] = e_pri; l[3] = e_dual;
        if (r_t <= e_pri && s_t <= e_dual) { status[i] = 1; events += 1; }
    }
    return events;
}

// X_k <- fac_g X_k for the instances of every point (group instances per point), queued on the stream without a host wait:
// the pinned slot is rewritten only after the NEXT iteration's synchronisation.
static int batch_rescale(ggl_ctx* c, const double* fac, int n, int group)
{
    bool any = false;
    for (int g = 0; g < n; ++g) any = any || fac[g] != 1.0;
    if (!any) return GGL_OK;
    double* h = c->par_h + 5 * (size_t)c->K;
    for (int g = 0; g < n; ++g)
        for (int k = 0; k < group; ++k) h[(size_t)g * group + k] = fac[g];
    // (the kernel reads its K factors from the pinned slot itself: one launch, no copy in front of it)
    launch_scale_batch(c->stream, c->X, h, c->K, c->p);
    HIPCHK(hipGetLastError());
    return GGL_OK;
}

static void batch_marks(ggl_ctx* c, int n, int group, unsigned char* marked)
{
    for (int g = 0; g < n; ++g) {
        marked[g] = 0;
        if (!c->failed) continue;
        for (int k = 0; k < group; ++k) marked[g] |= c->failed[(size_t)g * group + k] ? 1 : 0;
    }
}

// Up to n_iters iterations of ggl_sgl_batch_step with everything the host loop of gglasso_amd.batch.ADMM_SGL_batch does between
// two of them -- per-point stopping test, rho rule, X rescale (single_admm_solver.py:186-214; the grid walk it serves:
// helper/model_selection.py:619-633) -- done here, per iteration a loop over the K points instead of a Python round trip
// (~100 us where the device needs 50 us at p <= 64).
//   rho (K) in/out;  dims (K): (p_k^2 + p_k) / 2;  status (K) in/out: 0 live, 1 converged, 2 failed -- points that are not 0
//   on entry are finished (dragged along, no decisions);  last (K,4) in/out: {r_t, s_t, e_pri, e_dual} of the last
//   iteration a point was live in;  fin_iter (K) in/out: for a point that finishes during this call, it_base + the
//   iteration of this call it finished in (1-based).
//   snap_ctx == NULL: returns after the first iteration in which a live point converges or fails (the caller collects it).
//   snap_ctx != NULL (may be ctx itself; snap_index (K): the slot in snap_ctx of every slot of ctx): a point that finishes
//   is snapshotted there on the device (ggl_snapshot_state_from: Omega, Theta, L, X at that iteration, after the X rescale
//   -- single_admm_solver.py:205 comes before the break), a failed one is then parked on the identity problem
//   (ggl_reset_instance), and the loop goes on; it returns when every point is finished, when at least stop_after points
//   are (stop_after > 0: the caller may want to compact the batch) or after n_iters.
// Returns the number of iterations run (>= 1), < 0 on error.
struct BatchRun {
    int n, group;                  // points, instances per point
    double* rho; const double* dims; double tol, rtol; int update_rho;
    double* last; int* status; int* fin_iter; int it_base;
    ggl_ctx* snap_ctx; const int* snap_index; int stop_after;
};

static int snapshot_many(ggl_ctx* c, const int* kd, ggl_ctx* src, const int* ks, int n, bool with_state);

// after one iteration's step (sums in sq): decisions, rescale, snapshots; *stop = the call should return now
static int batch_after_step(ggl_ctx* c, const BatchRun& b, const double* sq, int it, bool last_iter, bool* stop)
{
    std::vector<double> fac(b.n);
    std::vector<unsigned char> live(b.n), marked(b.n);
    std::vector<int> ev(b.n);
    for (int g = 0; g < b.n; ++g) live[g] = b.status[g] == 0 ? 1 : 0;
    batch_marks(c, b.n, b.group, marked.data());
    const int events = ggl_batch_decide(b.n, sq, live.data(), marked.data(), b.rho, b.dims, b.tol, b.rtol, b.update_rho, b.last,
                                        fac.data(), ev.data());
    if (events < 0) return events;
    int rc = batch_rescale(c, fac.data(), b.n, b.group);
    if (rc) return rc;
    int finished = 0;
    std::vector<int> kd, ks;
    for (int g = 0; g < b.n; ++g) {
        if (ev[g] != 0) {
            b.status[g] = ev[g];
            b.fin_iter[g] = b.it_base + it + 1;
            if (b.snap_ctx)
                for (int k = 0; k < b.group; ++k) {
                    ks.push_back(g * b.group + k);
                    kd.push_back(b.snap_index[g * b.group + k]);
                }
        }
        finished += b.status[g] != 0 ? 1 : 0;
    }
    if (!ks.empty()) {
        // every point that finishes in this iteration in ONE hand-over, the converged ones together with the failed ones and
        // before those are parked (collecting reads what the ctx knows about the last L-step of the WHOLE batch)
        rc = snapshot_many(b.snap_ctx, kd.data(), c, ks.data(), (int)ks.size(), true);
        if (rc) return rc;
        for (int g = 0; g < b.n; ++g)
            if (ev[g] == 2)
                for (int k = 0; k < b.group; ++k) {
                    rc = ggl_reset_instance(c, g * b.group + k);
                    if (rc) return rc;
                }
    }
    *stop = last_iter || finished == b.n || (events > 0 && (!b.snap_ctx || (b.stop_after > 0 && finished >= b.stop_after)));
    if (*stop) HIPCHK(hipStreamSynchronize(c->stream));       // (the caller reads the state / rewrites the pinned slots next)
    return GGL_OK;
}

extern "C" int ggl_sgl_batch_run(ggl_ctx* c, int n_iters, double* rho, const double* lambda1, int latent, const double* mu1,
                                 const double* dims, double tol, double rtol, int update_rho, double* last, int* status,
                                 int* fin_iter, int it_base, ggl_ctx* snap_ctx, const int* snap_index, int stop_after)
{
    ARGCHK(c && rho && lambda1 && dims && last && status && fin_iter, "ctx, rho, lambda1, dims, last, status, fin_iter");
    ARGCHK(n_iters >= 1, "n_iters >= 1");
    ARGCHK(!latent || mu1, "latent needs mu1");
    ARGCHK(!snap_ctx || snap_index, "snapshots need the destination slots");
    HIPCHK(hipSetDevice(c->device));
    DROP_PRE(c);
    const int K = c->K;
    const BatchRun b = {K, 1, rho, dims, tol, rtol, update_rho, last, status, fin_iter, it_base, snap_ctx, snap_index, stop_after};
    std::vector<double> sq((size_t)K * GGL_NNORM);
    for (int it = 0; it < n_iters; ++it) {
        int rc = sgl_batch_step_impl(c, rho, lambda1, latent, mu1, sq.data());
        if (rc) return rc;
        bool stop = false;
        rc = batch_after_step(c, b, sq.data(), it, it == n_iters - 1, &stop);
        if (rc) return rc;
        if (stop) return it + 1;
    }
    return n_iters;
}

// The same for G multiple-graph problems in one stack (ggl_mgl_batch_step; admm_solver.py:215-237, the grid walk
// helper/model_selection.py:208-224).  rho, lambda1, lambda2, dims, status, fin_iter: (G); last (G,4); snap_index: (K)
// per INSTANCE slot.
extern "C" int ggl_mgl_batch_run(ggl_ctx* c, int G, int n_iters, double* rho, const double* lambda1, const double* lambda2,
                                 int reg, int latent, const double* mu1, const double* nk, const double* dims, double tol,
                                 double rtol, int update_rho, double* last, int* status, int* fin_iter, int it_base,
                                 ggl_ctx* snap_ctx, const int* snap_index, int stop_after)
{
    ARGCHK(c && rho && lambda1 && lambda2 && dims && last && status && fin_iter,
           "ctx, rho, lambda1, lambda2, dims, last, status, fin_iter");
    ARGCHK(n_iters >= 1, "n_iters >= 1");
    ARGCHK(reg == GGL_REG_GGL || reg == GGL_REG_FGL, "reg");
    ARGCHK(G >= 1 && c->K % G == 0, "the ctx holds G problems of K/G instances each");
    ARGCHK(!latent || mu1, "latent needs mu1");
    ARGCHK(!snap_ctx || snap_index, "snapshots need the destination slots");
    ARGCHK(c->state_symmetric, "the batched Theta-step needs exactly symmetric dual / latent start points");
    HIPCHK(hipSetDevice(c->device));
    DROP_PRE(c);
    const BatchRun b = {G, c->K / G, rho, dims, tol, rtol, update_rho, last, status, fin_iter, it_base, snap_ctx, snap_index,
                        stop_after};
    std::vector<double> sq((size_t)G * GGL_NNORM);
    for (int it = 0; it < n_iters; ++it) {
        int rc = mgl_batch_step_impl(c, G, rho, lambda1, lambda2, reg, latent, mu1, nk, sq.data());
        if (rc > 0) return fail(GGL_E_SOLVER, "batched MGL step: speculative step rejected twice");
        if (rc) return rc;
        bool stop = false;
        rc = batch_after_step(c, b, sq.data(), it, it == n_iters - 1, &stop);
        if (rc) return rc;
        if (stop) return it + 1;
    }
    return n_iters;
}

extern "C" int ggl_get_state_k(ggl_ctx* c, int k, double* Omega, double* Theta, double* L, double* X)
{
    ARGCHK(c && k >= 0 && k < c->K, "ctx, k");
    HIPCHK(hipSetDevice(c->device));
    DROP_PRE(c);
    const size_t pp = (size_t)c->p * c->p, nb = pp * sizeof(double), off = (size_t)k * pp;
    if (Omega) HIPCHK(hipMemcpyAsync(Omega, c->Om[c->cur] + off, nb, hipMemcpyDeviceToHost, c->stream));
    if (Theta) HIPCHK(hipMemcpyAsync(Theta, c->Theta + off, nb, hipMemcpyDeviceToHost, c->stream));
    if (L) HIPCHK(hipMemcpyAsync(L, c->L + off, nb, hipMemcpyDeviceToHost, c->stream));
    if (X) HIPCHK(hipMemcpyAsync(X, c->X + off, nb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return GGL_OK;
}

// GGL_OPT_ISOLATE: out[k] = 1 for every instance marked since the ctx was created (non-finite data, eigensolver failure);
// returns how many, < 0 on error.
extern "C" int ggl_failed_reason(ggl_ctx* c, int k, double out[2])
{
    ARGCHK(c && out && k >= 0 && k < c->K, "ctx, out, k");
    const bool f = c->failed && c->failed[k];
    out[0] = f ? (double)c->fail_why[k] : 0.0;
    out[1] = f ? c->fail_value[k] : 0.0;
    return GGL_OK;
}

extern "C" int ggl_failed_instances(ggl_ctx* c, int* out)
{
    ARGCHK(c, "ctx");
    int n = 0;
    for (int k = 0; k < c->K; ++k) {
        const int f = (c->failed && c->failed[k]) ? 1 : 0;
        if (out) out[k] = f;
        n += f;
    }
    return n;
}

// Parks instance k on the identity problem: S_k = Omega_k = Theta_k = I, L_k = X_k = 0 -- a fixed point of every step up to
// the penalties' shrinkage of zeros -- so that a failed point of a batch goes on harmlessly (finite data, shortest schedules)
// while the other points finish.  Its mark (ggl_failed_instances) stays.
extern "C" int ggl_reset_instance(ggl_ctx* c, int k)
{
    ARGCHK(c && k >= 0 && k < c->K, "ctx, k");
    HIPCHK(hipSetDevice(c->device));
    DROP_PRE(c);
    const size_t pp = (size_t)c->p * c->p, off = (size_t)k * pp;
    double* ident[] = {c->S + off, c->Om[0] + off, c->Om[1] + off, c->Theta + off};
    double* zero[] = {c->L + off, c->X + off, c->W + off};
    for (double* z : zero) HIPCHK(hipMemsetAsync(z, 0, pp * sizeof(double), c->stream));
    for (double* a : ident) launch_set_identity(c->stream, a, 1, c->p);
    // ext_ADMM_MGL state (ggl_ext_*): the copy variable Lambda = I, the second dual X1 = 0
    for (double* lam : c->Lam)
        if (lam) launch_set_identity(c->stream, lam + off, 1, c->p);
    if (c->X1) HIPCHK(hipMemsetAsync(c->X1 + off, 0, pp * sizeof(double), c->stream));
    HIPCHK(hipGetLastError());
    c->spec_have = false;
    c->cw_have = false;
    c->cwL_have = false;
    // (l_ns stays: it says where the OTHER instances' L came from -- a point that converges in the iteration another one
    // fails in is still snapshotted with its C and rebuilt by ggl_finalize_L)
    return GGL_OK;
}

// A new ctx holding the m instances idx[0..m) of `src` (their S, Omega, Theta, L, X, masks and dimensions; the options of
// src; nothing carried from earlier iterations): what is left of a batch of independent problems once a good part of it has
// converged goes on in a smaller stack instead of dragging the finished points through every product.
// Device-to-device; src is unchanged and stays valid (its snapshots are the finished points' results).
extern "C" int ggl_ctx_create_subset(ggl_ctx* src, const int* idx, int m, ggl_ctx** out)
{
    ARGCHK(src && idx && out, "ctx, idx, out");
    ARGCHK(m >= 1 && m <= src->K, "1 <= m <= K");
    for (int i = 0; i < m; ++i) ARGCHK(idx[i] >= 0 && idx[i] < src->K, "instance index");
    HIPCHK(hipSetDevice(src->device));
    { int rc_ = drop_prelaunch(src); if (rc_) return rc_; }
    ggl_ctx* c = nullptr;
    int rc = ggl_ctx_create(src->device, m, src->p, src->flags & ~GGL_CTX_STREAM_GIVEN, nullptr, &c);
    if (rc) return rc;
    c->spec_enable = src->spec_enable; c->spec_factor = src->spec_factor; c->ns_force = src->ns_force;
    c->ns_degrees = src->ns_degrees; c->theta_flat = src->theta_flat; c->rank_eig = src->rank_eig;
    c->rank_ns = c->omega_ns && !c->rank_eig; c->ns_parts = src->ns_parts; c->parts_max_tiles = src->parts_max_tiles;
    c->symm_variant = src->symm_variant; c->spin_wait = src->spin_wait; c->fused_bounds = src->fused_bounds;
    c->pipeline = src->pipeline; c->fused_start = src->fused_start; c->parts_small = src->parts_small; c->download_threads = src->download_threads; c->ns_tol = src->ns_tol;
    c->cw_warm = src->cw_warm; c->rank_l0 = src->rank_l0; c->rank_l0_coarse = src->rank_l0_coarse;
    c->group_sched = src->group_sched;
    c->isolate = src->isolate; c->lds_omega = src->lds_omega; c->lds_waves = src->lds_waves; c->early_part = src->early_part; c->rank_deflate = src->rank_deflate; c->rank_l0_deflate = src->rank_l0_deflate;
    int* didx = nullptr;
    hipError_t e = hipMalloc(&didx, m * sizeof(int));
    if (e == hipSuccess) e = hipMemcpyAsync(didx, idx, m * sizeof(int), hipMemcpyHostToDevice, src->stream);
    if (e != hipSuccess) { ggl_ctx_destroy(c); return fail(GGL_E_HIP, "subset: %s", hipGetErrorString(e)); }
    const size_t pp = (size_t)src->p * src->p;
    const double* from[] = {src->S, src->Om[src->cur], src->Om[src->cur ^ 1], src->Theta, src->L, src->X};
    double* to[] = {c->S, c->Om[0], c->Om[1], c->Theta, c->L, c->X};
    for (int i = 0; i < 6; ++i) launch_copy_instances(src->stream, to[i], from[i], didx, m, pp, false);
    c->cur = 0;
    c->state_symmetric = src->state_symmetric;
    c->S_symmetric = src->S_symmetric;
    c->fused_w = src->fused_w;
#ifdef GGL_DEV
    c->parts_bias = src->parts_bias; c->parts_order = src->parts_order; c->chain_mode = src->chain_mode; c->fused_cw = src->fused_cw;
    c->rank_cw = src->rank_cw; c->bound_side = src->bound_side;
#endif
    c->lds_pinned = src->lds_pinned;
    c->join_flag = src->join_flag;
    c->cw_rider = src->cw_rider;
    c->copy_rider = src->copy_rider;
    c->red_rider = src->red_rider;
    c->step_latent = src->step_latent;
    c->nk_valid = false;
    if (src->l_ns && src->Ckeep && src->Ckeep_beta) {
        // the kept input of the last (sign-iteration) L-step moves along: a point collected from the new ctx before its
        // first L-step there (max_iter right after a compaction) is still rebuilt by ggl_finalize_L
        e = hipMalloc(&c->Ckeep_alloc, c->n * sizeof(double) + STACK_SLACK);
        if (e != hipSuccess) { (void)hipFree(didx); ggl_ctx_destroy(c); return fail(GGL_E_HIP, "subset: %s", hipGetErrorString(e)); }
        c->Ckeep = c->Ckeep_alloc;
        c->Ckeep_beta = (double*)malloc(m * sizeof(double));
        for (int i = 0; i < m; ++i) c->Ckeep_beta[i] = src->Ckeep_beta[idx[i]];
        launch_copy_instances(src->stream, c->Ckeep, src->Ckeep, didx, m, pp, false);
        c->l_ns = true;
    }
    if (src->has_mask) {
        e = hipMemcpyAsync(c->mask, src->mask, pp * sizeof(double), hipMemcpyDeviceToDevice, src->stream);
        c->has_mask = true;
    }
    if (e == hipSuccess && src->has_maskK && src->maskK) {
        e = hipMalloc(&c->maskK, c->n * sizeof(double));
        if (e == hipSuccess) launch_copy_instances(src->stream, c->maskK, src->maskK, didx, m, pp, false);
        c->has_maskK = true;
    }
    if (e == hipSuccess && src->has_dims && src->inst_pk) {
        std::vector<int> all(src->K), sub(m);
        e = hipMemcpyAsync(all.data(), src->inst_pk, src->K * sizeof(int), hipMemcpyDeviceToHost, src->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(src->stream);
        for (int i = 0; i < m; ++i) sub[i] = all[idx[i]];
        if (e == hipSuccess) e = hipMalloc(&c->inst_pk, m * sizeof(int));
        if (e == hipSuccess) e = hipMemcpy(c->inst_pk, sub.data(), m * sizeof(int), hipMemcpyHostToDevice);
        c->has_dims = true;
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(src->stream);
    (void)hipFree(didx);
    if (e != hipSuccess) { ggl_ctx_destroy(c); return fail(GGL_E_HIP, "subset: %s", hipGetErrorString(e)); }
    *out = c;
    return GGL_OK;
}

extern "C" int ggl_scale_X(ggl_ctx* c, double factor)
{
    ARGCHK(c, "ctx");
    HIPCHK(hipSetDevice(c->device));
    DROP_PRE(c);
    launch_scale(c->stream, c->X, factor, c->n);
    HIPCHK(hipGetLastError());
    return GGL_OK;
}

extern "C" int ggl_profile_enable(ggl_ctx* c, int on)
{
    ARGCHK(c, "ctx");
    HIPCHK(hipSetDevice(c->device));
    if (on && !c->ev[0][0]) {
        for (int ph = 0; ph < GGL_NPHASE; ++ph)
            for (int e = 0; e < 2; ++e) HIPCHK(hipEventCreate(&c->ev[ph][e]));
        for (int q = 0; q < 2; ++q)
            for (int e = 0; e < 2; ++e) HIPCHK(hipEventCreate(&c->ev_early[q][e]));
    }
    c->prof_on = (on == 2) ? 2 : (on != 0 ? 1 : 0);
    return GGL_OK;
}

extern "C" int ggl_ns_stats(ggl_ctx* c, long long out[16])
{
    ARGCHK(c && out, "ctx, out");
    long long lds[4] = {0, 0, 0, 0};
    if (c->lds_calls) { int rc_ = ggl_lds_stats(c, lds); if (rc_) return rc_; }
    out[0] = c->ns_calls;
    out[1] = c->ns_steps_total + (lds[3] + c->K / 2) / c->K;        // (the LDS kernel's instances run their own schedules: batch averages)
    out[2] = c->ns_stable_calls;
    out[3] = c->ns_units_total + (lds[2] + c->K / 2) / c->K;
    out[4] = c->ns_launches_total;
    out[5] = c->rank_calls;
    out[6] = c->rank_retries;
    out[7] = c->rank_fallbacks;
    out[8] = c->rank_launches;
    out[9] = c->spec_calls;
    out[10] = c->spec_misses;
    out[11] = c->spin_timeouts;
    out[12] = c->last_parts;
    out[13] = c->last_variant;
    out[14] = c->ns_eigh_fallbacks;
    out[15] = c->pre_dropped;
    return GGL_OK;
}

// The LDS-resident Omega-step: { launches, launches an instance fell outside the kernel's range (step repeated on the launch
// chain), products summed over all instances of all launches, Newton-Schulz steps likewise }.  Waits for the stream.
// GGL_OPT_GROUP_SCHED: out = { Omega-steps that ran as groups with their own schedules, groups of the last step (1: whole),
// lengths of its groups [4], product units (A', B' included) of their schedules [4], grouped steps whose split differed from
// the grouped step before };
// units_sum (may be null) [4]: the units of every group slot summed over the grouped steps.
extern "C" int ggl_group_stats(ggl_ctx* c, long long out[11], double* units_sum)
{
    ARGCHK(c && out, "ctx, out");
    out[10] = c->group_changes;
    out[0] = c->group_steps;
    out[1] = c->last_groups;
    for (int g = 0; g < 4; ++g) {
        out[2 + g] = c->last_groups > 1 ? c->last_group_len[g] : 0;
        out[6 + g] = c->last_groups > 1 ? c->last_group_units[g] : 0;
        if (units_sum) units_sum[g] = c->group_units_sum[g];
    }
    return GGL_OK;
}

// The spectral bounds c_k >= lambda_max(W_k^2 + 4 beta_k I) and the beta_k of the last VALIDATED matrix-function Omega-step
// (what the next step's speculative schedule is built from); returns 0 when there are none yet, 1 otherwise.
extern "C" int ggl_spectral_bounds(ggl_ctx* c, double* c_out, double* beta_out)
{
    ARGCHK(c && c_out && beta_out, "ctx, c_out, beta_out");
    if (!c->omega_ns || !c->spec_have) return 0;
    for (int k = 0; k < c->K; ++k) { c_out[k] = c->spec_c[k]; beta_out[k] = c->spec_beta[k]; }
    return 1;
}

extern "C" int ggl_lds_stats(ggl_ctx* c, long long out[4])
{
    ARGCHK(c && out, "ctx, out");
    out[0] = c->lds_calls;
    out[1] = c->lds_misses;
    out[2] = out[3] = 0;
    if (c->lds_tab) {
        HIPCHK(hipSetDevice(c->device));
        unsigned long long cnt[2] = {0, 0};
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipMemcpy(cnt, c->lds_tab + (size_t)OMEGA_LDS_MAXTAB * OMEGA_LDS_ENT, sizeof(cnt), hipMemcpyDeviceToHost));
        out[2] = (long long)cnt[0];
        out[3] = (long long)cnt[1];
    }
    return GGL_OK;
}

// Pipelining across iterations (GGL_OPT_PIPELINE): { whole chains launched ahead of the caller's next step, of those forgotten
// (rho changed), early first parts put into the stream before the wait for the residuals, of those continued, fresh streams the
// concurrency probe of the part streams had to try (0: the part stream ran beside the main stream; -1: not probed yet) }
extern "C" int ggl_pipeline_stats(ggl_ctx* c, long long out[10])
{
    ARGCHK(c && out, "ctx, out");
    out[0] = c->pre_launched;
    out[1] = c->pre_dropped;
    out[2] = c->early_launched;
    out[3] = c->early_used;
    out[4] = c->parts_probed ? c->parts_replaced : -1;
    out[5] = c->wf_written;
    out[6] = c->wf_used;
    out[7] = c->cw_rides;
    out[8] = c->copy_rides;
    out[9] = c->red_rides;
    return GGL_OK;
}

// Per-instance status word of the last eigensolver launch as the last ADMM step fetched it: the LDS Jacobi kernel reports the
// sweeps it took (-1: not converged within its limit), rocSOLVER its info (0 = converged).
// Event timeline without a profiler.  ggl_trace_start: from now on every launch of ggl_admm_step's iteration is followed by an
// event on its stream (up to max_events; product launches through the hook of gemm_sym.hip) and the host notes when it
// entered the step, queued the Theta-step, queued the early part, saw the residuals and returned.  ggl_trace_read stops the
// recording, waits for the device and returns rows {kind 0 device / 1 host, lane (0 main stream, 1.. part streams), tag,
// microseconds since the start}: device rows give the COMPLETION time of the launch they follow.  Returns the rows written.
extern "C" int ggl_trace_start(ggl_ctx* c, int max_events)
{
    ARGCHK(c && max_events >= 16 && max_events <= (1 << 16), "ctx, 16 <= max_events <= 65536");
    HIPCHK(hipSetDevice(c->device));
    DROP_PRE(c);
    ggl_ctx::Trace& t = c->trace;
    for (hipEvent_t e : t.ev) (void)hipEventDestroy(e);
    if (t.base) (void)hipEventDestroy(t.base);
    t.ev.assign(max_events, nullptr);
    for (hipEvent_t& e : t.ev) HIPCHK(hipEventCreate(&e));
    t.tag.assign(max_events, 0); t.lane.assign(max_events, 0);
    t.host_us.assign(max_events, 0.0); t.host_tag.assign(max_events, 0);
    t.cap = max_events; t.n = 0; t.nhost = 0;
    HIPCHK(hipEventCreate(&t.base));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipEventRecord(t.base, c->stream));
    HIPCHK(hipEventSynchronize(t.base));
    t.t0 = std::chrono::steady_clock::now();
    symm_set_launch_hook(trace_symm_hook, c);
    t.on = true;
    return GGL_OK;
}

extern "C" int ggl_trace_read(ggl_ctx* c, double* out /*(cap,4)*/, int cap)
{
    ARGCHK(c && out && cap >= 1, "ctx, out, cap");
    ggl_ctx::Trace& t = c->trace;
    t.on = false;
    symm_set_launch_hook(nullptr, nullptr);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    int n = 0;
    for (int i = 0; i < t.n && n < cap; ++i, ++n) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.base, t.ev[i]) != hipSuccess) ms = -1.f;
        double* o = out + 4 * (size_t)n;
        o[0] = 0.0; o[1] = t.lane[i]; o[2] = t.tag[i]; o[3] = 1e3 * (double)ms;
    }
    for (int i = 0; i < t.nhost && n < cap; ++i, ++n) {
        double* o = out + 4 * (size_t)n;
        o[0] = 1.0; o[1] = -1.0; o[2] = t.host_tag[i]; o[3] = t.host_us[i];
    }
    for (hipEvent_t e : t.ev) (void)hipEventDestroy(e);
    t.ev.clear();
    if (t.base) { (void)hipEventDestroy(t.base); t.base = nullptr; }
    t.n = t.nhost = 0;
    return n;
}

extern "C" int ggl_eig_info(ggl_ctx* c, int* out)
{
    ARGCHK(c && out, "ctx, out");
    for (int k = 0; k < c->K; ++k) out[k] = c->info_h[k];
    return GGL_OK;
}

// What ran last: { concurrent parts, product-kernel variant of the last matrix-function step, code of the Theta kernel the
// process's last Theta-step launched (theta_pair.hip: theta_last_kernel), eigendecompositions ggl_finalize_L ran on this ctx }
extern "C" int ggl_last_dispatch(ggl_ctx* c, long long out[4])
{
    ARGCHK(c && out, "ctx, out");
    out[0] = c->last_parts;
    out[1] = c->last_variant;
    out[2] = theta_last_kernel();
    out[3] = c->finalize_calls;
    return GGL_OK;
}

// L-step calls whose first pass was followed by the deflation, and the instances that had something to deflate
extern "C" int ggl_deflate_stats(ggl_ctx* c, long long out[2])
{
    ARGCHK(c && out, "ctx, out");
    out[0] = c->rank_deflated_calls;
    out[1] = c->rank_deflated_instances;
    return GGL_OK;
}

extern "C" int ggl_rank_stats(ggl_ctx* c, long long out[4])
{
    ARGCHK(c && out, "ctx, out");
    out[0] = c->rank_calls;
    out[1] = c->rank_continued;
    out[2] = c->rank_cont_instances;
    out[3] = c->rank_fallbacks;
    return GGL_OK;
}

extern "C" int ggl_profile_read(ggl_ctx* c, double ms[GGL_NPHASE], long long count[GGL_NPHASE], int reset)
{
    ARGCHK(c && ms && count, "ctx, ms, count");
    for (int ph = 0; ph < GGL_NPHASE; ++ph) {
        ms[ph] = c->ph_ms[ph];
        count[ph] = c->ph_cnt[ph];
        if (reset) { c->ph_ms[ph] = 0.0; c->ph_cnt[ph] = 0; }
    }
    return GGL_OK;
}

// ---------------------------------------------------------------------------------------------
// exit checks / objective / kkt
// ---------------------------------------------------------------------------------------------
static int host_reduce(ggl_ctx* c, int rows, int nv, double* out /*nv*/, bool take_max)
{
    HIPCHK(hipMemcpyAsync(c->norms_h, c->norms, (size_t)rows * nv * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int v = 0; v < nv; ++v) {
        double s = take_max ? -INFINITY : 0.0;
        for (int r = 0; r < rows; ++r) {
            const double x = c->norms_h[(size_t)r * nv + v];
            s = take_max ? std::max(s, x) : s + x;
        }
        out[v] = s;
    }
    return GGL_OK;
}

// per-instance smallest eigenvalue of the stack A (destroyed)
static int min_eig_k(ggl_ctx* c, double* A, double* outK)
{
    int rc = eigvals_only(c, A, c->DvL);
    if (rc) return rc;
    std::vector<double> d((size_t)c->K * c->p);
    HIPCHK(hipMemcpyAsync(d.data(), c->DvL, d.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(c->info_h, c->info, c->K * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    rc = check_info(c, "exit check");
    if (rc) return rc;
    for (int k = 0; k < c->K; ++k) outK[k] = *std::min_element(d.begin() + (size_t)k * c->p, d.begin() + (size_t)(k + 1) * c->p);
    return GGL_OK;
}

extern "C" int ggl_exit_checks_k(ggl_ctx* c, int latent, double* out /*(K,5)*/)
{
    ARGCHK(c && out, "ctx, out");
    HIPCHK(hipSetDevice(c->device));
    DROP_PRE(c);
    const int K = c->K;
    const double* stacks[3] = {c->Om[c->cur], c->Theta, c->L};
    for (int i = 0; i < 3; ++i) {
        launch_asym_max(c->stream, stacks[i], K, c->p, c->norms);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(c->norms_h, c->norms, K * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        for (int k = 0; k < K; ++k) out[(size_t)k * 5 + i] = c->norms_h[k];
    }
    std::vector<double> mn(K);
    launch_sub(c->stream, c->W, c->Theta, c->L, c->n);   // admm_solver.py:294
    int rc = min_eig_k(c, c->W, mn.data());
    if (rc) return rc;
    for (int k = 0; k < K; ++k) { out[(size_t)k * 5 + 3] = mn[k]; out[(size_t)k * 5 + 4] = 0.0; }
    if (latent) {
        HIPCHK(hipMemcpyAsync(c->W, c->L, c->n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        rc = min_eig_k(c, c->W, mn.data());               // admm_solver.py:299
        if (rc) return rc;
        for (int k = 0; k < K; ++k) out[(size_t)k * 5 + 4] = mn[k];
    }
    return GGL_OK;
}

// Batched Cholesky factorisation of the stack A (destroyed) as a definiteness TEST: okK[k] = 1 iff A_k is (numerically)
// positive definite.  rocSOLVER's potrf; a failed pivot is what info reports.
static int chol_pd_k(ggl_ctx* c, double* A, unsigned char* okK)
{
    int rc = blas_handle(c, &c->blas);
    if (rc) return rc;
    c->info_dirty = true;
    rocblas_status st = rocsolver_dpotrf_strided_batched(c->blas, rocblas_fill_upper, c->p, A, c->p, (rocblas_stride)c->p * c->p,
                                                         c->info, c->K);
    if (st != rocblas_status_success) return fail(GGL_E_SOLVER, "rocsolver_dpotrf_strided_batched: status %d", (int)st);
    HIPCHK(hipMemcpyAsync(c->info_h, c->info, c->K * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int k = 0; k < c->K; ++k) okK[k] = (c->info_h[k] == 0) ? 1 : 0;
    // (info is an eigensolver status word elsewhere: leave it clean)
    HIPCHK(hipMemsetAsync(c->info, 0, c->K * sizeof(int), c->stream));
    memset(c->info_h, 0, c->K * sizeof(int));
    c->info_dirty = false;
    return GGL_OK;
}

// The exit checks of a solve (admm_solver.py:284-301, single_admm_solver.py:244-263, ext_admm_solver.py:290-311) as the
// DECISIONS the reference takes, without the eigenvalues: out[k*5..] = { max asymmetry of Omega, Theta, L as ggl_exit_checks_k,
// 1 if Theta_k - L_k - shift_tl I is positive definite else 0, 1 if L_k + shift_l I is positive definite else 0 (1 when not
// latent) } -- two batched Cholesky factorisations instead of two eigendecompositions (measured: 20 ms of eigenvalues behind
// a 25 ms solve at (32,500), 26 ms behind a 5 ms solve at (64,100): tools/time_exit_checks.py).  The reference warns when
// min eig(Theta - L) <= shift_tl resp. min eig(L) < -shift_l: exactly the instances whose flag is 0; the caller fetches the
// eigenvalues (ggl_exit_checks_k) only for the message of a warning it has to print.
extern "C" int ggl_exit_checks_fast_k(ggl_ctx* c, int latent, double shift_tl, double shift_l, double* out /*(K,5)*/)
{
    ARGCHK(c && out, "ctx, out");
    HIPCHK(hipSetDevice(c->device));
    DROP_PRE(c);
    const int K = c->K;
    const double* stacks[3] = {c->Om[c->cur], c->Theta, c->L};
    for (int i = 0; i < 3; ++i) {
        launch_asym_max(c->stream, stacks[i], K, c->p, c->norms);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(c->norms_h, c->norms, K * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        for (int k = 0; k < K; ++k) out[(size_t)k * 5 + i] = c->norms_h[k];
    }
    std::vector<unsigned char> ok(K);
    launch_sub(c->stream, c->W, c->Theta, c->L, c->n);
    if (shift_tl != 0.0) launch_add_diag(c->stream, c->W, K, c->p, -shift_tl);
    HIPCHK(hipGetLastError());
    int rc = chol_pd_k(c, c->W, ok.data());
    if (rc) return rc;
    for (int k = 0; k < K; ++k) { out[(size_t)k * 5 + 3] = ok[k]; out[(size_t)k * 5 + 4] = 1.0; }
    if (latent) {
        HIPCHK(hipMemcpyAsync(c->W, c->L, c->n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        launch_add_diag(c->stream, c->W, K, c->p, shift_l);
        HIPCHK(hipGetLastError());
        rc = chol_pd_k(c, c->W, ok.data());
        if (rc) return rc;
        for (int k = 0; k < K; ++k) out[(size_t)k * 5 + 4] = ok[k];
    }
    return GGL_OK;
}

extern "C" int ggl_exit_checks(ggl_ctx* c, int latent, double out[5])
{
    ARGCHK(c && out, "ctx, out");
    std::vector<double> per((size_t)c->K * 5);
    int rc = ggl_exit_checks_k(c, latent, per.data());
    if (rc) return rc;
    for (int v = 0; v < 5; ++v) {
        double s = per[v];
        for (int k = 1; k < c->K; ++k) s = (v < 3) ? std::max(s, per[(size_t)k * 5 + v]) : std::min(s, per[(size_t)k * 5 + v]);
        out[v] = s;
    }
    return GGL_OK;
}

// Snapshots of the instances ks[0..n) of `src` into the slots kd[0..n) of `c` (c == src, kd == ks: ggl_snapshot_k).  The two-ctx
// form serves a batch that was compacted (ggl_ctx_create_subset): a point that converges in the smaller ctx is snapshotted into
// the ORIGINAL ctx at its original index, where the selection statistics and ggl_finalize_L run over all points at once.
// with_state: Omega and X as well (ggl_snapshot_state_from).  ONE wait for src and one for c whatever n (the C loop hands over
// all points that finish in an iteration together; round 5 synchronised both streams per instance -- ADVICE r5).
static int snapshot_many(ggl_ctx* c, const int* kd, ggl_ctx* src, const int* ks, int n, bool with_state)
{
    ARGCHK(c && src && kd && ks && n >= 1, "ctx, indices");
    for (int i = 0; i < n; ++i) ARGCHK(kd[i] >= 0 && kd[i] < c->K && ks[i] >= 0 && ks[i] < src->K, "instance index");
    ARGCHK(c->p == src->p && c->device == src->device, "snapshot between ctxs of different dimension / device");
    HIPCHK(hipSetDevice(c->device));
    // (before anything is queued: a pre-launched chain of either ctx holds Omega[cur ^ 1] and scratch; what is snapshotted is
    // the iterate the caller can observe -- ADVICE r5: the state copies used to be queued ahead of the drop)
    DROP_PRE(c);
    if (src != c) {
        int rc_ = drop_prelaunch(src);
        if (rc_) return rc_;
        HIPCHK(hipStreamSynchronize(src->stream));          // the copies below run on c's stream
    }
    const size_t pp = (size_t)c->p * c->p;
    // (fills and copies of the snapshots are ordinary kernels on c's stream, launch_copy_block: their order is the queue's)
    auto lazy = [&](double** b) -> int {
        if (!*b) {
            HIPCHK(hipMalloc(b, c->n * sizeof(double)));
            launch_copy_block(c->stream, *b, nullptr, c->n);                      // (slots never snapshotted read as zeros)
        }
        return GGL_OK;
    };
    int rc = lazy(&c->snapT);
    if (rc) return rc;
    if (with_state) {
        if ((rc = lazy(&c->snapOm)) != GGL_OK || (rc = lazy(&c->snapX)) != GGL_OK) return rc;
    }
    if (src->step_latent) {
        if ((rc = lazy(&c->snapL)) != GGL_OK) return rc;
        if (!c->snap_ns) {
            c->snap_ns = (unsigned char*)calloc(c->K, 1);
            c->snap_beta = (double*)calloc(c->K, sizeof(double));
        }
        if (src->l_ns && (rc = lazy(&c->snapC)) != GGL_OK) return rc;
    }
    for (int i = 0; i < n; ++i) {
        const size_t od = (size_t)kd[i] * pp, os = (size_t)ks[i] * pp;
        if (with_state) {
            launch_copy_block(c->stream, c->snapOm + od, src->Om[src->cur] + os, pp);
            launch_copy_block(c->stream, c->snapX + od, src->X + os, pp);
        }
        launch_copy_block(c->stream, c->snapT + od, src->Theta + os, pp);
        if (src->step_latent) {
            launch_copy_block(c->stream, c->snapL + od, src->L + os, pp);
            c->snap_ns[kd[i]] = src->l_ns ? 1 : 0;
            if (src->l_ns) {
                // the sign iteration's L: keep its input C as well, ggl_finalize_L(which = 1) rebuilds the snapshot from it
                launch_copy_block(c->stream, c->snapC + od, src->Ckeep + os, pp);
                c->snap_beta[kd[i]] = src->Ckeep_beta[ks[i]];
            }
        }
    }
    HIPCHK(hipGetLastError());
    if (src != c) HIPCHK(hipStreamSynchronize(c->stream));  // src may go on (or away) right after the call
    return GGL_OK;
}

extern "C" int ggl_snapshot_from(ggl_ctx* c, int kd, ggl_ctx* src, int ks)
{
    ARGCHK(c && src, "ctx");
    return snapshot_many(c, &kd, src, &ks, 1, false);
}

extern "C" int ggl_snapshot_k(ggl_ctx* c, int k) { return ggl_snapshot_from(c, k, c, k); }

// ggl_snapshot_from plus Omega and X of the instance: the WHOLE solution of a point of a batch stays on the device at the
// iteration it finished, and the batch driver fetches all points' solutions at the end with ONE download per stack
// (ggl_get_snapshots) instead of three or four small ones per point (~40 us each: 4 ms of a 9 ms 100-point grid).
extern "C" int ggl_snapshot_state_from(ggl_ctx* c, int kd, ggl_ctx* src, int ks)
{
    ARGCHK(c && src, "ctx");
    return snapshot_many(c, &kd, src, &ks, 1, true);
}

/* whole snapshot stacks (K,p,p), any may be null: Omega, Theta, L, X as ggl_snapshot_state_from left them */
extern "C" int ggl_get_snapshots(ggl_ctx* c, double* Omega, double* Theta, double* L, double* X)
{
    ARGCHK(c, "ctx");
    ARGCHK(!Theta || c->snapT, "no snapshot taken");
    ARGCHK((!Omega && !X) || (c->snapOm && c->snapX), "no state snapshot taken (ggl_snapshot_state_from)");
    HIPCHK(hipSetDevice(c->device));
    const size_t nb = c->n * sizeof(double);
    std::vector<Xfer> xs;
    if (Omega) xs.push_back({Omega, c->snapOm, nb});
    if (Theta) xs.push_back({Theta, c->snapT, nb});
    if (L && c->snapL) xs.push_back({L, c->snapL, nb});
    else if (L) memset(L, 0, nb);          // no latent step ever ran: L is what the solvers return then, zeros (admm_solver.py:150)
    if (X) xs.push_back({X, c->snapX, nb});
    return download_stacks(c, xs);
}

// The latent component a solve returns (solver/ggl_helper.py:29-36: L = Q diag(max(d - beta, 0)) Q^T, whose null space is
// exact to rounding -- the reference's callers apply numpy.linalg.matrix_rank to it, helper/model_selection.py:254, :638).
// Where the L-step ran as the sign iteration, L is rebuilt here from ONE eigendecomposition of that step's input C (kept by
// rank_step / ggl_snapshot_k): the reference's own L-step, executed once per solve instead of once per iteration.  The dual X
// keeps the sign iteration's L in its last update (a difference of ~1e-13 |L|).
//   which 0: the live iterate's L;  1: the snapshots' L (ggl_snapshot_k).
//   rank_out (K ints, may be null): #{ eigenvalues of C_k above mu1_k / rho } for the instances rebuilt, -1 for the others.
// Returns the number of instances rebuilt (0: every L already came from an eigendecomposition, nothing done), < 0 on error.
extern "C" int ggl_finalize_L(ggl_ctx* c, int which, int* rank_out)
{
    ARGCHK(c, "ctx");
    ARGCHK(which == 0 || which == 1, "which");
    HIPCHK(hipSetDevice(c->device));
    DROP_PRE(c);
    const int K = c->K, p = c->p;
    const size_t pp = (size_t)p * p, kp = (size_t)K * p;
    if (rank_out) for (int k = 0; k < K; ++k) rank_out[k] = -1;
    std::vector<unsigned char> todo(K, 0);
    int n_todo = 0;
    const double* beta_src = nullptr;
    if (which == 0) {
        if (c->l_ns) { std::fill(todo.begin(), todo.end(), 1); n_todo = K; beta_src = c->Ckeep_beta; }
    } else if (c->snap_ns && c->snapC) {
        for (int k = 0; k < K; ++k) if (c->snap_ns[k]) { todo[k] = 1; n_todo += 1; }
        beta_src = c->snap_beta;
    }
    if (!n_todo) return 0;
    // parameter slot 2 (mu1_k / rho) is saved and restored: the solve may go on after a snapshot was finalised
    double* slot = c->par_h + 2 * (size_t)K;
    std::vector<double> saved(slot, slot + K);
    for (int k = 0; k < K; ++k) slot[k] = todo[k] ? beta_src[k] : 0.0;
    HIPCHK(hipMemcpyAsync(c->par + 2 * (size_t)K, slot, K * sizeof(double), hipMemcpyHostToDevice, c->stream));
    double* Csrc = which == 0 ? c->Ckeep : c->snapC;
    double* out = which == 0 ? c->L : c->W;
    int rc = eig_recon(c, Csrc, out, c->DvL, MAP_RANK, c->par + 2 * (size_t)K);      // (destroys Csrc)
    if (rc) return rc;
    std::vector<double> d(kp);
    HIPCHK(hipMemcpyAsync(d.data(), c->DvL, kp * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(c->info_h, c->info, K * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (which == 1)
        for (int k = 0; k < K; ++k)
            if (todo[k]) launch_copy_block(c->stream, c->snapL + k * pp, c->W + k * pp, pp);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(slot, saved.data(), K * sizeof(double));
    HIPCHK(hipMemcpyAsync(c->par + 2 * (size_t)K, slot, K * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    rc = check_info(c, "final L");
    if (rc) return rc;
    for (int k = 0; k < K; ++k) {
        if (!todo[k]) continue;
        int r = 0;
        bool finite = true;
        for (int e = 0; e < p; ++e) {
            const double v = d[(size_t)k * p + e];
            finite = finite && std::isfinite(v);
            r += v > beta_src[k] ? 1 : 0;
        }
        // eigenvalues that are not finite: the kept C was not (a diverged instance) -- never a rank of zero that reads like a result
        if (!finite) {
            if (c->isolate) { mark_failed(c, k, 2, NAN); r = -1; }
            else return fail(GGL_E_SOLVER, "final L: the eigenvalues of instance %d's L-step input are not finite", k);
        }
        if (rank_out) rank_out[k] = r;
    }
    if (which == 0) c->l_ns = false;                  // L is an eigendecomposition's now (and Ckeep is spent)
    else memset(c->snap_ns, 0, K);                    // (snapC is spent; a later snapshot of an instance sets its flag again)
    if (which == 1) launch_copy_block(c->stream, c->snapC, nullptr, c->n);
    c->finalize_calls += 1;
    return n_todo;
}

/* Theta and L of instance k's snapshot (ggl_snapshot_k), either may be null */
extern "C" int ggl_get_snapshot_k(ggl_ctx* c, int k, double* Theta, double* L)
{
    ARGCHK(c && k >= 0 && k < c->K, "ctx, k");
    ARGCHK(c->snapT, "no snapshot taken (ggl_snapshot_k)");
    ARGCHK(!L || c->snapL, "no snapshot of L (ggl_snapshot_k after a latent step)");
    HIPCHK(hipSetDevice(c->device));
    const size_t pp = (size_t)c->p * c->p, nb = pp * sizeof(double), off = (size_t)k * pp;
    if (Theta) HIPCHK(hipMemcpyAsync(Theta, c->snapT + off, nb, hipMemcpyDeviceToHost, c->stream));
    if (L) HIPCHK(hipMemcpyAsync(L, c->snapL + off, nb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return GGL_OK;
}

/* Omega and X of instance k's snapshot (ggl_snapshot_state_from), either may be null */
extern "C" int ggl_get_snapshot_state_k(ggl_ctx* c, int k, double* Omega, double* X)
{
    ARGCHK(c && k >= 0 && k < c->K, "ctx, k");
    ARGCHK(c->snapOm && c->snapX, "no state snapshot taken (ggl_snapshot_state_from)");
    HIPCHK(hipSetDevice(c->device));
    const size_t pp = (size_t)c->p * c->p, nb = pp * sizeof(double), off = (size_t)k * pp;
    if (Omega) HIPCHK(hipMemcpyAsync(Omega, c->snapOm + off, nb, hipMemcpyDeviceToHost, c->stream));
    if (X) HIPCHK(hipMemcpyAsync(X, c->snapX + off, nb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return GGL_OK;
}

extern "C" int ggl_selection_stats(ggl_ctx* c, double* out)
{
    ARGCHK(c && out, "ctx, out");
    ARGCHK(c->snapT, "no snapshot taken (ggl_snapshot_k)");
    HIPCHK(hipSetDevice(c->device));
    DROP_PRE(c);
    const int K = c->K, p = c->p;
    const size_t kp = (size_t)K * p;
    const int nblk = elementwise_blocks(p);
    std::vector<double> d(kp), dot(K), nnz(K);
    // <S_k, Theta_k> and the non-zero count, per instance
    launch_dot(c->stream, c->snapT, c->S, K, p, c->partials);
    launch_reduce_partials(c->stream, c->partials, K, nblk, 1, c->norms);
    HIPCHK(hipMemcpyAsync(dot.data(), c->norms, K * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    launch_count_nonzero(c->stream, c->snapT, K, p, c->partials);
    launch_reduce_partials(c->stream, c->partials, K, nblk, 1, c->norms + K);
    HIPCHK(hipMemcpyAsync(nnz.data(), c->norms + K, K * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipGetLastError());
    // eigenvalues of Theta_k: log det and the smallest one (robust_logdet, model_selection.py:884-894)
    HIPCHK(hipMemcpyAsync(c->W, c->snapT, c->n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    int rc = eigvals_only(c, c->W, c->DvO);
    if (rc) return rc;
    c->dvo_valid = false;
    HIPCHK(hipMemcpyAsync(d.data(), c->DvO, kp * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(c->info_h, c->info, K * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    rc = check_info(c, "selection statistics");
    if (rc) return rc;
    for (int k = 0; k < K; ++k) {
        double mn = INFINITY, ld = 0.0;
        for (int m = 0; m < p; ++m) {
            const double v = d[(size_t)k * p + m];
            mn = std::min(mn, v);
            ld += std::log(v);
        }
        out[k * 4 + 0] = dot[k];
        out[k * 4 + 1] = (mn <= 1e-12 || !(mn == mn)) ? -INFINITY : ld;
        out[k * 4 + 2] = nnz[k];
        out[k * 4 + 3] = mn;
    }
    return GGL_OK;
}

extern "C" int ggl_objective(ggl_ctx* c, double lambda1, double lambda2, int reg, double out[3])
{
    ARGCHK(c && out, "ctx, out");
    ARGCHK(reg == GGL_REG_GGL || reg == GGL_REG_FGL, "reg");
    HIPCHK(hipSetDevice(c->device));
    DROP_PRE(c);
    // -log det Omega_k = -sum_m log phip(d_m): eigenvalues of the last Omega-step (ggl_helper.py:266-270);
    // the Newton-Schulz Omega-step has none, so there the eigenvalues of Omega itself are computed.
    const size_t kp = (size_t)c->K * c->p;
    std::vector<double> d(kp);
    const bool from_w = c->dvo_valid;
    bool from_chol = false;
    if (!from_w) {
        // log det Omega_k = 2 sum_i log R_ii of the Cholesky factor (Omega = phiplus(...) is positive definite by construction):
        // one batched potrf instead of the eigenvalues -- measure=True evaluates this EVERY iteration, and the eigenvalues cost
        // 20 ms against a 0.8 ms iteration at (32,500) (tools/time_ctx.py).  A failed factorisation (a non-finite iterate)
        // falls back to the eigenvalues, whose logarithms then say what went wrong.
        HIPCHK(hipMemcpyAsync(c->W, c->Om[c->cur], c->n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        std::vector<unsigned char> ok(c->K);
        int rc0 = chol_pd_k(c, c->W, ok.data());
        if (rc0) return rc0;
        from_chol = true;
        for (int k = 0; k < c->K; ++k) from_chol = from_chol && ok[k];
        if (from_chol) {
            launch_get_diag(c->stream, c->W, c->K, c->p, c->DvO);
        } else {
            HIPCHK(hipMemcpyAsync(c->W, c->Om[c->cur], c->n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
            rc0 = eigvals_only(c, c->W, c->DvO);
            if (rc0) return rc0;
        }
    }
    HIPCHK(hipMemcpyAsync(d.data(), c->DvO, kp * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    launch_dot(c->stream, c->Om[c->cur], c->S, c->K, c->p, c->partials);
    launch_reduce_partials(c->stream, c->partials, c->K, elementwise_blocks(c->p), 1, c->norms);
    HIPCHK(hipGetLastError());
    int rc = host_reduce(c, c->K, 1, &out[1], false);
    if (rc) return rc;
    double ld = 0.0;
    for (int k = 0; k < c->K; ++k) {
        const double beta = c->par_h[k];
        for (int m = 0; m < c->p; ++m) {
            const double dv = d[(size_t)k * c->p + m];
            ld -= from_w ? std::log(0.5 * (std::sqrt(dv * dv + 4.0 * beta) + dv)) : (from_chol ? 2.0 * std::log(dv) : std::log(dv));
        }
    }
    out[0] = ld;
    const int nb = pval_blocks(c->p);
    launch_pval(c->stream, reg, c->Theta, lambda1, lambda2, c->K, c->p, c->partials);
    launch_reduce_partials(c->stream, c->partials, 1, nb, 1, c->norms);
    HIPCHK(hipGetLastError());
    return host_reduce(c, 1, 1, &out[2], false);
}

static int stack_sq(ggl_ctx* c, const double* A, const double* B, double* out)
{
    launch_sqdiff(c->stream, A, B, c->K, c->p, c->partials);
    launch_reduce_partials(c->stream, c->partials, c->K, elementwise_blocks(c->p), 1, c->norms);
    HIPCHK(hipGetLastError());
    return host_reduce(c, c->K, 1, out, false);
}

extern "C" int ggl_kkt_residual(ggl_ctx* c, double rho, double lambda1, double lambda2, int reg, int latent,
                                const double* mu1, const double* nk, double* out)
{
    ARGCHK(c && out, "ctx, out");
    ARGCHK(!latent || mu1, "latent needs mu1");
    HIPCHK(hipSetDevice(c->device));
    DROP_PRE(c);
    double* Om = c->Om[c->cur];
    double* T1 = c->W;             // scratch
    double* T2 = c->Om[c->cur ^ 1]; // Omega_{t-1} is dead once the step's norms are out
    double nTheta, nOmega, v;
    int rc;
    if ((rc = stack_sq(c, c->Theta, nullptr, &nTheta))) return rc;
    if ((rc = stack_sq(c, Om, nullptr, &nOmega))) return rc;
    nTheta = std::sqrt(nTheta);
    nOmega = std::sqrt(nOmega);
    // term1: |Theta - prox(Theta + rho X)| / (1 + |Theta|)
    launch_axpy(c->stream, T1, c->Theta, rho, c->X, c->n);
    if (reg == GGL_REG_SGL) {
        // prox_od_1norm with l = lambda1 (scalar or the (p,p) mask array), per instance
        for (int k = 0; k < c->K; ++k)
            launch_prox_od(c->stream, T2 + (size_t)k * c->p * c->p, T1 + (size_t)k * c->p * c->p, lambda1,
                           c->has_mask ? c->mask : nullptr, c->p);
    } else {
        HIPCHK(launch_prox_p(c->stream, reg, T2, T1, lambda1, lambda2, c->K, c->p, c->sqwork));
    }
    if ((rc = stack_sq(c, c->Theta, T2, &v))) return rc;
    double res = std::sqrt(v) / (1.0 + nTheta);
    // term2: |Theta - Omega - L| / (1 + |Theta|)
    launch_sub(c->stream, T1, c->Theta, Om, c->n);
    if ((rc = stack_sq(c, T1, latent ? c->L : nullptr, &v))) return rc;
    res = std::max(res, std::sqrt(v) / (1.0 + nTheta));
    // term3: |Omega - phiplus(eigh(Omega - nk S - rho X), nk)| / (1 + |Omega|)
    if ((rc = upload_par(c, 3, nk, 1.0, 1.0))) return rc;
    const double* nkd = c->par + 3 * (size_t)c->K;
    launch_kkt_w(c->stream, T1, Om, c->S, c->X, nkd, rho, c->K, c->p);
    if ((rc = eig_recon(c, T1, T2, c->DvL, MAP_PHIPLUS, nkd))) return rc;
    if ((rc = stack_sq(c, Om, T2, &v))) return rc;
    res = std::max(res, std::sqrt(v) / (1.0 + nOmega));
    if (latent) {
        double nL;
        if ((rc = stack_sq(c, c->L, nullptr, &nL))) return rc;
        if ((rc = upload_par(c, 2, mu1, 0.0, 1.0))) return rc;
        launch_axpy(c->stream, T1, c->L, -rho, c->X, c->n);
        if ((rc = eig_recon(c, T1, T2, c->DvL, MAP_RANK, c->par + 2 * (size_t)c->K))) return rc;
        if ((rc = stack_sq(c, c->L, T2, &v))) return rc;
        res = std::max(res, std::sqrt(v) / (1.0 + std::sqrt(nL)));
    }
    HIPCHK(hipMemcpyAsync(c->info_h, c->info, c->K * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if ((rc = check_info(c, "kkt residual"))) return rc;
    *out = res;
    return GGL_OK;
}

// ---------------------------------------------------------------------------------------------
// RCCL behind the ABI: the K-sharded GGL iteration as ONE call
// ---------------------------------------------------------------------------------------------
#define NCCLCHK(api, expr)                                                                              \
    do {                                                                                                \
        int r_ = (expr);                                                                                \
        if (r_ != 0) return fail(GGL_E_COMM, "%s failed: %s", #expr, (api)->GetErrorString(r_));      \
    } while (0)

extern "C" int ggl_comm_unique_id(char id_out[128])
{
    ARGCHK(id_out, "id_out");
    const char* err = nullptr;
    const RcclApi* api = rccl_api(&err);
    if (!api) return fail(GGL_E_COMM, "RCCL unavailable: %s", err ? err : "?");
    RcclApi::UniqueId id;
    NCCLCHK(api, api->GetUniqueId(&id));
    memcpy(id_out, id.internal, RcclApi::UNIQUE_ID_BYTES);
    return GGL_OK;
}

extern "C" int ggl_comm_init(ggl_ctx* c, int rank, int nranks, const char id[128])
{
    ARGCHK(c && id, "ctx, id");
    ARGCHK(nranks >= 1 && rank >= 0 && rank < nranks, "0 <= rank < nranks");
    ARGCHK(c->comm == nullptr, "the ctx already has a communicator");
    HIPCHK(hipSetDevice(c->device));
    DROP_PRE(c);
    const char* err = nullptr;
    const RcclApi* api = rccl_api(&err);
    if (!api) return fail(GGL_E_COMM, "RCCL unavailable: %s", err ? err : "?");
    RcclApi::UniqueId uid;
    memcpy(uid.internal, id, RcclApi::UNIQUE_ID_BYTES);
    RcclApi::Comm comm = nullptr;
    NCCLCHK(api, api->CommInitRank(&comm, nranks, uid, rank));
    c->comm = comm;
    c->comm_rank = rank;
    c->comm_nranks = nranks;
    return GGL_OK;
}

extern "C" int ggl_comm_count(ggl_ctx* c, int* nranks_out)
{
    ARGCHK(c && nranks_out, "ctx, nranks_out");
    ARGCHK(c->comm, "ggl_comm_init first");
    const RcclApi* api = rccl_api(nullptr);
    if (!api || !api->CommCount) return fail(GGL_E_COMM, "RCCL unavailable: ncclCommCount");
    NCCLCHK(api, api->CommCount(c->comm, nranks_out));
    return GGL_OK;
}

extern "C" int ggl_comm_destroy(ggl_ctx* c)
{
    ARGCHK(c, "ctx");
    if (!c->comm) return GGL_OK;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    const RcclApi* api = rccl_api(nullptr);
    if (api) NCCLCHK(api, api->CommDestroy(c->comm));
    c->comm = nullptr;
    return GGL_OK;
}

// sum over ranks of GROUPSQ (p*p sums of squares + the speculation flag) / of the five local sums, in place, on the ctx stream
extern "C" int ggl_allreduce_groupsq(ggl_ctx* c)
{
    ARGCHK(c && c->comm, "ctx with a communicator (ggl_comm_init)");
    const RcclApi* api = rccl_api(nullptr);
    // the packed upper triangle + the flag: 8 (p (p + 1) / 2 + 1) bytes on the wire
    NCCLCHK(api, api->AllReduce(c->groupsq, c->groupsq, ggl::tri_len(c->p) + 1, RcclApi::Float64, RcclApi::Sum, c->comm, c->stream));
    return GGL_OK;
}

extern "C" int ggl_allreduce_norms(ggl_ctx* c)
{
    ARGCHK(c && c->comm, "ctx with a communicator (ggl_comm_init)");
    const RcclApi* api = rccl_api(nullptr);
    NCCLCHK(api, api->AllReduce(c->norms, c->norms, GGL_NNORM, RcclApi::Float64, RcclApi::Sum, c->comm, c->stream));
    return GGL_OK;
}

static int sharded_pass(ggl_ctx* c, double rho, double lambda1, double lambda2, int latent, const double* mu1,
                        const double* nk, bool speculate, double out_norms[5])
{
    CopySegs sg;
    int rc = upload_par(c, 0, nk, 1.0, rho, &sg);
    if (rc) return rc;
    if (latent) DROP_PRE(c);       // (a latent step never speculates and never takes over a pre-launched chain)
    if (latent || !(speculate && take_prelaunched(c, 0))) {
        // with MAX_PARTS parts there is no flag slot left for the all-reduced flag
        rc = omega_step(c, latent, &sg, speculate && !latent && c->ns_parts < ggl_ctx::MAX_PARTS);
        if (rc) return rc;
    }
    // this rank's packed sums with its validation flag behind them: one launch (the flag used to be a kernel of its own
    // before the collective and another one after it)
    launch_group_sums_packed(c->stream, c->groupsq, c->sqwork, c->Om[c->cur], latent ? c->L : nullptr, c->X,
                             (1.0 / rho) * lambda1, c->K, c->p, c->spec_pending ? c->spec_flag : nullptr);
    HIPCHK(hipGetLastError());
    PB(c, GGL_PH_ALLREDUCE_GROUPSQ);         // (HIP events on the ctx stream: what the collective costs THIS rank, waiting included)
    rc = ggl_allreduce_groupsq(c);
    PE(c, GGL_PH_ALLREDUCE_GROUPSQ);
    if (rc) return rc;
    // latent: Theta from the reduced sums, then the L-step and the dual update on the local slab (admm_solver.py:197-208:
    // per instance, no exchange), one row of local sums; norms stay on the device
    rc = ggl_step_finish_impl(c, rho, lambda1, lambda2, GGL_REG_GGL, latent, mu1, 1 | 2, out_norms);
    if (rc) return rc;
    PB(c, GGL_PH_ALLREDUCE_NORMS);
    rc = ggl_allreduce_norms(c);
    PE(c, GGL_PH_ALLREDUCE_NORMS);
    if (rc) return rc;
    // (no early first part of the next chain here, as ggl_admm_step queues one: measured behind the two collectives it is
    // neutral to negative -- round 4, with its own form_W pass: K = 4 / 8 / 16 slabs 4182 / 2855 / 1978 it/s with it, 4224 /
    // 3070 / 2009 without; round 5, with the W written by the Theta kernel (GGL_OPT_FUSED_W): 4397 / 2877 / 2003 with,
    // 4352 / 3035 / 1990 without)
    return finish_norms(c, 1, out_norms);
}

extern "C" int ggl_admm_step_sharded_latent(ggl_ctx* c, double rho, double lambda1, double lambda2, int latent,
                                            const double* mu1, const double* nk, double out_norms[5])
{
    ARGCHK(c && out_norms, "ctx, out_norms");
    ARGCHK(c->comm, "ggl_comm_init first");
    ARGCHK(rho > 0 && lambda1 > 0 && lambda2 > 0, "rho, lambda1, lambda2 must be positive");
    ARGCHK(!latent || mu1, "latent needs mu1");
    HIPCHK(hipSetDevice(c->device));
    int rc = sharded_pass(c, rho, lambda1, lambda2, latent, mu1, nk, true, out_norms);
    if (rc == GGL_SPEC_RETRY) {
        // the reduced validation flag says some rank's schedule did not cover its spectrum: every rank left its iterate
        // alone and repeats the iteration bounds-first (all ranks take this branch together: the flag is the all-reduced one)
        rc = sharded_pass(c, rho, lambda1, lambda2, latent, mu1, nk, false, out_norms);
        if (rc == GGL_SPEC_RETRY) return fail(GGL_E_SOLVER, "K-sharded step: the non-speculative repeat was rejected");
    }
    if (rc != GGL_OK || latent) return rc;
    // the sums are the GLOBAL ones: every rank takes the same decision here (and the chain is local work anyway)
    return (c->ns_parts < ggl_ctx::MAX_PARTS) ? maybe_prelaunch(c, rho, out_norms) : GGL_OK;
}

extern "C" int ggl_admm_step_sharded(ggl_ctx* c, double rho, double lambda1, double lambda2, const double* nk,
                                     double out_norms[5])
{
    return ggl_admm_step_sharded_latent(c, rho, lambda1, lambda2, 0, nullptr, nk, out_norms);
}

// ---------------------------------------------------------------------------------------------
// ext_ADMM_MGL: instances of different dimension (solver/ext_admm_solver.py:18-323), csrc/ext_group.hip
// ---------------------------------------------------------------------------------------------
static int ext_setup_impl(ggl_ctx* c, int nprob, const int* pk, const int* G, int L);

extern "C" int ggl_ext_setup(ggl_ctx* c, const int* pk, const int* G, int L)
{
    ARGCHK(c, "ctx");
    return ext_setup_impl(c, 1, pk, G, L);
}

extern "C" int ggl_ext_setup_batch(ggl_ctx* c, int nprob, const int* pk, const int* G, int L)
{
    ARGCHK(c, "ctx");
    ARGCHK(nprob >= 1 && c->K % nprob == 0, "the ctx holds nprob problems of K/nprob instances each");
    return ext_setup_impl(c, nprob, pk, G, L);
}

static int ext_setup_impl(ggl_ctx* c, int nprob, const int* pk_all, const int* G, int L)
{
    // nprob > 1: the stack holds nprob independent problems with the same instance dimensions and the same bookkeeping array
    // G (a model-selection grid); pk_all lists the dimensions of ONE problem's instances, G covers those instances
    ARGCHK(c && pk_all, "ctx, pk");
    ARGCHK(L >= 0 && (L == 0 || G), "G, L");
    HIPCHK(hipSetDevice(c->device));
    DROP_PRE(c);
    const int Ktot = c->K, K = Ktot / nprob, p = c->p;
    const int* pk = pk_all;
    for (int k = 0; k < K; ++k) ARGCHK(pk[k] >= 1 && pk[k] <= p, "1 <= p_k <= p (the padded dimension of the ctx)");
    // the checks of helper/ext_admm_helper.py:82-102 (check_G) plus: no entry listed twice (the groups are then
    // independent, which is what lets them run in parallel; the reference processes them one after the other)
    std::vector<int> Gt((size_t)2 * K * std::max(L, 1)), gs(std::max(L, 1), 0);
    std::vector<unsigned char> seen((size_t)K * p * p, 0);
    for (int l = 0; l < L; ++l) {
        for (int k = 0; k < K; ++k) {
            const int i = G[((size_t)0 * L + l) * K + k], j = G[((size_t)1 * L + l) * K + k];
            if ((i == -1) != (j == -1)) return fail(GGL_E_ARG, "bad argument: Only row or column index specified in some group");
            if (i < -1 || j < -1)
                return fail(GGL_E_ARG, "bad argument: No negative indices allowed (only -1 for indicating a missing feature)");
            if (i >= 0) {
                if (i == j) return fail(GGL_E_ARG, "bad argument: G has entries on the diagonal!");
                if (i > j) return fail(GGL_E_ARG, "bad argument: Only upper diagonal entries should be contained in G");
                if (j >= pk[k]) return fail(GGL_E_ARG, "bad argument: indices larger as dimension were found");
                unsigned char& s = seen[((size_t)k * p + i) * p + j];
                if (s) return fail(GGL_E_ARG, "bad argument: entry (%d,%d) of instance %d is listed in more than one group", i, j, k);
                s = 1;
                gs[l] += 1;
            }
            Gt[((size_t)0 * K + k) * L + l] = i;
            Gt[((size_t)1 * K + k) * L + l] = j;
        }
        if (gs[l] == 0) return fail(GGL_E_ARG, "bad argument: G has rows with only -1 entries");
    }
    for (int* b : {c->ext_pk, c->ext_Gt, c->ext_gsize})
        if (b) (void)hipFree(b);
    c->ext_pk = c->ext_Gt = c->ext_gsize = nullptr;
    std::vector<int> pkrep((size_t)Ktot);
    for (int k = 0; k < Ktot; ++k) pkrep[k] = pk[k % K];
    HIPCHK(hipMalloc(&c->ext_pk, Ktot * sizeof(int)));
    HIPCHK(hipMalloc(&c->ext_Gt, Gt.size() * sizeof(int)));
    HIPCHK(hipMalloc(&c->ext_gsize, gs.size() * sizeof(int)));
    HIPCHK(hipMemcpyAsync(c->ext_pk, pkrep.data(), Ktot * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->ext_Gt, Gt.data(), Gt.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->ext_gsize, gs.data(), gs.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    const size_t nb = c->n * sizeof(double);
    if (!c->X1) {
        for (int i = 0; i < 2; ++i) HIPCHK(malloc_filled(&c->Lam[i], nb, c->stream));
        HIPCHK(malloc_filled(&c->X1, nb, c->stream));
        // the ext kernels write GGL_NNORM sums per (instance, chunk) twice per iteration
        int rcp = ensure_partials(c, 2 * (size_t)Ktot * ext_blocks(p) * GGL_NNORM);
        if (rcp) return rcp;
    }
    HIPCHK(hipMemsetAsync(c->X1, 0, nb, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));      // Gt / gs / pkrep are host temporaries
    c->ext_L = L;
    c->ext_nprob = nprob;
    c->lcur = 0;
    c->spec_have = false;
    return GGL_OK;
}

extern "C" int ggl_ext_set_state(ggl_ctx* c, const double* Lambda, const double* X1)
{
    ARGCHK(c && c->ext_L >= 0, "ctx (ggl_ext_setup first)");
    HIPCHK(hipSetDevice(c->device));
    DROP_PRE(c);
    const size_t nb = c->n * sizeof(double);
    if (Lambda) HIPCHK(hipMemcpyAsync(c->Lam[c->lcur], Lambda, nb, hipMemcpyHostToDevice, c->stream));
    if (X1) HIPCHK(hipMemcpyAsync(c->X1, X1, nb, hipMemcpyHostToDevice, c->stream));
    else HIPCHK(hipMemsetAsync(c->X1, 0, nb, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return GGL_OK;
}

extern "C" int ggl_ext_get_state(ggl_ctx* c, double* Lambda, double* X1)
{
    ARGCHK(c && c->ext_L >= 0, "ctx (ggl_ext_setup first)");
    HIPCHK(hipSetDevice(c->device));
    DROP_PRE(c);
    const size_t nb = c->n * sizeof(double);
    if (Lambda) HIPCHK(hipMemcpyAsync(Lambda, c->Lam[c->lcur], nb, hipMemcpyDeviceToHost, c->stream));
    if (X1) HIPCHK(hipMemcpyAsync(X1, c->X1, nb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return GGL_OK;
}

// everything of one iteration after the Omega-step
static int ext_finish(ggl_ctx* c, double rho, const double* lambda1K, const double* lambda2G, int latent, const double* mu1,
                      double* out_norms)
{
    // lambda2G: one value per problem (ext_nprob of them); out_norms: (ext_nprob, 5)
    const int K = c->K, p = c->p, nprob = c->ext_nprob, Kp = K / nprob;
    double* Om = c->Om[c->cur];
    double* OmPrev = c->Om[c->cur ^ 1];
    double* LamOld = c->Lam[c->lcur];
    double* LamNew = c->Lam[c->lcur ^ 1];
    const int* skip = c->spec_pending ? c->spec_flag : nullptr;
    int rc = upload_par(c, 1, lambda1K, 0.0, 2.0 * rho);      // lambda1_k / (2 rho)     (ext_admm_solver.py:210)
    if (rc) return rc;
    {   // lambda2 / rho of every instance slot's problem (:225)
        double* h = c->par_h + 6 * (size_t)K;
        for (int k = 0; k < K; ++k) h[k] = lambda2G[k / Kp] / rho;
        CopySegs sg;
        sg.add(c->par + 6 * (size_t)K, h, (size_t)K * sizeof(double));
        launch_copy_small(c->stream, sg);
    }
    const int nblk = ext_blocks(p);
    double* partA = c->partials;
    double* partB = c->partials + (size_t)K * nblk * GGL_NNORM;
    PB(c, GGL_PH_THETA);
    launch_ext_theta(c->stream, c->Theta, c->X, LamNew, c->W, Om, OmPrev, c->L, LamOld, c->X1, c->par + K, c->ext_pk, latent,
                     partA, K, p, skip);
    PE(c, GGL_PH_THETA);
    HIPCHK(hipGetLastError());
    if (latent) {
        rc = upload_par(c, 2, mu1, 0.0, rho);                  // mu1_k / rho             (:218)
        if (rc) return rc;
        rc = rank_step(c);
        if (rc) return rc;
    }
    PB(c, GGL_PH_DUAL);
    launch_ext_group(c->stream, LamNew, c->ext_Gt, c->ext_gsize, c->par + 6 * (size_t)K, c->ext_L, Kp, p, skip, nprob);   // :225
    launch_ext_dual(c->stream, c->X, c->X1, Om, OmPrev, c->Theta, c->L, LamNew, LamOld, c->ext_pk, latent,
                    latent ? partA : partB, K, p, skip);
    PE(c, GGL_PH_DUAL);
    PB(c, GGL_PH_REDUCE);
    c->norms_host = true;
    if (nprob == 1) {
        if (c->seq_h && c->spin_wait) c->stamp_want = c->seq_wait = ++c->seq_next;
        launch_reduce_partials(c->stream, c->partials, 1, (latent ? 1 : 2) * K * nblk, GGL_NNORM, c->norms_h,
                               c->seq_wait ? c->seq_h : nullptr, c->seq_wait);
        PE(c, GGL_PH_REDUCE);
        HIPCHK(hipGetLastError());
        rc = finish_norms(c, 1, out_norms);
    } else {
        // per problem: the rows of its Kp instances in the Theta-step's partials and (not latent) in the dual update's
        launch_reduce_partials(c->stream, partA, nprob, Kp * nblk, GGL_NNORM, c->norms_h);
        if (!latent) launch_reduce_partials(c->stream, partB, nprob, Kp * nblk, GGL_NNORM, c->norms_h + (size_t)nprob * GGL_NNORM);
        PE(c, GGL_PH_REDUCE);
        HIPCHK(hipGetLastError());
        std::vector<double> tmp((size_t)2 * nprob * GGL_NNORM, 0.0);
        rc = finish_norms(c, (latent ? 1 : 2) * nprob, tmp.data(), 1);
        for (int g = 0; rc == GGL_OK && g < nprob; ++g)
            for (int v = 0; v < GGL_NNORM; ++v)
                out_norms[(size_t)g * GGL_NNORM + v] = tmp[(size_t)g * GGL_NNORM + v] + (latent ? 0.0 : tmp[(size_t)(nprob + g) * GGL_NNORM + v]);
    }
    if (rc == GGL_OK) c->lcur ^= 1;        // a rejected speculative step leaves Lambda where it was
    return rc;
}

extern "C" int ggl_ext_admm_step(ggl_ctx* c, double rho, const double* lambda1K, double lambda2, int latent,
                                 const double* mu1, double out_norms[5])
{
    ARGCHK(c && lambda1K && out_norms, "ctx, lambda1, out_norms");
    ARGCHK(c->ext_L >= 0 && c->ext_nprob == 1, "ggl_ext_setup first");
    ARGCHK(rho > 0 && lambda2 > 0, "rho, lambda2 must be positive");
    return ggl_ext_batch_step(c, rho, lambda1K, &lambda2, latent, mu1, out_norms);
}

extern "C" int ggl_ext_batch_step(ggl_ctx* c, double rho, const double* lambda1K, const double* lambda2G, int latent,
                                  const double* mu1, double* out_norms)
{
    ARGCHK(c && lambda1K && lambda2G && out_norms, "ctx, lambda1, lambda2, out_norms");
    ARGCHK(c->ext_L >= 0, "ggl_ext_setup / ggl_ext_setup_batch first");
    ARGCHK(rho > 0, "rho must be positive");
    for (int g = 0; g < c->ext_nprob; ++g) ARGCHK(lambda2G[g] > 0, "lambda2 must be positive");
    ARGCHK(!latent || mu1, "latent needs mu1");
    HIPCHK(hipSetDevice(c->device));
    DROP_PRE(c);
    CopySegs sg;
    int rc = upload_par(c, 0, nullptr, 1.0, rho, &sg);     // beta = 1/rho for every instance   (:203)
    if (rc) return rc;
    rc = omega_step(c, latent, &sg, /*allow_spec=*/true);
    if (rc) return rc;
    rc = ext_finish(c, rho, lambda1K, lambda2G, latent, mu1, out_norms);
    if (rc != GGL_SPEC_RETRY) return rc;
    rc = omega_step(c, latent, nullptr, false);
    if (rc) return rc;
    return ext_finish(c, rho, lambda1K, lambda2G, latent, mu1, out_norms);
}

// out[k] = sum over the leading (p_k,p_k) block of ((A - B) + C)^2
static int ext_sq_k(ggl_ctx* c, const double* A, const double* B, const double* C, double* outK)
{
    launch_ext_sq(c->stream, A, B, C, c->ext_pk, c->K, c->p, c->partials);
    launch_reduce_partials(c->stream, c->partials, c->K, ext_blocks(c->p), 1, c->norms);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->norms_h, c->norms, c->K * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int k = 0; k < c->K; ++k) outK[k] = c->norms_h[k];
    return GGL_OK;
}

extern "C" int ggl_ext_kkt_residual(ggl_ctx* c, double rho, const double* lambda1K, double lambda2, int latent,
                                    const double* mu1, double* out)
{
    // solver/ext_admm_solver.py:347-392; the duals there are rho * X0, rho * X1
    ARGCHK(c && lambda1K && out, "ctx, lambda1, out");
    ARGCHK(c->ext_L >= 0, "ggl_ext_setup first");
    ARGCHK(!latent || mu1, "latent needs mu1");
    HIPCHK(hipSetDevice(c->device));
    DROP_PRE(c);
    const int K = c->K;
    double* Om = c->Om[c->cur];
    double* Lam = c->Lam[c->lcur];
    double* T1 = c->W;
    double* T2 = c->Om[c->cur ^ 1];       // Omega_{t-1} is dead once the step's sums are out
    std::vector<double> nOm(K), nTh(K), nL(K), nLam(K), v(K);
    double term[6] = {0, 0, 0, 0, 0, 0};
    int rc;
    if ((rc = ext_sq_k(c, Om, nullptr, nullptr, nOm.data()))) return rc;
    if ((rc = ext_sq_k(c, c->Theta, nullptr, nullptr, nTh.data()))) return rc;
    if ((rc = ext_sq_k(c, Lam, nullptr, nullptr, nLam.data()))) return rc;
    auto add = [&](int t, const std::vector<double>& den) {
        for (int k = 0; k < K; ++k) {
            const double r = std::sqrt(v[k]) / (1.0 + std::sqrt(den[k]));
            term[t] += r * r;
        }
    };
    // term1: Omega - phiplus(eigh(Omega - S - rho X0), 1)
    if ((rc = upload_par(c, 3, nullptr, 1.0, 1.0))) return rc;
    launch_lin3(c->stream, T1, 1.0, Om, -1.0, c->S, -rho, c->X, c->n);
    if ((rc = eig_recon(c, T1, T2, c->DvL, MAP_PHIPLUS, c->par + 3 * (size_t)K))) return rc;
    if ((rc = ext_sq_k(c, Om, T2, nullptr, v.data()))) return rc;
    add(0, nOm);
    // term2: Theta - prox_od_1norm(Theta + rho X0 - rho X1, lambda1_k)
    if ((rc = upload_par(c, 1, lambda1K, 0.0, 1.0))) return rc;
    launch_lin3(c->stream, T1, 1.0, c->Theta, rho, c->X, -rho, c->X1, c->n);
    launch_ext_prox_od(c->stream, T2, T1, c->par + K, K, c->p);
    if ((rc = ext_sq_k(c, c->Theta, T2, nullptr, v.data()))) return rc;
    add(1, nTh);
    if (latent) {
        if ((rc = ext_sq_k(c, c->L, nullptr, nullptr, nL.data()))) return rc;
        if ((rc = upload_par(c, 2, mu1, 0.0, 1.0))) return rc;
        launch_lin3(c->stream, T1, 1.0, c->L, -rho, c->X, 0.0, nullptr, c->n);
        if ((rc = eig_recon(c, T1, T2, c->DvL, MAP_RANK, c->par + 2 * (size_t)K))) return rc;
        if ((rc = ext_sq_k(c, c->L, T2, nullptr, v.data()))) return rc;
        add(2, nL);
    }
    // term4: prox_2norm_G(Lambda + rho X1, G, lambda2) - Lambda
    launch_lin3(c->stream, T1, 1.0, Lam, rho, c->X1, 0.0, nullptr, c->n);
    if ((rc = upload_par(c, 6, nullptr, lambda2, 1.0))) return rc;      // the group shrink reads its threshold per instance slot
    launch_ext_group(c->stream, T1, c->ext_Gt, c->ext_gsize, c->par + 6 * (size_t)K, c->ext_L, K, c->p, nullptr);
    if ((rc = ext_sq_k(c, T1, Lam, nullptr, v.data()))) return rc;
    add(3, nLam);
    // term5 / term6: the two equality constraints
    if ((rc = ext_sq_k(c, Om, c->Theta, latent ? c->L : nullptr, v.data()))) return rc;
    add(4, nTh);
    if ((rc = ext_sq_k(c, Lam, c->Theta, nullptr, v.data()))) return rc;
    add(5, nTh);
    HIPCHK(hipMemcpyAsync(c->info_h, c->info, K * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if ((rc = check_info(c, "kkt residual"))) return rc;
    double res = 0.0;
    for (int t = 0; t < 6; ++t) res = std::max(res, std::sqrt(term[t]));
    *out = res;
    return GGL_OK;
}

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

// ---------------------------------------------------------------------------------------------
// model selection on the snapshots: thresholded estimates and the rank of the latent component
// ---------------------------------------------------------------------------------------------
// tune_threshold (helper/model_selection.py:707-737) scores every tau of a range by AIC / eBIC of the thresholded
// estimate.  Two passes over the snapshots: (1) <S,T> and count_nonzero(T) of every (instance, tau) -- one light
// launch per tau; (2) log det T needs eigenvalues, but a larger tau zeroes a superset of entries, so two thresholds with
// the same non-zero count give the SAME matrix: only the distinct (instance, count) pairs are materialised, K at a time,
// and sent through the batched eigenvalue kernel.
extern "C" int ggl_threshold_scan(ggl_ctx* c, const double* tau, int ntau, double* out, int* n_eig)
{
    ARGCHK(c && tau && out, "ctx, tau, out");
    ARGCHK(ntau >= 1 && ntau <= 4096, "ntau in 1..4096");
    ARGCHK(c->snapT, "no snapshot taken (ggl_snapshot_k)");
    for (int j = 0; j < ntau; ++j) ARGCHK(tau[j] > 0.0, "thresholds must be positive (model_selection.py:716)");
    HIPCHK(hipSetDevice(c->device));
    DROP_PRE(c);
    const int K = c->K, p = c->p;
    const int nblk = elementwise_blocks(p);
    // ---- pass 1: sums of every (k, j)
    std::vector<int> src((size_t)K);
    for (int k = 0; k < K; ++k) src[k] = k;
    std::vector<double> tauK((size_t)ntau * K);
    for (int j = 0; j < ntau; ++j)
        for (int k = 0; k < K; ++k) tauK[(size_t)j * K + k] = tau[j];
    DevBuf dtau, dsums;
    int* dsrc = nullptr;
    HIPCHK(dtau.alloc(tauK.size()));
    HIPCHK(dsums.alloc((size_t)ntau * K * 2));
    HIPCHK(hipMalloc(&dsrc, (size_t)K * sizeof(int)));
    struct IntFree { int* p; ~IntFree() { (void)hipFree(p); } } srcfree{dsrc};
    HIPCHK(hipMemcpyAsync(dtau.p, tauK.data(), tauK.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(dsrc, src.data(), (size_t)K * sizeof(int), hipMemcpyHostToDevice, c->stream));
    for (int j = 0; j < ntau; ++j) {
        launch_threshold_sums(c->stream, c->snapT, c->S, dsrc, dtau.p + (size_t)j * K, K, p, c->partials);
        launch_reduce_partials(c->stream, c->partials, K, nblk, 2, dsums.p + (size_t)j * K * 2);
    }
    HIPCHK(hipGetLastError());
    std::vector<double> sums((size_t)ntau * K * 2);
    HIPCHK(hipMemcpyAsync(sums.data(), dsums.p, sums.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    // ---- pass 2: one eigenvalue problem per distinct thresholded matrix
    struct Item { int k, j; };
    std::vector<Item> work;
    std::vector<int> rep((size_t)K * ntau, -1);          // (k, j) -> index into work
    for (int k = 0; k < K; ++k) {
        std::vector<std::pair<double, int>> seen;       // (non-zero count, work index) of this instance
        for (int j = 0; j < ntau; ++j) {
            const double cnt = sums[((size_t)j * K + k) * 2 + 1];
            int w = -1;
            for (auto& sc : seen)
                if (sc.first == cnt) w = sc.second;
            if (w < 0) {
                w = (int)work.size();
                work.push_back({k, j});
                seen.push_back({cnt, w});
            }
            rep[(size_t)k * ntau + j] = w;
        }
    }
    const int nwork = (int)work.size();
    const int nchunk = (nwork + K - 1) / K;
    std::vector<int> wsrc((size_t)nchunk * K);
    std::vector<double> wtau((size_t)nchunk * K);
    for (int i = 0; i < nchunk * K; ++i) {
        const Item& it = work[i < nwork ? i : 0];        // the tail of the last chunk repeats a valid problem
        wsrc[i] = it.k;
        wtau[i] = tau[it.j];
    }
    DevBuf dwtau;
    int* dwsrc = nullptr;
    HIPCHK(dwtau.alloc(wtau.size()));
    HIPCHK(hipMalloc(&dwsrc, wsrc.size() * sizeof(int)));
    IntFree wsrcfree{dwsrc};
    HIPCHK(hipMemcpyAsync(dwtau.p, wtau.data(), wtau.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(dwsrc, wsrc.data(), wsrc.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    std::vector<double> d((size_t)nchunk * K * p);
    for (int ch = 0; ch < nchunk; ++ch) {
        launch_threshold_write(c->stream, c->snapT, dwsrc + (size_t)ch * K, dwtau.p + (size_t)ch * K, K, p, c->W);
        HIPCHK(hipGetLastError());
        int rc = eigvals_only(c, c->W, c->DvO);
        if (rc) return rc;
        c->dvo_valid = false;
        HIPCHK(hipMemcpyAsync(d.data() + (size_t)ch * K * p, c->DvO, (size_t)K * p * sizeof(double), hipMemcpyDeviceToHost,
                              c->stream));
        HIPCHK(hipMemcpyAsync(c->info_h, c->info, K * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        rc = check_info(c, "threshold scan");
        if (rc) return rc;
    }
    if (n_eig) *n_eig = nwork;
    std::vector<double> ld((size_t)nwork), mn((size_t)nwork);
    for (int w = 0; w < nwork; ++w) {
        double m = INFINITY, l = 0.0;
        for (int e = 0; e < p; ++e) {
            const double v = d[(size_t)w * p + e];
            m = std::min(m, v);
            l += std::log(v);
        }
        mn[w] = m;
        ld[w] = (m <= 1e-12 || !(m == m)) ? -INFINITY : l;      // robust_logdet, model_selection.py:884-894
    }
    for (int k = 0; k < K; ++k)
        for (int j = 0; j < ntau; ++j) {
            double* o = out + ((size_t)k * ntau + j) * 4;
            const int w = rep[(size_t)k * ntau + j];
            o[0] = sums[((size_t)j * K + k) * 2 + 0];
            o[1] = ld[w];
            o[2] = sums[((size_t)j * K + k) * 2 + 1];
            o[3] = mn[w];
        }
    return GGL_OK;
}

// numpy.linalg.matrix_rank of the snapshot of L_k (model_selection.py:256, :638): the number of eigenvalues with
// |lambda| > rel_tol * max|lambda|; rel_tol <= 0 selects numpy's p * eps.  out[k*4..] = { rank, max|lambda|,
// largest |lambda| NOT counted, smallest |lambda| counted } (0 where there is none): the caller sees how far the
// decision was from the tolerance.
extern "C" int ggl_selection_rank(ggl_ctx* c, double rel_tol, double* out)
{
    ARGCHK(c && out, "ctx, out");
    ARGCHK(c->snapL, "no snapshot of L (ggl_snapshot_k after a latent step)");
    HIPCHK(hipSetDevice(c->device));
    DROP_PRE(c);
    const int K = c->K, p = c->p;
    const size_t kp = (size_t)K * p;
    if (!(rel_tol > 0.0)) rel_tol = (double)p * 2.220446049250313e-16;
    HIPCHK(hipMemcpyAsync(c->W, c->snapL, c->n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    int rc = eigvals_only(c, c->W, c->DvO);
    if (rc) return rc;
    c->dvo_valid = false;
    std::vector<double> d(kp);
    HIPCHK(hipMemcpyAsync(d.data(), c->DvO, kp * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(c->info_h, c->info, K * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    rc = check_info(c, "rank of the latent component");
    if (rc) return rc;
    for (int k = 0; k < K; ++k) {
        double mx = 0.0;
        for (int e = 0; e < p; ++e) mx = std::max(mx, std::fabs(d[(size_t)k * p + e]));
        const double tol = mx * rel_tol;
        int r = 0;
        double below = 0.0, above = INFINITY;
        for (int e = 0; e < p; ++e) {
            const double a = std::fabs(d[(size_t)k * p + e]);
            if (a > tol) {
                r += 1;
                above = std::min(above, a);
            } else
                below = std::max(below, a);
        }
        out[k * 4 + 0] = r;
        out[k * 4 + 1] = mx;
        out[k * 4 + 2] = below;
        out[k * 4 + 3] = (r > 0) ? above : 0.0;
    }
    return GGL_OK;
}

static int eig_common(int K, int p, const double* A, const double* beta, double* D, double* Q, double* out, int map,
                      int eig_method)
{
    ARGCHK(K >= 1 && p >= 1 && A, "K, p, A");
    ARGCHK(eig_method != GGL_EIG_JACOBI || jacobi_fits(p), "GGL_EIG_JACOBI needs p <= GGL_JACOBI_MAX_P");
    const bool jac = (eig_method == GGL_EIG_JACOBI) || (eig_method == GGL_EIG_AUTO && jacobi_fits(p));
    const size_t n = (size_t)K * p * p, kp = (size_t)K * p;
    DevBuf dA, dD, dR, dO, dB, dE, dS;
    int* dinfo = nullptr;
    HIPCHK(dA.alloc(n));
    HIPCHK(dD.alloc(kp));
    HIPCHK(dB.alloc(K));
    HIPCHK(hipMalloc(&dinfo, K * sizeof(int)));
    struct InfoFree { int* p; ~InfoFree() { (void)hipFree(p); } } infofree{dinfo};
    UP(dA.p, A, n);
    if (beta) UP(dB.p, beta, K);
    if (out) HIPCHK(dO.alloc(n));
    std::vector<int> info(K);
    if (jac) {
        if (Q) HIPCHK(dR.alloc(n));
        HIPCHK(launch_jacobi(nullptr, dA.p, dD.p, dR.p, dO.p, map, beta ? dB.p : nullptr, dinfo, K, p));
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(info.data(), dinfo, K * sizeof(int), hipMemcpyDeviceToHost));
        for (int k = 0; k < K; ++k)
            if (info[k] < 0) return fail(GGL_E_SOLVER, "Jacobi eigensolver did not converge (instance %d)", k);
    } else {
        rocblas_handle h;
        if (rocblas_create_handle(&h) != rocblas_status_success) return fail(GGL_E_SOLVER, "rocblas_create_handle");
        HIPCHK(dE.alloc(kp));
        rocblas_status st = rocsolver_dsyevd_strided_batched(h, (Q || out) ? rocblas_evect_original : rocblas_evect_none,
                                                             rocblas_fill_upper, p, dA.p, p, (rocblas_stride)p * p,
                                                             dD.p, p, dE.p, p, dinfo, K);
        if (st == rocblas_status_success && out) {
            hipError_t e = dS.alloc(2 * kp);
            if (e == hipSuccess) launch_recon(nullptr, dO.p, dA.p, dD.p, beta ? dB.p : nullptr, map, K, p, dS.p);
        }
        (void)hipDeviceSynchronize();
        rocblas_destroy_handle(h);
        if (st != rocblas_status_success) return fail(GGL_E_SOLVER, "rocsolver_dsyevd_strided_batched: status %d", (int)st);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpy(info.data(), dinfo, K * sizeof(int), hipMemcpyDeviceToHost));
        for (int k = 0; k < K; ++k)
            if (info[k] != 0) return fail(GGL_E_SOLVER, "rocSOLVER syevd did not converge (instance %d)", k);
    }
    if (out) DOWN(out, dO.p, n);
    if (D || Q) {
        // NumPy convention across the ABI: ascending eigenvalues, eigenvectors in columns.
        std::vector<double> hd(kp), hr;
        DOWN(hd.data(), dD.p, kp);
        if (Q) { hr.resize(n); DOWN(hr.data(), jac ? dR.p : dA.p, n); }
        std::vector<int> idx(p);
        for (int k = 0; k < K; ++k) {
            std::iota(idx.begin(), idx.end(), 0);
            const double* dk = hd.data() + (size_t)k * p;
            std::stable_sort(idx.begin(), idx.end(), [dk](int a, int b) { return dk[a] < dk[b]; });
            for (int m = 0; m < p; ++m) {
                if (D) D[(size_t)k * p + m] = dk[idx[m]];
                if (Q) {
                    const double* row = hr.data() + (size_t)k * p * p + (size_t)idx[m] * p;
                    for (int i = 0; i < p; ++i) Q[(size_t)k * p * p + (size_t)i * p + m] = row[i];
                }
            }
        }
    }
    return GGL_OK;
}

extern "C" int ggl_dev_symm(int K, int p, const double* A, const double* B, const double* E, const double* coef5K,
                            double* C, double* C2, int variant)
{
    ARGCHK(K >= 1 && p >= 1 && A && B && coef5K && C, "arguments");
    ARGCHK(variant < 0 || symm_variant_built(variant), "product-kernel variant not in this build");
    const size_t n = (size_t)K * p * p;
    DevBuf dA, dB, dE, dC, dC2, dcoef;
    HIPCHK(dA.alloc(n));
    HIPCHK(dB.alloc(n));
    HIPCHK(dC.alloc(n));
    std::vector<double> cw((size_t)K * NS_NCOEF, 0.0);     // rows {cI,cAcc,cE,dI,dC} widened by dE = 0
    for (int k = 0; k < K; ++k) std::copy(coef5K + (size_t)k * 5, coef5K + (size_t)k * 5 + 5, cw.begin() + (size_t)k * NS_NCOEF);
    HIPCHK(dcoef.alloc(cw.size()));
    UP(dA.p, A, n);
    UP(dB.p, B, n);
    UP(dcoef.p, cw.data(), cw.size());
    if (E) { HIPCHK(dE.alloc(n)); UP(dE.p, E, n); }
    if (C2) HIPCHK(dC2.alloc(n));
    launch_symm(nullptr, dA.p, dB.p, dC.p, C2 ? dC2.p : nullptr, E ? dE.p : nullptr, dcoef.p, K, p, variant);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    DOWN(C, dC.p, n);
    if (C2) DOWN(C2, dC2.p, n);
    return GGL_OK;
}

// kernel unit test of the bound partials: C = A B with the product kernel's epilogue partials, then the row sums of
// |C| (K,p), |C|_F^2 (K) and the spectral bound sqrt(min(|C|_inf, Collatz-Wielandt, |C|_F)) (K) from them
extern "C" int ggl_dev_symm_bounds(int K, int p, const double* A, const double* B, int variant, double* C,
                                   double* rowsum_out, double* fro2_out, double* bound_out)
{
    ARGCHK(K >= 1 && p >= 1 && A && B && C && rowsum_out && fro2_out && bound_out, "arguments");
    ARGCHK(variant < 0 || symm_variant_built(variant), "product-kernel variant not in this build");
    const int tile = symm_bounds_tile(K, p, variant);
    ARGCHK(tile != 0, "this variant / p has no bound partials (direct-to-LDS kernels, even p)");
    const int T = (p + tile - 1) / tile, ntile = T * (T + 1) / 2, nib = bound_rows_blocks(p);
    const size_t n = (size_t)K * p * p;
    DevBuf dA, dB, dC, dcoef, drow, dfro, dd, dinf, dout;
    unsigned long long* cw = nullptr;
    unsigned* cnt = nullptr;
    HIPCHK(dA.alloc(n)); HIPCHK(dB.alloc(n)); HIPCHK(dC.alloc(n));
    HIPCHK(drow.alloc((size_t)K * T * p)); HIPCHK(dfro.alloc((size_t)K * ntile)); HIPCHK(dd.alloc((size_t)K * p));
    HIPCHK(dinf.alloc((size_t)K * nib)); HIPCHK(dout.alloc(K));
    HIPCHK(hipMalloc(&cw, K * sizeof(unsigned long long)));
    HIPCHK(hipMalloc(&cnt, K * sizeof(unsigned)));
    struct Free2 { void *a, *b; ~Free2() { (void)hipFree(a); (void)hipFree(b); } } free2{cw, cnt};
    HIPCHK(hipMemset(cw, 0, K * sizeof(unsigned long long)));
    HIPCHK(hipMemset(cnt, 0, K * sizeof(unsigned)));
    HIPCHK(hipMemset(drow.p, 0xff, (size_t)K * T * p * sizeof(double)));      // every slot must be written by the kernel
    HIPCHK(hipMemset(dfro.p, 0xff, (size_t)K * ntile * sizeof(double)));
    std::vector<double> coef((size_t)K * NS_NCOEF, 0.0);
    for (int k = 0; k < K; ++k) coef[(size_t)k * NS_NCOEF + 1] = 1.0;
    HIPCHK(dcoef.alloc(coef.size()));
    UP(dA.p, A, n);
    UP(dB.p, B, n);
    UP(dcoef.p, coef.data(), coef.size());
    launch_symm(nullptr, dA.p, dB.p, dC.p, nullptr, nullptr, dcoef.p, K, p, variant, nullptr, drow.p, dfro.p);
    launch_bound_rows(nullptr, drow.p, T, K, p, dd.p, dinf.p);
    launch_cw_final(nullptr, dC.p, dd.p, K, p, dinf.p, dfro.p, ntile, cw, cnt, dout.p, nullptr, nullptr, nullptr, 0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    DOWN(C, dC.p, n);
    DOWN(rowsum_out, dd.p, (size_t)K * p);
    DOWN(bound_out, dout.p, K);
    std::vector<double> fr((size_t)K * ntile);
    DOWN(fr.data(), dfro.p, fr.size());
    for (int k = 0; k < K; ++k) {
        double sq = 0.0;
        for (int t = 0; t < ntile; ++t) sq += fr[(size_t)k * ntile + t];
        fro2_out[k] = sq;
    }
    return GGL_OK;
}

extern "C" int ggl_dev_symm_bench(int K, int p, int variant, int iters, double* ms_out)
{
    ARGCHK(K >= 1 && p >= 1 && iters >= 1 && ms_out, "arguments");
    ARGCHK(variant < 0 || symm_variant_built(variant), "product-kernel variant not in this build");
    const size_t n = (size_t)K * p * p;
    std::vector<double> h(n), coef((size_t)K * NS_NCOEF, 0.0);
    unsigned long long s = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        h[i] = (double)(s >> 11) / 9007199254740992.0 - 0.5;
    }
    for (int k = 0; k < K; ++k) coef[(size_t)k * NS_NCOEF + 1] = 1.0 / p;
    DevBuf dA, dB, dC, dcoef;
    HIPCHK(dA.alloc(n));
    HIPCHK(dB.alloc(n));
    HIPCHK(dC.alloc(n));
    HIPCHK(dcoef.alloc(coef.size()));
    UP(dA.p, h.data(), n);
    UP(dB.p, h.data(), n);
    UP(dcoef.p, coef.data(), coef.size());
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch_symm(nullptr, dA.p, dB.p, dC.p, nullptr, nullptr, dcoef.p, K, p, variant);
    HIPCHK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) launch_symm(nullptr, dA.p, dB.p, dC.p, nullptr, nullptr, dcoef.p, K, p, variant);
    HIPCHK(hipEventRecord(e1, nullptr));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    HIPCHK(hipGetLastError());
    *ms_out = ms / iters;
    return GGL_OK;
}

#ifdef GGL_DEV
// What does a change of kernel between dependent launches cost?  mode 0: `iters` products; 1: `iters` x (product, then a
// one-thread kernel that stores a word); 2: `iters` of the one-thread kernel; 3: `iters` x (product, elementwise scale by 1 of
// the output: an LDS-free kernel over the same data).  ms per repetition (tools/kernel_switch_cost.py).
extern "C" int ggl_dev_switch_bench(int K, int p, int variant, int iters, int mode, double* ms_out)
{
    ARGCHK(K >= 1 && p >= 1 && iters >= 1 && ms_out && mode >= 0 && mode <= 3, "arguments");
    const size_t n = (size_t)K * p * p;
    std::vector<double> h(n, 0.25), coef((size_t)K * NS_NCOEF, 0.0);
    for (int k = 0; k < K; ++k) coef[(size_t)k * NS_NCOEF + 1] = 1.0 / p;
    DevBuf dA, dB, dC, dcoef, dflag;
    HIPCHK(dA.alloc(n));
    HIPCHK(dB.alloc(n));
    HIPCHK(dC.alloc(n));
    HIPCHK(dcoef.alloc(coef.size()));
    HIPCHK(dflag.alloc(8));
    UP(dA.p, h.data(), n);
    UP(dB.p, h.data(), n);
    UP(dcoef.p, coef.data(), coef.size());
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    auto rep = [&](int i) {
        if (mode != 2) launch_symm(nullptr, dA.p, dB.p, dC.p, nullptr, nullptr, dcoef.p, K, p, variant);
        if (mode == 1 || mode == 2) launch_set_flag(nullptr, (unsigned long long*)dflag.p, (unsigned long long)i);
        if (mode == 3) launch_scale(nullptr, dC.p, 1.0, n);
    };
    for (int i = 0; i < 3; ++i) rep(i);
    HIPCHK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) rep(i);
    HIPCHK(hipEventRecord(e1, nullptr));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    HIPCHK(hipGetLastError());
    *ms_out = ms / iters;
    return GGL_OK;
}

// A yardstick for the product kernel: what the vendor's FP64 GEMM reaches on this chip at the same shapes.
// mode 0: rocblas_dgemm_strided_batched C = A B (N,N);  1: C = A^T B (the operand layout of k_symm_tn / k_symm_dl);
// 2: rocblas_dsyrk_strided_batched C = A A^T, one triangle (p^3 flop per instance, like a symmetric product);
// 3: rocblas_dsyrkx_strided_batched C = A B^T, one triangle -- the library's form of OUR product (A B symmetric);
// 4: k_symm (variant by size) for comparison in the same process.  ms per call.  Dev library only; nothing on the solver's path.
extern "C" int ggl_dev_vendor_bench(int K, int p, int mode, int iters, double* ms_out)
{
    ARGCHK(K >= 1 && p >= 1 && iters >= 1 && ms_out && mode >= 0 && mode <= 4, "arguments");
    const size_t n = (size_t)K * p * p;
    std::vector<double> h(n), coef((size_t)K * NS_NCOEF, 0.0);
    unsigned long long s = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        h[i] = (double)(s >> 11) / 9007199254740992.0 - 0.5;
    }
    for (int k = 0; k < K; ++k) coef[(size_t)k * NS_NCOEF + 1] = 1.0 / p;
    DevBuf dA, dB, dC, dcoef;
    HIPCHK(dA.alloc(n));
    HIPCHK(dB.alloc(n));
    HIPCHK(dC.alloc(n));
    HIPCHK(dcoef.alloc(coef.size()));
    UP(dA.p, h.data(), n);
    UP(dB.p, h.data(), n);
    UP(dcoef.p, coef.data(), coef.size());
    rocblas_handle hd = nullptr;
    if (rocblas_create_handle(&hd) != rocblas_status_success) return fail(GGL_E_SOLVER, "rocblas_create_handle failed");
    const double one = 1.0 / p, zero = 0.0;
    const rocblas_stride st = (rocblas_stride)p * p;
    rocblas_status rs = rocblas_status_success;
    auto rep = [&]() {
        switch (mode) {
            case 0: rs = rocblas_dgemm_strided_batched(hd, rocblas_operation_none, rocblas_operation_none, p, p, p, &one, dA.p, p, st,
                                                       dB.p, p, st, &zero, dC.p, p, st, K); break;
            case 1: rs = rocblas_dgemm_strided_batched(hd, rocblas_operation_transpose, rocblas_operation_none, p, p, p, &one, dA.p, p,
                                                       st, dB.p, p, st, &zero, dC.p, p, st, K); break;
            case 2: rs = rocblas_dsyrk_strided_batched(hd, rocblas_fill_upper, rocblas_operation_none, p, p, &one, dA.p, p, st, &zero,
                                                       dC.p, p, st, K); break;
            case 3: rs = rocblas_dsyrkx_strided_batched(hd, rocblas_fill_upper, rocblas_operation_none, p, p, &one, dA.p, p, st, dB.p, p,
                                                        st, &zero, dC.p, p, st, K); break;
            default: launch_symm(nullptr, dA.p, dB.p, dC.p, nullptr, nullptr, dcoef.p, K, p, -1); break;
        }
    };
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) rep();
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) rep();
    HIPCHK(hipEventRecord(e1, nullptr));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)rocblas_destroy_handle(hd);
    if (rs != rocblas_status_success) return fail(GGL_E_SOLVER, "rocBLAS call failed: status %d", (int)rs);
    HIPCHK(hipGetLastError());
    *ms_out = ms / iters;
    return GGL_OK;
}
#endif

// LDS stages of the int8 product kernel: 1 (default: two workgroups per CU cover each other's loads) or 2 (double buffer)
#ifdef GGL_DEV
extern "C" int ggl_dev_i8_stages(int n)
{
    ARGCHK(n == 1 || n == 2, "1 or 2 stages");
    symm_i8_set_stages(n);
    return GGL_OK;
}

// Symmetric product on the INT8 matrix cores (gemm_i8.hip): C = A B from S int8 slices per operand, slice
// pairs t + u <= dmax.  A, B, C: (K,p,p) host arrays, |A| <= scaleA, |B| <= scaleB entrywise (powers of two).
// ms_out[0]: slicing both operands (two launches), ms_out[1]: one product launch (mean of iters), ms_out[2]: overflow flag.
extern "C" int ggl_dev_symm_i8(int K, int p, int S, int dmax, const double* A, const double* B, double scaleA, double scaleB,
                               double* C, int iters, double* ms_out)
{
    ARGCHK(K >= 1 && p >= 1 && iters >= 1 && A && B && C && ms_out, "arguments");
    ARGCHK(S >= 2 && S <= 8, "2 <= S <= 8");
    const size_t n = (size_t)K * p * p;
    const int P = (p + 63) / 64 * 64;
    const size_t nslice = (size_t)S * K * P * P;
    DevBuf dA, dB, dC, dsc;
    HIPCHK(dA.alloc(n));
    HIPCHK(dB.alloc(n));
    HIPCHK(dC.alloc(n));
    HIPCHK(dsc.alloc(2 * (size_t)K));
    DevBuf dpar;
    HIPCHK(dpar.alloc(12 * (size_t)K));
    {
        std::vector<double> par(12 * (size_t)K, 0.0);
        for (int k = 0; k < K; ++k) { par[12 * k + 1] = 1.0; par[12 * k + 8] = scaleA * scaleB; par[12 * k + 9] = par[12 * k + 10] = 1.0; }
        UP(dpar.p, par.data(), par.size());
    }
    int8_t *sA = nullptr, *sB = nullptr;
    int* flag = nullptr;
    HIPCHK(hipMalloc(&sA, nslice));
    HIPCHK(hipMalloc(&sB, nslice));
    HIPCHK(hipMalloc(&flag, sizeof(int)));
    HIPCHK(hipMemset(flag, 0, sizeof(int)));
    std::vector<double> sc(2 * (size_t)K);
    for (int k = 0; k < K; ++k) { sc[k] = scaleA; sc[K + k] = scaleB; }
    UP(dA.p, A, n);
    UP(dB.p, B, n);
    UP(dsc.p, sc.data(), sc.size());
    hipEvent_t e0, e1, e2;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventCreate(&e2));
    int rc = GGL_OK;
    launch_slice_i8(nullptr, dA.p, dsc.p, sA, K, p, S, flag);       // warm-up
    HIPCHK(hipEventRecord(e0, nullptr));
    launch_slice_i8(nullptr, dA.p, dsc.p, sA, K, p, S, flag);
    launch_slice_i8(nullptr, dB.p, dsc.p + K, sB, K, p, S, flag);
    HIPCHK(hipEventRecord(e1, nullptr));
    if (!launch_symm_i8(nullptr, sA, sB, dpar.p, dC.p, K, p, S, dmax))
        rc = fail(GGL_E_ARG, "bad argument: (S, dmax) = (%d, %d) is not instantiated", S, dmax);
    if (!rc) {
        HIPCHK(hipEventRecord(e1, nullptr));
        for (int i = 0; i < iters; ++i) launch_symm_i8(nullptr, sA, sB, dpar.p, dC.p, K, p, S, dmax);
        HIPCHK(hipEventRecord(e2, nullptr));
        HIPCHK(hipEventSynchronize(e2));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e1, e2));
        ms_out[1] = ms / iters;
        int hflag = 0;
        HIPCHK(hipMemcpy(&hflag, flag, sizeof(int), hipMemcpyDeviceToHost));
        ms_out[2] = hflag;
        HIPCHK(hipGetLastError());
        DOWN(C, dC.p, n);
    }
    {
        // slicing time, measured on its own
        HIPCHK(hipEventRecord(e0, nullptr));
        launch_slice_i8(nullptr, dA.p, dsc.p, sA, K, p, S, flag);
        launch_slice_i8(nullptr, dB.p, dsc.p + K, sB, K, p, S, flag);
        HIPCHK(hipEventRecord(e1, nullptr));
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        ms_out[0] = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipEventDestroy(e2);
    (void)hipFree(sA);
    (void)hipFree(sB);
    (void)hipFree(flag);
    return rc;
}

// The whole Omega-step on the int8 matrix cores (gemm_i8.hip: i8_omega_plan / i8_omega_run), stand-alone: Omega = phiplus(W)
// for a (K,p,p) host stack W, beta (K), spectral bounds cbound (K) >= lambda_max(W^2 + 4 beta I).  cfg = {s_full, s_f2, s_gf2,
// s_ye, d_ye} (0: defaults).  ms_out = {mean milliseconds of one step (slicing of W + products), products, overflow flag,
// algorithmic units (fp64 products the schedule stands for)}.
extern "C" int ggl_dev_omega_i8(int K, int p, const double* W, const double* beta, const double* cbound, const int* cfg5,
                                double tol, double* Omega, int iters, double* ms_out)
{
    ARGCHK(K >= 1 && p >= 1 && W && beta && cbound && Omega && iters >= 1 && ms_out, "arguments");
    const size_t n = (size_t)K * p * p;
    I8Omega w;
    if (i8_omega_alloc(&w, K, p) != 0) return fail(GGL_E_HIP, "i8 workspace: allocation failed");
    DevBuf dW, dA, dB, dY, dF, dF2, dOm;
    int rc = GGL_OK;
    do {
        if (dW.alloc(n) || dA.alloc(n) || dB.alloc(n) || dY.alloc(n) || dF.alloc(n) || dF2.alloc(n) || dOm.alloc(n)) {
            rc = fail(GGL_E_HIP, "allocation failed");
            break;
        }
        if (hipMemcpy(dW.p, W, n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { rc = fail(GGL_E_HIP, "upload"); break; }
        I8Cfg cfg;
        if (cfg5 && cfg5[0] > 0) { cfg.s_full = cfg5[0]; cfg.s_f2 = cfg5[1]; cfg.s_gf2 = cfg5[2]; cfg.s_ye = cfg5[3]; cfg.d_ye = cfg5[4]; }
        I8Bufs bufs = {dW.p, dA.p, dB.p, dY.p, dF.p, dF2.p, dOm.p};
        I8Prog prog;
        const int np = i8_omega_plan(&w, cbound, beta, 0, K, tol, 9, cfg, bufs, &prog);
        if (np <= 0) { rc = fail(GGL_E_ARG, "bad argument: no two-step schedule for these bounds (%d)", np); break; }
        if (hipMemcpy(w.par, w.par_h, (size_t)I8_MAXPROD * K * 12 * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(w.wscale, w.wscale_h, K * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
            rc = fail(GGL_E_HIP, "upload of the parameter rows");
            break;
        }
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        bool ok = i8_omega_run(nullptr, &w, prog, dW.p);      // warm-up (and the result)
        (void)hipEventRecord(e0, nullptr);
        for (int i = 0; ok && i < iters; ++i) ok = i8_omega_run(nullptr, &w, prog, dW.p);
        (void)hipEventRecord(e1, nullptr);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        if (!ok) { rc = fail(GGL_E_ARG, "bad argument: a slice configuration of the plan is not instantiated"); break; }
        if (hipGetLastError() != hipSuccess) { rc = fail(GGL_E_HIP, "i8 Omega-step: launch failed"); break; }
        int hflag = 0;
        (void)hipMemcpy(&hflag, w.flag, sizeof(int), hipMemcpyDeviceToHost);
        ms_out[0] = ms / iters;
        ms_out[1] = np;
        ms_out[2] = hflag;
        ms_out[3] = prog.units;
        if (hipMemcpy(Omega, dOm.p, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(GGL_E_HIP, "download"); break; }
    } while (0);
    i8_omega_free(&w);
    return rc;
}
#endif  // GGL_DEV (the int8 route: measured, rejected -- DESIGN 9.4)

// the LDS-resident Omega-step (omega_lds.hip) stand-alone: Omega = phiplus(Theta - L - X - beta S, beta) of K instances in one
// launch.  L may be NULL.  out = {ms per launch, fallback flag, products summed over the instances, table entries}
extern "C" int ggl_dev_omega_lds(int K, int p, const double* Theta, const double* L, const double* X, const double* S,
                                 const double* beta, double tol, int degrees, double* Omega, double* cbound, int iters, double* out)
{
    ARGCHK(K >= 1 && p >= 1 && Theta && X && S && beta && Omega && iters >= 1 && out, "arguments");
    ARGCHK(p <= omega_lds_max_p(), "p above the LDS-resident kernel's range");
    const int waves = degrees / 1000;             // degrees + 1000 * waves: 4 or 8 waves per workgroup (0: by size)
    degrees %= 1000;
    ARGCHK(waves == 0 || waves == 4 || waves == 8, "waves per workgroup: 4 or 8");
    const size_t n = (size_t)K * p * p;
    std::vector<double> tab((size_t)OMEGA_LDS_MAXTAB * OMEGA_LDS_ENT);
    double lnq = 0.0;
    const int ntab = omega_lds_build_table(tol, degrees, tab.data(), OMEGA_LDS_MAXTAB, &lnq);
    ARGCHK(ntab >= 1, "empty schedule table");
    DevBuf dT, dL, dX, dS, dB, dO, dTab, dC, dMisc;
    HIPCHK(dT.alloc(n)); HIPCHK(dX.alloc(n)); HIPCHK(dS.alloc(n)); HIPCHK(dB.alloc(K)); HIPCHK(dO.alloc(n));
    HIPCHK(dTab.alloc(tab.size())); HIPCHK(dC.alloc(K)); HIPCHK(dMisc.alloc(24));
    if (L) { HIPCHK(dL.alloc(n)); UP(dL.p, L, n); }
    UP(dT.p, Theta, n); UP(dX.p, X, n); UP(dS.p, S, n); UP(dB.p, beta, K); UP(dTab.p, tab.data(), tab.size());
    HIPCHK(hipMemset(dMisc.p, 0, 24 * sizeof(double)));
    int* flag = (int*)dMisc.p;
    int* flag_h = flag + 2;                       // (device memory stands in for the pinned mirror here)
    unsigned long long* units = (unsigned long long*)(dMisc.p + 20);
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    bool ok = launch_omega_lds(nullptr, dT.p, L ? dL.p : nullptr, dX.p, dS.p, dB.p, dO.p, dTab.p, ntab, lnq, K, p, flag,
                               flag_h, 0, units, dC.p, nullptr, waves);
    (void)hipEventRecord(e0, nullptr);
    for (int i = 0; ok && i < iters; ++i)
        launch_omega_lds(nullptr, dT.p, L ? dL.p : nullptr, dX.p, dS.p, dB.p, dO.p, dTab.p, ntab, lnq, K, p, flag, flag_h, 0,
                         nullptr, dC.p, (long long*)(dMisc.p + 4), waves);
    (void)hipEventRecord(e1, nullptr);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (!ok || hipGetLastError() != hipSuccess) return fail(GGL_E_HIP, "LDS Omega-step: launch failed");
    int hflag = 0;
    unsigned long long hu = 0;
    HIPCHK(hipMemcpy(&hflag, flag, sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&hu, units, sizeof(hu), hipMemcpyDeviceToHost));
    out[0] = ms / iters; out[1] = hflag; out[2] = (double)hu; out[3] = ntab;
    {
        // phase stamps of instance 0 (100 MHz wall clock): out[4..12] = us since the kernel's first instruction; out[13] = its products
        long long ts[16];
        HIPCHK(hipMemcpy(ts, dMisc.p + 4, sizeof(ts), hipMemcpyDeviceToHost));
        for (int i = 0; i < 9; ++i) out[4 + i] = (double)(ts[i] - ts[0]) * 0.01;
        out[13] = (double)ts[9];
        out[14] = (double)(ts[14] - ts[10]) * 0.01;                                        // the A' product alone
    }
    DOWN(Omega, dO.p, n);
    if (cbound) DOWN(cbound, dC.p, K);
    return GGL_OK;
}

#ifdef GGL_DEV
// persistent-chain probe (gemm_sym.hip): out = {ms per chain as nprod launches, ms per chain as one cooperative launch,
// grid of the cooperative launch, max |difference| between the two chains' results (same tile code: 0 unless a workgroup
// read stale data across a grid barrier), barrier time-out flag}.  The chain is X <- I - 1.5 X^2 on a dense symmetric
// start of norm <= 1/2 (the quadratic map keeps the spectrum in [-1, 1], so it can run for any number of products).
extern "C" int ggl_dev_chain_probe(int K, int p, int variant, int nprod, int iters, int two_level, double* out)
{
    ARGCHK(K >= 1 && p >= 2 && (p & 1) == 0 && nprod >= 1 && iters >= 1 && out, "arguments (p even)");
    const size_t n = (size_t)K * p * p;
    std::vector<double> h(n, 0.0), coef((size_t)K * NS_NCOEF, 0.0);
    unsigned long long s = 88172645463325252ull;
    for (int k = 0; k < K; ++k) {
        double* M = h.data() + (size_t)k * p * p;
        for (int i = 0; i < p; ++i)
            for (int j = i; j < p; ++j) {
                s ^= s << 13; s ^= s >> 7; s ^= s << 17;
                M[(size_t)i * p + j] = M[(size_t)j * p + i] = ((double)(s >> 11) / 9007199254740992.0 - 0.5) / p;
            }
        coef[(size_t)k * NS_NCOEF + 0] = 1.0;
        coef[(size_t)k * NS_NCOEF + 1] = -1.5;
    }
    DevBuf dX0, dX1, dcoef, dbar;
    HIPCHK(dX0.alloc(n));
    HIPCHK(dX1.alloc(n));
    HIPCHK(dcoef.alloc(coef.size()));
    HIPCHK(dbar.alloc(8));
    UP(dcoef.p, coef.data(), coef.size());
    unsigned* bar = reinterpret_cast<unsigned*>(dbar.p);
    double* last = (nprod & 1) ? dX1.p : dX0.p;
    auto chain_launches = [&]() {
        for (int j = 0; j < nprod; ++j)
            launch_symm(nullptr, (j & 1) ? dX1.p : dX0.p, (j & 1) ? dX1.p : dX0.p, (j & 1) ? dX0.p : dX1.p, nullptr, nullptr,
                        dcoef.p, K, p, variant);
    };
    int grid = 0;
    auto chain_persistent = [&]() -> int {
        hipError_t e = hipMemsetAsync(dbar.p, 0, 8 * sizeof(double), nullptr);
        if (e != hipSuccess) return -1;
        grid = launch_chain_probe(nullptr, dX0.p, dX1.p, dcoef.p, K, p, nprod, variant, bar, bar + 1, two_level);
        return grid;
    };
    // the two chains from the same start must agree bit for bit
    std::vector<double> r1(n), r2(n);
    UP(dX0.p, h.data(), n);
    chain_launches();
    DOWN(r1.data(), last, n);
    UP(dX0.p, h.data(), n);
    HIPCHK(hipMemset(dX1.p, 0, n * sizeof(double)));
    int g = chain_persistent();
    ARGCHK(g != 0, "no probe instance of this variant (16, 17, 20)");
    if (g < 0) return fail(GGL_E_HIP, "cooperative launch of the chain probe failed: %s", hipGetErrorString(hipGetLastError()));
    DOWN(r2.data(), last, n);
    double dev = 0.0;
    for (size_t i = 0; i < n; ++i) dev = std::max(dev, std::fabs(r1[i] - r2[i]));
    out[3] = dev;
    unsigned flags[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpy(flags, dbar.p, sizeof(flags), hipMemcpyDeviceToHost));
    out[4] = flags[1];
    out[2] = grid;
    if (flags[1]) return GGL_OK;          // a barrier timed out: do not time it
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    float ms = 0.f;
    for (int i = 0; i < 3; ++i) chain_launches();
    HIPCHK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) chain_launches();
    HIPCHK(hipEventRecord(e1, nullptr));
    HIPCHK(hipEventSynchronize(e1));
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    out[0] = ms / iters;
    for (int i = 0; i < 3 + iters; ++i) {
        if (i == 3) HIPCHK(hipEventRecord(e0, nullptr));
        if (chain_persistent() <= 0) return fail(GGL_E_HIP, "cooperative launch of the chain probe failed");
    }
    HIPCHK(hipEventRecord(e1, nullptr));
    HIPCHK(hipEventSynchronize(e1));
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    out[1] = ms / iters;
    HIPCHK(hipMemcpy(flags, dbar.p, sizeof(flags), hipMemcpyDeviceToHost));
    out[4] = flags[1];
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    HIPCHK(hipGetLastError());
    return GGL_OK;
}

// k_omega_chain on a synthetic program: nprod dependent products X <- I - 1.5 X^2 (ping-pong between two stacks) as nprod
// launches of the three-stage 64x64 kernel (out[0], ms per chain) and as ONE persistent launch with per-instance
// dependencies (out[1]); out[2] = persistent workgroups, out[3] = max |difference| of the results (same tile code: 0 unless
// a hand-off delivered stale data), out[4] = completion flag of k_chain_check, out[5..5+K) = done counters after the run
extern "C" int ggl_dev_chain_run(int K, int p, int nprod, int iters, double* out)
{
    ARGCHK(K >= 1 && p >= 2 && (p & 1) == 0 && nprod >= 1 && nprod <= CHAIN_MAX_OPS && iters >= 1 && out, "arguments (p even)");
    const size_t n = (size_t)K * p * p;
    std::vector<double> h(n, 0.0), coef((size_t)K * NS_NCOEF, 0.0);
    unsigned long long s = 88172645463325252ull;
    for (int k = 0; k < K; ++k) {
        double* M = h.data() + (size_t)k * p * p;
        for (int i = 0; i < p; ++i)
            for (int j = i; j < p; ++j) {
                s ^= s << 13; s ^= s >> 7; s ^= s << 17;
                M[(size_t)i * p + j] = M[(size_t)j * p + i] = ((double)(s >> 11) / 9007199254740992.0 - 0.5) / p;
            }
        coef[(size_t)k * NS_NCOEF + 0] = 1.0;
        coef[(size_t)k * NS_NCOEF + 1] = -1.5;
    }
    DevBuf dX0, dX1, dcoef, dcnt;
    HIPCHK(dX0.alloc(n));
    HIPCHK(dX1.alloc(n));
    HIPCHK(dcoef.alloc(coef.size()));
    const size_t ncw = (size_t)K * CHAIN_CNT_STRIDE;          // 32-bit words
    HIPCHK(dcnt.alloc(ncw / 2 + 8));
    UP(dcoef.p, coef.data(), coef.size());
    unsigned* cnt = reinterpret_cast<unsigned*>(dcnt.p);
    int* flag = reinterpret_cast<int*>(cnt + ncw);
    const int aux = getenv("GGL_CHAIN_AUX") ? atoi(getenv("GGL_CHAIN_AUX")) : 16;
    double* last = (nprod & 1) ? dX1.p : dX0.p;
    const int T = (p + 63) / 64;
    ChainProg P;
    P.nops = nprod; P.K = K; P.p = p; P.ntiles = T * (T + 1) / 2;
    P.begin[0] = 0;
    for (int j = 0; j < nprod; ++j) {
        SymmOp o{};
        o.A = o.B = (j & 1) ? dX1.p : dX0.p;
        o.C = (j & 1) ? dX0.p : dX1.p;
        o.coef = dcoef.p;
        P.op[j] = o;
        P.begin[j + 1] = P.begin[j] + P.ntiles;
    }
    auto chain_launches = [&]() {
        for (int j = 0; j < nprod; ++j)
            launch_symm(nullptr, (j & 1) ? dX1.p : dX0.p, (j & 1) ? dX1.p : dX0.p, (j & 1) ? dX0.p : dX1.p, nullptr, nullptr,
                        dcoef.p, K, p, 17);
    };
    int grid = 0;
    auto chain_persistent = [&]() -> int {
        if (hipMemsetAsync(dcnt.p, 0, (ncw / 2 + 8) * sizeof(double), nullptr) != hipSuccess) return -1;
        grid = launch_omega_chain(nullptr, P, cnt, flag, flag + 1, aux);
        return grid;
    };
    std::vector<double> r1(n), r2(n);
    UP(dX0.p, h.data(), n);
    chain_launches();
    DOWN(r1.data(), last, n);
    UP(dX0.p, h.data(), n);
    HIPCHK(hipMemset(dX1.p, 0, n * sizeof(double)));
    if (chain_persistent() <= 0) return fail(GGL_E_HIP, "k_omega_chain launch failed");
    DOWN(r2.data(), last, n);
    if (getenv("GGL_CHAIN_PROF")) {
        // one more run with the per-workgroup time accounting: claim / idle / tile time (100 MHz ticks) and tiles served
        DevBuf dprof;
        HIPCHK(dprof.alloc((size_t)grid * 8));
        HIPCHK(hipMemset(dprof.p, 0, (size_t)grid * 8 * sizeof(double)));
        P.prof = reinterpret_cast<long long*>(dprof.p);
        UP(dX0.p, h.data(), n);
        if (chain_persistent() <= 0) return fail(GGL_E_HIP, "k_omega_chain launch failed");
        std::vector<long long> hp((size_t)grid * 8);
        HIPCHK(hipMemcpy(hp.data(), dprof.p, hp.size() * sizeof(long long), hipMemcpyDeviceToHost));
        P.prof = nullptr;
        double cl = 0, id = 0, ti = 0, nt = 0, span = 0, cyc = 0;
        long long t0 = hp[4], t1 = hp[5];
        int perx[8] = {};
        for (int g = 0; g < grid; ++g) {
            const long long* o = hp.data() + (size_t)g * 8;
            cl += o[0]; id += o[1]; ti += o[2]; nt += o[3]; span += o[5] - o[4]; cyc += o[7];
            t0 = std::min(t0, o[4]); t1 = std::max(t1, o[5]);
            perx[o[6] & 7] += 1;
        }
        fprintf(stderr, "chain prof: kernel span %.1f us; per workgroup (avg over %d): claim %.1f us, idle %.1f us, tile %.1f us, "
                "%.2f tiles, %.1f us per tile, alive %.1f us; workgroups per XCD:", (t1 - t0) * 0.01, grid, cl * 0.01 / grid,
                id * 0.01 / grid, ti * 0.01 / grid, nt / grid, ti * 0.01 / std::max(nt, 1.0), span * 0.01 / grid);
        for (int x = 0; x < 8; ++x) fprintf(stderr, " %d", perx[x]);
        fprintf(stderr, "; clock64 ticks per us of wall_clock64 while alive: %.1f\n", cyc / (span * 0.01));
    }
    double dev = 0.0;
    for (size_t i = 0; i < n; ++i) dev = std::max(dev, std::fabs(r1[i] - r2[i]));
    out[3] = dev;
    out[2] = grid;
    std::vector<unsigned> hc(ncw + 2);
    HIPCHK(hipMemcpy(hc.data(), dcnt.p, hc.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    out[4] = (double)hc[ncw];
    for (int k = 0; k < K; ++k) out[5 + k] = (double)hc[(size_t)k * CHAIN_CNT_STRIDE];
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    float ms = 0.f;
    for (int i = 0; i < 3; ++i) chain_launches();
    HIPCHK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) chain_launches();
    HIPCHK(hipEventRecord(e1, nullptr));
    HIPCHK(hipEventSynchronize(e1));
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    out[0] = ms / iters;
    for (int i = 0; i < 3 + iters; ++i) {
        if (i == 3) HIPCHK(hipEventRecord(e0, nullptr));
        if (chain_persistent() <= 0) return fail(GGL_E_HIP, "k_omega_chain launch failed");
    }
    HIPCHK(hipEventRecord(e1, nullptr));
    HIPCHK(hipEventSynchronize(e1));
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    out[1] = ms / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    HIPCHK(hipGetLastError());
    return GGL_OK;
}

// timeline probe: one launch of variant 10; out = [nblocks][5] long long {start, loop, loop_end, end, xcc}
extern "C" int ggl_dev_symm_timeline(int K, int p, long long* out, int max_blocks, int* nblocks_out)
{
    ARGCHK(K >= 1 && p >= 1 && out && nblocks_out, "arguments");
    const size_t n = (size_t)K * p * p;
    std::vector<double> h(n, 0.25), coef((size_t)K * NS_NCOEF, 0.0);
    for (int k = 0; k < K; ++k) coef[(size_t)k * NS_NCOEF + 1] = 1.0 / p;
    const int T = (p + 63) / 64;
    const int nb = (K >= 8 ? 8 * ((K + 7) / 8) : K) * (T * (T + 1) / 2);
    ARGCHK(nb <= max_blocks, "max_blocks too small");
    DevBuf dA, dB, dC, dcoef, dT;
    HIPCHK(dA.alloc(n));
    HIPCHK(dB.alloc(n));
    HIPCHK(dC.alloc(n));
    HIPCHK(dcoef.alloc(coef.size()));
    HIPCHK(dT.alloc((size_t)nb * 5));
    UP(dA.p, h.data(), n);
    UP(dB.p, h.data(), n);
    UP(dcoef.p, coef.data(), coef.size());
    HIPCHK(hipMemset(dT.p, 0, (size_t)nb * 5 * sizeof(double)));
    for (int i = 0; i < 3; ++i) launch_symm(nullptr, dA.p, dB.p, dC.p, nullptr, nullptr, dcoef.p, K, p, 0);
    launch_symm(nullptr, dA.p, dB.p, dC.p, nullptr, nullptr, dcoef.p, K, p, 10, dT.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, dT.p, (size_t)nb * 5 * sizeof(long long), hipMemcpyDeviceToHost));
    *nblocks_out = nb;
    return GGL_OK;
}

#endif   // GGL_DEV

extern "C" int ggl_dev_ns_schedule(double l, int degrees, int max_steps, int* deg_out, double* coef_out, int* units_out)
{
    ARGCHK(deg_out && coef_out && units_out, "output pointers");
    ARGCHK(l > 0.0 && l <= 1.0, "l must be in (0,1]");
    const int n = ns_schedule_query(l, degrees, max_steps, deg_out, coef_out, units_out);
    if (n < 0) return fail(GGL_E_ARG, "no schedule within %d steps", max_steps);
    return n;
}

// host only: the grouping rule of GGL_OPT_GROUP_SCHED (ns_group_partition) and the product units of a schedule
extern "C" int ggl_dev_group_partition(const int* units, int K, int p, int max_groups, int* len_out)
{
    ARGCHK(units && len_out && K >= 1 && p >= 1 && max_groups >= 1 && max_groups <= 3, "units, len_out, K, p, 1 <= max_groups <= 3");
    return ns_group_partition(units, K, p, max_groups, len_out);
}
extern "C" int ggl_dev_ns_units(double l, int degrees, double tol) { return ns_units_query(l, degrees, tol); }

extern "C" int ggl_dev_ns_schedule_tol(double l, int degrees, double tol, int max_steps, int* deg_out, double* coef_out,
                                       int* units_out)
{
    ARGCHK(deg_out && coef_out && units_out, "output pointers");
    ARGCHK(l > 0.0 && l <= 1.0, "l must be in (0,1]");
    ARGCHK(tol >= 0.0 && tol <= 1e-6, "tol in [0, 1e-6]");
    const int n = ns_schedule_query(l, degrees, max_steps, deg_out, coef_out, units_out, std::max(tol, NS_TOL_EXACT));
    if (n < 0) return fail(GGL_E_ARG, "no schedule within %d steps", max_steps);
    return n;
}

#ifdef GGL_DEV
extern "C" int ggl_dev_coissue_probe(double* out12)
{
    ARGCHK(out12, "out");
    DevBuf d;
    HIPCHK(d.alloc((size_t)512 * 512));
    coissue_probe(nullptr, d.p, out12);
    HIPCHK(hipGetLastError());
    return GGL_OK;
}

extern "C" int ggl_dev_mfma_lds_probe(double* out6)
{
    ARGCHK(out6, "out");
    DevBuf d;
    HIPCHK(d.alloc((size_t)2048 * 256));
    mfma_lds_probe(nullptr, d.p, out6);
    HIPCHK(hipGetLastError());
    return GGL_OK;
}

extern "C" int ggl_dev_mfma_f64_peak(double* tflops_out)
{
    ARGCHK(tflops_out, "tflops_out");
    const int blocks = 256 * 8;
    DevBuf d;
    HIPCHK(d.alloc((size_t)blocks * 256));
    double best = 0.0;
    for (int r = 0; r < 3; ++r) best = std::max(best, mfma_f64_peak_tflops(nullptr, d.p, blocks, 2000, 8));
    HIPCHK(hipGetLastError());
    if (getenv("GGL_MFMA_PROBE_VERBOSE")) {
        for (int layers : {1, 2, 4, 8})
            for (int nacc : {1, 2, 4, 8})
                fprintf(stderr, "mfma f64 probe: %d wave(s)/SIMD, %d accumulators: %.1f TF/s\n", layers, nacc,
                        mfma_f64_peak_tflops(nullptr, d.p, 256 * layers, 4000, nacc));
    }
    if (getenv("GGL_MFMA_MIX_VERBOSE")) {
        for (int layers : {1, 4})
            for (int nv : {0, 2, 4, 8, 16})
                fprintf(stderr, "mfma+valu mix: %d wave(s)/SIMD, %2d VALU per MFMA: %.1f TF/s\n", layers, nv,
                        mfma_valu_mix_tflops(nullptr, d.p, 256 * layers, 4000, nv));
    }
    *tflops_out = best;
    return GGL_OK;
}
#endif   // GGL_DEV

extern "C" int ggl_eigh_batched(int K, int p, const double* A, double* D, double* Q, int eig_method)
{
    ARGCHK(D, "D");
    return eig_common(K, p, A, nullptr, D, Q, nullptr, MAP_IDENT, eig_method & 0xff);
}

extern "C" int ggl_phiplus_matrix(int K, int p, const double* beta, const double* W, double* out, int eig_method)
{
    ARGCHK(beta && out && W, "beta, W, out");
    ARGCHK(K >= 1 && p >= 1, "K, p");
    if (use_ns(eig_method & 0xff, p)) {
        // run the Omega-step of a scratch ctx with Theta = W, X = S = 0, nk = beta, rho = 1
        ggl_ctx* c = nullptr;
        int rc = ggl_ctx_create(0, K, p, (eig_method & ~0xff) | GGL_EIG_NEWTON_SCHULZ, nullptr, &c);
        if (rc) return rc;
        c->ns_tol = NS_TOL_EXACT;          // the operator-level entry point iterates to fp64 resolution
        std::vector<double> zero((size_t)K * p * p, 0.0);
        rc = ggl_set_S(c, zero.data());
        if (!rc) rc = ggl_set_state(c, zero.data(), W, nullptr, zero.data());
        if (!rc) rc = ggl_step_omega(c, 1.0, 0, beta);
        if (!rc) rc = ggl_get_state(c, out, nullptr, nullptr, nullptr);
        ggl_ctx_destroy(c);
        return rc;
    }
    return eig_common(K, p, W, beta, nullptr, nullptr, out, MAP_PHIPLUS, eig_method & 0xff);
}

static int rank_matrix_impl(int K, int p, const double* beta, const double* C, double* out, int eig_method, double l0_coarse,
                            double l0_deflate, long long* stats, int nstats);

// l0_coarse >= 0: the two-tier iteration at that first-pass resolution, WITHOUT the deflation (0: one tier); < 0: the ctx
// defaults (deflation after a first pass at GGL_OPT_RANK_L0_DEFLATE)
extern "C" int ggl_rank_matrix_ex(int K, int p, const double* beta, const double* C, double* out, int eig_method,
                                  double l0_coarse, long long stats[6])
{
    return rank_matrix_impl(K, p, beta, C, out, eig_method, l0_coarse, l0_coarse >= 0.0 ? 0.0 : -1.0, stats, 6);
}

// the deflating L-step with its first-pass resolution exposed (<= 0: the default); stats[8] = the six of ggl_rank_matrix_ex +
// { calls followed by the deflation, instances that had directions to deflate }
extern "C" int ggl_rank_matrix_deflate(int K, int p, const double* beta, const double* C, double* out, int eig_method,
                                       double l0_deflate, long long stats[8])
{
    return rank_matrix_impl(K, p, beta, C, out, eig_method, -1.0, l0_deflate > 0.0 ? l0_deflate : -1.0, stats, 8);
}

static int rank_matrix_impl(int K, int p, const double* beta, const double* C, double* out, int eig_method, double l0_coarse,
                            double l0_deflate, long long* stats, int nstats)
{
    ARGCHK(beta && out && C, "beta, C, out");
    ARGCHK(K >= 1 && p >= 1, "K, p");
    if (stats) for (int i = 0; i < nstats; ++i) stats[i] = 0;
    if (use_ns(eig_method & 0xff, p)) {
        // the L-step of a scratch ctx: C into the work stack, beta into the mu/rho parameter slot
        ggl_ctx* c = nullptr;
        int rc = ggl_ctx_create(0, K, p, (eig_method & ~0xff) | GGL_EIG_NEWTON_SCHULZ, nullptr, &c);
        if (rc) return rc;
        if (l0_coarse >= 0.0) c->rank_l0_coarse = l0_coarse;
        if (l0_deflate == 0.0) c->rank_deflate = false;
        else if (l0_deflate > 0.0) c->rank_l0_deflate = l0_deflate;
        hipError_t e = hipMemcpyAsync(c->W, C, c->n * sizeof(double), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            rc = upload_par(c, 2, beta, 0.0, 1.0);
            if (!rc) rc = rank_step(c);
            if (!rc) e = hipMemcpyAsync(out, c->L, c->n * sizeof(double), hipMemcpyDeviceToHost, c->stream);
            if (!rc && e == hipSuccess) e = hipStreamSynchronize(c->stream);
        }
        if (stats) {
            stats[0] = c->rank_calls; stats[1] = c->rank_continued; stats[2] = c->rank_cont_instances;
            stats[3] = c->rank_fallbacks; stats[4] = c->rank_retries; stats[5] = c->rank_launches;
            if (nstats >= 8) { stats[6] = c->rank_deflated_calls; stats[7] = c->rank_deflated_instances; }
        }
        ggl_ctx_destroy(c);
        if (e != hipSuccess) return fail(GGL_E_HIP, "ggl_rank_matrix: %s", hipGetErrorString(e));
        return rc;
    }
    return eig_common(K, p, C, beta, nullptr, nullptr, out, MAP_RANK, eig_method & 0xff);
}

extern "C" int ggl_rank_matrix(int K, int p, const double* beta, const double* C, double* out, int eig_method)
{
    return ggl_rank_matrix_ex(K, p, beta, C, out, eig_method, -1.0, nullptr);
}

static int recon_common(int K, int p, const double* beta, const double* D, const double* Q, double* out, int map)
{
    ARGCHK(K >= 1 && p >= 1 && beta && D && Q && out, "arguments");
    const size_t n = (size_t)K * p * p, kp = (size_t)K * p;
    // Q has eigenvectors in columns; the kernel wants them in rows
    std::vector<double> R(n);
    for (int k = 0; k < K; ++k)
        for (int i = 0; i < p; ++i)
            for (int m = 0; m < p; ++m) R[(size_t)k * p * p + (size_t)m * p + i] = Q[(size_t)k * p * p + (size_t)i * p + m];
    DevBuf dR, dD, dB, dO, dS;
    HIPCHK(dR.alloc(n));
    HIPCHK(dD.alloc(kp));
    HIPCHK(dB.alloc(K));
    HIPCHK(dO.alloc(n));
    HIPCHK(dS.alloc(2 * kp));
    UP(dR.p, R.data(), n);
    UP(dD.p, D, kp);
    UP(dB.p, beta, K);
    launch_recon(nullptr, dO.p, dR.p, dD.p, dB.p, map, K, p, dS.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    DOWN(out, dO.p, n);
    return GGL_OK;
}

extern "C" int ggl_phiplus(int K, int p, const double* beta, const double* D, const double* Q, double* out)
{
    return recon_common(K, p, beta, D, Q, out, MAP_PHIPLUS);
}

extern "C" int ggl_prox_rank_norm(int K, int p, const double* beta, const double* D, const double* Q, double* out)
{
    return recon_common(K, p, beta, D, Q, out, MAP_RANK);
}

extern "C" int ggl_prox_od_1norm(int p, const double* A, double lam, const double* lam_pp, double* out)
{
    ARGCHK(p >= 1 && A && out, "arguments");
    const size_t n = (size_t)p * p;
    DevBuf dA, dM, dO;
    HIPCHK(dA.alloc(n));
    HIPCHK(dO.alloc(n));
    UP(dA.p, A, n);
    if (lam_pp) { HIPCHK(dM.alloc(n)); UP(dM.p, lam_pp, n); }
    launch_prox_od(nullptr, dO.p, dA.p, lam, lam_pp ? dM.p : nullptr, p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    DOWN(out, dO.p, n);
    return GGL_OK;
}

extern "C" int ggl_prox_p(int K, int p, const double* X, double l1, double l2, int reg, double* out)
{
    ARGCHK(K >= 1 && p >= 1 && X && out, "arguments");
    ARGCHK(reg == GGL_REG_GGL || reg == GGL_REG_FGL, "reg");
    ARGCHK(l1 > 0 && l2 > 0, "lambda 1 and lambda2 have to be positive");
    if (reg == GGL_REG_FGL && K > fgl_max_K())
        return fail(GGL_E_ARG, "prox_p (FGL): K = %d exceeds the %d instances the LDS scan buffer holds", K, fgl_max_K());
    const size_t n = (size_t)K * p * p;
    DevBuf dX, dO, dW;
    HIPCHK(dX.alloc(n));
    HIPCHK(dO.alloc(n));
    HIPCHK(dW.alloc((size_t)ggl_chunks(K, p) * p * p));
    UP(dX.p, X, n);
    HIPCHK(launch_prox_p(nullptr, reg, dO.p, dX.p, l1, l2, K, p, dW.p));
    HIPCHK(hipDeviceSynchronize());
    DOWN(out, dO.p, n);
    return GGL_OK;
}

static int vec_common(int mode, int n, int K, const double* Y, double l1, double l2, double* out)
{
    ARGCHK(n >= 1 && K >= 1 && Y && out, "arguments");
    const size_t tot = (size_t)n * K;
    DevBuf dY, dO;
    HIPCHK(dY.alloc(tot));
    HIPCHK(dO.alloc(tot));
    UP(dY.p, Y, tot);
    HIPCHK(launch_vec_prox(nullptr, mode, dY.p, dO.p, n, K, l1, l2));
    HIPCHK(hipDeviceSynchronize());
    DOWN(out, dO.p, tot);
    return GGL_OK;
}

extern "C" int ggl_prox_tv(int n, int K, const double* Y, double lam, double* out)
{
    return vec_common(0, n, K, Y, lam, 0.0, out);
}

extern "C" int ggl_prox_2norm(int n, int K, const double* Y, double lam, double* out)
{
    return vec_common(1, n, K, Y, lam, 0.0, out);
}

extern "C" int ggl_prox_phi(int n, int K, const double* Y, double l1, double l2, int reg, double* out)
{
    ARGCHK(reg == GGL_REG_GGL || reg == GGL_REG_FGL, "reg");
    return vec_common(reg == GGL_REG_GGL ? 2 : 3, n, K, Y, l1, l2, out);
}
