// Part of the C ABI of libggl_hip.so (include/ggl_hip.h); see capi_internal.hpp for the map of the translation units.
#include "capi_internal.hpp"

// ---- K independent single problems with their own rho / lambda1 (batched lambda path) ----------
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

int sgl_batch_step_impl(ggl_ctx* c, const double* rho, const double* lambda1, int latent, const double* mu1,
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

int ensure_partials(ggl_ctx* c, size_t need)
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

int mgl_batch_step_impl(ggl_ctx* c, int G, const double* rho, const double* lambda1, const double* lambda2,
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

