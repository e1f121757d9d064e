// Part of the C ABI of libggl_hip.so (include/ggl_hip.h); see capi_internal.hpp for the map of the translation units.
#include "capi_internal.hpp"

// ---------------------------------------------------------------------------------------------
// eigen-decomposition + eigenvalue map + reconstruction of a device stack (in: A, destroyed when
// the rocSOLVER path is taken; out may alias nothing).  Dv receives the eigenvalues.
// ---------------------------------------------------------------------------------------------
int eig_recon(ggl_ctx* c, double* A, double* out, double* Dv, int map, const double* betaK, int ph_eig,
                     int ph_recon)
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

int eigvals_only(ggl_ctx* c, double* A, double* Dv)
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
void mark_failed(ggl_ctx* c, int k, int why, double value)
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
void sanitize_bounds(ggl_ctx* c, double* b, const double* repl, double repl_scale, double repl_scalar)
{
    if (!c->isolate) return;
    for (int k = 0; k < c->K; ++k)
        if (!std::isfinite(b[k]) || !(b[k] > 0.0)) {
            mark_failed(c, k, 1, b[k]);
            b[k] = repl ? repl_scale * repl[k] : repl_scalar;
        }
}

int check_info(ggl_ctx* c, const char* what)
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
int upload_par(ggl_ctx* c, int slot, const double* vals, double scalar, double div, CopySegs* pending)
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
int probe_part_streams(ggl_ctx* c)
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
int lds_table(ggl_ctx* c)
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

void lds_missed(ggl_ctx* c)
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
int trace_lane(const ggl_ctx* c, hipStream_t st)
{
    if (st == c->stream) return 0;
    for (int i = 0; i < ggl_ctx::MAX_PARTS - 1; ++i)
        if (st == c->streamx[i]) return i + 1;
    return -1;
}
void trace_mark(ggl_ctx* c, hipStream_t st, int tag)
{
    ggl_ctx::Trace& t = c->trace;
    if (!t.on || t.n >= t.cap) return;
    if (hipEventRecord(t.ev[t.n], st) != hipSuccess) return;
    t.tag[t.n] = tag;
    t.lane[t.n] = trace_lane(c, st);
    t.n += 1;
}
void trace_host(ggl_ctx* c, int tag)
{
    ggl_ctx::Trace& t = c->trace;
    if (!t.on || t.nhost >= t.cap) return;
    t.host_us[t.nhost] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t.t0).count();
    t.host_tag[t.nhost] = tag;
    t.nhost += 1;
}
void trace_symm_hook(hipStream_t st, int kind, void* arg) { trace_mark((ggl_ctx*)arg, st, 10 + kind); }

// GGL_OPT_GROUP_SCHED: contiguous groups of a batch whose instances need different product counts (ns_group_partition).
// cb[k] >= lambda_max(A'_k), beta_k = nk/rho.  Returns the number of groups (1: the batch stays whole, Kh / k0h untouched).
static int omega_groups(const ggl_ctx* c, const double* cb, const double* beta_h, int K, int* Kh, int* k0h, int* gunits)
{
    if (!c->group_sched || K < 2 || c->ns_force == 2 || c->chain_mode || c->comm) return 1;
    {
        // the K planner queries below cost ~0.1 us each on the host, in front of the chain's launches: only where a step is
        // long enough not to notice (a step is >= 7 launches of F + K I seconds; same constants as ns_group_partition)
        const double I = 2.5e-14 * (double)c->p * c->p * c->p, F = 6.5e-6;
        if (c->group_sched < 10 && (double)K * 1e-7 > 0.01 * 7.0 * (F + K * I)) return 1;
    }
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
int omega_step(ggl_ctx* c, int latent, CopySegs* pending, bool allow_spec, bool only_spec)
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
        // (the instance the WHOLE batch would take -- the groups share the chip -- with three DMA stages where that is the
        // 64x64 kernel, as for any concurrent parts)
        const int var_grouped = c->symm_variant >= 0 ? c->symm_variant : (symm_auto_variant(K, c->p) == 16 ? 17 : symm_auto_variant(K, c->p));
        const int var_parts = grouped ? var_grouped
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
                var_b = var_grouped;
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
int validate_spec(ggl_ctx* c)
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

int finish_norms(ggl_ctx* c, int rows, double* out_norms, int group)
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


extern "C" int ggl_step_finish(ggl_ctx* c, double rho, double lambda1, double lambda2, int reg, int latent,
                               const double* mu1, int groupsq_ready, double out_norms[5])
{
    ARGCHK(c, "ctx");
    DROP_PRE(c);
    return ggl_step_finish_impl(c, rho, lambda1, lambda2, reg, latent, mu1, groupsq_ready, out_norms);
}

int ggl_step_finish_impl(ggl_ctx* c, double rho, double lambda1, double lambda2, int reg, int latent,
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
bool early_wanted(const ggl_ctx* c)
{
    if (!c->early_caller || !c->early_part || !c->pipeline || !c->omega_ns || !c->spec_enable || c->prof_on == 1 || c->last_step_hint || !c->ratio_calm ||
        c->chain_mode || c->pre_valid)
        return false;
    if (c->lds_omega && c->p <= omega_lds_max_p()) return false;       // (one kernel writes Omega there: nothing to split)
    return true;
}

int maybe_early(ggl_ctx* c)
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
bool take_prelaunched(ggl_ctx* c, int latent)
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
int maybe_prelaunch(ggl_ctx* c, double rho, const double out_norms[5])
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

