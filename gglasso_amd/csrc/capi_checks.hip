// Part of the C ABI of libggl_hip.so (include/ggl_hip.h); see capi_internal.hpp for the map of the translation units.
#include "capi_internal.hpp"

// ---------------------------------------------------------------------------------------------
// exit checks / objective / kkt
// ---------------------------------------------------------------------------------------------
int host_reduce(ggl_ctx* c, int rows, int nv, double* out /*nv*/, bool take_max)
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

