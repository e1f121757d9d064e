// Part of the C ABI of libggl_hip.so (include/ggl_hip.h); see capi_internal.hpp for the map of the translation units.
#include "capi_internal.hpp"

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

