// Part of the C ABI of libggl_hip.so (include/ggl_hip.h); see capi_internal.hpp for the map of the translation units.
#include "capi_internal.hpp"

// L = (C - mu I)_+ with C in c->W and mu_k/rho in parameter slot 2 (pinned mirror par_h + 2K).
// Sign Newton-Schulz with a-posteriori verification; retries at a finer resolution, then falls back to the
// eigendecomposition, so the result always meets the eigh route's accuracy.

int rank_step(ggl_ctx* c)
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

int rank_step_impl(ggl_ctx* c)
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
    // The ONE eigendecomposition a solve's returned L (and its rank) rests on is checked: trace(C_k) = the sum of the eigenvalues,
    // taken before the eigensolver overwrites C_k.  An eigensolver that returns something else (reading (B) of round 5's
    // intermittent RANK table) is run a second time on a kept copy of C; if that fails too the instance is
    // marked / the call fails -- never a count that reads like a result.
    std::vector<double> tr(K), d(kp);
    launch_trace(c->stream, Csrc, K, p, 0.0, c->norms);
    HIPCHK(hipMemcpyAsync(tr.data(), c->norms, K * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    double* keep = c->nsT;                                 // (scratch of the matrix-function steps: free between steps)
    if (keep) launch_copy_block(c->stream, keep, Csrc, c->n);
    std::vector<unsigned char> wrong(K, 0);
    for (int attempt = 0; attempt < 2; ++attempt) {
        int rc = eig_recon(c, Csrc, out, c->DvL, MAP_RANK, c->par + 2 * (size_t)K);      // (destroys Csrc)
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(d.data(), c->DvL, kp * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(c->info_h, c->info, K * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        bool any_wrong = false;
        for (int k = 0; k < K; ++k) {
            wrong[k] = 0;
            if (!todo[k]) continue;
            double sum = 0.0, mag = 0.0;
            for (int e = 0; e < p; ++e) { sum += d[(size_t)k * p + e]; mag += std::fabs(d[(size_t)k * p + e]); }
            if (!std::isfinite(sum) || !std::isfinite(tr[k]) || !(std::fabs(sum - tr[k]) <= 1e-8 * std::max(mag, 1e-300))) {
                wrong[k] = 1;
                any_wrong = true;
            }
        }
        if (!any_wrong || attempt == 1 || !keep) break;
        c->finalize_retries += 1;
        launch_copy_block(c->stream, Csrc, keep, c->n);
    }
    if (which == 1)
        for (int k = 0; k < K; ++k)
            if (todo[k]) launch_copy_block(c->stream, c->snapL + k * pp, c->W + k * pp, pp);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(slot, saved.data(), K * sizeof(double));
    HIPCHK(hipMemcpyAsync(c->par + 2 * (size_t)K, slot, K * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    int rc = check_info(c, "final L");
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
        // eigenvalues that are not finite or do not add up to trace(C_k): the kept C was not finite (a diverged instance) or the
        // eigensolver failed twice -- never a rank of zero that reads like a result
        if (!finite || wrong[k]) {
            if (c->isolate) { mark_failed(c, k, 2, NAN); r = -1; }
            else return fail(GGL_E_SOLVER, "final L: the eigenvalues of instance %d's L-step input are not finite or do not add up "
                             "to its trace", k);
        }
        if (rank_out) rank_out[k] = r;
    }
    if (which == 0) c->l_ns = false;                  // L is an eigendecomposition's now (and Ckeep is spent)
    else memset(c->snap_ns, 0, K);                    // (snapC is spent; a later snapshot of an instance sets its flag again)
    if (which == 1) launch_copy_block(c->stream, c->snapC, nullptr, c->n);
    c->finalize_calls += 1;
    return n_todo;
}

