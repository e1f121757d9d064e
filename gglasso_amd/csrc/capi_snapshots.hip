// Part of the C ABI of libggl_hip.so (include/ggl_hip.h); see capi_internal.hpp for the map of the translation units.
#include "capi_internal.hpp"

// Snapshots of the instances ks[0..n) of `src` into the slots kd[0..n) of `c` (c == src, kd == ks: ggl_snapshot_k).  The two-ctx
// form serves a batch that was compacted (ggl_ctx_create_subset): a point that converges in the smaller ctx is snapshotted into
// the ORIGINAL ctx at its original index, where the selection statistics and ggl_finalize_L run over all points at once.
// with_state: Omega and X as well (ggl_snapshot_state_from).  ONE wait for src and one for c whatever n (the C loop hands over
// all points that finish in an iteration together; round 5 synchronised both streams per instance -- ADVICE r5).
int snapshot_many(ggl_ctx* c, const int* kd, ggl_ctx* src, const int* ks, int n, bool with_state)
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

