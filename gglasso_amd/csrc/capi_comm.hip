// Part of the C ABI of libggl_hip.so (include/ggl_hip.h); see capi_internal.hpp for the map of the translation units.
#include "capi_internal.hpp"

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

