// Part of the C ABI of libggl_hip.so (include/ggl_hip.h); see capi_internal.hpp for the map of the translation units.
#include "capi_internal.hpp"

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
// ggl_finalize_L: out = { eigendecompositions run, of those repeated because the eigenvalues did not add up to trace(C) }
extern "C" int ggl_finalize_stats(ggl_ctx* c, long long out[2])
{
    ARGCHK(c && out, "ctx, out");
    out[0] = c->finalize_calls;
    out[1] = c->finalize_retries;
    return GGL_OK;
}

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

