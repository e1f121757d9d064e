// Part of the C ABI of libggl_hip.so (include/ggl_hip.h); see capi_internal.hpp for the map of the translation units.
#include "capi_internal.hpp"

static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

void prof_collect(ggl_ctx* c)   // call after a stream sync
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
int blas_handle(ggl_ctx* c, rocblas_handle* out)
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

bool use_jacobi(const ggl_ctx* c)
{
    if (c->eig == GGL_EIG_JACOBI) return true;
    if (c->eig == GGL_EIG_ROCSOLVER) return false;
    return jacobi_fits(c->p);   // AUTO and NEWTON_SCHULZ: eigenvalue consumers use Jacobi when it fits
}

bool use_ns(int eig, int p)
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
int poison_fill() { return g_poison; }
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
int drop_prelaunch(ggl_ctx* c)
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
    ARGCHK(period >= 0 && period <= c->K, "period: 0 (all K) or a divisor of K");
    if (period == 0 || period == c->K) {
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
int download_stacks(ggl_ctx* c, const std::vector<Xfer>& xs)
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

