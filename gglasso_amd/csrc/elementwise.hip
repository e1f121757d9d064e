// HBM-streaming kernels of the ADMM iteration: W formation, SGL Theta-step, dual update,
// stopping-test norms, small utilities.  grid = (chunks of p*p, K); every block writes its
// five partial sums to a fixed slot so the reduction order (and the result) is deterministic.
#include <algorithm>
#include "common.hpp"
#include "kernels.hpp"

namespace ggl {

static constexpr int EW_THREADS = 256;
static constexpr int EW_EPT = 4;   // elements per thread
static constexpr int EW_CHUNK = EW_THREADS * EW_EPT;

int elementwise_blocks(int p)
{
    size_t pp = (size_t)p * p;
    return (int)((pp + EW_CHUNK - 1) / EW_CHUNK);
}

// ---------------------------------------------------------------------------------------------
template <bool HAS_L>
__global__ __launch_bounds__(EW_THREADS) void k_form_W(double* __restrict__ W, const double* __restrict__ Theta,
                                                       const double* __restrict__ L, const double* __restrict__ X,
                                                       const double* __restrict__ S, const double* __restrict__ betaK,
                                                       size_t pp)
{
    const int k = blockIdx.y;
    const double beta = betaK[k];
    const size_t base = (size_t)k * pp;
    size_t i = (size_t)blockIdx.x * EW_CHUNK + threadIdx.x;
#pragma unroll
    for (int e = 0; e < EW_EPT; ++e, i += EW_THREADS) {
        if (i < pp) {
            double t = Theta[base + i];
            if (HAS_L) t -= L[base + i];
            W[base + i] = (t - X[base + i]) - beta * S[base + i];
        }
    }
}

void launch_form_W(hipStream_t st, double* W, const double* Theta, const double* L, const double* X,
                   const double* S, const double* betaK, int K, int p)
{
    size_t pp = (size_t)p * p;
    dim3 grid(elementwise_blocks(p), K);
    if (L)
        hipLaunchKernelGGL(k_form_W<true>, grid, dim3(EW_THREADS), 0, st, W, Theta, L, X, S, betaK, pp);
    else
        hipLaunchKernelGGL(k_form_W<false>, grid, dim3(EW_THREADS), 0, st, W, Theta, L, X, S, betaK, pp);
}

// ---------------------------------------------------------------------------------------------
// SGL Theta-step.  single_admm_solver.py:169 (Theta), :178 (X), :277-291 (norms).
template <bool LATENT, bool MASK>
__global__ __launch_bounds__(EW_THREADS) void k_theta_sgl(double* __restrict__ Theta, double* __restrict__ X,
                                                          double* __restrict__ C, const double* __restrict__ Omega,
                                                          const double* __restrict__ OmegaPrev,
                                                          const double* __restrict__ L, const double* __restrict__ l1K,
                                                          const double* __restrict__ mask,
                                                          const double* __restrict__ invrhoK,
                                                          double* __restrict__ partials, int p,
                                                          const int* __restrict__ skip, const int* __restrict__ pk,
                                                          size_t mask_stride)
{
    // pk != null: instance k is the leading (pk[k], pk[k]) block of its slot (identity padding behind it, a fixed point of
    // the iteration): the stopping-test sums run over that block only (single_admm_solver.py:277-291 on the block itself)
    // mask_stride: 0 = one (p,p) threshold array for all instances, p*p = one per instance
    __shared__ double scratch[GGL_NNORM * (EW_THREADS / 64)];
    if (spec_failed(skip)) return;
    const int k = blockIdx.y;
    const size_t pp = (size_t)p * p;
    const size_t base = (size_t)k * pp;
    const int pin = pk ? pk[k] : p;
    if (MASK) mask += (size_t)k * mask_stride;
    const double lk = MASK ? 0.0 : l1K[k];
    const double inv_rho = MASK ? invrhoK[k] : 0.0;
    double acc[GGL_NNORM] = {0, 0, 0, 0, 0};
    size_t i = (size_t)blockIdx.x * EW_CHUNK + threadIdx.x;
#pragma unroll
    for (int e = 0; e < EW_EPT; ++e, i += EW_THREADS) {
        if (i < pp) {
            const int r = (int)(i / p), c = (int)(i - (size_t)r * p);
            const double om = Omega[base + i];
            const double x = X[base + i];
            const double l = LATENT ? L[base + i] : 0.0;
            const double v = (om + l) + x;
            const double thr = MASK ? inv_rho * mask[i] : lk;
            const double th = (r == c) ? v : soft(v, thr);
            Theta[base + i] = th;
            if (LATENT) {
                C[base + i] = (th - x) - om;
            } else {
                const double xn = (x + om) - th;   // single_admm_solver.py:178
                X[base + i] = xn;
                const double dp = om - OmegaPrev[base + i];
                if (r < pin && c < pin) {
                    acc[0] += om * om;
                    acc[1] += th * th;
                    acc[2] += xn * xn;
                    acc[3] += (om - th) * (om - th);
                    acc[4] += dp * dp;
                }
            }
        }
    }
    if (!LATENT) {
        block_sum<GGL_NNORM>(acc, scratch);
        if (threadIdx.x == 0) {
            double* o = partials + ((size_t)k * gridDim.x + blockIdx.x) * GGL_NNORM;
#pragma unroll
            for (int v = 0; v < GGL_NNORM; ++v) o[v] = acc[v];
        }
    }
}

void launch_theta_sgl(hipStream_t st, double* Theta, double* X, double* C, const double* Omega,
                      const double* OmegaPrev, const double* L, const double* l1K, const double* mask,
                      const double* invrhoK, int latent, double* partials, int K, int p, const int* skip, const int* pk,
                      size_t mask_stride)
{
    dim3 grid(elementwise_blocks(p), K), blk(EW_THREADS);
#define GGL_TS(LAT, MSK)                                                                              \
    hipLaunchKernelGGL((k_theta_sgl<LAT, MSK>), grid, blk, 0, st, Theta, X, C, Omega, OmegaPrev, L, l1K, \
                       mask, invrhoK, partials, p, skip, pk, mask_stride)
    if (latent) {
        if (mask) GGL_TS(true, true); else GGL_TS(true, false);
    } else {
        if (mask) GGL_TS(false, true); else GGL_TS(false, false);
    }
#undef GGL_TS
}

// ---------------------------------------------------------------------------------------------
// X += (Omega - Theta) + L ; norms (admm_solver.py:208, 316-331)
template <bool HAS_L>
__global__ __launch_bounds__(EW_THREADS) void k_dual_update(double* __restrict__ X, const double* __restrict__ Omega,
                                                            const double* __restrict__ OmegaPrev,
                                                            const double* __restrict__ Theta,
                                                            const double* __restrict__ L,
                                                            double* __restrict__ partials, size_t pp)
{
    __shared__ double scratch[GGL_NNORM * (EW_THREADS / 64)];
    const int k = blockIdx.y;
    const size_t base = (size_t)k * pp;
    double acc[GGL_NNORM] = {0, 0, 0, 0, 0};
    size_t i = (size_t)blockIdx.x * EW_CHUNK + threadIdx.x;
#pragma unroll
    for (int e = 0; e < EW_EPT; ++e, i += EW_THREADS) {
        if (i < pp) {
            const double om = Omega[base + i], th = Theta[base + i];
            const double l = HAS_L ? L[base + i] : 0.0;
            const double res = (om - th) + l;
            const double xn = X[base + i] + res;
            X[base + i] = xn;
            const double dp = om - OmegaPrev[base + i];
            acc[0] += om * om;
            acc[1] += (th - l) * (th - l);
            acc[2] += xn * xn;
            acc[3] += res * res;
            acc[4] += dp * dp;
        }
    }
    block_sum<GGL_NNORM>(acc, scratch);
    if (threadIdx.x == 0) {
        double* o = partials + ((size_t)k * gridDim.x + blockIdx.x) * GGL_NNORM;
#pragma unroll
        for (int v = 0; v < GGL_NNORM; ++v) o[v] = acc[v];
    }
}

void launch_dual_update(hipStream_t st, double* X, const double* Omega, const double* OmegaPrev,
                        const double* Theta, const double* L, double* partials, int K, int p)
{
    size_t pp = (size_t)p * p;
    dim3 grid(elementwise_blocks(p), K), blk(EW_THREADS);
    if (L)
        hipLaunchKernelGGL(k_dual_update<true>, grid, blk, 0, st, X, Omega, OmegaPrev, Theta, L, partials, pp);
    else
        hipLaunchKernelGGL(k_dual_update<false>, grid, blk, 0, st, X, Omega, OmegaPrev, Theta, L, partials, pp);
}

// ---------------------------------------------------------------------------------------------
// out[k][v] = sum_b partials[k][b][v]; one block per k, all nv (<= 8) sums in ONE pass over the partials: strided
// per-thread sums, wave shuffles, four wave results added in a fixed order (deterministic).
__global__ __launch_bounds__(256) void k_reduce_partials(const double* __restrict__ partials, int nblk, int nv,
                                                         double* __restrict__ out, unsigned long long* seq,
                                                         unsigned long long seq_val, unsigned* __restrict__ arrive)
{
    __shared__ double sh[4][8];
    const int k = blockIdx.x;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const double* base = partials + (size_t)k * nblk * nv;
    // four rows per trip, all their loads in flight before the first add (one workgroup walks up to ~2000 rows: with one row
    // per trip the loop paid a global-memory latency per row, 10.5 us at the headline); the adds keep the ascending order
    for (int b = threadIdx.x; b < nblk; b += 1024) {
        double t[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int bb = b + 256 * u;
#pragma unroll
            for (int v = 0; v < 8; ++v) t[u][v] = (v < nv && bb < nblk) ? base[(size_t)bb * nv + v] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (b + 256 * u < nblk) {
#pragma unroll
                for (int v = 0; v < 8; ++v)
                    if (v < nv) acc[v] += t[u][v];
            }
        }
    }
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        if (v < nv) {
            const double s = wave_sum(acc[v]);
            if (lane == 0) sh[wid][v] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x < nv) out[(size_t)k * nv + threadIdx.x] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
    // single-row reductions into pinned host memory: publish a sequence number AFTER the sums, so that the host can
    // wait for this kernel by polling one word instead of a stream synchronisation
    if (seq != nullptr && gridDim.x == 1) {
        // (the row carries the sequence number itself, behind the sums: the host's second check, finish_norms)
        if (threadIdx.x == nv) out[nv] = (double)seq_val;
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence_system();
            *(volatile unsigned long long*)seq = seq_val;
        }
    } else if (seq != nullptr) {
        // several rows (a batch of independent problems): every workgroup makes its row visible, then counts itself in; the
        // last one to arrive publishes the sequence number and leaves the counter at zero for the next launch
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence_system();
            const unsigned prev = atomicAdd(arrive, 1u);
            if (prev == gridDim.x - 1) {
                *arrive = 0u;
                __threadfence_system();
                *(volatile unsigned long long*)seq = seq_val;
            }
        }
    }
}

void launch_reduce_partials(hipStream_t st, const double* partials, int K, int nblk, int nv, double* out,
                            unsigned long long* seq, unsigned long long seq_val, unsigned* arrive)
{
    // seq with K > 1 rows needs the arrival counter (device memory, zero between launches)
    hipLaunchKernelGGL(k_reduce_partials, dim3(K), dim3(256), 0, st, partials, nblk, nv, out,
                       (K == 1 || arrive) ? seq : nullptr, seq_val, arrive);
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_scale(double* __restrict__ X, double f, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) X[i] *= f;
}

__global__ __launch_bounds__(EW_THREADS) void k_scale_batch(double* __restrict__ X, const double* __restrict__ fK,
                                                            size_t pp)
{
    const int k = blockIdx.y;
    const double f = fK[k];
    if (f == 1.0) return;
    const size_t base = (size_t)k * pp;
    size_t i = (size_t)blockIdx.x * EW_CHUNK + threadIdx.x;
#pragma unroll
    for (int e = 0; e < EW_EPT; ++e, i += EW_THREADS)
        if (i < pp) X[base + i] *= f;
}

void launch_scale_batch(hipStream_t st, double* X, const double* fK, int K, int p)
{
    size_t pp = (size_t)p * p;
    hipLaunchKernelGGL(k_scale_batch, dim3(elementwise_blocks(p), K), dim3(EW_THREADS), 0, st, X, fK, pp);
}

__global__ __launch_bounds__(256) void k_copy_small(CopySegs sg)
{
    const int s = blockIdx.y;
    if (s >= sg.n) return;
    unsigned* d = (unsigned*)sg.dst[s];
    const unsigned* src = (const unsigned*)sg.src[s];
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < sg.words[s]; i += gridDim.x * 256) d[i] = src ? src[i] : 0u;
}

// Join of concurrent launch sequences WITHOUT a cross-queue event (GGL_OPT_JOIN_FLAG).  An event wait between two hardware
// queues costs ~25 us of idle time on the waiting queue after the event has fired (tools/event_timeline.py: the last product
// of the second part done at 649 us, the Theta kernel behind the wait done at 740 us, the kernel itself 57 us).  Instead the
// part's stream ends with k_set_flag (an agent-scope store of the join's sequence number, performed at the memory side: the
// kernel before it has completed, end-of-kernel release included), and the main stream carries k_wait_flags -- ONE wave that
// polls the words and exits -- in front of the kernel that needs the parts' results: the dependency becomes an ordinary
// in-queue one.  A wave that waits longer than the time-out (the two streams share a hardware queue after all, or the other
// part died) raises the step's validation flag: the kernels behind it leave the iterate alone and the host repeats the step
// on the synchronising route.
__global__ void k_set_flag(unsigned long long* f, unsigned long long v)
{
    __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void k_wait_flags(const unsigned long long* f, int n, unsigned long long v, int* skip, int* skip_host, int slot,
                             long long timeout_ticks)
{
    if ((int)threadIdx.x >= n) return;
    const long long t0 = (long long)wall_clock64();
    while (__hip_atomic_load(f + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v) {
        __builtin_amdgcn_s_sleep(4);
        if ((long long)wall_clock64() - t0 > timeout_ticks) {
            atomicOr(skip + slot, 1);
            skip_host[slot] = 1;
            break;
        }
    }
}

void launch_set_flag(hipStream_t st, unsigned long long* f, unsigned long long v)
{
    hipLaunchKernelGGL(k_set_flag, dim3(1), dim3(1), 0, st, f, v);
}

void launch_wait_flags(hipStream_t st, const unsigned long long* f, int n, unsigned long long v, int* skip, int* skip_host,
                       int slot, double timeout_ms)
{
    hipLaunchKernelGGL(k_wait_flags, dim3(1), dim3(64), 0, st, f, n, v, skip, skip_host, slot, (long long)(timeout_ms * 1e5));
}

// one wave that does nothing for `us` microseconds (100 MHz wall clock): the stream-concurrency probe of capi_omega.hip
__global__ void k_spin_us(long long us)
{
    const long long t0 = (long long)wall_clock64();
    while ((long long)wall_clock64() - t0 < us * 100) __builtin_amdgcn_s_sleep(32);
}

void launch_spin_us(hipStream_t st, int us) { hipLaunchKernelGGL(k_spin_us, dim3(1), dim3(64), 0, st, (long long)us); }

// the same by ONE workgroup, which then publishes a sequence number in (coherent) pinned memory: the host waits for the
// copies by polling that word instead of a stream synchronisation (finish_norms of a K-sharded step)
__global__ __launch_bounds__(256) void k_copy_small_seq(CopySegs sg, unsigned long long* seq, unsigned long long seq_val)
{
    for (int s = 0; s < sg.n; ++s) {
        unsigned* d = (unsigned*)sg.dst[s];
        const unsigned* src = (const unsigned*)sg.src[s];
        for (unsigned i = threadIdx.x; i < sg.words[s]; i += 256) d[i] = src ? src[i] : 0u;
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        *(volatile unsigned long long*)seq = seq_val;
    }
}

void launch_copy_small_seq(hipStream_t st, const CopySegs& segs, unsigned long long* seq, unsigned long long seq_val)
{
    hipLaunchKernelGGL(k_copy_small_seq, dim3(1), dim3(256), 0, st, segs, seq, seq_val);
}

void launch_copy_small(hipStream_t st, const CopySegs& segs)
{
    if (segs.n == 0) return;
    unsigned mx = 1;
    for (int i = 0; i < segs.n; ++i) mx = std::max(mx, segs.words[i]);
    const unsigned bx = std::min((mx + 1023u) / 1024u, 64u);
    hipLaunchKernelGGL(k_copy_small, dim3(bx, segs.n), dim3(256), 0, st, segs);
}

void launch_scale(hipStream_t st, double* X, double f, size_t n)
{
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_scale, dim3(blocks), dim3(256), 0, st, X, f, n);
}

__global__ __launch_bounds__(256) void k_sub(double* __restrict__ D, const double* __restrict__ A,
                                             const double* __restrict__ B, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        D[i] = B ? A[i] - B[i] : A[i];
}

void launch_sub(hipStream_t st, double* D, const double* A, const double* B, size_t n)
{
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_sub, dim3(blocks), dim3(256), 0, st, D, A, B, n);
}

// out[k] = max |A[k,i,j] - A[k,j,i]|   (exit check admm_solver.py:284-291; runs once per solve)
__global__ __launch_bounds__(256) void k_asym_max(const double* __restrict__ A, int p, double* __restrict__ out)
{
    __shared__ double sh[4];
    const int k = blockIdx.x;
    const double* a = A + (size_t)k * p * p;
    double m = 0.0;
    const size_t pp = (size_t)p * p;
    for (size_t i = threadIdx.x; i < pp; i += 256) {
        const int r = (int)(i / p), c = (int)(i - (size_t)r * p);
        if (c > r) m = fmax(m, fabs(a[i] - a[(size_t)c * p + r]));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[k] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}

void launch_asym_max(hipStream_t st, const double* A, int K, int p, double* out)
{
    hipLaunchKernelGGL(k_asym_max, dim3(K), dim3(256), 0, st, A, p, out);
}

// partials[k][b][0] = sum_chunk A*B
__global__ __launch_bounds__(EW_THREADS) void k_dot(const double* __restrict__ A, const double* __restrict__ B,
                                                    size_t pp, double* __restrict__ partials)
{
    __shared__ double scratch[EW_THREADS / 64];
    const int k = blockIdx.y;
    const size_t base = (size_t)k * pp;
    double acc[1] = {0.0};
    size_t i = (size_t)blockIdx.x * EW_CHUNK + threadIdx.x;
#pragma unroll
    for (int e = 0; e < EW_EPT; ++e, i += EW_THREADS)
        if (i < pp) acc[0] += A[base + i] * B[base + i];
    block_sum<1>(acc, scratch);
    if (threadIdx.x == 0) partials[(size_t)k * gridDim.x + blockIdx.x] = acc[0];
}

// partials[k][b][0] = number of non-zero entries of the chunk (exact: counts are far below 2^53)
__global__ __launch_bounds__(EW_THREADS) void k_count_nonzero(const double* __restrict__ A, size_t pp,
                                                              double* __restrict__ partials)
{
    __shared__ double scratch[EW_THREADS / 64];
    const int k = blockIdx.y;
    const size_t base = (size_t)k * pp;
    double acc[1] = {0.0};
    size_t i = (size_t)blockIdx.x * EW_CHUNK + threadIdx.x;
#pragma unroll
    for (int e = 0; e < EW_EPT; ++e, i += EW_THREADS)
        if (i < pp && A[base + i] != 0.0) acc[0] += 1.0;
    block_sum<1>(acc, scratch);
    if (threadIdx.x == 0) partials[(size_t)k * gridDim.x + blockIdx.x] = acc[0];
}

void launch_count_nonzero(hipStream_t st, const double* A, int K, int p, double* partials)
{
    size_t pp = (size_t)p * p;
    hipLaunchKernelGGL(k_count_nonzero, dim3(elementwise_blocks(p), K), dim3(EW_THREADS), 0, st, A, pp, partials);
}

// out[k] = trace(A_k) - shift: one workgroup per instance, strided partial sums, then a fixed-order tree (deterministic)
__global__ __launch_bounds__(256) void k_trace(const double* __restrict__ A, int p, double shift, double* __restrict__ out)
{
    __shared__ double sh[256];
    const double* a = A + (size_t)blockIdx.x * p * p;
    double t = 0.0;
    for (int i = threadIdx.x; i < p; i += 256) t += a[(size_t)i * p + i];
    sh[threadIdx.x] = t;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = sh[0] - shift;
}

void launch_trace(hipStream_t st, const double* A, int K, int p, double shift, double* out)
{
    hipLaunchKernelGGL(k_trace, dim3(K), dim3(256), 0, st, A, p, shift, out);
}

// instance gather / scatter between stacks (the two-tier L-step's compact sub-batch)
__global__ __launch_bounds__(256) void k_copy_instances(double* __restrict__ dst, const double* __restrict__ src,
                                                        const int* __restrict__ idx, size_t pp, int scatter)
{
    const int i = blockIdx.y;
    const size_t so = (size_t)(scatter ? i : idx[i]) * pp, dp = (size_t)(scatter ? idx[i] : i) * pp;
    if (pp & 1) {
        // odd p: every second slot offset is only 8-byte aligned -- element copies (the double2 form was a
        // misaligned access on those slots)
        for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < pp; e += (size_t)gridDim.x * 256) dst[dp + e] = src[so + e];
        return;
    }
    const size_t n2 = pp / 2;                             // p even: slots are 16-byte aligned, 16-byte copies
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n2; e += (size_t)gridDim.x * 256)
        reinterpret_cast<double2*>(dst + dp)[e] = reinterpret_cast<const double2*>(src + so)[e];
}

void launch_copy_instances(hipStream_t st, double* dst, const double* src, const int* idx, int m, size_t pp, bool scatter)
{
    const int bx = (int)std::min<size_t>((pp / 2 + 255) / 256, 256);
    hipLaunchKernelGGL(k_copy_instances, dim3(std::max(bx, 1), m), dim3(256), 0, st, dst, src, idx, pp, scatter ? 1 : 0);
}

// One contiguous block of doubles copied / filled by an ordinary kernel in the stream: what the snapshots of a batch use instead
// of hipMemcpyAsync / hipMemsetAsync, whose device-to-device copies and fills are the runtime's (blit kernels or SDMA, ordered
// against each other by the runtime's own signals) -- a kernel behind a kernel in one stream is ordered by the queue itself.
__global__ __launch_bounds__(256) void k_copy_block(double* __restrict__ dst, const double* __restrict__ src, size_t n)
{
    const size_t step = (size_t)gridDim.x * 256;
    if (((reinterpret_cast<size_t>(dst) | reinterpret_cast<size_t>(src)) & 15) == 0) {
        const size_t n2 = n / 2;
        for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n2; e += step)
            reinterpret_cast<double2*>(dst)[e] = src ? reinterpret_cast<const double2*>(src)[e] : double2{0.0, 0.0};
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) dst[n - 1] = src ? src[n - 1] : 0.0;
        return;
    }
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += step) dst[e] = src ? src[e] : 0.0;
}

void launch_copy_block(hipStream_t st, double* dst, const double* src, size_t n)
{
    if (n == 0) return;
    const int bx = (int)std::min<size_t>((n / 2 + 255) / 256, 2048);
    hipLaunchKernelGGL(k_copy_block, dim3(std::max(bx, 1)), dim3(256), 0, st, dst, src, n);
}

// A_k += shift * I for the K instances of a stack (the shifted definiteness tests of ggl_exit_checks_fast)
__global__ __launch_bounds__(256) void k_add_diag(double* __restrict__ A, int p, double shift)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < p) A[(size_t)blockIdx.y * p * p + (size_t)i * p + i] += shift;
}

void launch_add_diag(hipStream_t st, double* A, int K, int p, double shift)
{
    hipLaunchKernelGGL(k_add_diag, dim3((p + 255) / 256, K), dim3(256), 0, st, A, p, shift);
}

// d[k][i] = A[k][i][i]
__global__ __launch_bounds__(256) void k_get_diag(const double* __restrict__ A, int p, double* __restrict__ d)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < p) d[(size_t)blockIdx.y * p + i] = A[(size_t)blockIdx.y * p * p + (size_t)i * p + i];
}

void launch_get_diag(hipStream_t st, const double* A, int K, int p, double* d)
{
    hipLaunchKernelGGL(k_get_diag, dim3((p + 255) / 256, K), dim3(256), 0, st, A, p, d);
}

// A_k = I for the K instances of a stack (ggl_reset_instance)
__global__ __launch_bounds__(256) void k_set_identity(double* __restrict__ A, int p)
{
    double* a = A + (size_t)blockIdx.y * p * p;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < (size_t)p * p; e += (size_t)gridDim.x * 256)
        a[e] = (e / p == e % p) ? 1.0 : 0.0;
}

void launch_set_identity(hipStream_t st, double* A, int K, int p)
{
    const int bx = (int)std::min<size_t>(((size_t)p * p + 255) / 256, 512);
    hipLaunchKernelGGL(k_set_identity, dim3(bx, K), dim3(256), 0, st, A, p);
}

void launch_dot(hipStream_t st, const double* A, const double* B, int K, int p, double* partials)
{
    size_t pp = (size_t)p * p;
    hipLaunchKernelGGL(k_dot, dim3(elementwise_blocks(p), K), dim3(EW_THREADS), 0, st, A, B, pp, partials);
}

// ---------------------------------------------------------------------------------------------
// Thresholded estimates of the model selection (helper/model_selection.py:698-705: off-diagonal entries with
// |a| <= tau are zeroed, the diagonal is kept).  Slot s of the launch reads instance src[s] with threshold tauS[s]:
// SUMS: partials[s][b] = { <S_src, T>, count_nonzero(T) } (the terms of aic_single / ebic_single, :812-856);
// WRITE: T is stored to out[s] (input of the eigenvalue pass for log det T).
template <bool SUMS, bool WRITE>
__global__ __launch_bounds__(EW_THREADS) void k_threshold(const double* __restrict__ A, const double* __restrict__ S,
                                                          const int* __restrict__ src, const double* __restrict__ tauS,
                                                          int p, size_t pp, double* __restrict__ out,
                                                          double* __restrict__ partials)
{
    __shared__ double scratch[2 * (EW_THREADS / 64)];
    const int s = blockIdx.y;
    const size_t base = (size_t)src[s] * pp;
    const double tau = tauS[s];
    double acc[2] = {0.0, 0.0};
    size_t i = (size_t)blockIdx.x * EW_CHUNK + threadIdx.x;
#pragma unroll
    for (int e = 0; e < EW_EPT; ++e, i += EW_THREADS) {
        if (i < pp) {
            const double a = A[base + i];
            const bool keep = fabs(a) > tau || i % ((size_t)p + 1) == 0;
            const double t = keep ? a : 0.0;
            if (WRITE) out[(size_t)s * pp + i] = t;
            if (SUMS) {
                acc[0] += t * S[base + i];
                acc[1] += (t != 0.0) ? 1.0 : 0.0;
            }
        }
    }
    if (SUMS) {
        block_sum<2>(acc, scratch);
        if (threadIdx.x == 0) {
            double* o = partials + ((size_t)s * gridDim.x + blockIdx.x) * 2;
            o[0] = acc[0];
            o[1] = acc[1];
        }
    }
}

void launch_threshold_sums(hipStream_t st, const double* A, const double* S, const int* src, const double* tauS,
                           int nslot, int p, double* partials)
{
    size_t pp = (size_t)p * p;
    hipLaunchKernelGGL((k_threshold<true, false>), dim3(elementwise_blocks(p), nslot), dim3(EW_THREADS), 0, st, A, S, src,
                       tauS, p, pp, (double*)nullptr, partials);
}

void launch_threshold_write(hipStream_t st, const double* A, const int* src, const double* tauS, int nslot, int p,
                            double* out)
{
    size_t pp = (size_t)p * p;
    hipLaunchKernelGGL((k_threshold<false, true>), dim3(elementwise_blocks(p), nslot), dim3(EW_THREADS), 0, st, A,
                       (const double*)nullptr, src, tauS, p, pp, out, (double*)nullptr);
}

__global__ __launch_bounds__(256) void k_axpy(double* __restrict__ out, const double* __restrict__ A, double c,
                                              const double* __restrict__ B, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = A[i] + c * B[i];
}

void launch_axpy(hipStream_t st, double* out, const double* A, double c, const double* B, size_t n)
{
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_axpy, dim3(blocks), dim3(256), 0, st, out, A, c, B, n);
}

__global__ __launch_bounds__(EW_THREADS) void k_kkt_w(double* __restrict__ out, const double* __restrict__ Omega,
                                                      const double* __restrict__ S, const double* __restrict__ X,
                                                      const double* __restrict__ nkK, double rho, size_t pp)
{
    const int k = blockIdx.y;
    const double nk = nkK[k];
    const size_t base = (size_t)k * pp;
    size_t i = (size_t)blockIdx.x * EW_CHUNK + threadIdx.x;
#pragma unroll
    for (int e = 0; e < EW_EPT; ++e, i += EW_THREADS)
        if (i < pp) out[base + i] = (Omega[base + i] - nk * S[base + i]) - rho * X[base + i];
}

void launch_kkt_w(hipStream_t st, double* out, const double* Omega, const double* S, const double* X,
                  const double* nkK, double rho, int K, int p)
{
    size_t pp = (size_t)p * p;
    hipLaunchKernelGGL(k_kkt_w, dim3(elementwise_blocks(p), K), dim3(EW_THREADS), 0, st, out, Omega, S, X, nkK, rho, pp);
}

__global__ __launch_bounds__(EW_THREADS) void k_sqdiff(const double* __restrict__ A, const double* __restrict__ B,
                                                       size_t pp, double* __restrict__ partials)
{
    __shared__ double scratch[EW_THREADS / 64];
    const int k = blockIdx.y;
    const size_t base = (size_t)k * pp;
    double acc[1] = {0.0};
    size_t i = (size_t)blockIdx.x * EW_CHUNK + threadIdx.x;
#pragma unroll
    for (int e = 0; e < EW_EPT; ++e, i += EW_THREADS)
        if (i < pp) {
            const double d = B ? A[base + i] - B[base + i] : A[base + i];
            acc[0] += d * d;
        }
    block_sum<1>(acc, scratch);
    if (threadIdx.x == 0) partials[(size_t)k * gridDim.x + blockIdx.x] = acc[0];
}

void launch_sqdiff(hipStream_t st, const double* A, const double* B, int K, int p, double* partials)
{
    size_t pp = (size_t)p * p;
    hipLaunchKernelGGL(k_sqdiff, dim3(elementwise_blocks(p), K), dim3(EW_THREADS), 0, st, A, B, pp, partials);
}

// stateless prox_od_1norm (ggl_helper.py:16-27)
__global__ __launch_bounds__(256) void k_prox_od(double* __restrict__ out, const double* __restrict__ A, double lam,
                                                 const double* __restrict__ lam_pp, int p)
{
    const size_t pp = (size_t)p * p;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < pp; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / p), c = (int)(i - (size_t)r * p);
        const double v = A[i];
        out[i] = (r == c) ? v : soft(v, lam_pp ? lam_pp[i] : lam);
    }
}

void launch_prox_od(hipStream_t st, double* out, const double* A, double lam, const double* lam_pp, int p)
{
    size_t pp = (size_t)p * p;
    int blocks = (int)((pp + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_prox_od, dim3(blocks), dim3(256), 0, st, out, A, lam, lam_pp, p);
}

}  // namespace ggl
