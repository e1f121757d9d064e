// Theta-step of ADMM_MGL: prox_p (solver/ggl_helper.py:190-207) over the (K,p,p) stack.
//
// The penalty acts on the K-vector X[:,i,j] of every upper-triangle pair i<j and the result is
// mirrored.  One workgroup owns a tile pair (I,J), I<=J: the "upper role" of a thread computes
// theta for element (i,j) from coalesced row reads of the upper tile, hands it through an LDS
// tile to the thread that owns the transposed element (j,i) ("lower role"), and both roles then
// finish their own element in its native, coalesced orientation: Theta write, dual update
// X += Omega - Theta (admm_solver.py:208) and the five stopping-test sums (admm_solver.py:316-331).
// Nothing is read or written with a stride: the only transposition happens in LDS.
//
// GGL (ggl_helper.py:68-71): two sweeps over k (sum of squares, then scale).
// FGL (ggl_helper.py:131-134): the K-vector of each pair sits in an LDS column ([k][thread], bank
// = thread, conflict free however the threads' scan positions diverge) and Condat's scan
// (fgl_helper.py:11-68) runs on it in place, one pair per thread.
#include <type_traits>
#include "common.hpp"
#include "kernels.hpp"

namespace ggl {

static constexpr int PT = 32;        // GGL tile edge
static constexpr int PTY = 8;        // thread rows of the 32x8 block
static constexpr int PQ = PT / PTY;  // elements per thread

__device__ __forceinline__ void decode_pair(int b, int T, int& I, int& J)
{
    int i = 0;
    while (b >= T - i) { b -= T - i; ++i; }
    I = i;
    J = i + b;
}

static inline int ntiles(int p, int t) { return (p + t - 1) / t; }
static inline int fgl_tile(int K) { return (K <= 32) ? 16 : 8; }
static constexpr int FGL_MAX_K_TD8 = (160 * 1024 - 1024) / (8 * 8 * 8);   // LDS scan buffer bound

int ggl_chunks(int K, int p);
int fgl_max_K() { return FGL_MAX_K_TD8; }

int pval_blocks(int p)
{
    const int T = ntiles(p, PT);
    return T * (T + 1) / 2;
}

static inline int flat_blocks(int p) { return (int)(((size_t)p * p + 255) / 256); }

// flat: 0 tile-pair kernels, 1 per-element kernel, 2 per-element kernel with the K-column split over the four waves of a
// workgroup where that pays (K > FLAT4_MIN_K)
static constexpr int FLAT4_MIN_K = 8;
static constexpr int FLAT4_CHUNKS = 2;     // 64-element chunks per workgroup (halves the partial-sum rows of the reduction)
static constexpr int GGL_FLAT1_MAX_K = 32;   // the one-thread-per-element kernel keeps the whole K-column in ONE lane's registers
static inline bool use_flat4(int K, int flat) { return (flat == 2 && K > FLAT4_MIN_K) || (flat != 0 && K > GGL_FLAT1_MAX_K); }

// Many instances of a SMALL matrix (one problem, K > 64, fewer than 256 workgroups of 128 elements): the workgroup takes 16
// elements instead of 128 (k_theta_ggl_flat16), e.g. K = 256, p = 64: 256 workgroups instead of 32 on the chip's 256 CUs.
static inline bool use_flat16(int K, int p, int flat, int G)
{
    return G == 1 && use_flat4(K, flat) && K > 64 && ((size_t)p * p + 127) / 128 < 256;
}

static inline int fgl_flat_nt(int K);
int theta_partial_blocks(int p, int reg, int K, int flat, int G)
{
    if (reg == 2 && flat && K <= FGL_MAX_K_TD8) return (int)(((size_t)p * p + fgl_flat_nt(K) - 1) / fgl_flat_nt(K));
    if (reg == 1 && flat && K <= GGL_FLAT_MAX_K && use_flat16(K, p, flat, G)) return (int)(((size_t)p * p + 15) / 16);
    if (reg == 1 && flat && K <= GGL_FLAT_MAX_K) return use_flat4(K, flat) ? (int)(((size_t)p * p + 64 * FLAT4_CHUNKS - 1) / (64 * FLAT4_CHUNKS)) : flat_blocks(p);
    return pair_blocks(p, reg, K);
}

int pair_blocks(int p, int reg, int K)
{
    if (reg == 2) {
        const int T = ntiles(p, fgl_tile(K));
        return T * (T + 1) / 2;
    }
    const int T = ntiles(p, PT);
    return T * (T + 1) / 2 * ggl_chunks(K, p);
}

// ---------------------------------------------------------------------------------------------
// GGL.  Two kernels so that the work spreads over (tile pair) x (K-chunk) workgroups:
//   k_group_partial: sq[c][i][j] = sum over the k of chunk c of soft(Omega+L+X, l1)^2      (i<j)
//   k_theta_ggl:     ss = sum_c sq[c][i][j]; theta for the k of its chunk; mirror; dual update; norms
// A K-sharded run all-reduces the (p,p) sum between the two (gglasso_amd/dist.py).
// ---------------------------------------------------------------------------------------------
int ggl_chunks(int K, int p)
{
    const int T = ntiles(p, PT);
    const int pairs = T * (T + 1) / 2;
    int kc = (1024 + pairs - 1) / pairs;      // aim at >= ~1024 workgroups
    if (kc < 1) kc = 1;
    if (kc > K) kc = K;
    const int len = (K + kc - 1) / kc;        // k per chunk
    return (K + len - 1) / len;
}
static inline int ggl_chunk_len(int K, int p)
{
    const int kc = ggl_chunks(K, p);
    return (K + kc - 1) / kc;
}

__global__ __launch_bounds__(256) void k_group_partial(double* __restrict__ sq, const double* __restrict__ Omega,
                                                       const double* __restrict__ L, const double* __restrict__ X,
                                                       double l1, int K, int p, int klen)
{
    const int T = (p + PT - 1) / PT;
    int I, J;
    decode_pair(blockIdx.x, T, I, J);
    const bool diag = (I == J);
    const int I0 = I * PT, J0 = J * PT;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const size_t pp = (size_t)p * p;
    const int k0 = blockIdx.y * klen, k1 = min(K, k0 + klen);
    double ss[PQ];
    size_t off[PQ];
    bool ok[PQ];
#pragma unroll
    for (int q = 0; q < PQ; ++q) {
        const int r = ty + PTY * q;
        ok[q] = (I0 + r < p) && (J0 + tx < p) && (!diag || r < tx);
        off[q] = (size_t)(I0 + r) * p + (J0 + tx);
        ss[q] = 0.0;
    }
#pragma unroll 4
    for (int k = k0; k < k1; ++k) {
        const size_t base = (size_t)k * pp;
#pragma unroll
        for (int q = 0; q < PQ; ++q) {
            if (ok[q]) {
                double v = Omega[base + off[q]];
                if (L) v += L[base + off[q]];
                if (X) v += X[base + off[q]];
                const double u = soft(v, l1);
                ss[q] += u * u;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < PQ; ++q)
        if (ok[q]) sq[(size_t)blockIdx.y * pp + off[q]] = ss[q];
}

// out = the PACKED upper triangle (tri_index, common.hpp) of sum_c sq[c][i][j] (sq holds the strict upper triangle; diagonal
// 0) followed by the speculation flag of this rank: what the ranks of a K-sharded run all-reduce -- p (p + 1) / 2 + 1
// doubles, half the full matrix -- whichever Theta kernel each of them runs afterwards
__global__ __launch_bounds__(256) void k_sum_chunks(double* __restrict__ out, const double* __restrict__ sq, int nsq,
                                                    int p, const int* __restrict__ flags)
{
    const size_t pp = (size_t)p * p;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[tri_len(p)] = spec_failed(flags) ? 1.0 : 0.0;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < pp; e += (size_t)gridDim.x * 256) {
        const int i = (int)(e / p), j = (int)(e - (size_t)i * p);
        if (i > j) continue;
        double s = 0.0;
        if (i != j)
            for (int c = 0; c < nsq; ++c) s += sq[(size_t)c * pp + e];
        out[tri_index(i, j, p)] = s;
    }
}

// the same vector from one thread per element of the upper triangle (K <= GGL_FLAT_MAX_K): sum_k soft(Omega + L + X, l1)^2 of
// the element's own K-column, no tile pairs, no chunk buffer; the lower triangle is neither read nor written (half the pass)
__global__ __launch_bounds__(256) void k_group_partial_flat(double* __restrict__ out, const double* __restrict__ Omega,
                                                            const double* __restrict__ L, const double* __restrict__ X,
                                                            double l1, int K, int p, const int* __restrict__ flags)
{
    const size_t pp = (size_t)p * p;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e == 0) out[tri_len(p)] = spec_failed(flags) ? 1.0 : 0.0;
    if (e >= pp) return;
    const int i = (int)(e / p), j = (int)(e - (size_t)i * p);
    if (i > j) return;
    double ss = 0.0;
    if (i != j) {
#pragma unroll 4
        for (int k = 0; k < K; ++k) {
            const size_t o = (size_t)k * pp + e;
            double v = Omega[o];
            if (L) v += L[o];
            if (X) v += X[o];
            const double u = soft(v, l1);
            ss += u * u;
        }
    }
    out[tri_index(i, j, p)] = ss;
}

template <bool FUSE_DUAL>
__global__ __launch_bounds__(256) void k_theta_ggl(double* __restrict__ Theta, double* __restrict__ X,
                                                   double* __restrict__ C, const double* __restrict__ Omega,
                                                   const double* __restrict__ OmegaPrev,
                                                   const double* __restrict__ L, double l1, double l2,
                                                   const double* __restrict__ sq, int nsq,
                                                   double* __restrict__ partials, int K, int p, int klen,
                                                   const int* __restrict__ skip, int packed)
{
    // packed: sq is the all-reduced GROUPSQ of a K-sharded run (packed upper triangle + flag, common.hpp), nsq = 1
    __shared__ double tile[2][PT][PT + 1];
    __shared__ double scratch[GGL_NNORM * 4];
    if (spec_failed(skip)) return;
    if (packed && gsq_rejected(sq, p)) return;
    const int T = (p + PT - 1) / PT;
    int I, J;
    decode_pair(blockIdx.x, T, I, J);
    const bool diag = (I == J);
    const int I0 = I * PT, J0 = J * PT;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const size_t pp = (size_t)p * p;
    const int k0 = blockIdx.y * klen, k1 = min(K, k0 + klen);

    bool up_ok[PQ], pr_ok[PQ], lo_ok[PQ];
    size_t up_off[PQ], lo_off[PQ];
    double amul[PQ], adiv[PQ];   // theta = u * (a - l2) / a     (ggl_helper.py:38-43)
#pragma unroll
    for (int q = 0; q < PQ; ++q) {
        const int r = ty + PTY * q;
        up_ok[q] = (I0 + r < p) && (J0 + tx < p) && (!diag || r <= tx);
        pr_ok[q] = up_ok[q] && !(diag && r == tx);
        up_off[q] = (size_t)(I0 + r) * p + (J0 + tx);
        lo_ok[q] = (J0 + r < p) && (I0 + tx < p) && (!diag || r > tx);   // element (J0+r, I0+tx)
        lo_off[q] = (size_t)(J0 + r) * p + (I0 + tx);
        double ss = 0.0;
        if (pr_ok[q]) {
            if (packed) ss = sq[tri_index(I0 + r, J0 + tx, p)];
            else
                for (int c = 0; c < nsq; ++c) ss += sq[(size_t)c * pp + up_off[q]];
        }
        const double a = fmax(sqrt(ss), l2);
        amul[q] = a - l2;
        adiv[q] = a;
    }

    double acc[GGL_NNORM] = {0, 0, 0, 0, 0};
    for (int k = k0; k < k1; ++k) {
        const size_t base = (size_t)k * pp;
        const int buf = (k - k0) & 1;
        // issue the lower-role loads early: they do not depend on the LDS hand-off
        double lom[PQ], lx[PQ], lop[PQ];
#pragma unroll
        for (int q = 0; q < PQ; ++q) {
            lom[q] = lx[q] = lop[q] = 0.0;
            if (lo_ok[q]) {
                const size_t o = base + lo_off[q];
                if (FUSE_DUAL) { lom[q] = Omega[o]; lx[q] = X[o]; lop[q] = OmegaPrev[o]; }
                else if (C) { lom[q] = Omega[o]; lx[q] = X[o]; }
            }
        }
#pragma unroll
        for (int q = 0; q < PQ; ++q) {
            if (up_ok[q]) {
                const size_t o = base + up_off[q];
                const double om = Omega[o];
                const double l = L ? L[o] : 0.0;
                const double x = X ? X[o] : 0.0;
                const double v = (om + l) + x;
                const double th = pr_ok[q] ? soft(v, l1) * amul[q] / adiv[q] : v;
                Theta[o] = th;
                tile[buf][ty + PTY * q][tx] = th;
                if (FUSE_DUAL) {
                    const double xn = x + (om - th);
                    X[o] = xn;
                    const double dp = om - OmegaPrev[o];
                    acc[0] += om * om;
                    acc[1] += th * th;
                    acc[2] += xn * xn;
                    acc[3] += (om - th) * (om - th);
                    acc[4] += dp * dp;
                } else if (C) {
                    C[o] = (th - x) - om;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PQ; ++q) {
            if (lo_ok[q]) {
                const size_t o = base + lo_off[q];
                const double th = tile[buf][tx][ty + PTY * q];
                Theta[o] = th;
                if (FUSE_DUAL) {
                    const double om = lom[q];
                    const double xn = lx[q] + (om - th);
                    X[o] = xn;
                    const double dp = om - lop[q];
                    acc[0] += om * om;
                    acc[1] += th * th;
                    acc[2] += xn * xn;
                    acc[3] += (om - th) * (om - th);
                    acc[4] += dp * dp;
                } else if (C) {
                    C[o] = (th - lx[q]) - lom[q];
                }
            }
        }
    }
    if (FUSE_DUAL) {
        const int lane = (ty * PT + tx) & 63, wid = (ty * PT + tx) >> 6;
#pragma unroll
        for (int v = 0; v < GGL_NNORM; ++v) acc[v] = wave_sum(acc[v]);
        if (lane == 0) {
#pragma unroll
            for (int v = 0; v < GGL_NNORM; ++v) scratch[wid * GGL_NNORM + v] = acc[v];
        }
        __syncthreads();
        if (tx == 0 && ty == 0) {
            double* o = partials + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * GGL_NNORM;
#pragma unroll
            for (int v = 0; v < GGL_NNORM; ++v)
                o[v] = (scratch[v] + scratch[GGL_NNORM + v]) + (scratch[2 * GGL_NNORM + v] + scratch[3 * GGL_NNORM + v]);
        }
    }
}

// sq: ggl_chunks(K,p) * p * p doubles.  Writes the per-chunk partial sums of squares.
void launch_group_partial(hipStream_t st, double* sq, const double* Omega, const double* L, const double* X,
                          double l1, int K, int p)
{
    const int T = ntiles(p, PT);
    hipLaunchKernelGGL(k_group_partial, dim3(T * (T + 1) / 2, ggl_chunks(K, p)), dim3(PT, PTY), 0, st, sq, Omega, L, X,
                       l1, K, p, ggl_chunk_len(K, p));
}

void launch_sum_chunks(hipStream_t st, double* out, const double* sq, int nsq, int p, const int* flags)
{
    const size_t pp = (size_t)p * p;
    int blocks = (int)((pp + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_sum_chunks, dim3(blocks), dim3(256), 0, st, out, sq, nsq, p, flags);
}

// GROUPSQ of a K-sharded run: the packed upper triangle of this rank's sum_k u^2 + this rank's speculation flag (flags: the
// validation flags of the step's parts, or null) -- ONE launch, tri_len(p) + 1 doubles
void launch_group_sums_packed(hipStream_t st, double* out, double* sqwork, const double* Omega, const double* L,
                              const double* X, double l1, int K, int p, const int* flags)
{
    if (K <= GGL_FLAT_MAX_K) {
        hipLaunchKernelGGL(k_group_partial_flat, dim3(flat_blocks(p)), dim3(256), 0, st, out, Omega, L, X, l1, K, p, flags);
        return;
    }
    launch_group_partial(st, sqwork, Omega, L, X, l1, K, p);
    launch_sum_chunks(st, out, sqwork, ggl_chunks(K, p), p, flags);
}

// ---------------------------------------------------------------------------------------------
// FGL
// ---------------------------------------------------------------------------------------------
// Condat's direct 1-D TV prox (fgl_helper.py:11-68) on y[0], y[s], ..., y[(N-1)s], in place:
// a segment is only written once the scan has moved past it, so x may overwrite y.
__device__ __forceinline__ void condat_inplace(double* y, const int s, const int N, const double lam)
{
    int k = 0, k0 = 0, kplus = 0, kminus = 0;
    double vmin = y[0] - lam, vmax = y[0] + lam, umin = lam, umax = -lam;
    for (;;) {
        while (k == N - 1) {
            if (umin < 0.0) {
                for (int i = k0; i <= kminus; ++i) y[i * s] = vmin;
                kminus += 1;
                k = k0 = kminus;
                umin = lam;
                vmin = y[k * s];
                umax = y[k * s] + lam - vmax;
            } else if (umax > 0.0) {
                for (int i = k0; i <= kplus; ++i) y[i * s] = vmax;
                kplus += 1;
                k = k0 = kplus;
                umax = -lam;
                vmax = y[k * s];
                umin = y[k * s] - lam - vmin;
            } else {
                const double v = vmin + umin / (double)(k - k0 + 1);
                for (int i = k0; i < N; ++i) y[i * s] = v;
                return;
            }
            if (k == N - 1) {
                y[k * s] = vmin + umin;
                return;
            }
        }
        const double yn = y[(k + 1) * s];
        if (yn + umin - vmin < -lam) {
            for (int i = k0; i <= kminus; ++i) y[i * s] = vmin;
            kminus += 1;
            k = kplus = k0 = kminus;
            vmin = y[k * s];
            vmax = y[k * s] + 2 * lam;
            umin = lam;
            umax = -lam;
        } else if (yn + umax - vmax > lam) {
            for (int i = k0; i <= kplus; ++i) y[i * s] = vmax;
            kplus += 1;
            k = kminus = k0 = kplus;
            vmin = y[k * s] - 2 * lam;
            vmax = y[k * s];
            umin = lam;
            umax = -lam;
        } else {
            k += 1;
            umin = umin + yn - vmin;
            umax = umax + yn - vmax;
            if (umin >= lam) {
                vmin += (umin - lam) / (double)(k - k0 + 1);
                umin = lam;
                kminus = k;
            }
            if (umax <= -lam) {
                vmax += (umax + lam) / (double)(k - k0 + 1);
                umax = -lam;
                kplus = k;
            }
        }
    }
}


// Condat's direct 1-D TV prox (fgl_helper.py:11-68), same arithmetic in the same order as condat_inplace, restructured for a
// wavefront: ONE loop whose body is one decision of the scan (extend the segment / negative jump / positive jump / the three
// end-of-signal cases), no inner loops.  Closing a segment [k0 .. kminus|kplus] does not fill it: the value goes to y[k0] and
// k0 is marked as a segment start; elements behind a closed segment are never read again, and every element is written
// exactly once by the uniform pass at the end that carries each start's value forward (f: applied to the values on the way).
template <int NWORDS, class F>
__device__ __forceinline__ void condat_uniform(double* y, const int s, const int N, const double lam, F f)
{
    int k = 0, k0 = 0, kplus = 0, kminus = 0;
    double vmin = y[0] - lam, vmax = y[0] + lam, umin = lam, umax = -lam;
    unsigned long long starts[NWORDS];
#pragma unroll
    for (int w = 0; w < NWORDS; ++w) starts[w] = 0ull;
    auto close = [&](int a, double v) {
        y[a * s] = v;
#pragma unroll
        for (int w = 0; w < NWORDS; ++w) starts[w] |= ((a >> 6) == w) ? (1ull << (a & 63)) : 0ull;
    };
    bool done = false;
    while (!done) {
        if (k == N - 1) {
            if (umin < 0.0) {
                close(k0, vmin);
                kminus += 1;
                k = k0 = kminus;
                umin = lam;
                vmin = y[k * s];
                umax = vmin + lam - vmax;
                if (k == N - 1) { close(k, vmin + umin); done = true; }
            } else if (umax > 0.0) {
                close(k0, vmax);
                kplus += 1;
                k = k0 = kplus;
                umax = -lam;
                vmax = y[k * s];
                umin = vmax - lam - vmin;
                if (k == N - 1) { close(k, vmin + umin); done = true; }
            } else {
                close(k0, vmin + umin / (double)(k - k0 + 1));
                done = true;
            }
        } else {
            const double yn = y[(k + 1) * s];
            if (yn + umin - vmin < -lam) {
                close(k0, vmin);
                kminus += 1;
                k = kplus = k0 = kminus;
                vmin = y[k * s];
                vmax = vmin + 2 * lam;
                umin = lam;
                umax = -lam;
            } else if (yn + umax - vmax > lam) {
                close(k0, vmax);
                kplus += 1;
                k = kminus = k0 = kplus;
                vmax = y[k * s];
                vmin = vmax - 2 * lam;
                umin = lam;
                umax = -lam;
            } else {
                k += 1;
                umin = umin + yn - vmin;
                umax = umax + yn - vmax;
                if (umin >= lam) {
                    vmin += (umin - lam) / (double)(k - k0 + 1);
                    umin = lam;
                    kminus = k;
                }
                if (umax <= -lam) {
                    vmax += (umax + lam) / (double)(k - k0 + 1);
                    umax = -lam;
                    kplus = k;
                }
            }
        }
    }
    double cur = 0.0;
    for (int i = 0; i < N; ++i) {
        bool st = false;
#pragma unroll
        for (int w = 0; w < NWORDS; ++w) st = st || (((i >> 6) == w) && ((starts[w] >> (i & 63)) & 1ull));
        if (st) cur = y[i * s];
        y[i * s] = f(cur);
    }
}

// the scan for a K-vector of any admissible length (FGL_MAX_K_TD8 = 318 <= 5 * 64): N is uniform over the launch
template <class F>
__device__ __forceinline__ void condat_scan(double* y, const int s, const int N, const double lam, F f)
{
    if (N <= 64) condat_uniform<1>(y, s, N, lam, f);
    else if (N <= 128) condat_uniform<2>(y, s, N, lam, f);
    else if (N <= 192) condat_uniform<3>(y, s, N, lam, f);
    else if (N <= 256) condat_uniform<4>(y, s, N, lam, f);
    else condat_uniform<5>(y, s, N, lam, f);
}

// A batch of G independent problems of K instances each (model-selection grid, ggl_mgl_batch_step): blockIdx.y = g,
// the stacks are (G*K,p,p), l1G / l2G hold the thresholds of instance g*K (null: the scalars l1 / l2, G = 1), the
// partial sums are rows [g][blocks].
// ABL (GGL_DEV builds, timing ablations, 1-3 with wrong results): 1 no Condat scan, 2 no scan and no soft-threshold pass, 4 the
// nested-loop scan,
// 3 scan only (no global loads / stores)
template <int TD, bool FUSE_DUAL, int ABL = 0>
__global__ __launch_bounds__(TD * TD) void k_theta_fgl(double* __restrict__ Theta, double* __restrict__ X,
                                                       double* __restrict__ C, const double* __restrict__ Omega,
                                                       const double* __restrict__ OmegaPrev,
                                                       const double* __restrict__ L, double l1, double l2,
                                                       double* __restrict__ partials, int K, int p,
                                                       const int* __restrict__ skip, const double* __restrict__ l1G,
                                                       const double* __restrict__ l2G)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];   // [K][TD*TD] then scratch
    constexpr int NT = TD * TD;
    if (spec_failed(skip)) return;
    {
        const size_t goff = (size_t)blockIdx.y * K * p * p;
        Theta += goff; Omega += goff;
        if (X) X += goff;
        if (C) C += goff;
        if (OmegaPrev) OmegaPrev += goff;
        if (L) L += goff;
        if (l1G) { l1 = l1G[(size_t)blockIdx.y * K]; l2 = l2G[(size_t)blockIdx.y * K]; }
        if (partials) partials += (size_t)blockIdx.y * gridDim.x * GGL_NNORM;
    }
    const int T = (p + TD - 1) / TD;
    int I, J;
    decode_pair(blockIdx.x, T, I, J);
    const bool diag = (I == J);
    const int I0 = I * TD, J0 = J * TD;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int tid = ty * TD + tx;
    const size_t pp = (size_t)p * p;
    double* ycol = lds + tid;
    double* scratch = lds + (size_t)K * NT;

    const bool up_ok = (I0 + ty < p) && (J0 + tx < p) && (!diag || ty <= tx);
    const bool pr_ok = up_ok && !(diag && ty == tx);
    const size_t up_off = (size_t)(I0 + ty) * p + (J0 + tx);
    const bool lo_ok = (J0 + ty < p) && (I0 + tx < p) && (!diag || ty > tx);
    const size_t lo_off = (size_t)(J0 + ty) * p + (I0 + tx);

    if (up_ok) {
        // eight instances' loads in flight per thread: with one wave per workgroup and ~6 workgroups per CU (the K-vectors
        // fill the LDS) nothing else hides the latency of these loads
        int k = 0;
        for (; k + 8 <= K; k += 8) {
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = Omega[(size_t)(k + q) * pp + up_off];
            if (L) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] += L[(size_t)(k + q) * pp + up_off];
            }
            if (X) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] += X[(size_t)(k + q) * pp + up_off];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) ycol[(k + q) * NT] = v[q];
        }
        for (; k < K; ++k) {
            const size_t o = (size_t)k * pp + up_off;
            double v = Omega[o];
            if (L) v += L[o];
            if (X) v += X[o];
            ycol[k * NT] = v;
        }
    }
    if (pr_ok) {
        if (ABL == 4) {                   // the nested-loop scan (what ran until round 3): timing comparison, same results
            condat_inplace(ycol, NT, K, l2);
            for (int k = 0; k < K; ++k) ycol[k * NT] = soft(ycol[k * NT], l1);
        } else if (ABL != 1 && ABL != 2) {
            condat_scan(ycol, NT, K, l2, [l1](double v) { return soft(v, l1); });     // prox_phi_fgl: prox_1norm(prox_tv(v))
        } else if (ABL != 2) {
            for (int k = 0; k < K; ++k) ycol[k * NT] = soft(ycol[k * NT], l1);
        }
    }
    __syncthreads();
    if (ABL == 3) { if (tid == 0 && ycol[0] == 1.2345e300) Theta[0] = 0.0; return; }

    double acc[GGL_NNORM] = {0, 0, 0, 0, 0};
#pragma unroll 4
    for (int k = 0; k < K; ++k) {
        const size_t base = (size_t)k * pp;
        if (up_ok) {
            const size_t o = base + up_off;
            const double th = ycol[k * NT];
            Theta[o] = th;
            if (FUSE_DUAL) {
                const double om = Omega[o], x = X[o];
                const double xn = x + (om - th);
                X[o] = xn;
                const double dp = om - OmegaPrev[o];
                acc[0] += om * om;
                acc[1] += th * th;
                acc[2] += xn * xn;
                acc[3] += (om - th) * (om - th);
                acc[4] += dp * dp;
            } else if (C) {
                C[o] = (th - X[o]) - Omega[o];
            }
        }
        if (lo_ok) {
            const size_t o = base + lo_off;
            const double th = lds[(size_t)k * NT + tx * TD + ty];
            Theta[o] = th;
            if (FUSE_DUAL) {
                const double om = Omega[o], x = X[o];
                const double xn = x + (om - th);
                X[o] = xn;
                const double dp = om - OmegaPrev[o];
                acc[0] += om * om;
                acc[1] += th * th;
                acc[2] += xn * xn;
                acc[3] += (om - th) * (om - th);
                acc[4] += dp * dp;
            } else if (C) {
                C[o] = (th - X[o]) - Omega[o];
            }
        }
    }
    if (FUSE_DUAL) {
        constexpr int NW = (NT + 63) / 64;
#pragma unroll
        for (int v = 0; v < GGL_NNORM; ++v) acc[v] = wave_sum(acc[v]);
        if ((tid & 63) == 0) {
#pragma unroll
            for (int v = 0; v < GGL_NNORM; ++v) scratch[(tid >> 6) * GGL_NNORM + v] = acc[v];
        }
        __syncthreads();
        if (tid == 0) {
            double* o = partials + (size_t)blockIdx.x * GGL_NNORM;
#pragma unroll
            for (int v = 0; v < GGL_NNORM; ++v) {
                double s = 0.0;
                for (int w = 0; w < NW; ++w) s += scratch[w * GGL_NNORM + v];
                o[v] = s;
            }
        }
    }
}

// FGL per ELEMENT (exactly symmetric state, like the GGL per-element kernels): a workgroup takes NT consecutive elements of the
// flattened (p,p) index, every thread the K-column of its own element -- (i,j) and (j,i) are both computed, each from its own
// (bitwise equal) inputs, so the output is bitwise symmetric without a mirror pass.  Twice the scans of the tile-pair kernel
// above (cheap since round 3's uniform scan), but every global access is a full row of the wave (512 contiguous bytes per
// instance) where the 8 x 8 tiles read 64-byte row segments and wrote the mirror tile column-wise through LDS: the memory
// side was 202 of the tile kernel's 263 us at (50,500).
// Reference: prox_phi_fgl = prox_1norm(prox_tv(v, l2), l1) on the K-vector of every off-diagonal element
// (solver/ggl_helper.py:73-82, fgl_helper.py:11-68), the diagonal passes through (ggl_helper.py:191-208).
template <int NT, bool FUSE_DUAL>
__global__ __launch_bounds__(NT) void k_theta_fgl_flat(double* __restrict__ Theta, double* __restrict__ X,
                                                       double* __restrict__ C, const double* __restrict__ Omega,
                                                       const double* __restrict__ OmegaPrev,
                                                       const double* __restrict__ L, double l1, double l2,
                                                       double* __restrict__ partials, int K, int p,
                                                       const int* __restrict__ skip, const double* __restrict__ l1G,
                                                       const double* __restrict__ l2G)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];   // [K][NT] then scratch
    if (spec_failed(skip)) return;
    {
        const size_t goff = (size_t)blockIdx.y * K * p * p;
        Theta += goff; Omega += goff;
        if (X) X += goff;
        if (C) C += goff;
        if (OmegaPrev) OmegaPrev += goff;
        if (L) L += goff;
        if (l1G) { l1 = l1G[(size_t)blockIdx.y * K]; l2 = l2G[(size_t)blockIdx.y * K]; }
        if (partials) partials += (size_t)blockIdx.y * gridDim.x * GGL_NNORM;
    }
    const size_t pp = (size_t)p * p;
    const int tid = threadIdx.x;
    const size_t e = (size_t)blockIdx.x * NT + tid;
    const bool live = e < pp;
    double* ycol = lds + tid;
    double* scratch = lds + (size_t)K * NT;
    double acc[GGL_NNORM] = {0, 0, 0, 0, 0};
    // The K-columns fill the LDS (K = 50: six waves per CU), so nothing but a thread's own loads hides the memory latency:
    // the loads of N instances of all three stacks are issued before the first is used (8 in flight: 0.33 ms at (50,500),
    // 16: 0.20 ms, 32: 0.25 ms), and the odd instances at the end go in batches of 8 / 4 / 2 / 1, not one by one.
    auto load = [&](int k, auto nconst) {
        constexpr int N = decltype(nconst)::value;
        double v[N], l[N], x[N];
#pragma unroll
        for (int q = 0; q < N; ++q) {
            const size_t o = (size_t)(k + q) * pp + e;
            v[q] = Omega[o];
            l[q] = L ? L[o] : 0.0;
            x[q] = X ? X[o] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < N; ++q) {
            double u = v[q];
            if (L) u += l[q];
            if (X) u += x[q];
            ycol[(k + q) * NT] = u;
        }
    };
    auto store = [&](int k, auto nconst) {
        constexpr int N = decltype(nconst)::value;
        double om[N], x[N], op[N];
#pragma unroll
        for (int q = 0; q < N; ++q) {
            const size_t o = (size_t)(k + q) * pp + e;
            if (FUSE_DUAL || C) { om[q] = Omega[o]; x[q] = X[o]; }
            if (FUSE_DUAL) op[q] = OmegaPrev[o];
        }
#pragma unroll
        for (int q = 0; q < N; ++q) {
            const size_t o = (size_t)(k + q) * pp + e;
            const double th = ycol[(k + q) * NT];
            Theta[o] = th;
            if (FUSE_DUAL) {
                const double xn = x[q] + (om[q] - th);
                X[o] = xn;
                const double dp = om[q] - op[q];
                acc[0] += om[q] * om[q];
                acc[1] += th * th;
                acc[2] += xn * xn;
                acc[3] += (om[q] - th) * (om[q] - th);
                acc[4] += dp * dp;
            } else if (C) {
                C[o] = (th - x[q]) - om[q];
            }
        }
    };
    using I16 = std::integral_constant<int, 16>;
    using I8 = std::integral_constant<int, 8>;
    using I4 = std::integral_constant<int, 4>;
    using I2 = std::integral_constant<int, 2>;
    using I1 = std::integral_constant<int, 1>;
    if (live) {
        int k = 0;
        for (; k + 16 <= K; k += 16) load(k, I16{});
        if (k + 8 <= K) { load(k, I8{}); k += 8; }
        if (k + 4 <= K) { load(k, I4{}); k += 4; }
        if (k + 2 <= K) { load(k, I2{}); k += 2; }
        if (k < K) load(k, I1{});
        const int i = (int)(e / p), j = (int)(e - (size_t)i * p);
        if (i != j) condat_scan(ycol, NT, K, l2, [l1](double v) { return soft(v, l1); });     // prox_phi_fgl: prox_1norm(prox_tv(v))
        // (every thread reads back its own column only: no barrier)
        k = 0;
        for (; k + 16 <= K; k += 16) store(k, I16{});
        if (k + 8 <= K) { store(k, I8{}); k += 8; }
        if (k + 4 <= K) { store(k, I4{}); k += 4; }
        if (k + 2 <= K) { store(k, I2{}); k += 2; }
        if (k < K) store(k, I1{});
    }
    if (FUSE_DUAL) {
        constexpr int NW = (NT + 63) / 64;
#pragma unroll
        for (int v = 0; v < GGL_NNORM; ++v) acc[v] = wave_sum(acc[v]);
        if ((tid & 63) == 0) {
#pragma unroll
            for (int v = 0; v < GGL_NNORM; ++v) scratch[(tid >> 6) * GGL_NNORM + v] = acc[v];
        }
        __syncthreads();
        if (tid == 0) {
            double* o = partials + (size_t)blockIdx.x * GGL_NNORM;
#pragma unroll
            for (int v = 0; v < GGL_NNORM; ++v) {
                double s = 0.0;
                for (int w = 0; w < NW; ++w) s += scratch[w * GGL_NNORM + v];
                o[v] = s;
            }
        }
    }
}

// elements per workgroup of the per-element FGL kernel: 128 while the K-columns of 128 elements fit the LDS, else 64
static inline int fgl_flat_nt(int K) { return ((size_t)K * 128 * 8 + 1024 <= (size_t)150 * 1024) ? 128 : 64; }

template <int NT>
static hipError_t launch_fgl_flat(hipStream_t st, double* Theta, double* X, double* C, const double* Omega,
                                  const double* OmegaPrev, const double* L, double l1, double l2, int fuse_dual,
                                  double* partials, int K, int p, const int* skip, int G, const double* l1G, const double* l2G)
{
    const size_t lds = ((size_t)K * NT + GGL_NNORM * 4) * sizeof(double);
    dim3 grid((unsigned)(((size_t)p * p + NT - 1) / NT), G), blk(NT);
    hipError_t e;
    if (fuse_dual) {
        e = hipFuncSetAttribute((const void*)k_theta_fgl_flat<NT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_theta_fgl_flat<NT, true>), grid, blk, lds, st, Theta, X, C, Omega, OmegaPrev, L, l1, l2, partials, K, p, skip, l1G, l2G);
    } else {
        e = hipFuncSetAttribute((const void*)k_theta_fgl_flat<NT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_theta_fgl_flat<NT, false>), grid, blk, lds, st, Theta, X, C, Omega, OmegaPrev, L, l1, l2, partials, K, p, skip, l1G, l2G);
    }
    return hipGetLastError();
}

static hipError_t launch_fgl_flat_any(hipStream_t st, double* Theta, double* X, double* C, const double* Omega,
                                      const double* OmegaPrev, const double* L, double l1, double l2, int fuse_dual,
                                      double* partials, int K, int p, const int* skip, int G, const double* l1G, const double* l2G);

template <int TD>
static hipError_t launch_fgl_td(hipStream_t st, double* Theta, double* X, double* C, const double* Omega,
                                const double* OmegaPrev, const double* L, double l1, double l2, int fuse_dual,
                                double* partials, int K, int p, const int* skip, int G = 1, const double* l1G = nullptr,
                                const double* l2G = nullptr)
{
    const int T = ntiles(p, TD);
    const size_t lds = ((size_t)K * TD * TD + GGL_NNORM * 4) * sizeof(double);
#ifdef GGL_DEV
    if (const char* ab = getenv("GGL_FGL_ABL")) {
        const int a = atoi(ab);
        dim3 grid(T * (T + 1) / 2, G), blk(TD, TD);
#define GGL_FA(N) do { (void)hipFuncSetAttribute((const void*)k_theta_fgl<TD, true, N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((k_theta_fgl<TD, true, N>), grid, blk, lds, st, Theta, X, C, Omega, OmegaPrev, L, l1, l2, partials, K, p, skip, l1G, l2G); } while (0)
        if (fuse_dual && a >= 1 && a <= 4) {
            if (a == 1) GGL_FA(1); else if (a == 2) GGL_FA(2); else if (a == 3) GGL_FA(3); else GGL_FA(4);
            return hipGetLastError();
        }
#undef GGL_FA
    }
#endif
    dim3 grid(T * (T + 1) / 2, G), blk(TD, TD);
    hipError_t e;
    if (fuse_dual) {
        e = hipFuncSetAttribute((const void*)k_theta_fgl<TD, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_theta_fgl<TD, true>), grid, blk, lds, st, Theta, X, C, Omega, OmegaPrev, L, l1, l2, partials, K, p, skip, l1G, l2G);
    } else {
        e = hipFuncSetAttribute((const void*)k_theta_fgl<TD, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_theta_fgl<TD, false>), grid, blk, lds, st, Theta, X, C, Omega, OmegaPrev, L, l1, l2, partials, K, p, skip, l1G, l2G);
    }
    return hipGetLastError();
}

// GGL with the whole K-column of an element in registers (K <= GGL_FLAT_MAX_K): one thread per (i,j), all k.
// Plain streaming -- no tile pairs, no LDS, no second pass over Omega and X for the group norm: the element's
// soft-thresholded values are summed, scaled and written in one go.  Every (i,j) is computed from its OWN
// inputs, which equals the reference's "upper triangle, then mirror" (ggl_helper.py:191-208) only for an
// exactly symmetric state; the caller enables this path only then (the ADMM state is, by construction).
template <int KMAX, bool FUSE_DUAL>
__global__ __launch_bounds__(256) void k_theta_ggl_flat(double* __restrict__ Theta, double* __restrict__ X,
                                                        double* __restrict__ C, const double* __restrict__ Omega,
                                                        const double* __restrict__ OmegaPrev,
                                                        const double* __restrict__ L, double l1, double l2,
                                                        double* __restrict__ partials, int K, int p,
                                                        const int* __restrict__ skip, const double* __restrict__ l1G,
                                                        const double* __restrict__ l2G, const double* __restrict__ gsq, WNext wn)
{
    // gsq != null (K-sharded run): the full (p,p) matrix of sum_k u^2 over ALL ranks' instances replaces the local sum
    __shared__ double scratch[GGL_NNORM * 4];
    if (spec_failed(skip)) return;
    if (gsq_rejected(gsq, p)) return;      // (K-sharded run: the all-reduced flag behind the packed sums)
    const size_t pp = (size_t)p * p;
    {   // grid-point dimension of a batch of independent problems (see k_theta_fgl)
        const size_t goff = (size_t)blockIdx.y * K * pp;
        Theta += goff; Omega += goff;
        if (X) X += goff;
        if (C) C += goff;
        if (OmegaPrev) OmegaPrev += goff;
        if (L) L += goff;
        if (l1G) { l1 = l1G[(size_t)blockIdx.y * K]; l2 = l2G[(size_t)blockIdx.y * K]; }
        if (partials) partials += (size_t)blockIdx.y * gridDim.x * GGL_NNORM;
    }
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    double acc[GGL_NNORM] = {0, 0, 0, 0, 0};
    if (e < pp) {
        const int i = (int)(e / p), j = (int)(e - (size_t)i * p);
        const bool offd = (i != j);
        double om[KMAX], x[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (k < K) {
                om[k] = Omega[(size_t)k * pp + e];
                x[k] = X ? X[(size_t)k * pp + e] : 0.0;
            }
        }
        double ss = 0.0;
        if (gsq) {
            ss = gsq[tri_index(i, j, p)];
        } else {
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                if (k < K) {
                    const double l = L ? L[(size_t)k * pp + e] : 0.0;
                    const double u = soft((om[k] + l) + x[k], l1);
                    ss += u * u;
                }
            }
        }
        const double a = fmax(sqrt(ss), l2);
        const double amul = a - l2;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (k < K) {
                const size_t o = (size_t)k * pp + e;
                const double l = L ? L[o] : 0.0;
                const double v = (om[k] + l) + x[k];
                const double th = offd ? soft(v, l1) * amul / a : v;
                Theta[o] = th;
                if (FUSE_DUAL) {
                    const double xn = x[k] + (om[k] - th);
                    X[o] = xn;
                    if (wn.S) C[o] = (th - xn) - wn.beta[k] * wn.S[o];      // W of the NEXT Omega-step (k_form_W_sym's arithmetic)
                    const double dp = om[k] - OmegaPrev[o];
                    acc[0] += om[k] * om[k];
                    acc[1] += th * th;
                    acc[2] += xn * xn;
                    acc[3] += (om[k] - th) * (om[k] - th);
                    acc[4] += dp * dp;
                } else if (C) {
                    C[o] = (th - x[k]) - om[k];
                }
            }
        }
    }
    if (FUSE_DUAL) {
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
        for (int v = 0; v < GGL_NNORM; ++v) acc[v] = wave_sum(acc[v]);
        if (lane == 0) {
#pragma unroll
            for (int v = 0; v < GGL_NNORM; ++v) scratch[wid * GGL_NNORM + v] = acc[v];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double* o = partials + (size_t)blockIdx.x * GGL_NNORM;
#pragma unroll
            for (int v = 0; v < GGL_NNORM; ++v)
                o[v] = (scratch[v] + scratch[GGL_NNORM + v]) + (scratch[2 * GGL_NNORM + v] + scratch[3 * GGL_NNORM + v]);
        }
    }
}

// The same per-element Theta-step with the K-column of an element split over the four waves of a workgroup: wave w holds
// the instances w*KQ .. (w+1)*KQ-1 of 64 consecutive elements (KQ values of Omega and X per lane instead of K: 8 waves
// per SIMD instead of 3 at K = 32, and four times the workgroups, so the tail of the last round of workgroups is short),
// the four partial sums of squares meet in LDS (fixed order).  Every access is still a full 512-byte wave row.
// NW waves per workgroup (4, 8 or 16): K <= NW * KQ.  The partial sums of squares of the NW waves are added as a fixed
// balanced tree (for NW = 4 the order it always had)
template <int NW> __device__ __forceinline__ double tree_sum(const double (&v)[NW])
{
    if constexpr (NW == 4) return (v[0] + v[1]) + (v[2] + v[3]);
    else {
        double h0[NW / 2], h1[NW / 2];
#pragma unroll
        for (int i = 0; i < NW / 2; ++i) { h0[i] = v[i]; h1[i] = v[NW / 2 + i]; }
        return tree_sum<NW / 2>(h0) + tree_sum<NW / 2>(h1);
    }
}

template <int KQ, bool FUSE_DUAL, int NW = 4>
__global__ __launch_bounds__(NW * 64) void k_theta_ggl_flat4(double* __restrict__ Theta, double* __restrict__ X,
                                                         double* __restrict__ C, const double* __restrict__ Omega,
                                                         const double* __restrict__ OmegaPrev,
                                                         const double* __restrict__ L, double l1, double l2,
                                                         double* __restrict__ partials, int K, int p,
                                                         const int* __restrict__ skip, const double* __restrict__ l1G,
                                                         const double* __restrict__ l2G, const double* __restrict__ gsq, WNext wn)
{
    __shared__ double ssh[NW][64];
    __shared__ double scratch[GGL_NNORM * NW];
    if (spec_failed(skip)) return;
    if (gsq_rejected(gsq, p)) return;      // (K-sharded run: the all-reduced flag behind the packed sums)
    const size_t pp = (size_t)p * p;
    {   // grid-point dimension of a batch of independent problems (see k_theta_fgl)
        const size_t goff = (size_t)blockIdx.y * K * pp;
        Theta += goff; Omega += goff;
        if (X) X += goff;
        if (C) C += goff;
        if (OmegaPrev) OmegaPrev += goff;
        if (L) L += goff;
        if (l1G) { l1 = l1G[(size_t)blockIdx.y * K]; l2 = l2G[(size_t)blockIdx.y * K]; }
        if (partials) partials += (size_t)blockIdx.y * gridDim.x * GGL_NNORM;
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int kb = wid * KQ;
    double acc[GGL_NNORM] = {0, 0, 0, 0, 0};
    for (int chunk = 0; chunk < FLAT4_CHUNKS; ++chunk) {
        const size_t e = ((size_t)blockIdx.x * FLAT4_CHUNKS + chunk) * 64 + lane;
        const bool live = e < pp;
        double om[KQ], x[KQ], u[KQ];
        double ss = 0.0;
        if (live) {
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                if (kb + q < K) {
                    const size_t o = (size_t)(kb + q) * pp + e;
                    om[q] = Omega[o];
                    x[q] = X ? X[o] : 0.0;
                }
            }
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                if (kb + q < K) {
                    const double l = L ? L[(size_t)(kb + q) * pp + e] : 0.0;
                    u[q] = (om[q] + l) + x[q];
                    const double sv = soft(u[q], l1);
                    ss += sv * sv;
                }
            }
        }
        if (chunk) __syncthreads();          // the previous chunk's sums have been read
        ssh[wid][lane] = ss;
        __syncthreads();
        if (live) {
            double sv[NW];
#pragma unroll
            for (int w = 0; w < NW; ++w) sv[w] = ssh[w][lane];
            const int i = (int)(e / p), j = (int)(e - (size_t)i * p);
            const double tot = gsq ? gsq[tri_index(i, j, p)] : tree_sum<NW>(sv);
            const double a = fmax(sqrt(tot), l2);
            const double amul = a - l2;
            const bool offd = (i != j);
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                if (kb + q < K) {
                    const size_t o = (size_t)(kb + q) * pp + e;
                    const double v = u[q];
                    const double th = offd ? soft(v, l1) * amul / a : v;
                    Theta[o] = th;
                    if (FUSE_DUAL) {
                        const double xn = x[q] + (om[q] - th);
                        X[o] = xn;
                        if (wn.S) C[o] = (th - xn) - wn.beta[kb + q] * wn.S[o];
                        const double dp = om[q] - OmegaPrev[o];
                        acc[0] += om[q] * om[q];
                        acc[1] += th * th;
                        acc[2] += xn * xn;
                        acc[3] += (om[q] - th) * (om[q] - th);
                        acc[4] += dp * dp;
                    } else if (C) {
                        C[o] = (th - x[q]) - om[q];
                    }
                }
            }
        }
    }
    if (FUSE_DUAL) {
#pragma unroll
        for (int v = 0; v < GGL_NNORM; ++v) acc[v] = wave_sum(acc[v]);
        if (lane == 0) {
#pragma unroll
            for (int v = 0; v < GGL_NNORM; ++v) scratch[wid * GGL_NNORM + v] = acc[v];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double* o = partials + (size_t)blockIdx.x * GGL_NNORM;
#pragma unroll
            for (int v = 0; v < GGL_NNORM; ++v) {
                double sv[NW];
#pragma unroll
                for (int w = 0; w < NW; ++w) sv[w] = scratch[w * GGL_NNORM + v];
                o[v] = tree_sum<NW>(sv);
            }
        }
    }
}

// The same with 16-byte accesses: a lane holds TWO consecutive elements (p^2 even), a wave row is 1 KiB.  8-byte
// accesses run at 0.54-0.70x the 16-byte rate on this chip (MI355X_MICROARCH.md), and this kernel is nothing but accesses.
// Same 128 elements per workgroup as the scalar form (one chunk of 2 x 64), so the partial-sum layout is unchanged.
template <int KQ, bool FUSE_DUAL, int NW = 4>
__global__ __launch_bounds__(NW * 64) void k_theta_ggl_flat4v(double* __restrict__ Theta, double* __restrict__ X,
                                                          double* __restrict__ C, const double* __restrict__ Omega,
                                                          const double* __restrict__ OmegaPrev,
                                                          const double* __restrict__ L, double l1, double l2,
                                                          double* __restrict__ partials, int K, int p,
                                                          const int* __restrict__ skip, const double* __restrict__ l1G,
                                                          const double* __restrict__ l2G, const double* __restrict__ gsq, WNext wn)
{
    static_assert(FLAT4_CHUNKS == 2, "128 elements per workgroup");
    __shared__ double2 ssh[NW][64];
    __shared__ double scratch[GGL_NNORM * NW];
    if (spec_failed(skip)) return;
    if (gsq_rejected(gsq, p)) return;      // (K-sharded run: the all-reduced flag behind the packed sums)
    const size_t pp = (size_t)p * p;
    {
        const size_t goff = (size_t)blockIdx.y * K * pp;
        Theta += goff; Omega += goff;
        if (X) X += goff;
        if (C) C += goff;
        if (OmegaPrev) OmegaPrev += goff;
        if (L) L += goff;
        if (l1G) { l1 = l1G[(size_t)blockIdx.y * K]; l2 = l2G[(size_t)blockIdx.y * K]; }
        if (partials) partials += (size_t)blockIdx.y * gridDim.x * GGL_NNORM;
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int kb = wid * KQ;
    double acc[GGL_NNORM] = {0, 0, 0, 0, 0};
    const size_t e = ((size_t)blockIdx.x * 64 + lane) * 2;
    // p even: e + 1 < pp as well, in the same row, and every pair is 16-byte aligned.  Odd p (round 6; rounds 2-5 ran the
    // one-element kernel there, 0.54-0.70x the access rate): the pairs of every other instance start on an 8-byte boundary
    // (the hardware takes that), a pair may straddle two rows, and the last element of a matrix has no partner -- that one
    // lane moves single elements and its second half counts as zero everywhere.
    const bool live = e < pp, live1 = e + 1 < pp;
    double2 om[KQ], x[KQ], u[KQ];
    double2 ss = {0.0, 0.0};
    auto ld2 = [live1](const double* q) { return live1 ? *reinterpret_cast<const double2*>(q) : double2{*q, 0.0}; };
    auto st2 = [live1](double* q, const double2 v) { if (live1) *reinterpret_cast<double2*>(q) = v; else *q = v.x; };
    if (live) {
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            if (kb + q < K) {
                const size_t o = (size_t)(kb + q) * pp + e;
                om[q] = ld2(Omega + o);
                x[q] = X ? ld2(X + o) : double2{0.0, 0.0};
            }
        }
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            if (kb + q < K) {
                const double2 l = L ? ld2(L + (size_t)(kb + q) * pp + e) : double2{0.0, 0.0};
                u[q].x = (om[q].x + l.x) + x[q].x;
                u[q].y = (om[q].y + l.y) + x[q].y;
                const double s0 = soft(u[q].x, l1), s1 = soft(u[q].y, l1);
                ss.x += s0 * s0;
                ss.y += s1 * s1;
            }
        }
    }
    ssh[wid][lane] = ss;
    __syncthreads();
    if (live) {
        double2 tot;
        if (gsq) {
            const int gi = (int)(e / p), gj = (int)(e - (size_t)gi * p);
            const int gj1 = (gj + 1 == p) ? 0 : gj + 1, gi1 = (gj + 1 == p) ? gi + 1 : gi;
            tot.x = gsq[tri_index(gi, gj, p)];
            tot.y = live1 ? gsq[tri_index(gi1, gj1, p)] : 0.0;
        } else {
            double sx[NW], sy[NW];
#pragma unroll
            for (int w = 0; w < NW; ++w) { sx[w] = ssh[w][lane].x; sy[w] = ssh[w][lane].y; }
            tot.x = tree_sum<NW>(sx);
            tot.y = tree_sum<NW>(sy);
        }
        const double a0 = fmax(sqrt(tot.x), l2), a1 = fmax(sqrt(tot.y), l2);
        const double m0 = a0 - l2, m1 = a1 - l2;
        const int i0 = (int)(e / p), j0 = (int)(e - (size_t)i0 * p);
        const bool off0 = (i0 != j0);
        // (p even and e even: e + 1 is the next column of the same row; odd p: possibly the first column of the next one)
        const bool off1 = (j0 + 1 == p) || (i0 != j0 + 1);       // (element (i0 + 1, 0) is never on the diagonal)
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            if (kb + q < K) {
                const size_t o = (size_t)(kb + q) * pp + e;
                double2 th;
                th.x = off0 ? soft(u[q].x, l1) * m0 / a0 : u[q].x;
                th.y = off1 ? soft(u[q].y, l1) * m1 / a1 : u[q].y;
                st2(Theta + o, th);
                if (FUSE_DUAL) {
                    double2 xn;
                    xn.x = x[q].x + (om[q].x - th.x);
                    xn.y = x[q].y + (om[q].y - th.y);
                    st2(X + o, xn);
                    if (wn.S) {
                        const double2 sv = ld2(wn.S + o);
                        const double bk = wn.beta[kb + q];
                        double2 wv;
                        wv.x = (th.x - xn.x) - bk * sv.x;
                        wv.y = (th.y - xn.y) - bk * sv.y;
                        st2(C + o, wv);
                    }
                    const double2 op = ld2(OmegaPrev + o);
                    const double d0 = om[q].x - op.x, d1 = om[q].y - op.y;
                    acc[0] += om[q].x * om[q].x + om[q].y * om[q].y;
                    acc[1] += th.x * th.x + th.y * th.y;
                    acc[2] += xn.x * xn.x + xn.y * xn.y;
                    acc[3] += (om[q].x - th.x) * (om[q].x - th.x) + (om[q].y - th.y) * (om[q].y - th.y);
                    acc[4] += d0 * d0 + d1 * d1;
                } else if (C) {
                    double2 cv;
                    cv.x = (th.x - x[q].x) - om[q].x;
                    cv.y = (th.y - x[q].y) - om[q].y;
                    st2(C + o, cv);
                }
            }
        }
    }
    if (FUSE_DUAL) {
#pragma unroll
        for (int v = 0; v < GGL_NNORM; ++v) acc[v] = wave_sum(acc[v]);
        if (lane == 0) {
#pragma unroll
            for (int v = 0; v < GGL_NNORM; ++v) scratch[wid * GGL_NNORM + v] = acc[v];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double* o = partials + (size_t)blockIdx.x * GGL_NNORM;
#pragma unroll
            for (int v = 0; v < GGL_NNORM; ++v) {
                double sv[NW];
#pragma unroll
                for (int w = 0; w < NW; ++w) sv[w] = scratch[w * GGL_NNORM + v];
                o[v] = tree_sum<NW>(sv);
            }
        }
    }
}

// Few elements, many instances: a workgroup of 16 waves takes 16 consecutive elements (a 128-byte segment of every
// instance) and ALL K <= 64 * KQ instances of them; lanes 0-15 / 16-31 / 32-47 / 48-63 of wave w hold the instance groups
// 4w .. 4w+3 (KQ instances each).  The sums of squares meet by two shuffles inside the wave and one LDS round over the waves,
// in a fixed order.  p^2 / 16 workgroups where the 128-element kernels above would launch p^2 / 128.
template <int KQ, bool FUSE_DUAL>
__global__ __launch_bounds__(1024) void k_theta_ggl_flat16(double* __restrict__ Theta, double* __restrict__ X,
                                                           double* __restrict__ C, const double* __restrict__ Omega,
                                                           const double* __restrict__ OmegaPrev,
                                                           const double* __restrict__ L, double l1, double l2,
                                                           double* __restrict__ partials, int K, int p,
                                                           const int* __restrict__ skip, const double* __restrict__ gsq, WNext wn)
{
    constexpr int NW = 16;
    __shared__ double ssh[NW][16];
    __shared__ double scratch[GGL_NNORM * NW];
    if (spec_failed(skip)) return;
    if (gsq_rejected(gsq, p)) return;      // (K-sharded run: the all-reduced flag behind the packed sums)
    const size_t pp = (size_t)p * p;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int el = lane & 15, kb = (wid * 4 + (lane >> 4)) * KQ;
    const size_t e = (size_t)blockIdx.x * 16 + el;
    const bool live = e < pp;
    double acc[GGL_NNORM] = {0, 0, 0, 0, 0};
    double om[KQ], x[KQ], u[KQ];
    double ss = 0.0;
    if (live) {
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            if (kb + q < K) {
                const size_t o = (size_t)(kb + q) * pp + e;
                om[q] = Omega[o];
                x[q] = X ? X[o] : 0.0;
            }
        }
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            if (kb + q < K) {
                const double l = L ? L[(size_t)(kb + q) * pp + e] : 0.0;
                u[q] = (om[q] + l) + x[q];
                const double sv = soft(u[q], l1);
                ss += sv * sv;
            }
        }
    }
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    if (lane < 16) ssh[wid][lane] = ss;
    __syncthreads();
    if (live) {
        double sv[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) sv[w] = ssh[w][el];
        const int i = (int)(e / p), j = (int)(e - (size_t)i * p);
        const double tot = gsq ? gsq[tri_index(i, j, p)] : tree_sum<NW>(sv);
        const double a = fmax(sqrt(tot), l2);
        const double amul = a - l2;
        const bool offd = (i != j);
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            if (kb + q < K) {
                const size_t o = (size_t)(kb + q) * pp + e;
                const double v = u[q];
                const double th = offd ? soft(v, l1) * amul / a : v;
                Theta[o] = th;
                if (FUSE_DUAL) {
                    const double xn = x[q] + (om[q] - th);
                    X[o] = xn;
                    if (wn.S) C[o] = (th - xn) - wn.beta[kb + q] * wn.S[o];
                    const double dp = om[q] - OmegaPrev[o];
                    acc[0] += om[q] * om[q];
                    acc[1] += th * th;
                    acc[2] += xn * xn;
                    acc[3] += (om[q] - th) * (om[q] - th);
                    acc[4] += dp * dp;
                } else if (C) {
                    C[o] = (th - x[q]) - om[q];
                }
            }
        }
    }
    if (FUSE_DUAL) {
#pragma unroll
        for (int v = 0; v < GGL_NNORM; ++v) acc[v] = wave_sum(acc[v]);
        if (lane == 0) {
#pragma unroll
            for (int v = 0; v < GGL_NNORM; ++v) scratch[wid * GGL_NNORM + v] = acc[v];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double* o = partials + (size_t)blockIdx.x * GGL_NNORM;
#pragma unroll
            for (int v = 0; v < GGL_NNORM; ++v) {
                double sv[NW];
#pragma unroll
                for (int w = 0; w < NW; ++w) sv[w] = scratch[w * GGL_NNORM + v];
                o[v] = tree_sum<NW>(sv);
            }
        }
    }
}

template <int KQ>
static void launch_flat16(hipStream_t st, double* Theta, double* X, double* C, const double* Omega, const double* OmegaPrev,
                          const double* L, double l1, double l2, int fuse_dual, double* partials, int K, int p,
                          const int* skip, const double* gsq, WNext wn = WNext())
{
    dim3 grid((unsigned)(((size_t)p * p + 15) / 16)), blk(1024);
    if (fuse_dual)
        hipLaunchKernelGGL((k_theta_ggl_flat16<KQ, true>), grid, blk, 0, st, Theta, X, C, Omega, OmegaPrev, L, l1, l2, partials, K, p, skip, gsq, wn);
    else
        hipLaunchKernelGGL((k_theta_ggl_flat16<KQ, false>), grid, blk, 0, st, Theta, X, C, Omega, OmegaPrev, L, l1, l2, partials, K, p, skip, gsq, wn);
}

static inline int flat4_blocks(int p) { return (int)(((size_t)p * p + 64 * FLAT4_CHUNKS - 1) / (64 * FLAT4_CHUNKS)); }

template <int KQ, int NW = 4, bool VEC_OK = true>
static void launch_flat4(hipStream_t st, double* Theta, double* X, double* C, const double* Omega,
                         const double* OmegaPrev, const double* L, double l1, double l2, int fuse_dual, double* partials,
                         int K, int p, const int* skip, int G = 1, const double* l1G = nullptr, const double* l2G = nullptr,
                         const double* gsq = nullptr, WNext wn = WNext())
{
    dim3 grid(flat4_blocks(p), G), blk(NW * 64);
    if constexpr (VEC_OK) {
        {                            // two consecutive elements per lane, 16-byte accesses (any p since round 6)
            if (fuse_dual)
                hipLaunchKernelGGL((k_theta_ggl_flat4v<KQ, true, NW>), grid, blk, 0, st, Theta, X, C, Omega, OmegaPrev, L, l1, l2, partials, K, p, skip, l1G, l2G, gsq, wn);
            else
                hipLaunchKernelGGL((k_theta_ggl_flat4v<KQ, false, NW>), grid, blk, 0, st, Theta, X, C, Omega, OmegaPrev, L, l1, l2, partials, K, p, skip, l1G, l2G, gsq, wn);
            return;
        }
    }
    if (fuse_dual)
        hipLaunchKernelGGL((k_theta_ggl_flat4<KQ, true, NW>), grid, blk, 0, st, Theta, X, C, Omega, OmegaPrev, L, l1, l2, partials, K, p, skip, l1G, l2G, gsq, wn);
    else
        hipLaunchKernelGGL((k_theta_ggl_flat4<KQ, false, NW>), grid, blk, 0, st, Theta, X, C, Omega, OmegaPrev, L, l1, l2, partials, K, p, skip, l1G, l2G, gsq, wn);
}

// the K-column of an element over 4, 8 or 16 waves: K <= 16: 4 x 4, <= 32: 4 x 8, <= 64: 8 x 8, <= 128: 16 x 8 (16-byte
// accesses throughout), <= 256: 16 x 16 with 8-byte accesses (two elements per lane would need 192 registers per lane, a
// 16-wave workgroup has 128)
// Which Theta kernel the last launch_theta_pair / launch_theta_batch of this process ran (ggl_last_dispatch; the parity tests
// assert the dispatch of every BASELINE configuration): 0 GGL tile pairs; 100 + KMAX per-element, K-column in one thread;
// 100 * KQ + NW per-element, K-column over NW waves of KQ values per lane (404, 408: four waves; 808 / 816 / 1616: K <= 64 /
// 128 / 256); 3000 + elements per workgroup: FGL per element (3128 / 3064); 10000 + 100 * KQ + 16: the same for few elements and many instances, 16 elements per workgroup of 16 waves
// (10216: K <= 128, 10416: K <= 256); 2000 + tile edge: FGL Condat tiles.
static int g_theta_kernel = -1;
int theta_last_kernel() { return g_theta_kernel; }

static void launch_flat4_any(hipStream_t st, double* Theta, double* X, double* C, const double* Omega,
                             const double* OmegaPrev, const double* L, double l1, double l2, int fuse_dual,
                             double* partials, int K, int p, const int* skip, int G, const double* l1G, const double* l2G,
                             const double* gsq, WNext wn = WNext())
{
    if (use_flat16(K, p, 2, G)) {
        g_theta_kernel = K <= 128 ? 10216 : 10416;
        if (K <= 128) launch_flat16<2>(st, Theta, X, C, Omega, OmegaPrev, L, l1, l2, fuse_dual, partials, K, p, skip, gsq, wn);
        else launch_flat16<4>(st, Theta, X, C, Omega, OmegaPrev, L, l1, l2, fuse_dual, partials, K, p, skip, gsq, wn);
        return;
    }
#define GGL_F4(...) launch_flat4<__VA_ARGS__>(st, Theta, X, C, Omega, OmegaPrev, L, l1, l2, fuse_dual, partials, K, p, skip, G, l1G, l2G, gsq, wn)
    g_theta_kernel = K <= 16 ? 404 : K <= 32 ? 408 : K <= 64 ? 808 : K <= 128 ? 816 : 1616;
    if (K <= 16) GGL_F4(4);
    else if (K <= 32) GGL_F4(8);
    else if (K <= 64) GGL_F4(8, 8);
    else if (K <= 128) GGL_F4(8, 16);
    else GGL_F4(16, 16, false);
#undef GGL_F4
}

template <int KMAX>
static void launch_flat(hipStream_t st, double* Theta, double* X, double* C, const double* Omega,
                        const double* OmegaPrev, const double* L, double l1, double l2, int fuse_dual, double* partials,
                        int K, int p, const int* skip, int G = 1, const double* l1G = nullptr, const double* l2G = nullptr,
                        const double* gsq = nullptr, WNext wn = WNext())
{
    dim3 grid(flat_blocks(p), G), blk(256);
    g_theta_kernel = 100 + KMAX;
    if (fuse_dual)
        hipLaunchKernelGGL((k_theta_ggl_flat<KMAX, true>), grid, blk, 0, st, Theta, X, C, Omega, OmegaPrev, L, l1, l2, partials, K, p, skip, l1G, l2G, gsq, wn);
    else
        hipLaunchKernelGGL((k_theta_ggl_flat<KMAX, false>), grid, blk, 0, st, Theta, X, C, Omega, OmegaPrev, L, l1, l2, partials, K, p, skip, l1G, l2G, gsq, wn);
}

hipError_t launch_theta_pair(hipStream_t st, int reg, double* Theta, double* X, double* C, const double* Omega,
                             const double* OmegaPrev, const double* L, double l1, double l2,
                             const double* groupsq, double* sqwork, int fuse_dual, double* partials, int K, int p,
                             int flat, const int* skip, WNext wn, int* wn_done)
{
    // wn (S, beta given; fuse_dual): the per-element GGL kernels also write C = (Theta - X_new) - beta_k S, the W of the NEXT
    // Omega-step (admm_solver.py:180), from the values they hold anyway; *wn_done = 1 when the kernel launched did so
    if (wn_done) *wn_done = 0;
    if (!fuse_dual || reg != 1) wn = WNext();
    // per-element kernels (exactly symmetric state): the group sums are the element's own K-column, or -- K-sharded run --
    // the all-reduced FULL matrix `groupsq` (launch_group_sums_full on every rank)
    if (reg == 1 && K <= GGL_FLAT_MAX_K && use_flat4(K, flat)) {
        launch_flat4_any(st, Theta, X, C, Omega, OmegaPrev, L, l1, l2, fuse_dual, partials, K, p, skip, 1, nullptr, nullptr, groupsq, wn);
        if (wn_done && wn.S) *wn_done = 1;
        return hipGetLastError();
    }
    if (reg == 1 && flat && K <= GGL_FLAT1_MAX_K) {
        if (K <= 8) launch_flat<8>(st, Theta, X, C, Omega, OmegaPrev, L, l1, l2, fuse_dual, partials, K, p, skip, 1, nullptr, nullptr, groupsq, wn);
        else if (K <= 16) launch_flat<16>(st, Theta, X, C, Omega, OmegaPrev, L, l1, l2, fuse_dual, partials, K, p, skip, 1, nullptr, nullptr, groupsq, wn);
        else launch_flat<32>(st, Theta, X, C, Omega, OmegaPrev, L, l1, l2, fuse_dual, partials, K, p, skip, 1, nullptr, nullptr, groupsq, wn);
        if (wn_done && wn.S) *wn_done = 1;
        return hipGetLastError();
    }
    if (reg == 1) {
        const int T = ntiles(p, PT);
        const int kc = ggl_chunks(K, p), klen = ggl_chunk_len(K, p);
        dim3 grid(T * (T + 1) / 2, kc), blk(PT, PTY);
        const double* sq = groupsq;
        int nsq = 1;
        g_theta_kernel = 0;
        if (!sq) {   // single rank: per-chunk sums of squares, summed on the fly by the second kernel
            if (!sqwork) return hipErrorInvalidValue;
            hipLaunchKernelGGL(k_group_partial, grid, blk, 0, st, sqwork, Omega, L, X, l1, K, p, klen);
            sq = sqwork;
            nsq = kc;
        }
        const int packed = groupsq ? 1 : 0;
        if (fuse_dual)
            hipLaunchKernelGGL(k_theta_ggl<true>, grid, blk, 0, st, Theta, X, C, Omega, OmegaPrev, L, l1, l2, sq, nsq, partials, K, p, klen, skip, packed);
        else
            hipLaunchKernelGGL(k_theta_ggl<false>, grid, blk, 0, st, Theta, X, C, Omega, OmegaPrev, L, l1, l2, sq, nsq, partials, K, p, klen, skip, packed);
        return hipGetLastError();
    }
    if (K > FGL_MAX_K_TD8) return hipErrorInvalidValue;
    if (flat) return launch_fgl_flat_any(st, Theta, X, C, Omega, OmegaPrev, L, l1, l2, fuse_dual, partials, K, p, skip, 1, nullptr, nullptr);
    g_theta_kernel = 2000 + fgl_tile(K);
    if (fgl_tile(K) == 16)
        return launch_fgl_td<16>(st, Theta, X, C, Omega, OmegaPrev, L, l1, l2, fuse_dual, partials, K, p, skip);
    return launch_fgl_td<8>(st, Theta, X, C, Omega, OmegaPrev, L, l1, l2, fuse_dual, partials, K, p, skip);
}

// 3000 + elements per workgroup: the per-element FGL kernel (ggl_last_dispatch)
static hipError_t launch_fgl_flat_any(hipStream_t st, double* Theta, double* X, double* C, const double* Omega,
                                      const double* OmegaPrev, const double* L, double l1, double l2, int fuse_dual,
                                      double* partials, int K, int p, const int* skip, int G, const double* l1G, const double* l2G)
{
    const int nt = fgl_flat_nt(K);
    g_theta_kernel = 3000 + nt;
    if (nt == 128) return launch_fgl_flat<128>(st, Theta, X, C, Omega, OmegaPrev, L, l1, l2, fuse_dual, partials, K, p, skip, G, l1G, l2G);
    return launch_fgl_flat<64>(st, Theta, X, C, Omega, OmegaPrev, L, l1, l2, fuse_dual, partials, K, p, skip, G, l1G, l2G);
}

// Theta-step of G independent problems of K instances each in one launch (stacks (G*K,p,p); thresholds of problem g at
// l1G[g*K], l2G[g*K]; partial sums [g][theta_partial_blocks]).  Exactly symmetric states only (the callers start from
// symmetric points): GGL runs the per-element kernel (K <= GGL_FLAT_MAX_K), FGL the Condat tile kernel.
hipError_t launch_theta_batch(hipStream_t st, int reg, double* Theta, double* X, double* C, const double* Omega,
                              const double* OmegaPrev, const double* L, const double* l1G, const double* l2G,
                              int fuse_dual, double* partials, int G, int K, int p, const int* skip)
{
    if (reg == 1) {
        if (K > GGL_FLAT_MAX_K) return hipErrorInvalidValue;
        if (use_flat4(K, 2)) {       // partial-sum rows: theta_partial_blocks(p, reg, K, 2) per problem
            launch_flat4_any(st, Theta, X, C, Omega, OmegaPrev, L, 0.0, 0.0, fuse_dual, partials, K, p, skip, G, l1G, l2G, nullptr);
            return hipGetLastError();
        }
        if (K <= 8) launch_flat<8>(st, Theta, X, C, Omega, OmegaPrev, L, 0.0, 0.0, fuse_dual, partials, K, p, skip, G, l1G, l2G);
        else if (K <= 16) launch_flat<16>(st, Theta, X, C, Omega, OmegaPrev, L, 0.0, 0.0, fuse_dual, partials, K, p, skip, G, l1G, l2G);
        else launch_flat<32>(st, Theta, X, C, Omega, OmegaPrev, L, 0.0, 0.0, fuse_dual, partials, K, p, skip, G, l1G, l2G);
        return hipGetLastError();
    }
    if (K > FGL_MAX_K_TD8) return hipErrorInvalidValue;
    // (the batched grids start from exactly symmetric points: always the per-element kernel; theta_partial_blocks(.., 2, G))
    return launch_fgl_flat_any(st, Theta, X, C, Omega, OmegaPrev, L, 0.0, 0.0, fuse_dual, partials, K, p, skip, G, l1G, l2G);
}

hipError_t launch_prox_p(hipStream_t st, int reg, double* out, const double* V, double l1, double l2, int K, int p,
                         double* sqwork)
{
    return launch_theta_pair(st, reg, out, nullptr, nullptr, V, nullptr, nullptr, l1, l2, nullptr, sqwork, 0, nullptr, K,
                             p, 0, nullptr);
}

// ---------------------------------------------------------------------------------------------
// P_val (ggl_helper.py:162-176): 2 * sum_{i<j} [ l1 |v|_1 + l2 |v|_2 ]  (GGL)
//                                2 * sum_{i<j} [ l1 |v|_1 + l2 sum_k |v_{k+1}-v_k| ]  (FGL)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pval(int reg, const double* __restrict__ Theta, double l1, double l2, int K,
                                              int p, double* __restrict__ partials)
{
    __shared__ double scratch[4];
    const int T = (p + PT - 1) / PT;
    int I, J;
    decode_pair(blockIdx.x, T, I, J);
    const bool diag = (I == J);
    const int I0 = I * PT, J0 = J * PT;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const size_t pp = (size_t)p * p;
    double tot = 0.0;
#pragma unroll
    for (int q = 0; q < PQ; ++q) {
        const int r = ty + PTY * q;
        const bool ok = (I0 + r < p) && (J0 + tx < p) && (!diag || r < tx);
        if (!ok) continue;
        const size_t off = (size_t)(I0 + r) * p + (J0 + tx);
        double s1 = 0.0, s2 = 0.0, prev = 0.0;
        for (int k = 0; k < K; ++k) {
            const double v = Theta[(size_t)k * pp + off];
            s1 += fabs(v);
            if (reg == 1) s2 += v * v;
            else if (k > 0) s2 += fabs(v - prev);
            prev = v;
        }
        tot += l1 * s1 + l2 * (reg == 1 ? sqrt(s2) : s2);
    }
    tot = wave_sum(tot);
    const int tid = ty * PT + tx;
    if ((tid & 63) == 0) scratch[tid >> 6] = tot;
    __syncthreads();
    if (tid == 0) partials[blockIdx.x] = 2.0 * ((scratch[0] + scratch[1]) + (scratch[2] + scratch[3]));
}

void launch_pval(hipStream_t st, int reg, const double* Theta, double l1, double l2, int K, int p, double* partials)
{
    const int T = ntiles(p, PT);
    hipLaunchKernelGGL(k_pval, dim3(T * (T + 1) / 2), dim3(PT, PTY), 0, st, reg, Theta, l1, l2, K, p, partials);
}

// ---------------------------------------------------------------------------------------------
// n independent K-vectors (operator-level parity entry points)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_vec_prox(int mode, const double* __restrict__ Y, double* __restrict__ out, int n,
                                                 int K, double l1, double l2)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];   // [K][64]
    const int v = blockIdx.x * 64 + threadIdx.x;
    if (v >= n) return;
    double* y = lds + threadIdx.x;
    for (int k = 0; k < K; ++k) y[k * 64] = Y[(size_t)v * K + k];
    if (mode == 0) {                      // prox_tv
        condat_scan(y, 64, K, l1, [](double v) { return v; });
    } else if (mode == 1 || mode == 2) {  // prox_2norm / prox_phi_ggl
        double ss = 0.0;
        for (int k = 0; k < K; ++k) {
            double u = y[k * 64];
            if (mode == 2) u = soft(u, l1);
            y[k * 64] = u;
            ss += u * u;
        }
        const double l = (mode == 1) ? l1 : l2;
        const double a = fmax(sqrt(ss), l);
        for (int k = 0; k < K; ++k) y[k * 64] = y[k * 64] * (a - l) / a;
    } else {                              // prox_phi_fgl
        condat_scan(y, 64, K, l2, [l1](double v) { return soft(v, l1); });
    }
    for (int k = 0; k < K; ++k) out[(size_t)v * K + k] = y[k * 64];
}

hipError_t launch_vec_prox(hipStream_t st, int mode, const double* Y, double* out, int n, int K, double l1, double l2)
{
    const size_t lds = (size_t)K * 64 * sizeof(double);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute((const void*)k_vec_prox, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_vec_prox, dim3((n + 63) / 64), dim3(64), lds, st, mode, Y, out, n, K, l1, l2);
    return hipGetLastError();
}

}  // namespace ggl
