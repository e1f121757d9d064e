// ext_ADMM_MGL (solver/ext_admm_solver.py:18-323): Group Graphical Lasso over K instances of DIFFERENT dimension p_k.
//
// Layout: the instances live in the ctx's (K,p,p) stacks padded to p = max_k p_k -- instance k is the leading
// (p_k,p_k) block of its slot and the remaining diagonal is an identity block (S = Omega = Theta = Lambda = I there,
// duals zero).  Every step of the iteration acts block-wise on a block-diagonal matrix (matrix functions, the
// elementwise prox, the group shrink that only touches listed entries), so the padding stays a decoupled fixed point
// and never mixes with the data; the stopping-test sums skip it (i, j < p_k only).
//
// Per iteration, after the shared Omega-step (and before / after the shared L-step when latent):
//   k_ext_theta   V = (Omega + L + X0 + Lambda - X1)/2, Theta = prox_od_1norm(V, lambda1_k/(2 rho))      (:207-210)
//                 Z = Theta + X1 into the other Lambda buffer                                            (:221-223)
//                 not latent: X0 += Omega - Theta (:229) and its share of the stopping-test sums; latent: C = Theta-X0-Omega
//   k_ext_group   prox_2norm_G (:394-453): gather the listed entries of a group over the instances that hold the
//                 pair, shrink by lambda2/rho * sqrt(group size), scatter to (i,j) and (j,i)
//   k_ext_dual    X1 += Theta - Lambda (:230), latent: X0 += Omega - Theta + L; the rest of the sums        (:325-345)
#include "common.hpp"
#include "kernels.hpp"

namespace ggl {

static constexpr int XT = 256, XE = 4, XCHUNK = XT * XE;

int ext_blocks(int p) { return (int)(((size_t)p * p + XCHUNK - 1) / XCHUNK); }

// sums: [0] |Omega|^2 + |Lambda|^2, [1] |Theta - L|^2 + |Theta|^2, [2] |X0|^2 + |X1|^2,
//       [3] |Omega - Theta + L|^2 + |Lambda - Theta|^2, [4] |Omega - Omega_prev|^2 + |Lambda - Lambda_prev|^2
template <bool LATENT>
__global__ __launch_bounds__(XT) void k_ext_theta(double* __restrict__ Theta, double* __restrict__ X0,
                                                  double* __restrict__ Znew, double* __restrict__ C,
                                                  const double* __restrict__ Omega, const double* __restrict__ OmegaPrev,
                                                  const double* __restrict__ L, const double* __restrict__ Lambda,
                                                  const double* __restrict__ X1, const double* __restrict__ l1K,
                                                  const int* __restrict__ pk, double* __restrict__ partials, int p,
                                                  const int* __restrict__ skip)
{
    __shared__ double scratch[GGL_NNORM * (XT / 64)];
    if (spec_failed(skip)) return;
    const int k = blockIdx.y;
    const size_t pp = (size_t)p * p, base = (size_t)k * pp;
    const double thr = l1K[k];
    const int pdim = pk[k];
    double acc[GGL_NNORM] = {0, 0, 0, 0, 0};
    size_t i = (size_t)blockIdx.x * XCHUNK + threadIdx.x;
#pragma unroll
    for (int e = 0; e < XE; ++e, i += XT) {
        if (i < pp) {
            const int r = (int)(i / p), c = (int)(i - (size_t)r * p);
            const size_t o = base + i;
            const double om = Omega[o], x0 = X0[o], x1 = X1[o];
            const double l = LATENT ? L[o] : 0.0;
            const double v = ((((om + l) + x0) + Lambda[o]) - x1) * 0.5;
            const double th = (r == c) ? v : soft(v, thr);
            Theta[o] = th;
            Znew[o] = th + x1;
            if (LATENT) {
                C[o] = (th - x0) - om;
            } else {
                const double res = om - th;
                const double xn = x0 + res;
                X0[o] = xn;
                if (r < pdim && c < pdim) {
                    const double dp = om - OmegaPrev[o];
                    acc[0] += om * om;
                    acc[1] += 2.0 * (th * th);
                    acc[2] += xn * xn;
                    acc[3] += res * res;
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

void launch_ext_theta(hipStream_t st, double* Theta, double* X0, double* Znew, double* C, const double* Omega,
                      const double* OmegaPrev, const double* L, const double* Lambda, const double* X1, const double* l1K,
                      const int* pk, int latent, double* partials, int K, int p, const int* skip)
{
    dim3 grid(ext_blocks(p), K), blk(XT);
    if (latent)
        hipLaunchKernelGGL(k_ext_theta<true>, grid, blk, 0, st, Theta, X0, Znew, C, Omega, OmegaPrev, L, Lambda, X1, l1K, pk,
                           partials, p, skip);
    else
        hipLaunchKernelGGL(k_ext_theta<false>, grid, blk, 0, st, Theta, X0, Znew, C, Omega, OmegaPrev, L, Lambda, X1, l1K, pk,
                           partials, p, skip);
}

// Gt: [2][K][L] (instance-major, groups contiguous: neighbouring threads read neighbouring words); -1 = the instance does
// not hold the pair.  gsize[l] = instances that do.  One thread per group; the listed entries are distinct over the
// whole array (checked on the host), so the groups are independent.
// blockIdx.y: problem g of a batch of independent problems with the SAME group structure (a model-selection grid): its K
// instances are the slots g*K .. of the stack, its threshold l2K[g*K] (one value per instance slot, equal within a problem)
__global__ __launch_bounds__(256) void k_ext_group(double* __restrict__ Lam, const int* __restrict__ Gt,
                                                   const int* __restrict__ gsize, const double* __restrict__ l2K, int L, int K,
                                                   int p, const int* __restrict__ skip)
{
    if (spec_failed(skip)) return;
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= L) return;
    const size_t pp = (size_t)p * p;
    Lam += (size_t)blockIdx.y * K * pp;
    const double l2 = l2K[(size_t)blockIdx.y * K];
    const int* gi = Gt + l;
    const int* gj = Gt + (size_t)K * L + l;
    double ss = 0.0;
    for (int k = 0; k < K; ++k) {
        const int i = gi[(size_t)k * L];
        if (i < 0) continue;
        const double v = Lam[(size_t)k * pp + (size_t)i * p + gj[(size_t)k * L]];
        ss += v * v;
    }
    const double lam = l2 * sqrt((double)gsize[l]);
    const double a = fmax(sqrt(ss), lam);
    const double amul = a - lam;
    for (int k = 0; k < K; ++k) {
        const int i = gi[(size_t)k * L];
        if (i < 0) continue;
        const int j = gj[(size_t)k * L];
        double* m = Lam + (size_t)k * pp;
        const double z = m[(size_t)i * p + j] * amul / a;
        m[(size_t)i * p + j] = z;
        m[(size_t)j * p + i] = z;
    }
}

void launch_ext_group(hipStream_t st, double* Lam, const int* Gt, const int* gsize, const double* l2K, int L, int K, int p,
                      const int* skip, int nprob)
{
    if (L <= 0) return;
    hipLaunchKernelGGL(k_ext_group, dim3((L + 255) / 256, nprob), dim3(256), 0, st, Lam, Gt, gsize, l2K, L, K, p, skip);
}

template <bool LATENT>
__global__ __launch_bounds__(XT) void k_ext_dual(double* __restrict__ X0, double* __restrict__ X1,
                                                 const double* __restrict__ Omega, const double* __restrict__ OmegaPrev,
                                                 const double* __restrict__ Theta, const double* __restrict__ L,
                                                 const double* __restrict__ Lam, const double* __restrict__ LamPrev,
                                                 const int* __restrict__ pk, double* __restrict__ partials, int p,
                                                 const int* __restrict__ skip)
{
    __shared__ double scratch[GGL_NNORM * (XT / 64)];
    if (spec_failed(skip)) return;
    const int k = blockIdx.y;
    const size_t pp = (size_t)p * p, base = (size_t)k * pp;
    const int pdim = pk[k];
    double acc[GGL_NNORM] = {0, 0, 0, 0, 0};
    size_t i = (size_t)blockIdx.x * XCHUNK + threadIdx.x;
#pragma unroll
    for (int e = 0; e < XE; ++e, i += XT) {
        if (i < pp) {
            const int r = (int)(i / p), c = (int)(i - (size_t)r * p);
            const size_t o = base + i;
            const double th = Theta[o], la = Lam[o];
            const double d1 = th - la;
            const double x1n = X1[o] + d1;
            X1[o] = x1n;
            const bool in = r < pdim && c < pdim;
            const double dl = la - LamPrev[o];
            if (in) {
                acc[0] += la * la;
                acc[2] += x1n * x1n;
                acc[3] += d1 * d1;
                acc[4] += dl * dl;
            }
            if (LATENT) {
                const double om = Omega[o], l = L[o];
                const double res = (om - th) + l;
                const double xn = X0[o] + res;
                X0[o] = xn;
                if (in) {
                    const double dp = om - OmegaPrev[o];
                    acc[0] += om * om;
                    acc[1] += (th - l) * (th - l) + th * th;
                    acc[2] += xn * xn;
                    acc[3] += res * res;
                    acc[4] += dp * dp;
                }
            }
        }
    }
    block_sum<GGL_NNORM>(acc, scratch);
    if (threadIdx.x == 0) {
        double* o = partials + ((size_t)k * gridDim.x + blockIdx.x) * GGL_NNORM;
#pragma unroll
        for (int v = 0; v < GGL_NNORM; ++v) o[v] = acc[v];
    }
}

void launch_ext_dual(hipStream_t st, double* X0, double* X1, const double* Omega, const double* OmegaPrev,
                     const double* Theta, const double* L, const double* Lam, const double* LamPrev, const int* pk,
                     int latent, double* partials, int K, int p, const int* skip)
{
    dim3 grid(ext_blocks(p), K), blk(XT);
    if (latent)
        hipLaunchKernelGGL(k_ext_dual<true>, grid, blk, 0, st, X0, X1, Omega, OmegaPrev, Theta, L, Lam, LamPrev, pk, partials,
                           p, skip);
    else
        hipLaunchKernelGGL(k_ext_dual<false>, grid, blk, 0, st, X0, X1, Omega, OmegaPrev, Theta, L, Lam, LamPrev, pk, partials,
                           p, skip);
}

// ---- pieces of the KKT residual (solver/ext_admm_solver.py:347-392): per-instance quantities -------------------------
// out = a A + b B + c C (B, C may be null)
__global__ __launch_bounds__(256) void k_lin3(double* __restrict__ out, double a, const double* __restrict__ A, double b,
                                              const double* __restrict__ B, double c, const double* __restrict__ C, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        double v = a * A[i];
        if (B) v += b * B[i];
        if (C) v += c * C[i];
        out[i] = v;
    }
}

void launch_lin3(hipStream_t st, double* out, double a, const double* A, double b, const double* B, double c,
                 const double* C, size_t n)
{
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_lin3, dim3(blocks), dim3(256), 0, st, out, a, A, b, B, c, C, n);
}

// partials[k][b] = sum over the chunk of ((A - B) + C)^2 restricted to the leading (p_k,p_k) block (B, C may be null)
__global__ __launch_bounds__(XT) void k_ext_sq(const double* __restrict__ A, const double* __restrict__ B,
                                               const double* __restrict__ C, const int* __restrict__ pk, int p,
                                               double* __restrict__ partials)
{
    __shared__ double scratch[XT / 64];
    const int k = blockIdx.y;
    const size_t pp = (size_t)p * p, base = (size_t)k * pp;
    const int pdim = pk[k];
    double acc[1] = {0.0};
    size_t i = (size_t)blockIdx.x * XCHUNK + threadIdx.x;
#pragma unroll
    for (int e = 0; e < XE; ++e, i += XT) {
        if (i < pp) {
            const int r = (int)(i / p), c = (int)(i - (size_t)r * p);
            if (r < pdim && c < pdim) {
                double d = A[base + i];
                if (B) d -= B[base + i];
                if (C) d += C[base + i];
                acc[0] += d * d;
            }
        }
    }
    block_sum<1>(acc, scratch);
    if (threadIdx.x == 0) partials[(size_t)k * gridDim.x + blockIdx.x] = acc[0];
}

void launch_ext_sq(hipStream_t st, const double* A, const double* B, const double* C, const int* pk, int K, int p,
                   double* partials)
{
    hipLaunchKernelGGL(k_ext_sq, dim3(ext_blocks(p), K), dim3(XT), 0, st, A, B, C, pk, p, partials);
}

// out[k] = prox_od_1norm(A[k], l1K[k]) (ggl_helper.py:16-27), per-instance threshold
__global__ __launch_bounds__(XT) void k_ext_prox_od(double* __restrict__ out, const double* __restrict__ A,
                                                    const double* __restrict__ l1K, int p)
{
    const int k = blockIdx.y;
    const size_t pp = (size_t)p * p, base = (size_t)k * pp;
    const double thr = l1K[k];
    size_t i = (size_t)blockIdx.x * XCHUNK + threadIdx.x;
#pragma unroll
    for (int e = 0; e < XE; ++e, i += XT) {
        if (i < pp) {
            const int r = (int)(i / p), c = (int)(i - (size_t)r * p);
            const double v = A[base + i];
            out[base + i] = (r == c) ? v : soft(v, thr);
        }
    }
}

void launch_ext_prox_od(hipStream_t st, double* out, const double* A, const double* l1K, int K, int p)
{
    hipLaunchKernelGGL(k_ext_prox_od, dim3(ext_blocks(p), K), dim3(XT), 0, st, out, A, l1K, p);
}

}  // namespace ggl
