// L-step: the eigenvalues next to the threshold are DEFLATED instead of iterated down (CPU prototype
// tools/proto_deflate.py).  Reference: prox_rank_norm, solver/ggl_helper.py:29-36 -- L = Q diag(max(d - mu, 0)) Q^T.
//
// L = (C - mu I)_+ = B (I + sign(B)) / 2, B = C - mu I.  The sign iteration X <- X t(X^2) (newton_schulz.hip) resolves the
// eigenvalues of B down to l0 |B|, and its length is set by the few that sit next to zero: 22 products at l0 = 1e-3, 31 at
// 1e-5, 36 at 1e-6.  After a COARSE pass every eigenvalue farther than l0 |B| from zero is at +-1 to rounding and
// R = I - X^2 has numerical rank r = the one or two eigenvalues inside (measured on the C4 iterates: r <= 2, the rest of R's
// spectrum at 5e-15: profiles/r4_lstep_deflation_prototype.txt).  Per instance:
//   range finder   Y = R G = G - X (X G),  G p x 8 fixed Gaussian           (two tall-skinny passes over X)
//   basis          Q1 = orth(first 6 columns of Y), columns whose remainder is below tau1 dropped (R's resolved directions
//                  sit at ~1e-13 here).  A direction with a SMALL residual eigenvalue r_i comes out of this contaminated by
//                  the noise floor at the relative level 1e-13 / r_i -- and an angle error delta costs the correction
//                  delta |B| |D| in L and 2 delta^2 in the trace -- so the basis is PURIFIED by one more application:
//                  V = orth(R Q1), accepted above tau2 = 1e-10 (relative rounding 1e-16 / r_i; what is dropped is resolved
//                  to a sign error below 5e-11 on an eigenvalue within l0 |B| of zero).  The last 2 columns are PROBES:
//                  what is left of R g after projecting V out must be noise, or rank(R) > 6 and the instance goes the
//                  iteration's way
//   small problem  H = V^T B V (r x r) eigendecomposed exactly;  D = sign(H) - V^T X V
//   correction     sign(B) = X + V D V^T   =>   L += (B V D V^T + V D V^T B) / 4,   trace(sign B) = trace X + trace D
// p^2 * 8 work per pass, no product.
#include "common.hpp"
#include "kernels.hpp"

namespace ggl {

// ---------------------------------------------------------------------------------------------
// tall-skinny products with a SYMMETRIC (K,p,p) stack: Vout[k][q][i] = sum_j A[k][i][j] Vin[k][q][j]
//   MODE 0: A Vin;   MODE 1: G - A Vin (G [q][p] shared);   MODE 2: A Vin - mu_k Vin
// vectors are stored [K][DEFL_Q][p] (every column contiguous), the DEFL_Q columns of Vin in LDS.
// A is symmetric, so the sum runs down COLUMN i: a lane owns one output index, a workgroup 64 of them, its four waves every
// fourth row j (each row segment a full 512-byte wave row, eight rows of loads in flight), Vin[q][j] is an LDS broadcast and
// no value crosses lanes until the four waves' partial sums meet in LDS, in a fixed order.  (The first version walked the
// ROWS, one wave per four rows, and reduced 32 sums per step across the wave with shuffles: 50 us per pass at K = 50,
// p = 500, 2 TB/s -- the shuffles, not the memory.)
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void k_defl_tall(const double* __restrict__ A, const double* __restrict__ Vin, size_t vin_stride,
                                                   const double* __restrict__ G, size_t g_stride, const double* __restrict__ muK,
                                                   double* __restrict__ Vout, int p, const double* __restrict__ meta, int need)
{
    // need: 0 always; 1 only instances with accepted columns (meta[k][0] > 0)
    extern __shared__ __attribute__((aligned(16))) double vs[];          // [DEFL_Q][p], then [4][DEFL_Q][64] partial sums
    const int k = blockIdx.y;
    if (need == 1 && meta[k * 4 + 0] == 0.0) return;
    const double* vin = Vin + (size_t)k * vin_stride;
    for (int e = threadIdx.x; e < DEFL_Q * p; e += 256) vs[e] = vin[e];
    __syncthreads();
    double* red = vs + (size_t)DEFL_Q * p;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 64 + lane;
    const double* col = A + (size_t)k * p * p + min(i, p - 1);
    double acc[DEFL_Q];
#pragma unroll
    for (int q = 0; q < DEFL_Q; ++q) acc[q] = 0.0;
    constexpr int U = 8;
    for (int j = wave; j < p; j += 4 * U) {
        double a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = (j + 4 * u < p) ? col[(size_t)(j + 4 * u) * p] : 0.0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int jj = min(j + 4 * u, p - 1);
#pragma unroll
            for (int q = 0; q < DEFL_Q; ++q) acc[q] = fma(a[u], vs[q * p + jj], acc[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < DEFL_Q; ++q) red[(wave * DEFL_Q + q) * 64 + lane] = acc[q];
    __syncthreads();
    if (i < p) {
#pragma unroll
        for (int h = 0; h < DEFL_Q / 4; ++h) {
            const int q = wave + 4 * h;
            double v = (red[(0 * DEFL_Q + q) * 64 + lane] + red[(1 * DEFL_Q + q) * 64 + lane]) +
                       (red[(2 * DEFL_Q + q) * 64 + lane] + red[(3 * DEFL_Q + q) * 64 + lane]);
            if (MODE == 1) v = G[(size_t)k * g_stride + (size_t)q * p + i] - v;
            if (MODE == 2) v -= muK[k] * vs[q * p + i];
            Vout[(size_t)k * DEFL_Q * p + (size_t)q * p + i] = v;
        }
    }
}

template <int MODE>
static void launch_tall(hipStream_t st, const double* A, const double* Vin, size_t vin_stride, const double* G, size_t g_stride,
                        const double* muK, double* Vout, int K, int p, const double* meta = nullptr, int need = 0)
{
    static_assert(DEFL_Q % 4 == 0, "the columns are dealt over the four waves on the way out");
    const size_t lds = ((size_t)DEFL_Q * p + 4 * DEFL_Q * 64) * sizeof(double);
    // per launch, not once per process: the attribute belongs to the CURRENT device's copy of the kernel
    (void)hipFuncSetAttribute((const void*)k_defl_tall<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    hipLaunchKernelGGL(k_defl_tall<MODE>, dim3((p + 63) / 64, K), dim3(256), lds, st, A, Vin, vin_stride, G, g_stride, muK, Vout,
                       p, meta, need);
}

// block-wide sums of NV values per thread (256 threads); results in every thread.  scratch: 4 * NV doubles.
template <int NV>
__device__ __forceinline__ void block_sums256(double (&v)[NV], double* scratch)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) scratch[wave * NV + i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = (scratch[i] + scratch[NV + i]) + (scratch[2 * NV + i] + scratch[3 * NV + i]);
}

// ---------------------------------------------------------------------------------------------
// basis: Y[k] (DEFL_Q columns, in place) -> orthonormal V (first r <= DEFL_Q0 columns, the others zeroed), probe columns
// projected.  meta[k] = { r, largest norm of a projected probe column, -, - }
// Classical Gram-Schmidt, twice (every column against ALL accepted ones in one fused reduction, then again); a column is
// accepted while its remainder is above `tau` (absolute: X is scaled to |X|_2 <= 1, so R's entries are on the scale of 1).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_defl_basis(double* __restrict__ Y, double* __restrict__ meta, int p, double tau,
                                                    const double* __restrict__ Gprobe)
{
    // Gprobe != null (first pass): the probe columns are REPLACED by G's on the way out, so that the second pass's
    // application of R produces fresh probes.  Second pass: an instance whose first pass found nothing (r = 0: its probes
    // were pure noise already) is final and skipped.
    extern __shared__ __attribute__((aligned(16))) double ys[];          // [DEFL_Q][p] + 4 * DEFL_Q
    double* scratch = ys + (size_t)DEFL_Q * p;
    const int k = blockIdx.x, tid = threadIdx.x;
    double* y = Y + (size_t)k * DEFL_Q * p;
    if (!Gprobe && meta[k * 4 + 0] == 0.0) return;
    for (int e = tid; e < DEFL_Q * p; e += 256) ys[e] = y[e];
    __syncthreads();
    int r = 0;
    for (int a = 0; a < DEFL_Q0; ++a) {
        for (int pass = 0; pass < 2 && r > 0; ++pass) {
            double d[DEFL_Q0];
#pragma unroll
            for (int b = 0; b < DEFL_Q0; ++b) d[b] = 0.0;
            for (int i = tid; i < p; i += 256) {
                const double ya = ys[a * p + i];
#pragma unroll
                for (int b = 0; b < DEFL_Q0; ++b) if (b < r) d[b] = fma(ys[b * p + i], ya, d[b]);
            }
            block_sums256<DEFL_Q0>(d, scratch);
            for (int i = tid; i < p; i += 256) {
                double ya = ys[a * p + i];
#pragma unroll
                for (int b = 0; b < DEFL_Q0; ++b) if (b < r) ya = fma(-d[b], ys[b * p + i], ya);
                ys[a * p + i] = ya;
            }
            __syncthreads();
        }
        double n2[1] = {0.0};
        for (int i = tid; i < p; i += 256) n2[0] = fma(ys[a * p + i], ys[a * p + i], n2[0]);
        block_sums256<1>(n2, scratch);
        const double nrm = sqrt(n2[0]);
        if (!(nrm > tau)) continue;                   // (uniform: every thread holds the same nrm) resolved / empty column
        const double inv = 1.0 / nrm;
        for (int i = tid; i < p; i += 256) ys[r * p + i] = ys[a * p + i] * inv;      // r <= a: never overwrites a later column
        __syncthreads();
        r += 1;
    }
    double leak = 0.0;
    {
        for (int b = DEFL_Q0; b < DEFL_Q; ++b) {
            for (int pass = 0; pass < 2 && r > 0; ++pass) {
                double d[DEFL_Q0];
#pragma unroll
                for (int a = 0; a < DEFL_Q0; ++a) d[a] = 0.0;
                for (int i = tid; i < p; i += 256) {
                    const double yb = ys[b * p + i];
#pragma unroll
                    for (int a = 0; a < DEFL_Q0; ++a) if (a < r) d[a] = fma(ys[a * p + i], yb, d[a]);
                }
                block_sums256<DEFL_Q0>(d, scratch);
                for (int i = tid; i < p; i += 256) {
                    double yb = ys[b * p + i];
#pragma unroll
                    for (int a = 0; a < DEFL_Q0; ++a) if (a < r) yb = fma(-d[a], ys[a * p + i], yb);
                    ys[b * p + i] = yb;
                }
                __syncthreads();
            }
            double n2[1] = {0.0};
            for (int i = tid; i < p; i += 256) n2[0] = fma(ys[b * p + i], ys[b * p + i], n2[0]);
            block_sums256<1>(n2, scratch);
            leak = fmax(leak, sqrt(n2[0]));
        }
    }
    for (int e = tid; e < DEFL_Q * p; e += 256) {
        double v = (e < r * p) ? ys[e] : 0.0;
        if (Gprobe && e >= DEFL_Q0 * p) v = Gprobe[e];
        y[e] = v;
    }
    if (tid == 0) {
        meta[k * 4 + 0] = r;
        meta[k * 4 + 1] = leak;
    }
}

// ---------------------------------------------------------------------------------------------
// small problem: H = V^T BV, M = V^T XV (r x r), sign(H) by a cyclic Jacobi eigendecomposition, D = sign(H) - M;
// Wm = BV D ([DEFL_Q][p], columns >= r zero); meta[k][2] = trace D
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_defl_small(const double* __restrict__ V, const double* __restrict__ BV,
                                                    const double* __restrict__ XV, double* __restrict__ Wm,
                                                    double* __restrict__ meta, int p)
{
    __shared__ double Hs[DEFL_Q0][DEFL_Q0], Ms[DEFL_Q0][DEFL_Q0], Ds[DEFL_Q0][DEFL_Q0];
    __shared__ double H[DEFL_Q0][DEFL_Q0], U[DEFL_Q0][DEFL_Q0];        // (thread 0's work arrays: dynamically indexed)
    const int k = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = (int)meta[k * 4 + 0];
    const size_t off = (size_t)k * DEFL_Q * p;
    const double *v = V + off, *bv = BV + off, *xv = XV + off;
    double* wm = Wm + off;
    if (r == 0) {
        for (int e = tid; e < DEFL_Q * p; e += 256) wm[e] = 0.0;
        if (tid == 0) meta[k * 4 + 2] = 0.0;
        return;
    }
    for (int pr = wave; pr < r * r; pr += 4) {
        const int a = pr / r, b = pr % r;
        double h = 0.0, m = 0.0;
        for (int i = lane; i < p; i += 64) {
            const double va = v[a * p + i];
            h = fma(va, bv[b * p + i], h);
            m = fma(va, xv[b * p + i], m);
        }
        h = wave_sum(h);
        m = wave_sum(m);
        if (lane == 0) { Hs[a][b] = h; Ms[a][b] = m; }
    }
    __syncthreads();
    if (tid == 0) {
        for (int a = 0; a < r; ++a)
            for (int b = 0; b < r; ++b) {
                H[a][b] = 0.5 * (Hs[a][b] + Hs[b][a]);
                U[a][b] = (a == b) ? 1.0 : 0.0;
            }
        // cyclic two-sided Jacobi on the r x r block (r <= 6)
        for (int sweep = 0; sweep < 30; ++sweep) {
            double offd = 0.0, dg = 0.0;
            for (int a = 0; a < r; ++a)
                for (int b = 0; b < r; ++b) (a == b ? dg : offd) += H[a][b] * H[a][b];
            if (!(offd > 1e-30 * dg)) break;
            for (int a = 0; a < r - 1; ++a)
                for (int b = a + 1; b < r; ++b) {
                    if (H[a][b] == 0.0) continue;
                    const double th = (H[b][b] - H[a][a]) / (2.0 * H[a][b]);
                    const double t = copysign(1.0, th) / (fabs(th) + sqrt(1.0 + th * th));
                    const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                    for (int m = 0; m < r; ++m) {
                        const double x = H[m][a], y = H[m][b];
                        H[m][a] = c * x - s * y;
                        H[m][b] = s * x + c * y;
                    }
                    for (int m = 0; m < r; ++m) {
                        const double x = H[a][m], y = H[b][m];
                        H[a][m] = c * x - s * y;
                        H[b][m] = s * x + c * y;
                    }
                    for (int m = 0; m < r; ++m) {
                        const double x = U[m][a], y = U[m][b];
                        U[m][a] = c * x - s * y;
                        U[m][b] = s * x + c * y;
                    }
                }
        }
        double tr = 0.0;
        for (int a = 0; a < r; ++a)
            for (int b = a; b < r; ++b) {
                double sh = 0.0;
                for (int m = 0; m < r; ++m) sh += U[a][m] * (H[m][m] > 0.0 ? 1.0 : (H[m][m] < 0.0 ? -1.0 : 0.0)) * U[b][m];
                const double d = sh - 0.5 * (Ms[a][b] + Ms[b][a]);
                Ds[a][b] = Ds[b][a] = d;
                if (a == b) tr += d;
            }
        meta[k * 4 + 2] = tr;
    }
    __syncthreads();
    for (int e = tid; e < DEFL_Q * p; e += 256) {
        const int a = e / p, i = e - a * p;
        double s = 0.0;
        if (a < r)
            for (int b = 0; b < r; ++b) s = fma(bv[b * p + i], Ds[b][a], s);
        wm[e] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// correction: L[k][i][j] += sum_a (Wm[a][i] V[a][j] + V[a][i] Wm[a][j]) / 4 for the instances with r > 0.
// The two products are rounded separately and added commutatively, so the update is bitwise symmetric in (i, j).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_defl_update(double* __restrict__ L, const double* __restrict__ V,
                                                     const double* __restrict__ Wm, const double* __restrict__ meta, int p)
{
#pragma clang fp contract(off)      // (hipcc contracts by default, also through __dmul_rn / __dadd_rn: an fma here would round
                                    // w_i v_j + v_i w_j differently from w_j v_i + v_j w_i -- one ulp of asymmetry in L)
    const int k = blockIdx.y;
    const int r = (int)meta[k * 4 + 0];
    if (r == 0) return;
    const size_t off = (size_t)k * DEFL_Q * p;
    const double *v = V + off, *w = Wm + off;
    double* Lk = L + (size_t)k * p * p;
    const int i = blockIdx.x;                       // one row per workgroup (16 rows per workgroup measured slower: 85 vs 64 us)
    double wi[DEFL_Q0], vi[DEFL_Q0];
#pragma unroll
    for (int a = 0; a < DEFL_Q0; ++a) { wi[a] = (a < r) ? w[a * p + i] : 0.0; vi[a] = (a < r) ? v[a * p + i] : 0.0; }
    for (int j = threadIdx.x; j < p; j += 256) {
        double s = 0.0;
#pragma unroll
        for (int a = 0; a < DEFL_Q0; ++a) {
            if (a >= r) break;
            const double t = wi[a] * v[a * p + j], u = vi[a] * w[a * p + j];
            s = s + (t + u);
        }
        Lk[(size_t)i * p + j] = Lk[(size_t)i * p + j] + 0.25 * s;
    }
}

// The whole deflation of a batch: X = the coarse pass's last iterate, C and mu_k/rho the L-step's input, L = B (I + X) / 2 as the
// closing product left it.  work: 4 * K * DEFL_Q * p doubles; G: [DEFL_Q][p]; meta: [K][4] (device) -> {r, probe leak, trace D, -}.
void launch_deflate(hipStream_t st, const double* X, const double* C, const double* muK, double* L, const double* G, double* work,
                    double* meta, int K, int p, double tau1, double tau2)
{
    const size_t vs = (size_t)K * DEFL_Q * p;
    double *Y = work, *BV = work + vs, *XV = work + 2 * vs, *Wm = work + 3 * vs;
    const size_t vst = (size_t)DEFL_Q * p;
    const size_t lds = ((size_t)DEFL_Q * p + 4 * DEFL_Q) * sizeof(double);
    (void)hipFuncSetAttribute((const void*)k_defl_basis, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    launch_tall<0>(st, X, G, 0, nullptr, 0, nullptr, XV, K, p);                    // X G      (scratch: XV)
    launch_tall<1>(st, X, XV, vst, G, 0, nullptr, Y, K, p);                        // Y = R G = G - X (X G)
    hipLaunchKernelGGL(k_defl_basis, dim3(K), dim3(256), lds, st, Y, meta, p, tau1, G);        // Q1, r1 (+ G's probe columns)
    // from here on only the instances that have something to deflate (r1 > 0)
    launch_tall<0>(st, X, Y, vst, nullptr, 0, nullptr, XV, K, p, meta, 1);         // X Q1
    launch_tall<1>(st, X, XV, vst, Y, vst, nullptr, BV, K, p, meta, 1);            // R Q1 = Q1 - X (X Q1)   (scratch: BV)
    hipLaunchKernelGGL(k_defl_basis, dim3(K), dim3(256), lds, st, BV, meta, p, tau2, (const double*)nullptr);   // V, r, leak
    launch_tall<2>(st, C, BV, vst, nullptr, 0, muK, Y, K, p, meta, 1);             // B V = C V - mu V     (V lives in BV's slot)
    launch_tall<0>(st, X, BV, vst, nullptr, 0, nullptr, XV, K, p, meta, 1);        // X V
    hipLaunchKernelGGL(k_defl_small, dim3(K), dim3(256), 0, st, BV /*V*/, Y /*BV*/, XV, Wm, meta, p);
    hipLaunchKernelGGL(k_defl_update, dim3(p, K), dim3(256), 0, st, L, BV /*V*/, Wm, meta, p);
}

int deflate_max_p() { return (int)((160 * 1024 - 256 - 64 - 4 * DEFL_Q * 64 * sizeof(double)) / (DEFL_Q * sizeof(double))); }

}  // namespace ggl
