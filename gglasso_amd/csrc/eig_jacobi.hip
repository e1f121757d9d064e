// One-workgroup-per-matrix symmetric eigensolver for matrices that fit one CU's LDS (p <= 128).
// Replaces numpy.linalg.eigh at solver/admm_solver.py:181,199 and single_admm_solver.py:164,174
// for small p, with the eigenvalue map and the reconstruction Q f(D) Q^T (phiplus,
// ggl_helper.py:280-303 / prox_rank_norm, ggl_helper.py:29-36) optionally fused behind it, so a
// whole Omega-step is one launch that reads W once and writes Omega once.
//
// Method: one-sided (Hestenes) Jacobi on the ROWS of G = A + sigma*I, sigma = 2*|A|_inf, so that G
// is positive definite and its singular vectors are the eigenvectors of A.  A round-robin
// tournament gives p/2 disjoint row pairs per step; each pair is owned by a group of 16 or 32
// lanes that keeps both rows in registers, reduces the three inner products with wavefront
// shuffles, and rotates.  One barrier per step; G never leaves LDS.  Converged when a full sweep
// applies no rotation (|g_a.g_b| <= tol |g_a||g_b|).  Eigenvalue_i = |g_i| - sigma, eigenvector_i
// = g_i/|g_i|.
#include "common.hpp"
#include "kernels.hpp"

namespace ggl {

static constexpr int JT = 1024;        // threads per workgroup
static constexpr int JMAXP = 128;
static constexpr int JMAXC = 8;        // row elements per lane (JMAXP / 16)
static constexpr int JMAXSWEEP = 48;

bool jacobi_fits(int p) { return p >= 1 && p <= JMAXP; }

__device__ __forceinline__ double eig_map(int map, double d, double beta)
{
    if (map == MAP_PHIPLUS) return 0.5 * (sqrt(d * d + 4.0 * beta) + d);
    if (map == MAP_RANK) return fmax(d - beta, 0.0);
    return d;
}

template <int LG>   // lanes per row pair: 16 or 32
__global__ __launch_bounds__(JT) void k_jacobi(const double* __restrict__ A, double* __restrict__ D,
                                               double* __restrict__ R, double* __restrict__ out, int map,
                                               const double* __restrict__ betaK, int* __restrict__ info, int p,
                                               int ld)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* G = lds;                          // [p][ld]
    double* dv = lds + (size_t)p * ld;        // [p] eigenvalues
    double* fv = dv + p;                      // [p] mapped eigenvalues / row sums
    int* flags = (int*)(fv + p);              // [JMAXSWEEP]
    const int tid = threadIdx.x;
    const int k = blockIdx.x;
    const double* a = A + (size_t)k * p * p;

    // symmetric load from the LOWER triangle (numpy.linalg.eigh default UPLO='L')
    for (int idx = tid; idx < p * p; idx += JT) {
        const int i = idx / p, j = idx - i * p;
        G[i * ld + j] = (j <= i) ? a[idx] : a[(size_t)j * p + i];
    }
    if (tid < JMAXSWEEP) flags[tid] = 0;
    __syncthreads();
    if (tid < p) {
        double s = 0.0;
        for (int j = 0; j < p; ++j) s += fabs(G[tid * ld + j]);
        fv[tid] = s;
    }
    __syncthreads();
    double sigma = 0.0;
    for (int i = 0; i < p; ++i) sigma = fmax(sigma, fv[i]);   // same value in every thread
    sigma = (sigma > 0.0) ? 2.0 * sigma : 1.0;
    __syncthreads();
    if (tid < p) G[tid * ld + tid] += sigma;
    __syncthreads();

    const int n = p + (p & 1);
    const int npairs = n >> 1;
    const int g = tid / LG, li = tid % LG;
    const double tol = sqrt((double)p) * 2.220446049250313e-16;
    int sweeps = -1;

    for (int sweep = 0; sweep < JMAXSWEEP; ++sweep) {
        for (int s = 0; s < n - 1; ++s) {
            if (g < npairs) {
                int ra, rb;
                if (g == 0) { ra = n - 1; rb = s; }
                else { ra = (s + g) % (n - 1); rb = (s - g + (n - 1)) % (n - 1); }
                if (ra < p && rb < p) {     // uniform over the LG lanes of the group
                    double va[JMAXC], vb[JMAXC];
                    double al = 0.0, be = 0.0, ga = 0.0;
#pragma unroll
                    for (int u = 0; u < JMAXC; ++u) {
                        const int c = li + u * LG;
                        if (c < p) {
                            va[u] = G[ra * ld + c];
                            vb[u] = G[rb * ld + c];
                        } else {
                            va[u] = 0.0;
                            vb[u] = 0.0;
                        }
                        al += va[u] * va[u];
                        be += vb[u] * vb[u];
                        ga += va[u] * vb[u];
                    }
#pragma unroll
                    for (int off = LG / 2; off > 0; off >>= 1) {
                        al += __shfl_xor(al, off, 64);
                        be += __shfl_xor(be, off, 64);
                        ga += __shfl_xor(ga, off, 64);
                    }
                    if (fabs(ga) > tol * sqrt(al * be)) {
                        const double zeta = (be - al) / (2.0 * ga);
                        const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                        const double c = 1.0 / sqrt(1.0 + t * t);
                        const double sn = c * t;
#pragma unroll
                        for (int u = 0; u < JMAXC; ++u) {
                            const int col = li + u * LG;
                            if (col < p) {
                                G[ra * ld + col] = c * va[u] - sn * vb[u];
                                G[rb * ld + col] = sn * va[u] + c * vb[u];
                            }
                        }
                        if (li == 0) flags[sweep] = 1;
                    }
                }
            }
            __syncthreads();
        }
        if (flags[sweep] == 0) { sweeps = sweep + 1; break; }
    }

    // eigenvalues and normalised rows
    const int ngroups = JT / LG;
    for (int r = g; r < p; r += ngroups) {
        double ss = 0.0;
        for (int c = li; c < p; c += LG) { const double v = G[r * ld + c]; ss += v * v; }
#pragma unroll
        for (int off = LG / 2; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
        const double nu = sqrt(ss);
        const double inv = 1.0 / nu;
        for (int c = li; c < p; c += LG) G[r * ld + c] *= inv;
        if (li == 0) {
            const double d = nu - sigma;
            dv[r] = d;
            fv[r] = eig_map(map, d, betaK ? betaK[k] : 0.0);
        }
    }
    __syncthreads();
    if (tid == 0 && info) info[k] = sweeps;
    if (D) for (int i = tid; i < p; i += JT) D[(size_t)k * p + i] = dv[i];
    if (R) {
        double* r = R + (size_t)k * p * p;
        for (int idx = tid; idx < p * p; idx += JT) {
            const int i = idx / p, j = idx - i * p;
            r[idx] = G[i * ld + j];
        }
    }
    if (out) {
        // out[i][j] = sum_m f_m G[m][i] G[m][j], 2x2 outputs per thread; t = gi*gj first so that the
        // result is bitwise symmetric.
        double* o = out + (size_t)k * p * p;
        const int nb = (p + 1) >> 1;
        for (int blk = tid; blk < nb * nb; blk += JT) {
            const int i0 = (blk / nb) * 2, j0 = (blk % nb) * 2;
            const int i1 = min(i0 + 1, p - 1), j1 = min(j0 + 1, p - 1);
            double c00 = 0.0, c01 = 0.0, c10 = 0.0, c11 = 0.0;
            for (int m = 0; m < p; ++m) {
                const double f = fv[m];
                const double a0 = G[m * ld + i0], a1 = G[m * ld + i1];
                const double b0 = G[m * ld + j0], b1 = G[m * ld + j1];
                c00 = fma(f, a0 * b0, c00);
                c01 = fma(f, a0 * b1, c01);
                c10 = fma(f, a1 * b0, c10);
                c11 = fma(f, a1 * b1, c11);
            }
            o[(size_t)i0 * p + j0] = c00;
            if (j0 + 1 < p) o[(size_t)i0 * p + j0 + 1] = c01;
            if (i0 + 1 < p) {
                o[(size_t)(i0 + 1) * p + j0] = c10;
                if (j0 + 1 < p) o[(size_t)(i0 + 1) * p + j0 + 1] = c11;
            }
        }
    }
}

hipError_t launch_jacobi(hipStream_t st, const double* A, double* D, double* R, double* out, int map,
                         const double* betaK, int* info, int K, int p)
{
    if (!jacobi_fits(p)) return hipErrorInvalidValue;
    const int ld = p | 1;   // odd row stride: rows start on different banks
    const size_t lds = ((size_t)p * ld + 2 * (size_t)p) * sizeof(double) + JMAXSWEEP * sizeof(int);
    const int npairs = (p + 1) / 2;
    hipError_t e;
    if (npairs <= 32) {
        e = hipFuncSetAttribute((const void*)k_jacobi<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_jacobi<32>, dim3(K), dim3(JT), lds, st, A, D, R, out, map, betaK, info, p, ld);
    } else {
        e = hipFuncSetAttribute((const void*)k_jacobi<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_jacobi<16>, dim3(K), dim3(JT), lds, st, A, D, R, out, map, betaK, info, p, ld);
    }
    return hipGetLastError();
}

}  // namespace ggl
