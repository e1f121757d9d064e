// Eigendecomposition-free Omega-step for matrices that do not fit one CU's LDS.
//
// The reference computes Omega = Q diag(phip(d)) Q^T from eigh(W) (solver/admm_solver.py:180-187,
// solver/ggl_helper.py:272-303).  phip(d) = (d + sqrt(d^2 + 4 beta))/2 is a matrix function,
//     Omega = ( W + (W^2 + 4 beta I)^{1/2} ) / 2,
// and the square root of the SPD matrix A = W^2 + 4 beta I (spectrum in [4 beta, |W|^2 + 4 beta], so
// its condition number is known a priori) is obtained with the coupled Newton-Schulz iteration
//     M = Z Y,  T = (3I - a^2 M)/2,  Y <- a Y T,  Z <- a T Z,        Y0 = A/c, Z0 = I,  Y -> (A/c)^{1/2}
// with the optimal per-step scaling a = sqrt(3/(1+l+l^2)) of Chen & Chow on the tracked spectral
// interval [l,1].  Everything is a product of commuting symmetric matrices, i.e. pure FP64
// matrix-core work (gemm_sym.hip); 3n-2 products for n steps (n = 6..9 for the condition numbers ADMM
// produces), versus the tridiagonalisation-bound syevd.  The result agrees with the eigh route to
// ~1e-14 relative (tests/test_gpu_ops.py).
#include <cmath>
#include <vector>

#include "common.hpp"
#include "kernels.hpp"

namespace ggl {

// ---------------------------------------------------------------------------------------------
// W = ((Theta - L) - X) - beta_k S evaluated on the LOWER triangle and mirrored (numpy.linalg.eigh
// reads the lower triangle: admm_solver.py:181), one 32x32 tile pair per workgroup.
// ---------------------------------------------------------------------------------------------
static constexpr int FT = 32, FTY = 8, FQ = FT / FTY;

// Also emits what the spectral bound needs, so that W is not read a second time:
//   rowpart[k][c][i] = sum over the columns j of column tile c of |W_ij|
//   sqpart[k][block] = sum of W_ij^2 over the block's elements
template <bool HAS_L>
__global__ __launch_bounds__(256) void k_form_W_sym(double* __restrict__ W, const double* __restrict__ Theta,
                                                    const double* __restrict__ L, const double* __restrict__ X,
                                                    const double* __restrict__ S, const double* __restrict__ betaK,
                                                    double* __restrict__ rowpart, double* __restrict__ sqpart, int p)
{
    __shared__ double tile[FT][FT + 1];
    __shared__ double shsq[4];
    const int k = blockIdx.y;
    const int T = (p + FT - 1) / FT;
    int I = 0, b = blockIdx.x;
    while (b >= T - I) { b -= T - I; ++I; }
    const int J = I + b;
    const int I0 = I * FT, J0 = J * FT;
    const bool diag = (I == J);
    const int tx = threadIdx.x, ty = threadIdx.y;
    const double beta = betaK[k];
    const size_t base = (size_t)k * p * p;
    double sq = 0.0;
    double nat[FQ];
    // native element of the lower tile: (J0 + r, I0 + tx)
#pragma unroll
    for (int q = 0; q < FQ; ++q) {
        const int r = ty + FTY * q;
        const int gi = J0 + r, gj = I0 + tx;
        double w = 0.0;
        if (gi < p && gj < p && (!diag || r >= tx)) {
            const size_t o = base + (size_t)gi * p + gj;
            double t = Theta[o];
            if (HAS_L) t -= L[o];
            w = (t - X[o]) - beta * S[o];
            W[o] = w;
        }
        tile[r][tx] = w;
        nat[q] = fabs(w);
        sq += w * w;
    }
    __syncthreads();
    // mirrored element (I0 + r, J0 + tx) = tile[tx][r]
#pragma unroll
    for (int q = 0; q < FQ; ++q) {
        const int r = ty + FTY * q;
        const int gi = I0 + r, gj = J0 + tx;
        double m = 0.0;
        if (gi < p && gj < p && (!diag || tx > r)) {
            m = tile[tx][r];
            W[base + (size_t)gi * p + gj] = m;
            sq += m * m;
        }
        // row sums over the 32 lanes that share this row (a half wave: tx is the fast lane index)
        double rs_nat = nat[q], rs_mir = fabs(m);
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) {
            rs_nat += __shfl_xor(rs_nat, off, 64);
            rs_mir += __shfl_xor(rs_mir, off, 64);
        }
        if (tx == 0) {
            double* rp = rowpart + (size_t)k * T * p;
            if (diag) {
                if (I0 + r < p) rp[(size_t)I * p + I0 + r] = rs_nat + rs_mir;
            } else {
                if (J0 + r < p) rp[(size_t)I * p + J0 + r] = rs_nat;     // row J0+r, column tile I
                if (I0 + r < p) rp[(size_t)J * p + I0 + r] = rs_mir;     // row I0+r, column tile J
            }
        }
    }
    sq = wave_sum(sq);
    const int tid = ty * FT + tx;
    if ((tid & 63) == 0) shsq[tid >> 6] = sq;
    __syncthreads();
    if (tid == 0) sqpart[(size_t)k * gridDim.x + blockIdx.x] = (shsq[0] + shsq[1]) + (shsq[2] + shsq[3]);
}

// bounds[k] = { max_i sum_j |W_ij| , sum_ij W_ij^2 } from the partials above (fixed order)
__global__ __launch_bounds__(256) void k_bounds_final(const double* __restrict__ rowpart,
                                                      const double* __restrict__ sqpart, int T, int nblk, int p,
                                                      double* __restrict__ bounds)
{
    __shared__ double sh[256];
    const int k = blockIdx.x;
    const double* rp = rowpart + (size_t)k * T * p;
    double mx = 0.0;
    for (int i = threadIdx.x; i < p; i += 256) {
        double s = 0.0;
        for (int c = 0; c < T; ++c) s += rp[(size_t)c * p + i];
        mx = fmax(mx, s);
    }
    sh[threadIdx.x] = mx;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + off]);
        __syncthreads();
    }
    const double rowmax = sh[0];
    __syncthreads();
    double sq = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) sq += sqpart[(size_t)k * nblk + b];
    sh[threadIdx.x] = sq;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        bounds[2 * k + 0] = rowmax;
        bounds[2 * k + 1] = sh[0];
    }
}

int form_W_tiles(int p) { return (p + FT - 1) / FT; }

// rowpart: K * form_W_tiles(p) * p doubles; sqpart: K * T(T+1)/2 doubles; bounds: K * 2 doubles
void launch_form_W_sym(hipStream_t st, double* W, const double* Theta, const double* L, const double* X,
                       const double* S, const double* betaK, double* rowpart, double* sqpart, double* bounds, int K,
                       int p)
{
    const int T = (p + FT - 1) / FT;
    const int nblk = T * (T + 1) / 2;
    dim3 grid(nblk, K), blk(FT, FTY);
    if (L)
        hipLaunchKernelGGL(k_form_W_sym<true>, grid, blk, 0, st, W, Theta, L, X, S, betaK, rowpart, sqpart, p);
    else
        hipLaunchKernelGGL(k_form_W_sym<false>, grid, blk, 0, st, W, Theta, L, X, S, betaK, rowpart, sqpart, p);
    if (bounds) hipLaunchKernelGGL(k_bounds_final, dim3(K), dim3(256), 0, st, rowpart, sqpart, T, nblk, p, bounds);
}

// ---------------------------------------------------------------------------------------------
// host side: scaling schedule and the product sequence
// ---------------------------------------------------------------------------------------------
static std::vector<double> ns_schedule(double l, int max_steps = NS_MAX_STEPS)
{
    std::vector<double> al;
    for (int it = 0; it < max_steps; ++it) {
        const double a = (l < 0.99) ? std::sqrt(3.0 / (1.0 + l + l * l)) : 1.0;
        al.push_back(a);
        const double gl = 0.5 * a * l * (3.0 - a * a * l * l);
        const double g1 = 0.5 * a * (3.0 - a * a);
        l = std::fmin(gl, g1);
        if (1.0 - l < 4e-16) break;
    }
    return al;
}

// Start of the iteration from the unscaled A' = W^2 + 4 beta I and B' = A'^2 (both already formed as products):
//   Y1 = a0 Y0 T0 = (1.5 a0/c) A' - (0.5 a0^3/c^2) B',   Z1 = a0 T0 = 1.5 a0 I - (0.5 a0^3/c) A'   (Y0 = A'/c)
// so the spectral bound c may come from B' itself: lambda_max(A')^2 <= |A'^2|_inf, a fourth-root-of-W^4 bound
// that is 2-3x tighter than |W|_inf and saves one to two Newton-Schulz steps.  st[k] = {y1a, y1b, z1i, z1a, h}
// with h = 0.5 sqrt(c) when the start is already the end (one step: Omega = W/2 + h Y1).
__global__ __launch_bounds__(256) void k_ns_start(double* __restrict__ Y1, double* __restrict__ Z1,
                                                  const double* __restrict__ Ap, const double* __restrict__ Bp,
                                                  const double* __restrict__ W, const double* __restrict__ st, int p,
                                                  int final_step)
{
    const int k = blockIdx.y;
    const size_t pp = (size_t)p * p, base = (size_t)k * pp;
    const double y1a = st[k * 5 + 0], y1b = st[k * 5 + 1], z1i = st[k * 5 + 2], z1a = st[k * 5 + 3], h = st[k * 5 + 4];
    size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int e = 0; e < 4; ++e, i += 256) {
        if (i < pp) {
            const double a = Ap[base + i], b2 = Bp[base + i];
            const double y = y1a * a + y1b * b2;
            if (final_step) {
                Y1[base + i] = 0.5 * W[base + i] + h * y;          // Y1 is Omega here
            } else {
                const int r = (int)(i / p), c = (int)(i - (size_t)r * p);
                Y1[base + i] = y;
                Z1[base + i] = z1a * a + (r == c ? z1i : 0.0);
            }
        }
    }
}

// cbound_h[k] >= lambda_max(A'_k).  Returns 0, -1 (non-finite) or -2 (condition number above NS_KAPPA_LIMIT).
// Coefficient slots are numbered from the first launch AFTER the start kernel; start_h: [K][5].
int ns_plan(const double* cbound_h, const double* beta_h, int K, double* coef_h, double* start_h, NsPlan* plan,
            int force_mode)
{
    std::vector<std::vector<double>> al(K);
    std::vector<double> c(K);
    int n = 1;
    double kappa = 1.0;
    for (int k = 0; k < K; ++k) {
        c[k] = cbound_h[k] * (1.0 + 1e-10);
        if (!(c[k] > 0.0) || !std::isfinite(c[k]) || !(beta_h[k] > 0.0)) return -1;
        if (c[k] < 4.0 * beta_h[k]) c[k] = 4.0 * beta_h[k];          // lambda_min(A') = 4 beta is exact
        kappa = std::fmax(kappa, c[k] / (4.0 * beta_h[k]));
        al[k] = ns_schedule(std::sqrt(4.0 * beta_h[k] / c[k]));
        n = std::max(n, (int)al[k].size());
    }
    const bool stable = (force_mode == 2) || (force_mode == 0 && kappa > NS_SYM_KAPPA_MAX);
    // Beyond this the schedule would be cut off at NS_MAX_STEPS and fp64 (error ~ eps*sqrt(kappa)) could not
    // deliver the accuracy anyway: the caller takes the eigendecomposition route for this call.
    if (kappa > NS_KAPPA_LIMIT) return -2;
    plan->steps = n;
    plan->stable = stable;
    plan->kappa = kappa;
    plan->products = 2 + 2 * (n - 1);      // kernel launches of symmetric / right-multiply products (incl. A', B')
    for (int k = 0; k < K; ++k) {
        auto a_of = [&](int it) { return it < (int)al[k].size() ? al[k][it] : 1.0; };
        auto put = [&](int g, double cI, double cAcc, double cE, double dI, double dC) {
            double* o = coef_h + (size_t)g * NS_SLOT(K) + (size_t)k * 5;
            o[0] = cI; o[1] = cAcc; o[2] = cE; o[3] = dI; o[4] = dC;
        };
        const double sc = std::sqrt(c[k]);
        double a = a_of(0);
        double* s = start_h + (size_t)k * 5;
        s[0] = 1.5 * a / c[k];
        s[1] = -0.5 * a * a * a / (c[k] * c[k]);
        s[2] = 1.5 * a;
        s[3] = -0.5 * a * a * a / c[k];
        s[4] = 0.5 * sc;
        int g = 0;
        for (int it = 1; it < n; ++it) {
            a = a_of(it);
            put(g++, 1.5, -0.5 * a * a, 0.0, 0.0, 0.0);                      // T = 1.5 I - 0.5 a^2 (Z Y)
            if (it == n - 1) put(g++, 0.0, 0.5 * sc * a, 0.5, 0.0, 0.0);    // Omega = W/2 + sqrt(c) a (Y T)/2
            else if (stable) {
                double* o = coef_h + (size_t)g * NS_SLOT(K);                // [Y <- a Y T ; P <- a P T]
                o[k] = a;
                o[K + k] = a;
                ++g;
            } else {
                // one launch, 2K instances: [Y <- a Y T ; Z <- a T Z]; its slot holds 2K coefficient rows
                double* o = coef_h + (size_t)g * NS_SLOT(K);
                double* y = o + (size_t)k * 5;
                double* z = o + ((size_t)K + k) * 5;
                y[0] = 0.0; y[1] = a; y[2] = 0.0; y[3] = 0.0; y[4] = 0.0;
                z[0] = 0.0; z[1] = a; z[2] = 0.0; z[3] = 0.0; z[4] = 0.0;
                ++g;
            }
        }
    }
    return 0;
}

// Phase A (before the bound is known): A' = W^2 + 4 beta I -> AB.Y, B' = A'^2 -> AB.Z.
// pre_d: two coefficient slots {4 beta, 1, 0, 0, 0} and {0, 1, 0, 0, 0} per instance.
void ns_prepare(hipStream_t st, const double* pre_d, const double* W, double* AB, int K, int p, int variant)
{
    const size_t cs = NS_SLOT(K), n1 = (size_t)K * p * p;
    launch_symm(st, W, W, AB, nullptr, nullptr, pre_d, K, p, variant);
    launch_symm(st, AB, AB, AB + n1, nullptr, nullptr, pre_d + cs, K, p, variant);
}

// Phase B.  AB = [A' | B'] from ns_prepare (free afterwards), YP = the other [Y | Z] scratch pair, Tb one stack,
// W preserved, out = Omega.
//
// fast schedule (plan.stable == false, small condition numbers): every product is a product of commuting
// symmetric matrices, computed as upper triangle + mirror.
// stable schedule: Z is replaced by P = Z^T, M = P^T Y is still a congruence (exactly symmetric), but
// Y <- a Y T and P <- a P T are full, unsymmetrised products in one 2K-batch launch: this keeps Y = Y0 P
// exactly, which is what makes the coupled iteration insensitive to rounding (the symmetrised form
// amplifies commutator errors by ~sqrt(kappa)/4 per step).
void ns_run(hipStream_t st, const NsPlan& plan, const double* coef_d, const double* start_d, const double* W,
            double* AB, double* YP, double* Tb, double* out, int K, int p, int variant, size_t pstride)
{
    // pstride: distance (doubles) between the two stacks of a [Y|Z] pair; 0 = contiguous (K*p*p).  A sub-batch
    // of a larger ctx (two-stream execution) passes the full-stack stride.
    const size_t cs = NS_SLOT(K), n1 = pstride ? pstride : (size_t)K * p * p;
    const int n = plan.steps;
    dim3 grid((unsigned)(((size_t)p * p + 1023) / 1024), K);
    if (n == 1) {
        hipLaunchKernelGGL(k_ns_start, grid, dim3(256), 0, st, out, nullptr, AB, AB + n1, W, start_d, p, 1);
        return;
    }
    hipLaunchKernelGGL(k_ns_start, grid, dim3(256), 0, st, YP, YP + n1, AB, AB + n1, W, start_d, p, 0);
    double *cur = YP, *nxt = AB;      // cur = [Y | Z]
    int g = 0;
    for (int it = 1; it < n; ++it) {
        // T = 1.5 I - 0.5 a^2 (Z Y)   [fast: Z Y = Z^T Y, Z symmetric; stable: P^T Y]
        launch_symm(st, cur + n1, cur, Tb, nullptr, nullptr, coef_d + cs * g++, K, p, variant);
        if (it == n - 1) {
            launch_symm(st, cur, Tb, out, nullptr, W, coef_d + cs * g++, K, p, variant);
        } else if (plan.stable) {
            launch_gemm_right(st, cur, Tb, nxt, coef_d + cs * g++, 2 * K, K, p, 0);   // needs a contiguous pair
            std::swap(cur, nxt);
        } else {
            launch_symm_pair(st, cur, Tb, nxt, Tb, cur + n1, nxt + n1, coef_d + cs * g++, K, p, variant);
            std::swap(cur, nxt);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// L-step without an eigendecomposition:  L = (C - mu I)_+ = (C - mu I)(I + sign(C - mu I))/2
// (prox_rank_norm, solver/ggl_helper.py:29-36 with D,Q from eigh, admm_solver.py:197-205).
// sign(B) by the scaled Newton-Schulz iteration X <- a X (3I - a^2 X^2)/2, X0 = B/|B|: a single sequence of
// polynomials in B, so every product is a product of commuting symmetric matrices and the symmetrised
// form is stable (unlike the coupled square-root iteration).  Eigenvalues of B closer to zero than
// l0*|B| are not resolved by a schedule built for l0; that is DETECTED from max|T_last - I| (the last
// step's residual) and the caller then retries with a smaller l0 or falls back to rocSOLVER, so the
// result is never silently inexact.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_norm_bounds(const double* __restrict__ W, int p, double* __restrict__ part)
{
    // 16 rows per workgroup, 4 per wave, all four streamed together (independent loads in flight)
    __shared__ double sh_abs[4], sh_sq[4];
    const int k = blockIdx.y;
    const double* w = W + (size_t)k * p * p;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * 16 + wave * 4;
    double a[4] = {0.0, 0.0, 0.0, 0.0}, sq = 0.0;
    size_t ro[4];
    bool ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        ok[q] = (r0 + q) < p;
        ro[q] = (size_t)min(r0 + q, p - 1) * p;
    }
    for (int j = lane; j < p; j += 64) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double v = ok[q] ? w[ro[q] + j] : 0.0;
            a[q] += fabs(v);
            sq += v * v;
        }
    }
    double mx = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) mx = fmax(mx, wave_sum(a[q]));
    sq = wave_sum(sq);
    if (lane == 0) { sh_abs[wave] = mx; sh_sq[wave] = sq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* o = part + 2 * ((size_t)k * gridDim.x + blockIdx.x);
        o[0] = fmax(fmax(sh_abs[0], sh_abs[1]), fmax(sh_abs[2], sh_abs[3]));
        o[1] = (sh_sq[0] + sh_sq[1]) + (sh_sq[2] + sh_sq[3]);
    }
}

int norm_bounds_blocks(int p) { return (p + 15) / 16; }

// part[k][blk] = {max row abs-sum, sum of squares} over row block blk (host finishes the reduction)
void launch_norm_bounds(hipStream_t st, const double* W, int K, int p, double* part)
{
    hipLaunchKernelGGL(k_norm_bounds, dim3(norm_bounds_blocks(p), K), dim3(256), 0, st, W, p, part);
}

// cnorm_h[k] >= |C_k|_2, mu_h[k] = mu1_k / rho.  Fills the coefficient table; returns steps.
int rank_ns_plan(const double* cnorm_h, const double* mu_h, int K, double l0, double* coef_h, NsPlan* plan)
{
    std::vector<double> al = ns_schedule(l0 < 0.5 ? l0 : 0.5, NS_RANK_MAX_STEPS);
    const int n = (int)al.size();
    if (n < 2) return -1;
    plan->steps = n;
    plan->products = 2 * n + 1;
    plan->stable = false;
    for (int k = 0; k < K; ++k) {
        const double mu = mu_h[k];
        const double nb = (cnorm_h[k] + mu) * (1.0 + 1e-10);      // |C - mu I| <= |C| + mu
        if (!(nb > 0.0) || !std::isfinite(nb)) return -1;
        auto put = [&](int g, double cI, double cAcc, double cE, double dI, double dC) {
            double* o = coef_h + (size_t)g * NS_SLOT(K) + (size_t)k * 5;
            o[0] = cI; o[1] = cAcc; o[2] = cE; o[3] = dI; o[4] = dC;
        };
        int g = 0;
        const double a0 = al[0], s2 = a0 * a0 / (nb * nb);
        // T0 = 1.5 I - 0.5 a0^2 X0^2, X0 = (C - mu I)/nb   [product C*C, E = C]
        put(g++, 1.5 - 0.5 * s2 * mu * mu, -0.5 * s2, s2 * mu, 0.0, 0.0);
        // X1 = a0 X0 T0 = (a0/nb) C T0 - (a0 mu/nb) T0      [product C*T0, E = T0]
        put(g++, 0.0, a0 / nb, -a0 * mu / nb, 0.0, 0.0);
        for (int it = 1; it < n; ++it) {
            const double a = al[it];
            put(g++, 1.5, -0.5 * a * a, 0.0, 0.0, 0.0);          // T = 1.5 I - 0.5 a^2 X^2
            put(g++, 0.0, a, 0.0, 1.0, 1.0);                     // X <- a X T ; second output P2 = I + X
        }
        // L = (C - mu I) P2 / 2 = 0.5 C P2 - 0.5 mu P2        [product C*P2, E = P2]
        put(g++, 0.0, 0.5, -0.5 * mu, 0.0, 0.0);
    }
    return 0;
}

// C preserved.  Xa, Xb, Tb, P2: scratch stacks.  maxdev: device [K], receives max|T_last - I|.
void rank_ns_run(hipStream_t st, const NsPlan& plan, const double* coef_d, const double* C, double* Xa, double* Xb,
                 double* Tb, double* P2, double* out, double* maxdev, int K, int p, int variant)
{
    const size_t cs = NS_SLOT(K);
    int g = 0;
    const int n = plan.steps;
    double *X = Xa, *Xn = Xb;
    (void)hipMemsetAsync(maxdev, 0, K * sizeof(double), st);
    launch_symm(st, C, C, Tb, nullptr, C, coef_d + cs * g++, K, p, variant);
    launch_symm(st, C, Tb, X, nullptr, Tb, coef_d + cs * g++, K, p, variant);
    for (int it = 1; it < n; ++it) {
        const bool last = (it == n - 1);
        launch_symm(st, X, X, Tb, nullptr, nullptr, coef_d + cs * g++, K, p, variant, last ? maxdev : nullptr);
        launch_symm(st, X, Tb, Xn, last ? P2 : nullptr, nullptr, coef_d + cs * g++, K, p, variant);
        std::swap(X, Xn);
    }
    launch_symm(st, C, P2, out, nullptr, P2, coef_d + cs * g++, K, p, variant);
}

}  // namespace ggl
