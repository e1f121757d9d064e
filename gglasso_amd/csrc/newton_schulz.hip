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

template <bool HAS_L>
__global__ __launch_bounds__(256) void k_form_W_sym(double* __restrict__ W, const double* __restrict__ Theta,
                                                    const double* __restrict__ L, const double* __restrict__ X,
                                                    const double* __restrict__ S, const double* __restrict__ betaK, int p)
{
    __shared__ double tile[FT][FT + 1];
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
    }
    __syncthreads();
    // mirrored element (I0 + r, J0 + tx) = tile[tx][r]
#pragma unroll
    for (int q = 0; q < FQ; ++q) {
        const int r = ty + FTY * q;
        const int gi = I0 + r, gj = J0 + tx;
        if (gi < p && gj < p && (!diag || tx > r)) W[base + (size_t)gi * p + gj] = tile[tx][r];
    }
}

void launch_form_W_sym(hipStream_t st, double* W, const double* Theta, const double* L, const double* X,
                       const double* S, const double* betaK, int K, int p)
{
    const int T = (p + FT - 1) / FT;
    dim3 grid(T * (T + 1) / 2, K), blk(FT, FTY);
    if (L)
        hipLaunchKernelGGL(k_form_W_sym<true>, grid, blk, 0, st, W, Theta, L, X, S, betaK, p);
    else
        hipLaunchKernelGGL(k_form_W_sym<false>, grid, blk, 0, st, W, Theta, L, X, S, betaK, p);
}

// Norm bounds of W: part[k][blk] = { max_i sum_j |W_ij| , sum_ij W_ij^2 } over the 64 rows of row-block
// blk (one wave per 16 rows).  The host finishes the reduction (max / sum over the blocks, fixed order).
static constexpr int NB_ROWS = 64;

int norm_bounds_blocks(int p) { return (p + NB_ROWS - 1) / NB_ROWS; }

__global__ __launch_bounds__(256) void k_norm_bounds(const double* __restrict__ W, int p, double* __restrict__ part)
{
    __shared__ double sh_abs[4], sh_sq[4];
    const int k = blockIdx.y;
    const double* w = W + (size_t)k * p * p;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * NB_ROWS + wave * 16;
    double mx = 0.0, sq = 0.0;
    for (int i = r0; i < min(r0 + 16, p); ++i) {
        double a = 0.0;
        for (int j = lane; j < p; j += 64) {
            const double v = w[(size_t)i * p + j];
            a += fabs(v);
            sq += v * v;
        }
        mx = fmax(mx, wave_sum(a));
    }
    sq = wave_sum(sq);
    if (lane == 0) { sh_abs[wave] = mx; sh_sq[wave] = sq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* o = part + 2 * ((size_t)k * gridDim.x + blockIdx.x);
        o[0] = fmax(fmax(sh_abs[0], sh_abs[1]), fmax(sh_abs[2], sh_abs[3]));
        o[1] = (sh_sq[0] + sh_sq[1]) + (sh_sq[2] + sh_sq[3]);
    }
}

void launch_norm_bounds(hipStream_t st, const double* W, int K, int p, double* part)
{
    hipLaunchKernelGGL(k_norm_bounds, dim3(norm_bounds_blocks(p), K), dim3(256), 0, st, W, p, part);
}

// ---------------------------------------------------------------------------------------------
// host side: scaling schedule and the product sequence
// ---------------------------------------------------------------------------------------------
static std::vector<double> ns_schedule(double l)
{
    std::vector<double> al;
    for (int it = 0; it < NS_MAX_STEPS; ++it) {
        const double a = (l < 0.99) ? std::sqrt(3.0 / (1.0 + l + l * l)) : 1.0;
        al.push_back(a);
        const double gl = 0.5 * a * l * (3.0 - a * a * l * l);
        const double g1 = 0.5 * a * (3.0 - a * a);
        l = std::fmin(gl, g1);
        if (1.0 - l < 4e-16) break;
    }
    return al;
}

int ns_plan(const double* bounds_h, const double* beta_h, int K, double* coef_h, NsPlan* plan, int force_mode)
{
    std::vector<std::vector<double>> al(K);
    std::vector<double> c(K);
    int n = 1;
    double kappa = 1.0;
    for (int k = 0; k < K; ++k) {
        const double w2 = std::fmin(bounds_h[2 * k], std::sqrt(bounds_h[2 * k + 1])) * (1.0 + 1e-10);
        if (!(w2 >= 0.0) || !std::isfinite(w2) || !(beta_h[k] > 0.0)) return -1;
        c[k] = w2 * w2 + 4.0 * beta_h[k];
        kappa = std::fmax(kappa, c[k] / (4.0 * beta_h[k]));
        al[k] = ns_schedule(std::sqrt(4.0 * beta_h[k] / c[k]));
        n = std::max(n, (int)al[k].size());
    }
    const bool stable = (force_mode == 2) || (force_mode == 0 && kappa > NS_SYM_KAPPA_MAX);
    plan->steps = n;
    plan->stable = stable;
    plan->kappa = kappa;
    plan->products = (n == 1) ? 2 : (stable ? 2 + 2 * (n - 2) + 2 : 3 * n - 2);   // kernel launches
    // launch g of the sequence (see ns_run) reads coef_h[(g*K + k)*5 ..]; a right-multiply launch reads
    // its 2K scalars from the start of its slot instead.
    for (int k = 0; k < K; ++k) {
        auto a_of = [&](int it) { return it < (int)al[k].size() ? al[k][it] : 1.0; };
        auto put = [&](int g, double cI, double cAcc, double cE, double dI, double dC) {
            double* o = coef_h + ((size_t)g * K + k) * 5;
            o[0] = cI; o[1] = cAcc; o[2] = cE; o[3] = dI; o[4] = dC;
        };
        const double sc = std::sqrt(c[k]);
        int g = 0;
        double a = a_of(0);
        // g0: Y0 = W^2/c + (4 beta/c) I ; second output T0 = 1.5 I - 0.5 a0^2 Y0
        put(g++, 4.0 * beta_h[k] / c[k], 1.0 / c[k], 0.0, 1.5, -0.5 * a * a);
        // step 0: Y1 = a0 Y0 T0 (final if n == 1); Z1 = P1 = zs * T0 with zs = a0 kept as a pending scalar
        if (n == 1) { put(g++, 0.0, 0.5 * sc * a, 0.5, 0.0, 0.0); continue; }
        put(g++, 0.0, a, 0.0, 0.0, 0.0);
        double zs = a;
        for (int it = 1; it < n; ++it) {
            a = a_of(it);
            put(g++, 1.5, -0.5 * a * a * zs, 0.0, 0.0, 0.0);                 // T = 1.5 I - 0.5 a^2 (Z Y)
            if (it == n - 1) put(g++, 0.0, 0.5 * sc * a, 0.5, 0.0, 0.0);    // Omega = W/2 + sqrt(c) a (Y T)/2
            else if (stable) {
                double* o = coef_h + (size_t)g * K * 5;                     // [Y <- a Y T ; P <- a zs P T]
                o[k] = a;
                o[K + k] = a * zs;
                ++g;
                zs = 1.0;
            } else {
                put(g++, 0.0, a, 0.0, 0.0, 0.0);                            // Y <- a Y T
                put(g++, 0.0, a * zs, 0.0, 0.0, 0.0);                       // Z <- a T Z
                zs = 1.0;
            }
        }
    }
    return 0;
}

// YP0 / YP1: [Y stack | Z stack] scratch pairs (2*K*p*p doubles each); Tb: one stack; W preserved; out: Omega.
//
// fast path (plan.stable == false, small condition numbers): every product is a product of
// commuting symmetric matrices, computed as upper triangle + mirror (3 per step).
// stable path: Z is replaced by P = Z^T, M = P^T Y is still a congruence (exactly symmetric), but
// Y <- a Y T and P <- a P T are full, unsymmetrised products in one 2K-batch launch: this keeps
// Y = Y0 P exactly, which is what makes the coupled iteration insensitive to rounding (the
// symmetrised form amplifies commutator errors by ~sqrt(kappa)/4 per step).
void ns_run(hipStream_t st, const NsPlan& plan, const double* coef_d, const double* W, double* YP0, double* YP1,
            double* Tb, double* out, int K, int p, int variant)
{
    const size_t cs = (size_t)K * 5, n1 = (size_t)K * p * p;
    int g = 0;
    const int n = plan.steps;
    double *cur = YP0, *nxt = YP1;      // cur = [Y | Z]
    // g0: Y0 -> cur.Y, T0 -> nxt.Z  (Z1 = P1 = a0 T0: the scalar is folded into later coefficients)
    launch_symm(st, W, W, cur, nxt + n1, nullptr, coef_d + cs * g++, K, p, variant);
    if (n == 1) {
        launch_symm(st, cur, nxt + n1, out, nullptr, W, coef_d + cs * g++, K, p, variant);
        return;
    }
    // step 0: Y1 = a0 Y0 T0 -> nxt.Y, right next to T0, so that nxt = [Y1 | Z1]
    launch_symm(st, cur, nxt + n1, nxt, nullptr, nullptr, coef_d + cs * g++, K, p, variant);
    std::swap(cur, nxt);
    for (int it = 1; it < n; ++it) {
        // T = 1.5 I - 0.5 a^2 zs (Z Y)   [fast: Z Y = Z^T Y, Z symmetric; stable: P^T Y]
        launch_symm(st, cur + n1, cur, Tb, nullptr, nullptr, coef_d + cs * g++, K, p, variant);
        if (it == n - 1) {
            launch_symm(st, cur, Tb, out, nullptr, W, coef_d + cs * g++, K, p, variant);
        } else if (plan.stable) {
            launch_gemm_right(st, cur, Tb, nxt, coef_d + cs * g++, 2 * K, K, p, 0);
            std::swap(cur, nxt);
        } else {
            launch_symm(st, cur, Tb, nxt, nullptr, nullptr, coef_d + cs * g++, K, p, variant);
            launch_symm(st, Tb, cur + n1, nxt + n1, nullptr, nullptr, coef_d + cs * g++, K, p, variant);
            std::swap(cur, nxt);
        }
    }
}

}  // namespace ggl
