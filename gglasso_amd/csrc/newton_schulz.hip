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
#include <unordered_map>

#include <functional>
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
                                                    const double* __restrict__ S, const double* __restrict__ betaK,
                                                    int p)
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

int form_W_tiles(int p) { return (p + FT - 1) / FT; }

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

// ---- mixed-degree schedule (fast, all-symmetric mode) --------------------------------------------------
// One step maps x = sqrt(eig(Z Y)) in [l,1] by x -> x t(x^2), t of degree 1 (cubic step, Chen-Chow scaling),
// 2 (quintic) or 4 (degree 9): the minimax polynomial for the constant 1 on [l,1] (cf. the "Polar Express"
// construction, Amsel et al. 2025).  Higher degrees cost more symmetric products per step but contract the
// interval much faster; which mix is cheapest depends on l.  Cost in symmetric products of the stack, on top of
// A' and B':   cubic first 0 / middle 3 / last 2,  quintic 1 / 4 / 3,  degree nine 2 / 5 / 4
// (t of degree 4 takes two products: Q = M^2 + a M, t(M) = t4 Q (Q + d I) + e M + f I).
struct NsStep { double t[5]; double lnew; };

static NsStep ns_cubic_step(double l)
{
    const double a = (l < 0.99) ? std::sqrt(3.0 / (1.0 + l + l * l)) : 1.0;
    const double gl = 0.5 * a * l * (3.0 - a * a * l * l);
    const double g1 = 0.5 * a * (3.0 - a * a);
    return {{1.5 * a, -0.5 * a * a * a, 0.0, 0.0, 0.0}, std::fmin(gl, g1)};
}

// Narrow intervals (s = 1 - x^2 in [0, smax], smax <= 0.05), where the minimax deviation drops below what an
// exchange algorithm can resolve in fp64: t(s) = the Taylor polynomial of (1-s)^(-1/2) of degree m with its
// leading term s^m replaced by the degree-(m-1) Chebyshev economisation on [0, smax].  Deviation
// <= a_m smax^m / 2^(2m-1) + sum_{j>m} a_j smax^j, a_j = binom(2j,j)/4^j: within a few per cent of the minimax.
static NsStep ns_econ_step(double l, int m)
{
    static const double cheb3[4] = {-1.0, 18.0, -48.0, 32.0};                       // T3*(u) = T3(2u-1)
    static const double cheb5[6] = {-1.0, 50.0, -400.0, 1120.0, -1280.0, 512.0};    // T5*(u)
    const double* ch = (m == 3) ? cheb3 : cheb5;
    const double smax = 1.0 - l * l;
    double a[48];
    a[0] = 1.0;
    for (int j = 1; j < 48; ++j) a[j] = a[j - 1] * (2.0 * j - 1.0) / (2.0 * j);
    double cs[5] = {0, 0, 0, 0, 0};                       // t in powers of s
    double smj = 1.0;                                     // smax^(m-i)
    for (int i = m - 1; i >= 0; --i) {
        smj *= smax;
        cs[i] = a[i] - a[m] * smj * ch[i] / ch[m];
    }
    double E = a[m] * std::pow(smax, m) / ch[m], sp = std::pow(smax, m + 1);
    for (int j = m + 1; j < 48; ++j, sp *= smax) E += a[j] * sp;
    E *= 1.0 + 1e-6;
    NsStep r = {{0, 0, 0, 0, 0}, 0.0};
    static const double binom[5][5] = {{1, 0, 0, 0, 0}, {1, 1, 0, 0, 0}, {1, 2, 1, 0, 0}, {1, 3, 3, 1, 0}, {1, 4, 6, 4, 1}};
    for (int j = 0; j < m; ++j)                           // s^j = (1 - x^2)^j
        for (int i = 0; i <= j; ++i) r.t[i] += cs[j] * binom[j][i] * ((i & 1) ? -1.0 : 1.0);
    const double sc = 1.0 / (1.0 + E);
    for (int i = 0; i < m; ++i) r.t[i] *= sc;
    r.lnew = (1.0 - E) * sc;
    return r;
}

// minimax p(x) = a x + b x^3 + c x^5 ~ 1 on [l,1] by Remez exchange: the reference is {l, q1, q2, 1} and the
// interior extrema q1,q2 are the roots of p' = a + 3b x^2 + 5c x^4, a quadratic in x^2.
static NsStep ns_quintic_step(double l)
{
    const double e = 1.0 - l;
    if (1.0 - l * l <= 0.05) return ns_econ_step(l, 3);
    double q1 = l + e / 3.0, q2 = l + 2.0 * e / 3.0, co[4] = {0, 0, 0, 0};
    for (int iter = 0; iter < 60; ++iter) {
        const double pts[4] = {l, q1, q2, 1.0};
        double A[4][5];
        for (int r = 0; r < 4; ++r) {
            const double x = pts[r], x2 = x * x;
            A[r][0] = x; A[r][1] = x * x2; A[r][2] = x * x2 * x2; A[r][3] = (r & 1) ? -1.0 : 1.0; A[r][4] = 1.0;
        }
        for (int i = 0; i < 4; ++i) {                     // Gauss-Jordan with partial pivoting
            int pv = i;
            for (int r = i + 1; r < 4; ++r) if (std::fabs(A[r][i]) > std::fabs(A[pv][i])) pv = r;
            for (int cc = 0; cc < 5; ++cc) std::swap(A[i][cc], A[pv][cc]);
            for (int r = 0; r < 4; ++r) {
                if (r == i) continue;
                const double f = A[r][i] / A[i][i];
                for (int cc = i; cc < 5; ++cc) A[r][cc] -= f * A[i][cc];
            }
        }
        for (int i = 0; i < 4; ++i) co[i] = A[i][4] / A[i][i];     // a, b, c, E  (p(l) = 1 - E, p(q1) = 1 + E, ...)
        const double disc = 9.0 * co[1] * co[1] - 20.0 * co[0] * co[2];
        if (!(disc > 0.0) || !(co[2] > 0.0)) break;
        const double sq = std::sqrt(disc);
        const double r1 = (-3.0 * co[1] - sq) / (10.0 * co[2]), r2 = (-3.0 * co[1] + sq) / (10.0 * co[2]);
        if (!(r1 > 0.0) || !(r2 > r1)) break;
        const double n1 = std::sqrt(r1), n2 = std::sqrt(r2);
        const bool done = std::fabs(n1 - q1) + std::fabs(n2 - q2) < 1e-14;
        q1 = n1; q2 = n2;
        if (done) break;
    }
    // the exchange converges from below: the true deviation is max |p - 1| over the final extrema
    auto pv = [&](double x) { const double x2 = x * x; return x * (co[0] + x2 * (co[1] + x2 * co[2])); };
    double E = std::fmax(std::fmax(std::fabs(pv(l) - 1.0), std::fabs(pv(q1) - 1.0)),
                         std::fmax(std::fabs(pv(q2) - 1.0), std::fabs(pv(1.0) - 1.0)));
    E *= 1.0 + 1e-9;
    const double s = 1.0 / (1.0 + E);                     // rescale: the image is [ (1-E)/(1+E), 1 ]
    return {{co[0] * s, co[1] * s, co[2] * s, 0.0, 0.0}, (1.0 - E) * s};
}

// degree 9: Remez exchange on a Chebyshev-spaced grid of [l,1], in the basis x ((1-x^2)/smax)^j (well conditioned
// whatever l); the grid maximum is inflated by 1e-3 for what lies between grid points (the error curve has six
// alternations; between neighbours of a 384-point grid it cannot rise by more than ~2e-4 of its amplitude).
static NsStep ns_nonic_step(double l)
{
    constexpr int m = 5, G = 384;
    const double smax = 1.0 - l * l;
    if (smax <= 0.05) return ns_econ_step(l, m);
    static thread_local double gx[G + 1], gb[G + 1][m], err[G + 1];
    for (int i = 0; i <= G; ++i) {
        const double x = l + (1.0 - l) * 0.5 * (1.0 - std::cos(M_PI * i / G));
        gx[i] = x;
        const double u = (1.0 - x * x) / smax;
        double b = x;
        for (int j = 0; j < m; ++j, b *= u) gb[i][j] = b;
    }
    int ref[m + 1];
    for (int k = 0; k <= m; ++k) ref[k] = (int)std::lround((double)G * k / m);
    double co[m + 1] = {};
    for (int iter = 0; iter < 30; ++iter) {
        double A[m + 1][m + 2];
        for (int r = 0; r <= m; ++r) {
            for (int j = 0; j < m; ++j) A[r][j] = gb[ref[r]][j];
            A[r][m] = (r & 1) ? -1.0 : 1.0;               // p(ref_0) = 1 - E, p(ref_1) = 1 + E, ...
            A[r][m + 1] = 1.0;
        }
        for (int i = 0; i <= m; ++i) {
            int pv = i;
            for (int r = i + 1; r <= m; ++r) if (std::fabs(A[r][i]) > std::fabs(A[pv][i])) pv = r;
            for (int cc = 0; cc <= m + 1; ++cc) std::swap(A[i][cc], A[pv][cc]);
            for (int r = 0; r <= m; ++r) {
                if (r == i) continue;
                const double f = A[r][i] / A[i][i];
                for (int cc = i; cc <= m + 1; ++cc) A[r][cc] -= f * A[i][cc];
            }
        }
        for (int i = 0; i <= m; ++i) co[i] = A[i][m + 1] / A[i][i];
        for (int i = 0; i <= G; ++i) {
            double v = 0.0;
            for (int j = 0; j < m; ++j) v += co[j] * gb[i][j];
            err[i] = v - 1.0;
        }
        // local extrema of the error (end points included), reduced to an alternating set of m+1
        int alt[G + 2], na = 0;
        for (int i = 0; i <= G; ++i) {
            const bool ext = (i == 0 || i == G) || ((err[i] - err[i - 1]) * (err[i + 1] - err[i]) <= 0.0);
            if (!ext) continue;
            if (na > 0 && (err[i] > 0.0) == (err[alt[na - 1]] > 0.0)) {
                if (std::fabs(err[i]) > std::fabs(err[alt[na - 1]])) alt[na - 1] = i;
            } else {
                alt[na++] = i;
            }
        }
        int lo = 0;
        while (na - lo > m + 1) {
            if (std::fabs(err[alt[lo]]) < std::fabs(err[alt[na - 1]])) ++lo; else --na;
        }
        if (na - lo < m + 1) break;
        bool same = true;
        for (int k = 0; k <= m; ++k) { same = same && (ref[k] == alt[lo + k]); ref[k] = alt[lo + k]; }
        if (same) break;
    }
    NsStep r = {{0, 0, 0, 0, 0}, 0.0};
    static const double binom[5][5] = {{1, 0, 0, 0, 0}, {1, 1, 0, 0, 0}, {1, 2, 1, 0, 0}, {1, 3, 3, 1, 0}, {1, 4, 6, 4, 1}};
    double sj = 1.0;
    for (int j = 0; j < m; ++j, sj *= smax)
        for (int i = 0; i <= j; ++i) r.t[i] += co[j] / sj * binom[j][i] * ((i & 1) ? -1.0 : 1.0);
    double E = 0.0;
    for (int i = 0; i <= G; ++i) {
        const double x2 = gx[i] * gx[i];
        const double pv = gx[i] * (r.t[0] + x2 * (r.t[1] + x2 * (r.t[2] + x2 * (r.t[3] + x2 * r.t[4]))));
        E = std::fmax(E, std::fabs(pv - 1.0));
    }
    E *= 1.0 + 1e-3;
    const double sc = 1.0 / (1.0 + E);
    for (int i = 0; i < m; ++i) r.t[i] *= sc;
    r.lnew = (1.0 - E) * sc;
    return r;
}

// kind 0: coupled square-root iteration (Omega-step); kind 1: sign iteration X <- X t(X^2) (L-step), where every
// step costs X^2, [t], X t: 2 / 3 / 4 products
static int ns_step_cost(int kind, int d, bool first, bool last)
{
    const int base = (d == 3) ? 0 : (d == 5 ? 1 : 2);     // products inside t(M) beyond M itself
    if (kind == 1) return base + 2;
    if (first) return base;                               // Z1 = T1 from A', B'; Y1 (or Omega) one product unless cubic
    return base + (last ? 2 : 3);                         // M, [t(M)], Y T, [T Z]
}

struct NsSeq { int n = 0; int cost = 1 << 30; unsigned char deg[NS_RANK_MAX_STEPS]; NsStep st[NS_RANK_MAX_STEPS]; };

// The schedule is a deterministic function of a QUANTISED interval: l is rounded down on a geometric grid
// (ratio 1.01) below 0.5 and 1-l is rounded up on a geometric grid (ratio 1.02) above, at the start and after
// every step.  Rounding only widens the interval, so the polynomials stay valid; it makes every step and the
// cost-to-go a function of an integer key, which is what the dynamic programme below memoises.
struct NsKey { int key; double l; };
static NsKey ns_quantise(double l)
{
    if (l < 0.5) {
        const double lg = std::log(1.01);
        const int idx = (int)std::floor(std::log(l) / lg);
        return {idx, std::exp(idx * lg)};
    }
    const double lg = std::log(1.02), e = std::fmax(1.0 - l, 1e-300);
    const int j = (int)std::ceil(std::log(e) / lg);
    const double eq = std::exp(j * lg);
    if (eq > 0.5) return {-70, std::exp(-70 * std::log(1.01))};        // just below 0.5 on the lower grid
    return {1000000 + j, 1.0 - eq};
}

struct NsPlanner {
    int degrees;
    int kind;
    struct Node { int cost; int d; };
    std::unordered_map<long long, NsStep> steps;          // (key, degree) -> step
    std::unordered_map<int, Node> togo;                   // key -> cheapest completion as a non-first step
    std::unordered_map<int, NsSeq> plans;                 // key -> whole schedule from a first step
    const NsStep& step(const NsKey& q, int d)
    {
        const long long id = (long long)q.key * 16 + d;
        auto it = steps.find(id);
        if (it != steps.end()) return it->second;
        const NsStep s = (d == 3) ? ns_cubic_step(q.l) : (d == 5 ? ns_quintic_step(q.l) : ns_nonic_step(q.l));
        return steps.emplace(id, s).first->second;
    }
    // a schedule ends when the spectrum is inside [1 - tol, 1]: tol = NS_TOL_EXACT drives it to what fp64 can represent;
    // a larger tol (GGL_OPT_NS_TOL) is the relative spectral accuracy of the matrix function the caller asks for
    double tol = NS_TOL_EXACT;
    bool converged(const NsStep& s) const { return 1.0 - s.lnew < tol; }
    void set_tol(double t)
    {
        if (t == tol) return;
        togo.clear();                 // the steps themselves do not depend on the tolerance, the plans do
        plans.clear();
        tol = t;
    }
    Node best_from(const NsKey& q, int depth)
    {
        // the cost-to-go of a key must not depend on how deep the query was that first reached it (it is memoised):
        // no depth cut here -- every step strictly narrows the interval, so the recursion is finite (< 40 levels
        // down to l = 1e-13); schedules that are too long are refused where they are used
        auto it = togo.find(q.key);
        if (it != togo.end()) return it->second;
        Node b = {1 << 29, 3};
        {
            for (int d = 3; d <= degrees; d += (d == 5 ? 4 : 2)) {
                const NsStep& s = step(q, d);
                int cost;
                if (converged(s)) cost = ns_step_cost(kind, d, false, true);
                else {
                    const NsKey nq = ns_quantise(s.lnew);
                    if (nq.l <= q.l) continue;            // no progress (cannot happen for l in (0,1))
                    cost = ns_step_cost(kind, d, false, false) + best_from(nq, depth + 1).cost;
                }
                if (cost < b.cost) b = {cost, d};
            }
        }
        togo.emplace(q.key, b);
        return b;
    }
    const NsSeq& plan(double l)
    {
        const NsKey q0 = ns_quantise(l);
        auto it = plans.find(q0.key);
        if (it != plans.end()) return it->second;
        if (plans.size() > 4096) { plans.clear(); }
        NsSeq best;
        int bd = 3, bc = 1 << 30;
        for (int d = 3; d <= degrees; d += (d == 5 ? 4 : 2)) {
            const NsStep& s = step(q0, d);
            int cost = ns_step_cost(kind, d, true, converged(s));
            if (!converged(s)) cost += best_from(ns_quantise(s.lnew), 1).cost;
            if (cost < bc) { bc = cost; bd = d; }
        }
        best.cost = bc;
        NsKey q = q0;
        int d = bd;
        bool done = false;
        for (int i = 0; i < NS_RANK_MAX_STEPS; ++i) {
            const NsStep& s = step(q, d);
            best.deg[i] = (unsigned char)d;
            best.st[i] = s;
            best.n = i + 1;
            if (converged(s)) { done = true; break; }
            q = ns_quantise(s.lnew);
            d = best_from(q, i + 1).d;
        }
        if (!done) best.cost = 1 << 30;            // longer than any table: refused by the callers
        return plans.emplace(q0.key, best).first->second;
    }
};

static const NsSeq& ns_mixed_schedule(double l, int degrees, double tol)
{
    static thread_local NsPlanner planners[3] = {{3, 0}, {5, 0}, {9, 0}};
    NsPlanner& pl = planners[degrees >= 9 ? 2 : (degrees >= 5 ? 1 : 0)];
    pl.set_tol(std::fmin(std::fmax(tol, NS_TOL_EXACT), 1e-6));
    return pl.plan(l);
}

static const NsSeq& ns_sign_schedule(double l, int degrees, double tol = NS_TOL_EXACT)
{
    static thread_local NsPlanner planners[3] = {{3, 1}, {5, 1}, {9, 1}};
    NsPlanner& pl = planners[degrees >= 9 ? 2 : (degrees >= 5 ? 1 : 0)];
    pl.set_tol(std::fmin(std::fmax(tol, NS_TOL_EXACT), 1e-6));
    return pl.plan(l);
}

int ns_schedule_query(double l, int degrees, int max_steps, int* deg, double* coef, int* units, double tol)
{
    if (!(l > 0.0) || !(l <= 1.0)) return -1;
    const NsSeq& sq = (degrees >= 100) ? ns_sign_schedule(l, degrees - 100, tol) : ns_mixed_schedule(l, degrees, tol);
    if (sq.n < 1 || sq.n > max_steps || sq.cost >= (1 << 29)) return -1;
    for (int i = 0; i < sq.n; ++i) {
        deg[i] = sq.deg[i];
        for (int j = 0; j < 5; ++j) coef[6 * i + j] = sq.st[i].t[j];
        coef[6 * i + 5] = sq.st[i].lnew;
    }
    *units = (degrees >= 100) ? sq.cost : 2 + sq.cost;
    return sq.n;
}

// Products (in units of one symmetric product of the instance, A' and B' included) of the all-symmetric schedule for a
// spectrum in [l, 1]; -1 where that schedule does not apply (condition number above NS_SYM_KAPPA_MAX, no schedule).
int ns_units_query(double l, int degrees, double tol)
{
    if (!(l > 0.0) || !(l <= 1.0) || 1.0 / (l * l) > NS_SYM_KAPPA_MAX) return -1;
    const NsSeq& sq = ns_mixed_schedule(l, degrees, tol);
    if (sq.n < 1 || sq.n > NS_MAX_STEPS || sq.cost >= (1 << 29)) return -1;
    return 2 + sq.cost;
}

// Contiguous groups of a batch whose instances differ in conditioning.  One launch sequence runs ONE schedule -- the degree
// sequence is common to a launch -- so ns_plan builds it for the worst instance, and a grid of independent problems
// (helper/model_selection.py:619-633 solves every point with its own eigh) pays the worst point's product count for all of
// them: 10-11 products per Omega-step for the 20-point p = 1000 lambda1 grid where its well-conditioned half needs 7-8.  The
// instances of such a grid are ordered by lambda1, i.e. by conditioning, so CONTIGUOUS groups with their own schedules --
// the concurrent parts the engine already has -- recover most of it.  units[k]: ns_units_query of instance k.  A partition
// into g <= max_groups runs of lengths len_i costs  sum_i U_i (F + len_i I),  U_i the largest unit count in run i,
// I = seconds per instance and product, F = fixed seconds per launch (a deterministic model, never a clock: the decision
// shows in the last digits of the iterates); it is taken when it beats the single schedule by at least 6 %.
// Returns the number of groups (1: leave the batch whole) and their lengths.
int ns_group_partition(const int* units, int K, int p, int max_groups, int* len_out)
{
    len_out[0] = K;
    if (K < 2 || max_groups < 2) return 1;
    int umin = units[0], umax = units[0];
    for (int k = 1; k < K; ++k) { umin = std::min(umin, units[k]); umax = std::max(umax, units[k]); }
    if (umin <= 0 || umin == umax) return 1;
    const double I = 2.5e-14 * (double)p * p * p, F = 6.5e-6;     // 64x64 direct-to-LDS kernel: 25 us per instance-product at p = 1000
    // the dynamic programme is O(groups K^2) host work per step: only where the step is long enough not to notice
    if (3.0 * K * K * 4e-9 > 0.02 * umax * (F + K * I)) return 1;
    const int G = std::min(max_groups, 3);
    const double INF = 1e300;
    std::vector<double> best((size_t)(G + 1) * (K + 1), INF);
    std::vector<int> from((size_t)(G + 1) * (K + 1), -1);
    best[0] = 0.0;
    for (int g = 1; g <= G; ++g)
        for (int j = g; j <= K; ++j) {
            int u = 0;
            for (int i = j - 1; i >= g - 1; --i) {            // run [i, j)
                u = std::max(u, units[i]);
                const double prev = best[(size_t)(g - 1) * (K + 1) + i];
                if (prev >= INF) continue;
                const double cst = prev + u * (F + (j - i) * I);
                if (cst < best[(size_t)g * (K + 1) + j]) { best[(size_t)g * (K + 1) + j] = cst; from[(size_t)g * (K + 1) + j] = i; }
            }
        }
    const double whole = best[(size_t)1 * (K + 1) + K];
    int gbest = 1;
    for (int g = 2; g <= G; ++g)
        if (best[(size_t)g * (K + 1) + K] < best[(size_t)gbest * (K + 1) + K]) gbest = g;
    if (gbest == 1 || !(best[(size_t)gbest * (K + 1) + K] <= 0.94 * whole)) return 1;
    int j = K;
    for (int g = gbest; g >= 1; --g) {
        const int i = from[(size_t)g * (K + 1) + j];
        len_out[g - 1] = j - i;
        j = i;
    }
    return gbest;
}

// Start of the iteration from the unscaled A' = W^2 + 4 beta I and B' = A'^2 (both already formed as products).
// cubic first step (mode 0/1):
//   Y1 = a0 Y0 T0 = (1.5 a0/c) A' - (0.5 a0^3/c^2) B',   Z1 = a0 T0 = 1.5 a0 I - (0.5 a0^3/c) A'   (Y0 = A'/c)
// quintic first step (mode 2):  Z1 = T1 = t0 I + (t1/c) A' + (t2/c^2) B'  only; Y1 = A' Z1 / c is a product.
// The spectral bound c may come from B' itself: lambda_max(A')^2 <= |A'^2|_inf, a fourth-root-of-W^4 bound
// that is 2-3x tighter than |W|_inf and saves one to two Newton-Schulz steps.  st[k] = {y1a, y1b, z1i, z1a, h}
// with h = 0.5 sqrt(c) when the start is already the end (one step: Omega = W/2 + h Y1); mode 2 reads
// {-, z1b, z1i, z1a, -}.
// degree-nine first step (mode 3):  Y1 <- U = (t3/c) A' + (t4/c^2) B' + t2 I   {ua, ub, ui, -, -};
// T1 = t0 I + (t1/c) A' + U B'/c^2  (the quartic t(m), m = A'/c, as t0 + t1 m + (t2 + t3 m + t4 m^2) m^2) and
// Y1 = A' T1 / c are products.  When the bound c is known before B' is formed (speculative step), U (degree nine) or
// Z1 (quintic) is the SECOND OUTPUT of the launch that forms B' and this kernel does not run at all.
__global__ __launch_bounds__(256) void k_ns_start(double* __restrict__ Y1, double* __restrict__ Z1,
                                                  const double* __restrict__ Ap, const double* __restrict__ Bp,
                                                  const double* __restrict__ W, const double* __restrict__ st, int p,
                                                  int mode)
{
    const int k = blockIdx.y;
    const size_t pp = (size_t)p * p, base = (size_t)k * pp;
    const double y1a = st[k * 5 + 0], y1b = st[k * 5 + 1], z1i = st[k * 5 + 2], z1a = st[k * 5 + 3], h = st[k * 5 + 4];
    size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int e = 0; e < 4; ++e, i += 256) {
        if (i < pp) {
            const double a = Ap[base + i], b2 = Bp[base + i];
            const int r = (int)(i / p), c = (int)(i - (size_t)r * p);
            if (mode == 2) {
                Z1[base + i] = __builtin_fma(z1a, a, y1b * b2) + (r == c ? z1i : 0.0);     // same order as c2val()
                continue;
            }
            if (mode == 3) {
                Y1[base + i] = __builtin_fma(y1a, a, y1b * b2) + (r == c ? z1i : 0.0);     // same order as c2val()
                continue;
            }
            const double y = y1a * a + y1b * b2;
            if (mode == 1) {
                Y1[base + i] = 0.5 * W[base + i] + h * y;          // Y1 is Omega here
            } else {
                Y1[base + i] = y;
                Z1[base + i] = z1a * a + (r == c ? z1i : 0.0);
            }
        }
    }
}

static void launch_ns_start(hipStream_t st, double* Y1, double* Z1, const double* Ap, const double* Bp, const double* W,
                            const double* stt, int K, int p, int mode)
{
    dim3 grid((unsigned)(((size_t)p * p + 1023) / 1024), K);
    hipLaunchKernelGGL(k_ns_start, grid, dim3(256), 0, st, Y1, Z1, Ap, Bp, W, stt, p, mode);
}

// cbound_h[k] >= lambda_max(A'_k).  Returns 0, -1 (non-finite) or -2 (condition number above NS_KAPPA_LIMIT).
// Coefficient slots are numbered from the first launch AFTER the start kernel; start_h: [K][5].
// One schedule for the whole batch, built for the smallest l_k = sqrt(4 beta_k / c_k) (every spectrum lies in
// [l_k, 1] after scaling by c_k, so the polynomials of the widest interval converge for all of them).
int ns_plan(const double* cbound_h, const double* beta_h, int K, double* coef_h, double* start_h, NsPlan* plan,
            int force_mode, int degrees, double tol)
{
    std::vector<double> c(K);
    double kappa = 1.0;
    for (int k = 0; k < K; ++k) {
        c[k] = cbound_h[k] * (1.0 + 1e-10);
        if (!(c[k] > 0.0) || !std::isfinite(c[k]) || !(beta_h[k] > 0.0)) return -1;
        if (c[k] < 4.0 * beta_h[k]) c[k] = 4.0 * beta_h[k];          // lambda_min(A') = 4 beta is exact
        kappa = std::fmax(kappa, c[k] / (4.0 * beta_h[k]));
    }
    const bool stable = (force_mode == 2) || (force_mode == 0 && kappa > NS_SYM_KAPPA_MAX);
    // Beyond this the schedule would be cut off at NS_MAX_STEPS and fp64 (error ~ eps*sqrt(kappa)) could not
    // deliver the accuracy anyway: the caller takes the eigendecomposition route for this call.
    if (kappa > NS_KAPPA_LIMIT) return -2;
    const double lmin = 1.0 / std::sqrt(kappa);
    plan->stable = stable;
    plan->kappa = kappa;
    auto put = [&](int g, int k, double cI, double cAcc, double cE, double dI = 0.0, double dC = 0.0, double dE = 0.0) {
        double* o = coef_h + (size_t)g * NS_SLOT(K) + (size_t)k * NS_NCOEF;
        o[0] = cI; o[1] = cAcc; o[2] = cE; o[3] = dI; o[4] = dC; o[5] = dE;
    };
    if (!stable) {
        const NsSeq& sq = ns_mixed_schedule(lmin, degrees, tol);
        if (sq.n < 1 || sq.n > NS_MAX_STEPS || sq.cost >= (1 << 29)) return -2;
        const int n = sq.n;
        plan->steps = n;
        plan->units = 2 + sq.cost;
        int g = 0;
        for (int it = 0; it < n; ++it) plan->deg[it] = sq.deg[it];
        for (int k = 0; k < K; ++k) {
            const double ck = c[k], sc = std::sqrt(ck);
            double* s = start_h + (size_t)k * 5;
            const double* t = sq.st[0].t;
            g = 0;
            if (sq.deg[0] == 3) {
                s[0] = t[0] / ck; s[1] = t[1] / (ck * ck); s[2] = t[0]; s[3] = t[1] / ck; s[4] = 0.5 * sc;
            } else {
                if (sq.deg[0] == 5) {
                    s[0] = 0.0; s[1] = t[2] / (ck * ck); s[2] = t[0]; s[3] = t[1] / ck; s[4] = 0.0;
                } else {
                    // t(m) = t0 + t1 m + (t2 + t3 m + t4 m^2) m^2,  m = A'/c,  m^2 = B'/c^2
                    s[0] = t[3] / ck; s[1] = t[4] / (ck * ck); s[2] = t[2]; s[3] = 0.0; s[4] = 0.0;
                    put(g++, k, t[0], 1.0 / (ck * ck), t[1] / ck);    // Z1 = t0 I + (U B')/c^2 + (t1/c) A'
                }
                if (n == 1) put(g++, k, 0.0, 0.5 * sc / ck, 0.5);     // Omega = W/2 + sqrt(c)/2 (A'/c) Z1
                else put(g++, k, 0.0, 1.0 / ck, 0.0);                 // Y1 = (A'/c) Z1
            }
            for (int it = 1; it < n; ++it) {
                t = sq.st[it].t;
                if (sq.deg[it] == 3) {
                    put(g++, k, t[0], t[1], 0.0);                    // T = t0 I + t1 (Z Y)
                } else if (sq.deg[it] == 5) {
                    put(g++, k, 0.0, 1.0, 0.0);                      // M = Z Y
                    put(g++, k, t[0], t[2], t[1]);                   // T = t0 I + t2 (M M) + t1 M
                } else {
                    const double a = t[3] / (2.0 * t[4]), d = t[2] / t[4] - a * a, e = t[1] - t[4] * d * a;
                    put(g++, k, 0.0, 1.0, 0.0);                      // M = Z Y
                    put(g++, k, 0.0, 1.0, a, d, 1.0);                // Q = M M + a M;  Q2 = Q + d I
                    put(g++, k, t[0], t[4], e);                      // T = f I + t4 (Q Q2) + e M
                }
                if (it == n - 1) put(g++, k, 0.0, 0.5 * sc, 0.5);    // Omega = W/2 + sqrt(c) (Y T)/2
                else {
                    // one launch, 2K instances: [Y <- Y T ; Z <- T Z]; its slot holds 2K coefficient rows
                    put(g, k, 0.0, 1.0, 0.0);
                    put(g, K + k, 0.0, 1.0, 0.0);
                    ++g;
                }
            }
        }
        plan->products = 2 + g;
        return 0;
    }
    // stable schedule: cubic steps, per-instance scaling
    std::vector<std::vector<double>> al(K);
    int n = 1;
    for (int k = 0; k < K; ++k) {
        al[k] = ns_schedule(std::sqrt(4.0 * beta_h[k] / c[k]));
        n = std::max(n, (int)al[k].size());
    }
    plan->steps = n;
    plan->products = 2 + 2 * (n - 1);      // kernel launches of symmetric / right-multiply products (incl. A', B')
    plan->units = (n == 1) ? 2 : 5 * n - 6;
    for (int it = 0; it < n && it < NS_MAX_STEPS; ++it) plan->deg[it] = 3;
    for (int k = 0; k < K; ++k) {
        auto a_of = [&](int it) { return it < (int)al[k].size() ? al[k][it] : 1.0; };
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
            put(g++, k, 1.5, -0.5 * a * a, 0.0);                             // T = 1.5 I - 0.5 a^2 (P^T Y)
            if (it == n - 1) put(g++, k, 0.0, 0.5 * sc * a, 0.5);           // Omega = W/2 + sqrt(c) a (Y T)/2
            else {
                double* o = coef_h + (size_t)g * NS_SLOT(K);                // [Y <- a Y T ; P <- a P T]
                o[k] = a;
                o[K + k] = a;
                ++g;
            }
        }
    }
    return 0;
}

// Phase A (before the bound is known): A' = W^2 + 4 beta I -> Ap, B' = A'^2 -> Bp.
// pre0_d / pre1_d: the coefficient rows {4 beta, 1, 0, 0, 0} and {0, 1, 0, 0, 0} of the K instances.
void ns_prepare(hipStream_t st, const double* pre0_d, const double* pre1_d, const double* W, double* Ap, double* Bp, int K,
                int p, int variant, double* start2, double* rowpart, double* fropart)
{
    // start2 != null (the bound is already known): the B' launch also writes start2 = dI I + dC B' + dE A' -- the
    // first step's U (degree nine) or Z1 (quintic) -- with {dI, dC, dE} in pre1_d
    launch_symm(st, W, W, Ap, nullptr, nullptr, pre0_d, K, p, variant);
    symm_flush_rider(st);         // (tables riding in the A' launch: a launch of another kernel family did not take them)
    launch_symm(st, Ap, Ap, Bp, start2, start2 ? Ap : nullptr, pre1_d, K, p, variant, nullptr, rowpart, fropart);
}

// where the first step's elementwise start goes (ns_run's layout) and its coefficients {dI, dC, dE} from the start
// table of ns_plan; null when the first step has no single-matrix start (cubic first step, stable schedule)
double* ns_fused_start(const NsPlan& plan, const double* start_hk, double* YP, double* Tb, size_t n1, double out3[3])
{
    if (plan.stable || plan.deg[0] < 5) return nullptr;
    if (plan.deg[0] == 5) { out3[0] = start_hk[2]; out3[1] = start_hk[1]; out3[2] = start_hk[3]; return YP + n1; }   // Z1
    out3[0] = start_hk[2]; out3[1] = start_hk[1]; out3[2] = start_hk[0];                                              // U
    return Tb;
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
            double* AB, double* YP, double* Tb, double* out, int K, int p, int variant, size_t pstride, bool fused_start,
            hipEvent_t bprime_free)
{
    // bprime_free (speculative steps with GGL_OPT_BOUND_SIDE): the bound kernels that validate the assumed bound read B'
    // (AB + n1) on a SIDE stream while this chain's first products run; the event is waited for before the first launch
    // that overwrites B', or at the end
    bool bwaited = bprime_free == nullptr;
    const double* bprime = AB + (pstride ? pstride : (size_t)K * p * p);
    auto before_write = [&](const double* o1, const double* o2) {
        if (!bwaited && (o1 == bprime || o2 == bprime)) { (void)hipStreamWaitEvent(st, bprime_free, 0); bwaited = true; }
    };
    struct AtExit { std::function<void()> f; ~AtExit() { f(); } } at_exit{[&]() { before_write(bprime, nullptr); }};
    // fused_start: the first step's elementwise start (ns_fused_start) was written by ns_prepare's B' launch
    // pstride: distance (doubles) between the two stacks of a [Y|Z] pair; 0 = contiguous (K*p*p).  A sub-batch
    // of a larger ctx (two-stream execution) passes the full-stack stride.
    const size_t cs = NS_SLOT(K), n1 = pstride ? pstride : (size_t)K * p * p;
    const int n = plan.steps;
    int g = 0;
    double *cur = YP, *nxt = AB;      // cur = [Y | Z]
    if (plan.deg[0] >= 5 && !plan.stable) {
        if (plan.deg[0] == 5) {
            // quintic first step: Z1 = T1 elementwise
            if (!fused_start)
                launch_ns_start(st, nullptr, YP + n1, AB, AB + n1, W, start_d, K, p, 2);
        } else {
            // degree nine: U -> Tb elementwise, Z1 = T1 = t0 I + U B'/c^2 + (t1/c) A'
            if (!fused_start)
                launch_ns_start(st, Tb, nullptr, AB, AB + n1, W, start_d, K, p, 3);
            launch_symm(st, Tb, AB + n1, YP + n1, nullptr, AB, coef_d + cs * g++, K, p, variant);
        }
        // Y1 = (A'/c) Z1 (or Omega directly when it is the only step)
        if (n == 1) {
            launch_symm(st, AB, YP + n1, out, nullptr, W, coef_d + cs * g++, K, p, variant);
            return;
        }
        launch_symm(st, AB, YP + n1, YP, nullptr, nullptr, coef_d + cs * g++, K, p, variant);
    } else {
        if (n == 1) {
            launch_ns_start(st, out, nullptr, AB, AB + n1, W, start_d, K, p, 1);
            return;
        }
        launch_ns_start(st, YP, YP + n1, AB, AB + n1, W, start_d, K, p, 0);
    }
    for (int it = 1; it < n; ++it) {
        if (plan.deg[it] == 5 && !plan.stable) {
            // M = Z Y into the (free) Y slot of the other pair, then T = t0 I + t1 M + t2 M^2
            launch_symm(st, cur + n1, cur, nxt, nullptr, nullptr, coef_d + cs * g++, K, p, variant);
            launch_symm(st, nxt, nxt, Tb, nullptr, nxt, coef_d + cs * g++, K, p, variant);
        } else if (plan.deg[it] == 9 && !plan.stable) {
            // M = Z Y into the output stack (scratch until the last launch), Q and Q + d I into the other pair,
            // T = f I + t4 Q (Q + d I) + e M
            launch_symm(st, cur + n1, cur, out, nullptr, nullptr, coef_d + cs * g++, K, p, variant);
            before_write(nxt, nxt + n1);
            launch_symm(st, out, out, nxt, nxt + n1, out, coef_d + cs * g++, K, p, variant);
            launch_symm(st, nxt, nxt + n1, Tb, nullptr, out, coef_d + cs * g++, K, p, variant);
        } else {
            // T = t0 I + t1 (Z Y)   [fast: Z Y = Z^T Y, Z symmetric; stable: P^T Y]
            launch_symm(st, cur + n1, cur, Tb, nullptr, nullptr, coef_d + cs * g++, K, p, variant);
        }
        if (it == n - 1) {
            launch_symm(st, cur, Tb, out, nullptr, W, coef_d + cs * g++, K, p, variant);
        } else if (plan.stable) {
            before_write(nxt + n1, nullptr);
            launch_gemm_right(st, cur, Tb, nxt, coef_d + cs * g++, 2 * K, K, p, 0);   // needs a contiguous pair
            std::swap(cur, nxt);
        } else {
            before_write(nxt, nxt + n1);
            launch_symm_pair(st, cur, Tb, nxt, Tb, cur + n1, nxt + n1, coef_d + cs * g++, K, p, variant);
            std::swap(cur, nxt);
        }
    }
}

#ifdef GGL_DEV
// The launches of ns_prepare + ns_run (all-symmetric schedule, first step's start fused into the B' launch) as a list of
// product descriptors for k_omega_chain (gemm_sym.hip) -- same operands, same coefficient slots, same order, hence the same
// bits.  One difference in WHERE things live: the iteration ping-pongs between YP and the extra pair NX instead of YP and
// AB, so A' and B' survive the chain and the bound kernels can read B' after it.  Returns the number of ops, 0 when the
// plan is not a pure chain of symmetric products (stable schedule, cubic first step) or does not fit max_ops.
int ns_chain_ops(const NsPlan& plan, const double* pre0_d, const double* pre1_d, const double* coef_d, const double* W,
                 double* AB, double* YP, double* NX, double* Tb, double* out, int K, int p, size_t pstride, double* start2,
                 double* rowpart, double* fropart, SymmOp* ops, int max_ops)
{
    if (plan.stable || plan.deg[0] < 5 || !start2) return 0;
    const size_t cs = NS_SLOT(K), n1 = pstride ? pstride : (size_t)K * p * p;
    const int n = plan.steps;
    int no = 0, g = 0;
    bool ok = true;
    auto add = [&](const double* A, const double* B, double* C, double* C2, const double* E, const double* coef,
                   double* rp = nullptr, double* fp = nullptr) -> SymmOp* {
        if (no >= max_ops) { ok = false; return nullptr; }
        SymmOp& o = ops[no++];
        o = SymmOp{};
        o.A = A; o.B = B; o.C = C; o.C2 = C2; o.E = E; o.coef = coef; o.rowpart = rp; o.fropart = fp;
        return &o;
    };
    double* Ap = AB;
    double* Bp = AB + n1;
    add(W, W, Ap, nullptr, nullptr, pre0_d);
    add(Ap, Ap, Bp, start2, Ap, pre1_d, rowpart, fropart);
    double *cur = YP, *nxt = NX;
    if (plan.deg[0] == 9) add(Tb, Bp, YP + n1, nullptr, Ap, coef_d + cs * g++);
    if (n == 1) {
        add(Ap, YP + n1, out, nullptr, W, coef_d + cs * g++);
        return ok ? no : 0;
    }
    add(Ap, YP + n1, YP, nullptr, nullptr, coef_d + cs * g++);
    for (int it = 1; it < n; ++it) {
        if (plan.deg[it] == 5) {
            add(cur + n1, cur, nxt, nullptr, nullptr, coef_d + cs * g++);
            add(nxt, nxt, Tb, nullptr, nxt, coef_d + cs * g++);
        } else if (plan.deg[it] == 9) {
            add(cur + n1, cur, out, nullptr, nullptr, coef_d + cs * g++);
            add(out, out, nxt, nxt + n1, out, coef_d + cs * g++);
            add(nxt, nxt + n1, Tb, nullptr, out, coef_d + cs * g++);
        } else {
            add(cur + n1, cur, Tb, nullptr, nullptr, coef_d + cs * g++);
        }
        if (it == n - 1) {
            add(cur, Tb, out, nullptr, W, coef_d + cs * g++);
        } else {
            SymmOp* o = add(cur, Tb, nxt, nullptr, nullptr, coef_d + cs * g++);
            if (o) { o->A1 = Tb; o->B1 = cur + n1; o->C1 = nxt + n1; o->pair = 1; }
            std::swap(cur, nxt);
        }
    }
    return ok ? no : 0;
}
#endif   // GGL_DEV (ns_chain_ops)

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
__global__ __launch_bounds__(256) void k_norm_bounds(const double* __restrict__ W, int p, double* __restrict__ part,
                                                      double* __restrict__ rowsum)
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
    for (int q = 0; q < 4; ++q) {
        const double rs = wave_sum(a[q]);
        mx = fmax(mx, rs);
        if (rowsum && lane == 0 && ok[q]) rowsum[(size_t)k * p + r0 + q] = rs;
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

int norm_bounds_blocks(int p) { return (p + 15) / 16; }

// part[k][blk] = {max row abs-sum, sum of squares} over row block blk (host finishes the reduction)
void launch_norm_bounds(hipStream_t st, const double* W, int K, int p, double* part, double* rowsum)
{
    hipLaunchKernelGGL(k_norm_bounds, dim3(norm_bounds_blocks(p), K), dim3(256), 0, st, W, p, part, rowsum);
}

// Collatz-Wielandt bound: for any positive vector d,  lambda_max(B) <= rho(|B|) <= max_i (|B| d)_i / d_i.
// With d = the row sums of |B| (one power step towards the Perron vector of |B|) this is markedly tighter than
// |B|_inf = max_i d_i: at the headline workload the scaled lower end l of the spectrum rises from ~0.55 to ~0.67,
// which is what lets three quintic steps converge.  part[k][blk] = max ratio over row block blk.
__global__ __launch_bounds__(256) void k_cw_bounds(const double* __restrict__ W, const double* __restrict__ d, int p,
                                                    double* __restrict__ part)
{
    __shared__ double sh[4];
    const int k = blockIdx.y;
    const double* w = W + (size_t)k * p * p;
    const double* dk = d + (size_t)k * p;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * 16 + wave * 4;
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    size_t ro[4];
    bool ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        ok[q] = (r0 + q) < p;
        ro[q] = (size_t)min(r0 + q, p - 1) * p;
    }
    for (int j = lane; j < p; j += 64) {
        const double dj = dk[j];
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] += fabs(w[ro[q] + j]) * dj;
    }
    double mx = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const double y = wave_sum(a[q]);
        if (ok[q]) mx = fmax(mx, y / dk[r0 + q]);
    }
    if (lane == 0) sh[wave] = mx;
    __syncthreads();
    if (threadIdx.x == 0) part[(size_t)k * gridDim.x + blockIdx.x] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}

// out[k] = sqrt(min(max_b part2[k][b][0], cw bound, sqrt(sum_b part2[k][b][1])))   (mode 0: spectral bound of A' from B')
//        = min(max_b part2[k][b][0], sqrt(sum_b part2[k][b][1]))                   (mode 1: norm bound of C)
// `out` may be pinned host memory: K doubles.
__global__ __launch_bounds__(256) void k_bound_final(const double* __restrict__ part2, const double* __restrict__ cwpart,
                                                     int nbb, int K, double* __restrict__ out, int mode,
                                                     const double* __restrict__ cuse, int* __restrict__ flag,
                                                     int* __restrict__ flag_host)
{
    for (int k = blockIdx.x * 256 + threadIdx.x; k < K; k += gridDim.x * 256) {
        double mx = 0.0, sq = 0.0, cw = 0.0;
        for (int b = 0; b < nbb; ++b) {
            const double m = part2[2 * ((size_t)k * nbb + b)];
            mx = (mx < m) ? m : mx;
            sq += part2[2 * ((size_t)k * nbb + b) + 1];
            if (cwpart) { const double w = cwpart[(size_t)k * nbb + b]; cw = (cw < w) ? w : cw; }
        }
        const double fr = sqrt(sq);
        if (mode == 0) {
            // the ratio is computed in floating point from ~p terms, hence the small inflation
            if (cwpart && isfinite(cw) && cw > 0.0) { const double w = cw * (1.0 + 1e-12); mx = (w < mx) ? w : mx; }
            const double b = sqrt((fr < mx) ? fr : mx);
            out[k] = b;
            // the schedule already running was built for a spectrum inside [4 beta, cuse[k]]
            if (flag && !(b <= cuse[k])) {
                atomicOr(flag, 1);          // for the kernels of this step
                *flag_host = 1;             // pinned: for the host, after its stream sync
            }
        } else {
            out[k] = (fr < mx) ? fr : mx;
        }
    }
}

void launch_bound_final(hipStream_t st, const double* part2, const double* cwpart, int nbb, int K, double* out, int mode,
                        const double* cuse, int* flag, int* flag_host)
{
    hipLaunchKernelGGL(k_bound_final, dim3((K + 255) / 256), dim3(256), 0, st, part2, cwpart, nbb, K, out, mode, cuse,
                       flag, flag_host);
}

// ---- the same bound from the partials the product launch leaves behind (launch_symm rowpart / fropart) -------------
// d[k][i] = sum_s rowpart[k][s][i] (row sums of |B'|, fixed order) and infpart[k][blk] = max of d over the block's rows
__global__ __launch_bounds__(256) void k_bound_rows(const double* __restrict__ rowpart, int T, int p,
                                                     double* __restrict__ d, double* __restrict__ infpart)
{
    __shared__ double sh[4];
    const int k = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    double v = 0.0;
    if (i < p) {
        const double* rp = rowpart + (size_t)k * T * p + i;
        for (int s2 = 0; s2 < T; ++s2) v += rp[(size_t)s2 * p];
        d[(size_t)k * p + i] = v;
    }
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) infpart[(size_t)k * gridDim.x + blockIdx.x] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}

int bound_rows_blocks(int p) { return (p + 255) / 256; }

void launch_bound_rows(hipStream_t st, const double* rowpart, int T, int K, int p, double* d, double* infpart)
{
    hipLaunchKernelGGL(k_bound_rows, dim3(bound_rows_blocks(p), K), dim3(256), 0, st, rowpart, T, p, d, infpart);
}

// Collatz-Wielandt ratios max_i (|B'| d)_i / d_i and -- by the last workgroup of an instance to finish --
// the bound itself: out[k] = sqrt(min(|B'|_inf, cw (1 + 1e-12), |B'|_F)), compared with the bound the running schedule
// assumes (cuse).  The maximum over the row blocks is an atomic max on the bit pattern of a non-negative double (order
// independent, hence deterministic); |B'|_F^2 is summed in tile order.  cwmax / cnt: [K], zero on entry, left zero.
// dprev / dnext (optional): the Collatz-Wielandt ratio max_i (|B'| v)_i / v_i bounds the Perron root of |B'| from above for
// ANY positive v; with v = the row sums it is one power step away from the infinity norm, with v = the Perron vector it IS
// the root.  B' moves little between ADMM iterations, so the ctx keeps v across them: this pass reads the previous
// iteration's vector (dprev, if there is one) and leaves (|B'| v) / |B'|_inf behind for the next (dnext) -- a power
// iteration at one step per ADMM iteration and no extra pass, still a rigorous bound at every step.
__global__ __launch_bounds__(256) void k_cw_final(const double* __restrict__ B, const double* __restrict__ d,
                                                   const double* __restrict__ dprev, double* __restrict__ dnext, int p,
                                                   const double* __restrict__ infpart, int ninf,
                                                   const double* __restrict__ fropart, int ntile,
                                                   unsigned long long* __restrict__ cwmax, unsigned* __restrict__ cnt,
                                                   double* __restrict__ out, const double* __restrict__ cuse,
                                                   int* __restrict__ flag, int* __restrict__ flag_host, int flag_slot)
{
    // 16 rows per workgroup, 4 per wave, all four streamed together (independent loads in flight)
    __shared__ double sh[4];
    __shared__ double shfro[256];
    __shared__ int shlast;
    const int k = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * 16 + wave * 4;
    const double* w = B + (size_t)k * p * p;
    const double* dk = (dprev ? dprev : d) + (size_t)k * p;
    double scale = 0.0;
    if (dnext) {
        for (int b2 = 0; b2 < ninf; ++b2) scale = fmax(scale, infpart[(size_t)k * ninf + b2]);
        scale = 1.0 / scale;
    }
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    size_t ro[4];
    bool ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        ok[q] = (r0 + q) < p;
        ro[q] = (size_t)min(r0 + q, p - 1) * p;
    }
    for (int j = lane; j < p; j += 64) {
        const double dj = dk[j];
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] += fabs(w[ro[q] + j]) * dj;
    }
    double mx = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const double y = wave_sum(a[q]);
        if (ok[q]) {
            mx = fmax(mx, y / dk[r0 + q]);              // a zero row gives 0/0: fmax drops the NaN
            if (dnext && lane == 0) {
                const double v = y * scale;             // anything positive keeps the bound rigorous
                dnext[(size_t)k * p + r0 + q] = (v > 0.0 && isfinite(v)) ? v : 1.0;
            }
        }
    }
    if (lane == 0) sh[wave] = mx;
    __syncthreads();
    // Everything the workgroups exchange travels in agent-scope atomics (performed at the memory side, coherent across
    // the XCDs' L2s) -- no cache write-back / invalidate fences: a __threadfence() per workgroup writes back the whole XCD
    // L2, which the product kernels of the other part keep dirtying (measured: +150 us per iteration).  One atomic max
    // and one arrival per workgroup (thousands of atomics on sixteen addresses serialise: measured 74 us for one per
    // row).  The arrival depends on the RETURNED value of the max, so it cannot be performed before it.
    if (threadIdx.x == 0) {
        mx = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
        const unsigned long long old = __hip_atomic_fetch_max(cwmax + k, (unsigned long long)__double_as_longlong(mx),
                                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned arrive = 1u + (unsigned)(old >> 63);               // old is a non-negative double: + 0
        shlast = __hip_atomic_fetch_add(cnt + k, arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    }
    __syncthreads();
    if (!shlast) return;
    // last workgroup of instance k: every block maximum has been merged.  The Frobenius shares are fetched by all threads
    // at once and added by one in tile order (round 5: the event timeline showed this kernel at 12.6 us on (20, 200) and
    // 18.8 us on a K = 4 slab of p = 500 -- T (T + 1) / 2 = 36 dependent loads by one thread were most of the second figure).
    double cw = 0.0;
    if (threadIdx.x == 0) {
        cw = __longlong_as_double((long long)__hip_atomic_exchange(cwmax + k, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        __hip_atomic_store(cnt + k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    double sq = 0.0;
    for (int t0 = 0; t0 < ntile; t0 += 256) {
        const int t = t0 + (int)threadIdx.x;
        shfro[threadIdx.x] = (t < ntile) ? fropart[(size_t)k * ntile + t] : 0.0;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int m = min(256, ntile - t0);
            for (int q = 0; q < m; ++q) sq += shfro[q];
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    double inf = 0.0;
    for (int b2 = 0; b2 < ninf; ++b2) inf = fmax(inf, infpart[(size_t)k * ninf + b2]);
    const double fr = sqrt(sq);
    if (isfinite(cw) && cw > 0.0) { const double wv = cw * (1.0 + 1e-12); inf = (wv < inf) ? wv : inf; }
    const double b = sqrt((fr < inf) ? fr : inf);
    out[k] = b;
    if (flag && !(b <= cuse[k])) {
        atomicOr(flag + flag_slot, 1);
        flag_host[flag_slot] = 1;
    }
}

void launch_cw_final(hipStream_t st, const double* B, const double* d, int K, int p, const double* infpart,
                     const double* fropart, int ntile, unsigned long long* cwmax, unsigned* cnt, double* out,
                     const double* cuse, int* flag, int* flag_host, int flag_slot, const double* dprev, double* dnext)
{
    hipLaunchKernelGGL(k_cw_final, dim3((p + 15) / 16, K), dim3(256), 0, st, B, d, dprev, dnext, p, infpart,
                       bound_rows_blocks(p), fropart, ntile, cwmax, cnt, out, cuse, flag, flag_host, flag_slot);
}

// k_bound_rows + k_cw_final as ONE launch (round 4, VERDICT r3 item 6: the K = 4 slab's iteration carried the two as 5.7 + 14.2
// us of its 216): every workgroup adds up the row sums d_j = sum_s rowpart[k][s][j] of ITS instance itself (T * p loads out of
// L2 -- 64 KB at p = 500 -- against a kernel boundary) and takes |B'|_inf = max_j d_j from them; RW rows per wave instead of
// four (a slab of K = 4 has 2000 rows: 125 workgroups of 16 rows were half a workgroup per CU); the last workgroup of an
// instance sums the Frobenius shares with all its threads (fixed order) instead of one.  Same arithmetic, same bound, same bits
// in the Collatz-Wielandt vector as the two-kernel form.  MEASURED (interleaved A/B, bench.py --opt fused_cw=0/1): K = 4 slab
// 4654 / 4892 it/s without against 4602 / 4774 with, K = 8 3158 / 3258 against 3178 / 3358, (20,200) 5270 / 5340 against 5384 /
// 5267, headline 1340 / 1307 against 1214 / 1253 -- no gain where it was meant to help, a loss where the bound kernels run in
// the shadow of the other part's products (every workgroup re-adding the row sums is work on the chain's critical path).
// GGL_OPT_FUSED_CW, default off.
template <int RW>
__global__ __launch_bounds__(256) void k_bound_cw(const double* __restrict__ B, const double* __restrict__ rowpart, int T,
                                                  double* __restrict__ d_out, const double* __restrict__ dprev,
                                                  double* __restrict__ dnext, int p, const double* __restrict__ fropart,
                                                  int ntile, unsigned long long* __restrict__ cwmax, unsigned* __restrict__ cnt,
                                                  double* __restrict__ out, const double* __restrict__ cuse,
                                                  int* __restrict__ flag, int* __restrict__ flag_host, int flag_slot)
{
    extern __shared__ __attribute__((aligned(16))) double dl[];          // [p] row sums | [8] scratch
    double* sh = dl + p;
    __shared__ int last;
    const int k = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double mxd = 0.0;
    for (int j = threadIdx.x; j < p; j += 256) {
        const double* rp = rowpart + (size_t)k * T * p + j;
        double v = 0.0;
        for (int s2 = 0; s2 < T; ++s2) v += rp[(size_t)s2 * p];
        dl[j] = v;
        mxd = fmax(mxd, v);
        if (blockIdx.x == 0 && d_out) d_out[(size_t)k * p + j] = v;
    }
    mxd = wave_max(mxd);
    if (lane == 0) sh[wave] = mxd;
    __syncthreads();
    const double inf0 = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
    __syncthreads();
    const double scale = 1.0 / inf0;
    const double* dg = dprev ? dprev + (size_t)k * p : nullptr;
    const int r0 = (blockIdx.x * 4 + wave) * RW;
    const double* w = B + (size_t)k * p * p;
    double a[RW];
    size_t ro[RW];
    bool ok[RW];
#pragma unroll
    for (int q = 0; q < RW; ++q) {
        a[q] = 0.0;
        ok[q] = (r0 + q) < p;
        ro[q] = (size_t)min(r0 + q, p - 1) * p;
    }
    for (int j = lane; j < p; j += 64) {
        const double dj = dg ? dg[j] : dl[j];
#pragma unroll
        for (int q = 0; q < RW; ++q) a[q] += fabs(w[ro[q] + j]) * dj;
    }
    double mx = 0.0;
#pragma unroll
    for (int q = 0; q < RW; ++q) {
        const double y = wave_sum(a[q]);
        if (ok[q]) {
            mx = fmax(mx, y / (dg ? dg[r0 + q] : dl[r0 + q]));       // a zero row gives 0/0: fmax drops the NaN
            if (dnext && lane == 0) {
                const double v = y * scale;                           // anything positive keeps the bound rigorous
                dnext[(size_t)k * p + r0 + q] = (v > 0.0 && isfinite(v)) ? v : 1.0;
            }
        }
    }
    if (lane == 0) sh[wave] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
        // (agent-scope atomics carry everything the workgroups exchange: see k_cw_final)
        const unsigned long long old =
            __hip_atomic_fetch_max(cwmax + k, (unsigned long long)__double_as_longlong(mx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned arrive = 1u + (unsigned)(old >> 63);
        last = (__hip_atomic_fetch_add(cnt + k, arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!last) return;
    // last workgroup of instance k: |B'|_F^2 in a fixed order (strided per-thread sums, wave sums, four waves)
    double sq = 0.0;
    for (int t = threadIdx.x; t < ntile; t += 256) sq += fropart[(size_t)k * ntile + t];
    sq = wave_sum(sq);
    __syncthreads();
    if (lane == 0) sh[wave] = sq;
    __syncthreads();
    if (threadIdx.x != 0) return;
    sq = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    const double cw = __longlong_as_double((long long)__hip_atomic_exchange(cwmax + k, 0ull, __ATOMIC_RELAXED,
                                                                            __HIP_MEMORY_SCOPE_AGENT));
    __hip_atomic_store(cnt + k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double inf = inf0;
    const double fr = sqrt(sq);
    if (isfinite(cw) && cw > 0.0) { const double wv = cw * (1.0 + 1e-12); inf = (wv < inf) ? wv : inf; }
    const double b = sqrt((fr < inf) ? fr : inf);
    out[k] = b;
    if (flag && !(b <= cuse[k])) {
        atomicOr(flag + flag_slot, 1);
        flag_host[flag_slot] = 1;
    }
}

void launch_bound_cw(hipStream_t st, const double* B, const double* rowpart, int T, int K, int p, double* d_out,
                     const double* fropart, int ntile, unsigned long long* cwmax, unsigned* cnt, double* out,
                     const double* cuse, int* flag, int* flag_host, int flag_slot, const double* dprev, double* dnext)
{
    const size_t lds = ((size_t)p + 8) * sizeof(double);
    // rows per workgroup: 16 while that still gives every CU a few workgroups, 4 for the small slabs
    const long wg16 = (long)((p + 15) / 16) * K;
    if (wg16 >= 1024)
        hipLaunchKernelGGL(k_bound_cw<4>, dim3((p + 15) / 16, K), dim3(256), lds, st, B, rowpart, T, d_out, dprev, dnext, p, fropart,
                           ntile, cwmax, cnt, out, cuse, flag, flag_host, flag_slot);
    else
        hipLaunchKernelGGL(k_bound_cw<1>, dim3((p + 3) / 4, K), dim3(256), lds, st, B, rowpart, T, d_out, dprev, dnext, p, fropart,
                           ntile, cwmax, cnt, out, cuse, flag, flag_host, flag_slot);
}

void launch_cw_bounds(hipStream_t st, const double* W, const double* rowsum, int K, int p, double* part)
{
    hipLaunchKernelGGL(k_cw_bounds, dim3(norm_bounds_blocks(p), K), dim3(256), 0, st, W, rowsum, p, part);
}

// ---- the L-step's norm bound from C^2 -------------------------------------------------------------------------------------
// The sign iteration scales with a bound nb >= |C - mu I|_2 and its resolution is relative to nb, so every factor 2.6 of slack
// in the bound costs a cubic step (two products).  min(|C|_inf, |C|_F) is ~10x the spectral radius on the C = Theta - X - Omega
// of an ADMM run (profiles/r3_two_tier_lstep.txt); sqrt(min(|C^2|_inf, |C^2|_F)) is 2.4x -- and C^2 is the iteration's first
// product anyway.  So: P = C C as a plain product whose epilogue leaves the row sums and Frobenius shares of P behind
// (launch_symm rowpart / fropart), launch_bound_rows + this kernel turn them into b_k >= rho(C_k), the host plans with
// nb_k = b_k + mu_k, and T0 = cI I + cAcc P + cE C (the first launch's affine epilogue, same arithmetic) is formed in place.
__global__ void k_bound_sqrt_inf_fro(const double* __restrict__ infpart, int ninf, const double* __restrict__ fropart,
                                     int ntile, int K, double* __restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    double inf = 0.0, sq = 0.0;
    for (int b = 0; b < ninf; ++b) inf = fmax(inf, infpart[(size_t)k * ninf + b]);
    for (int t = 0; t < ntile; ++t) sq += fropart[(size_t)k * ntile + t];
    const double fr = sqrt(sq);
    out[k] = sqrt((fr < inf ? fr : inf)) * (1.0 + 1e-12);          // rho(C)^2 = rho(C^2) <= min(|C^2|_inf, |C^2|_F)
}

void launch_bound_sqrt_inf_fro(hipStream_t st, const double* infpart, int ninf, const double* fropart, int ntile, int K,
                               double* out)
{
    hipLaunchKernelGGL(k_bound_sqrt_inf_fro, dim3((K + 63) / 64), dim3(64), 0, st, infpart, ninf, fropart, ntile, K, out);
}

// T0 = cI I + cAcc P + cE C in place over P; coef: slot 0 of rank_ns_plan ({cI, cAcc, cE, ...} per instance)
__global__ __launch_bounds__(256) void k_rank_t0(double* __restrict__ P, const double* __restrict__ C,
                                                 const double* __restrict__ coef, int p)
{
    const int k = blockIdx.y;
    const double cI = coef[k * NS_NCOEF + 0], cAcc = coef[k * NS_NCOEF + 1], cE = coef[k * NS_NCOEF + 2];
    const size_t pp = (size_t)p * p, base = (size_t)k * pp;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < pp; e += (size_t)gridDim.x * 256) {
        double v = cAcc * P[base + e];
        if (e % ((size_t)p + 1) == 0) v += cI;
        v += cE * C[base + e];
        P[base + e] = v;
    }
}

void launch_rank_t0(hipStream_t st, double* P, const double* C, const double* coef_d, int K, int p)
{
    const size_t pp = (size_t)p * p;
    const int bx = (int)std::min<size_t>((pp + 255) / 256, 512);
    hipLaunchKernelGGL(k_rank_t0, dim3(bx, K), dim3(256), 0, st, P, C, coef_d, p);
}

// Residual check of the sign iteration's result: an eigenvalue x of the iterate before the last step shows up as
// |t(x^2) - 1| ~ e = 1 - |x| in T_last, and the last step turns it into a sign error of 1.5 e^2 (cubic), 2.5 e^3 (quintic) or
// 7.9 e^5 (degree nine).  An eigenvalue the schedule leaves short of convergence sits within l0 |B| of the threshold (the
// schedule is planned for everything further away), so its contribution to L is |lambda - mu| * error <= l0 |B| * error:
// accept a sign error of 4e-9 (e up to the values below).  NOTE what this check sees: the ENTRYWISE maximum of T_last - I,
// which for one unresolved eigenvalue with eigenvector v is its residual times max_i v_i^2 -- ~2 ln(p) / p of it for a
// delocalised vector.  It is a coarse net; the sharp one is rank_trace_tolerance() below.
static double rank_check(int deg_last, double l0)
{
    (void)l0;
    return (deg_last == 3) ? 5e-5 : (deg_last == 5 ? 1.1e-3 : 8e-3);
}

// The sharp check: the iterate converges to sign(C - mu I), whose trace is an INTEGER (#eigenvalues above the threshold minus
// #below).  Eigenvalue magnitudes approach 1 from below, so an eigenvalue left with a sign error s moves the trace by s, in the
// direction of its sign; the products' rounding moves it by ~1e-15 p.  What is left unresolved sits within l0 |B| of the
// threshold and costs L at most l0 |B| s: tolerate s up to 1e-14 / l0 (1e-8 at the fine resolution 1e-6, 2e-10 for a first
// pass at 5e-5).  (Two unresolved eigenvalues on opposite sides of the threshold whose errors cancel to that accuracy would
// slip through this net -- and are left to the entrywise one.)
double rank_trace_tolerance(double l0, int p)
{
    // never below what the rounding of the products puts into the trace of a CONVERGED iterate (a few 1e-15 per diagonal entry)
    return std::max(std::min(1e-8, 1e-14 / std::max(l0, 1e-300)), 2e-14 * p);
}

// cnorm_h[k] >= |C_k|_2, mu_h[k] = mu1_k / rho.  Fills the coefficient table; returns steps.
int rank_ns_plan(const double* cnorm_h, const double* mu_h, int K, double l0, double* coef_h, NsPlan* plan, int degrees)
{
    // first step: cubic with the shift by mu folded into its two products; then the cheapest mix of cubic, quintic
    // and degree-nine steps for what is left of [l,1] (every step X^2, [t], X t: 2 / 3 / 4 products)
    const double lq = l0 < 0.5 ? l0 : 0.5;
    const NsStep s0 = ns_cubic_step(lq);
    const double a0 = (lq < 0.99) ? std::sqrt(3.0 / (1.0 + lq + lq * lq)) : 1.0;
    const bool more = 1.0 - s0.lnew >= 4e-16;
    const NsSeq* sq = more ? &ns_sign_schedule(s0.lnew, degrees) : nullptr;
    const int n = 1 + (sq ? sq->n : 0);
    if (n < 2 || n > NS_RANK_MAX_STEPS || (sq && sq->cost >= (1 << 29))) return -1;
    plan->steps = n;
    plan->stable = false;
    plan->deg[0] = 3;
    for (int i = 1; i < n; ++i) plan->deg[i] = sq->deg[i - 1];
    // Residual check of the result: rank_check() below.
    plan->check = rank_check(plan->deg[n - 1], l0);
    int g = 0;
    for (int k = 0; k < K; ++k) {
        const double mu = mu_h[k];
        const double nb = (cnorm_h[k] + mu) * (1.0 + 1e-10);      // |C - mu I| <= |C| + mu
        if (!(nb > 0.0) || !std::isfinite(nb)) return -1;
        auto put = [&](int gg, double cI, double cAcc, double cE, double dI, double dC) {
            double* o = coef_h + (size_t)gg * NS_SLOT(K) + (size_t)k * NS_NCOEF;
            o[0] = cI; o[1] = cAcc; o[2] = cE; o[3] = dI; o[4] = dC; o[5] = 0.0;
        };
        g = 0;
        const double s2 = a0 * a0 / (nb * nb);
        // T0 = 1.5 I - 0.5 a0^2 X0^2, X0 = (C - mu I)/nb   [product C*C, E = C]
        put(g++, 1.5 - 0.5 * s2 * mu * mu, -0.5 * s2, s2 * mu, 0.0, 0.0);
        // X1 = a0 X0 T0 = (a0/nb) C T0 - (a0 mu/nb) T0      [product C*T0, E = T0]
        put(g++, 0.0, a0 / nb, -a0 * mu / nb, 0.0, 0.0);
        for (int it = 1; it < n; ++it) {
            const double* t = sq->st[it - 1].t;
            const int d = sq->deg[it - 1];
            if (d == 3) {
                put(g++, t[0], t[1], 0.0, 0.0, 0.0);             // T = t0 I + t1 X^2
            } else if (d == 5) {
                put(g++, 0.0, 1.0, 0.0, 0.0, 0.0);               // M = X^2
                put(g++, t[0], t[2], t[1], 0.0, 0.0);            // T = t0 I + t2 M^2 + t1 M
            } else {
                const double a = t[3] / (2.0 * t[4]), dl = t[2] / t[4] - a * a, e = t[1] - t[4] * dl * a;
                put(g++, 0.0, 1.0, 0.0, 0.0, 0.0);               // M = X^2
                put(g++, 0.0, 1.0, a, dl, 1.0);                  // Q = M^2 + a M; second output Q + d I
                put(g++, t[0], t[4], e, 0.0, 0.0);               // T = f I + t4 Q (Q + d I) + e M
            }
            put(g++, 0.0, 1.0, 0.0, 1.0, 1.0);                   // X <- X T ; second output P2 = I + X (last step)
        }
        // L = (C - mu I) P2 / 2 = 0.5 C P2 - 0.5 mu P2        [product C*P2, E = P2]
        put(g++, 0.0, 0.5, -0.5 * mu, 0.0, 0.0);
    }
    plan->products = g;
    return 0;
}
// Where the sign iteration of rank_ns_plan(l0) leaves an eigenvalue that started at x (|x| <= 1, relative to the norm bound):
// the scalar image under the first cubic step and every step of the schedule.  For x below the plan's l0 this is how far the
// first pass got with an eigenvalue it was not planned for.
double rank_ns_image(double l0, int degrees, double x)
{
    const double lq = l0 < 0.5 ? l0 : 0.5;
    const NsStep s0 = ns_cubic_step(lq);
    const double a0 = (lq < 0.99) ? std::sqrt(3.0 / (1.0 + lq + lq * lq)) : 1.0;
    x = a0 * x * (1.5 - 0.5 * a0 * a0 * x * x);
    if (1.0 - s0.lnew < 4e-16) return x;
    const NsSeq& sq = ns_sign_schedule(s0.lnew, degrees);
    for (int i = 0; i < sq.n; ++i) {
        const double* t = sq.st[i].t;
        const double m = x * x;
        const int d = sq.deg[i];
        const double tm = (d == 3) ? t[0] + t[1] * m : (d == 5 ? t[0] + (t[1] + t[2] * m) * m
                                                                 : t[0] + (t[1] + (t[2] + (t[3] + t[4] * m) * m) * m) * m);
        x *= tm;
    }
    return x;
}

// Continuation plan for m instances whose iterate X already has its eigenvalues in [-1,1] and at least lp away from 0: the
// schedule for [lp,1] and the closing product.  plan->steps = schedule steps + 1 (rank_ns_steps runs steps 1 .. steps-1);
// coefficient slot g of instance i at coef_h + g * slot + i * NS_NCOEF.
int rank_ns_plan_continue(const double* mu_h, int m, double lp, double* coef_h, NsPlan* plan, int degrees, size_t slot)
{
    if (!(lp > 0.0) || !(lp < 1.0)) return -1;
    const NsSeq& sq = ns_sign_schedule(lp, degrees);
    const int n = 1 + sq.n;
    if (sq.n < 1 || n > NS_RANK_MAX_STEPS || sq.cost >= (1 << 29)) return -1;
    plan->steps = n;
    plan->stable = false;
    plan->deg[0] = 0;
    for (int i = 1; i < n; ++i) plan->deg[i] = sq.deg[i - 1];
    plan->check = rank_check(plan->deg[n - 1], 1e-6);   // what is left short here started within the FINE resolution of the threshold
    int g = 0;
    for (int k = 0; k < m; ++k) {
        auto put = [&](int gg, double cI, double cAcc, double cE, double dI, double dC) {
            double* o = coef_h + (size_t)gg * slot + (size_t)k * NS_NCOEF;
            o[0] = cI; o[1] = cAcc; o[2] = cE; o[3] = dI; o[4] = dC; o[5] = 0.0;
        };
        g = 0;
        for (int it = 1; it < n; ++it) {
            const double* t = sq.st[it - 1].t;
            const int d = sq.deg[it - 1];
            if (d == 3) {
                put(g++, t[0], t[1], 0.0, 0.0, 0.0);
            } else if (d == 5) {
                put(g++, 0.0, 1.0, 0.0, 0.0, 0.0);
                put(g++, t[0], t[2], t[1], 0.0, 0.0);
            } else {
                const double a = t[3] / (2.0 * t[4]), dl2 = t[2] / t[4] - a * a, e = t[1] - t[4] * dl2 * a;
                put(g++, 0.0, 1.0, 0.0, 0.0, 0.0);
                put(g++, 0.0, 1.0, a, dl2, 1.0);
                put(g++, t[0], t[4], e, 0.0, 0.0);
            }
            put(g++, 0.0, 1.0, 0.0, 1.0, 1.0);
        }
        put(g++, 0.0, 0.5, -0.5 * mu_h[k], 0.0, 0.0);
    }
    plan->products = g;
    return 0;
}

void rank_ns_run(hipStream_t st, const NsPlan& plan, const double* coef_d, const double* C, double* Xa, double* Xb,
                 double* Tb, double* P2, double* out, double* maxdev, int K, int p, int variant, size_t cslot, bool t0_ready)
{
    // maxdev must be zero on entry; cslot: doubles between the coefficient slots of successive launches
    // (0 = NS_SLOT(K); a sub-batch of a larger table passes the table's slot size).
    // scratch of the higher-degree steps: M = X^2 in `out` (free until the last launch), Q in P2 (free until the last
    // step's second output), Q + d I in the other X buffer (free until X T is written there)
    const size_t cs = cslot ? cslot : NS_SLOT(K);
    if (!t0_ready) launch_symm(st, C, C, Tb, nullptr, C, coef_d, K, p, variant);     // else: Tb holds T0 (launch_rank_t0)
    launch_symm(st, C, Tb, Xa, nullptr, Tb, coef_d + cs, K, p, variant);
    rank_ns_steps(st, plan, coef_d + 2 * cs, C, Xa, Xb, Tb, P2, out, maxdev, K, p, variant, cs);
}

// Steps 1 .. steps-1 of a plan from an iterate X (eigenvalues in [-1,1], those of interest at least the plan's l away from 0)
// and the closing product L = (C - mu I)(I + X_last)/2: the tail of rank_ns_run, and the whole of a CONTINUATION
// (rank_ns_plan_continue) -- more steps on the instances whose eigenvalues next to the threshold the first pass did not resolve.
void rank_ns_steps(hipStream_t st, const NsPlan& plan, const double* coef_d, const double* C, double* X, double* Xn, double* Tb,
                   double* P2, double* out, double* maxdev, int K, int p, int variant, size_t cs)
{
    int g = 0;
    const int n = plan.steps;
    for (int it = 1; it < n; ++it) {
        const bool last = (it == n - 1);
        double* md = last ? maxdev : nullptr;        // max |T_last - I| = |t(X^2) - 1|: the residual check
        if (plan.deg[it] == 5) {
            launch_symm(st, X, X, out, nullptr, nullptr, coef_d + cs * g++, K, p, variant);
            launch_symm(st, out, out, Tb, nullptr, out, coef_d + cs * g++, K, p, variant, md);
        } else if (plan.deg[it] == 9) {
            launch_symm(st, X, X, out, nullptr, nullptr, coef_d + cs * g++, K, p, variant);
            launch_symm(st, out, out, P2, Xn, out, coef_d + cs * g++, K, p, variant);
            launch_symm(st, P2, Xn, Tb, nullptr, out, coef_d + cs * g++, K, p, variant, md);
        } else {
            launch_symm(st, X, X, Tb, nullptr, nullptr, coef_d + cs * g++, K, p, variant, md);
        }
        launch_symm(st, X, Tb, Xn, last ? P2 : nullptr, nullptr, coef_d + cs * g++, K, p, variant);
        std::swap(X, Xn);
    }
    launch_symm(st, C, P2, out, nullptr, P2, coef_d + cs * g++, K, p, variant);
}

}  // namespace ggl
