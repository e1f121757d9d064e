// The Omega-step of a SMALL matrix as ONE launch: one workgroup per instance, every matrix of the Newton-Schulz chain resident in
// that CU's LDS, products on the FP64 matrix cores straight out of LDS, the spectral bound and the choice of the schedule on the
// device.  Replaces, for p <= 64, the launch-per-product chain (a dozen launches for 0.1 ms of Omega-step at p <= 128, all of it
// launch and tile-kernel latency) -- and is what the north_star's "one-block-per-matrix kernel for small p" became once the
// LDS Jacobi eigensolver had been measured (2.4 us per rotation step).
//
// Reference: phiplus, solver/ggl_helper.py:272-303 -- Omega = Q diag((d + sqrt(d^2 + 4 beta)) / 2) Q^T = (W + (W^2 + 4 beta I)^(1/2)) / 2,
// W = Theta - L - X - beta S (solver/admm_solver.py:180-187).
//
//   W (formed from the iterate: Theta, L, X, S are read once)  ->  A' = W W + 4 beta I  ->  B' = A' A'
//   c = sqrt(min(|B'|_inf, |B'|_F, Collatz-Wielandt ratio of the row sums)) >= lambda_max(A'),   l = sqrt(4 beta / c)
//   schedule = table[ceil(-ln l / ln 1.02)]: the planner's mixed-degree minimax schedule for the interval [1.02^-idx, 1]
//              (newton_schulz.hip: ns_schedule_query; built on the host once per stopping tolerance and degree set)
//   the coupled iteration Y <- Y t(ZY), Z <- t(ZY) Z with the same polynomials, affine epilogues and product count as the
//   launch chain, every instance on ITS OWN schedule
//   Omega = (W + sqrt(c) Y_last) / 2, bitwise symmetric (upper blocks computed, mirrored)
// An instance whose condition number exceeds the symmetric schedule's range (kappa > 300), whose schedule is longer than the
// table holds or whose data are not finite raises flag[0]: the host repeats the step on the launch chain.
#include "common.hpp"
#include "kernels.hpp"

namespace ggl {

typedef double v4d __attribute__((ext_vector_type(4)));

// LDS row stride (doubles) of a PT x PT matrix: ODD, so that the 16 lanes of a lane group that write a block's MIRROR image (one
// column, 16 consecutive rows) land on 16 different bank pairs (an even stride makes those writes 8-way conflicted; measured, the
// difference is within noise -- the epilogue's cost is its instruction count).  The fragment reads (two
// k-rows of 16 consecutive doubles per lane group) are 2-way conflicted with any stride that is not 16 mod 32; they stay far
// below the matrix pipe's time.
template <int PT> struct LdsDim {
    static constexpr int LD = PT + 1, NB = PT / 16;
    static constexpr int NE = (PT / 2) * (PT + 1);          // elements of the lower triangle incl. the diagonal
};

// element e of the lower triangle, rows folded in pairs (row r with row PT-1-r: PT+1 elements per pair, row-contiguous)
template <int PT>
__device__ __forceinline__ void tri_index(int e, int& i, int& j)
{
    const int pr = e / (PT + 1), r = e - pr * (PT + 1);
    if (r <= pr) { i = pr; j = r; }
    else { i = PT - 1 - pr; j = r - pr - 1; }
}

// out = cI I + cA (A B) + cE E for commuting symmetric A, B (and symmetric E) in LDS, all zero beyond p; only the upper 16 x 16
// blocks are computed (dealt over the NW waves: block w, w + NW, ...) and mirrored.  A wave works through its blocks one
// after the other and spreads the epilogue of block s (accumulator reads, affine combination, the stores of the block and of
// its mirror image) over the first k-steps of block s+1: a wave64 vector instruction takes 4 cycles and there are ~150 of
// them per block, which next to a 1.6 us matrix phase cost 1.1 us when they ran after it; under the matrix instructions of
// the next block (64 cycles each, 56 of them free for the vector unit) only the last block's epilogue is exposed.
// NW = 8 (round 5): TWO waves per SIMD -- ten blocks at PT = 64 are 2/2/1/1/1/1/1/1, i.e. the same three blocks on the busiest
// SIMD as with four waves (3/3/2/2), but what a lone wave exposes (the last block's epilogue, the first fragments' LDS
// latency, the barrier) now runs under the other wave's matrix instructions; a wave without a block in a slot issues nothing
// there (with four waves it multiplied a dummy block, which two waves per SIMD would pay for).
// ALIAS: out is one of the operands -- every value is kept in registers until all waves have finished reading.
template <int PT, bool ALIAS, int NW>
__device__ __forceinline__ void lds_symm(const double* A, const double* B, double* out, double cI, double cA, const double* E,
                                         double cE, int p, long long* ts = nullptr)
{
    if (ts && threadIdx.x == 0) ts[0] = (long long)wall_clock64();
    constexpr int LD = LdsDim<PT>::LD, NB = LdsDim<PT>::NB, NPAIR = NB * (NB + 1) / 2;
    constexpr int MAXB = (NPAIR + NW - 1) / NW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    constexpr int KS = PT / 4, NT = MAXB * KS, D = 4, RING = 8;      // fragments are read D k-steps ahead: the LDS latency is
                                                                      // ~2 matrix instructions, and a wave has one accumulator chain at a time
    v4d acc = {0.0, 0.0, 0.0, 0.0}, pacc = {0.0, 0.0, 0.0, 0.0};
    int pij = 0, pji = 0, prd = -1, prmax = -1;          // finished block: element offsets (block, mirror), diagonal register, last register to store
    double keep[ALIAS ? MAXB : 1][4];
    int kij[ALIAS ? MAXB : 1], kji[ALIAS ? MAXB : 1], krmax[ALIAS ? MAXB : 1];
    const double* ap[MAXB];
    const double* bp[MAXB];
    int Ib[MAXB], Jb[MAXB];
#pragma unroll
    for (int s = 0; s < MAXB; ++s) {
        int pr = wave + NW * s, I = -1, J = 0;
        if (pr < NPAIR) {
            I = 0;
            while (pr >= NB - I) { pr -= NB - I; ++I; }
            J = I + pr;
        }
        Ib[s] = I; Jb[s] = J;
        ap[s] = A + fk * LD + 16 * max(I, 0) + fr;       // a(i, k) = A[k][i] (symmetric): lanes of a k-row read contiguously
        bp[s] = B + fk * LD + 16 * J + fr;
        if (ALIAS) krmax[s] = -1;
    }
    // register r of the finished block (C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg)
    auto epilogue = [&](int sdone, int r) {
        if (r <= prmax) {
            double x = cA * pacc[r];
            if (E) x += cE * E[pij + 4 * r * LD];
            if (r == prd) x += cI;
            if (ALIAS) keep[sdone][r] = x;
            else { out[pij + 4 * r * LD] = x; out[pji + 4 * r] = x; }
        }
    };
    double af[RING], bf[RING];
    // (a wave's idle slots are its LAST ones: Ib[s] < 0 implies Ib[s+1] < 0; the conditions below are wave-uniform)
    // (Skipping the k-steps entirely beyond p -- zeros times zeros; p = 50 in a 64-tile would run 13 of 16 -- was measured: a
    // guard per step on the loads and the matrix instruction costs every size ~30 %, K = 256, p = 64: 40.1 -> 51.6 us, for the
    // 10 % it saves at p = 50.  Not kept: the steps stay unconditional inside a block.)
#pragma unroll
    for (int t = 0; t < D && t < NT; ++t)
        if (Ib[t / KS] >= 0) { af[t] = ap[t / KS][(t % KS) * 4 * LD]; bf[t] = bp[t / KS][(t % KS) * 4 * LD]; }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int s = t / KS, kq = t % KS;
        if (Ib[s] >= 0) {
            if (t + D < NT && Ib[(t + D) / KS] >= 0) {
                af[(t + D) % RING] = ap[(t + D) / KS][((t + D) % KS) * 4 * LD];
                bf[(t + D) % RING] = bp[(t + D) / KS][((t + D) % KS) * 4 * LD];
            }
            if (kq == 0) acc = (v4d){0.0, 0.0, 0.0, 0.0};
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[t % RING], bf[t % RING], acc, 0, 0, 0);
            if (s > 0 && kq < 4) epilogue(s - 1, kq);
            if (kq == KS - 1) {
                const int I = Ib[s], J = Jb[s];
                pacc = acc;
                pij = (16 * I + fk) * LD + 16 * J + fr;
                pji = (16 * J + fr) * LD + 16 * I + fk;
                // diagonal blocks: the registers with row <= col are stored (and mirrored), one of them holds the diagonal element
                const int dr = fr - fk;
                prmax = (I != J) ? 3 : (dr >= 0 ? (dr >> 2) : -1);
                prd = (I == J && dr >= 0 && (dr & 3) == 0 && 16 * I + fr < p) ? (dr >> 2) : -1;
                if (ALIAS) { kij[s] = pij; kji[s] = pji; krmax[s] = prmax; }
            }
        }
    }
    // the last block this wave computed (slot MAXB - 1, or an earlier one when its last slots are idle)
    {
        int slast = -1;
#pragma unroll
        for (int s = 0; s < MAXB; ++s) if (Ib[s] >= 0) slast = s;
#pragma unroll
        for (int s = 0; s < MAXB; ++s)
            if (s == slast) {
#pragma unroll
                for (int r = 0; r < 4; ++r) epilogue(s, r);
            }
    }
    if (ALIAS) {
        __syncthreads();
#pragma unroll
        for (int s = 0; s < MAXB; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r <= krmax[s]) { out[kij[s] + 4 * r * LD] = keep[s][r]; out[kji[s] + 4 * r] = keep[s][r]; }
    }
    __syncthreads();
    if (ts && threadIdx.x == 0) ts[4] = (long long)wall_clock64();
}

// thread 0 of a workgroup of the SGL form: this instance's row (or its failure mark) is written -- make it visible to the host,
// count the workgroup in, and as the last one publish the sequence number (cf. k_reduce_partials)
__device__ __forceinline__ void lds_sgl_arrive(const LdsSgl& sg)
{
    __threadfence_system();
    const unsigned prev = atomicAdd(sg.arrive, 1u);
    if (prev == gridDim.x - 1) {
        *sg.arrive = 0u;
        __threadfence_system();
        if (sg.seq) *(volatile unsigned long long*)sg.seq = sg.seq_val;
    }
}

// table entry: { n, deg[OMEGA_LDS_MAXSTEP], (pad to 8), coef[OMEGA_LDS_MAXSTEP][6] = {t0..t4, l_after} } = OMEGA_LDS_ENT doubles
// SGL (round 5): K INDEPENDENT single problems (single_admm_solver.py:157-214) -- the workgroup that holds an instance's Omega
// in LDS goes straight on with that instance's Theta-step, dual update and the five sums of the stopping test
// (prox_od_1norm, solver/ggl_helper.py:16-27; single_admm_solver.py:169, :178, :277-291; k_theta_sgl's arithmetic), writes the
// row of sums to pinned memory and, as the last workgroup to arrive, the sequence number the host polls: a batch iteration is
// this ONE kernel behind its parameter copy instead of four dependent launches.  An instance outside the kernel's range marks
// itself (sg.fail) and leaves its iterate alone -- the caller redoes exactly those instances on the launch chain.
template <int PT, int NW, bool SGL>
__global__ __launch_bounds__(64 * NW) void k_omega_lds(const double* __restrict__ Theta, const double* __restrict__ Lm,
                                                   const double* __restrict__ X, const double* __restrict__ S,
                                                   const double* __restrict__ betaK, double* __restrict__ Omega,
                                                   const double* __restrict__ table, int ntab,
                                                   double lnq, int p, int* __restrict__ flag, int* __restrict__ flag_host,
                                                   int flag_slot, unsigned long long* __restrict__ units,
                                                   double* __restrict__ cbound, long long* __restrict__ dbg, LdsSgl sg)
{
    constexpr int LD = LdsDim<PT>::LD, NE = LdsDim<PT>::NE, NTH = 64 * NW, NIT = (NE + NTH - 1) / NTH;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* b0 = lds;
    double* b1 = b0 + PT * LD;
    double* b2 = b1 + PT * LD;
    double* b3 = b2 + PT * LD;
    double* vec = b3 + PT * LD;              // [PT] row sums
    double* sh = vec + PT;                   // [8]
    double* part = sh + 8;                   // [2][NTH] partial column sums
    const int k = blockIdx.x, tid = threadIdx.x;
#define GGL_TS(i) do { if (dbg && k == 0 && tid == 0) dbg[i] = (long long)wall_clock64(); } while (0)
    GGL_TS(0);
    const size_t off = (size_t)k * p * p;
    const double beta = betaK[k];
    // W = ((Theta - L) - X) - beta S from the LOWER triangle, mirrored (what numpy.linalg.eigh reads; k_form_W_sym's arithmetic).
    // Every thread owns NIT elements of the triangle: all their loads are issued before the first is used (row-contiguous, no
    // upper-triangle traffic), and the values stay in registers until the final Omega = (W + sqrt(c) Y) / 2.
    // (SGL form: Theta and X are also WRITTEN by this kernel, through sg -- they are read through sg as well, never through
    // the __restrict__ parameters)
    const double* Thr = SGL ? sg.Theta : Theta;
    const double* Xr = SGL ? sg.X : X;
    double wreg[NIT];
    {
        double th[NIT], xx[NIT], ss[NIT], ll[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int i, j;
            tri_index<PT>(tid + NTH * it, i, j);
            const bool in = (tid + NTH * it) < NE && i < p;
            const size_t o = off + (size_t)i * p + j;
            th[it] = in ? Thr[o] : 0.0;
            ll[it] = (in && Lm) ? Lm[o] : 0.0;
            xx[it] = in ? Xr[o] : 0.0;
            ss[it] = in ? S[o] : 0.0;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int i, j;
            tri_index<PT>(tid + NTH * it, i, j);
            double t = th[it];
            if (Lm) t -= ll[it];
            wreg[it] = (t - xx[it]) - beta * ss[it];
            if (tid + NTH * it < NE) { b0[i * LD + j] = wreg[it]; b0[j * LD + i] = wreg[it]; }
        }
    }
    __syncthreads();
    GGL_TS(1);
    // A' = W W + 4 beta I -> b1;  B' = A' A' -> b2
    lds_symm<PT, false, NW>(b0, b0, b1, 4.0 * beta, 1.0, nullptr, 0.0, p, (dbg && k == 0) ? dbg + 10 : nullptr);
    GGL_TS(2);
    lds_symm<PT, false, NW>(b1, b1, b2, 0.0, 1.0, nullptr, 0.0, p);
    GGL_TS(3);
    // the bound: row sums d, |B'|_inf, |B'|_F^2, Collatz-Wielandt ratio max_i (|B'| d)_i / d_i.  B' is symmetric: a thread walks
    // down a COLUMN (lanes on consecutive columns: conflict-free), NQ threads per column; the PT column totals are finished by
    // the first PT threads, which all sit in wave 0 (PT <= 64), so each block-wide reduction is one shuffle chain.
    double fro, inf;
    {
        constexpr int NQ = 256 / PT;          // (not NTH / PT: the order of these sums must not depend on the number of waves)
        const int col = tid % PT, q = tid / PT;
        double s1 = 0.0, s2 = 0.0;
        if (q < NQ)
            for (int j = q; j < PT; j += NQ) { const double x = b2[j * LD + col]; s1 += fabs(x); s2 += x * x; }
        part[tid] = s1;
        part[NTH + tid] = s2;
        __syncthreads();
        if (tid < 64) {
            double d = 0.0, f = 0.0;
            if (tid < PT)
                for (int qq = 0; qq < NQ; ++qq) { d += part[qq * PT + tid]; f += part[NTH + qq * PT + tid]; }
            if (tid < PT) vec[tid] = d;
            const double dm = wave_max(d), fs = wave_sum(f);
            if (tid == 0) { sh[0] = dm; sh[1] = fs; }
        }
        __syncthreads();
        double y = 0.0;
        if (q < NQ)
            for (int j = q; j < PT; j += NQ) y += fabs(b2[j * LD + col]) * vec[j];
        part[tid] = y;
        __syncthreads();
        if (tid < 64) {
            double r = 0.0;
            if (tid < p) {
                double yy = 0.0;
                for (int qq = 0; qq < NQ; ++qq) yy += part[qq * PT + tid];
                r = yy / vec[tid];
                if (!(r == r)) r = 0.0;                   // (a zero row: 0/0)
            }
            r = wave_max(r);
            if (tid == 0) sh[2] = r;
        }
        __syncthreads();
        inf = sh[0];
        fro = sh[1];
        const double cw = sh[2];
        if (isfinite(cw) && cw > 0.0) inf = fmin(inf, cw * (1.0 + 1e-12));
    }
    double c = sqrt(fmin(inf, sqrt(fro))) * (1.0 + 1e-10);
    if (c < 4.0 * beta) c = 4.0 * beta;                   // lambda_min(A') = 4 beta is exact
    const double kappa = c / (4.0 * beta);
    int idx = (int)ceil(log(sqrt(kappa)) / lnq - 1e-12);  // l = kappa^-1/2 >= q^-idx
    if (idx < 0) idx = 0;
    // (fmax / fmin drop NaNs: the sum of squares is what a non-finite entry cannot hide from)
    const bool bad = !(isfinite(fro) && isfinite(c) && c > 0.0 && beta > 0.0) || kappa > NS_SYM_KAPPA_MAX || idx >= ntab;
    const double* ent = table + (size_t)min(idx, ntab - 1) * OMEGA_LDS_ENT;
    const int n = bad ? 0 : (int)ent[0];
    if (n < 1 || n > OMEGA_LDS_MAXSTEP) {
        if (tid == 0) {
            atomicOr(flag + flag_slot, 1);
            flag_host[flag_slot] = 1;
            if (SGL) { sg.fail[k] = 1; lds_sgl_arrive(sg); }
        }
        return;
    }
    if (tid == 0 && cbound) cbound[k] = c;
    GGL_TS(4);
    const double sc = sqrt(c);
    double *Ap = b1, *Bp = b2, *f0 = b0, *f1 = b3;        // A', B', two free buffers
    double *Y, *Z;
    unsigned nprod = 2;
    {
        const double* t = ent + 8;
        const int d0 = (int)ent[1];
        // T1 into B''s buffer
        if (d0 == 9) {
            // U = t2 I + (t3/c) A' + (t4/c^2) B' -> f0 (elementwise), T1 = t0 I + (t1/c) A' + (U B') / c^2 -> f1
            for (int e = tid; e < PT * PT; e += NTH) {
                const int i = e / PT, j = e - i * PT, o = i * LD + j;
                double u = (t[3] / c) * Ap[o] + (t[4] / (c * c)) * Bp[o];
                if (i == j && i < p) u += t[2];
                f0[o] = u;
            }
            __syncthreads();
            lds_symm<PT, false, NW>(f0, Bp, f1, t[0], 1.0 / (c * c), Ap, t[1] / c, p);
            nprod += 1;
            Z = f1;                                       // T1
            f1 = Bp;                                      // B' is dead
        } else {
            const double c2 = (d0 == 5) ? t[2] / (c * c) : 0.0;
            for (int e = tid; e < PT * PT; e += NTH) {
                const int i = e / PT, j = e - i * PT, o = i * LD + j;
                double x = (t[1] / c) * Ap[o] + c2 * Bp[o];
                if (i == j && i < p) x += t[0];
                Bp[o] = x;
            }
            __syncthreads();
            Z = Bp;                                       // T1 in place of B'
        }
        // Y1 = (A'/c) T1 -> f0
        lds_symm<PT, false, NW>(Ap, Z, f0, 0.0, 1.0 / c, nullptr, 0.0, p);
        nprod += 1;
        Y = f0;
        f0 = Ap;                                          // A' is dead: free buffers f0, f1 (deg 9) or f0 and b3
        if (d0 != 9) f1 = b3;
    }
    GGL_TS(5);
    for (int it = 1; it < n; ++it) {
        const double* t = ent + 8 + 6 * it;
        const int d = (int)ent[1 + it];
        const bool last = (it == n - 1);
        double* T;
        if (d == 3) {
            lds_symm<PT, false, NW>(Z, Y, f0, t[0], t[1], nullptr, 0.0, p);          // T = t0 I + t1 (Z Y)
            nprod += 1;
            T = f0;
        } else if (d == 5) {
            lds_symm<PT, false, NW>(Z, Y, f0, 0.0, 1.0, nullptr, 0.0, p);            // M = Z Y
            lds_symm<PT, false, NW>(f0, f0, f1, t[0], t[2], f0, t[1], p);            // T = t0 I + t2 M M + t1 M
            nprod += 2;
            T = f1;
            double* sw = f0; f0 = f1; f1 = sw;            // T lives in (new) f0; M's buffer is free as f1
        } else {
            const double a = t[3] / (2.0 * t[4]), dl = t[2] / t[4] - a * a, e = t[1] - t[4] * dl * a;
            lds_symm<PT, false, NW>(Z, Y, f0, 0.0, 1.0, nullptr, 0.0, p);            // M = Z Y -> f0
            lds_symm<PT, false, NW>(f0, f0, f1, 0.0, 1.0, f0, a, p);                 // Q = M M + a M -> f1
            // T = t0 I + t4 (Q Q + dl Q) + e M -> in place of Q (reads finish before the writes)
            {
                // out = f I + t4 (Q Q) + [t4 dl Q + e M]: two E terms -- fold the second into a pre-pass on M's buffer
                for (int q = tid; q < PT * PT; q += NTH) {
                    const int i = q / PT, j = q - i * PT, o = i * LD + j;
                    f0[o] = e * f0[o] + t[4] * dl * f1[o];
                }
                __syncthreads();
                lds_symm<PT, true, NW>(f1, f1, f1, t[0], t[4], f0, 1.0, p);
            }
            nprod += 3;
            T = f1;
            double* sw = f0; f0 = f1; f1 = sw;            // T in f0, the other buffer free as f1
        }
        if (last) {
            lds_symm<PT, false, NW>(Y, T, f1, 0.0, 1.0, nullptr, 0.0, p);            // Y_last = Y T -> f1
            nprod += 1;
            Y = f1;
        } else {
            // Ynew = Y T -> f1;  Znew = T Z -> old Y's buffer
            lds_symm<PT, false, NW>(Y, T, f1, 0.0, 1.0, nullptr, 0.0, p);
            lds_symm<PT, false, NW>(T, Z, Y, 0.0, 1.0, nullptr, 0.0, p);
            nprod += 2;
            double* oldY = Y;
            double* oldZ = Z;
            Y = f1;
            Z = oldY;
            f1 = oldZ;                                    // free: T's buffer (f0) and old Z's
        }
    }
    // Omega = (W + sqrt(c) Y) / 2 by the owners of W's elements, both mirror images into a buffer the chain is done with
    // (bitwise symmetric), then row-contiguous stores
    GGL_TS(6);
    double* wb = (Y == b0) ? b1 : b0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        int i, j;
        tri_index<PT>(tid + NTH * it, i, j);
        if (tid + NTH * it < NE) {
            const double om = 0.5 * wreg[it] + (0.5 * sc) * Y[i * LD + j];
            wb[i * LD + j] = om;
            wb[j * LD + i] = om;
        }
    }
    __syncthreads();
    GGL_TS(7);
    if (!SGL) {
        for (int e = tid; e < PT * PT; e += NTH) {
            const int i = e / PT, j = e - i * PT;
            if (i < p && j < p) Omega[off + (size_t)i * p + j] = wb[i * LD + j];
        }
    } else {
        // Theta = prox_od_1norm(Omega + X, lambda1 / rho), X += Omega - Theta, the five sums over the instance's own block
        // (in place: every element is read and written by its own thread only, and the reads for W are long done)
        double* ThW = sg.Theta;
        double* XW = sg.X;
        const int pin = sg.pk ? sg.pk[k] : p;
        const double* mask = sg.mask ? sg.mask + (size_t)k * sg.mask_stride : nullptr;
        const double lk = mask ? 0.0 : sg.l1K[k];
        const double inv_rho = mask ? sg.invrhoK[k] : 0.0;
        double acc[GGL_NNORM] = {0, 0, 0, 0, 0};
        for (int e = tid; e < PT * PT; e += NTH) {
            const int i = e / PT, j = e - i * PT;
            if (i < p && j < p) {
                const size_t o = (size_t)i * p + j;
                const double om = wb[i * LD + j];
                const double x = XW[off + o];
                const double v = om + x;                                   // (k_theta_sgl: (om + l) + x with l = 0)
                const double thr = mask ? inv_rho * mask[o] : lk;
                const double th = (i == j) ? v : soft(v, thr);
                const double xn = (x + om) - th;                           // single_admm_solver.py:178
                const double dp = om - sg.OmegaPrev[off + o];
                Omega[off + o] = om;
                ThW[off + o] = th;
                XW[off + o] = xn;
                if (i < pin && j < pin) {
                    acc[0] += om * om;
                    acc[1] += th * th;
                    acc[2] += xn * xn;
                    acc[3] += (om - th) * (om - th);
                    acc[4] += dp * dp;
                }
            }
        }
        __syncthreads();                                 // (part[] is free again: the bound is long done)
        block_sum<GGL_NNORM>(acc, part);
        if (tid == 0) {
#pragma unroll
            for (int v = 0; v < GGL_NNORM; ++v) sg.norms[(size_t)k * GGL_NNORM + v] = acc[v];
            lds_sgl_arrive(sg);
        }
    }
    GGL_TS(8);
    if (dbg && k == 0 && tid == 0) dbg[9] = nprod;
    if (tid == 0 && units) { atomicAdd(units, (unsigned long long)nprod); atomicAdd(units + 1, (unsigned long long)n); }
#undef GGL_TS
}

int omega_lds_max_p() { return 64; }

// table for the stopping tolerance `tol`: entry idx serves every l >= q^-idx, q = 1.02; returns the number of entries written
// (<= max_entries; the table ends where the symmetric schedule's range does, kappa = 300; an entry whose schedule needs more
// than OMEGA_LDS_MAXSTEP steps has n = 0: the kernel raises its flag there)
int omega_lds_build_table(double tol, int degrees, double* table_h, int max_entries, double* lnq_out)
{
    const double lnq = std::log(1.02);
    *lnq_out = lnq;
    int n = 0;
    for (; n < max_entries; ++n) {
        const double l = std::exp(-n * lnq);
        if (1.0 / (l * l) > NS_SYM_KAPPA_MAX * 1.05) break;
        int deg[8];
        double co[8 * 6];
        int units = 0;
        const int steps = ns_schedule_query(l, degrees, 8, deg, co, &units, tol);
        double* ent = table_h + (size_t)n * OMEGA_LDS_ENT;
        for (int i = 0; i < OMEGA_LDS_ENT; ++i) ent[i] = 0.0;
        if (steps < 1 || steps > OMEGA_LDS_MAXSTEP) { ent[0] = 0.0; continue; }        // (n = 0: the kernel raises the flag for this entry)
        ent[0] = steps;
        for (int i = 0; i < steps; ++i) {
            ent[1 + i] = deg[i];
            for (int j = 0; j < 6; ++j) ent[8 + 6 * i + j] = co[6 * i + j];
        }
    }
    return n;
}

bool launch_omega_lds(hipStream_t st, const double* Theta, const double* L, const double* X, const double* S, const double* betaK,
                      double* Omega, const double* table, int ntab, double lnq, int K, int p, int* flag,
                      int* flag_host, int flag_slot, unsigned long long* units, double* cbound, long long* dbg, int waves,
                      const LdsSgl* sgl)
{
    const LdsSgl sg = sgl ? *sgl : LdsSgl();
    // waves: 4 or 8 per workgroup, 0 = by size (two waves per SIMD from PT = 48 on, where a wave has more than one block)
#define GGL_OL(PT, NW)                                                                                                          \
    do {                                                                                                                        \
        const size_t lds = ((size_t)4 * PT * LdsDim<PT>::LD + PT + 8 + 2 * 64 * NW) * sizeof(double);                               \
        /* per launch: the attribute belongs to the current device's copy of the kernel */                            \
        if (sgl) {                                                                                                              \
            if (hipFuncSetAttribute((const void*)k_omega_lds<PT, NW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
                return false;                                                                                                    \
            hipLaunchKernelGGL((k_omega_lds<PT, NW, true>), dim3(K), dim3(64 * NW), lds, st, Theta, L, X, S, betaK, Omega, table, ntab, lnq, p, \
                               flag, flag_host, flag_slot, units, cbound, dbg, sg);                                                  \
        } else {                                                                                                                 \
            if (hipFuncSetAttribute((const void*)k_omega_lds<PT, NW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
                return false;                                                                                                    \
            hipLaunchKernelGGL((k_omega_lds<PT, NW, false>), dim3(K), dim3(64 * NW), lds, st, Theta, L, X, S, betaK, Omega, table, ntab, lnq, p, \
                               flag, flag_host, flag_slot, units, cbound, dbg, sg);                                                  \
        }                                                                                                                        \
    } while (0)
    const bool w8 = waves == 8 || (waves == 0 && p > 32);
    if (p <= 16) GGL_OL(16, 4);
    else if (p <= 32) GGL_OL(32, 4);
    else if (p <= 48) { if (w8) GGL_OL(48, 8); else GGL_OL(48, 4); }
    else if (p <= 64) { if (w8) GGL_OL(64, 8); else GGL_OL(64, 4); }
    else return false;
#undef GGL_OL
    return true;
}

}  // namespace ggl
