// Symmetric products on the INT8 matrix cores (error-free split, after Ozaki et al.): C = A B for commuting symmetric A, B
// whose entries are bounded by a known power of two (every operand of the scaled Newton-Schulz iteration has |.|_2 <= 1 up
// to a known factor, so one scale per matrix is enough).
//
//   slices   A / scale_A = sum_t D_t 2^-(6 + 7 t) + r,  D_t int8 with |digit| <= 64, |r| <= 2^-(7 S)   (k_slice_i8)
//   product  C = scale_A scale_B sum_{d <= DMAX} 2^-(12 + 7 d) sum_{t + u = d} D^A_t (D^B_u)^T           (k_symm_i8)
//            v_mfma_i32_16x16x64_i8 accumulates in int32 EXACTLY: one k = 64 step adds at most 64 * 64 * 64 = 2^18 per slice
//            pair; the pairs of equal weight d share one accumulator: (d + 1) pairs * p_pad / 64 steps, p_pad <= 4096 keeps
//            |sum| <= 8 * 64 * 2^18 = 2^27 < 2^31.
//   The MI355X's int8 matrix rate is >= 3944 TOPS against 78.6 TF/s for FP64 MFMA (MI355X_MICROARCH.md): S = 7 slices with
//   the triangular truncation are 28 slice products for one fp64-accurate product (tools/proto_ozaki.py prices the Omega-step).
//
// Layout: slices are int8 stacks [S][K][P][P], P = p rounded up to 64, zero padded, row-major -- both operands are read
// along rows (B symmetric: B[k][j] = B[j][k]), 64-byte row segments per k = 64 step, DMA'd straight into LDS
// (global_load_lds, 16 bytes per lane: one wave instruction = 16 rows x 64 bytes).  The 16-byte chunk c of row r is stored at
// chunk position c ^ ((r >> 2) & 3) -- applied on the SOURCE address, undone in the fragment read -- so that the sixteen
// lanes of a fragment read (rows r .. r+15, same chunk) cover all 64 banks.
// Tile 64 x 64, 4 waves of 32 x 32 (2 x 2 MFMA blocks), k-step 64, two LDS stages of 2 * S * 4 KiB (112 KiB at S = 7: one
// workgroup per CU; every wave holds 2 S A-fragments + 2 S B-fragments and 4 * (DMAX + 1) accumulators in registers).
// Measured and rejected: built into libggl_hip_dev.so only (-DGGL_DEV), like the other rejected variants.
#include "common.hpp"
#include "kernels.hpp"

#ifdef GGL_DEV
namespace ggl {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr8_t;
typedef __attribute__((address_space(3))) void* lptr8_t;

// ---------------------------------------------------------------------------------------------
// fp64 stack (K,p,p) -> S int8 slice stacks [S][K][P][P].  flag[0] is raised when an entry exceeds the scale.
// ---------------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(256) void k_slice_i8(const double* __restrict__ A, const double* __restrict__ scaleK,
                                                  int8_t* __restrict__ out, size_t sstride, int p, int P,
                                                  int* __restrict__ flag)
{
    // one thread = 4 consecutive columns of one row (a 4-byte store per slice)
    const int k = blockIdx.z, row = blockIdx.y;
    const int c0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (c0 >= P) return;
    const double inv = 1.0 / scaleK[k];
    double r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = c0 + e;
        r[e] = (row < p && c < p) ? A[((size_t)k * p + row) * p + c] * inv : 0.0;
    }
    bool over = false;
    int8_t* o = out + ((size_t)k * P + row) * P + c0;
#pragma unroll
    for (int t = 0; t < S; ++t) {
        const double w = __builtin_ldexp(1.0, 6 + 7 * t), wi = __builtin_ldexp(1.0, -(6 + 7 * t));
        int packed = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            double d = rint(r[e] * w);
            if (fabs(d) > 127.0) { over = true; d = d > 0 ? 127.0 : -127.0; }
            r[e] -= d * wi;
            packed |= ((int)d & 0xff) << (8 * e);
        }
        *reinterpret_cast<int*>(o + t * sstride) = packed;
    }
    if (over && flag) atomicOr(flag, 1);
}

void launch_slice_i8(hipStream_t st, const double* A, const double* scaleK, int8_t* out, int K, int p, int S, int* flag,
                     size_t sstride)
{
    const int P = (p + 63) / 64 * 64;
    if (!sstride) sstride = (size_t)K * P * P;
    dim3 grid((P / 4 + 255) / 256, P, K), blk(256);
    switch (S) {
        case 2: hipLaunchKernelGGL(k_slice_i8<2>, grid, blk, 0, st, A, scaleK, out, sstride, p, P, flag); break;
        case 3: hipLaunchKernelGGL(k_slice_i8<3>, grid, blk, 0, st, A, scaleK, out, sstride, p, P, flag); break;
        case 4: hipLaunchKernelGGL(k_slice_i8<4>, grid, blk, 0, st, A, scaleK, out, sstride, p, P, flag); break;
        case 5: hipLaunchKernelGGL(k_slice_i8<5>, grid, blk, 0, st, A, scaleK, out, sstride, p, P, flag); break;
        case 6: hipLaunchKernelGGL(k_slice_i8<6>, grid, blk, 0, st, A, scaleK, out, sstride, p, P, flag); break;
        case 7: hipLaunchKernelGGL(k_slice_i8<7>, grid, blk, 0, st, A, scaleK, out, sstride, p, P, flag); break;
        default: hipLaunchKernelGGL(k_slice_i8<8>, grid, blk, 0, st, A, scaleK, out, sstride, p, P, flag); break;
    }
}

// ---------------------------------------------------------------------------------------------
// the product
//   acc  = scale_A scale_B sum_{t + u <= DMAX} 2^-(12 + 7 (t + u)) D^A_t (D^B_u)^T
//   out1 = cI I + cAcc acc + cE1 E1 + cE2 E2        out2 = dI I + dAcc acc + dE1 E1 + dE2 E2
// each output optionally as an fp64 stack (C1 / C2) and / or as int8 slices of out / scale (S1 / S2, nS slices) -- the next
// product's operand, written by the producer: tile and mirror, so that every consumer reads rows.
// par[k][12] = { cI, cAcc, cE1, cE2, dI, dAcc, dE1, dE2, scale_A scale_B, 1 / scale_1, 1 / scale_2, - }
// ---------------------------------------------------------------------------------------------
static constexpr int I8_TLD = 65;    // row stride (doubles) of the staged output tile

template <int SA, int SB, int DMAX, int NSTG>
__global__ __launch_bounds__(256) void k_symm_i8(const I8Op op)
{
    constexpr int ND = DMAX + 1;
    constexpr int SLICE = 64 * 64;                          // bytes of one slice's 64 x 64 tile
    constexpr int STAGE = (SA + SB) * SLICE;
    extern __shared__ __attribute__((aligned(16))) int8_t smem8[];
    const int K = op.K, p = op.p, P = op.P;
    const int T = P / 64;
    // tile pair (I <= J) and instance: instances fastest over the XCDs (blockIdx % 8 = XCD)
    int b = blockIdx.x;
    const int kk = b % K;
    b /= K;
    int I = 0;
    while (b >= T - I) { b -= T - I; ++I; }
    const int J = I + b;
    const int I0 = I * 64, J0 = J * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
    const int8_t* Ak = op.As + (size_t)kk * P * P;
    const int8_t* Bk = op.Bs + (size_t)kk * P * P;

    v4i acc[2][2][ND];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int d = 0; d < ND; ++d) acc[i][j][d] = (v4i){0, 0, 0, 0};

    // DMA geometry: per slice one instruction per 16 rows (1 KiB); wave w moves rows 16 w .. 16 w + 15 of every slice;
    // lane l -> row 16 w + (l >> 2), stored chunk l & 3 = source chunk (l & 3) ^ ((l >> 4) & 3).
    const int drow = lane >> 2;
    const int dchunk = (lane & 3) ^ ((lane >> 4) & 3);
    const int8_t* srcA = Ak + (size_t)(I0 + 16 * wave + drow) * P + 16 * dchunk;
    const int8_t* srcB = Bk + (size_t)(J0 + 16 * wave + drow) * P + 16 * dchunk;
    auto issue = [&](int s, int buf) {
        int8_t* base = smem8 + buf * STAGE + wave * 1024;
#pragma unroll
        for (int t = 0; t < SA; ++t)
            __builtin_amdgcn_global_load_lds((gptr8_t)(srcA + t * op.sstrideA + 64 * s), (lptr8_t)(base + t * SLICE), 16, 0, 0);
#pragma unroll
        for (int u = 0; u < SB; ++u)
            __builtin_amdgcn_global_load_lds((gptr8_t)(srcB + u * op.sstrideB + 64 * s), (lptr8_t)(base + (SA + u) * SLICE), 16, 0, 0);
    };

    // fragment addresses: row r = base row + (lane & 15), logical chunk g = lane >> 4 -> stored chunk g ^ ((r >> 2) & 3);
    // (r >> 2) & 3 = ((lane & 15) >> 2) & 3 because the base rows are multiples of 16
    const int frow = lane & 15;
    const int fch = (lane >> 4) ^ ((frow >> 2) & 3);
    const int foff = frow * 64 + fch * 16;

    const int NS = P / 64;
    issue(0, 0);
    for (int s = 0; s < NS; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (NSTG == 2 && s + 1 < NS) issue(s + 1, (s + 1) & 1);
        const int8_t* st = smem8 + (NSTG == 2 ? (s & 1) : 0) * STAGE;
        v4i af[2][SA], bf[2][SB];
#pragma unroll
        for (int t = 0; t < SA; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                af[i][t] = *reinterpret_cast<const v4i*>(st + t * SLICE + (wr + 16 * i) * 64 + foff);
#pragma unroll
        for (int u = 0; u < SB; ++u)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                bf[j][u] = *reinterpret_cast<const v4i*>(st + (SA + u) * SLICE + (wc + 16 * j) * 64 + foff);
#pragma unroll
        for (int t = 0; t < SA; ++t)
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                if (t + u > DMAX) continue;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j][t + u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[i][t], bf[j][u], acc[i][j][t + u], 0, 0, 0);
            }
        if (NSTG == 1 && s + 1 < NS) {
            // single stage (two workgroups per CU cover each other's loads): the fragments are in registers, the stage is
            // free as soon as every wave has read it
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            issue(s + 1, 0);
        }
    }

    // ---- epilogue.  fp64 recombination (smallest weights first; C/D layout of the 16x16 i32 MFMA: col = lane & 15,
    // row = 4 * (lane >> 4) + reg), then per output: the combination staged as a 64 x 64 fp64 tile in LDS, from which the
    // tile and its mirror leave as rows (fp64: one 512-byte row per wave instruction; slices: 16 digits per lane and slice).
    const double* pr = op.par + (size_t)kk * 12;
    const double sc = pr[8];
    double v[2][2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double a = 0.0;
#pragma unroll
                for (int d = ND - 1; d >= 0; --d) a += (double)acc[i][j][d][r] * __builtin_ldexp(1.0, -(12 + 7 * d));
                v[i][j][r] = a * sc;
            }
    double* Tl = reinterpret_cast<double*>(smem8);
    const size_t pp = (size_t)p * p;
    const double* E1k = op.E1 ? op.E1 + (size_t)kk * pp : nullptr;
    const double* E2k = op.E2 ? op.E2 + (size_t)kk * pp : nullptr;
    const bool diag = (I == J);
#pragma unroll 1
    for (int o = 0; o < 2; ++o) {
        double* Cg = o == 0 ? op.C1 : op.C2;
        int8_t* Sg = o == 0 ? op.S1 : op.S2;
        if (!Cg && !Sg) continue;
        const int nS = o == 0 ? op.nS1 : op.nS2;
        const size_t sstr = o == 0 ? op.sstride1 : op.sstride2;
        const double cI = pr[4 * o + 0], cA = pr[4 * o + 1], cE1 = pr[4 * o + 2], cE2 = pr[4 * o + 3];
        const double inv = pr[9 + o];
        __syncthreads();                                    // the stage (or the previous output's tile) is free
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rl = wr + 16 * i + 4 * (lane >> 4) + r, cl = wc + 16 * j + (lane & 15);
                    const int gr = I0 + rl, gc = J0 + cl;
                    double val = 0.0;
                    if (gr < p && gc < p) {
                        val = cA * v[i][j][r];
                        if (gr == gc) val += cI;
                        if (E1k && cE1 != 0.0) val += cE1 * E1k[(size_t)gr * p + gc];
                        if (E2k && cE2 != 0.0) val += cE2 * E2k[(size_t)gr * p + gc];
                    }
                    Tl[rl * I8_TLD + cl] = val;
                }
        __syncthreads();
        if (Cg) {
            double* Ck = Cg + (size_t)kk * pp;
            // wave w writes rows 16 w .. 16 w + 15, lane = column
#pragma unroll 4
            for (int rr = 0; rr < 16; ++rr) {
                const int rl = 16 * wave + rr;
                if (I0 + rl < p && J0 + lane < p) {
                    const double x = diag ? Tl[min(rl, lane) * I8_TLD + max(rl, lane)] : Tl[rl * I8_TLD + lane];
                    Ck[(size_t)(I0 + rl) * p + J0 + lane] = x;
                }
                if (!diag && J0 + rl < p && I0 + lane < p) Ck[(size_t)(J0 + rl) * p + I0 + lane] = Tl[lane * I8_TLD + rl];
            }
        }
        if (Sg) {
            // thread -> row tr, 16 consecutive columns from tc: 16 digits = one 16-byte store per slice
            const int tr = tid >> 2, tc = (tid & 3) * 16;
            bool over = false;
#pragma unroll 1
            for (int orient = 0; orient < (diag ? 1 : 2); ++orient) {
                double x[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int c = tc + e;
                    const double t = orient ? Tl[c * I8_TLD + tr] : (diag ? Tl[min(tr, c) * I8_TLD + max(tr, c)] : Tl[tr * I8_TLD + c]);
                    x[e] = t * inv;
                }
                int8_t* dst = Sg + ((size_t)kk * P + (orient ? J0 : I0) + tr) * P + (orient ? I0 : J0) + tc;
                for (int t = 0; t < nS; ++t) {
                    const double w = __builtin_ldexp(1.0, 6 + 7 * t), wi = __builtin_ldexp(1.0, -(6 + 7 * t));
                    int pk[4] = {0, 0, 0, 0};
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        double d = rint(x[e] * w);
                        if (fabs(d) > 127.0) { over = true; d = d > 0 ? 127.0 : -127.0; }
                        x[e] -= d * wi;
                        pk[e >> 2] |= ((int)d & 0xff) << (8 * (e & 3));
                    }
                    *reinterpret_cast<v4i*>(dst + t * sstr) = (v4i){pk[0], pk[1], pk[2], pk[3]};
                }
            }
            if (over && op.flag) atomicOr(op.flag, 1);
        }
    }
}

template <int SA, int SB, int DMAX, int NSTG>
static void launch_i8_stg(hipStream_t st, const I8Op& op)
{
    const int T = op.P / 64;
    const size_t lds = std::max<size_t>(NSTG * (size_t)(SA + SB) * 4096, 64 * I8_TLD * sizeof(double));
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_symm_i8<SA, SB, DMAX, NSTG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((k_symm_i8<SA, SB, DMAX, NSTG>), dim3(T * (T + 1) / 2 * op.K), dim3(256), lds, st, op);
}

static int g_i8_stages = 0;          // 0: one stage (two workgroups per CU cover each other's loads: 84 vs 103 us at S = 7, K = 32)
void symm_i8_set_stages(int n) { g_i8_stages = n; }

template <int SA, int SB, int DMAX>
static void launch_i8(hipStream_t st, const I8Op& op)
{
    if (g_i8_stages == 2) launch_i8_stg<SA, SB, DMAX, 2>(st, op);
    else launch_i8_stg<SA, SB, DMAX, 1>(st, op);
}

// SA / SB slices of the operands, pairs t + u <= dmax.  Returns false for a combination that is not instantiated.
bool launch_symm_i8_op(hipStream_t st, const I8Op& op, int SA, int SB, int dmax)
{
#define GGL_I8(a, b, d) if (SA == a && SB == b && dmax == d) { launch_i8<a, b, d>(st, op); return true; }
    GGL_I8(7, 7, 6)
    GGL_I8(6, 6, 5)
    GGL_I8(5, 5, 4)
    GGL_I8(4, 4, 3)
    GGL_I8(3, 3, 2)
    GGL_I8(2, 2, 1)
    GGL_I8(8, 8, 7)
    GGL_I8(5, 5, 3)
    GGL_I8(4, 4, 2)
#undef GGL_I8
    return false;
}

// plain product C = A B (development entry point / tests): par built by the caller
bool launch_symm_i8(hipStream_t st, const int8_t* As, const int8_t* Bs, const double* par, double* C, int K, int p, int S, int dmax)
{
    I8Op op = {};
    const int P = (p + 63) / 64 * 64;
    op.As = As; op.Bs = Bs;
    op.sstrideA = op.sstrideB = (size_t)K * P * P;
    op.par = par;
    op.C1 = C;
    op.K = K; op.p = p; op.P = P;
    return launch_symm_i8_op(st, op, S, S, dmax);
}

// ---------------------------------------------------------------------------------------------
// The Omega-step on the int8 matrix cores: phiplus(W) = (W + (W^2 + 4 beta I)^(1/2)) / 2 (solver/ggl_helper.py:272-303) by the
// two-step Newton-Schulz schedules of newton_schulz.hip -- (5,9) at the headline -- with every product error-free split and
// the slice budget spent where it is needed (tools/proto_ozaki.py):
//   full products (S_FULL slices, triangular truncation):  A' = W W + 4 beta I;  B' = A' A' (second output: T1, the first
//   step's polynomial in A', B';  degree nine: U, then T1 = t0 I + t1 M + U B'/c^2 is one more product);  Y1 = (A'/c) T1;
//   M2 = T1 Y1 -- written as F = I - M2, the SMALL matrix (|F| <= 1 - l1^2) everything after it is a polynomial in:
//   t(M2) = g(F) = g0 + g1 F + F^2 (g2 + g3 F + g4 F^2),  E = g(F) - g0 I,  Omega = (W + sqrt(c) (g0 Y1 + Y1 E)) / 2.
//   Products of small matrices need few slices relative to their own scale: F F (S_F2), (g3 F + g4 F^2) F^2 (S_GF2), Y1 E
//   (S_YE slices, pairs t + u <= D_YE).
// Every scale is a power of two known on the host from the spectral bound c the step assumes (|W|_2^2 <= c - 4 beta, |A'|_2 <=
// c, |T1|_2 <= max t, |Y1|_2 <= 1, |F|_2 <= 1 - l1^2): a digit that does not fit raises `flag` (the bound was violated).
// ---------------------------------------------------------------------------------------------
static double pow2_ge(double x)
{
    if (!(x > 0.0)) return 1.0;
    int e;
    const double m = std::frexp(x, &e);          // x = m 2^e, 0.5 <= m < 1
    return std::ldexp(1.0, m == 0.5 ? e - 1 : e);
}

int i8_omega_alloc(I8Omega* w, int K, int p)
{
    w->K = K; w->p = p; w->P = (p + 63) / 64 * 64;
    w->sl = (size_t)K * w->P * w->P;
    if (hipMalloc(&w->slab, (size_t)I8_NSLICES * w->sl) != hipSuccess) return -1;
    if (hipMalloc(&w->par, (size_t)I8_MAXPROD * K * 12 * sizeof(double)) != hipSuccess) return -1;
    if (hipHostMalloc(&w->par_h, (size_t)I8_MAXPROD * K * 12 * sizeof(double)) != hipSuccess) return -1;
    if (hipMalloc(&w->wscale, K * sizeof(double)) != hipSuccess) return -1;
    if (hipHostMalloc(&w->wscale_h, K * sizeof(double)) != hipSuccess) return -1;
    if (hipMalloc(&w->flag, sizeof(int)) != hipSuccess) return -1;
    if (hipMemset(w->flag, 0, sizeof(int)) != hipSuccess) return -1;
    // the pad region of every slice stack is zero and stays zero (the epilogues write zeros there)
    if (hipMemset(w->slab, 0, (size_t)I8_NSLICES * w->sl) != hipSuccess) return -1;
    return 0;
}

void i8_omega_free(I8Omega* w)
{
    if (w->slab) (void)hipFree(w->slab);
    if (w->par) (void)hipFree(w->par);
    if (w->par_h) (void)hipHostFree(w->par_h);
    if (w->wscale) (void)hipFree(w->wscale);
    if (w->wscale_h) (void)hipHostFree(w->wscale_h);
    if (w->flag) (void)hipFree(w->flag);
    *w = I8Omega();
}

// slice-stack slots of the workspace (first slice of each)
enum { I8_SW = 0, I8_SA = 7, I8_ST = 14, I8_SY = 21, I8_SF = 28, I8_SF2 = 32, I8_SG = 36, I8_SE = 40, I8_SU = 45, I8_SB = 52 };
static_assert(I8_SB + 7 <= I8_NSLICES, "slice slots");

// Plans the step of the instances k0 .. k0 + Kp - 1 (one schedule for the part, built for its smallest l; scales and
// coefficients per instance).  Fills w->par_h / w->wscale_h rows of these instances and prog.  Returns the number of
// products, 0 when the schedule is not a two-step one (the caller takes the fp64 path), < 0 for non-finite input.
int i8_omega_plan(I8Omega* w, const double* cuse_h, const double* beta_h, int k0, int Kp, double tol, int degrees,
                  const I8Cfg& cfg, const I8Bufs& bufs, I8Prog* prog)
{
    const int K = w->K, p = w->p, P = w->P;
    double kappa = 1.0;
    std::vector<double> c(Kp);
    for (int i = 0; i < Kp; ++i) {
        const int k = k0 + i;
        c[i] = cuse_h[k] * (1.0 + 1e-10);
        if (!(c[i] > 0.0) || !std::isfinite(c[i]) || !(beta_h[k] > 0.0)) return -1;
        if (c[i] < 4.0 * beta_h[k]) c[i] = 4.0 * beta_h[k];
        kappa = std::fmax(kappa, c[i] / (4.0 * beta_h[k]));
    }
    const double lmin = 1.0 / std::sqrt(kappa);
    int deg[8];
    double co[8 * 6];
    int units = 0;
    const int n = ns_schedule_query(lmin, degrees, 8, deg, co, &units, tol);
    if (n != 2) return 0;
    const double* t1 = co;
    const double* t2 = co + 6;
    const int d1 = deg[0], d2 = deg[1];
    const double lo1 = t1[5];
    // g(f) = t2(1 - f)
    double g[5] = {0, 0, 0, 0, 0};
    {
        const double binom[5][5] = {{1, 0, 0, 0, 0}, {1, 1, 0, 0, 0}, {1, 2, 1, 0, 0}, {1, 3, 3, 1, 0}, {1, 4, 6, 4, 1}};
        for (int j = 0; j < 5; ++j)
            for (int i = 0; i <= j; ++i) g[i] += t2[j] * binom[j][i] * ((i & 1) ? -1.0 : 1.0);
    }
    // largest value of t1 on [lmin^2, 1] (T1's spectral norm)
    double tmax = 0.0, umax = 0.0;
    for (int i = 0; i <= 256; ++i) {
        const double m = lmin * lmin + (1.0 - lmin * lmin) * i / 256.0;
        tmax = std::fmax(tmax, std::fabs(t1[0] + m * (t1[1] + m * (t1[2] + m * (t1[3] + m * t1[4])))));
        umax = std::fmax(umax, std::fabs(t1[2] + m * (t1[3] + m * t1[4])));
    }
    const double sT = pow2_ge(tmax * 1.001), sU = pow2_ge(umax * 1.001);
    const double fmax_ = std::fmax(1.0 - lo1 * lo1, 1e-300) * 1.0001;
    const double sF = pow2_ge(fmax_), sF2 = sF * sF;
    const double sG = pow2_ge(std::fabs(g[3]) * sF + std::fabs(g[4]) * sF2);
    double sE;
    if (d2 == 3) sE = pow2_ge(std::fabs(g[1]) * sF);
    else if (d2 == 5) sE = pow2_ge(std::fabs(g[1]) * sF + std::fabs(g[2]) * sF2);
    else sE = pow2_ge(std::fabs(g[1]) * sF + std::fabs(g[2]) * sF2 + sG * sF2);

    const size_t pp = (size_t)p * p, PP = (size_t)P * P;
    int np = 0;
    auto slot = [&](int first) { return w->slab + (size_t)first * w->sl + (size_t)k0 * PP; };
    auto new_op = [&](int first_a, int first_b, int SA, int SB, int dmax) -> I8Op& {
        I8Prog::Prod& pr = prog->prod[np];
        pr.SA = SA; pr.SB = SB; pr.dmax = dmax;
        I8Op& op = pr.op;
        op = I8Op();
        op.As = slot(first_a); op.Bs = slot(first_b);
        op.sstrideA = op.sstrideB = w->sl;
        op.par = w->par + ((size_t)np * K + k0) * 12;
        op.sstride1 = op.sstride2 = w->sl;
        op.flag = w->flag;
        op.K = Kp; op.p = p; op.P = P;
        return op;
    };
    auto row = [&](int g_, int i) { return w->par_h + ((size_t)g_ * K + k0 + i) * 12; };
    auto zero_rows = [&](int g_) { for (int i = 0; i < Kp; ++i) for (int j = 0; j < 12; ++j) row(g_, i)[j] = 0.0; };
    const int SF = cfg.s_full, DF = cfg.s_full - 1;
    std::vector<double> sW(Kp), sA(Kp);
    for (int i = 0; i < Kp; ++i) {
        sW[i] = pow2_ge(std::sqrt(std::fmax(c[i] - 4.0 * beta_h[k0 + i], 1e-300)) * 1.0001);
        sA[i] = pow2_ge(c[i] * 1.0001);
        w->wscale_h[k0 + i] = sW[i];
    }
    // P1: A' = W W + 4 beta I
    {
        I8Op& op = new_op(I8_SW, I8_SW, SF, SF, DF);
        op.C1 = bufs.Ap + k0 * pp; op.S1 = slot(I8_SA); op.nS1 = SF;
        zero_rows(np);
        for (int i = 0; i < Kp; ++i) { double* r = row(np, i); r[0] = 4.0 * beta_h[k0 + i]; r[1] = 1.0; r[8] = sW[i] * sW[i]; r[9] = 1.0 / sA[i]; r[10] = 1.0; }
        ++np;
    }
    // P2: B' = A' A';  second output T1 (cubic / quintic first step) or U (degree nine)
    {
        I8Op& op = new_op(I8_SA, I8_SA, SF, SF, DF);
        op.C1 = bufs.Bp + k0 * pp; op.E1 = bufs.Ap + k0 * pp;
        if (d1 == 9) { op.S1 = slot(I8_SB); op.nS1 = SF; op.S2 = slot(I8_SU); op.nS2 = SF; }
        else { op.S2 = slot(I8_ST); op.nS2 = SF; }
        zero_rows(np);
        for (int i = 0; i < Kp; ++i) {
            double* r = row(np, i);
            const double ck = c[i];
            r[1] = 1.0; r[8] = sA[i] * sA[i]; r[9] = 1.0 / (sA[i] * sA[i]);
            if (d1 == 9) { r[4] = t1[2]; r[5] = t1[4] / (ck * ck); r[6] = t1[3] / ck; r[10] = 1.0 / sU; }
            else { r[4] = t1[0]; r[5] = (d1 == 5 ? t1[2] / (ck * ck) : 0.0); r[6] = t1[1] / ck; r[10] = 1.0 / sT; }
        }
        ++np;
    }
    if (d1 == 9) {
        // T1 = t0 I + (t1/c) A' + U B' / c^2
        I8Op& op = new_op(I8_SU, I8_SB, SF, SF, DF);
        op.E1 = bufs.Ap + k0 * pp; op.S1 = slot(I8_ST); op.nS1 = SF;
        zero_rows(np);
        for (int i = 0; i < Kp; ++i) {
            double* r = row(np, i);
            const double ck = c[i];
            r[0] = t1[0]; r[1] = 1.0 / (ck * ck); r[2] = t1[1] / ck; r[8] = sU * sA[i] * sA[i]; r[9] = 1.0 / sT; r[10] = 1.0;
        }
        ++np;
    }
    // P3: Y1 = (A'/c) T1
    {
        I8Op& op = new_op(I8_SA, I8_ST, SF, SF, DF);
        op.C1 = bufs.Y1 + k0 * pp; op.S1 = slot(I8_SY); op.nS1 = SF;
        zero_rows(np);
        for (int i = 0; i < Kp; ++i) { double* r = row(np, i); r[1] = 1.0 / c[i]; r[8] = sA[i] * sT; r[9] = 1.0; r[10] = 1.0; }
        ++np;
    }
    // P4: F = I - T1 Y1  (cubic last step: second output E = g1 F)
    {
        I8Op& op = new_op(I8_ST, I8_SY, SF, SF, DF);
        if (d2 >= 5) { op.C1 = bufs.F + k0 * pp; op.S1 = slot(I8_SF); op.nS1 = cfg.s_f2; }
        else { op.S2 = slot(I8_SE); op.nS2 = cfg.s_ye; }
        zero_rows(np);
        for (int i = 0; i < Kp; ++i) {
            double* r = row(np, i);
            r[0] = 1.0; r[1] = -1.0; r[8] = sT; r[9] = 1.0 / sF; r[4] = g[1]; r[5] = -g[1]; r[10] = 1.0 / sE;
        }
        ++np;
    }
    if (d2 >= 5) {
        // P5: F^2;  quintic: E = g1 F + g2 F^2;  degree nine: second output G = g3 F + g4 F^2
        I8Op& op = new_op(I8_SF, I8_SF, cfg.s_f2, cfg.s_f2, cfg.s_f2 - 1);
        op.E1 = bufs.F + k0 * pp;
        if (d2 == 9) { op.C1 = bufs.F2 + k0 * pp; op.S1 = slot(I8_SF2); op.nS1 = cfg.s_gf2; op.S2 = slot(I8_SG); op.nS2 = cfg.s_gf2; }
        else { op.S1 = slot(I8_SE); op.nS1 = cfg.s_ye; }
        zero_rows(np);
        for (int i = 0; i < Kp; ++i) {
            double* r = row(np, i);
            r[8] = sF2;
            if (d2 == 9) { r[1] = 1.0; r[9] = 1.0 / sF2; r[5] = g[4]; r[6] = g[3]; r[10] = 1.0 / sG; }
            else { r[1] = g[2]; r[2] = g[1]; r[9] = 1.0 / sE; r[10] = 1.0; }
        }
        ++np;
    }
    if (d2 == 9) {
        // P6: E = g1 F + g2 F^2 + G F^2
        I8Op& op = new_op(I8_SG, I8_SF2, cfg.s_gf2, cfg.s_gf2, cfg.s_gf2 - 1);
        op.E1 = bufs.F + k0 * pp; op.E2 = bufs.F2 + k0 * pp; op.S1 = slot(I8_SE); op.nS1 = cfg.s_ye;
        zero_rows(np);
        for (int i = 0; i < Kp; ++i) { double* r = row(np, i); r[1] = 1.0; r[2] = g[1]; r[3] = g[2]; r[8] = sG * sF2; r[9] = 1.0 / sE; r[10] = 1.0; }
        ++np;
    }
    // P7: Omega = (W + sqrt(c) (g0 Y1 + Y1 E)) / 2
    {
        I8Op& op = new_op(I8_SY, I8_SE, cfg.s_ye, cfg.s_ye, cfg.d_ye);
        op.E1 = bufs.Y1 + k0 * pp; op.E2 = bufs.W + k0 * pp; op.C1 = bufs.Om + k0 * pp;
        zero_rows(np);
        for (int i = 0; i < Kp; ++i) {
            double* r = row(np, i);
            const double sc = std::sqrt(c[i]);
            r[1] = 0.5 * sc; r[2] = 0.5 * sc * g[0]; r[3] = 0.5; r[8] = sE; r[9] = 1.0; r[10] = 1.0;
        }
        ++np;
    }
    prog->nprod = np;
    prog->units = units;
    prog->k0 = k0; prog->Kp = Kp;
    prog->S_W = SF;
    return np;
}

// Launches the planned step of one part on `st` (the parameter rows must have been uploaded: i8_omega_upload).
bool i8_omega_run(hipStream_t st, I8Omega* w, const I8Prog& prog, const double* W)
{
    const size_t pp = (size_t)w->p * w->p, PP = (size_t)w->P * w->P;
    launch_slice_i8(st, W + prog.k0 * pp, w->wscale + prog.k0, w->slab + (size_t)I8_SW * w->sl + (size_t)prog.k0 * PP, prog.Kp, w->p,
                    prog.S_W, w->flag, w->sl);
    for (int i = 0; i < prog.nprod; ++i)
        if (!launch_symm_i8_op(st, prog.prod[i].op, prog.prod[i].SA, prog.prod[i].SB, prog.prod[i].dmax)) return false;
    return true;
}

}  // namespace ggl
#endif  // GGL_DEV
