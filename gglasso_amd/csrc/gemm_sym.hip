// Batched FP64 matrix-core product of COMMUTING SYMMETRIC matrices, the workhorse of the
// eigendecomposition-free Omega-step (newton_schulz.hip):
//
//     C[k] = cI[k]*I + cAcc[k]*(A[k] * B[k]) + cE[k]*E[k]        (+ optional C2[k] = dI[k]*I + dC[k]*C[k])
//
// A, B symmetric and commuting  =>  A*B = A^T*B is symmetric.  Written as a "TN" product both
// operands are read along contiguous rows (row m of A, columns I0.. and row m of B, columns J0..),
// only tile pairs I<=J are computed (v_mfma_f64_16x16x4_f64), and the I<J tiles are mirrored through
// an LDS transpose, so every global access is row-contiguous and the output is bitwise symmetric.
//
// Tiling: BM x BM block tile, NW waves as a (BM/WM) x (BM/WN) grid, each wave WM x WN = (WM/16) x
// (WN/16) MFMA tiles; k-slab BK.  The next slab is fetched from HBM/L2 into registers while the
// current one is consumed from LDS (software prefetch), one barrier pair per slab.
#include <algorithm>
#include <initializer_list>
#include <mutex>
#include <unordered_map>
#include <type_traits>

#include "common.hpp"
#include "kernels.hpp"

namespace ggl {

typedef double v4d __attribute__((ext_vector_type(4)));

// second output of a product launch, C2 = dI I + dC C + dE E: one fixed evaluation order at every store site (upper
// triangle, diagonal-tile mirror, LDS-transposed mirror), so that the output is bitwise symmetric
__device__ __forceinline__ double c2val(double dC, double v, double dE, double e) { return __builtin_fma(dE, e, dC * v); }

template <int BM, int BK, int WM, int WN, bool LM>
struct SymCfg {
    static constexpr int NWR = BM / WM, NWC = BM / WN, NW = NWR * NWC, NT = NW * 64;
    static constexpr int TI = WM / 16, TJ = WN / 16;
    static constexpr int LDS_LD = BM + 16;           // consecutive rows start on opposite bank halves
    static constexpr int SLAB = BK * LDS_LD;          // doubles per operand slab
    static constexpr int CLD = BM + 1;
    static constexpr int LDS_DOUBLES = (!LM || 2 * SLAB > BM * CLD) ? 2 * SLAB : BM * CLD;   // LM: mirror via LDS
    static constexpr int LPT = (BK * BM) / NT;       // elements per thread per operand slab
    static_assert((BK * BM) % NT == 0, "slab must divide evenly over the threads");
    static_assert(NT % BM == 0, "a row of the slab must be covered by whole thread rows");
};

constexpr int sym_nt(int BM, int WM, int WN) { return (BM / WM) * (BM / WN) * 64; }

// XCD-aware work decode.  The 8 XCDs have private 4 MiB L2s and the dispatcher deals consecutive
// workgroup ids round-robin over them (id % 8, observed; only speed depends on it).  A 1-D grid of
// 8 * ceil(K/8) * ntiles ids is decoded so that all tiles of instance k run on XCD k % 8 and
// follow each other in dispatch order: the two operand matrices of an instance (2 x 8 p^2 bytes)
// are then fetched into ONE L2 once and reused by all of its tiles, instead of being streamed
// into all eight.  Batches smaller than 8 keep the plain (tile, k) order so that every XCD has work.
static constexpr int NXCD = 8;
// batches of 1, 2 or 4 instances: 8 / K XCDs share one instance (its tiles interleaved over them), so an XCD's L2 still
// holds the operands of ONE instance only instead of slices of all of them
__host__ __device__ inline int xcd_share(int K) { return (K == 1 || K == 2 || K == 4) ? NXCD / K : 0; }
// K >= 8: whole rounds of eight instances as above; the K % 8 instances that are left over are dealt like a small batch
// of their own (8 / r XCDs per instance for r = 1, 2, 4, else plain tile order) -- as a "ninth, tenth, ..." instance of the
// first XCDs they would leave the other XCDs idle for a whole instance (K = 20: 3 against 2 instances per XCD; measured
// with an uneven split of the headline: 18 instead of 16 per launch costs 8 %).
__host__ __device__ inline int xcd_grid_small(int ntiles, int K)
{
    const int g = xcd_share(K);
    return g ? NXCD * ((ntiles + g - 1) / g) : NXCD * ((ntiles * K + NXCD - 1) / NXCD);
}
__host__ __device__ inline int xcd_grid(int ntiles, int K)
{
    if (K < NXCD) return xcd_grid_small(ntiles, K);
    const int r = K % NXCD;
    return NXCD * (K / NXCD) * ntiles + (r ? xcd_grid_small(ntiles, r) : 0);
}
__device__ __forceinline__ bool decode_block_small(int ntiles, int K, int& k, int& tile, int L)
{
    const int g = xcd_share(K);
    if (g) {
        const int xcd = L % NXCD, slot = L / NXCD;
        k = xcd / g;
        tile = slot * g + xcd % g;
        return tile < ntiles;
    }
    k = L / ntiles;
    tile = L % ntiles;
    return k < K;
}
__device__ __forceinline__ bool decode_block_xcd(int ntiles, int K, int& k, int& tile, int L)
{
    if (K < NXCD) return decode_block_small(ntiles, K, k, tile, L);
    const int nfull = NXCD * (K / NXCD) * ntiles;          // a multiple of 8: L % 8 is still the XCD behind it
    if (L < nfull) {
        const int xcd = L % NXCD, slot = L / NXCD;
        k = (slot / ntiles) * NXCD + xcd;
        tile = slot % ntiles;
        return true;
    }
    const bool ok = decode_block_small(ntiles, K % NXCD, k, tile, L - nfull);
    k += (K / NXCD) * NXCD;
    return ok;
}
__device__ __forceinline__ bool decode_block_xcd(int ntiles, int K, int& k, int& tile)
{
    return decode_block_xcd(ntiles, K, k, tile, (int)blockIdx.x);
}

// ABL != 0: timing ablations and the timeline probe, instantiated by GGL_DEV builds only (tools/bench_tail.py)
template <int BM, int BK, int WM, int WN, bool LM, int ABL = 0>
__global__ __launch_bounds__(sym_nt(BM, WM, WN)) void k_symm_tn(
    const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C, double* __restrict__ C2,
    const double* __restrict__ E, const double* __restrict__ coef, int K, int p, const double* __restrict__ A1,
    const double* __restrict__ B1, double* __restrict__ C1, int K1, double* __restrict__ maxdev)
{
    // maxdev != null: maxdev[k] = max over the instance of |C - I| (atomic max on the bit pattern of a
    // non-negative double: order independent, hence deterministic).
    // Instances k < K use (A, B, C, C2, E); instances K <= k < K + K1 use the second set (A1, B1, C1):
    // two independent products of one Newton-Schulz step share a launch, so that the chip sees 2K
    // instances' worth of tiles at once.  coef is indexed by the combined k.
    using Cfg = SymCfg<BM, BK, WM, WN, LM>;
    __shared__ __attribute__((aligned(16))) double smem[Cfg::LDS_DOUBLES];
    double* As = smem;
    double* Bs = smem + Cfg::SLAB;
    const int T = (p + BM - 1) / BM;
    int k, b;
    if (!decode_block_xcd(T * (T + 1) / 2, K + K1, k, b)) return;
    const bool second = k >= K;
    const int kk = second ? k - K : k;
    int I = 0;
    while (b >= T - I) { b -= T - I; ++I; }
    const int J = I + b;
    const int I0 = I * BM, J0 = J * BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = (wave / Cfg::NWC) * WM, wc = (wave % Cfg::NWC) * WN;
    const size_t pp = (size_t)p * p;
    const double* Ak = (second ? A1 : A) + (size_t)kk * pp;
    const double* Bk = (second ? B1 : B) + (size_t)kk * pp;

#ifdef GGL_DEV
    long long t_start = 0, t_loop = 0, t_loop_end = 0;
    if (ABL == 3) t_start = clock64();
#endif

    v4d acc[Cfg::TI][Cfg::TJ];
#pragma unroll
    for (int i = 0; i < Cfg::TI; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TJ; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

    // slab element e = tid + q*NT -> row e / BM, col e % BM
    constexpr int RSTEP = Cfg::NT / BM;
    constexpr int NST = 1;                      // slabs in flight (register stages); 1 measured best, 3 was -8%
    const int lcol = tid % BM, lrow = tid / BM;
    const bool aok = (I0 + lcol) < p, bok = (J0 + lcol) < p;
    double ra[NST][Cfg::LPT], rb[NST][Cfg::LPT];

    // Unconditional fetch from clamped (always valid) addresses with 32-bit element offsets (p*p < 2^31);
    // out-of-range elements are zeroed only when the slab is staged into LDS, so nothing consumes a load
    // early and NST slabs (NST * 2 * BK * BM * 8 bytes per workgroup) stay in flight: the product is
    // latency-bound on the L2/MALL path otherwise.
    const unsigned pu = (unsigned)p, pm1 = (unsigned)(p - 1);
    const unsigned ca = min((unsigned)(I0 + lcol), pm1), cb = min((unsigned)(J0 + lcol), pm1);
    auto fetch = [&](int m0, double (&xa)[Cfg::LPT], double (&xb)[Cfg::LPT]) {
#pragma unroll
        for (int q = 0; q < Cfg::LPT; ++q) {
            if (ABL == 1 || ABL == 4 || ABL == 5) { xa[q] = 1.0 + q; xb[q] = 2.0 + q; continue; }   // ablation: no global loads
            const unsigned ro = min((unsigned)(m0 + lrow + q * RSTEP), pm1) * pu;
            xa[q] = Ak[ro + ca];
            xb[q] = Bk[ro + cb];
        }
    };
    auto stage = [&](int m0, const double (&xa)[Cfg::LPT], const double (&xb)[Cfg::LPT]) {
#pragma unroll
        for (int q = 0; q < Cfg::LPT; ++q) {
            const int row = lrow + q * RSTEP;
            const bool in = (m0 + row) < p;
            As[row * Cfg::LDS_LD + lcol] = (in && aok) ? xa[q] : 0.0;
            Bs[row * Cfg::LDS_LD + lcol] = (in && bok) ? xb[q] : 0.0;
        }
    };
    // all fragment reads of the slab first, then the MFMAs back to back: one LDS round trip per slab
    // instead of one per k-quad
    // nq: k-quads of the slab that lie inside the matrix (the last slab of p = 200 at BK = 32 has 2 of 8)
    auto compute = [&](int nq) {
        double af[BK / 4][Cfg::TI], bf[BK / 4][Cfg::TJ];
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            if (kk >= nq) break;
            const int row = kk * 4 + (lane >> 4);
#pragma unroll
            for (int i = 0; i < Cfg::TI; ++i) af[kk][i] = As[row * Cfg::LDS_LD + wr + i * 16 + (lane & 15)];
#pragma unroll
            for (int j = 0; j < Cfg::TJ; ++j) bf[kk][j] = Bs[row * Cfg::LDS_LD + wc + j * 16 + (lane & 15)];
        }
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            if (kk >= nq) break;
#pragma unroll
            for (int i = 0; i < Cfg::TI; ++i)
#pragma unroll
                for (int j = 0; j < Cfg::TJ; ++j) {
                    if (ABL == 2) { acc[i][j][0] += af[kk][i] * bf[kk][j]; continue; }   // ablation: no MFMA
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk][i], bf[kk][j], acc[i][j], 0, 0, 0);
                }
        }
    };

    // On a diagonal tile a wave whose sub-tile lies entirely below the diagonal produces nothing that is
    // kept (only gi <= gj survives there): it still stages and synchronises, but issues no MFMA.
    const bool dead_wave = (I == J) && (wr >= wc + WN);
#pragma unroll
    for (int s = 0; s < NST; ++s) fetch(s * BK, ra[s], rb[s]);
#ifdef GGL_DEV
    if (ABL == 3) t_loop = clock64();
#endif
    for (int m0 = 0; m0 < p; m0 += NST * BK) {
#pragma unroll
        for (int s = 0; s < NST; ++s) {
            const int ms = m0 + s * BK;
            if (ms < p) {                                   // uniform over the workgroup
                if (ABL != 4) stage(ms, ra[s], rb[s]);       // ABL 4: ds_read + MFMA only; ABL 5: + ds_write
                if (ABL != 4 && ABL != 5) __syncthreads();
                fetch(ms + NST * BK, ra[s], rb[s]);          // past the end: clamped, never staged
                if (!dead_wave) compute(min(BK / 4, (p - ms + 3) / 4));
                if (ABL != 4 && ABL != 5) __syncthreads();
            }
        }
    }

#ifdef GGL_DEV
    if (ABL == 3) t_loop_end = clock64();
#endif
    // epilogue.  C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
    const double cI = coef[k * NS_NCOEF + 0], cAcc = coef[k * NS_NCOEF + 1], cE = coef[k * NS_NCOEF + 2];
    const double dI = coef[k * NS_NCOEF + 3], dC = coef[k * NS_NCOEF + 4], dE = coef[k * NS_NCOEF + 5];
    double* Ck = (second ? C1 : C) + (size_t)kk * pp;
    double* C2k = (C2 && !second) ? C2 + (size_t)kk * pp : nullptr;
    const double* Ek = (E && !second) ? E + (size_t)kk * pp : nullptr;
#pragma unroll
    for (int ti = 0; ti < Cfg::TI; ++ti)
#pragma unroll
        for (int tj = 0; tj < Cfg::TJ; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wr + ti * 16 + (lane >> 4) + 4 * r;
                const int col = wc + tj * 16 + (lane & 15);
                const int gi = I0 + row, gj = J0 + col;
                double v = cAcc * acc[ti][tj][r];
                const bool in = gi < p && gj < p;
                if (in && (I != J || gi <= gj)) {
                    if (gi == gj) v += cI;
                    const double e0 = Ek ? Ek[(size_t)gi * p + gj] : 0.0;
                    v += cE * e0;
                    Ck[(size_t)gi * p + gj] = v;
                    if (C2k) C2k[(size_t)gi * p + gj] = c2val(dC, v, dE, e0) + (gi == gj ? dI : 0.0);
                    // mirrored store.  Diagonal tiles: only the upper triangle is kept, so the result is
                    // bitwise symmetric even though A != B.  Off-diagonal tiles without the LDS transpose:
                    // the four r-values of a lane quad complete a 128-B line.
                    if ((I == J && gi != gj) || (!LM && I != J)) {
                        Ck[(size_t)gj * p + gi] = v;
                        if (C2k) C2k[(size_t)gj * p + gi] = c2val(dC, v, dE, e0);
                    }
                }
                if (LM && I != J) smem[row * Cfg::CLD + col] = v;
            }
    if (maxdev && ABL != 3) {
        double dev = 0.0;
#pragma unroll
        for (int ti = 0; ti < Cfg::TI; ++ti)
#pragma unroll
            for (int tj = 0; tj < Cfg::TJ; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gi = I0 + wr + ti * 16 + (lane >> 4) + 4 * r;
                    const int gj = J0 + wc + tj * 16 + (lane & 15);
                    if (gi < p && gj < p && (I != J || gi <= gj)) {
                        double v = cAcc * acc[ti][tj][r] + (gi == gj ? cI - 1.0 : 0.0);
                        if (Ek) v += cE * Ek[(size_t)gi * p + gj];      // the deviation of the OUTPUT, E term included
                        dev = fmax(dev, fabs(v));
                    }
                }
        dev = wave_max(dev);
        if (lane == 0 && dev > 0.0)
            atomicMax(reinterpret_cast<unsigned long long*>(maxdev + k), (unsigned long long)__double_as_longlong(dev));
    }
    if (LM && I != J) {
        __syncthreads();
        for (int e = tid; e < BM * BM; e += Cfg::NT) {
            const int a = e / BM, c = e % BM;   // out[J0+a][I0+c] = tile[c][a]
            if (J0 + a < p && I0 + c < p) {
                const double v = smem[c * Cfg::CLD + a];
                Ck[(size_t)(J0 + a) * p + I0 + c] = v;
                if (C2k) {
                    // E is bitwise symmetric (an output of this kernel family): its mirrored entry IS the upper one
                    const double e0 = (dE != 0.0 && Ek) ? Ek[(size_t)(J0 + a) * p + I0 + c] : 0.0;
                    C2k[(size_t)(J0 + a) * p + I0 + c] = c2val(dC, v, dE, e0);
                }
            }
        }
    }
#ifdef GGL_DEV
    if (ABL == 3 && tid == 0) {
        // timeline probe: maxdev is (ab)used as a [gridDim.x][5] long long buffer
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        long long* tl = reinterpret_cast<long long*>(maxdev) + (size_t)blockIdx.x * 5;
        unsigned xcc = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        tl[0] = t_start; tl[1] = t_loop; tl[2] = t_loop_end; tl[3] = clock64(); tl[4] = (long long)xcc;
    }
#endif
}

template <int BM, int BK, int WM, int WN, bool LM>
static void launch_cfg(hipStream_t st, const double* A, const double* B, double* C, double* C2, const double* E,
                       const double* coef, int K, int p, const double* A1 = nullptr, const double* B1 = nullptr,
                       double* C1 = nullptr, int K1 = 0, double* maxdev = nullptr)
{
    using Cfg = SymCfg<BM, BK, WM, WN, LM>;
    const int T = (p + BM - 1) / BM;
    hipLaunchKernelGGL((k_symm_tn<BM, BK, WM, WN, LM>), dim3(xcd_grid(T * (T + 1) / 2, K + K1)), dim3(Cfg::NT), 0, st, A,
                       B, C, C2, E, coef, K, p, A1, B1, C1, K1, maxdev);
}

// ---------------------------------------------------------------------------------------------
// k_symm_dl: the same symmetric product with the operand slabs DMA'd straight into LDS
// (global_load_lds, 16 bytes per lane, no VGPR round trip and no ds_write pass) and a double-buffered
// LDS image, one barrier per k-slab.  64x64 tile, 4 waves (32x32 each), k-slab 16.
//   * one wave instruction moves 1 KiB = two 64-double rows; its LDS destination is wave-linear
//     (base + lane*16), so the slab is stored unpadded [16][64]; rows r and r+1 would put the two 16-lane
//     groups of a half wave on the same banks, so odd rows are stored with their 16-double halves
//     swapped -- done on the SOURCE address (the DMA cannot scatter), and undone in the fragment read.
//   * needs 16-byte aligned rows: p even (odd p takes k_symm_tn).  Out-of-range columns read a
//     clamped address (their products only reach outputs that are never stored); out-of-range k-rows of
//     the last slab are zeroed in LDS after the DMA has landed.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N> __device__ __forceinline__ void wait_vmcnt()
{
    static_assert(N >= 0 && N <= 63, "vmcnt range");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else static_assert(N == 0, "add the immediate");
}

// BK: k-slab depth (16 or 32); NSTG: slabs resident in LDS (2 = double buffer; more = deeper DMA prefetch)
// BM: tile edge (64, or 32 for under-filled batches: 4x the workgroups, and with NSTG = 4 enough slabs in flight to
// cover the load latency that bounds the small-batch regime); ABL 1: no mirror write (timing ablation)
// NW: waves per workgroup (4: 2 x 2 waves of BM/2 x BM/2; 8: 4 x 2 waves of BM/4 x BM/2 -- two waves per SIMD, for batches
// so small that only one workgroup lands on a CU and a lone wave per SIMD cannot hide its LDS round trips and barriers)
// symm_dl_tile: one output tile (instance k of the combined batch, tile pair b) -- the whole body of k_symm_dl
// AUX: cache policy of every global READ of the tile body (operand DMA and the E term): 0 = default; 16 = sc1, i.e. served by
// the XCD's L2 past this CU's vector L1 -- what a persistent kernel needs to read tiles that ANOTHER CU of the same XCD
// wrote earlier in the same launch (a CU's L1 is never refreshed by other CUs' stores; MI355X_MICROARCH.md).
template <int AUX> __device__ __forceinline__ double ld_pol(const double* p)
{
    if constexpr (AUX == 0) return *p;
    else return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // global_load_dwordx2 ... sc1
}

template <int BK, int NSTG, int ABL, int BM, int NW, int AUX = 0>
__device__ __forceinline__ void symm_dl_tile(const double* __restrict__ A, const double* __restrict__ B,
                                             double* __restrict__ C, double* __restrict__ C2,
                                             const double* __restrict__ E, const double* __restrict__ coef, int K,
                                             int p, const double* __restrict__ A1, const double* __restrict__ B1,
                                             double* __restrict__ C1, int K1, double* __restrict__ maxdev,
                                             double* __restrict__ rowpart, double* __restrict__ fropart, int k, int b,
                                             double* smem)
{
    // smem: the caller's NSTG * 2 * BK * BM doubles of LDS ([buf][A|B][BK][BM]; later the mirror tile)
    // rowpart / fropart != null: the launch also leaves what a spectral bound of C needs, with no pass over C --
    // rowpart[k][s][i] = sum over the columns of tile-column s of |C[i][.]| (summed over s: the row sums of |C|) and
    // fropart[k][tile] = the tile's share of |C|_F^2 (mirror included), every slot written by exactly one workgroup
    constexpr int NT = NW * 64;
    constexpr int WM = BM / (NW / 2), WN = BM / 2, TI = WM / 16, TJ = WN / 16;
    constexpr int RPI = 128 / BM;                        // slab rows per DMA instruction (1 KiB)
    constexpr int LPR = 64 / RPI;                        // lanes per row
    constexpr int SLAB = BK * BM;                        // doubles per operand slab (8 KiB at BK = 16)
    constexpr int IPW = BK / (NW * RPI);                 // DMA instructions per wave, operand and slab
    constexpr int RPW = BK / NW;                         // slab rows a wave's DMA writes
    static_assert(NW == 4 || NW == 8, "waves per workgroup");
    static_assert(TI >= 1 && TJ >= 1, "wave tile");
    static_assert(BM == 64 || BM == 32, "tile edge");
    static_assert(IPW >= 1, "k-slab too shallow for this tile");
    static_assert(NSTG * 2 * SLAB >= BM * BM, "the mirror tile reuses the slab storage");
    const int T = (p + BM - 1) / BM;
    const int blockTile = b;
    const bool second = k >= K;
    const int kk = second ? k - K : k;
    int I = 0;
    while (b >= T - I) { b -= T - I; ++I; }
    const int J = I + b;
    const int I0 = I * BM, J0 = J * BM;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // wave-uniform: everything derived from it is scalar
    const int wr = (wave >> 1) * WM, wc = (wave & 1) * WN;
    const size_t pp = (size_t)p * p;
    const double* Ak = (second ? A1 : A) + (size_t)kk * pp;
    const double* Bk = (second ? B1 : B) + (size_t)kk * pp;

    v4d acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

    // DMA geometry: wave w issues instructions i = 2w, 2w+1 per operand; instruction i covers slab rows
    // 2i, 2i+1; lane l -> row 2i + (l >> 5), LDS position (l & 31) * 2, source column position ^ 16*(row & 1).
    // The per-lane source pointers are kept for the NEXT slab to be issued and advanced by BK rows per slab: every vector
    // instruction beside an MFMA takes matrix-pipe issue slots (tools/probe_mfma_mix.py: 2 integer VALU per MFMA cost 11 % of
    // the FP64 MFMA rate, 8 cost 17 %), and recomputing min / mul / 64-bit adds per instruction was ~25 of them per slab.
    // Only the last, partial slab (rows beyond the matrix clamped to its last row) goes the long way.
    // (pm2: the last column PAIR a lane may start at -- p - 2 for even p; for odd p the pair (p - 1, p), whose second half is
    // the first element of the next row: in the stack for every row but the last row of the last instance, where it is the 8
    // bytes behind the stack -- every stack that can be an operand is allocated with that slack, stack_bytes())
    const unsigned pu = (unsigned)p, pm1 = (unsigned)(p - 1), pm2 = (unsigned)(p - 1) & ~1u;
    const int lrow = lane / LPR;
    const unsigned cpos = (unsigned)((lane % LPR) * 2) ^ (unsigned)(16 * (lrow & 1));
    const unsigned ca = min((unsigned)I0 + cpos, pm2), cb = min((unsigned)J0 + cpos, pm2);
    const double* pa[IPW];
    const double* pb[IPW];
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
        const unsigned r0 = min((unsigned)(RPI * (wave * IPW + j) + lrow), pm1);
        pa[j] = Ak + (size_t)r0 * pu + ca;
        pb[j] = Bk + (size_t)r0 * pu + cb;
    }
    const size_t slab_step = (size_t)BK * pu;
    const int Sfull = p / BK;                                            // slabs that lie inside the matrix entirely
    auto issue = [&](int sidx, int buf) {                                // slabs are issued in increasing order
        if (ABL == 2) return;                                            // timing ablation: no operand loads at all
        double* base = smem + (size_t)buf * 2 * SLAB + (wave * IPW) * 128;   // wave-uniform LDS base
        if (sidx < Sfull) {
#pragma unroll
            for (int j = 0; j < IPW; ++j) {
                double* la = base + j * 128;
                __builtin_amdgcn_global_load_lds((gptr_t)pa[j], (lptr_t)la, 16, 0, AUX);
                __builtin_amdgcn_global_load_lds((gptr_t)pb[j], (lptr_t)(la + SLAB), 16, 0, AUX);
                pa[j] += slab_step;
                pb[j] += slab_step;
            }
        } else {
            const int m0 = sidx * BK;
#pragma unroll
            for (int j = 0; j < IPW; ++j) {
                const int i = wave * IPW + j;
                const unsigned ro = min((unsigned)(m0 + RPI * i + lrow), pm1) * pu;
                double* la = base + j * 128;
                __builtin_amdgcn_global_load_lds((gptr_t)(Ak + ro + ca), (lptr_t)la, 16, 0, AUX);
                __builtin_amdgcn_global_load_lds((gptr_t)(Bk + ro + cb), (lptr_t)(la + SLAB), 16, 0, AUX);
            }
        }
    };
    // Which 16x16 blocks this wave owns.  acc[i][j] belongs to rows rbk[i].., columns cbk[j].. of the tile and the MFMAs
    // of a k-quad run in the order X = acc[0][0], Y = acc[0][1], Z = acc[1][1], U = acc[1][0].  Off-diagonal tiles: the
    // wave's own 32x32 quadrant, all four.  A DIAGONAL tile is written from its upper triangle (in-tile mirror), i.e. 10
    // of its 16 blocks: the two diagonal quadrants need X, Y, Z (U lies below the diagonal); the upper-right quadrant needs
    // all four, and the wave of the lower-left quadrant would have nothing to do -- it takes that quadrant's block (1,1)
    // instead, and the quadrant's own wave names its column blocks the other way round so that what is left to it is again
    // X, Y, Z.  Every wave of a diagonal tile then skips U: 3 MFMAs per k-quad instead of 4 on the 8 diagonal tiles of the
    // 36 of a p = 500 product, for ONE wave-uniform branch per k-quad (the fourth wave runs Y and Z on accumulators nobody
    // reads: its SIMD would otherwise idle).  Where that branch sits matters more than what it saves: U is issued FIRST,
    // straight behind the k-quad's exit check (two adjacent scalar branches: +1.9 % at the headline, 0.0 % at p = 1000);
    // behind X, Y, Z, or as X / branch / Y Z / branch / U, the off-diagonal tiles lose 0.8-1.5 % at p = 1000, and a
    // separate copy of the k-quad loop for the partial waves makes the allocator shuffle the accumulators between the two
    // (-21 %).  In-box A/B: profiles/r3_product_kernel_round3.txt.  Same blocks, same k order: same bits.
    constexpr bool REBAL = (TI == 2 && TJ == 2 && NW == 4);
    int rbk[TI], cbk[TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) rbk[i] = wr + 16 * i;
#pragma unroll
    for (int j = 0; j < TJ; ++j) cbk[j] = wc + 16 * j;
    const bool skipU = REBAL && I == J;
    int nval = TI * TJ;                                   // accumulators that hold blocks of the tile: X [, Y, Z [, U]]
    if (skipU) {
        nval = 3;
        if (wave == 1) { cbk[0] = wc + 16; cbk[TJ - 1] = wc; }                                   // X=(0,3) Y=(0,2) Z=(1,2)
        else if (wave == 2) { rbk[0] = 16; rbk[TI - 1] = 0; cbk[0] = WN + 16; cbk[TJ - 1] = WN; nval = 1; }   // X=(1,3)
    }
    const bool dead_wave = !REBAL && (I == J) && (wr >= wc + WN);
    const int S = (p + BK - 1) / BK;
#pragma unroll
    for (int q = 0; q < NSTG - 1; ++q)
        if (q < S) issue(q, q);
    // per-lane LDS positions of the wave's fragments in k-quad 0 of a slab image: row (lane >> 4), column XOR-swizzled by the
    // row's parity -- rows of k-quad q are 4q + (lane >> 4), same parity -- so every fragment read of the slab loop is one
    // of these bases plus a COMPILE-TIME offset (stage, operand, k-quad): the loop is unrolled over the NSTG stages to make
    // the stage a constant, and the fragment reads carry no address arithmetic at all (see the note on issue slots above)
    const int fg = lane >> 4, fsw = 16 * (fg & 1);
    const double* pfa[TI];
    const double* pfb[TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) pfa[i] = smem + fg * BM + ((rbk[i] + (lane & 15)) ^ fsw);
#pragma unroll
    for (int j = 0; j < TJ; ++j) pfb[j] = smem + SLAB + fg * BM + ((cbk[j] + (lane & 15)) ^ fsw);
    auto slab = [&](const int s, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        // this wave's DMA of slab s has landed: at most the later slabs' instructions may still be in flight
        const int ahead = min(NSTG - 2, S - 1 - s);
        if (NSTG >= 6 && ahead >= 4) wait_vmcnt<(NSTG >= 6 ? 8 * IPW : 0)>();
        else if (NSTG >= 5 && ahead >= 3) wait_vmcnt<(NSTG >= 5 ? 6 * IPW : 0)>();
        else if (NSTG >= 4 && ahead >= 2) wait_vmcnt<4 * IPW>();
        else if (NSTG >= 3 && ahead >= 1) wait_vmcnt<2 * IPW>();
        else wait_vmcnt<0>();
        const int valid = p - s * BK;                          // k-rows of this slab inside the matrix
        if (valid < BK) {
            // zero the rows beyond the matrix (each wave cleans the rows its own DMA wrote)
            for (int e = lane; e < RPW * BM; e += 64) {
                const int row = wave * RPW + e / BM;
                if (row >= valid) {
                    smem[(size_t)buf * 2 * SLAB + row * BM + (e % BM)] = 0.0;
                    smem[(size_t)buf * 2 * SLAB + SLAB + row * BM + (e % BM)] = 0.0;
                }
            }
        }
        // slab s visible to all, slab s-1 fully consumed.  A raw barrier with an LDS-only wait: __syncthreads() would
        // make hipcc drain vmcnt(0) here, i.e. wait for the slabs prefetched AHEAD as well and collapse the pipeline
        // to one slab of overlap whatever NSTG is (this wave's own slab-s DMA was waited for by the counted vmcnt above)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (ABL != 5) __builtin_amdgcn_s_barrier();                      // ABL 5: no slab barrier (timing ablation)
        asm volatile("" ::: "memory");
        if (s + NSTG - 1 < S) issue(s + NSTG - 1, (buf + NSTG - 1) % NSTG);
        if (!dead_wave) {
#if defined(GGL_EXP_SETPRIO) && GGL_EXP_SETPRIO == 1
            __builtin_amdgcn_s_setprio(1);                    // experiment (tools/r6_e.sh): the matrix segment of a wave outranks the epilogues of its SIMD
#endif
            const int nq = min(BK / 4, (valid + 3) / 4);      // k-quads inside the matrix (last slab: 1 of 4 at p = 500)
            // (The exit check between the k-quads stays even for slabs that lie inside the matrix.  Straight-line code --
            // with the compiler's own order, ds_read2st64 pairs ahead of 8 back-to-back MFMAs, or with a scheduling barrier
            // per k-quad that keeps read 4 / issue 4 -- measured 8-9 % SLOWER at the headline, in-box A/B: the scalar
            // branch is where a wave lets the other waves of its SIMD in.)
#pragma unroll
            for (int kq4 = 0; kq4 < BK / 4; ++kq4) {
                if (kq4 >= nq) break;
                constexpr int so = buf * 2 * SLAB;
                double af[TI], bf[TJ];
#pragma unroll
                for (int i = 0; i < TI; ++i) af[i] = pfa[i][so + kq4 * 4 * BM];
#pragma unroll
                for (int j = 0; j < TJ; ++j) bf[j] = pfb[j][so + kq4 * 4 * BM];
                if constexpr (REBAL && ABL != 3) {
                    if (!skipU) acc[TI - 1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[TI - 1], bf[0], acc[TI - 1][0], 0, 0, 0);
                    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[0], bf[0], acc[0][0], 0, 0, 0);
                    acc[0][TJ - 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[0], bf[TJ - 1], acc[0][TJ - 1], 0, 0, 0);
                    acc[TI - 1][TJ - 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[TI - 1], bf[TJ - 1], acc[TI - 1][TJ - 1], 0, 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < TI; ++i)
#pragma unroll
                        for (int j = 0; j < TJ; ++j) {
                            if (ABL == 3) { acc[i][j][0] += af[i] * bf[j]; continue; }   // timing ablation: no MFMA
                            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
                        }
                }
            }
#if defined(GGL_EXP_SETPRIO) && GGL_EXP_SETPRIO == 1
            __builtin_amdgcn_s_setprio(0);
#endif
        }
    };
    auto stages = [&](const int s0, auto... bufs) {
        (void)std::initializer_list<int>{((s0 + (int)decltype(bufs)::value < S) ? (slab(s0 + (int)decltype(bufs)::value, bufs), 0) : 0)...};
    };
    static_assert(NSTG == 2 || NSTG == 3 || NSTG == 4 || NSTG == 6, "stage counts with an unrolled slab loop");
    for (int s0 = 0; s0 < S; s0 += NSTG) {
        if constexpr (NSTG == 2) stages(s0, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        else if constexpr (NSTG == 3) stages(s0, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
        else if constexpr (NSTG == 4) stages(s0, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{});
        else stages(s0, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{}, std::integral_constant<int, 4>{}, std::integral_constant<int, 5>{});
    }
    __syncthreads();     // all fragment reads done before the slabs are reused as the mirror tile
#if defined(GGL_EXP_SETPRIO) && GGL_EXP_SETPRIO == 2
    __builtin_amdgcn_s_setprio(2);                                       // experiment: the epilogue outranks the other workgroups' slab loops
#endif
    if (ABL == 4) {                                                      // timing ablation: no epilogue
        double keepalive = 0.0;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) keepalive += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (keepalive == 1234.5678) C[0] = keepalive;
        return;
    }

    const double cI = coef[k * NS_NCOEF + 0], cAcc = coef[k * NS_NCOEF + 1], cE = coef[k * NS_NCOEF + 2];
    const double dI = coef[k * NS_NCOEF + 3], dC = coef[k * NS_NCOEF + 4], dE = coef[k * NS_NCOEF + 5];
    double* Ck = (second ? C1 : C) + (size_t)kk * pp;
    double* C2k = (C2 && !second) ? C2 + (size_t)kk * pp : nullptr;
    const double* Ek = (E && !second) ? E + (size_t)kk * pp : nullptr;
    double dev = 0.0;
    // Off-diagonal tiles: the accumulators go to the (XOR-swizzled) LDS tile first and BOTH copies -- the tile and its
    // mirror -- are written from there with 16-byte accesses, two consecutive columns per lane (p is even, tile columns start
    // on even indices): half the global store / E-load instructions of the layout the MFMA leaves the values in (one column
    // per lane: 8-byte accesses, which run at 0.54-0.70x the 16-byte rate on this chip).  Same arithmetic per element
    // (cAcc * acc, then + cE * E): same bits.  Diagonal tiles keep the direct path (upper triangle only, in-tile mirror).
    const bool wide = (I != J) && ABL != 1;
    if (wide) {
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
            for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wr + ti * 16 + (lane >> 4) + 4 * r;
                    const int col = wc + tj * 16 + (lane & 15);
                    smem[row * BM + (col ^ row)] = cAcc * acc[ti][tj][r];
                }
        __syncthreads();
        auto ld2 = [](const double* q) { return *reinterpret_cast<const double2*>(q); };
        auto lde2 = [&](const double* q) {
            if constexpr (AUX == 0) return *reinterpret_cast<const double2*>(q);
            else { double2 e; e.x = ld_pol<AUX>(q); e.y = ld_pol<AUX>(q + 1); return e; }
        };
        for (int e2 = tid; e2 < BM * BM / 2; e2 += NT) {
            const int row = e2 / (BM / 2), c2 = (e2 % (BM / 2)) * 2;
            const int gi = I0 + row, gj = J0 + c2;
            // columns c2, c2 + 1 of row `row` sit at (c2 ^ row), (c2 ^ row) ^ 1: one aligned pair, swapped for odd rows
            double2 t = ld2(smem + row * BM + ((c2 ^ row) & ~1));
            if (row & 1) { const double h = t.x; t.x = t.y; t.y = h; }
            if (gi < p && gj + 1 < p) {
                double2 e = {0.0, 0.0};
                if (Ek) e = lde2(Ek + (size_t)gi * p + gj);
                double2 v;
                v.x = t.x + cE * e.x;
                v.y = t.y + cE * e.y;
                dev = fmax(dev, fmax(fabs(v.x), fabs(v.y)));
                *reinterpret_cast<double2*>(Ck + (size_t)gi * p + gj) = v;
                if (C2k) {
                    double2 w;
                    w.x = c2val(dC, v.x, dE, e.x);
                    w.y = c2val(dC, v.y, dE, e.y);
                    *reinterpret_cast<double2*>(C2k + (size_t)gi * p + gj) = w;
                }
                if (rowpart) {          // the bound partials (and then the mirror) read the FINAL values from the tile
                    if (row & 1) { const double h = v.x; v.x = v.y; v.y = h; }
                    *reinterpret_cast<double2*>(smem + row * BM + ((c2 ^ row) & ~1)) = v;
                }
            } else if (gi < p && gj < p) {
                // odd p: the last column is the first half of a pair whose second half lies outside the matrix
                const double e0 = Ek ? ld_pol<AUX>(Ek + (size_t)gi * p + gj) : 0.0;
                const double v0 = t.x + cE * e0;
                dev = fmax(dev, fabs(v0));
                Ck[(size_t)gi * p + gj] = v0;
                if (C2k) C2k[(size_t)gi * p + gj] = c2val(dC, v0, dE, e0);
                if (rowpart) {
                    double2 v = {v0, 0.0};
                    if (row & 1) { v.x = 0.0; v.y = v0; }
                    *reinterpret_cast<double2*>(smem + row * BM + ((c2 ^ row) & ~1)) = v;
                }
            } else if (rowpart) {
                *reinterpret_cast<double2*>(smem + row * BM + ((c2 ^ row) & ~1)) = double2{0.0, 0.0};
            }
        }
    } else {
    if (REBAL && I == J && rowpart && ABL != 1) {
        // the staged tile of the bound partials wants zeros below the diagonal: the six blocks no wave owns any more
        for (int e = tid; e < 6 * 256; e += NT) {
            const int blk = e >> 8, br = (blk < 1) ? 1 : (blk < 3 ? 2 : 3), bc = blk - (br == 1 ? 0 : (br == 2 ? 1 : 3));
            const int row = br * 16 + ((e >> 4) & 15), col = bc * 16 + (e & 15);
            smem[row * BM + (col ^ row)] = 0.0;
        }
    }
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj) {
            if (REBAL && ((ti != tj && tj == 0 && nval <= 3) || ((ti | tj) != 0 && nval <= 1))) continue;   // not a block of the tile
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rbk[ti] + (lane >> 4) + 4 * r;
                const int col = cbk[tj] + (lane & 15);
                const int gi = I0 + row, gj = J0 + col;
                double v = cAcc * acc[ti][tj][r];
                const bool keep = gi < p && gj < p && (I != J || gi <= gj);
                if (keep) {
                    if (gi == gj) v += cI;
                    const double e0 = Ek ? ld_pol<AUX>(Ek + (size_t)gi * p + gj) : 0.0;
                    v += cE * e0;
                    dev = fmax(dev, fabs(v - (gi == gj ? 1.0 : 0.0)));
                    Ck[(size_t)gi * p + gj] = v;
                    if (C2k) C2k[(size_t)gi * p + gj] = c2val(dC, v, dE, e0) + (gi == gj ? dI : 0.0);
                    if (I == J && gi != gj) {
                        Ck[(size_t)gj * p + gi] = v;
                        if (C2k) C2k[(size_t)gj * p + gi] = c2val(dC, v, dE, e0);
                    }
                }
                // XOR-swizzled BM x BM tile in LDS: the mirror pass of an off-diagonal tile reads it transposed; with
                // bound partials every tile is staged (entries that are not stored count as zero)
                if ((I != J || rowpart) && ABL != 1) smem[row * BM + (col ^ row)] = (rowpart && !keep) ? 0.0 : v;
            }
        }
    }
    if (maxdev) {
        dev = wave_max(dev);
        if (lane == 0 && dev > 0.0)
            atomicMax(reinterpret_cast<unsigned long long*>(maxdev + k), (unsigned long long)__double_as_longlong(dev));
    }
    if (((I != J && !wide) || rowpart) && ABL != 1) __syncthreads();
    if (rowpart && ABL != 1 && tid < 2 * BM) {
        // threads 0..BM-1: row t of the staged tile; threads BM..2BM-1 (off-diagonal tiles): column t of it, i.e. row
        // J0+t of the mirrored tile.  A diagonal tile holds its upper triangle U only: row t of the full symmetric tile
        // is row t of U plus column t of U minus the diagonal entry.
        const int t = tid & (BM - 1);
        const bool second = tid >= BM;
        const size_t slot_stride = (size_t)p;
        double rs = 0.0, sq = 0.0;
        if (!second) {
#pragma unroll 8
            for (int c = 0; c < BM; ++c) { const double x = smem[t * BM + (c ^ t)]; rs += fabs(x); sq += x * x; }
        }
        if (second || I == J) {
            double cs = 0.0;
#pragma unroll 8
            for (int r = 0; r < BM; ++r) cs += fabs(smem[r * BM + (t ^ r)]);
            if (I == J) {
                if (!second) {
                    const double dg = smem[t * BM];
                    rs = (rs + cs) - fabs(dg);
                    sq = 2.0 * sq - dg * dg;
                }
            } else {
                rs = cs;
            }
        } else {
            sq *= 2.0;            // off-diagonal tile: its mirror has the same squares
        }
        if (I != J || !second) {
            const int T2 = (p + BM - 1) / BM;
            const int slot = second ? I : J, gi = (second ? J0 : I0) + t;
            if (gi < p) rowpart[((size_t)k * T2 + slot) * slot_stride + gi] = rs;
        }
        if (tid < 64) {                      // wave 0 holds the BM row threads
            sq = wave_sum((tid < BM) ? sq : 0.0);
            if (tid == 0) fropart[(size_t)k * (T * (T + 1) / 2) + blockTile] = sq;
        }
    }
    if (wide) {
        // the mirror: row a of it is column a of the staged tile; two consecutive columns c2, c2 + 1 per lane.  Without bound
        // partials the tile still holds cAcc * acc and the E term is added here as well (E is bitwise symmetric -- an output
        // of this kernel family -- so its mirrored entries ARE the upper ones); with them it holds the final values.
        auto lde2 = [&](const double* q) {
            if constexpr (AUX == 0) return *reinterpret_cast<const double2*>(q);
            else { double2 e; e.x = ld_pol<AUX>(q); e.y = ld_pol<AUX>(q + 1); return e; }
        };
        for (int e2 = tid; e2 < BM * BM / 2; e2 += NT) {
            const int a = e2 / (BM / 2), c2 = (e2 % (BM / 2)) * 2;      // out[J0+a][I0+c2 .. +1] = tile[c2 .. +1][a]
            if (J0 + a < p && I0 + c2 + 1 < p) {
                double2 v;
                v.x = smem[c2 * BM + (a ^ c2)];
                v.y = smem[(c2 + 1) * BM + (a ^ (c2 + 1))];
                double2 e = {0.0, 0.0};
                const bool need_e = Ek && (!rowpart || (C2k && dE != 0.0));
                if (need_e) e = lde2(Ek + (size_t)(J0 + a) * p + I0 + c2);
                if (!rowpart) { v.x += cE * e.x; v.y += cE * e.y; }
                *reinterpret_cast<double2*>(Ck + (size_t)(J0 + a) * p + I0 + c2) = v;
                if (C2k) {
                    double2 w;
                    w.x = c2val(dC, v.x, dE, e.x);
                    w.y = c2val(dC, v.y, dE, e.y);
                    *reinterpret_cast<double2*>(C2k + (size_t)(J0 + a) * p + I0 + c2) = w;
                }
            } else if (J0 + a < p && I0 + c2 < p) {
                // (never taken: I < J, so the rows I0 .. I0 + BM - 1 of an off-diagonal tile all lie inside the matrix and an
                // odd last column can only belong to tile column J; kept so that the pair test above stands on its own)
                double v0 = smem[c2 * BM + (a ^ c2)];
                const bool need_e = Ek && (!rowpart || (C2k && dE != 0.0));
                const double e0 = need_e ? ld_pol<AUX>(Ek + (size_t)(J0 + a) * p + I0 + c2) : 0.0;
                if (!rowpart) v0 += cE * e0;
                Ck[(size_t)(J0 + a) * p + I0 + c2] = v0;
                if (C2k) C2k[(size_t)(J0 + a) * p + I0 + c2] = c2val(dC, v0, dE, e0);
            }
        }
    }
}

// The bound validation of a speculative Omega-step as extra workgroups of a product launch (CwRider, kernels.hpp): workgroup
// idx of the rider takes rows 16 bx .. 16 bx + 15 of instance k.  Arithmetic and order are those of k_bound_rows (row sums in
// tile-column order, their maximum) and k_cw_final (lane-strided products, wave sums, memory-side atomic max and arrival, the
// Frobenius shares added in tile order by the last workgroup of the instance) -- the bound and the vector left for the next
// iteration are the same bits.  Every workgroup adds the row sums of its instance itself (T p loads out of L2; the only thing
// it needs of them is their maximum, the scale of the next vector): hidden work on CUs the product leaves idle.
__device__ __forceinline__ void cw_rider_body(const CwRider& r, int idx, double* sh)
{
    const int k = idx / r.nbx, bx = idx - k * r.nbx;
    if (k >= r.K) return;
    const int p = r.p;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double mxd = 0.0;
    for (int j = threadIdx.x; j < p; j += 256) {
        const double* rp = r.rowpart + (size_t)k * r.T * p + j;
        double v = 0.0;
        for (int s2 = 0; s2 < r.T; ++s2) v += rp[(size_t)s2 * p];
        mxd = fmax(mxd, v);
        if (bx == 0 && r.d_out) r.d_out[(size_t)k * p + j] = v;
    }
    mxd = wave_max(mxd);
    if (lane == 0) sh[wave] = mxd;
    __syncthreads();
    const double inf0 = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
    __syncthreads();
    const double scale = 1.0 / inf0;
    const double* dk = r.dprev + (size_t)k * p;
    const int r0 = bx * 16 + wave * 4;
    const double* w = r.B + (size_t)k * p * p;
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
            mx = fmax(mx, y / dk[r0 + q]);
            if (r.dnext && lane == 0) {
                const double v = y * scale;
                r.dnext[(size_t)k * p + r0 + q] = (v > 0.0 && isfinite(v)) ? v : 1.0;
            }
        }
    }
    if (lane == 0) sh[wave] = mx;
    __syncthreads();
    int* last = reinterpret_cast<int*>(sh + 4);
    if (threadIdx.x == 0) {
        mx = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
        const unsigned long long old = __hip_atomic_fetch_max(r.cwmax + k, (unsigned long long)__double_as_longlong(mx),
                                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned arrive = 1u + (unsigned)(old >> 63);
        *last = __hip_atomic_fetch_add(r.cnt + k, arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)r.nbx - 1;
    }
    __syncthreads();
    if (!*last) return;
    double cw = 0.0;
    if (threadIdx.x == 0) {
        cw = __longlong_as_double((long long)__hip_atomic_exchange(r.cwmax + k, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        __hip_atomic_store(r.cnt + k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    double sq = 0.0;
    double* shfro = sh + 8;
    for (int t0 = 0; t0 < r.ntile; t0 += 256) {
        const int t = t0 + (int)threadIdx.x;
        __syncthreads();
        shfro[threadIdx.x] = (t < r.ntile) ? r.fropart[(size_t)k * r.ntile + t] : 0.0;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int m = min(256, r.ntile - t0);
            for (int q = 0; q < m; ++q) sq += shfro[q];
        }
    }
    if (threadIdx.x != 0) return;
    double inf = inf0;
    const double fr = sqrt(sq);
    if (isfinite(cw) && cw > 0.0) { const double wv = cw * (1.0 + 1e-12); inf = (wv < inf) ? wv : inf; }
    const double b = sqrt((fr < inf) ? fr : inf);
    r.out[k] = b;
    if (r.flag && !(b <= r.cuse[k])) {
        atomicOr(r.flag + r.flag_slot, 1);
        r.flag_host[r.flag_slot] = 1;
    }
}

// the norm reduction as a rider (RedRider): k_reduce_partials' single row -- four rows per trip, strided per-thread sums in
// ascending order, wave sums, the four waves added in a fixed order -- then the sums and, behind them, the sequence number as
// system-scope stores.  No release FENCE here: inside a product launch it would write back an XCD's L2 (the tiles' output).
// What orders the two instead: the sums are system-scope (write-through, uncached) stores of ONE wave, `s_waitcnt vmcnt(0)`
// returns when they have been acknowledged, the barrier puts thread 0's store of the sequence number behind that -- and the
// row carries the sequence number a second time, in the same store instruction as the sums (slot nv), which the host checks
// after the poll (finish_norms): a sequence number that overtook its sums would be seen without its stamp.
__device__ __forceinline__ void red_rider_body(const RedRider& r, double* smem)
{
    double (*sh)[8] = reinterpret_cast<double (*)[8]>(smem);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nblk = r.nblk, nv = r.nv;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const double* base = r.partials;
    // (two rows per trip where k_reduce_partials takes four -- the same rows per thread in the same ascending order, hence the
    // same sums -- so that this branch stays inside the product kernel's 81 registers: with four the 32x32 kernel went to 108
    // and from five to four workgroups per CU)
    for (int b = threadIdx.x; b < nblk; b += 512) {
        double t[2][8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int bb = b + 256 * u;
#pragma unroll
            for (int v = 0; v < 8; ++v) t[u][v] = (v < nv && bb < nblk) ? base[(size_t)bb * nv + v] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
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
    if ((int)threadIdx.x <= nv) {
        const double s = (int)threadIdx.x == nv ? (double)r.seq_val
                                                : (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
        __hip_atomic_store(r.out + threadIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(r.seq, r.seq_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// a rider nobody took (symm_flush_rider)
__global__ __launch_bounds__(256) void k_cw_rider(const CwRider rider)
{
    __shared__ __attribute__((aligned(16))) double sh[8 + 256];
    cw_rider_body(rider, (int)blockIdx.x, sh);
}

template <int BK, int NSTG, int ABL = 0, int BM = 64, int NW = 4>
__global__ __launch_bounds__(NW * 64) void k_symm_dl(const double* __restrict__ A, const double* __restrict__ B,
                                                 double* __restrict__ C, double* __restrict__ C2,
                                                 const double* __restrict__ E, const double* __restrict__ coef, int K,
                                                 int p, const double* __restrict__ A1, const double* __restrict__ B1,
                                                 double* __restrict__ C1, int K1, double* __restrict__ maxdev,
                                                 double* __restrict__ rowpart, double* __restrict__ fropart, const CwRider rider,
                                                 const CopySegs cps, const RedRider red)
{
    __shared__ __attribute__((aligned(16))) double smem[NSTG * 2 * BK * BM];
    const int T = (p + BM - 1) / BM;
    int bid = (int)blockIdx.x;
    if constexpr (NW == 4) {
        // The norm reduction the host is waiting for rides in FRONT of the product's workgroups (dispatched first: behind them
        // its result reached the host ~10 us after the Theta kernel had ended, in front ~5); NXCD slots, one of them working,
        // so that the tiles keep their XCDs.
        if (red.nblk > 0) {
            if (bid < NXCD) {
                if (bid == 0) red_rider_body(red, smem);
                return;
            }
            bid -= NXCD;
        }
        // workgroups behind the product's own: the bound validation of the previous launch's output (CwRider), table transfers
        const int base = xcd_grid(T * (T + 1) / 2, K + K1);
        if (bid >= base) {
            const int idx = bid - base, ncw = rider.K * rider.nbx;
            if (idx < ncw) cw_rider_body(rider, idx, smem);
            else {
                // (k_copy_small: words from the pinned tables, or zeros; 1024 words per workgroup, so that no thread waits for
                // more than four reads over PCIe one after the other -- one workgroup per segment took 45 us for the 30 KB of a
                // K = 64 schedule)
                int blk = idx - ncw;
                for (int sgi = 0; sgi < cps.n; ++sgi) {
                    const int nb = (int)((cps.words[sgi] + 1023u) / 1024u);
                    if (blk < nb) {
                        unsigned* d = (unsigned*)cps.dst[sgi];
                        const unsigned* src = (const unsigned*)cps.src[sgi];
                        const unsigned i0 = (unsigned)blk * 1024u + threadIdx.x;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const unsigned i = i0 + 256u * e;
                            if (i < cps.words[sgi]) d[i] = src ? src[i] : 0u;
                        }
                        break;
                    }
                    blk -= nb;
                }
            }
            return;
        }
    }
    int k, b;
    if (!decode_block_xcd(T * (T + 1) / 2, K + K1, k, b, bid)) return;
    symm_dl_tile<BK, NSTG, ABL, BM, NW>(A, B, C, C2, E, coef, K, p, A1, B1, C1, K1, maxdev, rowpart, fropart, k, b, smem);
}

#ifdef GGL_DEV
// ---------------------------------------------------------------------------------------------
// k_symm_sk: the product for batches that leave the chip UNDER-FILLED (the K-sharded slabs K = 2..8 at p = 500, (20,200)):
// 32x32 output tiles, and the four waves of a workgroup split the k-range of ONE tile instead of its area.
//
// MEASURED, NOT SHIPPED (dev build only; profiles/r3_small_batch_split_k.txt): as fast as the 32x32 direct-to-LDS kernel,
// not faster, and slower with the k-range split over workgroups -- the under-filled launches are bound by fixed latencies
// (launch, first DMA, epilogue: ~6.5 us) plus whole tiles per CU, not by what a wave does per k-quad.
// The 32x32 direct-to-LDS kernel gives each wave one 16x16 block: per k-quad two fragment reads, an LDS round trip and ONE
// MFMA, a workgroup barrier per slab.  Here every wave owns the whole tile (2x2 blocks: two + two
// fragment reads feed FOUR independent MFMAs) over the k-slabs s = wave, wave+4, wave+8, ...; it streams ITS slabs into
// ITS quarter of the LDS through its own DMA pipeline (NSTG stages) and waits for nobody -- no barrier in the main loop at
// all.  The four partial tiles meet in LDS at the end (one barrier), are summed in a fixed order (wave 0..3:
// deterministic), and the epilogue (affine terms, second output, mirror, bound partials) runs on the sums.
// Diagonal tiles take the upper triangle's sums for both halves: the output is symmetric bit for bit.
// ---------------------------------------------------------------------------------------------
//
// SPLIT > 1: the k-range of a tile is dealt over SPLIT workgroups as well (slab s goes to workgroup (s / 4) % SPLIT, wave
// s % 4) -- 544 tiles on 256 CUs leave some CUs three tiles and the launch ends when those do; 2176 quarter-tiles pack to
// within 6 %.  Each workgroup leaves the sum of its four waves in a global scratch tile (write-through stores), then one
// memory-side atomic per workgroup counts the arrivals; the LAST one adds the SPLIT scratch tiles in part order (so the
// result does not depend on who was last) and runs the epilogue.  No cache write-back / invalidate fences anywhere
// (see k_cw_final): stores, loads and the counter are agent-scope atomics.  The counter is left at zero.
template <int SPLIT> __device__ __forceinline__ bool sk_decode(int ntiles, int K, int L, int& k, int& tile, int& part)
{
    // as decode_block_xcd, over ntiles * SPLIT units per instance with the SPLIT parts of a tile in consecutive slots
    auto small = [&](int Ks, int Ls) {
        const int g = xcd_share(Ks);
        if (g) {
            const int xcd = Ls % NXCD, slot = Ls / NXCD;
            k = xcd / g;
            tile = (slot / SPLIT) * g + xcd % g;
            part = slot % SPLIT;
            return tile < ntiles;
        }
        const int u = Ls % (ntiles * SPLIT);
        k = Ls / (ntiles * SPLIT);
        tile = u / SPLIT;
        part = u % SPLIT;
        return k < Ks;
    };
    if (K < NXCD) return small(K, L);
    const int nfull = NXCD * (K / NXCD) * ntiles * SPLIT;
    if (L < nfull) {
        const int xcd = L % NXCD, slot = L / NXCD, u = slot % (ntiles * SPLIT);
        k = (slot / (ntiles * SPLIT)) * NXCD + xcd;
        tile = u / SPLIT;
        part = u % SPLIT;
        return true;
    }
    const bool ok = small(K % NXCD, L - nfull);
    k += (K / NXCD) * NXCD;
    return ok;
}
inline int sk_grid_small(int ntiles, int K, int split)
{
    const int g = xcd_share(K);
    return g ? NXCD * ((ntiles + g - 1) / g) * split : NXCD * ((ntiles * split * K + NXCD - 1) / NXCD);
}
inline int sk_grid(int ntiles, int K, int split)
{
    if (K < NXCD) return sk_grid_small(ntiles, K, split);
    const int r = K % NXCD;
    return NXCD * (K / NXCD) * ntiles * split + (r ? sk_grid_small(ntiles, r, split) : 0);
}

template <int BK, int NSTG, int SPLIT>
__global__ __launch_bounds__(256) void k_symm_sk(const double* __restrict__ A, const double* __restrict__ B,
                                                 double* __restrict__ C, double* __restrict__ C2,
                                                 const double* __restrict__ E, const double* __restrict__ coef, int K,
                                                 int p, double* __restrict__ maxdev, double* __restrict__ rowpart,
                                                 double* __restrict__ fropart, double* __restrict__ scratch,
                                                 unsigned* __restrict__ arrivals)
{
    constexpr int BM = 32;
    constexpr int SLAB = BK * BM;                       // doubles per operand slab
    constexpr int IPS = BK / 4;                         // DMA instructions per operand and slab: 4 rows x 32 doubles = 1 KiB each
    constexpr int WREG = NSTG * 2 * SLAB;               // doubles of LDS per wave: [stage][A|B][BK][32]; later its partial tile
    static_assert(BK % 4 == 0 && IPS >= 1, "slab depth");
    static_assert(WREG >= BM * BM, "a wave's region takes its partial tile");
    static_assert(NSTG >= 2 && NSTG <= 4, "stages");
    __shared__ __attribute__((aligned(16))) double smem[4 * WREG];
    __shared__ int last_s;
    const int T = (p + BM - 1) / BM;
    const int ntiles = T * (T + 1) / 2;
    int k, b, part = 0;
    if constexpr (SPLIT > 1) { if (!sk_decode<SPLIT>(ntiles, K, (int)blockIdx.x, k, b, part)) return; }
    else { if (!decode_block_xcd(ntiles, K, k, b)) return; }
    const int blockTile = b;
    int I = 0;
    while (b >= T - I) { b -= T - I; ++I; }
    const int J = I + b;
    const int I0 = I * BM, J0 = J * BM;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t pp = (size_t)p * p;
    const double* Ak = A + (size_t)k * pp;
    const double* Bk = B + (size_t)k * pp;
    double* wbase = smem + (size_t)wave * WREG;
    const int sfirst = 4 * part + wave, sstep = 4 * SPLIT;          // this wave's slabs: sfirst, sfirst + sstep, ...

    v4d acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

    // DMA geometry (as k_symm_dl at BM = 32): lane l -> slab row 4i + (l >> 4), LDS position (l & 15) * 2, source column
    // position ^ 16 * (row & 1) -- the XOR keeps the fragment reads of a k-quad (four consecutive rows) off each other's banks
    const unsigned pu = (unsigned)p, pm1 = (unsigned)(p - 1), pm2 = (unsigned)(p - 2);
    const int lrow = lane >> 4;
    const unsigned cpos = (unsigned)((lane & 15) * 2) ^ (unsigned)(16 * (lrow & 1));
    const unsigned ca = min((unsigned)I0 + cpos, pm2), cb = min((unsigned)J0 + cpos, pm2);
    const int S = (p + BK - 1) / BK;                    // slabs of the whole k-range
    const int nw = (S > sfirst) ? (S - sfirst + sstep - 1) / sstep : 0;
    auto issue = [&](int j, int stage) {
        const int m0 = (sfirst + sstep * j) * BK;
        double* base = wbase + (size_t)stage * 2 * SLAB;
#pragma unroll
        for (int i = 0; i < IPS; ++i) {
            const unsigned ro = min((unsigned)(m0 + 4 * i + lrow), pm1) * pu;
            __builtin_amdgcn_global_load_lds((gptr_t)(Ak + ro + ca), (lptr_t)(base + i * 128), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(Bk + ro + cb), (lptr_t)(base + SLAB + i * 128), 16, 0, 0);
        }
    };
#pragma unroll
    for (int q = 0; q < NSTG - 1; ++q)
        if (q < nw) issue(q, q);
    const int fg = lane >> 4, fsw = 16 * (fg & 1);
    int fa[2], fb[2];                                   // fragment positions in k-quad 0 of a stage image
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        fa[i] = fg * BM + ((i * 16 + (lane & 15)) ^ fsw);
        fb[i] = SLAB + fg * BM + ((i * 16 + (lane & 15)) ^ fsw);
    }
    int stage = 0;
    for (int j = 0; j < nw; ++j) {
        // this wave's DMA of its slab j has landed: at most the slabs issued after it may still be in flight
        const int ahead = min(NSTG - 2, nw - 1 - j);
        if (NSTG >= 4 && ahead >= 2) wait_vmcnt<4 * IPS>();
        else if (NSTG >= 3 && ahead >= 1) wait_vmcnt<2 * IPS>();
        else wait_vmcnt<0>();
        const double* st = wbase + (size_t)stage * 2 * SLAB;
        const int valid = p - (sfirst + sstep * j) * BK;      // k-rows of this slab inside the matrix
        if (valid < BK) {
            double* stw = wbase + (size_t)stage * 2 * SLAB;
            for (int e = lane; e < BK * BM; e += 64)
                if (e / BM >= valid) { stw[e] = 0.0; stw[SLAB + e] = 0.0; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        // the stage slab j-1 was read from is free: its fragment reads completed before the MFMAs that consumed them issued
        if (j + NSTG - 1 < nw) issue(j + NSTG - 1, (stage + NSTG - 1) % NSTG);
        const int nq = min(BK / 4, (valid + 3) / 4);
#pragma unroll
        for (int kq = 0; kq < BK / 4; ++kq) {
            if (kq >= nq) break;
            const double a0 = st[fa[0] + kq * 4 * BM], a1 = st[fa[1] + kq * 4 * BM];
            const double b0 = st[fb[0] + kq * 4 * BM], b1 = st[fb[1] + kq * 4 * BM];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        stage = (stage + 1 == NSTG) ? 0 : stage + 1;
    }
    // the wave's partial tile over its own slabs, plain [row][col]
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                wbase[(ti * 16 + (lane >> 4) + 4 * r) * BM + tj * 16 + (lane & 15)] = acc[ti][tj][r];
    __syncthreads();

    const double cI = coef[k * NS_NCOEF + 0], cAcc = coef[k * NS_NCOEF + 1], cE = coef[k * NS_NCOEF + 2];
    const double dI = coef[k * NS_NCOEF + 3], dC = coef[k * NS_NCOEF + 4], dE = coef[k * NS_NCOEF + 5];
    double* Ck = C + (size_t)k * pp;
    double* C2k = C2 ? C2 + (size_t)k * pp : nullptr;
    const double* Ek = E ? E + (size_t)k * pp : nullptr;
    // thread -> row (tid >> 3), columns c0 .. c0 + 3 of the tile; the four partials are added in wave order
    const int row = tid >> 3, c0 = (tid & 7) * 4;
    const int gi = I0 + row;
    double sum[4];
    if constexpr (SPLIT > 1) {
        // this workgroup's share of the tile -> scratch; the last workgroup of the tile to arrive carries on with the total
        double* mine = scratch + (((size_t)k * ntiles + blockTile) * SPLIT + part) * (BM * BM);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = row * BM + c0 + i;
            double t = smem[o];
#pragma unroll
            for (int w = 1; w < 4; ++w) t += smem[(size_t)w * WREG + o];
            __hip_atomic_store(mine + o, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            unsigned* cell = arrivals + (size_t)k * ntiles + blockTile;
            const unsigned old = __hip_atomic_fetch_add(cell, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_s = (old == SPLIT - 1);
            if (old == SPLIT - 1) __hip_atomic_store(cell, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!last_s) return;
        const double* all = scratch + ((size_t)k * ntiles + blockTile) * SPLIT * (BM * BM);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = c0 + i, o = (I != J || row <= c) ? row * BM + c : c * BM + row;
            double t = __hip_atomic_load(all + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int g = 1; g < SPLIT; ++g)
                t += __hip_atomic_load(all + (size_t)g * (BM * BM) + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sum[i] = t;
        }
    } else if (I != J) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double2 t = *reinterpret_cast<const double2*>(smem + row * BM + c0 + 2 * h);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const double2 u = *reinterpret_cast<const double2*>(smem + (size_t)w * WREG + row * BM + c0 + 2 * h);
                t.x += u.x;
                t.y += u.y;
            }
            sum[2 * h] = t.x;
            sum[2 * h + 1] = t.y;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = c0 + i, o = (row <= c) ? row * BM + c : c * BM + row;       // the upper triangle's sums, both ways
            double t = smem[o];
#pragma unroll
            for (int w = 1; w < 4; ++w) t += smem[(size_t)w * WREG + o];
            sum[i] = t;
        }
    }
    double dev = 0.0, val[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int gj = J0 + c0 + 2 * h;
        val[2 * h] = 0.0;
        val[2 * h + 1] = 0.0;
        if (gi < p && gj < p) {                         // p is even and gj is: the pair is inside or outside together
            double2 e = {0.0, 0.0};
            if (Ek) e = *reinterpret_cast<const double2*>(Ek + (size_t)gi * p + gj);
            double2 v;
            v.x = cAcc * sum[2 * h];
            v.y = cAcc * sum[2 * h + 1];
            if (gi == gj) v.x += cI;
            if (gi == gj + 1) v.y += cI;
            v.x += cE * e.x;
            v.y += cE * e.y;
            dev = fmax(dev, fmax(fabs(v.x - (gi == gj ? 1.0 : 0.0)), fabs(v.y - (gi == gj + 1 ? 1.0 : 0.0))));
            *reinterpret_cast<double2*>(Ck + (size_t)gi * p + gj) = v;
            if (C2k) {
                double2 w2;
                w2.x = c2val(dC, v.x, dE, e.x) + (gi == gj ? dI : 0.0);
                w2.y = c2val(dC, v.y, dE, e.y) + (gi == gj + 1 ? dI : 0.0);
                *reinterpret_cast<double2*>(C2k + (size_t)gi * p + gj) = w2;
            }
            val[2 * h] = v.x;
            val[2 * h + 1] = v.y;
        }
    }
    if (maxdev) {
        dev = wave_max(dev);
        if (lane == 0 && dev > 0.0)
            atomicMax(reinterpret_cast<unsigned long long*>(maxdev + k), (unsigned long long)__double_as_longlong(dev));
    }
    if (I == J && !rowpart) return;
    // the final tile (zeros outside the matrix) for the mirror and the bound partials: over wave 0's partial, once everybody
    // has read the partials
    __syncthreads();
    *reinterpret_cast<double2*>(smem + row * BM + c0) = double2{val[0], val[1]};
    *reinterpret_cast<double2*>(smem + row * BM + c0 + 2) = double2{val[2], val[3]};
    __syncthreads();
    if (I != J) {
        // mirror: row a of it is column a of the tile; E is bitwise symmetric (an output of this kernel family)
        const int a = row, gm = J0 + a;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = c0 + 2 * h, gc = I0 + c;
            if (gm < p && gc < p) {
                double2 v;
                v.x = smem[c * BM + a];
                v.y = smem[(c + 1) * BM + a];
                *reinterpret_cast<double2*>(Ck + (size_t)gm * p + gc) = v;
                if (C2k) {
                    double2 e = {0.0, 0.0};
                    if (Ek && dE != 0.0) e = *reinterpret_cast<const double2*>(Ek + (size_t)gm * p + gc);
                    double2 w2;
                    w2.x = c2val(dC, v.x, dE, e.x);
                    w2.y = c2val(dC, v.y, dE, e.y);
                    *reinterpret_cast<double2*>(C2k + (size_t)gm * p + gc) = w2;
                }
            }
        }
    }
    if (rowpart && tid < 2 * BM) {
        // threads 0..31: row t of the tile; threads 32..63 (off-diagonal tiles): column t, i.e. row J0 + t of the mirror.
        // A diagonal tile is complete (both triangles) here.
        const int t = tid & (BM - 1);
        const bool second = tid >= BM;
        double rs = 0.0, sq = 0.0;
        if (!second) {
#pragma unroll 8
            for (int c = 0; c < BM; ++c) { const double x = smem[t * BM + ((c + t) & (BM - 1))]; rs += fabs(x); sq += x * x; }
        } else if (I != J) {
#pragma unroll 8
            for (int r = 0; r < BM; ++r) rs += fabs(smem[((r + t) & (BM - 1)) * BM + t]);
        }
        if (I != J) sq *= 2.0;                          // the mirror has the same squares
        if (I != J || !second) {
            const int slot = second ? I : J, gr = (second ? J0 : I0) + t;
            if (gr < p) rowpart[((size_t)k * T + slot) * (size_t)p + gr] = rs;
        }
        sq = wave_sum(second ? 0.0 : sq);               // tid < 64: wave 0
        if (tid == 0) fropart[(size_t)k * (T * (T + 1) / 2) + blockTile] = sq;
    }
}

#endif   // GGL_DEV (k_symm_sk)

#ifdef GGL_DEV      // (round 6: the persistent chain is an option of the development library only -- measured slower)
// ---------------------------------------------------------------------------------------------
// k_omega_chain: the WHOLE product chain of an Omega-step (A', B', the Newton-Schulz products, Omega) of a batch in ONE
// persistent launch, with the dependencies kept where they are: per INSTANCE.  Product s+1 of instance k needs product s of
// instance k and nothing else, yet a launch boundary makes every instance wait for the slowest tile of all of them, 7 times
// per iteration (a K = 32, p = 500 launch keeps the matrix pipes ~52 % busy, two concurrent half-batch launch sequences
// ~57 %).  Here the workgroups are persistent (as many as are resident: 3 per CU with 48 KiB of LDS), each serves the
// instances of the XCD it REALLY runs on (HW_REG_XCC_ID; instance k lives on XCD k % 8, so its operands stay in one L2 and
// every hand-off is XCD-local), and pulls tiles through two counters per instance:
//     next[k]  tickets handed out (ticket t = product s, tile t - begin[s])      done[k]  tiles completed
// A ticket of product s is handed out only when done[k] >= begin[s], i.e. all tiles of the products before it are complete.
// While one instance drains the last tiles of a product, the workgroups run the other instances of the XCD: fixed priority
// by instance index, so the instances drift apart instead of marching in lockstep.
// Visibility (MI355X_MICROARCH.md, inter-workgroup visibility): producer = plain stores (they stay in the XCD's L2) ->
// every wave s_waitcnt vmcnt(0) -> barrier -> one relaxed agent-scope fetch_add on done[k]; consumer = relaxed agent-scope
// (sc1) load of done[k] -> every global READ of the tile with sc1 (L2-served, never this CU's L1).  No L2 write-back and no
// L1 invalidate anywhere: producer and consumer share one L2 by construction (a workgroup only serves its own XCD's list).
// Should the hardware ever place no workgroup on an XCD, that XCD's instances stay unfinished: k_chain_check raises the
// step's validation flag and the host repeats the step on the launch-per-product path (correct either way).
// ---------------------------------------------------------------------------------------------
// state of instance k: one 128-byte line of CHAIN_CNT_STRIDE words -- [0] tiles completed (all products), [1 + s] tickets
// handed out of product s.  A claim is ONE fetch_add on the ticket word of the product the workgroup last saw ready for
// that instance (a ticket beyond the product's tiles is void: nothing is lost, nobody waits); only when a product is
// exhausted does the workgroup read the completion word to see whether the next one may start.
template <int BK, int NSTG, int BM, int AUX>
__global__ __launch_bounds__(256) void k_omega_chain(const ChainProg P, unsigned* __restrict__ state)
{
    __shared__ __attribute__((aligned(16))) double smem[NSTG * 2 * BK * BM];
    __shared__ int sh_job[3];
    __shared__ unsigned char sh_step[CHAIN_MAX_INST];         // per instance of this XCD: the product last seen ready
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int x = (int)(xcc & 0xfu) % NXCD;
    const int ninst = min((P.K - x + NXCD - 1) / NXCD, CHAIN_MAX_INST);   // instances x, x + 8, ...
    const int rot = ninst > 0 ? (int)(blockIdx.x / NXCD) % ninst : 0;     // claims start at different instances
    unsigned backoff = 4;
    if (threadIdx.x < CHAIN_MAX_INST) sh_step[threadIdx.x] = 0;           // product 0 has no predecessor: ready
#ifdef GGL_DEV
    long long t_claim = 0, t_idle = 0, t_tile = 0, n_tile = 0, t_prev = wall_clock64(), t_first = t_prev;
    const long long c_first = clock64();
#endif
    __syncthreads();
    for (;;) {
        if (threadIdx.x == 0) {
            int jk = -2, js = 0, jb = 0;                      // -2: every product of every instance of this XCD is handed out
            for (int ii = 0; ii < ninst && jk < 0; ++ii) {
                const int i = (ii + rot) % ninst;
                const int k = x + NXCD * i;
                unsigned* w = state + (size_t)k * CHAIN_CNT_STRIDE;
                int s = sh_step[i];
                while (s < P.nops) {
                    // L2-scope read-modify-write (no sc1): every workgroup that touches this line runs on THIS XCD (it
                    // serves the XCD it really sits on), so the XCD's L2 is the one place the line lives and its atomic
                    // unit the one place it changes -- a device-scope (memory-side) atomic cost ~26 us per claim under 96
                    // claimers per XCD
                    const unsigned nt = (unsigned)(P.begin[s + 1] - P.begin[s]);
                    const unsigned t = __hip_atomic_fetch_add(w + 1 + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (t < nt) { jk = k; js = s; jb = (int)t; break; }
                    // product s is handed out completely: may s + 1 start?  (sc1 load: served by the XCD's L2, never by this
                    // CU's L1 -- a workgroup-scope load would be an L1-hitting sc0 load)
                    const unsigned d = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (d < (unsigned)P.begin[s + 1]) { jk = max(jk, -1); break; }      // not yet: something is left, later
                    ++s;
                }
                sh_step[i] = (unsigned char)s;
            }
            sh_job[0] = jk; sh_job[1] = js; sh_job[2] = jb;
        }
        __syncthreads();
        const int jk = __builtin_amdgcn_readfirstlane(sh_job[0]);
        const int js = __builtin_amdgcn_readfirstlane(sh_job[1]);
        int jb = __builtin_amdgcn_readfirstlane(sh_job[2]);
        __syncthreads();                                      // everyone has read the slot before it is rewritten
#ifdef GGL_DEV
        if (jk == -2 && P.prof && threadIdx.x == 0) {
            long long* o = P.prof + (size_t)blockIdx.x * 8;
            o[0] = t_claim; o[1] = t_idle; o[2] = t_tile; o[3] = n_tile; o[4] = t_first; o[5] = wall_clock64(); o[6] = x; o[7] = clock64() - c_first;
        }
#endif
        if (jk == -2) return;
        if (jk == -1) {
            // nothing ready on this XCD: back off (0.2 .. 3 us) so that idle workgroups do not crowd the lines the working
            // ones need
            for (unsigned q = 0; q < backoff; ++q) __builtin_amdgcn_s_sleep(8);
            backoff = min(backoff * 2u, 64u);
#ifdef GGL_DEV
            { const long long t = wall_clock64(); t_idle += t - t_prev; t_prev = t; }
#endif
            continue;
        }
        backoff = 4;
#ifdef GGL_DEV
        { const long long t = wall_clock64(); t_claim += t - t_prev; t_prev = t; }
#endif
        const SymmOp& o = P.op[js];
        int kk = jk;
        if (o.pair && jb >= P.ntiles) { kk = P.K + jk; jb -= P.ntiles; }
        symm_dl_tile<BK, NSTG, 0, BM, 4, AUX>(o.A, o.B, o.C, o.C2, o.E, o.coef, P.K, P.p, o.A1, o.B1, o.C1, o.pair ? P.K : 0,
                                               nullptr, o.rowpart, o.fropart, kk, jb, smem);
        // this tile's stores are in the XCD's L2 before the arrival is counted; the barrier also frees the LDS image
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_fetch_add(state + (size_t)jk * CHAIN_CNT_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef GGL_DEV
        { const long long t = wall_clock64(); t_tile += t - t_prev; t_prev = t; n_tile += 1; }
#endif
    }
}

// flag[0] / flag_h[0] = 1 unless every instance completed all of its tiles
__global__ __launch_bounds__(64) void k_chain_check(const unsigned* __restrict__ state, int K, unsigned total,
                                                    int* __restrict__ flag, int* __restrict__ flag_h)
{
    bool bad = false;
    for (int k = threadIdx.x; k < K; k += 64) bad = bad || (state[(size_t)k * CHAIN_CNT_STRIDE] != total);
    if (__any(bad) && threadIdx.x == 0) { *flag = 1; *flag_h = 1; }
}

int chain_tile(int K, int p, bool force)
{
    // the chain kernel is built on the 64x64 three-stage DMA tile; it needs 16-byte rows (even p), a batch that covers the
    // XCDs (instance k lives on XCD k % 8) and enough tiles per XCD to keep its 96 workgroup slots busy across the
    // instances' seams
    if ((p & 1) != 0 || K < NXCD || K > NXCD * CHAIN_MAX_INST) return 0;
    const long T = (p + 63) / 64;
    return (force || T * (T + 1) / 2 * K >= 1100) ? 64 : 0;
}

int launch_omega_chain(hipStream_t st, const ChainProg& P, unsigned* state, int* flag, int* flag_h, int aux)
{
    static int slots = 0;
    if (!slots) {
        int dev = 0, per_cu = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_omega_chain<16, 3, 64, 16>, 256, 0) != hipSuccess || per_cu < 1)
            return -1;
        slots = std::min(per_cu, 3) * prop.multiProcessorCount;
        slots -= slots % NXCD;
    }
    if (aux == 0) hipLaunchKernelGGL((k_omega_chain<16, 3, 64, 0>), dim3(slots), dim3(256), 0, st, P, state);   // measurement only
    else
    hipLaunchKernelGGL((k_omega_chain<16, 3, 64, 16>), dim3(slots), dim3(256), 0, st, P, state);
    hipLaunchKernelGGL(k_chain_check, dim3(1), dim3(64), 0, st, state, P.K, (unsigned)P.begin[P.nops], flag, flag_h);
    return slots;
}

// ---- persistent-chain probe: nprod DEPENDENT products X <- X * X of a K-batch inside ONE cooperative
// launch, a grid-wide barrier between products, against the same chain as nprod launches.  What it prices is exactly the
// trade a persistent Omega-chain kernel would make: a kernel boundary (drain, launch, L2 write-back/invalidate by the
// command processor) against agent-scope release -> counter -> acquire by the workgroups themselves.
__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target, unsigned* err)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // this XCD's dirty L2 lines out to memory
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1L << 22)) {                              // ~0.3 s: a lost workgroup must not hang the GPU
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // this CU's L1 (and stale L2 lines) dropped
    }
    __syncthreads();
}

// two-level form: the workgroups of an XCD count themselves in on a per-XCD counter (their stores are in that XCD's L2 once
// the __syncthreads() has drained them); the LAST one writes the L2 back -- one buffer_wbl2 per XCD and barrier instead of one
// per workgroup -- and arrives at the global counter for the XCD.  bar: [0] global, [1] error, [8..15] per-XCD counters.
__device__ __forceinline__ void grid_barrier_xcd(unsigned* bar, unsigned phase, unsigned per_xcd, unsigned xcd)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(bar + 8 + xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == phase * per_xcd) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        long spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase * NXCD) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1L << 22)) {
                __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <int BK, int NSTG, int BM>
__global__ __launch_bounds__(256) void k_symm_chain_probe(double* X0, double* X1, const double* coef, int K, int p,
                                                          int nprod, unsigned* bar, unsigned* err, int two_level)
{
    __shared__ __attribute__((aligned(16))) double smem[NSTG * 2 * BK * BM];
    const int T = (p + BM - 1) / BM, ntiles = T * (T + 1) / 2;
    const int total = xcd_grid(ntiles, K);
    unsigned xcd = blockIdx.x % NXCD;
    if (two_level) {
        // the per-XCD counters need the XCD a workgroup REALLY runs on; the round-robin placement is observed, not promised
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if ((xcc & 0xf) != xcd && threadIdx.x == 0)
            __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int j = 0; j < nprod; ++j) {
        const double* src = (j & 1) ? X1 : X0;
        double* dst = (j & 1) ? X0 : X1;
        for (int L = blockIdx.x; L < total; L += gridDim.x) {       // gridDim.x is a multiple of 8: L % 8 = this XCD
            int k, b;
            if (decode_block_xcd(ntiles, K, k, b, L))
                symm_dl_tile<BK, NSTG, 0, BM, 4>(src, src, dst, nullptr, nullptr, coef, K, p, nullptr, nullptr, nullptr, 0,
                                                 nullptr, nullptr, nullptr, k, b, smem);
            __syncthreads();                                         // the slab storage is reused by the next tile
        }
        if (j + 1 < nprod) {
            if (two_level) grid_barrier_xcd(bar, (unsigned)(j + 1), gridDim.x / NXCD, xcd);
            else grid_barrier(bar, (unsigned)(j + 1) * gridDim.x, err);
        }
    }
}

// returns the grid used (0: the variant has no probe instance, < 0: HIP error)
int launch_chain_probe(hipStream_t st, double* X0, double* X1, const double* coef, int K, int p, int nprod, int variant,
                       unsigned* bar, unsigned* err, int two_level)
{
    const void* fn = nullptr;
    int BM = 64;
    if (variant == 16) fn = (const void*)k_symm_chain_probe<16, 2, 64>;
    else if (variant == 17) fn = (const void*)k_symm_chain_probe<16, 3, 64>;
    else if (variant == 20) { fn = (const void*)k_symm_chain_probe<32, 2, 32>; BM = 32; }
    else return 0;
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0) != hipSuccess || per_cu < 1) return -1;
    const int T = (p + BM - 1) / BM;
    const int total = xcd_grid(T * (T + 1) / 2, K);
    if (per_cu > 4) per_cu = 4;     // the occupancy query promised 5 (32 KiB of LDS) but 1088 / 1152 workgroups were not co-resident
    int grid = per_cu * prop.multiProcessorCount;
    if (grid > total) grid = total;
    grid -= grid % NXCD;
    if (grid < NXCD) return -1;
    void* args[] = {&X0, &X1, &coef, &K, &p, &nprod, &bar, &err, &two_level};
    if (hipLaunchCooperativeKernel(fn, dim3(grid), dim3(256), args, 0, st) != hipSuccess) return -1;
    return grid;
}
#endif

static thread_local CwRider g_rider;          // pending (K > 0): see symm_set_rider
static thread_local CopySegs g_copy_rider;    // pending (n > 0): see symm_set_copy_rider
static thread_local RedRider g_red_rider;     // pending (nblk > 0): see symm_set_reduce_rider
void symm_set_reduce_rider(const RedRider& r) { g_red_rider = r; }
void symm_set_rider(const CwRider& r) { g_rider = r; }
void symm_set_copy_rider(const CopySegs& sg) { g_copy_rider = sg; }
void symm_flush_rider(hipStream_t st)
{
    if (g_red_rider.nblk > 0) {
        launch_reduce_partials(st, g_red_rider.partials, 1, g_red_rider.nblk, g_red_rider.nv, g_red_rider.out, g_red_rider.seq,
                               g_red_rider.seq_val);
        g_red_rider = RedRider{};
    }
    if (g_copy_rider.n > 0) {
        launch_copy_small(st, g_copy_rider);
        g_copy_rider = CopySegs{};
    }
    if (g_rider.K <= 0) return;
    hipLaunchKernelGGL(k_cw_rider, dim3(g_rider.K * g_rider.nbx), dim3(256), 0, st, g_rider);
    g_rider = CwRider{};
}

static void launch_dl(hipStream_t st, const double* A, const double* B, double* C, double* C2, const double* E,
                      const double* coef, int K, int p, const double* A1, const double* B1, double* C1, int K1,
                      double* maxdev, int dl_cfg = 0, double* rowpart = nullptr, double* fropart = nullptr)
{
    // (the eight-wave development variants do not carry riders: the rider stays pending)
    const bool eight = dl_cfg == 6 || dl_cfg == 7;
    const CwRider rider = (g_rider.K > 0 && !eight) ? g_rider : CwRider{};
    if (rider.K > 0) g_rider = CwRider{};
    const CopySegs cps = (g_copy_rider.n > 0 && !eight) ? g_copy_rider : CopySegs{};
    if (cps.n > 0) g_copy_rider = CopySegs{};
    const RedRider red = (g_red_rider.nblk > 0 && !eight) ? g_red_rider : RedRider{};
    if (red.nblk > 0) g_red_rider = RedRider{};
    int ncopy = red.nblk > 0 ? NXCD : 0;          // (in front of the tiles, see k_symm_dl)
    for (int i = 0; i < cps.n; ++i) ncopy += (int)((cps.words[i] + 1023u) / 1024u);
    const int nride = rider.K * rider.nbx + ncopy;
#define GGL_DL(...) hipLaunchKernelGGL((k_symm_dl<__VA_ARGS__>), grid, dim3(256), 0, st, A, B, C, C2, E, coef, K, p, A1, B1, C1, K1, maxdev, rowpart, fropart, rider, cps, red)
    if (dl_cfg == 4 || (dl_cfg >= 8 && dl_cfg <= 13) || (dl_cfg >= 18 && dl_cfg <= 21)) {
        const int T32 = (p + 31) / 32;
        const dim3 grid(xcd_grid(T32 * (T32 + 1) / 2, K + K1) + nride);
        // 32x32 tiles, k-slab 32, double buffer: 32 KiB of LDS, five workgroups per CU.  Measured (MI355X, p = 500, us per
        // launch at K = 2 / 4 / 8 / 16): 15.9 / 21.8 / 35.0 / 58.6, against 16.3 / 26.4 / 40.2 / 70.3 with four slabs in
        // flight (64 KiB, two workgroups per CU: residency, not prefetch depth, is what the small batches lack)
        if (dl_cfg == 4) GGL_DL(32, 2, 0, 32);
#ifdef GGL_DEV
        // other slab / prefetch depths of the 32x32 kernel (LDS per workgroup: 32 / 48 / 48 / 64 / 16 / 24 KiB)
        else if (dl_cfg == 8) GGL_DL(16, 4, 0, 32);
        else if (dl_cfg == 9) GGL_DL(32, 3, 0, 32);
        else if (dl_cfg == 10) GGL_DL(16, 6, 0, 32);
        else if (dl_cfg == 11) GGL_DL(32, 4, 0, 32);
        else if (dl_cfg == 12) GGL_DL(16, 2, 0, 32);
        else if (dl_cfg == 13) GGL_DL(16, 3, 0, 32);
        // the same four timing ablations of the 32x32 kernel
        else if (dl_cfg == 18) GGL_DL(32, 2, 2, 32);
        else if (dl_cfg == 19) GGL_DL(32, 2, 3, 32);
        else if (dl_cfg == 20) GGL_DL(32, 2, 4, 32);
        else if (dl_cfg == 21) GGL_DL(32, 2, 5, 32);
#endif
        return;
    }
    const int T = (p + 63) / 64;
    const dim3 grid(xcd_grid(T * (T + 1) / 2, K + K1) + nride);
#ifdef GGL_DEV
    if (dl_cfg == 6 || dl_cfg == 7) {
        // eight waves per workgroup (two per SIMD) for batches that leave one workgroup per CU.  Measured (MI355X, p = 500):
        // K = 4: 27.0 / 26.3 us vs 26.6 us for the 32x32 kernel; K = 8: 43.9 / 49.3 vs 40.4; K = 16: 68.6 / 74.9 vs 63.9 --
        // no gain anywhere, so the shipped library does not carry them
#define GGL_DL8(...) hipLaunchKernelGGL((k_symm_dl<__VA_ARGS__>), grid, dim3(512), 0, st, A, B, C, C2, E, coef, K, p, A1, B1, C1, K1, maxdev, rowpart, fropart, rider, cps, red)
        if (dl_cfg == 6) GGL_DL8(16, 4, 0, 64, 8);
        else GGL_DL8(32, 3, 0, 64, 8);
#undef GGL_DL8
        return;
    }
#endif
    if (dl_cfg == 1) GGL_DL(16, 3);
#ifdef GGL_DEV
    else if (dl_cfg == 2) GGL_DL(16, 4);
    else if (dl_cfg == 3) GGL_DL(32, 2);
    else if (dl_cfg == 5) GGL_DL(16, 2, 1);      // no mirror write: timing ablation, wrong results
    // timing ablations of the three-stage 64x64 kernel (wrong results): no operand loads / no MFMA / no epilogue / no barrier
    else if (dl_cfg == 14) GGL_DL(16, 3, 2);
    else if (dl_cfg == 15) GGL_DL(16, 3, 3);
    else if (dl_cfg == 16) GGL_DL(16, 3, 4);
    else if (dl_cfg == 17) GGL_DL(16, 3, 5);
    // shallower slabs, more of them in flight: 32 KiB (4 x 8 KiB) / 48 KiB (6 x 8 KiB) of LDS
    else if (dl_cfg == 22) GGL_DL(8, 4);
    else if (dl_cfg == 23) GGL_DL(8, 6);
#endif
    else GGL_DL(16, 2);
#undef GGL_DL
}

// ---------------------------------------------------------------------------------------------
// General (non-symmetric result) product with a symmetric right factor:
//     C[b] = s[b] * A[b] * T[b % K],     b = 0 .. nbatch-1,
// used by the numerically stable Newton-Schulz path where Y and P = Z^T are both multiplied by
// the same T from the right (newton_schulz.hip); nbatch = 2K covers both in one launch.  A is
// read as row panels A[I0+i][m0+k] (k contiguous), T as row panels T[m0+k][J0+j] (T symmetric, so
// its rows are its columns); full output, no mirroring.
// ---------------------------------------------------------------------------------------------
template <int BM, int BK, int WM, int WN>
__global__ __launch_bounds__(sym_nt(BM, WM, WN)) void k_gemm_nt_right(const double* __restrict__ A,
                                                                      const double* __restrict__ T,
                                                                      double* __restrict__ C,
                                                                      const double* __restrict__ scal, int nbatch, int K, int p)
{
    constexpr int NWC = BM / WN, NT = sym_nt(BM, WM, WN), TI = WM / 16, TJ = WN / 16;
    constexpr int LDA = BK + 2;      // A tile [BM][BK]: (2 i + k) mod 32 distinct over a half wave
    constexpr int LDB = BM + 16;     // T tile [BK][BM]
    __shared__ __attribute__((aligned(16))) double smem[BM * LDA + BK * LDB];
    double* As = smem;
    double* Bs = smem + BM * LDA;
    const int Tn = (p + BM - 1) / BM;
    int b, tile;
    if (!decode_block_xcd(Tn * Tn, nbatch, b, tile)) return;
    const int kt = b % K;
    const int I0 = (tile / Tn) * BM, J0 = (tile % Tn) * BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = (wave / NWC) * WM, wc = (wave % NWC) * WN;
    const size_t pp = (size_t)p * p;
    const double* Ab = A + (size_t)b * pp;
    const double* Tb = T + (size_t)kt * pp;

    v4d acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

    constexpr int LPT = (BM * BK) / NT;
    static_assert((BM * BK) % NT == 0 && NT % BK == 0 && NT % BM == 0, "tile/thread mismatch");
    const int acol = tid % BK, arow = tid / BK;      // A tile element (arow + q*NT/BK, acol)
    const int bcol = tid % BM, brow = tid / BM;      // T tile element (brow + q*NT/BM, bcol)
    double ra[LPT], rb[LPT];
    auto fetch = [&](int m0) {
#pragma unroll
        for (int q = 0; q < LPT; ++q) {
            const int i = min(I0 + arow + q * (NT / BK), p - 1), m = min(m0 + acol, p - 1);
            ra[q] = Ab[(size_t)i * p + m];
            const int mb = min(m0 + brow + q * (NT / BM), p - 1), j = min(J0 + bcol, p - 1);
            rb[q] = Tb[(size_t)mb * p + j];
        }
    };
    auto stage = [&](int m0) {
#pragma unroll
        for (int q = 0; q < LPT; ++q) {
            const bool ina = (I0 + arow + q * (NT / BK)) < p && (m0 + acol) < p;
            const bool inb = (m0 + brow + q * (NT / BM)) < p && (J0 + bcol) < p;
            As[(arow + q * (NT / BK)) * LDA + acol] = ina ? ra[q] : 0.0;
            Bs[(brow + q * (NT / BM)) * LDB + bcol] = inb ? rb[q] : 0.0;
        }
    };
    fetch(0);
    for (int m0 = 0; m0 < p; m0 += BK) {
        stage(m0);
        __syncthreads();
        fetch(m0 + BK);
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            const int kq = kk * 4 + (lane >> 4);
            double af[TI], bf[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) af[i] = As[(wr + i * 16 + (lane & 15)) * LDA + kq];
#pragma unroll
            for (int j = 0; j < TJ; ++j) bf[j] = Bs[kq * LDB + wc + j * 16 + (lane & 15)];
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    const double sc = scal[b];
    double* Cb = C + (size_t)b * pp;
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = I0 + wr + ti * 16 + (lane >> 4) + 4 * r;
                const int gj = J0 + wc + tj * 16 + (lane & 15);
                if (gi < p && gj < p) Cb[(size_t)gi * p + gj] = sc * acc[ti][tj][r];
            }
}

void launch_gemm_right(hipStream_t st, const double* A, const double* T, double* C, const double* scal, int nbatch,
                       int K, int p, int variant)
{
    if (variant == 1) {
        const int Tn = (p + 127) / 128;
        hipLaunchKernelGGL((k_gemm_nt_right<128, 16, 32, 64>), dim3(xcd_grid(Tn * Tn, nbatch)), dim3(512), 0, st, A, T, C, scal, nbatch, K, p);
    } else {
        const int Tn = (p + 63) / 64;
        hipLaunchKernelGGL((k_gemm_nt_right<64, 32, 32, 32>), dim3(xcd_grid(Tn * Tn, nbatch)), dim3(256), 0, st, A, T, C, scal, nbatch, K, p);
    }
}

#ifdef GGL_DEV
// FP64 matrix-core ceiling probe: every wave issues `iters` x NACC MFMAs on NACC independent accumulators.
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma_f64_peak(double* __restrict__ out, int iters)
{
    v4d acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

// same probe with NV dependent 64-bit integer VALU ops (address arithmetic stand-ins) per MFMA
template <int NV>
__global__ __launch_bounds__(256) void k_mfma_valu_mix(double* __restrict__ out, int iters, unsigned long long seed)
{
    v4d acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    unsigned long long x = seed + threadIdx.x;
    double d = a;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (v & 1) x = (x << 3) + seed;           // v_lshl_add_u64
                else d = (x & 1) ? d : b;                 // v_cndmask pair
            }
        }
    }
    double s = d + (double)x;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV>
static double mix_probe(hipStream_t st, double* scratch, int blocks, int iters)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mfma_valu_mix<NV>, dim3(blocks), dim3(256), 0, st, scratch, 16, 12345ull);
    (void)hipEventRecord(e0, st);
    hipLaunchKernelGGL(k_mfma_valu_mix<NV>, dim3(blocks), dim3(256), 0, st, scratch, iters, 12345ull);
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return (double)blocks * 4.0 * iters * 4 * 2.0 * 16 * 16 * 4 / (ms * 1e-3) / 1e12;
}

double mfma_valu_mix_tflops(hipStream_t st, double* scratch, int blocks, int iters, int nv)
{
    switch (nv) {
        case 0: return mix_probe<0>(st, scratch, blocks, iters);
        case 2: return mix_probe<2>(st, scratch, blocks, iters);
        case 4: return mix_probe<4>(st, scratch, blocks, iters);
        case 8: return mix_probe<8>(st, scratch, blocks, iters);
        default: return mix_probe<16>(st, scratch, blocks, iters);
    }
}

template <int NACC>
static double mfma_probe(hipStream_t st, double* scratch, int blocks, int iters)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mfma_f64_peak<NACC>, dim3(blocks), dim3(256), 0, st, scratch, 16);
    (void)hipEventRecord(e0, st);
    hipLaunchKernelGGL(k_mfma_f64_peak<NACC>, dim3(blocks), dim3(256), 0, st, scratch, iters);
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    const double flops = (double)blocks * 4.0 * iters * NACC * 2.0 * 16 * 16 * 4;
    return flops / (ms * 1e-3) / 1e12;
}

// nacc in {1,2,4,8}; blocks = workgroups of 4 waves (256 per "one wave per SIMD" layer)
double mfma_f64_peak_tflops(hipStream_t st, double* scratch, int blocks, int iters, int nacc)
{
    switch (nacc) {
        case 1: return mfma_probe<1>(st, scratch, blocks, iters);
        case 2: return mfma_probe<2>(st, scratch, blocks, iters);
        case 4: return mfma_probe<4>(st, scratch, blocks, iters);
        default: return mfma_probe<8>(st, scratch, blocks, iters);
    }
}
#endif   // GGL_DEV

static constexpr int SMALL_DL_MIN_P = 16;         // (round 4: measured ahead of the register-staged 32x32 kernel at every even p from 16 up, tools/bench_symm_small.py)
static constexpr long SMALL_BATCH_TILES = 800;   // up to here the 32x32-tile kernel, above the 64x64 direct-to-LDS one

// Product-kernel variants.  The shipped library holds the instances the solvers dispatch to:
//    0  k_symm_tn 64x64 tile, k-slab 16 (register-staged; odd p, where the DMA kernel's 16-byte rows do not exist)
//    9  k_symm_tn 32x32 tile, k-slab 32 (register-staged; small batches of odd p)
//   16  k_symm_dl 64x64, double-buffered DMA      17  k_symm_dl 64x64, three DMA stages (concurrent parts)
//   20  k_symm_dl 32x32, k-slab 32, double-buffered DMA (small batches)
// A GGL_DEV build (libggl_hip_dev.so) adds the measured alternatives 1-5, 8, 11-13, 18, 19, 22 / 23 (64x64 with EIGHT
// waves per workgroup), 24-29 (other slab / prefetch depths of the 32x32 kernel, see launch_dl) and the ablations 6, 7, 10,
// 14, 15, 21 (tools/bench_*.py).
// 41-45: k_symm_sk, the split-K kernel of the under-filled batches: <BK, stages> = <8,3> <8,4> <16,2> <8,2> <4,4>
// (48 / 64 / 64 / 32 / 32 KiB of LDS per workgroup); 46-50: the same with the k-range of a tile dealt over 2 / 4 workgroups.  40 is not a launch variant: ggl_ns_stats reports it for k_omega_chain.
int symm_variants() { return 50; }
bool symm_variant_built(int v)
{
#ifdef GGL_DEV
    return v >= 0 && v <= 50 && v != 40;
#else
    return v == 0 || v == 9 || v == 16 || v == 17 || v == 20;
#endif
}

#ifdef GGL_DEV
// scratch tiles and arrival counters of the SPLIT > 1 launches, one set per stream (launches on a stream are ordered, so a
// set is never shared by two launches in flight); grown on demand, released by symm_release_workspace()
namespace {
struct SkWork { double* scratch = nullptr; unsigned* arrivals = nullptr; size_t tiles = 0, units = 0; };
std::mutex sk_mutex;
std::unordered_map<hipStream_t, SkWork> sk_work;
}
static bool sk_workspace(hipStream_t st, size_t tiles, size_t units, double** scratch, unsigned** arrivals)
{
    std::lock_guard<std::mutex> lock(sk_mutex);
    SkWork& w = sk_work[st];
    if (w.tiles < tiles || w.units < units) {
        if (w.scratch) { (void)hipStreamSynchronize(st); (void)hipFree(w.scratch); (void)hipFree(w.arrivals); w = SkWork(); }
        if (hipMalloc(&w.scratch, units * 1024 * sizeof(double)) != hipSuccess) return false;
        if (hipMalloc(&w.arrivals, tiles * sizeof(unsigned)) != hipSuccess) return false;
        if (hipMemset(w.arrivals, 0, tiles * sizeof(unsigned)) != hipSuccess) return false;
        w.tiles = tiles;
        w.units = units;
    }
    *scratch = w.scratch;
    *arrivals = w.arrivals;
    return true;
}
void symm_release_workspace(hipStream_t st)
{
    std::lock_guard<std::mutex> lock(sk_mutex);
    auto it = sk_work.find(st);
    if (it == sk_work.end()) return;
    (void)hipFree(it->second.scratch);
    (void)hipFree(it->second.arrivals);
    sk_work.erase(it);
}

static void launch_sk(hipStream_t st, const double* A, const double* B, double* C, double* C2, const double* E,
                      const double* coef, int K, int p, double* maxdev, double* rowpart, double* fropart, int variant)
{
    const int T32 = (p + 31) / 32;
    const int ntiles = T32 * (T32 + 1) / 2;
    double* scratch = nullptr;
    unsigned* arrivals = nullptr;
#define GGL_SK(BK_, ST_, SP_)                                                                                          \
    do {                                                                                                               \
        if (SP_ > 1 && !sk_workspace(st, (size_t)K * ntiles, (size_t)K * ntiles * SP_, &scratch, &arrivals)) return;   \
        hipLaunchKernelGGL((k_symm_sk<BK_, ST_, SP_>), dim3(SP_ > 1 ? sk_grid(ntiles, K, SP_) : xcd_grid(ntiles, K)),  \
                           dim3(256), 0, st, A, B, C, C2, E, coef, K, p, maxdev, rowpart, fropart, scratch, arrivals); \
    } while (0)
    switch (variant) {
        case 41: GGL_SK(8, 3, 1); break;
        case 42: GGL_SK(8, 4, 1); break;
        case 43: GGL_SK(16, 2, 1); break;
        case 44: GGL_SK(8, 2, 1); break;
        case 45: GGL_SK(4, 4, 1); break;
        case 46: GGL_SK(8, 3, 2); break;
        case 47: GGL_SK(8, 3, 4); break;
        case 48: GGL_SK(8, 2, 2); break;
        case 49: GGL_SK(8, 2, 4); break;
        default: GGL_SK(4, 4, 4); break;
    }
#undef GGL_SK
}
#else
void symm_release_workspace(hipStream_t) {}
#endif

// Measured on MI355X (tools/bench_small_batches.py, profiles/r2_small_batches_product_kernel.txt), us per launch,
// 32x32 (20) against 64x64 (16) direct-to-LDS tiles: p = 500: K = 2: 16.1 / 28.3, K = 4: 21.6 / 29.7, K = 8: 34.7 / 44.7,
// K = 16 (576 64x64 tile pairs): 59.5 / 63.5; p = 1000: K = 4 (544): 111.5 / 116.4, K = 8 (1088): 220.5 / 199.8;
// p = 200, K = 20: 11.8 / 16.4 (register-staged 32x32: 14.7).  Four times the workgroups and five of them resident per CU
// fill the chip where the 64x64 tiles leave it waiting; from ~1000 tile pairs on the 64x64 tiles' halved L2 -> LDS
// traffic wins.  Odd p: the register-staged kernels 9 / 0.
// Odd p on the direct-to-LDS kernel (round 6).  Its DMA moves 16 bytes per lane, and for odd p every other row of a stack starts
// on an 8-byte boundary only: rounds 1-5 sent every odd p to the register-staged kernel, 25-37 % slower per product.  The
// hardware takes dword-aligned addresses for global_load_lds_dwordx4 and for 16-byte global loads / stores (the queues run in
// unaligned-access mode), so what odd p needs is (i) the last column as the first half of a pair whose second half lies
// outside the matrix (symm_dl_tile: clamp, epilogue) and (ii) 8 bytes of slack behind every stack that can be an operand.
// g_odd_dl = false restores the old dispatch (GGL_OPT_ODD_DL, for A/B runs and tests).
static bool g_odd_dl = true;
void symm_set_odd_dl(bool on) { g_odd_dl = on; }
bool symm_dl_serves(int p) { return p >= 2 && ((p & 1) == 0 || g_odd_dl); }

// the kernel a variant really runs for this p: a direct-to-LDS variant that the dimension does not allow falls to the
// register-staged kernel of the same tile (what ggl_ns_stats reports: the dispatch tests assert the kernel, not the request)
int symm_effective_variant(int variant, int p)
{
    if (variant >= 16 && variant <= 39 && !symm_dl_serves(p)) return (variant == 20 || variant >= 24) ? 9 : 0;
    return variant;
}

int symm_auto_variant(int nprod, int p)
{
    const long T64 = (p + 63) / 64;
    return (T64 * (T64 + 1) / 2 * nprod <= SMALL_BATCH_TILES) ? (p >= SMALL_DL_MIN_P ? 20 : 9) : 16;
}

// Two independent products in one launch: C = coef[k]-affine(A*B) for k < K and C1 = coef[K+k]-scaled(A1*B1).
// Launch hook of the event timeline (ggl_trace_*, capi_stats.hip): called after every product launch with its stream.
static void (*g_symm_hook)(hipStream_t, int, void*) = nullptr;
static void* g_symm_hook_arg = nullptr;
void symm_set_launch_hook(void (*fn)(hipStream_t, int, void*), void* arg) { g_symm_hook = fn; g_symm_hook_arg = arg; }
struct SymmHookAtExit {
    hipStream_t st; int kind;
    ~SymmHookAtExit() { if (g_symm_hook) g_symm_hook(st, kind, g_symm_hook_arg); }
};

void launch_symm_pair(hipStream_t st, const double* A, const double* B, double* C, const double* A1, const double* B1,
                      double* C1, const double* coef2K, int K, int p, int variant)
{
    SymmHookAtExit hook{st, 1};
    if (variant < 0) variant = symm_auto_variant(2 * K, p);
    if (variant >= 41 && variant <= 50) variant = 20;       // k_symm_sk takes single products only
    switch (variant) {
        case 16: case 17: case 18: case 19: case 20: case 22: case 23: case 24: case 25: case 26: case 27: case 28: case 29: case 30: case 31: case 32: case 33: case 34: case 35: case 36: case 37: case 38: case 39:
            if (symm_dl_serves(p)) {
                launch_dl(st, A, B, C, nullptr, nullptr, coef2K, K, p, A1, B1, C1, K, nullptr, variant - 16);
                break;
            }
            if (variant == 20 || variant >= 24) launch_cfg<32, 32, 16, 16, true>(st, A, B, C, nullptr, nullptr, coef2K, K, p, A1, B1, C1, K);
            else launch_cfg<64, 16, 32, 32, true>(st, A, B, C, nullptr, nullptr, coef2K, K, p, A1, B1, C1, K);
            break;
        case 9: launch_cfg<32, 32, 16, 16, true>(st, A, B, C, nullptr, nullptr, coef2K, K, p, A1, B1, C1, K); break;
#ifdef GGL_DEV
        case 1: launch_cfg<64, 32, 32, 32, true>(st, A, B, C, nullptr, nullptr, coef2K, K, p, A1, B1, C1, K); break;
        case 3: launch_cfg<128, 16, 32, 64, false>(st, A, B, C, nullptr, nullptr, coef2K, K, p, A1, B1, C1, K); break;
        case 4: launch_cfg<64, 16, 32, 32, false>(st, A, B, C, nullptr, nullptr, coef2K, K, p, A1, B1, C1, K); break;
        case 8: launch_cfg<32, 16, 16, 16, true>(st, A, B, C, nullptr, nullptr, coef2K, K, p, A1, B1, C1, K); break;
#endif
        default: launch_cfg<64, 16, 32, 32, true>(st, A, B, C, nullptr, nullptr, coef2K, K, p, A1, B1, C1, K); break;
    }
}

// tile edge of the direct-to-LDS kernel a variant resolves to, or 0 if the variant (or an odd p) runs the register-staged
// kernel, which does not produce bound partials
int symm_bounds_tile(int K, int p, int variant)
{
    if (variant < 0) variant = symm_auto_variant(K, p);
    if (!symm_dl_serves(p)) return 0;
    if (variant == 20 || (variant >= 24 && variant <= 29) || (variant >= 34 && variant <= 37) || (variant >= 41 && variant <= 50)) return 32;
    if ((variant >= 16 && variant <= 19) || variant == 22 || variant == 23 || (variant >= 30 && variant <= 33) || variant == 38 || variant == 39) return 64;
    return 0;
}

void launch_symm(hipStream_t st, const double* A, const double* B, double* C, double* C2, const double* E,
                 const double* coef, int K, int p, int variant, double* maxdev, double* rowpart, double* fropart)
{
    SymmHookAtExit hook{st, 0};
    if (variant < 0) variant = symm_auto_variant(K, p);
#define GGL_TN(BM, BK, WM, WN, LM) \
    launch_cfg<BM, BK, WM, WN, LM>(st, A, B, C, C2, E, coef, K, p, nullptr, nullptr, nullptr, 0, maxdev)
#ifdef GGL_DEV
    if (variant >= 41 && variant <= 50) {
        if ((p & 1) == 0 && p >= 2) { launch_sk(st, A, B, C, C2, E, coef, K, p, maxdev, rowpart, fropart, variant); return; }
        variant = 9;                                     // odd p: the register-staged 32x32 kernel
    }
#endif
    switch (variant) {
        case 16: case 17: case 18: case 19: case 20: case 21: case 22: case 23: case 24: case 25: case 26: case 27: case 28: case 29: case 30: case 31: case 32: case 33: case 34: case 35: case 36: case 37: case 38: case 39:
            if (symm_dl_serves(p)) {
                launch_dl(st, A, B, C, C2, E, coef, K, p, nullptr, nullptr, nullptr, 0, maxdev, variant - 16, rowpart, fropart);
                break;
            }
            if (variant == 20 || variant >= 24) GGL_TN(32, 32, 16, 16, true);
            else GGL_TN(64, 16, 32, 32, true);
            break;
        case 9: GGL_TN(32, 32, 16, 16, true); break;
#ifdef GGL_DEV
        case 1: GGL_TN(64, 32, 32, 32, true); break;
        case 2: GGL_TN(128, 16, 64, 64, false); break;
        case 3: GGL_TN(128, 16, 32, 64, false); break;
        case 4: GGL_TN(64, 16, 32, 32, false); break;
        case 5: GGL_TN(128, 32, 64, 64, false); break;
        case 8: GGL_TN(32, 16, 16, 16, true); break;
        case 11: GGL_TN(64, 16, 16, 32, true); break;
        case 12: GGL_TN(64, 32, 16, 32, true); break;
        case 13: GGL_TN(64, 16, 32, 16, true); break;
        case 6: case 7: case 10: case 14: case 15: {
            // ablations of variant 0: 6 no global loads, 7 no MFMA, 14 ds_read + MFMA only, 15 + ds_write; 10 = timeline
            // probe (maxdev is a [grid][5] long long buffer)
            const int T = (p + 63) / 64;
            const dim3 grid(xcd_grid(T * (T + 1) / 2, K));
            double* md = (variant == 10) ? maxdev : nullptr;
#define GGL_ABL(N) hipLaunchKernelGGL((k_symm_tn<64, 16, 32, 32, true, N>), grid, dim3(256), 0, st, A, B, C, C2, E, coef, K, p, nullptr, nullptr, nullptr, 0, md)
            if (variant == 6) GGL_ABL(1); else if (variant == 7) GGL_ABL(2); else if (variant == 10) GGL_ABL(3);
            else if (variant == 14) GGL_ABL(4); else GGL_ABL(5);
#undef GGL_ABL
            break;
        }
#endif
        default: GGL_TN(64, 16, 32, 32, true); break;
    }
#undef GGL_TN
}

}  // namespace ggl
