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
#include "common.hpp"
#include "kernels.hpp"

namespace ggl {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr8_t;
typedef __attribute__((address_space(3))) void* lptr8_t;

// ---------------------------------------------------------------------------------------------
// fp64 stack (K,p,p) -> S int8 slice stacks [S][K][P][P].  flag[0] is raised when an entry exceeds the scale.
// ---------------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(256) void k_slice_i8(const double* __restrict__ A, const double* __restrict__ scaleK,
                                                  int8_t* __restrict__ out, int K, int p, int P, int* __restrict__ flag)
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
    const size_t sstride = (size_t)K * P * P;
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

void launch_slice_i8(hipStream_t st, const double* A, const double* scaleK, int8_t* out, int K, int p, int S, int* flag)
{
    const int P = (p + 63) / 64 * 64;
    dim3 grid((P / 4 + 255) / 256, P, K), blk(256);
    switch (S) {
        case 2: hipLaunchKernelGGL(k_slice_i8<2>, grid, blk, 0, st, A, scaleK, out, K, p, P, flag); break;
        case 3: hipLaunchKernelGGL(k_slice_i8<3>, grid, blk, 0, st, A, scaleK, out, K, p, P, flag); break;
        case 4: hipLaunchKernelGGL(k_slice_i8<4>, grid, blk, 0, st, A, scaleK, out, K, p, P, flag); break;
        case 5: hipLaunchKernelGGL(k_slice_i8<5>, grid, blk, 0, st, A, scaleK, out, K, p, P, flag); break;
        case 6: hipLaunchKernelGGL(k_slice_i8<6>, grid, blk, 0, st, A, scaleK, out, K, p, P, flag); break;
        case 7: hipLaunchKernelGGL(k_slice_i8<7>, grid, blk, 0, st, A, scaleK, out, K, p, P, flag); break;
        default: hipLaunchKernelGGL(k_slice_i8<8>, grid, blk, 0, st, A, scaleK, out, K, p, P, flag); break;
    }
}

// ---------------------------------------------------------------------------------------------
// the product
// ---------------------------------------------------------------------------------------------
template <int SA, int SB, int DMAX>
__global__ __launch_bounds__(256) void k_symm_i8(const int8_t* __restrict__ As, const int8_t* __restrict__ Bs,
                                                 const double* __restrict__ scaleA, const double* __restrict__ scaleB,
                                                 double* __restrict__ C, int K, int p, int P)
{
    constexpr int ND = DMAX + 1;
    constexpr int SLICE = 64 * 64;                          // bytes of one slice's 64 x 64 tile
    constexpr int STAGE = (SA + SB) * SLICE;
    extern __shared__ __attribute__((aligned(16))) int8_t smem8[];
    const int T = P / 64;
    // tile pair (I <= J) and instance: instances fastest over the XCDs (blockIdx % 8 = XCD)
    const int nt = T * (T + 1) / 2;
    int b = blockIdx.x;
    const int kk = b % K;
    b /= K;
    int I = 0;
    while (b >= T - I) { b -= T - I; ++I; }
    const int J = I + b;
    (void)nt;
    const int I0 = I * 64, J0 = J * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
    const size_t sstride = (size_t)K * P * P;
    const int8_t* Ak = As + (size_t)kk * P * P;
    const int8_t* Bk = Bs + (size_t)kk * P * P;

    v4i acc[2][2][ND];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int d = 0; d < ND; ++d) acc[i][j][d] = (v4i){0, 0, 0, 0};

    // DMA geometry: instruction n in [0, 4 (SA + SB)): operand slice n / 4 (A slices first), row group q = n % 4 (16 rows);
    // lane l -> row 16 q + (l >> 2), stored chunk l & 3 = source chunk (l & 3) ^ ((l >> 4) & 3).  Wave w issues n = w, w+4, ...
    const int drow = lane >> 2;
    const int dchunk = (lane & 3) ^ ((lane >> 4) & 3);
    auto issue = [&](int s, int buf) {
        int8_t* base = smem8 + buf * STAGE;
#pragma unroll
        for (int n0 = 0; n0 < SA + SB; ++n0) {
            // wave w handles row group q = w of every slice (n = 4 * n0 + w): uniform code, no divergence
            const int q = wave;
            const int8_t* src = (n0 < SA) ? Ak + (size_t)n0 * sstride + (size_t)(I0 + 16 * q + drow) * P
                                          : Bk + (size_t)(n0 - SA) * sstride + (size_t)(J0 + 16 * q + drow) * P;
            src += 64 * s + 16 * dchunk;
            __builtin_amdgcn_global_load_lds((gptr8_t)src, (lptr8_t)(base + n0 * SLICE + q * 1024), 16, 0, 0);
        }
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
        if (s + 1 < NS) issue(s + 1, (s + 1) & 1);
        const int8_t* st = smem8 + (s & 1) * STAGE;
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
    }

    // epilogue: fp64 recombination, smallest weights first; C/D layout of the 16x16 i32 MFMA: col = lane & 15,
    // row = 4 * (lane >> 4) + reg.  Tile and mirror are written (symmetric result).
    const double sc = scaleA[kk] * scaleB[kk];
    double* Ck = C + (size_t)kk * p * p;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double v = 0.0;
#pragma unroll
                for (int d = ND - 1; d >= 0; --d) v += (double)acc[i][j][d][r] * __builtin_ldexp(1.0, -(12 + 7 * d));
                v *= sc;
                const int row = I0 + wr + 16 * i + 4 * (lane >> 4) + r;
                const int col = J0 + wc + 16 * j + (lane & 15);
                if (row < p && col < p) {
                    if (I != J || row <= col) Ck[(size_t)row * p + col] = v;
                    if (I != J || row < col) Ck[(size_t)col * p + row] = v;
                }
            }
}

template <int SA, int SB, int DMAX>
static void launch_i8(hipStream_t st, const int8_t* As, const int8_t* Bs, const double* scaleA, const double* scaleB, double* C,
                      int K, int p)
{
    const int P = (p + 63) / 64 * 64, T = P / 64;
    const size_t lds = 2 * (size_t)(SA + SB) * 4096;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_symm_i8<SA, SB, DMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((k_symm_i8<SA, SB, DMAX>), dim3(T * (T + 1) / 2 * K), dim3(256), lds, st, As, Bs, scaleA, scaleB, C, K, p, P);
}

// S slices per operand, pairs t + u <= dmax.  Returns false for a combination that is not instantiated.
bool launch_symm_i8(hipStream_t st, const int8_t* As, const int8_t* Bs, const double* scaleA, const double* scaleB, double* C,
                    int K, int p, int S, int dmax)
{
#define GGL_I8(s, d) if (S == s && dmax == d) { launch_i8<s, s, d>(st, As, Bs, scaleA, scaleB, C, K, p); return true; }
    GGL_I8(7, 6)
    GGL_I8(6, 5)
    GGL_I8(5, 4)
    GGL_I8(4, 3)
    GGL_I8(3, 2)
    GGL_I8(2, 1)
    GGL_I8(8, 7)
    GGL_I8(5, 3)
    GGL_I8(4, 2)
#undef GGL_I8
    return false;
}

}  // namespace ggl
