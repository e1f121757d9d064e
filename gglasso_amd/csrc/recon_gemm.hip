// Reconstruction Omega_k = Q f(D) Q^T (phiplus, solver/ggl_helper.py:280-303; prox_rank_norm,
// ggl_helper.py:29-36; the reference does it with one BLAS dgemm per k, admm_solver.py:183-187).
//
// R holds the eigenvectors in ROWS (R[k,m,:] = m-th eigenvector), so
//   out[i][j] = sum_m (R[m][i] sa_m) (R[m][j] sb_m)
// is a batched "TN" product whose two operands are both read along contiguous rows.  Both maps
// are non-negative, so sa = sb = sqrt(f) turns it into a SYRK: only tile pairs I<=J are
// computed, with v_mfma_f64_16x16x4_f64 (FP64 matrix cores: 64x64 block tile, 4 waves, each
// 32x32 = 2x2 MFMA tiles, k-slab 16), and the I<J tiles are mirrored through an LDS transpose so
// that every global write is row-contiguous.  The result is bitwise symmetric.
#include "common.hpp"
#include "kernels.hpp"

namespace ggl {

typedef double v4d __attribute__((ext_vector_type(4)));

static constexpr int RB = 64;         // block tile edge
static constexpr int RK = 16;         // rows of R per slab
static constexpr int RLD = RB + 16;   // LDS row stride: consecutive rows land on opposite bank halves
static constexpr int CLD = RB + 1;    // transpose tile stride

// scale[k][0][m] / scale[k][1][m]: factors of the two operands
__global__ __launch_bounds__(256) void k_eigmap(const double* __restrict__ D, const double* __restrict__ betaK, int map,
                                                int p, double* __restrict__ scale)
{
    const int k = blockIdx.y;
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= p) return;
    const double d = D[(size_t)k * p + m];
    const double beta = betaK ? betaK[k] : 0.0;
    double sa, sb;
    if (map == MAP_PHIPLUS) sa = sb = sqrt(0.5 * (sqrt(d * d + 4.0 * beta) + d));
    else if (map == MAP_RANK) sa = sb = sqrt(fmax(d - beta, 0.0));
    else { sa = d; sb = 1.0; }
    scale[((size_t)k * 2 + 0) * p + m] = sa;
    scale[((size_t)k * 2 + 1) * p + m] = sb;
}

__global__ __launch_bounds__(256) void k_recon(double* __restrict__ out, const double* __restrict__ R,
                                               const double* __restrict__ scale, int p)
{
    __shared__ __attribute__((aligned(16))) double smem[RB * CLD];   // As | Bs, later the transpose tile
    double* As = smem;
    double* Bs = smem + RK * RLD;
    const int k = blockIdx.y;
    const int T = (p + RB - 1) / RB;
    int I = 0, b = blockIdx.x;
    while (b >= T - I) { b -= T - I; ++I; }
    const int J = I + b;
    const int I0 = I * RB, J0 = J * RB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
    const double* Rk = R + (size_t)k * p * p;
    const double* sa = scale + (size_t)k * 2 * p;
    const double* sb = sa + p;

    v4d acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

    const int lcol = tid & 63, lrow = tid >> 6;
    const bool aok = (I0 + lcol) < p, bok = (J0 + lcol) < p;

    for (int m0 = 0; m0 < p; m0 += RK) {
#pragma unroll
        for (int q = 0; q < RK / 4; ++q) {
            const int row = lrow + 4 * q;
            const int m = m0 + row;
            double av = 0.0, bv = 0.0;
            if (m < p) {
                if (aok) av = Rk[(size_t)m * p + I0 + lcol] * sa[m];
                if (bok) bv = Rk[(size_t)m * p + J0 + lcol] * sb[m];
            }
            As[row * RLD + lcol] = av;
            Bs[row * RLD + lcol] = bv;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < RK / 4; ++kk) {
            const int row = kk * 4 + (lane >> 4);
            const double a0 = As[row * RLD + wr + (lane & 15)];
            const double a1 = As[row * RLD + wr + 16 + (lane & 15)];
            const double b0 = Bs[row * RLD + wc + (lane & 15)];
            const double b1 = Bs[row * RLD + wc + 16 + (lane & 15)];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }

    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
    double* Ok = out + (size_t)k * p * p;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wr + ti * 16 + (lane >> 4) + 4 * r;
                const int col = wc + tj * 16 + (lane & 15);
                if (I0 + row < p && J0 + col < p) Ok[(size_t)(I0 + row) * p + J0 + col] = acc[ti][tj][r];
                if (I != J) smem[row * CLD + col] = acc[ti][tj][r];
            }
    if (I != J) {
        __syncthreads();
        for (int e = tid; e < RB * RB; e += 256) {
            const int a = e >> 6, c = e & 63;   // out[J0+a][I0+c] = tile[c][a]
            if (J0 + a < p && I0 + c < p) Ok[(size_t)(J0 + a) * p + I0 + c] = smem[c * CLD + a];
        }
    }
}

void launch_recon(hipStream_t st, double* out, const double* R, const double* D, const double* betaK, int map,
                  int K, int p, double* scale_work)
{
    hipLaunchKernelGGL(k_eigmap, dim3((p + 255) / 256, K), dim3(256), 0, st, D, betaK, map, p, scale_work);
    const int T = (p + RB - 1) / RB;
    hipLaunchKernelGGL(k_recon, dim3(T * (T + 1) / 2, K), dim3(256), 0, st, out, R, scale_work, p);
}

}  // namespace ggl
