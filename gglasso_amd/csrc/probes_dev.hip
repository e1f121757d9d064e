// Development probes (libggl_hip_dev.so only, -DGGL_DEV): hardware questions whose answers decide kernel designs.
// Nothing here is linked into the product library's dispatch.
#include "common.hpp"
#include "kernels.hpp"

#ifdef GGL_DEV
namespace ggl {

typedef double v4d __attribute__((ext_vector_type(4)));

// ---- FP64 VALU || FP64 MFMA co-issue ------------------------------------------------------
// On MI355X the FP64 vector and FP64 matrix peaks coincide (78.6 TF/s).  Are they the SAME multipliers, or can v_fma_f64
// of one wave retire while another wave's v_mfma_f64 occupies the matrix pipe?  mode:
//   0  every wave: MFMA only (4 independent accumulators)                         -> matrix peak
//   1  every wave: DFMA only (16 independent v_fma_f64 chains per lane)            -> vector peak
//   2  even waves of a SIMD MFMA, odd waves DFMA (workgroups of 8 waves: wave w and w+4 share a SIMD)
//   3  one wave: 1 MFMA followed by NV independent v_fma_f64, repeated
// out: flop executed by the MFMA instructions and by the DFMA instructions are counted separately by the host.
template <int MODE, int NV>
__global__ __launch_bounds__(512) void k_coissue(double* __restrict__ out, int iters)
{
    const int wave = threadIdx.x >> 6;
    v4d acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
    double f[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) f[i] = 1.0 + 1e-9 * (threadIdx.x + i);
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9, m = 0.999999, c = 1e-7;
    const bool do_mfma = MODE == 0 || MODE == 3 || (MODE == 2 && wave < 4);
    const bool do_fma = MODE == 1 || (MODE == 2 && wave >= 4);
    if (MODE == 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < NV; ++v) f[v & 15] = __builtin_fma(f[v & 15], m, c);
            }
        }
    } else if (do_mfma) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
    } else if (do_fma) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) f[i] = __builtin_fma(f[i], m, c);
        }
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += f[i];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE, int NV>
static double coissue_ms(hipStream_t st, double* scratch, int blocks, int threads, int iters)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k_coissue<MODE, NV>), dim3(blocks), dim3(threads), 0, st, scratch, 16);
    (void)hipEventRecord(e0, st);
    hipLaunchKernelGGL((k_coissue<MODE, NV>), dim3(blocks), dim3(threads), 0, st, scratch, iters);
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return ms;
}

// out[0..]: {mfma-only TF/s, dfma-only TF/s, split waves: mfma TF/s, dfma TF/s, same wave NV=4: mfma, dfma, NV=8: mfma, dfma,
//            NV=16: mfma, dfma, NV=32: mfma, dfma}.  Workgroups: 256 per "layer"; MODE 0/1/3 use 4 waves, MODE 2 eight.
void coissue_probe(hipStream_t st, double* scratch, double* out)
{
    const int iters = 4000;
    const double mf = 4.0 * 2.0 * 16 * 16 * 4;          // flop of the 4 MFMAs of one inner pass of one wave
    const double ff = 64.0 * 16 * 64 * 2.0 / 16;        // placeholder, replaced below per mode
    (void)ff;
    const int layers = 2;                                // two waves per SIMD in every mode
    const int blocks4 = 256 * layers;                    // 4-wave workgroups
    {
        const double ms = coissue_ms<0, 0>(st, scratch, blocks4, 256, iters);
        out[0] = blocks4 * 4.0 * iters * mf / (ms * 1e-3) / 1e12;
    }
    {
        const double ms = coissue_ms<1, 0>(st, scratch, blocks4, 256, iters);
        out[1] = blocks4 * 4.0 * iters * (64.0 * 64 * 2.0) / (ms * 1e-3) / 1e12;      // 64 v_fma_f64 x 64 lanes x 2 flop
    }
    {
        const int blocks8 = 256;                         // 8 waves per workgroup = the same two waves per SIMD
        const double ms = coissue_ms<2, 0>(st, scratch, blocks8, 512, iters);
        out[2] = blocks8 * 4.0 * iters * mf / (ms * 1e-3) / 1e12;
        out[3] = blocks8 * 4.0 * iters * (64.0 * 64 * 2.0) / (ms * 1e-3) / 1e12;
    }
    auto same = [&](auto tag, int nv, int o) {
        constexpr int NV = decltype(tag)::value;
        const double ms = coissue_ms<3, NV>(st, scratch, blocks4, 256, iters);
        out[o] = blocks4 * 4.0 * iters * mf / (ms * 1e-3) / 1e12;
        out[o + 1] = blocks4 * 4.0 * iters * (4.0 * nv * 64 * 2.0) / (ms * 1e-3) / 1e12;
    };
    same(std::integral_constant<int, 4>{}, 4, 4);
    same(std::integral_constant<int, 8>{}, 8, 6);
    same(std::integral_constant<int, 16>{}, 16, 8);
    same(std::integral_constant<int, 32>{}, 32, 10);
}

// ---- MFMA fed from LDS: what fragment-read : MFMA ratio does the matrix pipe tolerate? ----------------------------------
// The product kernel's inner loop without everything else (no DMA, no barriers, no epilogue): per k-quad a wave reads TI
// A-fragments and TJ B-fragments (ds_read_b64, immediate offsets off two base registers) and issues TI x TJ MFMAs.
// Workgroups of 4 waves with LDSB bytes of (static) LDS each, so that the residency per CU is the kernel's.
template <int TI, int TJ, int LDSB>
__global__ __launch_bounds__(256) void k_mfma_lds(double* __restrict__ out, int iters)
{
    __shared__ __attribute__((aligned(16))) double smem[LDSB / 8];
    const int lane = threadIdx.x & 63;
    for (int e = threadIdx.x; e < LDSB / 8; e += 256) smem[e] = 1.0 + 1e-9 * e;
    __syncthreads();
    v4d acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
    const double* As = smem + (lane >> 4) * 64 + (lane & 15);
    const double* Bs = As + 1024;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {                  // one 16-row slab: 4 k-quads at compile-time offsets
            double af[TI], bf[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) af[i] = As[kq * 256 + i * 16];
#pragma unroll
            for (int j = 0; j < TJ; ++j) bf[j] = Bs[kq * 256 + j * 16];
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("" ::: "memory");                    // the reads are re-issued every slab
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int TI, int TJ, int LDSB>
static double mfma_lds_tf(hipStream_t st, double* scratch, int blocks, int iters)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k_mfma_lds<TI, TJ, LDSB>), dim3(blocks), dim3(256), 0, st, scratch, 8);
    (void)hipEventRecord(e0, st);
    hipLaunchKernelGGL((k_mfma_lds<TI, TJ, LDSB>), dim3(blocks), dim3(256), 0, st, scratch, iters);
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return (double)blocks * 4.0 * iters * 4.0 * TI * TJ * 2048.0 / (ms * 1e-3) / 1e12;
}

// out[0..]: TF/s for {2x2 at 3 workgroups/CU (48 KiB), 2x2 at 5/CU (32 KiB), 2x4 at 3/CU, 4x4 at 2/CU (64 KiB), 4x4 at 3/CU
// (48 KiB; needs <= 170 registers), 1x1 at 5/CU}
void mfma_lds_probe(hipStream_t st, double* scratch, double* out)
{
    const int it = 1500;
    out[0] = mfma_lds_tf<2, 2, 49152>(st, scratch, 256 * 3, it);
    out[1] = mfma_lds_tf<2, 2, 32768>(st, scratch, 256 * 5, it);
    out[2] = mfma_lds_tf<2, 4, 49152>(st, scratch, 256 * 3, it);
    out[3] = mfma_lds_tf<4, 4, 65536>(st, scratch, 256 * 2, it);
    out[4] = mfma_lds_tf<4, 4, 49152>(st, scratch, 256 * 3, it);
    out[5] = mfma_lds_tf<1, 1, 32768>(st, scratch, 256 * 5, it);
}

}  // namespace ggl
#endif   // GGL_DEV
