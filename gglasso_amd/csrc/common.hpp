// Shared device helpers for libggl_hip (gfx950 only; wavefront = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.hpp"

#define GGL_WAVE 64

namespace ggl {

// soft threshold, solver/ggl_helper.py:12-14 (sign(v) * max(|v| - l, 0))
__device__ __forceinline__ double soft(double v, double l)
{
    double a = fmax(fabs(v) - l, 0.0);
    return v > 0.0 ? a : (v < 0.0 ? -a : 0.0);
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

// Deterministic block reduction of NV running sums; result valid in thread 0.
// scratch: NV * (blockDim/64) doubles of LDS.
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* scratch)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) scratch[wid * NV + i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            double s = 0.0;
            for (int w = 0; w < nw; ++w) s += scratch[w * NV + i];
            v[i] = s;
        }
    }
    __syncthreads();
}

// Speculative Omega-step (capi_omega.hip): the kernels that overwrite the iterate take the validation flags of the
// step's parts and do nothing when any is set -- the host then repeats the iteration without speculation.
__device__ __forceinline__ bool spec_failed(const int* __restrict__ skip)
{
    return skip != nullptr && (skip[0] | skip[1] | skip[2] | skip[3]) != 0;
}

// (tri_index / tri_len: kernels.hpp)
// the reduced flag behind the packed sums: > 0 when some rank's speculative schedule did not cover its spectrum
__device__ __forceinline__ bool gsq_rejected(const double* __restrict__ gsq, int p)
{
    return gsq != nullptr && gsq[tri_len(p)] > 0.5;
}

}  // namespace ggl
