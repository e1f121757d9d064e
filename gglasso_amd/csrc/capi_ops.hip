// Part of the C ABI of libggl_hip.so (include/ggl_hip.h); see capi_internal.hpp for the map of the translation units.
#include "capi_internal.hpp"

static int eig_common(int K, int p, const double* A, const double* beta, double* D, double* Q, double* out, int map,
                      int eig_method)
{
    ARGCHK(K >= 1 && p >= 1 && A, "K, p, A");
    ARGCHK(eig_method != GGL_EIG_JACOBI || jacobi_fits(p), "GGL_EIG_JACOBI needs p <= GGL_JACOBI_MAX_P");
    const bool jac = (eig_method == GGL_EIG_JACOBI) || (eig_method == GGL_EIG_AUTO && jacobi_fits(p));
    const size_t n = (size_t)K * p * p, kp = (size_t)K * p;
    DevBuf dA, dD, dR, dO, dB, dE, dS;
    int* dinfo = nullptr;
    HIPCHK(dA.alloc(n));
    HIPCHK(dD.alloc(kp));
    HIPCHK(dB.alloc(K));
    HIPCHK(hipMalloc(&dinfo, K * sizeof(int)));
    struct InfoFree { int* p; ~InfoFree() { (void)hipFree(p); } } infofree{dinfo};
    UP(dA.p, A, n);
    if (beta) UP(dB.p, beta, K);
    if (out) HIPCHK(dO.alloc(n));
    std::vector<int> info(K);
    if (jac) {
        if (Q) HIPCHK(dR.alloc(n));
        HIPCHK(launch_jacobi(nullptr, dA.p, dD.p, dR.p, dO.p, map, beta ? dB.p : nullptr, dinfo, K, p));
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(info.data(), dinfo, K * sizeof(int), hipMemcpyDeviceToHost));
        for (int k = 0; k < K; ++k)
            if (info[k] < 0) return fail(GGL_E_SOLVER, "Jacobi eigensolver did not converge (instance %d)", k);
    } else {
        rocblas_handle h;
        if (rocblas_create_handle(&h) != rocblas_status_success) return fail(GGL_E_SOLVER, "rocblas_create_handle");
        HIPCHK(dE.alloc(kp));
        rocblas_status st = rocsolver_dsyevd_strided_batched(h, (Q || out) ? rocblas_evect_original : rocblas_evect_none,
                                                             rocblas_fill_upper, p, dA.p, p, (rocblas_stride)p * p,
                                                             dD.p, p, dE.p, p, dinfo, K);
        if (st == rocblas_status_success && out) {
            hipError_t e = dS.alloc(2 * kp);
            if (e == hipSuccess) launch_recon(nullptr, dO.p, dA.p, dD.p, beta ? dB.p : nullptr, map, K, p, dS.p);
        }
        (void)hipDeviceSynchronize();
        rocblas_destroy_handle(h);
        if (st != rocblas_status_success) return fail(GGL_E_SOLVER, "rocsolver_dsyevd_strided_batched: status %d", (int)st);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpy(info.data(), dinfo, K * sizeof(int), hipMemcpyDeviceToHost));
        for (int k = 0; k < K; ++k)
            if (info[k] != 0) return fail(GGL_E_SOLVER, "rocSOLVER syevd did not converge (instance %d)", k);
    }
    if (out) DOWN(out, dO.p, n);
    if (D || Q) {
        // NumPy convention across the ABI: ascending eigenvalues, eigenvectors in columns.
        std::vector<double> hd(kp), hr;
        DOWN(hd.data(), dD.p, kp);
        if (Q) { hr.resize(n); DOWN(hr.data(), jac ? dR.p : dA.p, n); }
        std::vector<int> idx(p);
        for (int k = 0; k < K; ++k) {
            std::iota(idx.begin(), idx.end(), 0);
            const double* dk = hd.data() + (size_t)k * p;
            std::stable_sort(idx.begin(), idx.end(), [dk](int a, int b) { return dk[a] < dk[b]; });
            for (int m = 0; m < p; ++m) {
                if (D) D[(size_t)k * p + m] = dk[idx[m]];
                if (Q) {
                    const double* row = hr.data() + (size_t)k * p * p + (size_t)idx[m] * p;
                    for (int i = 0; i < p; ++i) Q[(size_t)k * p * p + (size_t)i * p + m] = row[i];
                }
            }
        }
    }
    return GGL_OK;
}

extern "C" int ggl_dev_symm(int K, int p, const double* A, const double* B, const double* E, const double* coef5K,
                            double* C, double* C2, int variant)
{
    ARGCHK(K >= 1 && p >= 1 && A && B && coef5K && C, "arguments");
    ARGCHK(variant < 0 || symm_variant_built(variant), "product-kernel variant not in this build");
    const size_t n = (size_t)K * p * p;
    DevBuf dA, dB, dE, dC, dC2, dcoef;
    HIPCHK(dA.alloc(n));
    HIPCHK(dB.alloc(n));
    HIPCHK(dC.alloc(n));
    std::vector<double> cw((size_t)K * NS_NCOEF, 0.0);     // rows {cI,cAcc,cE,dI,dC} widened by dE = 0
    for (int k = 0; k < K; ++k) std::copy(coef5K + (size_t)k * 5, coef5K + (size_t)k * 5 + 5, cw.begin() + (size_t)k * NS_NCOEF);
    HIPCHK(dcoef.alloc(cw.size()));
    UP(dA.p, A, n);
    UP(dB.p, B, n);
    UP(dcoef.p, cw.data(), cw.size());
    if (E) { HIPCHK(dE.alloc(n)); UP(dE.p, E, n); }
    if (C2) HIPCHK(dC2.alloc(n));
    launch_symm(nullptr, dA.p, dB.p, dC.p, C2 ? dC2.p : nullptr, E ? dE.p : nullptr, dcoef.p, K, p, variant);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    DOWN(C, dC.p, n);
    if (C2) DOWN(C2, dC2.p, n);
    return GGL_OK;
}

// kernel unit test of the bound partials: C = A B with the product kernel's epilogue partials, then the row sums of
// |C| (K,p), |C|_F^2 (K) and the spectral bound sqrt(min(|C|_inf, Collatz-Wielandt, |C|_F)) (K) from them
extern "C" int ggl_dev_symm_bounds(int K, int p, const double* A, const double* B, int variant, double* C,
                                   double* rowsum_out, double* fro2_out, double* bound_out)
{
    ARGCHK(K >= 1 && p >= 1 && A && B && C && rowsum_out && fro2_out && bound_out, "arguments");
    ARGCHK(variant < 0 || symm_variant_built(variant), "product-kernel variant not in this build");
    const int tile = symm_bounds_tile(K, p, variant);
    ARGCHK(tile != 0, "this variant / p has no bound partials (direct-to-LDS kernels, even p)");
    const int T = (p + tile - 1) / tile, ntile = T * (T + 1) / 2, nib = bound_rows_blocks(p);
    const size_t n = (size_t)K * p * p;
    DevBuf dA, dB, dC, dcoef, drow, dfro, dd, dinf, dout;
    unsigned long long* cw = nullptr;
    unsigned* cnt = nullptr;
    HIPCHK(dA.alloc(n)); HIPCHK(dB.alloc(n)); HIPCHK(dC.alloc(n));
    HIPCHK(drow.alloc((size_t)K * T * p)); HIPCHK(dfro.alloc((size_t)K * ntile)); HIPCHK(dd.alloc((size_t)K * p));
    HIPCHK(dinf.alloc((size_t)K * nib)); HIPCHK(dout.alloc(K));
    HIPCHK(hipMalloc(&cw, K * sizeof(unsigned long long)));
    HIPCHK(hipMalloc(&cnt, K * sizeof(unsigned)));
    struct Free2 { void *a, *b; ~Free2() { (void)hipFree(a); (void)hipFree(b); } } free2{cw, cnt};
    HIPCHK(hipMemset(cw, 0, K * sizeof(unsigned long long)));
    HIPCHK(hipMemset(cnt, 0, K * sizeof(unsigned)));
    HIPCHK(hipMemset(drow.p, 0xff, (size_t)K * T * p * sizeof(double)));      // every slot must be written by the kernel
    HIPCHK(hipMemset(dfro.p, 0xff, (size_t)K * ntile * sizeof(double)));
    std::vector<double> coef((size_t)K * NS_NCOEF, 0.0);
    for (int k = 0; k < K; ++k) coef[(size_t)k * NS_NCOEF + 1] = 1.0;
    HIPCHK(dcoef.alloc(coef.size()));
    UP(dA.p, A, n);
    UP(dB.p, B, n);
    UP(dcoef.p, coef.data(), coef.size());
    launch_symm(nullptr, dA.p, dB.p, dC.p, nullptr, nullptr, dcoef.p, K, p, variant, nullptr, drow.p, dfro.p);
    launch_bound_rows(nullptr, drow.p, T, K, p, dd.p, dinf.p);
    launch_cw_final(nullptr, dC.p, dd.p, K, p, dinf.p, dfro.p, ntile, cw, cnt, dout.p, nullptr, nullptr, nullptr, 0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    DOWN(C, dC.p, n);
    DOWN(rowsum_out, dd.p, (size_t)K * p);
    DOWN(bound_out, dout.p, K);
    std::vector<double> fr((size_t)K * ntile);
    DOWN(fr.data(), dfro.p, fr.size());
    for (int k = 0; k < K; ++k) {
        double sq = 0.0;
        for (int t = 0; t < ntile; ++t) sq += fr[(size_t)k * ntile + t];
        fro2_out[k] = sq;
    }
    return GGL_OK;
}

extern "C" int ggl_dev_symm_bench(int K, int p, int variant, int iters, double* ms_out)
{
    ARGCHK(K >= 1 && p >= 1 && iters >= 1 && ms_out, "arguments");
    ARGCHK(variant < 0 || symm_variant_built(variant), "product-kernel variant not in this build");
    const size_t n = (size_t)K * p * p;
    std::vector<double> h(n), coef((size_t)K * NS_NCOEF, 0.0);
    unsigned long long s = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        h[i] = (double)(s >> 11) / 9007199254740992.0 - 0.5;
    }
    for (int k = 0; k < K; ++k) coef[(size_t)k * NS_NCOEF + 1] = 1.0 / p;
    DevBuf dA, dB, dC, dcoef;
    HIPCHK(dA.alloc(n));
    HIPCHK(dB.alloc(n));
    HIPCHK(dC.alloc(n));
    HIPCHK(dcoef.alloc(coef.size()));
    UP(dA.p, h.data(), n);
    UP(dB.p, h.data(), n);
    UP(dcoef.p, coef.data(), coef.size());
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch_symm(nullptr, dA.p, dB.p, dC.p, nullptr, nullptr, dcoef.p, K, p, variant);
    HIPCHK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) launch_symm(nullptr, dA.p, dB.p, dC.p, nullptr, nullptr, dcoef.p, K, p, variant);
    HIPCHK(hipEventRecord(e1, nullptr));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    HIPCHK(hipGetLastError());
    *ms_out = ms / iters;
    return GGL_OK;
}

#ifdef GGL_DEV
// What does a change of kernel between dependent launches cost?  mode 0: `iters` products; 1: `iters` x (product, then a
// one-thread kernel that stores a word); 2: `iters` of the one-thread kernel; 3: `iters` x (product, elementwise scale by 1 of
// the output: an LDS-free kernel over the same data).  ms per repetition (tools/kernel_switch_cost.py).
extern "C" int ggl_dev_switch_bench(int K, int p, int variant, int iters, int mode, double* ms_out)
{
    ARGCHK(K >= 1 && p >= 1 && iters >= 1 && ms_out && mode >= 0 && mode <= 3, "arguments");
    const size_t n = (size_t)K * p * p;
    std::vector<double> h(n, 0.25), coef((size_t)K * NS_NCOEF, 0.0);
    for (int k = 0; k < K; ++k) coef[(size_t)k * NS_NCOEF + 1] = 1.0 / p;
    DevBuf dA, dB, dC, dcoef, dflag;
    HIPCHK(dA.alloc(n));
    HIPCHK(dB.alloc(n));
    HIPCHK(dC.alloc(n));
    HIPCHK(dcoef.alloc(coef.size()));
    HIPCHK(dflag.alloc(8));
    UP(dA.p, h.data(), n);
    UP(dB.p, h.data(), n);
    UP(dcoef.p, coef.data(), coef.size());
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    auto rep = [&](int i) {
        if (mode != 2) launch_symm(nullptr, dA.p, dB.p, dC.p, nullptr, nullptr, dcoef.p, K, p, variant);
        if (mode == 1 || mode == 2) launch_set_flag(nullptr, (unsigned long long*)dflag.p, (unsigned long long)i);
        if (mode == 3) launch_scale(nullptr, dC.p, 1.0, n);
    };
    for (int i = 0; i < 3; ++i) rep(i);
    HIPCHK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) rep(i);
    HIPCHK(hipEventRecord(e1, nullptr));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    HIPCHK(hipGetLastError());
    *ms_out = ms / iters;
    return GGL_OK;
}

// Two probes for the readings of round 5's intermittent RANK table.
// (A) ggl_dev_fill_copy_probe: what the snapshots of a batch used to do with the runtime's own operations -- hipMalloc a stack,
//     hipMemsetAsync it to zero, hipMemcpyAsync device-to-device `slices` instances into it on the SAME stream (with `gap_us` of
//     kernel work queued between the fill and the copies when > 0), synchronise, count the slices that came out zero.
//     out = { repetitions, slices lost }.
extern "C" int ggl_dev_fill_copy_probe(int reps, int K, int p, int slices, int gap_us, long long out[2])
{
    ARGCHK(reps >= 1 && K >= 1 && p >= 1 && slices >= 1 && slices <= K && out, "arguments");
    const size_t pp = (size_t)p * p, n = (size_t)K * pp;
    hipStream_t st = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    DevBuf src, flag;
    HIPCHK(src.alloc(n));
    HIPCHK(flag.alloc(K));
    std::vector<double> ones(n, 1.0), got(K);
    UP(src.p, ones.data(), n);
    long long lost = 0;
    for (int r = 0; r < reps; ++r) {
        double* dst = nullptr;
        HIPCHK(hipMalloc(&dst, n * sizeof(double)));
        HIPCHK(hipMemsetAsync(dst, 0, n * sizeof(double), st));
        if (gap_us > 0) launch_spin_us(st, gap_us);
        for (int i = 0; i < slices; ++i) {
            const int k = (i * 7 + r) % K;
            HIPCHK(hipMemcpyAsync(dst + (size_t)k * pp, src.p + (size_t)k * pp, pp * sizeof(double), hipMemcpyDeviceToDevice, st));
        }
        // one element of every slot, read back after the stream has drained
        HIPCHK(hipStreamSynchronize(st));
        for (int i = 0; i < slices; ++i) {
            const int k = (i * 7 + r) % K;
            double v = -1.0;
            HIPCHK(hipMemcpy(&v, dst + (size_t)k * pp + pp / 2, sizeof(double), hipMemcpyDeviceToHost));
            if (v != 1.0) lost += 1;
        }
        HIPCHK(hipFree(dst));
    }
    (void)hipStreamDestroy(st);
    out[0] = reps;
    out[1] = lost;
    return GGL_OK;
}

// A yardstick for the product kernel: what the vendor's FP64 GEMM reaches on this chip at the same shapes.
// mode 0: rocblas_dgemm_strided_batched C = A B (N,N);  1: C = A^T B (the operand layout of k_symm_tn / k_symm_dl);
// 2: rocblas_dsyrk_strided_batched C = A A^T, one triangle (p^3 flop per instance, like a symmetric product);
// 3: rocblas_dsyrkx_strided_batched C = A B^T, one triangle -- the library's form of OUR product (A B symmetric);
// 4: k_symm (variant by size) for comparison in the same process.  ms per call.  Dev library only; nothing on the solver's path.
extern "C" int ggl_dev_vendor_bench(int K, int p, int mode, int iters, double* ms_out)
{
    ARGCHK(K >= 1 && p >= 1 && iters >= 1 && ms_out && mode >= 0 && mode <= 4, "arguments");
    const size_t n = (size_t)K * p * p;
    std::vector<double> h(n), coef((size_t)K * NS_NCOEF, 0.0);
    unsigned long long s = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        h[i] = (double)(s >> 11) / 9007199254740992.0 - 0.5;
    }
    for (int k = 0; k < K; ++k) coef[(size_t)k * NS_NCOEF + 1] = 1.0 / p;
    DevBuf dA, dB, dC, dcoef;
    HIPCHK(dA.alloc(n));
    HIPCHK(dB.alloc(n));
    HIPCHK(dC.alloc(n));
    HIPCHK(dcoef.alloc(coef.size()));
    UP(dA.p, h.data(), n);
    UP(dB.p, h.data(), n);
    UP(dcoef.p, coef.data(), coef.size());
    rocblas_handle hd = nullptr;
    if (rocblas_create_handle(&hd) != rocblas_status_success) return fail(GGL_E_SOLVER, "rocblas_create_handle failed");
    const double one = 1.0 / p, zero = 0.0;
    const rocblas_stride st = (rocblas_stride)p * p;
    rocblas_status rs = rocblas_status_success;
    auto rep = [&]() {
        switch (mode) {
            case 0: rs = rocblas_dgemm_strided_batched(hd, rocblas_operation_none, rocblas_operation_none, p, p, p, &one, dA.p, p, st,
                                                       dB.p, p, st, &zero, dC.p, p, st, K); break;
            case 1: rs = rocblas_dgemm_strided_batched(hd, rocblas_operation_transpose, rocblas_operation_none, p, p, p, &one, dA.p, p,
                                                       st, dB.p, p, st, &zero, dC.p, p, st, K); break;
            case 2: rs = rocblas_dsyrk_strided_batched(hd, rocblas_fill_upper, rocblas_operation_none, p, p, &one, dA.p, p, st, &zero,
                                                       dC.p, p, st, K); break;
            case 3: rs = rocblas_dsyrkx_strided_batched(hd, rocblas_fill_upper, rocblas_operation_none, p, p, &one, dA.p, p, st, dB.p, p,
                                                        st, &zero, dC.p, p, st, K); break;
            default: launch_symm(nullptr, dA.p, dB.p, dC.p, nullptr, nullptr, dcoef.p, K, p, -1); break;
        }
    };
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) rep();
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) rep();
    HIPCHK(hipEventRecord(e1, nullptr));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)rocblas_destroy_handle(hd);
    if (rs != rocblas_status_success) return fail(GGL_E_SOLVER, "rocBLAS call failed: status %d", (int)rs);
    HIPCHK(hipGetLastError());
    *ms_out = ms / iters;
    return GGL_OK;
}
#endif

// LDS stages of the int8 product kernel: 1 (default: two workgroups per CU cover each other's loads) or 2 (double buffer)
#ifdef GGL_DEV
extern "C" int ggl_dev_i8_stages(int n)
{
    ARGCHK(n == 1 || n == 2, "1 or 2 stages");
    symm_i8_set_stages(n);
    return GGL_OK;
}

// Symmetric product on the INT8 matrix cores (gemm_i8.hip): C = A B from S int8 slices per operand, slice
// pairs t + u <= dmax.  A, B, C: (K,p,p) host arrays, |A| <= scaleA, |B| <= scaleB entrywise (powers of two).
// ms_out[0]: slicing both operands (two launches), ms_out[1]: one product launch (mean of iters), ms_out[2]: overflow flag.
extern "C" int ggl_dev_symm_i8(int K, int p, int S, int dmax, const double* A, const double* B, double scaleA, double scaleB,
                               double* C, int iters, double* ms_out)
{
    ARGCHK(K >= 1 && p >= 1 && iters >= 1 && A && B && C && ms_out, "arguments");
    ARGCHK(S >= 2 && S <= 8, "2 <= S <= 8");
    const size_t n = (size_t)K * p * p;
    const int P = (p + 63) / 64 * 64;
    const size_t nslice = (size_t)S * K * P * P;
    DevBuf dA, dB, dC, dsc;
    HIPCHK(dA.alloc(n));
    HIPCHK(dB.alloc(n));
    HIPCHK(dC.alloc(n));
    HIPCHK(dsc.alloc(2 * (size_t)K));
    DevBuf dpar;
    HIPCHK(dpar.alloc(12 * (size_t)K));
    {
        std::vector<double> par(12 * (size_t)K, 0.0);
        for (int k = 0; k < K; ++k) { par[12 * k + 1] = 1.0; par[12 * k + 8] = scaleA * scaleB; par[12 * k + 9] = par[12 * k + 10] = 1.0; }
        UP(dpar.p, par.data(), par.size());
    }
    int8_t *sA = nullptr, *sB = nullptr;
    int* flag = nullptr;
    HIPCHK(hipMalloc(&sA, nslice));
    HIPCHK(hipMalloc(&sB, nslice));
    HIPCHK(hipMalloc(&flag, sizeof(int)));
    HIPCHK(hipMemset(flag, 0, sizeof(int)));
    std::vector<double> sc(2 * (size_t)K);
    for (int k = 0; k < K; ++k) { sc[k] = scaleA; sc[K + k] = scaleB; }
    UP(dA.p, A, n);
    UP(dB.p, B, n);
    UP(dsc.p, sc.data(), sc.size());
    hipEvent_t e0, e1, e2;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventCreate(&e2));
    int rc = GGL_OK;
    launch_slice_i8(nullptr, dA.p, dsc.p, sA, K, p, S, flag);       // warm-up
    HIPCHK(hipEventRecord(e0, nullptr));
    launch_slice_i8(nullptr, dA.p, dsc.p, sA, K, p, S, flag);
    launch_slice_i8(nullptr, dB.p, dsc.p + K, sB, K, p, S, flag);
    HIPCHK(hipEventRecord(e1, nullptr));
    if (!launch_symm_i8(nullptr, sA, sB, dpar.p, dC.p, K, p, S, dmax))
        rc = fail(GGL_E_ARG, "bad argument: (S, dmax) = (%d, %d) is not instantiated", S, dmax);
    if (!rc) {
        HIPCHK(hipEventRecord(e1, nullptr));
        for (int i = 0; i < iters; ++i) launch_symm_i8(nullptr, sA, sB, dpar.p, dC.p, K, p, S, dmax);
        HIPCHK(hipEventRecord(e2, nullptr));
        HIPCHK(hipEventSynchronize(e2));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e1, e2));
        ms_out[1] = ms / iters;
        int hflag = 0;
        HIPCHK(hipMemcpy(&hflag, flag, sizeof(int), hipMemcpyDeviceToHost));
        ms_out[2] = hflag;
        HIPCHK(hipGetLastError());
        DOWN(C, dC.p, n);
    }
    {
        // slicing time, measured on its own
        HIPCHK(hipEventRecord(e0, nullptr));
        launch_slice_i8(nullptr, dA.p, dsc.p, sA, K, p, S, flag);
        launch_slice_i8(nullptr, dB.p, dsc.p + K, sB, K, p, S, flag);
        HIPCHK(hipEventRecord(e1, nullptr));
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        ms_out[0] = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipEventDestroy(e2);
    (void)hipFree(sA);
    (void)hipFree(sB);
    (void)hipFree(flag);
    return rc;
}

// The whole Omega-step on the int8 matrix cores (gemm_i8.hip: i8_omega_plan / i8_omega_run), stand-alone: Omega = phiplus(W)
// for a (K,p,p) host stack W, beta (K), spectral bounds cbound (K) >= lambda_max(W^2 + 4 beta I).  cfg = {s_full, s_f2, s_gf2,
// s_ye, d_ye} (0: defaults).  ms_out = {mean milliseconds of one step (slicing of W + products), products, overflow flag,
// algorithmic units (fp64 products the schedule stands for)}.
extern "C" int ggl_dev_omega_i8(int K, int p, const double* W, const double* beta, const double* cbound, const int* cfg5,
                                double tol, double* Omega, int iters, double* ms_out)
{
    ARGCHK(K >= 1 && p >= 1 && W && beta && cbound && Omega && iters >= 1 && ms_out, "arguments");
    const size_t n = (size_t)K * p * p;
    I8Omega w;
    if (i8_omega_alloc(&w, K, p) != 0) return fail(GGL_E_HIP, "i8 workspace: allocation failed");
    DevBuf dW, dA, dB, dY, dF, dF2, dOm;
    int rc = GGL_OK;
    do {
        if (dW.alloc(n) || dA.alloc(n) || dB.alloc(n) || dY.alloc(n) || dF.alloc(n) || dF2.alloc(n) || dOm.alloc(n)) {
            rc = fail(GGL_E_HIP, "allocation failed");
            break;
        }
        if (hipMemcpy(dW.p, W, n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { rc = fail(GGL_E_HIP, "upload"); break; }
        I8Cfg cfg;
        if (cfg5 && cfg5[0] > 0) { cfg.s_full = cfg5[0]; cfg.s_f2 = cfg5[1]; cfg.s_gf2 = cfg5[2]; cfg.s_ye = cfg5[3]; cfg.d_ye = cfg5[4]; }
        I8Bufs bufs = {dW.p, dA.p, dB.p, dY.p, dF.p, dF2.p, dOm.p};
        I8Prog prog;
        const int np = i8_omega_plan(&w, cbound, beta, 0, K, tol, 9, cfg, bufs, &prog);
        if (np <= 0) { rc = fail(GGL_E_ARG, "bad argument: no two-step schedule for these bounds (%d)", np); break; }
        if (hipMemcpy(w.par, w.par_h, (size_t)I8_MAXPROD * K * 12 * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(w.wscale, w.wscale_h, K * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
            rc = fail(GGL_E_HIP, "upload of the parameter rows");
            break;
        }
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        bool ok = i8_omega_run(nullptr, &w, prog, dW.p);      // warm-up (and the result)
        (void)hipEventRecord(e0, nullptr);
        for (int i = 0; ok && i < iters; ++i) ok = i8_omega_run(nullptr, &w, prog, dW.p);
        (void)hipEventRecord(e1, nullptr);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        if (!ok) { rc = fail(GGL_E_ARG, "bad argument: a slice configuration of the plan is not instantiated"); break; }
        if (hipGetLastError() != hipSuccess) { rc = fail(GGL_E_HIP, "i8 Omega-step: launch failed"); break; }
        int hflag = 0;
        (void)hipMemcpy(&hflag, w.flag, sizeof(int), hipMemcpyDeviceToHost);
        ms_out[0] = ms / iters;
        ms_out[1] = np;
        ms_out[2] = hflag;
        ms_out[3] = prog.units;
        if (hipMemcpy(Omega, dOm.p, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(GGL_E_HIP, "download"); break; }
    } while (0);
    i8_omega_free(&w);
    return rc;
}
#endif  // GGL_DEV (the int8 route: measured, rejected -- DESIGN 9.4)

// the LDS-resident Omega-step (omega_lds.hip) stand-alone: Omega = phiplus(Theta - L - X - beta S, beta) of K instances in one
// launch.  L may be NULL.  out = {ms per launch, fallback flag, products summed over the instances, table entries}
extern "C" int ggl_dev_omega_lds(int K, int p, const double* Theta, const double* L, const double* X, const double* S,
                                 const double* beta, double tol, int degrees, double* Omega, double* cbound, int iters, double* out)
{
    ARGCHK(K >= 1 && p >= 1 && Theta && X && S && beta && Omega && iters >= 1 && out, "arguments");
    ARGCHK(p <= omega_lds_max_p(), "p above the LDS-resident kernel's range");
    const int waves = degrees / 1000;             // degrees + 1000 * waves: 4 or 8 waves per workgroup (0: by size)
    degrees %= 1000;
    ARGCHK(waves == 0 || waves == 4 || waves == 8, "waves per workgroup: 4 or 8");
    const size_t n = (size_t)K * p * p;
    std::vector<double> tab((size_t)OMEGA_LDS_MAXTAB * OMEGA_LDS_ENT);
    double lnq = 0.0;
    const int ntab = omega_lds_build_table(tol, degrees, tab.data(), OMEGA_LDS_MAXTAB, &lnq);
    ARGCHK(ntab >= 1, "empty schedule table");
    DevBuf dT, dL, dX, dS, dB, dO, dTab, dC, dMisc;
    HIPCHK(dT.alloc(n)); HIPCHK(dX.alloc(n)); HIPCHK(dS.alloc(n)); HIPCHK(dB.alloc(K)); HIPCHK(dO.alloc(n));
    HIPCHK(dTab.alloc(tab.size())); HIPCHK(dC.alloc(K)); HIPCHK(dMisc.alloc(24));
    if (L) { HIPCHK(dL.alloc(n)); UP(dL.p, L, n); }
    UP(dT.p, Theta, n); UP(dX.p, X, n); UP(dS.p, S, n); UP(dB.p, beta, K); UP(dTab.p, tab.data(), tab.size());
    HIPCHK(hipMemset(dMisc.p, 0, 24 * sizeof(double)));
    int* flag = (int*)dMisc.p;
    int* flag_h = flag + 2;                       // (device memory stands in for the pinned mirror here)
    unsigned long long* units = (unsigned long long*)(dMisc.p + 20);
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    bool ok = launch_omega_lds(nullptr, dT.p, L ? dL.p : nullptr, dX.p, dS.p, dB.p, dO.p, dTab.p, ntab, lnq, K, p, flag,
                               flag_h, 0, units, dC.p, nullptr, waves);
    (void)hipEventRecord(e0, nullptr);
    for (int i = 0; ok && i < iters; ++i)
        launch_omega_lds(nullptr, dT.p, L ? dL.p : nullptr, dX.p, dS.p, dB.p, dO.p, dTab.p, ntab, lnq, K, p, flag, flag_h, 0,
                         nullptr, dC.p, (long long*)(dMisc.p + 4), waves);
    (void)hipEventRecord(e1, nullptr);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (!ok || hipGetLastError() != hipSuccess) return fail(GGL_E_HIP, "LDS Omega-step: launch failed");
    int hflag = 0;
    unsigned long long hu = 0;
    HIPCHK(hipMemcpy(&hflag, flag, sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&hu, units, sizeof(hu), hipMemcpyDeviceToHost));
    out[0] = ms / iters; out[1] = hflag; out[2] = (double)hu; out[3] = ntab;
    {
        // phase stamps of instance 0 (100 MHz wall clock): out[4..12] = us since the kernel's first instruction; out[13] = its products
        long long ts[16];
        HIPCHK(hipMemcpy(ts, dMisc.p + 4, sizeof(ts), hipMemcpyDeviceToHost));
        for (int i = 0; i < 9; ++i) out[4 + i] = (double)(ts[i] - ts[0]) * 0.01;
        out[13] = (double)ts[9];
        out[14] = (double)(ts[14] - ts[10]) * 0.01;                                        // the A' product alone
    }
    DOWN(Omega, dO.p, n);
    if (cbound) DOWN(cbound, dC.p, K);
    return GGL_OK;
}

#ifdef GGL_DEV
// persistent-chain probe (gemm_sym.hip): out = {ms per chain as nprod launches, ms per chain as one cooperative launch,
// grid of the cooperative launch, max |difference| between the two chains' results (same tile code: 0 unless a workgroup
// read stale data across a grid barrier), barrier time-out flag}.  The chain is X <- I - 1.5 X^2 on a dense symmetric
// start of norm <= 1/2 (the quadratic map keeps the spectrum in [-1, 1], so it can run for any number of products).
extern "C" int ggl_dev_chain_probe(int K, int p, int variant, int nprod, int iters, int two_level, double* out)
{
    ARGCHK(K >= 1 && p >= 2 && (p & 1) == 0 && nprod >= 1 && iters >= 1 && out, "arguments (p even)");
    const size_t n = (size_t)K * p * p;
    std::vector<double> h(n, 0.0), coef((size_t)K * NS_NCOEF, 0.0);
    unsigned long long s = 88172645463325252ull;
    for (int k = 0; k < K; ++k) {
        double* M = h.data() + (size_t)k * p * p;
        for (int i = 0; i < p; ++i)
            for (int j = i; j < p; ++j) {
                s ^= s << 13; s ^= s >> 7; s ^= s << 17;
                M[(size_t)i * p + j] = M[(size_t)j * p + i] = ((double)(s >> 11) / 9007199254740992.0 - 0.5) / p;
            }
        coef[(size_t)k * NS_NCOEF + 0] = 1.0;
        coef[(size_t)k * NS_NCOEF + 1] = -1.5;
    }
    DevBuf dX0, dX1, dcoef, dbar;
    HIPCHK(dX0.alloc(n));
    HIPCHK(dX1.alloc(n));
    HIPCHK(dcoef.alloc(coef.size()));
    HIPCHK(dbar.alloc(8));
    UP(dcoef.p, coef.data(), coef.size());
    unsigned* bar = reinterpret_cast<unsigned*>(dbar.p);
    double* last = (nprod & 1) ? dX1.p : dX0.p;
    auto chain_launches = [&]() {
        for (int j = 0; j < nprod; ++j)
            launch_symm(nullptr, (j & 1) ? dX1.p : dX0.p, (j & 1) ? dX1.p : dX0.p, (j & 1) ? dX0.p : dX1.p, nullptr, nullptr,
                        dcoef.p, K, p, variant);
    };
    int grid = 0;
    auto chain_persistent = [&]() -> int {
        hipError_t e = hipMemsetAsync(dbar.p, 0, 8 * sizeof(double), nullptr);
        if (e != hipSuccess) return -1;
        grid = launch_chain_probe(nullptr, dX0.p, dX1.p, dcoef.p, K, p, nprod, variant, bar, bar + 1, two_level);
        return grid;
    };
    // the two chains from the same start must agree bit for bit
    std::vector<double> r1(n), r2(n);
    UP(dX0.p, h.data(), n);
    chain_launches();
    DOWN(r1.data(), last, n);
    UP(dX0.p, h.data(), n);
    HIPCHK(hipMemset(dX1.p, 0, n * sizeof(double)));
    int g = chain_persistent();
    ARGCHK(g != 0, "no probe instance of this variant (16, 17, 20)");
    if (g < 0) return fail(GGL_E_HIP, "cooperative launch of the chain probe failed: %s", hipGetErrorString(hipGetLastError()));
    DOWN(r2.data(), last, n);
    double dev = 0.0;
    for (size_t i = 0; i < n; ++i) dev = std::max(dev, std::fabs(r1[i] - r2[i]));
    out[3] = dev;
    unsigned flags[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpy(flags, dbar.p, sizeof(flags), hipMemcpyDeviceToHost));
    out[4] = flags[1];
    out[2] = grid;
    if (flags[1]) return GGL_OK;          // a barrier timed out: do not time it
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    float ms = 0.f;
    for (int i = 0; i < 3; ++i) chain_launches();
    HIPCHK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) chain_launches();
    HIPCHK(hipEventRecord(e1, nullptr));
    HIPCHK(hipEventSynchronize(e1));
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    out[0] = ms / iters;
    for (int i = 0; i < 3 + iters; ++i) {
        if (i == 3) HIPCHK(hipEventRecord(e0, nullptr));
        if (chain_persistent() <= 0) return fail(GGL_E_HIP, "cooperative launch of the chain probe failed");
    }
    HIPCHK(hipEventRecord(e1, nullptr));
    HIPCHK(hipEventSynchronize(e1));
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    out[1] = ms / iters;
    HIPCHK(hipMemcpy(flags, dbar.p, sizeof(flags), hipMemcpyDeviceToHost));
    out[4] = flags[1];
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    HIPCHK(hipGetLastError());
    return GGL_OK;
}

// k_omega_chain on a synthetic program: nprod dependent products X <- I - 1.5 X^2 (ping-pong between two stacks) as nprod
// launches of the three-stage 64x64 kernel (out[0], ms per chain) and as ONE persistent launch with per-instance
// dependencies (out[1]); out[2] = persistent workgroups, out[3] = max |difference| of the results (same tile code: 0 unless
// a hand-off delivered stale data), out[4] = completion flag of k_chain_check, out[5..5+K) = done counters after the run
extern "C" int ggl_dev_chain_run(int K, int p, int nprod, int iters, double* out)
{
    ARGCHK(K >= 1 && p >= 2 && (p & 1) == 0 && nprod >= 1 && nprod <= CHAIN_MAX_OPS && iters >= 1 && out, "arguments (p even)");
    const size_t n = (size_t)K * p * p;
    std::vector<double> h(n, 0.0), coef((size_t)K * NS_NCOEF, 0.0);
    unsigned long long s = 88172645463325252ull;
    for (int k = 0; k < K; ++k) {
        double* M = h.data() + (size_t)k * p * p;
        for (int i = 0; i < p; ++i)
            for (int j = i; j < p; ++j) {
                s ^= s << 13; s ^= s >> 7; s ^= s << 17;
                M[(size_t)i * p + j] = M[(size_t)j * p + i] = ((double)(s >> 11) / 9007199254740992.0 - 0.5) / p;
            }
        coef[(size_t)k * NS_NCOEF + 0] = 1.0;
        coef[(size_t)k * NS_NCOEF + 1] = -1.5;
    }
    DevBuf dX0, dX1, dcoef, dcnt;
    HIPCHK(dX0.alloc(n));
    HIPCHK(dX1.alloc(n));
    HIPCHK(dcoef.alloc(coef.size()));
    const size_t ncw = (size_t)K * CHAIN_CNT_STRIDE;          // 32-bit words
    HIPCHK(dcnt.alloc(ncw / 2 + 8));
    UP(dcoef.p, coef.data(), coef.size());
    unsigned* cnt = reinterpret_cast<unsigned*>(dcnt.p);
    int* flag = reinterpret_cast<int*>(cnt + ncw);
    const int aux = getenv("GGL_CHAIN_AUX") ? atoi(getenv("GGL_CHAIN_AUX")) : 16;
    double* last = (nprod & 1) ? dX1.p : dX0.p;
    const int T = (p + 63) / 64;
    ChainProg P;
    P.nops = nprod; P.K = K; P.p = p; P.ntiles = T * (T + 1) / 2;
    P.begin[0] = 0;
    for (int j = 0; j < nprod; ++j) {
        SymmOp o{};
        o.A = o.B = (j & 1) ? dX1.p : dX0.p;
        o.C = (j & 1) ? dX0.p : dX1.p;
        o.coef = dcoef.p;
        P.op[j] = o;
        P.begin[j + 1] = P.begin[j] + P.ntiles;
    }
    auto chain_launches = [&]() {
        for (int j = 0; j < nprod; ++j)
            launch_symm(nullptr, (j & 1) ? dX1.p : dX0.p, (j & 1) ? dX1.p : dX0.p, (j & 1) ? dX0.p : dX1.p, nullptr, nullptr,
                        dcoef.p, K, p, 17);
    };
    int grid = 0;
    auto chain_persistent = [&]() -> int {
        if (hipMemsetAsync(dcnt.p, 0, (ncw / 2 + 8) * sizeof(double), nullptr) != hipSuccess) return -1;
        grid = launch_omega_chain(nullptr, P, cnt, flag, flag + 1, aux);
        return grid;
    };
    std::vector<double> r1(n), r2(n);
    UP(dX0.p, h.data(), n);
    chain_launches();
    DOWN(r1.data(), last, n);
    UP(dX0.p, h.data(), n);
    HIPCHK(hipMemset(dX1.p, 0, n * sizeof(double)));
    if (chain_persistent() <= 0) return fail(GGL_E_HIP, "k_omega_chain launch failed");
    DOWN(r2.data(), last, n);
    if (getenv("GGL_CHAIN_PROF")) {
        // one more run with the per-workgroup time accounting: claim / idle / tile time (100 MHz ticks) and tiles served
        DevBuf dprof;
        HIPCHK(dprof.alloc((size_t)grid * 8));
        HIPCHK(hipMemset(dprof.p, 0, (size_t)grid * 8 * sizeof(double)));
        P.prof = reinterpret_cast<long long*>(dprof.p);
        UP(dX0.p, h.data(), n);
        if (chain_persistent() <= 0) return fail(GGL_E_HIP, "k_omega_chain launch failed");
        std::vector<long long> hp((size_t)grid * 8);
        HIPCHK(hipMemcpy(hp.data(), dprof.p, hp.size() * sizeof(long long), hipMemcpyDeviceToHost));
        P.prof = nullptr;
        double cl = 0, id = 0, ti = 0, nt = 0, span = 0, cyc = 0;
        long long t0 = hp[4], t1 = hp[5];
        int perx[8] = {};
        for (int g = 0; g < grid; ++g) {
            const long long* o = hp.data() + (size_t)g * 8;
            cl += o[0]; id += o[1]; ti += o[2]; nt += o[3]; span += o[5] - o[4]; cyc += o[7];
            t0 = std::min(t0, o[4]); t1 = std::max(t1, o[5]);
            perx[o[6] & 7] += 1;
        }
        fprintf(stderr, "chain prof: kernel span %.1f us; per workgroup (avg over %d): claim %.1f us, idle %.1f us, tile %.1f us, "
                "%.2f tiles, %.1f us per tile, alive %.1f us; workgroups per XCD:", (t1 - t0) * 0.01, grid, cl * 0.01 / grid,
                id * 0.01 / grid, ti * 0.01 / grid, nt / grid, ti * 0.01 / std::max(nt, 1.0), span * 0.01 / grid);
        for (int x = 0; x < 8; ++x) fprintf(stderr, " %d", perx[x]);
        fprintf(stderr, "; clock64 ticks per us of wall_clock64 while alive: %.1f\n", cyc / (span * 0.01));
    }
    double dev = 0.0;
    for (size_t i = 0; i < n; ++i) dev = std::max(dev, std::fabs(r1[i] - r2[i]));
    out[3] = dev;
    out[2] = grid;
    std::vector<unsigned> hc(ncw + 2);
    HIPCHK(hipMemcpy(hc.data(), dcnt.p, hc.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    out[4] = (double)hc[ncw];
    for (int k = 0; k < K; ++k) out[5 + k] = (double)hc[(size_t)k * CHAIN_CNT_STRIDE];
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    float ms = 0.f;
    for (int i = 0; i < 3; ++i) chain_launches();
    HIPCHK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) chain_launches();
    HIPCHK(hipEventRecord(e1, nullptr));
    HIPCHK(hipEventSynchronize(e1));
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    out[0] = ms / iters;
    for (int i = 0; i < 3 + iters; ++i) {
        if (i == 3) HIPCHK(hipEventRecord(e0, nullptr));
        if (chain_persistent() <= 0) return fail(GGL_E_HIP, "k_omega_chain launch failed");
    }
    HIPCHK(hipEventRecord(e1, nullptr));
    HIPCHK(hipEventSynchronize(e1));
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    out[1] = ms / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    HIPCHK(hipGetLastError());
    return GGL_OK;
}

// timeline probe: one launch of variant 10; out = [nblocks][5] long long {start, loop, loop_end, end, xcc}
extern "C" int ggl_dev_symm_timeline(int K, int p, long long* out, int max_blocks, int* nblocks_out)
{
    ARGCHK(K >= 1 && p >= 1 && out && nblocks_out, "arguments");
    const size_t n = (size_t)K * p * p;
    std::vector<double> h(n, 0.25), coef((size_t)K * NS_NCOEF, 0.0);
    for (int k = 0; k < K; ++k) coef[(size_t)k * NS_NCOEF + 1] = 1.0 / p;
    const int T = (p + 63) / 64;
    const int nb = (K >= 8 ? 8 * ((K + 7) / 8) : K) * (T * (T + 1) / 2);
    ARGCHK(nb <= max_blocks, "max_blocks too small");
    DevBuf dA, dB, dC, dcoef, dT;
    HIPCHK(dA.alloc(n));
    HIPCHK(dB.alloc(n));
    HIPCHK(dC.alloc(n));
    HIPCHK(dcoef.alloc(coef.size()));
    HIPCHK(dT.alloc((size_t)nb * 5));
    UP(dA.p, h.data(), n);
    UP(dB.p, h.data(), n);
    UP(dcoef.p, coef.data(), coef.size());
    HIPCHK(hipMemset(dT.p, 0, (size_t)nb * 5 * sizeof(double)));
    for (int i = 0; i < 3; ++i) launch_symm(nullptr, dA.p, dB.p, dC.p, nullptr, nullptr, dcoef.p, K, p, 0);
    launch_symm(nullptr, dA.p, dB.p, dC.p, nullptr, nullptr, dcoef.p, K, p, 10, dT.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, dT.p, (size_t)nb * 5 * sizeof(long long), hipMemcpyDeviceToHost));
    *nblocks_out = nb;
    return GGL_OK;
}

#endif   // GGL_DEV

extern "C" int ggl_dev_ns_schedule(double l, int degrees, int max_steps, int* deg_out, double* coef_out, int* units_out)
{
    ARGCHK(deg_out && coef_out && units_out, "output pointers");
    ARGCHK(l > 0.0 && l <= 1.0, "l must be in (0,1]");
    const int n = ns_schedule_query(l, degrees, max_steps, deg_out, coef_out, units_out);
    if (n < 0) return fail(GGL_E_ARG, "no schedule within %d steps", max_steps);
    return n;
}

// host only: the grouping rule of GGL_OPT_GROUP_SCHED (ns_group_partition) and the product units of a schedule
extern "C" int ggl_dev_group_partition(const int* units, int K, int p, int max_groups, int* len_out)
{
    ARGCHK(units && len_out && K >= 1 && p >= 1 && max_groups >= 1 && max_groups <= 3, "units, len_out, K, p, 1 <= max_groups <= 3");
    return ns_group_partition(units, K, p, max_groups, len_out);
}
extern "C" int ggl_dev_ns_units(double l, int degrees, double tol) { return ns_units_query(l, degrees, tol); }

extern "C" int ggl_dev_ns_schedule_tol(double l, int degrees, double tol, int max_steps, int* deg_out, double* coef_out,
                                       int* units_out)
{
    ARGCHK(deg_out && coef_out && units_out, "output pointers");
    ARGCHK(l > 0.0 && l <= 1.0, "l must be in (0,1]");
    ARGCHK(tol >= 0.0 && tol <= 1e-6, "tol in [0, 1e-6]");
    const int n = ns_schedule_query(l, degrees, max_steps, deg_out, coef_out, units_out, std::max(tol, NS_TOL_EXACT));
    if (n < 0) return fail(GGL_E_ARG, "no schedule within %d steps", max_steps);
    return n;
}

#ifdef GGL_DEV
extern "C" int ggl_dev_coissue_probe(double* out12)
{
    ARGCHK(out12, "out");
    DevBuf d;
    HIPCHK(d.alloc((size_t)512 * 512));
    coissue_probe(nullptr, d.p, out12);
    HIPCHK(hipGetLastError());
    return GGL_OK;
}

extern "C" int ggl_dev_mfma_lds_probe(double* out6)
{
    ARGCHK(out6, "out");
    DevBuf d;
    HIPCHK(d.alloc((size_t)2048 * 256));
    mfma_lds_probe(nullptr, d.p, out6);
    HIPCHK(hipGetLastError());
    return GGL_OK;
}

extern "C" int ggl_dev_mfma_f64_peak(double* tflops_out)
{
    ARGCHK(tflops_out, "tflops_out");
    const int blocks = 256 * 8;
    DevBuf d;
    HIPCHK(d.alloc((size_t)blocks * 256));
    double best = 0.0;
    for (int r = 0; r < 3; ++r) best = std::max(best, mfma_f64_peak_tflops(nullptr, d.p, blocks, 2000, 8));
    HIPCHK(hipGetLastError());
    if (getenv("GGL_MFMA_PROBE_VERBOSE")) {
        for (int layers : {1, 2, 4, 8})
            for (int nacc : {1, 2, 4, 8})
                fprintf(stderr, "mfma f64 probe: %d wave(s)/SIMD, %d accumulators: %.1f TF/s\n", layers, nacc,
                        mfma_f64_peak_tflops(nullptr, d.p, 256 * layers, 4000, nacc));
    }
    if (getenv("GGL_MFMA_MIX_VERBOSE")) {
        for (int layers : {1, 4})
            for (int nv : {0, 2, 4, 8, 16})
                fprintf(stderr, "mfma+valu mix: %d wave(s)/SIMD, %2d VALU per MFMA: %.1f TF/s\n", layers, nv,
                        mfma_valu_mix_tflops(nullptr, d.p, 256 * layers, 4000, nv));
    }
    *tflops_out = best;
    return GGL_OK;
}
#endif   // GGL_DEV

// the eigensolver selector of the stateless entry points (low byte; GGL_EIG_NS_MODE / _DEGREES above it are checked by the ctx)
static bool eig_selector_ok(int eig_method)
{
    const int e = eig_method & 0xff;
    return e == GGL_EIG_AUTO || e == GGL_EIG_JACOBI || e == GGL_EIG_ROCSOLVER || e == GGL_EIG_NEWTON_SCHULZ;
}
// beta of the log-det prox (n_k / rho, admm_solver.py:180-187): positive and finite for every instance
static bool betas_ok(const double* beta, int K)
{
    for (int k = 0; k < K; ++k)
        if (!(beta[k] > 0.0) || !std::isfinite(beta[k])) return false;
    return true;
}

extern "C" int ggl_eigh_batched(int K, int p, const double* A, double* D, double* Q, int eig_method)
{
    ARGCHK(D, "D");
    ARGCHK(eig_selector_ok(eig_method), "eigensolver selector");
    return eig_common(K, p, A, nullptr, D, Q, nullptr, MAP_IDENT, eig_method & 0xff);
}

extern "C" int ggl_phiplus_matrix(int K, int p, const double* beta, const double* W, double* out, int eig_method)
{
    ARGCHK(beta && out && W, "beta, W, out");
    ARGCHK(K >= 1 && p >= 1, "K, p");
    ARGCHK(eig_selector_ok(eig_method), "eigensolver selector");
    ARGCHK(betas_ok(beta, K), "beta must be positive and finite");
    if (use_ns(eig_method & 0xff, p)) {
        // run the Omega-step of a scratch ctx with Theta = W, X = S = 0, nk = beta, rho = 1
        ggl_ctx* c = nullptr;
        int rc = ggl_ctx_create(0, K, p, (eig_method & ~0xff) | GGL_EIG_NEWTON_SCHULZ, nullptr, &c);
        if (rc) return rc;
        c->ns_tol = NS_TOL_EXACT;          // the operator-level entry point iterates to fp64 resolution
        std::vector<double> zero((size_t)K * p * p, 0.0);
        rc = ggl_set_S(c, zero.data());
        if (!rc) rc = ggl_set_state(c, zero.data(), W, nullptr, zero.data());
        if (!rc) rc = ggl_step_omega(c, 1.0, 0, beta);
        if (!rc) rc = ggl_get_state(c, out, nullptr, nullptr, nullptr);
        ggl_ctx_destroy(c);
        return rc;
    }
    return eig_common(K, p, W, beta, nullptr, nullptr, out, MAP_PHIPLUS, eig_method & 0xff);
}

static int rank_matrix_impl(int K, int p, const double* beta, const double* C, double* out, int eig_method, double l0_coarse,
                            double l0_deflate, long long* stats, int nstats);

// l0_coarse >= 0: the two-tier iteration at that first-pass resolution, WITHOUT the deflation (0: one tier); < 0: the ctx
// defaults (deflation after a first pass at GGL_OPT_RANK_L0_DEFLATE)
extern "C" int ggl_rank_matrix_ex(int K, int p, const double* beta, const double* C, double* out, int eig_method,
                                  double l0_coarse, long long stats[6])
{
    return rank_matrix_impl(K, p, beta, C, out, eig_method, l0_coarse, l0_coarse >= 0.0 ? 0.0 : -1.0, stats, 6);
}

// the deflating L-step with its first-pass resolution exposed (<= 0: the default); stats[8] = the six of ggl_rank_matrix_ex +
// { calls followed by the deflation, instances that had directions to deflate }
extern "C" int ggl_rank_matrix_deflate(int K, int p, const double* beta, const double* C, double* out, int eig_method,
                                       double l0_deflate, long long stats[8])
{
    return rank_matrix_impl(K, p, beta, C, out, eig_method, -1.0, l0_deflate > 0.0 ? l0_deflate : -1.0, stats, 8);
}

static int rank_matrix_impl(int K, int p, const double* beta, const double* C, double* out, int eig_method, double l0_coarse,
                            double l0_deflate, long long* stats, int nstats)
{
    ARGCHK(beta && out && C, "beta, C, out");
    ARGCHK(K >= 1 && p >= 1, "K, p");
    ARGCHK(eig_selector_ok(eig_method), "eigensolver selector");
    if (stats) for (int i = 0; i < nstats; ++i) stats[i] = 0;
    if (use_ns(eig_method & 0xff, p)) {
        // the L-step of a scratch ctx: C into the work stack, beta into the mu/rho parameter slot
        ggl_ctx* c = nullptr;
        int rc = ggl_ctx_create(0, K, p, (eig_method & ~0xff) | GGL_EIG_NEWTON_SCHULZ, nullptr, &c);
        if (rc) return rc;
        if (l0_coarse >= 0.0) c->rank_l0_coarse = l0_coarse;
        if (l0_deflate == 0.0) c->rank_deflate = false;
        else if (l0_deflate > 0.0) c->rank_l0_deflate = l0_deflate;
        hipError_t e = hipMemcpyAsync(c->W, C, c->n * sizeof(double), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            rc = upload_par(c, 2, beta, 0.0, 1.0);
            if (!rc) rc = rank_step(c);
            if (!rc) e = hipMemcpyAsync(out, c->L, c->n * sizeof(double), hipMemcpyDeviceToHost, c->stream);
            if (!rc && e == hipSuccess) e = hipStreamSynchronize(c->stream);
        }
        if (stats) {
            stats[0] = c->rank_calls; stats[1] = c->rank_continued; stats[2] = c->rank_cont_instances;
            stats[3] = c->rank_fallbacks; stats[4] = c->rank_retries; stats[5] = c->rank_launches;
            if (nstats >= 8) { stats[6] = c->rank_deflated_calls; stats[7] = c->rank_deflated_instances; }
        }
        ggl_ctx_destroy(c);
        if (e != hipSuccess) return fail(GGL_E_HIP, "ggl_rank_matrix: %s", hipGetErrorString(e));
        return rc;
    }
    return eig_common(K, p, C, beta, nullptr, nullptr, out, MAP_RANK, eig_method & 0xff);
}

extern "C" int ggl_rank_matrix(int K, int p, const double* beta, const double* C, double* out, int eig_method)
{
    return ggl_rank_matrix_ex(K, p, beta, C, out, eig_method, -1.0, nullptr);
}

static int recon_common(int K, int p, const double* beta, const double* D, const double* Q, double* out, int map)
{
    ARGCHK(K >= 1 && p >= 1 && beta && D && Q && out, "arguments");
    ARGCHK(map != MAP_PHIPLUS || betas_ok(beta, K), "beta must be positive and finite");
    const size_t n = (size_t)K * p * p, kp = (size_t)K * p;
    // Q has eigenvectors in columns; the kernel wants them in rows
    std::vector<double> R(n);
    for (int k = 0; k < K; ++k)
        for (int i = 0; i < p; ++i)
            for (int m = 0; m < p; ++m) R[(size_t)k * p * p + (size_t)m * p + i] = Q[(size_t)k * p * p + (size_t)i * p + m];
    DevBuf dR, dD, dB, dO, dS;
    HIPCHK(dR.alloc(n));
    HIPCHK(dD.alloc(kp));
    HIPCHK(dB.alloc(K));
    HIPCHK(dO.alloc(n));
    HIPCHK(dS.alloc(2 * kp));
    UP(dR.p, R.data(), n);
    UP(dD.p, D, kp);
    UP(dB.p, beta, K);
    launch_recon(nullptr, dO.p, dR.p, dD.p, dB.p, map, K, p, dS.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    DOWN(out, dO.p, n);
    return GGL_OK;
}

extern "C" int ggl_phiplus(int K, int p, const double* beta, const double* D, const double* Q, double* out)
{
    return recon_common(K, p, beta, D, Q, out, MAP_PHIPLUS);
}

extern "C" int ggl_prox_rank_norm(int K, int p, const double* beta, const double* D, const double* Q, double* out)
{
    return recon_common(K, p, beta, D, Q, out, MAP_RANK);
}

extern "C" int ggl_prox_od_1norm(int p, const double* A, double lam, const double* lam_pp, double* out)
{
    ARGCHK(p >= 1 && A && out, "arguments");
    const size_t n = (size_t)p * p;
    DevBuf dA, dM, dO;
    HIPCHK(dA.alloc(n));
    HIPCHK(dO.alloc(n));
    UP(dA.p, A, n);
    if (lam_pp) { HIPCHK(dM.alloc(n)); UP(dM.p, lam_pp, n); }
    launch_prox_od(nullptr, dO.p, dA.p, lam, lam_pp ? dM.p : nullptr, p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    DOWN(out, dO.p, n);
    return GGL_OK;
}

extern "C" int ggl_prox_p(int K, int p, const double* X, double l1, double l2, int reg, double* out)
{
    ARGCHK(K >= 1 && p >= 1 && X && out, "arguments");
    ARGCHK(reg == GGL_REG_GGL || reg == GGL_REG_FGL, "reg");
    ARGCHK(l1 > 0 && l2 > 0, "lambda 1 and lambda2 have to be positive");
    if (reg == GGL_REG_FGL && K > fgl_max_K())
        return fail(GGL_E_ARG, "prox_p (FGL): K = %d exceeds the %d instances the LDS scan buffer holds", K, fgl_max_K());
    const size_t n = (size_t)K * p * p;
    DevBuf dX, dO, dW;
    HIPCHK(dX.alloc(n));
    HIPCHK(dO.alloc(n));
    HIPCHK(dW.alloc((size_t)ggl_chunks(K, p) * p * p));
    UP(dX.p, X, n);
    HIPCHK(launch_prox_p(nullptr, reg, dO.p, dX.p, l1, l2, K, p, dW.p));
    HIPCHK(hipDeviceSynchronize());
    DOWN(out, dO.p, n);
    return GGL_OK;
}

static int vec_common(int mode, int n, int K, const double* Y, double l1, double l2, double* out)
{
    ARGCHK(n >= 1 && K >= 1 && Y && out, "arguments");
    const size_t tot = (size_t)n * K;
    DevBuf dY, dO;
    HIPCHK(dY.alloc(tot));
    HIPCHK(dO.alloc(tot));
    UP(dY.p, Y, tot);
    HIPCHK(launch_vec_prox(nullptr, mode, dY.p, dO.p, n, K, l1, l2));
    HIPCHK(hipDeviceSynchronize());
    DOWN(out, dO.p, tot);
    return GGL_OK;
}

extern "C" int ggl_prox_tv(int n, int K, const double* Y, double lam, double* out)
{
    return vec_common(0, n, K, Y, lam, 0.0, out);
}

extern "C" int ggl_prox_2norm(int n, int K, const double* Y, double lam, double* out)
{
    return vec_common(1, n, K, Y, lam, 0.0, out);
}

extern "C" int ggl_prox_phi(int n, int K, const double* Y, double l1, double l2, int reg, double* out)
{
    ARGCHK(reg == GGL_REG_GGL || reg == GGL_REG_FGL, "reg");
    return vec_common(reg == GGL_REG_GGL ? 2 : 3, n, K, Y, l1, l2, out);
}

