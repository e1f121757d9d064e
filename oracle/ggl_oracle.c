/*
 * CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/ggl_oracle.py for the rules).
 *
 * Plain-C restatement of the Numba-jitted half of the reference's Theta-step, so that the
 * checker and the cpu_baseline leg of bench.py are not timed on interpreted Python loops
 * (the real reference runs these bodies through numba.njit):
 *   prox_1norm      src/gglasso/solver/ggl_helper.py:12-14
 *   prox_2norm      src/gglasso/solver/ggl_helper.py:38-43
 *   prox_phi_ggl    src/gglasso/solver/ggl_helper.py:68-71
 *   condat_method   src/gglasso/solver/fgl_helper.py:11-68
 *   prox_phi_fgl    src/gglasso/solver/ggl_helper.py:131-134
 *   prox_p          src/gglasso/solver/ggl_helper.py:190-207
 * Pinned against tests/golden/ (G4-G6) by tests/test_oracle_golden.py.
 *
 * Build: oracle/build.sh  ->  oracle/libggl_oracle.so
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static double soft(double v, double l)
{
    double a = fabs(v) - l;
    if (a < 0.0) a = 0.0;
    return v > 0.0 ? a : (v < 0.0 ? -a : 0.0);
}

/* fgl_helper.py:11-68 -- y[0..N), x[0..N) distinct buffers. */
void oracle_condat(const double *y, double *x, int N, double lam)
{
    int k = 0, k0 = 0, kplus = 0, kminus = 0, i;
    double vmin = y[0] - lam, vmax = y[0] + lam, umin = lam, umax = -lam;
    for (;;) {
        while (k == N - 1) {
            if (umin < 0.0) {
                for (i = k0; i <= kminus; ++i) x[i] = vmin;
                kminus += 1;
                k = k0 = kminus;
                umin = lam; vmin = y[k]; umax = y[k] + lam - vmax;
            } else if (umax > 0.0) {
                for (i = k0; i <= kplus; ++i) x[i] = vmax;
                kplus += 1;
                k = k0 = kplus;
                umax = -lam; vmax = y[k]; umin = y[k] - lam - vmin;
            } else {
                double v = vmin + umin / (double)(k - k0 + 1);
                for (i = k0; i < N; ++i) x[i] = v;
                return;
            }
            if (k == N - 1) {
                x[k] = vmin + umin;
                return;
            }
        }
        if (y[k + 1] + umin - vmin < -lam) {
            for (i = k0; i <= kminus; ++i) x[i] = vmin;
            kminus += 1;
            k = kplus = k0 = kminus;
            vmin = y[k]; vmax = y[k] + 2 * lam;
            umin = lam; umax = -lam;
        } else if (y[k + 1] + umax - vmax > lam) {
            for (i = k0; i <= kplus; ++i) x[i] = vmax;
            kplus += 1;
            k = kminus = k0 = kplus;
            vmin = y[k] - 2 * lam; vmax = y[k];
            umin = lam; umax = -lam;
        } else {
            k += 1;
            umin = umin + y[k] - vmin;
            umax = umax + y[k] - vmax;
            if (umin >= lam) { vmin += (umin - lam) / (double)(k - k0 + 1); umin = lam; kminus = k; }
            if (umax <= -lam) { vmax += (umax + lam) / (double)(k - k0 + 1); umax = -lam; kplus = k; }
        }
    }
}

/* ggl_helper.py:190-207.  reg: 1 = GGL, 2 = FGL.  X, out: (K,p,p) row-major. */
int oracle_prox_p(const double *X, double *out, int K, int p, double l1, double l2, int reg)
{
    size_t pp = (size_t)p * p;
    double *v = (double *)malloc(sizeof(double) * 2 * (size_t)K);
    double *t = v + K;
    int i, j, k;
    if (!v) return -1;
    for (i = 0; i < p; ++i) {
        for (k = 0; k < K; ++k) out[k * pp + (size_t)i * p + i] = X[k * pp + (size_t)i * p + i];
        for (j = i + 1; j < p; ++j) {
            for (k = 0; k < K; ++k) v[k] = X[k * pp + (size_t)i * p + j];
            if (reg == 1) {
                double ss = 0.0, a;
                for (k = 0; k < K; ++k) { v[k] = soft(v[k], l1); ss += v[k] * v[k]; }
                a = sqrt(ss);
                if (a < l2) a = l2;
                for (k = 0; k < K; ++k) t[k] = v[k] * (a - l2) / a;
            } else {
                oracle_condat(v, t, K, l2);
                for (k = 0; k < K; ++k) t[k] = soft(t[k], l1);
            }
            for (k = 0; k < K; ++k) {
                out[k * pp + (size_t)i * p + j] = t[k];
                out[k * pp + (size_t)j * p + i] = t[k];
            }
        }
    }
    free(v);
    return 0;
}
