// Host-side launchers of the gfx950 kernels behind libggl_hip's C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#define GGL_NNORM 5   // |Omega|^2, |Theta-L|^2, |X|^2, |Omega-Theta+L|^2, |Omega-Omega_prev|^2

namespace ggl {

// eigenvalue maps applied inside reconstruction kernels
enum EigMap {
    MAP_PHIPLUS = 0,  // 0.5*(d + sqrt(d^2 + 4 beta))   solver/ggl_helper.py:272-274
    MAP_RANK = 1,     // max(d - beta, 0)                solver/ggl_helper.py:35
    MAP_IDENT = 2     // d (plain Q diag(d) Q^T, tests)
};

// ---- elementwise.hip --------------------------------------------------------------------
// number of partial-sum slots per instance the elementwise kernels produce for (p)
int elementwise_blocks(int p);
// W = ((Theta - L) - X) - beta_k * S        (admm_solver.py:180); L may be null (== 0)
void launch_form_W(hipStream_t st, double* W, const double* Theta, const double* L, const double* X,
                   const double* S, const double* betaK, int K, int p);
// SGL Theta-step (prox_od_1norm, ggl_helper.py:16-27) on every instance k with threshold
// l1K[k] (scalar) or invrhoK[k]*mask[i,j].  latent == 0: also X += Omega - Theta and the norms
// partials [K][nblk][5].  latent != 0: writes Theta and C = (Theta - X) - Omega.
void launch_theta_sgl(hipStream_t st, double* Theta, double* X, double* C, const double* Omega,
                      const double* OmegaPrev, const double* L, const double* l1K,
                      const double* mask, const double* invrhoK, int latent, double* partials, int K, int p, const int* skip = nullptr,
                      const int* pk = nullptr, size_t mask_stride = 0);     // pk: sums over the leading (pk[k],pk[k]) blocks only
// X += (Omega - Theta) + L and the norms partials (latent path, admm_solver.py:208)
void launch_dual_update(hipStream_t st, double* X, const double* Omega, const double* OmegaPrev,
                        const double* Theta, const double* L, double* partials, int K, int p);
// out[k][v] = sum_b partials[k][b][v]   (fixed order => deterministic)
// seq (optional): pinned word that receives seq_val after the sums have been written; with K > 1 rows it needs `arrive`,
// one device word that is zero between launches (the last workgroup to arrive publishes)
void launch_reduce_partials(hipStream_t st, const double* partials, int K, int nblk, int nv, double* out,
                            unsigned long long* seq = nullptr, unsigned long long seq_val = 0, unsigned* arrive = nullptr);
void launch_scale(hipStream_t st, double* X, double f, size_t n);
// Small transfers between pinned host memory (device-visible) and HBM as an ordinary kernel in the stream:
// a hipMemcpyAsync of a few KB costs ~15 us of queue idle time around its blit (measured, rocprofv3), this
// costs a launch.  Up to 10 segments of 32-bit words per launch; src == nullptr zero-fills.
struct CopySegs {
    static constexpr int MAX = 12;
    int n = 0;
    void* dst[MAX];
    const void* src[MAX];
    unsigned words[MAX];
    void add(void* d, const void* s, size_t bytes) { dst[n] = d; src[n] = s; words[n] = (unsigned)(bytes / 4); ++n; }
};
void launch_copy_small(hipStream_t st, const CopySegs& segs);
// dst[0..n) = src[0..n) (src null: zeros) as an ordinary kernel in the stream
void launch_copy_block(hipStream_t st, double* dst, const double* src, size_t n);
void launch_copy_small_seq(hipStream_t st, const CopySegs& segs, unsigned long long* seq, unsigned long long seq_val);
void launch_spin_us(hipStream_t st, int us);
// join of concurrent launch sequences through memory instead of a cross-queue event (elementwise.hip)
void launch_set_flag(hipStream_t st, unsigned long long* f, unsigned long long v);
void launch_wait_flags(hipStream_t st, const unsigned long long* f, int n, unsigned long long v, int* skip, int* skip_host,
                       int slot, double timeout_ms);      // one idle wave for `us` microseconds (stream-concurrency probe)
// X[k] *= fK[k]
void launch_scale_batch(hipStream_t st, double* X, const double* fK, int K, int p);
// out[k] = max_{i,j} |A[k,i,j] - A[k,j,i]|
void launch_asym_max(hipStream_t st, const double* A, int K, int p, double* out);
// partials[K][elementwise_blocks(p)]: non-zero entries per chunk of every instance
void launch_count_nonzero(hipStream_t st, const double* A, int K, int p, double* partials);
// dst[i] = src[idx[i]] (gather) or dst[idx[i]] = src[i] (scatter) for m instances of pp doubles
void launch_set_identity(hipStream_t st, double* A, int K, int p);
void launch_add_diag(hipStream_t st, double* A, int K, int p, double shift);
void launch_get_diag(hipStream_t st, const double* A, int K, int p, double* d);
void launch_copy_instances(hipStream_t st, double* dst, const double* src, const int* idx, int m, size_t pp, bool scatter);
// out[k] = trace(A_k) - shift, summed in a fixed order
void launch_trace(hipStream_t st, const double* A, int K, int p, double shift, double* out);
// D = A - B (B may be null)
void launch_sub(hipStream_t st, double* D, const double* A, const double* B, size_t n);
// out[k][0] = sum A*B
void launch_dot(hipStream_t st, const double* A, const double* B, int K, int p, double* partials);
// thresholded estimates (helper/model_selection.py:698-705): per slot s, instance src[s] at threshold tauS[s];
// sums: partials[s][block][2] = {<S,T>, count_nonzero(T)}; write: out[s] = T
void launch_threshold_sums(hipStream_t st, const double* A, const double* S, const int* src, const double* tauS,
                           int nslot, int p, double* partials);
void launch_threshold_write(hipStream_t st, const double* A, const int* src, const double* tauS, int nslot, int p,
                            double* out);
// out = A + c * B
void launch_axpy(hipStream_t st, double* out, const double* A, double c, const double* B, size_t n);
// out = (Omega - nk_k * S) - rho * X      (argument of the log-det prox in the KKT residual)
void launch_kkt_w(hipStream_t st, double* out, const double* Omega, const double* S, const double* X,
                  const double* nkK, double rho, int K, int p);
// partials[k][b][0] = sum_chunk (A - B)^2   (B may be null)
void launch_sqdiff(hipStream_t st, const double* A, const double* B, int K, int p, double* partials);
// plain elementwise prox_od_1norm for the operator entry point
void launch_prox_od(hipStream_t st, double* out, const double* A, double lam, const double* lam_pp, int p);

// ---- theta_pair.hip ---------------------------------------------------------------------
int pair_blocks(int p, int reg, int K);
// flat != 0 and K <= GGL_FLAT_MAX_K (GGL only): one thread per element with its K-column in registers; valid for an
// exactly symmetric state only.  theta_partial_blocks: norms partial rows written by launch_theta_pair.
static constexpr int GGL_FLAT_MAX_K = 256;     // K-column over up to 16 waves of 16 instances each (theta_pair.hip, launch_flat4_any)
int theta_partial_blocks(int p, int reg, int K, int flat, int G = 1);      // G: problems in one launch (batched grid)
// GGL/FGL Theta-step on upper-triangle K-vectors (prox_p, ggl_helper.py:190-207), mirrored.
//   fuse_dual != 0 (non-latent): also X += Omega - Theta and norms partials [K][nblk][5].
//   fuse_dual == 0 (latent): writes Theta and C = (Theta - X) - Omega  (admm_solver.py:198).
// groupsq != null (GGL only): use this PACKED upper triangle (+ trailing flag: a set flag makes the kernel leave the iterate
// alone) as sum_k u^2 instead of computing it (K-sharded).
// Returns hipErrorInvalidValue if K is beyond what the FGL kernel's LDS scan buffer holds.
// sqwork: ggl_chunks(K,p)*p*p doubles of scratch for the GGL sums of squares (unused when groupsq is given).
// wn (fuse_dual, GGL, per-element kernels only): S and beta_k of the NEXT Omega-step -- the kernel then also writes
// C = (Theta_new - X_new) - beta_k S, that step's W (admm_solver.py:180; k_form_W_sym's arithmetic per element, which equals
// its "lower triangle, mirrored" for a bitwise symmetric state and S), saving the k_form_W_sym pass and launch;
// *wn_done = 1 when the launched kernel wrote it.
struct WNext { const double* S = nullptr; const double* beta = nullptr; };
hipError_t launch_theta_pair(hipStream_t st, int reg, double* Theta, double* X, double* C,
                             const double* Omega, const double* OmegaPrev, const double* L,
                             double l1, double l2, const double* groupsq, double* sqwork, int fuse_dual,
                             double* partials, int K, int p, int flat = 0, const int* skip = nullptr, WNext wn = WNext(),
                             int* wn_done = nullptr);
// largest K the FGL Theta-step kernel serves (K-vectors of an 8x8 tile pair in one workgroup's LDS)
int fgl_max_K();
// G independent problems of K instances each in one launch: stacks (G*K,p,p), thresholds of problem g at l1G[g*K] /
// l2G[g*K], partial sums [g][theta_partial_blocks(p,reg,K,2)].  GGL: K <= GGL_FLAT_MAX_K; symmetric states only.
hipError_t launch_theta_batch(hipStream_t st, int reg, double* Theta, double* X, double* C, const double* Omega,
                              const double* OmegaPrev, const double* L, const double* l1G, const double* l2G,
                              int fuse_dual, double* partials, int G, int K, int p, const int* skip);
// number of K-chunks the GGL kernels split the stack into (grid.y)
int ggl_chunks(int K, int p);
// GGL pass 1 only: sq[c](p,p)[i<j] = sum_{k in chunk c} soft(Omega+L+X, l1)^2,  c < ggl_chunks(K,p)
void launch_group_partial(hipStream_t st, double* sq, const double* Omega, const double* L,
                          const double* X, double l1, int K, int p);
// out(p,p) = sum_c sq[c], mirrored into the FULL symmetric matrix (diagonal 0)
void launch_sum_chunks(hipStream_t st, double* out, const double* sq, int nsq, int p, const int* flags);
// GROUPSQ of a K-sharded run (theta_pair.hip, capi_comm.hip): the sums of squares sum_k u^2 are symmetric, so what the ranks
// exchange is the PACKED upper triangle, row-major with the diagonal -- p (p + 1) / 2 doubles -- followed by one double for
// the speculation flag.  Element (i,j) in either order:
__host__ __device__ __forceinline__ size_t tri_index(int i, int j, int p)
{
    const int a = i < j ? i : j, b = i < j ? j : i;
    return (size_t)a * p - ((size_t)a * (a - 1)) / 2 + (size_t)(b - a);
}
__host__ __device__ __forceinline__ size_t tri_len(int p) { return (size_t)p * (p + 1) / 2; }
// GROUPSQ of a K-sharded run: the PACKED upper triangle (tri_index, common.hpp) of this rank's sum_k soft(Omega+L+X, l1)^2
// followed by this rank's speculation flag (flags: the validation flags of the step's parts, or null): tri_len(p) + 1
// doubles, one launch (per-element kernel for K <= GGL_FLAT_MAX_K, tile pairs + launch_sum_chunks beyond; sqwork:
// ggl_chunks(K,p)*p*p doubles)
void launch_group_sums_packed(hipStream_t st, double* out, double* sqwork, const double* Omega, const double* L,
                              const double* X, double l1, int K, int p, const int* flags);
// stateless prox_p: out = prox_p(V)   (V symmetric stack; upper triangle decides)
hipError_t launch_prox_p(hipStream_t st, int reg, double* out, const double* V, double l1, double l2,
                         int K, int p, double* sqwork);
// P_val partials (ggl_helper.py:162-176): partials[b] per block, nblk = pval_blocks(p)
int pval_blocks(int p);
void launch_pval(hipStream_t st, int reg, const double* Theta, double l1, double l2, int K, int p,
                 double* partials);
// n independent K-vectors (n,K) row-major: mode 0 prox_tv, 1 prox_2norm, 2 prox_phi_ggl, 3 prox_phi_fgl
hipError_t launch_vec_prox(hipStream_t st, int mode, const double* Y, double* out, int n, int K,
                           double l1, double l2);

// ---- ext_group.hip (ext_ADMM_MGL, solver/ext_admm_solver.py) ------------------------------
int ext_blocks(int p);
// Theta-step + Z = Theta + X1 (into Znew) [+ X0 update and its sums when not latent, C = Theta - X0 - Omega when latent]
void launch_ext_theta(hipStream_t st, double* Theta, double* X0, double* Znew, double* C, const double* Omega,
                      const double* OmegaPrev, const double* L, const double* Lambda, const double* X1, const double* l1K,
                      const int* pk, int latent, double* partials, int K, int p, const int* skip);
// prox_2norm_G in place on Lam (holding Z): Gt [2][K][L] int32 (-1 = pair absent), gsize [L]
// nprob independent problems of K instances each with the same group structure (stack slots g*K ..); l2K: device, one
// threshold per instance slot (lambda2/rho of the slot's problem)
void launch_ext_group(hipStream_t st, double* Lam, const int* Gt, const int* gsize, const double* l2K, int L, int K, int p,
                      const int* skip, int nprob = 1);
// X1 += Theta - Lambda [latent: X0 += Omega - Theta + L] and the remaining stopping-test sums
void launch_ext_dual(hipStream_t st, double* X0, double* X1, const double* Omega, const double* OmegaPrev,
                     const double* Theta, const double* L, const double* Lam, const double* LamPrev, const int* pk,
                     int latent, double* partials, int K, int p, const int* skip);
void launch_lin3(hipStream_t st, double* out, double a, const double* A, double b, const double* B, double c,
                 const double* C, size_t n);
// partials[k][ext_blocks(p)] of |(A - B) + C|^2 over the leading (pk,pk) block of every instance
void launch_ext_sq(hipStream_t st, const double* A, const double* B, const double* C, const int* pk, int K, int p,
                   double* partials);
void launch_ext_prox_od(hipStream_t st, double* out, const double* A, const double* l1K, int K, int p);

// ---- eig_jacobi.hip ---------------------------------------------------------------------
bool jacobi_fits(int p);
int theta_last_kernel();
// deflate.hip: L-step, the eigenvalues next to the threshold deflated after a coarse sign iteration
static constexpr int DEFL_Q = 8;      // columns of the range finder: DEFL_Q0 for the basis + probes
static constexpr int DEFL_Q0 = 6;
void launch_deflate(hipStream_t st, const double* X, const double* C, const double* muK, double* L, const double* G, double* work,
                    double* meta, int K, int p, double tau1, double tau2);
int deflate_max_p();

// ---- the Omega-step of a small matrix as one launch, one workgroup per instance, the chain resident in LDS (omega_lds.hip) ----
static constexpr int OMEGA_LDS_MAXSTEP = 6;      // Newton-Schulz steps of a table entry's schedule (cubic-only schedules need five)
static constexpr int OMEGA_LDS_ENT = 8 + 6 * OMEGA_LDS_MAXSTEP;   // doubles per table entry {n, deg[6], -, coef[6][6]}
static constexpr int OMEGA_LDS_MAXTAB = 160;     // q^-idx with q = 1.02 down to l = 300^-1/2: 145 entries
int omega_lds_max_p();
// units: two counters {products, steps} summed over the instances (or null); cbound (K): the bound each instance used
int omega_lds_build_table(double tol, int degrees, double* table_h, int max_entries, double* lnq_out);
// SGL form (sgl != null, L == null): the workgroup goes on with the instance's Theta-step, dual update and stopping-test sums
// (k_theta_sgl's arithmetic) -- Theta / X are the stacks it READS for W and WRITES afterwards (sg.Theta, sg.X: the same stacks),
// OmegaPrev the previous iterate, l1K (K) lambda1 / rho or mask (+ mask_stride: 0 shared, p*p per instance) with invrhoK (K),
// pk (K, or null) the instances' own dimensions; norms (K,5) and fail (K ints, zeroed by the caller) in pinned memory; seq /
// seq_val / arrive as launch_reduce_partials.
struct LdsSgl {
    double* Theta = nullptr; double* X = nullptr; const double* OmegaPrev = nullptr;
    const double* l1K = nullptr; const double* mask = nullptr; size_t mask_stride = 0; const double* invrhoK = nullptr;
    const int* pk = nullptr; double* norms = nullptr; int* fail = nullptr;
    unsigned long long* seq = nullptr; unsigned long long seq_val = 0; unsigned* arrive = nullptr;
};
bool launch_omega_lds(hipStream_t st, const double* Theta, const double* L, const double* X, const double* S, const double* betaK,
                      double* Omega, const double* table, int ntab, double lnq, int K, int p, int* flag,
                      int* flag_host, int flag_slot, unsigned long long* units, double* cbound, long long* dbg = nullptr,
                      int waves = 0, const LdsSgl* sgl = nullptr);
// gemm_i8.hip: error-free split products on the INT8 matrix cores
void launch_slice_i8(hipStream_t st, const double* A, const double* scaleK, int8_t* out, int K, int p, int S, int* flag,
                     size_t sstride = 0);
void symm_i8_set_stages(int n);
bool launch_symm_i8(hipStream_t st, const int8_t* As, const int8_t* Bs, const double* par, double* C, int K, int p, int S, int dmax);
// one product launch of k_symm_i8 (gemm_i8.hip):  acc = scale_A scale_B sum_{t+u<=dmax} 2^-(12+7(t+u)) D^A_t (D^B_u)^T,
// out1 = cI I + cAcc acc + cE1 E1 + cE2 E2, out2 = dI I + dAcc acc + dE1 E1 + dE2 E2, each as fp64 (C) and / or int8 slices (S)
// par[k][12] = { cI, cAcc, cE1, cE2, dI, dAcc, dE1, dE2, scale_A scale_B, 1 / scale_1, 1 / scale_2, - }
struct I8Op {
    const int8_t* As = nullptr;
    const int8_t* Bs = nullptr;
    size_t sstrideA = 0, sstrideB = 0;       // bytes between the slices of an operand
    const double* par = nullptr;             // [K][12]
    const double* E1 = nullptr;
    const double* E2 = nullptr;
    double* C1 = nullptr;
    double* C2 = nullptr;
    int8_t* S1 = nullptr;
    int8_t* S2 = nullptr;
    int nS1 = 0, nS2 = 0;
    size_t sstride1 = 0, sstride2 = 0;
    int* flag = nullptr;                     // raised when a digit of an output slice does not fit (a scale was too small)
    int K = 0, p = 0, P = 0;
};
bool launch_symm_i8_op(hipStream_t st, const I8Op& op, int SA, int SB, int dmax);
// the Omega-step on the int8 matrix cores (gemm_i8.hip)
static constexpr int I8_NSLICES = 59;        // slice stacks of the workspace
static constexpr int I8_MAXPROD = 8;
struct I8Omega {
    int K = 0, p = 0, P = 0;
    size_t sl = 0;                           // bytes of one slice stack (K P P)
    int8_t* slab = nullptr;
    double *par = nullptr, *par_h = nullptr; // [I8_MAXPROD][K][12] device / pinned
    double *wscale = nullptr, *wscale_h = nullptr;   // [K] scale of W
    int* flag = nullptr;
};
struct I8Cfg { int s_full = 7, s_f2 = 4, s_gf2 = 3, s_ye = 5, d_ye = 4; };
struct I8Bufs { const double* W; double *Ap, *Bp, *Y1, *F, *F2, *Om; };   // fp64 stacks (K,p,p) of the whole batch
struct I8Prog {
    struct Prod { I8Op op; int SA, SB, dmax; } prod[I8_MAXPROD];
    int nprod = 0, units = 0, k0 = 0, Kp = 0, S_W = 7;
};
int i8_omega_alloc(I8Omega* w, int K, int p);
void i8_omega_free(I8Omega* w);
int i8_omega_plan(I8Omega* w, const double* cuse_h, const double* beta_h, int k0, int Kp, double tol, int degrees,
                  const I8Cfg& cfg, const I8Bufs& bufs, I8Prog* prog);
bool i8_omega_run(hipStream_t st, I8Omega* w, const I8Prog& prog, const double* W);   // theta_pair.hip: code of the Theta kernel the last launch ran
// One workgroup per matrix, matrix resident in LDS (one-sided Jacobi on rows, wave-shuffle
// reductions).  Reads the LOWER triangle of A (numpy.linalg.eigh default).
//   D (K,p) eigenvalues (unsorted), R (K,p,p) eigenvectors in ROWS; either may be null.
//   out != null: out[k] = R^T diag(map(D, betaK[k])) R fused in the same kernel.
//   info (K ints): sweeps used, negative if not converged.
hipError_t launch_jacobi(hipStream_t st, const double* A, double* D, double* R, double* out, int map,
                         const double* betaK, int* info, int K, int p);

// ---- recon_gemm.hip ---------------------------------------------------------------------
// out[k] = R[k]^T diag(map(D[k], betaK[k])) R[k]; R holds eigenvectors in ROWS. FP64 MFMA.
// scale_work: 2*K*p doubles of scratch.
void launch_recon(hipStream_t st, double* out, const double* R, const double* D, const double* betaK,
                  int map, int K, int p, double* scale_work);

// ---- gemm_sym.hip -----------------------------------------------------------------------
// C[k] = cI*I + cAcc*(A[k]*B[k]) + cE*E[k]  and optionally  C2[k] = dI*I + dC*C[k] + dE*E[k]  for commuting
// symmetric A, B (so that A*B = A^T*B is symmetric).  coef: device [K][NS_NCOEF] = {cI,cAcc,cE,dI,dC,dE}.
static constexpr int NS_NCOEF = 6;
// variant < 0: pick by problem size.  FP64 MFMA.
int symm_variants();
bool symm_variant_built(int v);          // the shipped library holds the dispatched instances only (gemm_sym.hip)
bool symm_dl_serves(int p);               // the direct-to-LDS product kernel takes this dimension (every p >= 2 unless switched off for odd p)
void symm_set_odd_dl(bool on);            // process-wide: odd p on the direct-to-LDS kernel (default) or on the register-staged one
int symm_effective_variant(int variant, int p);   // the kernel a variant resolves to at this p (reporting)
int symm_auto_variant(int nprod, int p); // what variant < 0 resolves to for nprod products of p x p matrices in a launch
// maxdev (optional, device [K], zeroed by the caller): max |C - I| per instance.
// rowpart / fropart (optional, only where symm_bounds_tile() != 0): partial row sums of |C| per tile column,
// [K][T][p] with T = ceil(p / tile), and the tiles' shares of |C|_F^2, [K][T(T+1)/2] -- what launch_bound_rows sums up
void launch_symm(hipStream_t st, const double* A, const double* B, double* C, double* C2, const double* E,
                 const double* coef, int K, int p, int variant, double* maxdev = nullptr, double* rowpart = nullptr,
                 double* fropart = nullptr);
int symm_bounds_tile(int K, int p, int variant);
// frees the split-K scratch (k_symm_sk with a tile's k-range over several workgroups) that launches on this stream allocated
void symm_release_workspace(hipStream_t st);

// ---- the Omega-step's product chain as one persistent launch (gemm_sym.hip, k_omega_chain) ----
// one product of the chain: C = coef-affine(A B) [+ C2] for all K instances (stack base pointers; instance k at + k p^2);
// pair: a second, independent product C1 = A1 B1 with the coefficient rows K + k rides in the same step
struct SymmOp {
    const double *A, *B, *E, *A1, *B1;
    double *C, *C2, *C1;
    const double* coef;
    double *rowpart, *fropart;
    int pair, pad_;
};
static constexpr int CHAIN_MAX_OPS = 16;
struct ChainProg {
    int nops, K, p, ntiles;                 // ntiles: 64x64 tile pairs of one product
    int begin[CHAIN_MAX_OPS + 1];           // tickets [begin[s], begin[s+1]) of an instance belong to op s
    SymmOp op[CHAIN_MAX_OPS];
    long long* prof = nullptr;              // GGL_DEV builds: [grid][8] per-workgroup time accounting (100 MHz ticks)
};
// tile edge of the chain kernel for this batch, 0 = the launch-per-product path serves it
int chain_tile(int K, int p, bool force = false);      // force: without the batch-size threshold (tests, measurements)
// state: K lines of CHAIN_CNT_STRIDE 32-bit words (128 bytes per instance: completion count + one ticket word per product),
// zeroed by the caller in the stream ahead of this; flag / flag_h: raised when an instance is left incomplete.  Returns the
// grid (persistent workgroups) or < 0.
static constexpr int CHAIN_CNT_STRIDE = 32;
static constexpr int CHAIN_MAX_INST = 64;       // instances per XCD (K <= 512)
static_assert(CHAIN_MAX_OPS + 1 <= CHAIN_CNT_STRIDE, "one line holds the completion word and the ticket words");
int launch_omega_chain(hipStream_t st, const ChainProg& P, unsigned* state, int* flag, int* flag_h, int aux = 16);

#ifdef GGL_DEV
// persistent-chain probe: nprod dependent products X <- X * X in one cooperative launch with grid barriers (gemm_sym.hip)
int launch_chain_probe(hipStream_t st, double* X0, double* X1, const double* coef, int K, int p, int nprod, int variant,
                       unsigned* bar, unsigned* err, int two_level);
#endif
#ifdef GGL_DEV
// FP64 VALU || MFMA co-issue probe (probes_dev.hip): out[12] TF/s pairs, see there
void coissue_probe(hipStream_t st, double* scratch, double* out);
void mfma_lds_probe(hipStream_t st, double* scratch, double* out);      // MFMA fed from LDS at several read : MFMA ratios
#endif
// measured FP64 matrix-core ceiling (MFMA-only probe kernel; GGL_DEV builds)
double mfma_f64_peak_tflops(hipStream_t st, double* scratch, int blocks, int iters, int nacc);
double mfma_valu_mix_tflops(hipStream_t st, double* scratch, int blocks, int iters, int nv);
// C[b] = scal[b % K] * A[b] * T[b % K] (T symmetric, A general), b < nbatch; full output.  FP64 MFMA.
void launch_gemm_right(hipStream_t st, const double* A, const double* T, double* C, const double* scal,
                       int nbatch, int K, int p, int variant);


// The speculative Omega-step's bound validation (launch_bound_rows + launch_cw_final) riding in the NEXT product launch: the
// launches of a small sequence leave most CUs idle (K = 4, p = 500: 144 workgroups on 256 CUs) and the schedule does not
// depend on the validation, so the validating workgroups are appended to the grid of the first product after B' instead of
// sitting between B' and that product as two dependent launches (event timeline, round 5: 8.5 + 18.8 us of a K = 4 slab's
// 220 us, 6.5 + 12.6 of (20,200)'s 192).  Same arithmetic in the same order as the two kernels: same bound, same vector.
// Needs the Collatz-Wielandt vector of the previous iteration (dprev); the cold first step keeps the two launches.
struct CwRider {
    const double* B = nullptr;            // B' [K][p][p]
    const double* rowpart = nullptr;      // [K][T][p] row-sum partials of the B' launch
    const double* fropart = nullptr;      // [K][ntile]
    const double* dprev = nullptr;        // [K][p]
    double* dnext = nullptr;              // [K][p] (optional)
    double* d_out = nullptr;              // [K][p] row sums (optional)
    unsigned long long* cwmax = nullptr;  // [K], zero on entry, left zero
    unsigned* cnt = nullptr;              // [K], zero on entry, left zero
    double* out = nullptr;                // [K] bounds (pinned host)
    const double* cuse = nullptr;         // [K] the bound the running schedule assumes
    int* flag = nullptr;
    int* flag_host = nullptr;
    int flag_slot = 0;
    int T = 0, ntile = 0, p = 0, K = 0;   // K = 0: no rider
    int nbx = 0;                          // workgroups per instance, 16 rows each
};
// The same for the small host-to-device transfers in front of a chain (launch_copy_small): one extra workgroup per segment of
// the NEXT direct-to-LDS product launch, for tables that launch itself does not read.
void symm_set_copy_rider(const CopySegs& sg);
// ... and for the single-row norm reduction behind a Theta-step (launch_reduce_partials with a sequence word) when the NEXT
// chain's A' launch follows it in the stream anyway: out[v] = sum_b partials[b][v] in k_reduce_partials' order, then the
// sequence number, by ONE extra workgroup (the first of the riders) of that launch.
struct RedRider {
    const double* partials = nullptr;
    double* out = nullptr;                 // pinned host
    unsigned long long* seq = nullptr;     // pinned host
    unsigned long long seq_val = 0;
    int nblk = 0, nv = 0;                  // nblk = 0: no rider
};
void symm_set_reduce_rider(const RedRider& r);
void symm_set_rider(const CwRider& r);    // taken by the next launch_symm of this host thread that runs the direct-to-LDS kernel
void symm_flush_rider(hipStream_t st);    // riders nobody took get launches of their own
// event timeline (ggl_trace_*): fn(stream, kind 0 single / 1 pair, arg) after every product launch; null switches it off
void symm_set_launch_hook(void (*fn)(hipStream_t, int, void*), void* arg);
// two independent symmetric products in one launch; coef2K: [2K][NS_NCOEF] (second half for the second product)
void launch_symm_pair(hipStream_t st, const double* A, const double* B, double* C, const double* A1, const double* B1,
                      double* C1, const double* coef2K, int K, int p, int variant);

// ---- newton_schulz.hip ------------------------------------------------------------------
static constexpr double NS_TOL_EXACT = 4e-16;  // schedules end with the spectrum inside [1 - tol, 1]; this is "all fp64 has"
// what a ctx iterates the Omega-step to unless told otherwise (GGL_OPT_NS_TOL): |Omega - phiplus(W)|_2 <= 2e-12 |sqrt(W^2 +
// 4 beta I)|_2 / 2.  Measured at GGL (32,500), solved to 1e-10 on both sides: |Theta - reference|_F = 1.3e-10 on a stack of
// norm 128 (max entry 2.4e-12) against 2.8e-13 in exact mode -- 75x inside the 1e-8 the comparison asks for -- for 7 instead
// of 8 products per iteration (+10 % iterations/s)
static constexpr double NS_TOL_DEFAULT = 2e-12;
static constexpr int NS_MAX_STEPS = 24;        // square-root schedule (condition number up to ~1e18)
static constexpr int NS_RANK_MAX_STEPS = 40;   // sign schedule (resolution down to ~1e-13)
static constexpr int NS_MAX_LAUNCHES = 2 * NS_RANK_MAX_STEPS + 1 + 3;   // + start table + two pre-bound slots
// doubles per launch slot of the coefficient table (a pair launch carries 2K rows of 5)
static inline size_t NS_SLOT(int K) { return (size_t)K * 2 * NS_NCOEF; }
// all-symmetric products are only accurate while the condition number of W^2 + 4 beta I is small
static constexpr double NS_SYM_KAPPA_MAX = 300.0;
// above this condition number of W^2 + 4 beta I the Omega-step uses the eigendecomposition instead
static constexpr double NS_KAPPA_LIMIT = 1e12;
// steps: polynomial steps; products: kernel launches of products (incl. A', B'); units: symmetric products
// of the whole stack (K p^3 flop each); deg[it]: degree (3, 5 or 9) of step it in x = sqrt(eig(Z Y))
struct NsPlan {
    int steps = 0; int products = 0; bool stable = false; double kappa = 0.0; int units = 0;
    double check = 5e-5;      // L-step: largest max|T_last - I| that still means "every eigenvalue resolved"
    unsigned char deg[NS_RANK_MAX_STEPS] = {};
};
// W = ((Theta - L) - X) - beta_k S from the lower triangle, mirrored (exactly symmetric)
int form_W_tiles(int p);
void launch_form_W_sym(hipStream_t st, double* W, const double* Theta, const double* L, const double* X,
                       const double* S, const double* betaK, int K, int p);
// Omega-step in two phases around the one host sync that fetches the spectral bound:
//   ns_prepare: A' = W^2 + 4 beta I and B' = A'^2 into AB = [A' | B'] (pre_d: 2 coefficient slots)
//   -- host: c_k = sqrt(min(|B'_k|_inf, |B'_k|_F)) >= lambda_max(A'_k); ns_plan --
//   ns_run: start kernel (Y1, Z1 as polynomials of A', B') + the remaining products
// ns_plan returns 0, -1 (non-finite input) or -2 (condition number above NS_KAPPA_LIMIT: use the
// eigendecomposition); fills coef_h[launch slots] and start_h[K][5].
// degrees: highest step degree of the fast schedule: 3 = cubic only, 5 = cubic/quintic mix, 9 = + degree nine.
int ns_plan(const double* cbound_h, const double* beta_h, int K, double* coef_h, double* start_h, NsPlan* plan,
            int force_mode, int degrees = 9, double tol = NS_TOL_EXACT);
// the schedule alone (host): returns steps, fills deg[max_steps], coef[max_steps*6] = {t0..t4,l_after}
int ns_schedule_query(double l, int degrees, int max_steps, int* deg, double* coef, int* units, double tol = NS_TOL_EXACT);
// products of the all-symmetric schedule for [l, 1] (A', B' included), -1 where it does not apply; contiguous groups of a
// batch whose instances need different product counts (newton_schulz.hip)
int ns_units_query(double l, int degrees, double tol);
int ns_group_partition(const int* units, int K, int p, int max_groups, int* len_out);
void ns_prepare(hipStream_t st, const double* pre0_d, const double* pre1_d, const double* W, double* Ap, double* Bp,
                int K, int p, int variant, double* start2 = nullptr, double* rowpart = nullptr, double* fropart = nullptr);
void ns_run(hipStream_t st, const NsPlan& plan, const double* coef_d, const double* start_d, const double* W,
            double* AB, double* YP, double* Tb, double* out, int K, int p, int variant, size_t pstride = 0,
            bool fused_start = false, hipEvent_t bprime_free = nullptr);
// the same launches as a product list for k_omega_chain; NX: an extra [Y|Z] pair (A', B' survive); 0 = not a pure chain
int ns_chain_ops(const NsPlan& plan, const double* pre0_d, const double* pre1_d, const double* coef_d, const double* W,
                 double* AB, double* YP, double* NX, double* Tb, double* out, int K, int p, size_t pstride, double* start2,
                 double* rowpart, double* fropart, SymmOp* ops, int max_ops);
// speculative step (bound known before B' exists): target buffer and {dI, dC, dE} of the start the B' launch can emit
// as its second output; null if the schedule's first step has none.  start_hk: row k of ns_plan's start table.
double* ns_fused_start(const NsPlan& plan, const double* start_hk, double* YP, double* Tb, size_t n1, double out3[3]);

// L-step (C - mu I)_+ by a sign-function Newton-Schulz iteration (newton_schulz.hip)
int norm_bounds_blocks(int p);
// part[k][blk] = {max row abs-sum, sum of squares}; rowsum (optional, K*p): every row's abs-sum
void launch_norm_bounds(hipStream_t st, const double* W, int K, int p, double* part, double* rowsum = nullptr);
// Collatz-Wielandt refinement of the row-sum bound: part[k][blk] = max_i (|W| d)_i / d_i, d = rowsum
void launch_cw_bounds(hipStream_t st, const double* W, const double* rowsum, int K, int p, double* part);
// per-instance reduction of the block partials on the device; out (K doubles) may be pinned host memory.
// mode 0: sqrt(min(|.|_inf, cw, |.|_F)) (cwpart may be null); mode 1: min(|.|_inf, |.|_F)
// cuse/flag/flag_host (optional, mode 0): speculative validation -- *flag (device) and *flag_host (pinned) are set
// to 1 if some out[k] exceeds cuse[k] or is not finite
void launch_bound_final(hipStream_t st, const double* part2, const double* cwpart, int nbb, int K, double* out, int mode,
                        const double* cuse = nullptr, int* flag = nullptr, int* flag_host = nullptr);
// the same bound without a pass over B' for the norms: from the partials of the product launch (launch_symm rowpart /
// fropart with tile edge symm_bounds_tile()): d = row sums, infpart[K][bound_rows_blocks(p)] = block maxima; then the
// Collatz-Wielandt pass with the final reduction done by the last workgroup of every instance
int bound_rows_blocks(int p);
void launch_bound_rows(hipStream_t st, const double* rowpart, int T, int K, int p, double* d, double* infpart);
// launch_bound_rows + launch_cw_final as one launch (the speculative Omega-step's bound validation)
void launch_bound_cw(hipStream_t st, const double* B, const double* rowpart, int T, int K, int p, double* d_out,
                     const double* fropart, int ntile, unsigned long long* cwmax, unsigned* cnt, double* out,
                     const double* cuse, int* flag, int* flag_host, int flag_slot, const double* dprev, double* dnext);
void launch_cw_final(hipStream_t st, const double* B, const double* d, int K, int p, const double* infpart,
                     const double* fropart, int ntile, unsigned long long* cwmax, unsigned* cnt, double* out,
                     const double* cuse, int* flag, int* flag_host, int flag_slot, const double* dprev = nullptr,
                     double* dnext = nullptr);
int rank_ns_plan(const double* cnorm_h, const double* mu_h, int K, double l0, double* coef_h, NsPlan* plan,
                 int degrees = 9);
// two-tier L-step: where rank_ns_plan(l0)'s iteration leaves an eigenvalue that started at x; the continuation plan for m
// instances whose iterate has its eigenvalues at least lp away from 0 (coefficient slots `slot` doubles apart); the steps
// and closing product of a plan from a given iterate (the tail of rank_ns_run / a continuation's whole run)
double rank_ns_image(double l0, int degrees, double x);
double rank_trace_tolerance(double l0, int p);      // accepted distance of trace(sign iterate) from an integer, plan resolution l0
int rank_ns_plan_continue(const double* mu_h, int m, double lp, double* coef_h, NsPlan* plan, int degrees, size_t slot);
void rank_ns_steps(hipStream_t st, const NsPlan& plan, const double* coef_d, const double* C, double* X, double* Xn, double* Tb,
                   double* P2, double* out, double* maxdev, int K, int p, int variant, size_t cs);
// t0_ready: Tb already holds the first step's T0 (launch_rank_t0 from P = C C; the bound then came from P as well)
void rank_ns_run(hipStream_t st, const NsPlan& plan, const double* coef_d, const double* C, double* Xa, double* Xb,
                 double* Tb, double* P2, double* out, double* maxdev, int K, int p, int variant, size_t cslot = 0,
                 bool t0_ready = false);
// b_k = sqrt(min(|P_k|_inf, |P_k|_F)) >= rho(C_k) for P = C C, from the product launch's partials (launch_bound_rows first)
void launch_bound_sqrt_inf_fro(hipStream_t st, const double* infpart, int ninf, const double* fropart, int ntile, int K,
                               double* out);
void launch_rank_t0(hipStream_t st, double* P, const double* C, const double* coef_d, int K, int p);

}  // namespace ggl
