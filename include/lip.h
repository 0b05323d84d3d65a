/*
 * lip.h — C ABI of the MI355X-native linearised-Laplace engine (liblip_hip.so).
 *
 * The reference (nrholm1/Laplace-Inducing-Points) has no FFI: its hot path is a set of
 * Python closures over jax.jvp / jax.vjp (src/ggn.py:9-146).  This header is the build-defined
 * drop-in boundary underneath the same Python call surface (last bullet):
 * plain pointers and sizes, integer status codes, a hipStream_t passed as void*, no torch
 * types, no exceptions across the boundary.  All pointers are DEVICE pointers borrowed from
 * the caller (who owns them) unless stated otherwise.  Thread-compatible, not thread-safe.
 *
 * Each entry point cites the reference code it replaces.
 */
#ifndef LIP_H
#define LIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LIP_OK 0
#define LIP_ERR_ARG 1       /* bad argument / shape the kernels do not support */
#define LIP_ERR_HIP 2       /* a HIP runtime call failed; see lip_last_error() */
#define LIP_ERR_STATE 3     /* engine not bound / tape missing */

/* ---- address spaces an op operand can live in -------------------------------------- */
enum {
  LIP_SP_NONE = -1,
  LIP_SP_THETA = 0,  /* flat MAP parameters theta (D floats)        src/utils.py:12-17     */
  LIP_SP_CONST = 1,  /* per-handle constants (BN factors, W^T)                              */
  LIP_SP_PRIM = 2,   /* cached primal tensors, one copy per example (never per probe)       */
  LIP_SP_WORK = 3,   /* tangent / cotangent workspace, [probe][example][H][W][C]            */
  LIP_SP_VIN = 4,    /* caller's input block V  (P, D) row-major                            */
  LIP_SP_YOUT = 5,   /* caller's output block Y (P, D) row-major                            */
  LIP_SP_HEAD = 6    /* caller's output-space block (P, n, K)                               */
};

/* operand = base[space] + off + probe * pstride   (all in floats) */
typedef struct {
  int32_t space;
  int32_t reserved;
  int64_t off;
  int64_t pstride;
} lip_ref_t;

/* one K-segment of an implicit GEMM:  acc[r][co] += sum_{kh,kw,c} A[img][ih][iw][c] * B[(kh*KW+kw)*C + c][co]
 * mode 0 (conv):            ih = oh*stride + kh - pad_h
 * mode 1 (transposed conv): t = oh + pad_h - kh, valid iff t >= 0 && t % stride == 0, ih = t/stride   */
typedef struct {
  lip_ref_t a;
  lip_ref_t b;
  int32_t IH, IW, C, KH, KW, stride, pad_h, pad_w, mode;
  int32_t flags;       /* LIP_SEG_B_TRANS: B is read as  B[(tap*C + c)][n] = b[(tap*N + n)*C + c]  — the HWIO kernel of the
                          convolution being transposed, used in place (per-probe weight tangents have no transposed copy) */
} lip_seg_t;
#define LIP_SEG_B_TRANS 1

enum {
  LIP_OP_IGEMM = 1,       /* implicit-GEMM conv / dense, fused epilogue (tangent fwd, data-grad, primal) */
  LIP_OP_WGRAD = 2,       /* weight-gradient GEMM reduced over examples and pixels, per probe            */
  LIP_OP_REDUCE = 3,      /* per-channel column sums (bias / BN-parameter cotangents)                    */
  LIP_OP_POOL_FWD = 4,    /* mean over pixels                                                            */
  LIP_OP_POOL_BWD = 5,    /* broadcast/HW, times dphi, plus parameter reductions                         */
  LIP_OP_PRIMAL_POST = 6, /* z -> xhat, y, a = act(y), dphi = act'(y)   (primal pass, once per binding)  */
  LIP_OP_SOFTMAX = 7,     /* logits -> p, sqrt(p)                        (primal pass)                   */
  LIP_OP_HEAD = 8,        /* output-space Hessian / square-root factor action  src/ggn.py:16-39,125-131  */
  /* window pools: with a NONE aux0 ref the three ops below compute the window AVERAGE (sum / (KH*KW), padding
     counted -- flax.linen.avg_pool, used by the reference's LeNet5 src/scalemodels.py:28,34) and its transpose */
  LIP_OP_MAXPOOL_PRIMAL = 9,  /* max pool of the primal activations + cached argmax (aux0)             */
  LIP_OP_MAXPOOL_FWD = 10,    /* tangent of max pool: gather at the cached argmax                      */
  LIP_OP_MAXPOOL_BWD = 11     /* cotangent of max pool (gather over the covering windows), times dphi,
                                 plus parameter reductions — same epilogue fields as LIP_OP_POOL_BWD   */
};

/* head modes */
enum {
  LIP_HEAD_GGN = 0,  /* g = c * (diag p - p p^T) Jv         src/ggn.py:125-131 (classifier), :112-113 (regressor: c*Jv) */
  LIP_HEAD_LT = 1,   /* U = c * L^T Jv  -> HEAD space       src/ggn.py:29-39  (W^T)   */
  LIP_HEAD_L = 2,    /* g = c * L U     <- HEAD space       src/ggn.py:16-27  (W)     */
  LIP_HEAD_OUT = 3,  /* U = Jv          -> HEAD space       raw JVP  (src/lla.py:153) */
  LIP_HEAD_IN = 4    /* g = U           <- HEAD space       raw VJP  (src/lla.py:66)  */
};

typedef struct {
  int32_t kind;
  int32_t nseg;
  lip_seg_t seg[3];
  int32_t n_img, OH, OW, N;      /* rows R = n_img*OH*OW, N output channels                       */
  int32_t act;                   /* PRIMAL_POST: activation id (0 none,1 relu,2 tanh,3 gelu-tanh) */
  int32_t ksplit;                /* WGRAD: split of the row reduction (atomics when > 1)          */
  int32_t M;                     /* WGRAD: KH*KW*Cin ; REDUCE/POOL: unused                        */
  int32_t classifier;            /* HEAD: 1 softmax-CE, 0 Gaussian regression                     */
  float   fscale;                /* POOL: 1/HW ; others unused                                    */
  float   bn_eps;
  lip_ref_t out;                 /* [probe][R][N]                                                 */
  lip_ref_t out2;                /* PRIMAL_POST: dphi ; SOFTMAX: sqrt(p)                          */
  lip_ref_t out3;                /* PRIMAL_POST: xhat                                             */
  lip_ref_t scale;               /* per-channel multiplier of the accumulator                     */
  lip_ref_t e0;                  /* per-channel addend (per probe if pstride != 0)                */
  lip_ref_t e1;                  /* per-channel multiplier of xhat, added                         */
  lip_ref_t xhat;                /* [R][N] primal                                                 */
  lip_ref_t res;                 /* residual addend [probe][R][N]                                 */
  lip_ref_t dphi;                /* [R][N] primal, multiplies the result                          */
  lip_ref_t red0;                /* += column sums of the result            (YOUT)                */
  lip_ref_t red1;                /* += column sums of result * xhat2        (YOUT)                */
  lip_ref_t xhat2;               /* [R][N] primal used by red1                                    */
  lip_ref_t aux0, aux1;          /* PRIMAL_POST: BN mean / rsqrt(var+eps) ; HEAD: p / sqrt(p)     */
} lip_op_t;

typedef struct lip_engine lip_engine_t;

/* ---- library ----------------------------------------------------------------------- */
int lip_abi_version(void);                 /* bumps when this header changes                     */
const char* lip_last_error(void);          /* text of the last failure on this thread            */
int lip_sizeof_op(void);                   /* sizeof(lip_op_t): lets the ctypes mirror self-check */
/* arithmetic of the MFMA kernels (process-wide): 0 = exact f32 MFMA (default, dtype "f32");
 * 1 = split-precision operands x = hi + lo in bf16, three bf16 MFMAs per product with f32 accumulation
 *     ("bf16x3": ~1e-5 relative error per product, ~5x fewer matrix-pipe cycles).                     */
int lip_set_precision(int32_t mode);
int lip_get_precision(void);
/* split-K implicit GEMM of under-filled launches (few probes): 1 = on (default), 0 = off.  The
 * results differ only in the summation order of the K axis; the switch exists so that a test can compare the two
 * orders in one process (the environment variable LIP_NOKSPLIT is read once).                          */
int lip_set_split_k(int32_t on);
/* Winograd F(2x2, 3x3) route of the 3x3 / stride-1 / pad-1 layers of the tangent and backward tapes (2.25x fewer
 * matrix-pipe multiplications, f32 in / f32 accumulate; the transforms move a layer's result by ~2e-7 relative;
 * DESIGN.md section 4): 0 = off (the direct implicit GEMMs everywhere), 1 = on (default: every eligible launch),
 * 2 = on (the value the tests set; the same launches as 1).  Never applied to the primal tape.  The environment
 * variable LIP_NOWINO is read once; this call overrides it.                                             */
int lip_set_winograd(int32_t mode);
int lip_get_winograd(void);

/* ---- engine: one per (network, theta_MAP, data slice Z) binding on one device --------
 * Replaces the closure factories compute_ggn_vp / compute_W_vps (src/ggn.py:97,9): they
 * snapshot flat_params and Z at factory time; so does the engine (theta/consts/prim are
 * fixed until the next lip_engine_bind).                                                */
enum { LIP_TAPE_PRIMAL = 0, LIP_TAPE_TANGENT = 1, LIP_TAPE_BACKWARD = 2 };

int lip_engine_create(lip_engine_t** out, int64_t D, int32_t n_img, int32_t K);
int lip_engine_destroy(lip_engine_t* e);
int lip_engine_set_tape(lip_engine_t* e, int32_t which, const lip_op_t* ops /*host*/, int32_t nops);
int lip_engine_bind(lip_engine_t* e, const float* theta, const float* consts, float* prim,
                    float* work, int64_t work_floats_per_probe, int32_t max_probes_per_chunk);
/* run the primal tape once (caches activations, BN-normalised values, act', softmax).   */
int lip_engine_primal(lip_engine_t* e, void* stream);

/* measurement hook: when enabled every op launch is bracketed by HIP events recorded on the launch
 * stream; lip_engine_profile_read sums elapsed ms and launch counts per op kind (index = LIP_OP_*)
 * and clears the log.  Used by bench.py for the live per-kernel roofline figure.                 */
int lip_engine_profile(lip_engine_t* e, int32_t enable);
int lip_engine_profile_read(lip_engine_t* e, double* ms_by_kind, int64_t* launches_by_kind, int32_t nkinds);

/* run ONE host-supplied op on a single probe chunk (P <= chunk size) against the engine's bound buffers and the
 * caller's V / Y / H blocks.  The second-order pass of the inducing-point gradient (reverse over the tangent tape,
 * src/train_inducing.py:195-232) is driven from the host op by op through this entry point.                    */
int lip_engine_run_op(lip_engine_t* e, const lip_op_t* op /*host*/, const float* V, float* Y, float* H, int32_t P,
                      int32_t head_mode, float head_c, void* stream);
/* test hook: run ops [first, first+count) of one tape on a single probe chunk (P <= chunk size) */
int lip_debug_run_ops(lip_engine_t* e, int32_t which, int32_t first, int32_t count, const float* V, float* Y,
                      float* H, int32_t P, int32_t head_mode, float head_c, void* stream);

/* Y[p] = scale * sum_i J_i^T H_i J_i V[p] + alpha * V[p]      src/ggn.py:133-144, src/lla.py:21-22
 * (scale carries N/M and, for the regressor, exp(-logvar): src/ggn.py:111-113)           */
int lip_ggn_vp(lip_engine_t* e, const float* V, float* Y, int32_t P, float scale, float alpha, void* stream);
/* U[p,i,:] = c * L_i^T J_i V[p]   (mode LIP_HEAD_LT, src/ggn.py:55-62,84-85) or J_i V[p] (LIP_HEAD_OUT) */
int lip_jvp(lip_engine_t* e, const float* V, float* U, int32_t P, int32_t head_mode, float c, void* stream);
/* Y[p] = sum_i J_i^T (c * L_i U[p,i,:])  (LIP_HEAD_L, src/ggn.py:64-76,87-91) or raw (LIP_HEAD_IN) */
int lip_vjp(lip_engine_t* e, const float* U, float* Y, int32_t P, int32_t head_mode, float c, void* stream);
/* per-example rows of the same product: Y[(p*n + i)] = J_i^T (c * L_i U[p,i,:]), Y is (P*n, D) — no reduction
 * crosses examples, so a probe that holds e_k on every example yields the n rows J_i^T L_i e_k of the GGN's
 * square-root factor in ONE sweep (the reference builds them column by column: src/ggn.py:64-93, 207-219) */
int lip_vjp_rows(lip_engine_t* e, const float* U, float* Y, int32_t P, int32_t head_mode, float c, void* stream);

/* ---- Krylov / trace primitives on blocks of vectors: X is (P, N) row-major -----------
 * They replace what XLA emits for matfree's tridiag_sym (called at src/sample.py:114-126),
 * jax.scipy.sparse.linalg.cg (src/stochtrace.py:146,192; src/sample.py:71) and the
 * Hutchinson quadratic forms (src/stochtrace.py:30-34).                                 */
int lip_bdot(const float* X, const float* Y, float* out /*[P], overwritten*/, int32_t P, int64_t N, void* stream);
int lip_axpby(float* Y, const float* X, const float* a /*[P] or NULL*/, float a_s, const float* b /*[P] or NULL*/,
              float b_s, int32_t P, int64_t N, void* stream);      /* Y[p] = (a_s*a[p]) X[p] + (b_s*b[p]) Y[p] */
/* Lanczos basis Q is (P, kmax, ldq): row stride ldq >= N with ldq % 4 == 0 and a 16-byte aligned base, so the
 * basis — (j+1) x the traffic of w — streams with aligned 16-byte loads; w stays (P, N).
 * c[p][j] = <Q[p][j], w[p]>, j < k.                                                       */
int lip_multi_dot(const float* Q, const float* w, float* c /*[P][kmax], first k overwritten*/, int32_t P,
                  int32_t k, int32_t kmax, int64_t N, int64_t ldq, void* stream);
/* w[p] -= sum_j c[p][j] Q[p][j];  nrm2[p] = ||w[p]||^2 after the update.                 */
int lip_multi_axpy_norm(const float* Q, const float* c, float* w, float* nrm2 /*[P], overwritten*/, int32_t P,
                        int32_t k, int32_t kmax, int64_t N, int64_t ldq, void* stream);
/* Q[p][j] = w[p] * rsqrt(nrm2[p])   (next Lanczos vector; the row padding is zeroed)     */
int lip_scale_store(const float* w, const float* nrm2, float* Q, int32_t j, int32_t P, int32_t kmax, int64_t N,
                    int64_t ldq, void* stream);
/* fused CG update (one pass): x += a p; r -= a Ap; rr[p] = <r,r>, a[p] = rr_old[p]/pAp[p]; inactive probes
 * (active[p]==0) are left untouched.                                                     */
int lip_cg_update(float* x, float* r, const float* p, const float* Ap, const float* rr_old, const float* pAp,
                  const int32_t* active, float* rr_new /*[P], overwritten*/, int32_t P, int64_t N, void* stream);
/* p = r + (rr_new/rr_old) p                                                              */
int lip_cg_direction(float* p, const float* r, const float* rr_new, const float* rr_old, const int32_t* active,
                     int32_t P, int64_t N, void* stream);
/* C (m, n) float64, overwritten = A B^T with A (m, K), B (n, K) float32 rows (row strides lda / ldb >= K, 4-byte
 * alignment suffices) accumulated in float64: the tall-skinny inner products whose float32 accumulation is too coarse —
 * the sampler's stiff-direction coefficients (src/sample.py:130-139 at cond 3e9), the Gram of Hutch++'s tall-skinny
 * orthonormalisation and its projections G Q^T (src/stochtrace.py:124-131).  Uses one partial-sum buffer per DEVICE:
 * calls on different streams of one device must not overlap (the package issues them on one stream).          */
int lip_dot_nt_f64(const float* A, int64_t lda, int32_t m, const float* B, int64_t ldb, int32_t n, int64_t K, double* C,
                   void* stream);
/* C (m, n) float32, overwritten = A B^T, same operand layout as lip_dot_nt_f64, float32 MFMA with the long reduction
 * axis split over the grid (partial tiles added by float atomics: the summation order, hence the last bits, vary from
 * run to run).  W^T applied to a block of draws on a materialised factor (src/sample.py:130-139).             */
int lip_gemm_nt(const float* A, int64_t lda, int32_t m, const float* B, int64_t ldb, int32_t n, int64_t K, float* C,
                void* stream);
/* Out (m, N) = T (m, k) B (k, N) + beta V (m, N): float32 MFMA, short reduction k (rows of a materialised factor), very
 * long N (= D); row strides ldt >= k, ldb / ldv / ldo >= N.  V may be NULL (no addend) or Out itself (in place); Out must
 * not alias T or B.  The second pass of a block of posterior draws, x = alpha^(-1/2) eps + (coefficients) Qm
 * (src/sample.py:139-143 in the Gram's eigenbasis), and the second product of the factor-mode GGN-vp.        */
int lip_gemm_nn_axpy(const float* T, int64_t ldt, int32_t m, int32_t k, const float* B, int64_t ldb, int64_t N, const float* V,
                     int64_t ldv, float beta, float* Out, int64_t ldo, void* stream);
/* Out[i] = zscale * Z[i] + sum_j Cm[i][j] Y[j],  i < r: r combinations of the s rows of Y (s, N) in one streaming pass
 * per 12 output rows; Cm (r, s) float64 row-major on the device, Z (r, N) optional (NULL: no addend).  Replaces
 * jnp.linalg.qr's Q of src/stochtrace.py:128 (as L^-1 Y after a Gram factorisation) and the deflation
 * G - (G Q) Q^T of :131.  Out must not alias Y or Z.                                                      */
int lip_rows_combine(const double* Cm, const float* Y, int64_t ldy, int32_t s, const float* Z, int64_t ldz, float zscale,
                     float* Out, int64_t ldo, int32_t r, int64_t N, void* stream);
/* counter-based Rademacher (+-1) / standard-normal fill of a (P, N) block               */
int lip_fill_rademacher(float* X, int32_t P, int64_t N, uint64_t seed, void* stream);
int lip_fill_normal(float* X, int32_t P, int64_t N, uint64_t seed, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LIP_H */
