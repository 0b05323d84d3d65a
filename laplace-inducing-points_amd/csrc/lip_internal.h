// Internal (non-ABI) declarations shared by the HIP translation units of liblip_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lip.h"

namespace lip {

// ---- division by a run-time constant without the ~40-instruction software divide --------------
// q = (umulhi(n, m) + n) >> s  (round-up magic number; exact for 0 <= n < 2^31, d >= 1)
struct FastDiv {
  unsigned m, s, d;
  FastDiv() : m(1), s(0), d(1) {}
  explicit FastDiv(unsigned dd) : d(dd) {
    s = 0;
    while ((1ull << s) < dd) ++s;
    m = (unsigned)((((1ull << s) - dd) << 32) / dd + 1);
  }
  __host__ __device__ __forceinline__ int div(int n) const {
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)((__umulhi((unsigned)n, m) + (unsigned)n) >> s);
#else
    return (int)((unsigned)n / d);
#endif
  }
};

// ---- resolved (device-pointer) parameter blocks handed to the kernels -------------------
struct SegP {
  const float* a; long long a_ps;
  const float* b; long long b_ps;
  int IH, IW, C, KH, KW, stride, pad_h, pad_w, mode, Ktot;
  int b_trans;                          // LIP_SEG_B_TRANS (generic kernel only)
  FastDiv dC, dKW;
  // branch-free gather coordinate (fast kernel; stride in {1, 2}):
  //   t0 = o * mul + sgn * k + off ;  valid iff (t0 & mask) == 0 && 0 <= (t0 >> sh) < lim
  int mul, sgn, off_h, off_w, mask, sh;
};

struct IgemmP {
  int nseg;
  SegP seg[3];
  int R, OHW, OW, N;
  FastDiv dOHW, dOW;
  float* out; long long out_ps;
  const float* scale;
  const float* e0; long long e0_ps;
  const float* e1; long long e1_ps;
  const float* xhat;
  const float* res; long long res_ps;
  const float* dphi;
  float* red0; long long red0_ps;
  float* red1; long long red1_ps;
  const float* xhat2;
  const float* zeros;                   // >= 16 floats of device zeros (masked gather rows)
  unsigned long long* dbg;              // diagnostic s_memtime stamps (null in every real run)
  // split-K launches (few probes, under-filled grids): block z of gridDim.z accumulates its share of the K-tiles and
  // stores the raw sums to partial + z * partial_zs + p * R * N; igemm_finish_kernel adds the shares and runs the epilogue
  float* partial; long long partial_zs;
  int no_ksplit;                        // outputs of the primal tape: ReLU gates / pooling arg-maxima are taken from these sums,
                                        // so their summation order stays the one-block order whatever the launch geometry
  // parity-class row order of a stride-2 data gradient (fast kernel, OH and OW even): the rows of one block all
  // share (oh & 1, ow & 1), so the taps whose parity cannot match are skipped instead of gathered as zeros
  int Rc, OHW2, OW2;                    // rows per class n*(OH/2)*(OW/2), (OH/2)*(OW/2), OW/2
  FastDiv dOHW2, dOW2;
};

struct WgradP {
  const float* a;                       // primal activations [n][IH][IW][C]
  int IH, IW, C, KH, KW, stride, pad_h, pad_w;
  const float* g; long long g_ps;       // cotangent [P][R][N]
  int R, OHW, OW, N, M;                 // M = KH*KW*C
  FastDiv dOHW, dOW;
  float* y; long long y_ps;             // Y + param offset, element [m*N + co]
  const float* scale;                   // per-channel [N] or null
  int ksplit;
  const float* zeros;
  int P;                                // probe-batched variant: probes in this launch (columns = P*N)
  // per-example rows (lip_vjp_rows): grid.z = example, the row reduction of block z covers only that example's
  // seg_rows = OH*OW rows and lands in Y row (p, z): y + p*y_ps + z*seg_ys
  int seg_rows; long long seg_ys;
  // fused output of a weight gradient whose launch reduces all R rows in one block (no row split):
  //   overwrite == 1:  y = s * acc + alpha * v     (v = the probe's own slice of V at the same offset, or null)
  // instead of y += s * acc on a block the caller initialised with alpha * V — the initialisation pass over these
  // parameters is then skipped (lip_ggn_vp).  Set by the engine only where wgrad_will_overwrite() says so.
  int overwrite; const float* v; long long v_ps; float alpha;
  int sk_mg0, sk_tiles, sk_tn0;         // wgrad_skinny_kernel: first m-group of the launch, column tiles per probe it walks, first of them
};

struct ReduceP {
  const float* g; long long g_ps; int R, N;
  const float* xhat;
  float* red0; long long red0_ps;
  float* red1; long long red1_ps;
  // per-example rows: nseg > 0 -> grid.z = example, R rows per example, outputs at + z*red_seg
  int nseg; long long red_seg;
  int rpb;                              // rows per block: set by launch_reduce
};

struct PoolP {
  const float* in; long long in_ps;     // fwd: [P][n][HW][C]   bwd: [P][n][C]
  float* out; long long out_ps;         // fwd: [P][n][C]       bwd: [P][n][HW][C]
  int n, HW, C; float inv;
  const float* dphi;                    // bwd: [n][HW][C] or null
  const float* xhat;                    // bwd: for red1
  float* red0; long long red0_ps;
  float* red1; long long red1_ps;
};

struct MaxPoolP {
  const float* in; long long in_ps;     // [P][n][IH][IW][C]   (bwd: cotangent of the pooled tensor [P][n][OH][OW][C])
  float* out; long long out_ps;         // [P][n][OH][OW][C]   (bwd: [P][n][IH][IW][C])
  float* amax_w; const float* amax;     // [n][OH][OW][C] argmax as linear pixel index ih*IW+iw (float)
  int n, IH, IW, OH, OW, C, KH, KW, stride, pad_h, pad_w;
  const float* dphi; const float* xhat;
  float* red0; long long red0_ps;
  float* red1; long long red1_ps;
};

struct PrimalPostP {
  const float* z; float* a; float* dphi; float* xhat;   // all [R][N]
  const float* bias;                    // [N] or null          (THETA)
  const float* gamma; const float* beta;                 // BN   (THETA) or null
  const float* mean; const float* rstd;                  // BN   (CONST)
  const float* res;                     // [R][N] or null
  long long count; int N; int act;
};

struct HeadP {
  const float* in; long long in_ps;     // [P][n][K]
  float* out; long long out_ps;         // [P][n][K]
  const float* p; const float* s;       // softmax probs / sqrt [n][K]
  int n, K, mode, classifier; float c;
};

// ---- launchers (return hipError_t of the launch) ------------------------------------------
hipError_t launch_igemm(const IgemmP& p, int P, hipStream_t st);
hipError_t launch_wgrad(const WgradP& p, int P, hipStream_t st);
hipError_t launch_reduce(const ReduceP& p, int P, hipStream_t st);
hipError_t launch_pool_fwd(const PoolP& p, int P, hipStream_t st);
hipError_t launch_pool_bwd(const PoolP& p, int P, hipStream_t st);
hipError_t launch_primal_post(const PrimalPostP& p, hipStream_t st);
hipError_t launch_maxpool_primal(const MaxPoolP& p, hipStream_t st);
hipError_t launch_maxpool_fwd(const MaxPoolP& p, int P, hipStream_t st);
hipError_t launch_maxpool_bwd(const MaxPoolP& p, int P, hipStream_t st);
hipError_t launch_softmax(const float* logits, float* prob, float* sqrtp, int n, int K, hipStream_t st);
hipError_t launch_head(const HeadP& p, int P, hipStream_t st);
hipError_t launch_gemm_nt(const float* A, long long lda, int m, const float* B, long long ldb, int n, long long K, float* C,
                          hipStream_t st);
hipError_t launch_gemm_nn_axpy(const float* T, long long ldt, int m, int k, const float* B, long long ldb, long long N,
                               const float* V, long long ldv, float beta, float* Out, long long ldo, hipStream_t st);
hipError_t launch_scale_copy(float* y, const float* x, float a, long long count, hipStream_t st);
// y[p][off + i] = a * x[p][off + i] (x null: 0) for i < len, p < P, row stride ld: the parameters a fused weight gradient
// does NOT write (biases, BN parameters, layers on the accumulate path)
hipError_t launch_scale_copy_range(float* y, const float* x, float a, long long off, long long len, int P, long long ld, hipStream_t st);
// true when launch_wgrad(p, P) will run the one-block-per-tile kernel that honours WgradP::overwrite
bool wgrad_will_overwrite(const WgradP& p, int P);

void set_error(const char* fmt, ...);
int precision_mode();
void set_precision_mode(int m);
void set_split_k_mode(int on);
void set_wino_mode(int m);
int wino_mode();

// ---- wave / block reductions (wave = 64 lanes on gfx950) --------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

}  // namespace lip
