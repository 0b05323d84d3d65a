// Krylov / trace primitives on blocks of vectors X (P, N) row-major — HBM-bound kernels:
// 16-byte loads where the row alignment allows (rows of an odd-length D-vector block are not
// 16-byte aligned, so every row is split into scalar head | float4 body | scalar tail),
// per-wave shuffle reductions, one float atomic per wave/block.
//
// Replaces what XLA emits for matfree's Lanczos (reference src/sample.py:114-126), JAX's CG
// (src/stochtrace.py:146,192; src/sample.py:71) and the Hutchinson quadratic forms
// (src/stochtrace.py:30-34).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <mutex>
#include "lip_internal.h"

namespace lip {

constexpr int KT = 256;        // threads per block
constexpr int CHUNK = KT * 8;  // elements of one row handled by one block

struct RowSplit { long long h, G; };   // head length, number of aligned quads

__device__ __forceinline__ RowSplit split_row(const float* x, long long N) {
  RowSplit s;
  const long long mis = ((unsigned long long)x >> 2) & 3;
  s.h = (4 - mis) & 3;
  if (s.h > N) s.h = N;
  s.G = (N - s.h) >> 2;
  return s;
}

__device__ __forceinline__ float block_sum(float v, float* sm /*[KT/64]*/) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x < KT / 64) t = sm[threadIdx.x];
  if (threadIdx.x < 64) t = wave_sum(t);
  __syncthreads();
  return t;   // valid in thread 0 (all of wave 0)
}

// Apply f to every element chunk of row [0, N): f4(offset) for aligned quads, f1(offset) for scalars.
// Work is distributed grid-stride over blockIdx.x; block 0 also takes the head / tail scalars.
template <typename F4, typename F1>
__device__ __forceinline__ void row_apply(const RowSplit s, long long N, F4 f4, F1 f1) {
  for (long long g = (long long)blockIdx.x * KT + threadIdx.x; g < s.G; g += (long long)gridDim.x * KT)
    f4(s.h + 4 * g);
  if (blockIdx.x == 0) {
    for (long long i = threadIdx.x; i < s.h; i += KT) f1(i);
    for (long long i = s.h + 4 * s.G + threadIdx.x; i < N; i += KT) f1(i);
  }
}

#define LD4(ptr, off) (*reinterpret_cast<const float4*>((ptr) + (off)))
#define ST4(ptr, off, v) (*reinterpret_cast<float4*>((ptr) + (off)) = (v))

// ---- out[p] = <X[p], Y[p]> --------------------------------------------------------------------------
__global__ __launch_bounds__(KT) void bdot_kernel(const float* X, const float* Y, float* out, long long N) {
  __shared__ float sm[KT / 64];
  const int p = blockIdx.y;
  const float* x = X + (long long)p * N;
  const float* y = Y + (long long)p * N;
  const RowSplit s = split_row(x, N);
  float acc = 0.f;
  row_apply(s, N,
            [&](long long o) { const float4 a = LD4(x, o), b = LD4(y, o); acc += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; },
            [&](long long o) { acc += x[o] * y[o]; });
  const float t = block_sum(acc, sm);
  if (threadIdx.x == 0) atomicAdd(out + p, t);
}

// ---- Y[p] = ca[p] X[p] + cb[p] Y[p] --------------------------------------------------------------------
__global__ __launch_bounds__(KT) void axpby_kernel(float* Y, const float* X, const float* a, float a_s,
                                                   const float* b, float b_s, long long N) {
  const int p = blockIdx.y;
  const float ca = a_s * (a ? a[p] : 1.f), cb = b_s * (b ? b[p] : 1.f);
  const float* x = X + (long long)p * N;
  float* y = Y + (long long)p * N;
  const RowSplit s = split_row(x, N);
  const bool useb = cb != 0.f;   // b == 0: pure scaled copy, Y may hold garbage (NaN-safe)
  row_apply(s, N,
            [&](long long o) {
              const float4 xv = LD4(x, o);
              float4 yv = make_float4(0.f, 0.f, 0.f, 0.f);
              if (useb) yv = LD4(y, o);
              ST4(y, o, make_float4(ca * xv.x + cb * yv.x, ca * xv.y + cb * yv.y, ca * xv.z + cb * yv.z, ca * xv.w + cb * yv.w));
            },
            [&](long long o) { y[o] = ca * x[o] + (useb ? cb * y[o] : 0.f); });
}

// ---- Lanczos basis kernels.  Q is (P, kmax, ldq) with ldq % 4 == 0 and a 16-byte aligned base: the rows of
// the basis — (j+1) x the traffic of w — are read with aligned float4 loads; w (P, N), whose rows inherit the
// caller's odd alignment, is read / written with coalesced dwords.  Each block keeps its 2048-element chunk of
// w in registers and streams the k rows of Q past it.
__device__ __forceinline__ float4 ld_q4(const float* q, long long o, long long N) {
  float4 v = LD4(q, o);
  if (o + 3 >= N) {                      // last, partial quad of the row: mask the padding
    if (o + 1 >= N) v.y = 0.f;
    if (o + 2 >= N) v.z = 0.f;
    v.w = 0.f;
  }
  return v;
}

// c[p][j] = <Q[p][j], w[p]>, j < k
__global__ __launch_bounds__(KT) void multi_dot_kernel(const float* Q, const float* W, float* c, int k, int kmax,
                                                       long long N, long long ldq) {
  extern __shared__ float cs[];   // [k]
  const int p = blockIdx.y;
  const float* w = W + (long long)p * N;
  const float* q0 = Q + (long long)p * kmax * ldq;
  for (int j = threadIdx.x; j < k; j += KT) cs[j] = 0.f;
  __syncthreads();
  const long long beg = (long long)blockIdx.x * CHUNK;
  float wr[8];
  long long o4[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    o4[i] = beg + 4ll * (threadIdx.x + i * KT);
#pragma unroll
    for (int e = 0; e < 4; ++e) wr[4 * i + e] = (o4[i] + e < N) ? w[o4[i] + e] : 0.f;
  }
  for (int j = 0; j < k; ++j) {
    const float* q = q0 + (long long)j * ldq;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (o4[i] < N) {
        const float4 v = ld_q4(q, o4[i], N);
        acc += v.x * wr[4 * i] + v.y * wr[4 * i + 1] + v.z * wr[4 * i + 2] + v.w * wr[4 * i + 3];
      }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) atomicAdd(&cs[j], acc);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < k; j += KT) atomicAdd(c + (long long)p * kmax + j, cs[j]);
}

// w[p] -= sum_j c[p][j] Q[p][j] ; nrm2[p] = ||w[p]||^2
__global__ __launch_bounds__(KT) void multi_axpy_norm_kernel(const float* Q, const float* c, float* W, float* nrm2,
                                                             int k, int kmax, long long N, long long ldq) {
  extern __shared__ float cs[];   // [k]
  __shared__ float sm[KT / 64];
  const int p = blockIdx.y;
  float* w = W + (long long)p * N;
  const float* q0 = Q + (long long)p * kmax * ldq;
  for (int j = threadIdx.x; j < k; j += KT) cs[j] = c[(long long)p * kmax + j];
  __syncthreads();
  const long long beg = (long long)blockIdx.x * CHUNK;
  float wr[8];
  long long o4[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    o4[i] = beg + 4ll * (threadIdx.x + i * KT);
#pragma unroll
    for (int e = 0; e < 4; ++e) wr[4 * i + e] = (o4[i] + e < N) ? w[o4[i] + e] : 0.f;
  }
  for (int j = 0; j < k; ++j) {
    const float* q = q0 + (long long)j * ldq;
    const float cj = cs[j];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (o4[i] < N) {
        const float4 v = ld_q4(q, o4[i], N);
        wr[4 * i] -= cj * v.x; wr[4 * i + 1] -= cj * v.y; wr[4 * i + 2] -= cj * v.z; wr[4 * i + 3] -= cj * v.w;
      }
    }
  }
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (o4[i] + e < N) { w[o4[i] + e] = wr[4 * i + e]; acc += wr[4 * i + e] * wr[4 * i + e]; }
  const float t = block_sum(acc, sm);
  if (threadIdx.x == 0) atomicAdd(nrm2 + p, t);
}

// Q[p][j] = w[p] / sqrt(nrm2[p])   (padding of the row is zeroed)
__global__ __launch_bounds__(KT) void scale_store_kernel(const float* W, const float* nrm2, float* Q, int j, int kmax,
                                                         long long N, long long ldq) {
  const int p = blockIdx.y;
  const float inv = rsqrtf(nrm2[p]);
  const float* w = W + (long long)p * N;
  float* q = Q + ((long long)p * kmax + j) * ldq;
  for (long long o = 4ll * ((long long)blockIdx.x * KT + threadIdx.x); o < ldq; o += 4ll * (long long)gridDim.x * KT) {
    float4 v;
    v.x = (o < N) ? w[o] * inv : 0.f;
    v.y = (o + 1 < N) ? w[o + 1] * inv : 0.f;
    v.z = (o + 2 < N) ? w[o + 2] * inv : 0.f;
    v.w = (o + 3 < N) ? w[o + 3] * inv : 0.f;
    ST4(q, o, v);
  }
}

// ---- fused CG update: a = rr_old/pAp ; x += a p ; r -= a Ap ; rr_new = <r, r> ------------------------------
__global__ __launch_bounds__(KT) void cg_update_kernel(float* X, float* R, const float* Pd, const float* AP,
                                                       const float* rr_old, const float* pAp, const int* active,
                                                       float* rr_new, long long N) {
  __shared__ float sm[KT / 64];
  const int p = blockIdx.y;
  if (active && !active[p]) return;
  const float a = rr_old[p] / pAp[p];
  float* x = X + (long long)p * N;
  float* r = R + (long long)p * N;
  const float* pd = Pd + (long long)p * N;
  const float* ap = AP + (long long)p * N;
  const RowSplit s = split_row(x, N);
  float acc = 0.f;
  row_apply(s, N,
            [&](long long o) {
              const float4 pv = LD4(pd, o), av = LD4(ap, o);
              float4 xv = LD4(x, o), rv = LD4(r, o);
              xv.x += a * pv.x; xv.y += a * pv.y; xv.z += a * pv.z; xv.w += a * pv.w;
              rv.x -= a * av.x; rv.y -= a * av.y; rv.z -= a * av.z; rv.w -= a * av.w;
              ST4(x, o, xv); ST4(r, o, rv);
              acc += rv.x * rv.x + rv.y * rv.y + rv.z * rv.z + rv.w * rv.w;
            },
            [&](long long o) {
              x[o] += a * pd[o];
              const float rv = r[o] - a * ap[o];
              r[o] = rv;
              acc += rv * rv;
            });
  const float t = block_sum(acc, sm);
  if (threadIdx.x == 0) atomicAdd(rr_new + p, t);
}

// ---- p = r + (rr_new/rr_old) p ------------------------------------------------------------------------------
__global__ __launch_bounds__(KT) void cg_direction_kernel(float* Pd, const float* R, const float* rr_new,
                                                          const float* rr_old, const int* active, long long N) {
  const int p = blockIdx.y;
  if (active && !active[p]) return;
  const float b = rr_new[p] / rr_old[p];
  float* pd = Pd + (long long)p * N;
  const float* r = R + (long long)p * N;
  const RowSplit s = split_row(pd, N);
  row_apply(s, N,
            [&](long long o) {
              const float4 rv = LD4(r, o); float4 pv = LD4(pd, o);
              pv.x = rv.x + b * pv.x; pv.y = rv.y + b * pv.y; pv.z = rv.z + b * pv.z; pv.w = rv.w + b * pv.w;
              ST4(pd, o, pv);
            },
            [&](long long o) { pd[o] = r[o] + b * pd[o]; });
}


// ---- C = A B^T with float64 accumulation: A (m, K), B (n, K) float32, rows K-contiguous (row strides lda / ldb, any
// 4-byte alignment), C (m, n) float64.  The tall-skinny inner products of the posterior engine: the coefficients
// <q_k, v> of the sampler's stiff directions (cond(A) ~ 3e9: a float32-accumulated dot of length 1e6 is two decades too
// coarse there), the Gram Y Y^T of CholeskyQR2 / Hutch++ (src/stochtrace.py:124-133), the projections G Q^T.
// Every f32 x f32 product is exact in float64, so the result carries one rounding per addition at 1e-16.
// gfx950 has no faster float64 matrix path than v_fma_f64 (78.6 TF either way), so this is a VALU kernel:
// block = 32 x 32 output tile x one K-range; K-chunks of 32 are converted to float64 once, on their way into k-major
// LDS tiles (the next chunk's global loads are in flight during the sweep); each of the four waves takes 8 of the 32
// k's with a 4 x 4 micro-tile per lane (four ds_read_b128 per 16 v_fma_f64); the waves' partial tiles meet in LDS.
// Few output tiles mean hundreds of K-ranges per tile: their partial tiles go to a scratch buffer and a second, tiny
// kernel adds them (float64 atomics on one 32 x 32 tile serialise: 1059 adds per address cost more than the sweep).
constexpr int DT_B = 32, DT_KC = 32, DT_LD = DT_B + 2;
constexpr long long DT_SCRATCH_TILES = 4096;          // partial tiles the scratch buffer holds (32 MiB of float64)
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

__global__ __launch_bounds__(256) void dot_nt_f64_kernel(const float* __restrict__ A, long long lda, int m,
                                                         const float* __restrict__ B, long long ldb, int n, long long K,
                                                         long long kper, double* __restrict__ C, double* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) double As[DT_KC * DT_LD];
  __shared__ __attribute__((aligned(16))) double Bs[DT_KC * DT_LD];
  __shared__ double red[3 * DT_B * DT_B];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane >> 3, lj = lane & 7;
  const int tiles_n = (n + DT_B - 1) / DT_B;
  const int m0 = (blockIdx.x / tiles_n) * DT_B, n0 = (blockIdx.x % tiles_n) * DT_B;
  const long long kb = (long long)blockIdx.y * kper, ke = (kb + kper < K) ? kb + kper : K;
  double acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
  // loader: quad tid -> (row = tid >> 3, k = 4 * (tid & 7)): one dword-aligned 16-byte load per operand and chunk
  const int lrow = tid >> 3, lkq = 4 * (tid & 7);
  const bool arow = m0 + lrow < m, brow = n0 + lrow < n;
  const float* ap = A + (long long)(arow ? m0 + lrow : 0) * lda + lkq;
  const float* bp = B + (long long)(brow ? n0 + lrow : 0) * ldb + lkq;
  auto fetch = [&](const float* __restrict__ src, bool rowok, long long k0, float (&v)[4]) {
    const long long k = k0 + lkq;
    v[0] = v[1] = v[2] = v[3] = 0.f;
    if (rowok && k < ke) {
      if (k + 3 < ke) { const f4u t = *reinterpret_cast<const f4u*>(src + k0); v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
      else { v[0] = src[k0]; if (k + 1 < ke) v[1] = src[k0 + 1]; if (k + 2 < ke) v[2] = src[k0 + 2]; }
    }
  };
  float ra[4], rb[4];
  fetch(ap, arow, kb, ra);
  fetch(bp, brow, kb, rb);
  for (long long k0 = kb; k0 < ke; k0 += DT_KC) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      As[(lkq + t) * DT_LD + lrow] = (double)ra[t];
      Bs[(lkq + t) * DT_LD + lrow] = (double)rb[t];
    }
    __syncthreads();
    if (k0 + DT_KC < ke) { fetch(ap, arow, k0 + DT_KC, ra); fetch(bp, brow, k0 + DT_KC, rb); }
#pragma unroll
    for (int kk = 0; kk < DT_KC / 4; ++kk) {
      const int k = wave * (DT_KC / 4) + kk;
      const double2 a0 = *reinterpret_cast<const double2*>(&As[k * DT_LD + 4 * li]);
      const double2 a1 = *reinterpret_cast<const double2*>(&As[k * DT_LD + 4 * li + 2]);
      const double2 b0 = *reinterpret_cast<const double2*>(&Bs[k * DT_LD + 4 * lj]);
      const double2 b1 = *reinterpret_cast<const double2*>(&Bs[k * DT_LD + 4 * lj + 2]);
      const double ad[4] = {a0.x, a0.y, a1.x, a1.y};
      const double bd[4] = {b0.x, b0.y, b1.x, b1.y};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fma(ad[i], bd[j], acc[i][j]);
    }
    __syncthreads();
  }
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[(wave - 1) * DT_B * DT_B + (4 * li + i) * DT_B + 4 * lj + j] = acc[i][j];
  }
  __syncthreads();
  if (wave == 0) {
    double* mine = part ? part + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * (DT_B * DT_B) : nullptr;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = (4 * li + i) * DT_B + 4 * lj + j;
        const double v = acc[i][j] + red[e] + red[DT_B * DT_B + e] + red[2 * DT_B * DT_B + e];
        const int r = m0 + 4 * li + i, c = n0 + 4 * lj + j;
        if (mine) mine[e] = v;
        else if (r < m && c < n) unsafeAtomicAdd(C + (long long)r * n + c, v);
      }
  }
}

// The same 32 x 32 x K-range block on the float64 matrix pipe (round 3): v_mfma_f64_16x16x4_f64, operands straight from
// global memory — lane (i = lane & 15, q = lane >> 4) loads the four consecutive k's 4q .. 4q+3 of its row as ONE 16-byte
// load per operand tile and 16-deep chunk (16 rows x 64 contiguous bytes per instruction) and k-step s of the chunk
// multiplies element s of every lane's quad (A and B agree on that choice, so the sum over the chunk is unchanged);
// converted to float64 in registers.  No LDS in the loop, no barrier: the four waves of a block take every fourth
// chunk of the block's K-range and meet in LDS at the end, exactly where the VALU kernel's waves do.  The VALU kernel
// reads 2 B of LDS per FLOP (a 4 x 4 micro-tile per lane) — half the LDS bandwidth at the float64 peak.
typedef double f64x4 __attribute__((ext_vector_type(4)));

// QUAD: a block covers 2 x 2 tiles of 32 x 32, ONE PER WAVE, each wave over the block's whole K-range (no reduction in
// LDS; the pairs of waves share operand rows through the L1) — taken when the output has at least 2 x 2 tiles.
template <bool QUAD>
__global__ __launch_bounds__(256) void dot_nt_f64_mfma_kernel(const float* __restrict__ A, long long lda, int m,
                                                              const float* __restrict__ B, long long ldb, int n, long long K,
                                                              long long kper, double* __restrict__ C, double* __restrict__ part) {
  __shared__ double red[QUAD ? 1 : 3 * DT_B * DT_B];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, q = lane >> 4;
  const int tiles_n = (n + DT_B - 1) / DT_B;
  int m0, n0, tile_id;
  if (QUAD) {
    const int bn = (tiles_n + 1) / 2;
    const int tm_ = 2 * ((int)blockIdx.x / bn) + (wave >> 1), tn_ = 2 * ((int)blockIdx.x % bn) + (wave & 1);
    m0 = tm_ * DT_B; n0 = tn_ * DT_B; tile_id = tm_ * tiles_n + tn_;
    if (m0 >= m || n0 >= n) return;                            // no tile behind this wave (odd tile counts)
  } else {
    m0 = ((int)blockIdx.x / tiles_n) * DT_B; n0 = ((int)blockIdx.x % tiles_n) * DT_B; tile_id = (int)blockIdx.x;
  }
  const long long kb = (long long)blockIdx.y * kper, ke = (kb + kper < K) ? kb + kper : K;
  f64x4 acc[2][2];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = f64x4{0.0, 0.0, 0.0, 0.0};
  // rows clamped (outputs past m / n are never written), this lane's quad at k = chunk + 4 q
  const float* ap[2];
  const float* bp[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    ap[t] = A + (long long)min(m0 + 16 * t + i16, m - 1) * lda + 4 * q;
    bp[t] = B + (long long)min(n0 + 16 * t + i16, n - 1) * ldb + 4 * q;
  }
  auto fetch = [&](long long k0, float (&ra)[2][4], float (&rb)[2][4]) __attribute__((always_inline)) {
    if (k0 + 16 <= ke) {                                      // whole chunk (uniform): unconditional 16-byte loads
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f4u va = *reinterpret_cast<const f4u*>(ap[t] + k0);
        const f4u vb = *reinterpret_cast<const f4u*>(bp[t] + k0);
#pragma unroll
        for (int e = 0; e < 4; ++e) { ra[t][e] = va[e]; rb[t][e] = vb[e]; }
      }
    } else {                                                  // the last, partial chunk of the K-range: element loads
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = k0 + 4 * q + e < ke;
          ra[t][e] = ok ? ap[t][k0 + e] : 0.f;
          rb[t][e] = ok ? bp[t][k0 + e] : 0.f;
        }
    }
  };
  auto sweep = [&](const float (&ra)[2][4], const float (&rb)[2][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const double a0 = (double)ra[0][e], a1 = (double)ra[1][e], b0 = (double)rb[0][e], b1 = (double)rb[1][e];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
  };
  float ra0[2][4], rb0[2][4], ra1[2][4], rb1[2][4];
  constexpr long long STEP = QUAD ? 16 : 64;
  long long k0 = QUAD ? kb : kb + 16ll * wave;                // !QUAD: this wave's chunks are every fourth one of the K-range
  if (k0 < ke) fetch(k0, ra0, rb0);
  while (k0 < ke) {
    const long long k1 = k0 + STEP;
    if (k1 < ke) fetch(k1, ra1, rb1);
    sweep(ra0, rb0);
    if (k1 >= ke) break;
    const long long k2 = k1 + STEP;
    if (k2 < ke) fetch(k2, ra0, rb0);
    sweep(ra1, rb1);
    k0 = k2;
  }
  // C/D map of v_mfma_f64_16x16x4_f64: register r of lane l holds (row = (l >> 4) + 4 r, col = l & 15)
  if (QUAD) {
    const int tiles = ((m + DT_B - 1) / DT_B) * tiles_n;
    double* mine = part ? part + ((long long)blockIdx.y * tiles + tile_id) * (DT_B * DT_B) : nullptr;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int e = (16 * tm + q + 4 * r) * DT_B + 16 * tn + i16;
          const int row = m0 + 16 * tm + q + 4 * r, col = n0 + 16 * tn + i16;
          if (mine) mine[e] = acc[tm][tn][r];
          else if (row < m && col < n) unsafeAtomicAdd(C + (long long)row * n + col, acc[tm][tn][r]);
        }
    return;
  }
  if (wave > 0) {
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave - 1) * DT_B * DT_B + (16 * tm + q + 4 * r) * DT_B + 16 * tn + i16] = acc[tm][tn][r];
  }
  __syncthreads();
  if (wave == 0) {
    double* mine = part ? part + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * (DT_B * DT_B) : nullptr;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int e = (16 * tm + q + 4 * r) * DT_B + 16 * tn + i16;
          const double v = acc[tm][tn][r] + red[e] + red[DT_B * DT_B + e] + red[2 * DT_B * DT_B + e];
          const int row = m0 + 16 * tm + q + 4 * r, col = n0 + 16 * tn + i16;
          if (mine) mine[e] = v;
          else if (row < m && col < n) unsafeAtomicAdd(C + (long long)row * n + col, v);
        }
  }
}

// second stage: C[r][c] = sum over the K-ranges of the partial tiles.  One block per 64 elements of a tile, sixteen
// waves each summing every sixteenth partial (coalesced 512-byte reads), LDS for the last step — a serial loop over
// hundreds of partials per thread would be latency-bound (measured: 190 us for 768 partials of one tile).
__global__ __launch_bounds__(1024) void dot_nt_reduce_kernel(const double* __restrict__ part, int tiles, int ks, int m, int n,
                                                             double* __restrict__ C) {
  __shared__ double sm[16][64];
  const int tile = blockIdx.x >> 4, e = ((blockIdx.x & 15) << 6) + (threadIdx.x & 63), ys = threadIdx.x >> 6;
  double s = 0.0;
  for (int y = ys; y < ks; y += 16) s += part[((long long)y * tiles + tile) * (DT_B * DT_B) + e];
  sm[ys][threadIdx.x & 63] = s;
  __syncthreads();
  if (ys == 0) {
#pragma unroll
    for (int y = 1; y < 16; ++y) s += sm[y][threadIdx.x];
    const int tiles_n = (n + DT_B - 1) / DT_B;
    const int r = (tile / tiles_n) * DT_B + e / DT_B, c = (tile % tiles_n) * DT_B + e % DT_B;
    if (r < m && c < n) C[(long long)r * n + c] = s;
  }
}

// Partial-tile scratch of lip_dot_nt_f64, one buffer per (device, stream): kernels of one stream are ordered, so launch
// i's partial tiles are reduced before launch i + 1 overwrites them; a second stream or thread gets its own buffer.
// Null when the table is full or the allocation fails — the caller then takes the float64-atomics path.
static double* dot_nt_scratch(hipStream_t st) {
  struct Entry { int dev; hipStream_t st; double* buf; };
  static Entry table[16];
  static int used = 0;
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  for (int i = 0; i < used; ++i)
    if (table[i].dev == dev && table[i].st == st) return table[i].buf;
  if (used == 16) return nullptr;
  double* buf = nullptr;
  if (hipMalloc((void**)&buf, sizeof(double) * DT_SCRATCH_TILES * DT_B * DT_B) != hipSuccess) return nullptr;
  table[used++] = Entry{dev, st, buf};
  return buf;
}

// ---- Out[i] = zscale * Z[i] + sum_j Cm[i][j] Y[j]: r combinations of the s rows of Y (s, N), streamed once per tile of
// RT output rows — the "triangular solve" of CholeskyQR2 (Q = L^-1 Y), the Hutch++ deflation G - (G Q^T) Q
// (src/stochtrace.py:131) and the change of basis of a factor.  Coefficients arrive in float64 and are rounded once.
template <int RT>
__global__ __launch_bounds__(256) void rows_combine_kernel(const double* __restrict__ Cm, const float* __restrict__ Y,
                                                           long long ldy, int s, const float* __restrict__ Z, long long ldz,
                                                           float zscale, float* __restrict__ Out, long long ldo, int r,
                                                           long long N) {
  extern __shared__ float cs[];   // [RT][s]
  const int r0 = blockIdx.x * RT;
  for (int e = threadIdx.x; e < RT * s; e += 256) {
    const int t = e / s, j = e - t * s;
    cs[e] = (r0 + t < r) ? (float)Cm[(long long)(r0 + t) * s + j] : 0.f;
  }
  __syncthreads();
  const long long col = 4ll * ((long long)blockIdx.y * 256 + threadIdx.x);
  if (col >= N) return;
  const bool full = col + 3 < N;
  float acc[RT][4];
#pragma unroll
  for (int t = 0; t < RT; ++t) { acc[t][0] = acc[t][1] = acc[t][2] = acc[t][3] = 0.f; }
  for (int j = 0; j < s; ++j) {
    const float* src = Y + (long long)j * ldy + col;
    float y0, y1 = 0.f, y2 = 0.f, y3 = 0.f;
    if (full) { const f4u v = *reinterpret_cast<const f4u*>(src); y0 = v[0]; y1 = v[1]; y2 = v[2]; y3 = v[3]; }
    else { y0 = src[0]; if (col + 1 < N) y1 = src[1]; if (col + 2 < N) y2 = src[2]; }
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const float c = cs[t * s + j];
      acc[t][0] += c * y0; acc[t][1] += c * y1; acc[t][2] += c * y2; acc[t][3] += c * y3;
    }
  }
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    if (r0 + t >= r) break;
    float* dst = Out + (long long)(r0 + t) * ldo + col;
    if (Z) {
      const float* zs = Z + (long long)(r0 + t) * ldz + col;
      acc[t][0] += zscale * zs[0];
      if (col + 1 < N) acc[t][1] += zscale * zs[1];
      if (col + 2 < N) acc[t][2] += zscale * zs[2];
      if (col + 3 < N) acc[t][3] += zscale * zs[3];
    }
    if (full) { f4u v; v[0] = acc[t][0]; v[1] = acc[t][1]; v[2] = acc[t][2]; v[3] = acc[t][3]; *reinterpret_cast<f4u*>(dst) = v; }
    else { dst[0] = acc[t][0]; if (col + 1 < N) dst[1] = acc[t][1]; if (col + 2 < N) dst[2] = acc[t][2]; }
  }
}

// ---- counter-based RNG: Philox4x32-10, counter = flat element index / 4, key = seed ---------------------------
__device__ __forceinline__ void philox4x32(unsigned long long ctr, unsigned long long key, unsigned int (&out)[4]) {
  unsigned int c0 = (unsigned int)ctr, c1 = (unsigned int)(ctr >> 32), c2 = 0x9E3779B9u, c3 = 0xBB67AE85u;
  unsigned int k0 = (unsigned int)key, k1 = (unsigned int)(key >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long m0 = (unsigned long long)0xD2511F53u * c0;
    const unsigned long long m1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned int n0 = (unsigned int)(m1 >> 32) ^ c1 ^ k0;
    const unsigned int n1 = (unsigned int)m1;
    const unsigned int n2 = (unsigned int)(m0 >> 32) ^ c3 ^ k1;
    const unsigned int n3 = (unsigned int)m0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

template <bool NORMAL>
__global__ __launch_bounds__(KT) void fill_kernel(float* X, long long total, unsigned long long seed) {
  const long long nq = (total + 3) >> 2;
  const bool aligned = (((unsigned long long)X) & 15ull) == 0;     // a slice of a block may start anywhere
  for (long long q = (long long)blockIdx.x * KT + threadIdx.x; q < nq; q += (long long)gridDim.x * KT) {
    unsigned int u[4];
    philox4x32((unsigned long long)q, seed, u);
    float v[4];
    if (NORMAL) {
      // Box-Muller on two uniform pairs
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float u1 = ((float)(u[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float u2 = ((float)(u[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        // hardware transcendentals (v_log_f32 / v_sin_f32 / v_cos_f32, ~1e-6 relative): the library versions made
        // this fill ALU-bound at 3.8 TB/s; u1 >= 2^-25, so the logarithm stays far from its denormal range
        const float rad = __fsqrt_rn(-2.0f * __logf(u1));
        float sn, cs;
        __sincosf(6.283185307179586f * u2, &sn, &cs);
        v[2 * h] = rad * cs; v[2 * h + 1] = rad * sn;
      }
    } else {
#pragma unroll
      for (int h = 0; h < 4; ++h) v[h] = (u[h] & 0x80000000u) ? 1.f : -1.f;
    }
    if (aligned && 4 * q + 3 < total) {
      ST4(X, 4 * q, make_float4(v[0], v[1], v[2], v[3]));
    } else {
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const long long o = 4 * q + h;
        if (o < total) X[o] = v[h];
      }
    }
  }
}

// Rademacher fill, one BIT per element: a Philox4x32-10 call is ~80 integer multiplies (v_mul_hi/lo at quarter rate),
// which bounds a fill that spends one call per 16 bytes at ~4 TB/s — measured 3.4.  One call yields 128 sign bits = 128
// elements = 512 bytes: thread t of a block writes the float4 at quad (chunk * 32 + i) * KT + t for i < 32, bit group i
// of its 128-bit word, so every store instruction of a wave is one contiguous kilobyte and the kernel is write-bound.
__global__ __launch_bounds__(KT) void fill_rademacher_kernel(float* X, long long total, unsigned long long seed) {
  const long long nq = (total + 3) >> 2;                               // quads
  const bool aligned = (((unsigned long long)X) & 15ull) == 0;
  const long long nchunk = (nq + 32ll * KT - 1) / (32ll * KT);
  for (long long ch = blockIdx.x; ch < nchunk; ch += gridDim.x) {
    unsigned int u[4];
    philox4x32((unsigned long long)(ch * KT + threadIdx.x), seed, u);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const long long q = (ch * 32 + i) * KT + threadIdx.x;
      if (q >= nq) break;
      const unsigned bits = u[i >> 3] >> (4 * (i & 7));
      float v[4];
#pragma unroll
      for (int h = 0; h < 4; ++h) v[h] = (bits >> h) & 1u ? 1.f : -1.f;
      if (aligned && 4 * q + 3 < total) ST4(X, 4 * q, make_float4(v[0], v[1], v[2], v[3]));
      else
        for (int h = 0; h < 4; ++h) if (4 * q + h < total) X[4 * q + h] = v[h];
    }
  }
}

static inline unsigned nblk_for(long long N, long long per_block, unsigned cap) {
  long long b = (N + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (unsigned)b;
}

static bool same_alignment(const void* a, const void* b) {
  return (((unsigned long long)a ^ (unsigned long long)b) & 15ull) == 0;
}

}  // namespace lip

using namespace lip;

#define LIP_CHECK_HIP(expr)                                                        \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess) { set_error("%s: %s", #expr, hipGetErrorString(_e)); return LIP_ERR_HIP; } \
  } while (0)

extern "C" {

int lip_bdot(const float* X, const float* Y, float* out, int32_t P, int64_t N, void* stream) {
  if (!X || !Y || !out || P <= 0 || N <= 0) { set_error("lip_bdot: bad argument"); return LIP_ERR_ARG; }
  if (!same_alignment(X, Y) ) { set_error("lip_bdot: X and Y must share 16-byte alignment"); return LIP_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  LIP_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(float) * P, st));
  hipLaunchKernelGGL(bdot_kernel, dim3(nblk_for(N, CHUNK * 4, 256), P), dim3(KT), 0, st, X, Y, out, (long long)N);
  LIP_CHECK_HIP(hipGetLastError());
  return LIP_OK;
}

int lip_axpby(float* Y, const float* X, const float* a, float a_s, const float* b, float b_s, int32_t P, int64_t N,
              void* stream) {
  if (!X || !Y || P <= 0 || N <= 0) { set_error("lip_axpby: bad argument"); return LIP_ERR_ARG; }
  if (!same_alignment(X, Y)) { set_error("lip_axpby: X and Y must share 16-byte alignment"); return LIP_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(axpby_kernel, dim3(nblk_for(N, CHUNK, 2048), P), dim3(KT), 0, st, Y, X, a, a_s, b, b_s, (long long)N);
  LIP_CHECK_HIP(hipGetLastError());
  return LIP_OK;
}

static int check_basis(const float* Q, int64_t N, int64_t ldq, const char* who) {
  if (ldq < N || (ldq & 3) || (((unsigned long long)Q) & 15ull)) {
    set_error("%s: the basis needs ldq >= N, ldq %% 4 == 0 and a 16-byte aligned base", who);
    return LIP_ERR_ARG;
  }
  return LIP_OK;
}

int lip_multi_dot(const float* Q, const float* w, float* c, int32_t P, int32_t k, int32_t kmax, int64_t N, int64_t ldq,
                  void* stream) {
  if (!Q || !w || !c || P <= 0 || k <= 0 || k > kmax || N <= 0 || k > 8192) { set_error("lip_multi_dot: bad argument"); return LIP_ERR_ARG; }
  if (check_basis(Q, N, ldq, "lip_multi_dot")) return LIP_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  LIP_CHECK_HIP(hipMemset2DAsync(c, sizeof(float) * kmax, 0, sizeof(float) * k, P, st));
  const unsigned nb = (unsigned)((N + CHUNK - 1) / CHUNK);
  hipLaunchKernelGGL(multi_dot_kernel, dim3(nb, P), dim3(KT), sizeof(float) * k, st, Q, w, c, k, kmax, (long long)N, (long long)ldq);
  LIP_CHECK_HIP(hipGetLastError());
  return LIP_OK;
}

int lip_multi_axpy_norm(const float* Q, const float* c, float* w, float* nrm2, int32_t P, int32_t k, int32_t kmax,
                        int64_t N, int64_t ldq, void* stream) {
  if (!Q || !w || !c || !nrm2 || P <= 0 || k <= 0 || k > kmax || N <= 0 || k > 8192) { set_error("lip_multi_axpy_norm: bad argument"); return LIP_ERR_ARG; }
  if (check_basis(Q, N, ldq, "lip_multi_axpy_norm")) return LIP_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  LIP_CHECK_HIP(hipMemsetAsync(nrm2, 0, sizeof(float) * P, st));
  const unsigned nb = (unsigned)((N + CHUNK - 1) / CHUNK);
  hipLaunchKernelGGL(multi_axpy_norm_kernel, dim3(nb, P), dim3(KT), sizeof(float) * k, st, Q, c, w, nrm2, k, kmax, (long long)N, (long long)ldq);
  LIP_CHECK_HIP(hipGetLastError());
  return LIP_OK;
}

int lip_scale_store(const float* w, const float* nrm2, float* Q, int32_t j, int32_t P, int32_t kmax, int64_t N,
                    int64_t ldq, void* stream) {
  if (!Q || !w || !nrm2 || P <= 0 || j < 0 || j >= kmax || N <= 0) { set_error("lip_scale_store: bad argument"); return LIP_ERR_ARG; }
  if (check_basis(Q, N, ldq, "lip_scale_store")) return LIP_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(scale_store_kernel, dim3(nblk_for(N, CHUNK * 2, 2048), P), dim3(KT), 0, st, w, nrm2, Q, j, kmax, (long long)N, (long long)ldq);
  LIP_CHECK_HIP(hipGetLastError());
  return LIP_OK;
}

int lip_cg_update(float* x, float* r, const float* p, const float* Ap, const float* rr_old, const float* pAp,
                  const int32_t* active, float* rr_new, int32_t P, int64_t N, void* stream) {
  if (!x || !r || !p || !Ap || !rr_old || !pAp || !rr_new || P <= 0 || N <= 0) { set_error("lip_cg_update: bad argument"); return LIP_ERR_ARG; }
  if (!same_alignment(x, r) || !same_alignment(x, p) || !same_alignment(x, Ap)) { set_error("lip_cg_update: blocks must share 16-byte alignment"); return LIP_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  LIP_CHECK_HIP(hipMemsetAsync(rr_new, 0, sizeof(float) * P, st));
  hipLaunchKernelGGL(cg_update_kernel, dim3(nblk_for(N, CHUNK * 2, 1024), P), dim3(KT), 0, st, x, r, p, Ap, rr_old, pAp,
                     (const int*)active, rr_new, (long long)N);
  LIP_CHECK_HIP(hipGetLastError());
  return LIP_OK;
}

int lip_cg_direction(float* p, const float* r, const float* rr_new, const float* rr_old, const int32_t* active, int32_t P,
                     int64_t N, void* stream) {
  if (!p || !r || !rr_new || !rr_old || P <= 0 || N <= 0) { set_error("lip_cg_direction: bad argument"); return LIP_ERR_ARG; }
  if (!same_alignment(p, r)) { set_error("lip_cg_direction: blocks must share 16-byte alignment"); return LIP_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(cg_direction_kernel, dim3(nblk_for(N, CHUNK, 2048), P), dim3(KT), 0, st, p, r, rr_new, rr_old,
                     (const int*)active, (long long)N);
  LIP_CHECK_HIP(hipGetLastError());
  return LIP_OK;
}


int lip_dot_nt_f64(const float* A, int64_t lda, int32_t m, const float* B, int64_t ldb, int32_t n, int64_t K, double* C,
                   void* stream) {
  if (!A || !B || !C || m <= 0 || n <= 0 || K <= 0 || lda < K || ldb < K) { set_error("lip_dot_nt_f64: bad argument"); return LIP_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  const long long tiles = (long long)((m + DT_B - 1) / DT_B) * ((n + DT_B - 1) / DT_B);
  // split K until the launch has ~3 blocks per CU (LDS allows 3), whole 32-chunks per block, at least 8 chunks each
  long long chunks = (K + DT_KC - 1) / DT_KC, ks = (768 + tiles - 1) / tiles;
  if (ks > chunks / 8) ks = chunks / 8;
  if (ks < 1) ks = 1;
  if (ks > 65535) ks = 65535;
  const long long kper = (chunks + ks - 1) / ks * DT_KC;
  ks = (K + kper - 1) / kper;
  // partial tiles through the scratch buffer of this (device, stream); float64 atomics when a launch has more partial
  // tiles than it holds or no buffer is to be had
  double* part = (ks > 1 && tiles * ks <= DT_SCRATCH_TILES) ? dot_nt_scratch(st) : nullptr;
  if (!part) LIP_CHECK_HIP(hipMemsetAsync(C, 0, sizeof(double) * (size_t)m * n, st));
  static const bool valu = getenv("LIP_DOT_NT_VALU") != nullptr;          // A/B switch: the VALU / LDS kernel
  static const bool noquad = getenv("LIP_DOT_NT_NOQUAD") != nullptr;      // A/B switch: one tile per block, waves split K
  const int tm32 = (m + DT_B - 1) / DT_B, tn32 = (n + DT_B - 1) / DT_B;
  if (valu)
    hipLaunchKernelGGL(dot_nt_f64_kernel, dim3((unsigned)tiles, (unsigned)ks), dim3(256), 0, st, A, (long long)lda, m, B,
                       (long long)ldb, n, (long long)K, kper, C, part);
  else if (!noquad && tm32 >= 2 && tn32 >= 2 && (!part || tiles * ks * 4 <= DT_SCRATCH_TILES) && chunks / (ks * 4) >= 4) {
    // a wave per tile and K-split: four times the K-splits keep the number of waves (a wave of the other form sweeps a
    // quarter of its block's K-range)
    long long ks4 = ks * 4;
    const long long kper4 = (chunks + ks4 - 1) / ks4 * DT_KC;
    ks4 = (K + kper4 - 1) / kper4;
    const unsigned blocks = (unsigned)(((tm32 + 1) / 2) * ((tn32 + 1) / 2));
    hipLaunchKernelGGL((dot_nt_f64_mfma_kernel<true>), dim3(blocks, (unsigned)ks4), dim3(256), 0, st, A, (long long)lda, m, B,
                       (long long)ldb, n, (long long)K, kper4, C, part);
    LIP_CHECK_HIP(hipGetLastError());
    if (part) {
      hipLaunchKernelGGL(dot_nt_reduce_kernel, dim3((unsigned)(tiles * 16)), dim3(1024), 0, st, part, (int)tiles, (int)ks4, m, n, C);
      LIP_CHECK_HIP(hipGetLastError());
    }
    return LIP_OK;
  } else
    hipLaunchKernelGGL((dot_nt_f64_mfma_kernel<false>), dim3((unsigned)tiles, (unsigned)ks), dim3(256), 0, st, A, (long long)lda, m, B,
                       (long long)ldb, n, (long long)K, kper, C, part);
  LIP_CHECK_HIP(hipGetLastError());
  if (part) {
    hipLaunchKernelGGL(dot_nt_reduce_kernel, dim3((unsigned)(tiles * 16)), dim3(1024), 0, st, part, (int)tiles, (int)ks, m, n, C);
    LIP_CHECK_HIP(hipGetLastError());
  }
  return LIP_OK;
}

int lip_gemm_nt(const float* A, int64_t lda, int32_t m, const float* B, int64_t ldb, int32_t n, int64_t K, float* C,
                void* stream) {
  if (!A || !B || !C || m <= 0 || n <= 0 || K <= 0 || lda < K || ldb < K) { set_error("lip_gemm_nt: bad argument"); return LIP_ERR_ARG; }
  LIP_CHECK_HIP(launch_gemm_nt(A, (long long)lda, m, B, (long long)ldb, n, (long long)K, C, (hipStream_t)stream));
  return LIP_OK;
}

int lip_gemm_nn_axpy(const float* T, int64_t ldt, int32_t m, int32_t k, const float* B, int64_t ldb, int64_t N, const float* V,
                     int64_t ldv, float beta, float* Out, int64_t ldo, void* stream) {
  if (!T || !B || !Out || m <= 0 || k < 4 || N < 4 || ldt < k || ldb < N || ldo < N || (V && ldv < N)) {
    set_error("lip_gemm_nn_axpy: bad argument (k and N must be at least 4)");
    return LIP_ERR_ARG;
  }
  if (Out == B || Out == T) { set_error("lip_gemm_nn_axpy: the output must not alias T or B (it may alias V)"); return LIP_ERR_ARG; }
  if (V && V != Out && V < Out + (size_t)(m - 1) * ldo + N && Out < V + (size_t)(m - 1) * ldv + N) {
    set_error("lip_gemm_nn_axpy: V overlaps the output without being it");
    return LIP_ERR_ARG;
  }
  LIP_CHECK_HIP(launch_gemm_nn_axpy(T, (long long)ldt, m, k, B, (long long)ldb, (long long)N, V, (long long)ldv, beta, Out,
                                    (long long)ldo, (hipStream_t)stream));
  return LIP_OK;
}

int lip_rows_combine(const double* Cm, const float* Y, int64_t ldy, int32_t s, const float* Z, int64_t ldz, float zscale,
                     float* Out, int64_t ldo, int32_t r, int64_t N, void* stream) {
  if (!Cm || !Y || !Out || s <= 0 || r <= 0 || N <= 0 || ldy < N || ldo < N || (Z && ldz < N) || s > 4096) {
    set_error("lip_rows_combine: bad argument");
    return LIP_ERR_ARG;
  }
  if (Out == Y || Out == Z) { set_error("lip_rows_combine: the output must not alias an input"); return LIP_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  const unsigned ny = (unsigned)((N + 1023) / 1024);
  if (r <= 4 || s > 1365)        // the coefficient tile lives in LDS (RT * s floats <= 64 KiB): 12 rows up to s = 1365, 4 rows up to 4096
    hipLaunchKernelGGL((rows_combine_kernel<4>), dim3((unsigned)((r + 3) / 4), ny), dim3(256), sizeof(float) * 4 * s, st, Cm, Y,
                       (long long)ldy, s, Z, (long long)ldz, zscale, Out, (long long)ldo, r, (long long)N);
  else
    hipLaunchKernelGGL((rows_combine_kernel<12>), dim3((unsigned)((r + 11) / 12), ny), dim3(256), sizeof(float) * 12 * s, st, Cm, Y,
                       (long long)ldy, s, Z, (long long)ldz, zscale, Out, (long long)ldo, r, (long long)N);
  LIP_CHECK_HIP(hipGetLastError());
  return LIP_OK;
}

int lip_fill_rademacher(float* X, int32_t P, int64_t N, uint64_t seed, void* stream) {
  if (!X || P <= 0 || N <= 0) { set_error("lip_fill_rademacher: bad argument"); return LIP_ERR_ARG; }
  const long long total = (long long)P * N;
  hipLaunchKernelGGL(fill_rademacher_kernel, dim3(nblk_for(total, KT * 128, 16384)), dim3(KT), 0, (hipStream_t)stream, X, total,
                     (unsigned long long)seed);
  LIP_CHECK_HIP(hipGetLastError());
  return LIP_OK;
}

int lip_fill_normal(float* X, int32_t P, int64_t N, uint64_t seed, void* stream) {
  if (!X || P <= 0 || N <= 0) { set_error("lip_fill_normal: bad argument"); return LIP_ERR_ARG; }
  const long long total = (long long)P * N;
  hipLaunchKernelGGL((fill_kernel<true>), dim3(nblk_for(total, KT * 16, 8192)), dim3(KT), 0, (hipStream_t)stream, X, total,
                     (unsigned long long)seed);
  LIP_CHECK_HIP(hipGetLastError());
  return LIP_OK;
}

}  // extern "C"
