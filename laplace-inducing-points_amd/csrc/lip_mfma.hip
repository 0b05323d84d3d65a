// MFMA kernels of the linearised-network engine (gfx950 / CDNA4, wave64).
//
//  igemm_kernel : implicit-GEMM convolution / dense layer with up to three K-segments and a
//                 fused epilogue.  One launch is
//                   * a tangent-forward layer   dz = conv(da, W) + conv(a, dW_p)   (K1 of SURVEY §2.2,
//                     reference src/ggn.py:59,139 jax.jvp) with the BN / bias / residual / act'
//                     tangent fused in the epilogue,
//                   * a data-gradient layer     g_in = act' * (convT(g, W^T s) [+ convT(g2, ..)] + res)
//                     (K3, reference src/ggn.py:75-76,142-143 jax.vjp) with the bias / BN parameter
//                     cotangents reduced from the tile,
//                   * or a primal layer (P = 1).
//  wgrad_kernel : per-probe weight cotangent  dW_p = s * sum_{i,pix} im2col(a_i)^T g_{i,p}, written
//                 straight into the caller's (P, D) output block — the sum over examples happens in
//                 the MFMA K dimension; no (M, D) per-example intermediate exists
//                 (the reference materialises it: src/ggn.py:89-91).
//
// Arithmetic: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bit-for-bit an fmaf chain), so the
// results are exact-f32 GEMMs; dtype "f32".  Operand maps (cdna_hip_programming.md §3):
//   A: lane l holds A[i = l&31][k = l>>5],  B: lane l holds B[k = l>>5][j = l&31],
//   C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
// LDS images are k-major (As[k][m], Bs[k][n]) so every MFMA operand read is 32 consecutive
// dwords per half-wave: conflict-free ds_read_b32.
#include <stdio.h>
#include <stdlib.h>
#include <mutex>
#include "lip_internal.h"

namespace lip {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;

template <int WM, int WN, int TM, int TN>
struct Tile {
  static constexpr int NT = WM * WN * 64;
  static constexpr int BM = WM * TM * 32;
  static constexpr int BN = WN * TN * 32;
  static constexpr int AE = BM * BK / NT;   // A floats per thread per K-tile
  static constexpr int AQ = AE / 4;         // A float4 per thread per K-tile
  static constexpr int BE = (BN * BK + NT - 1) / NT;   // B floats per thread per K-tile (last one partial when WM = 3)
  static constexpr bool BPART = (BN * BK) % NT != 0;
};

// One BK-deep MFMA sweep over the LDS tiles.
template <int WM, int WN, int TM, int TN, int LDA, int LDB>
__device__ __forceinline__ void mfma_sweep(const float* __restrict__ As, const float* __restrict__ Bs,
                                           f32x16 (&acc)[TM][TN], int wm, int wn, int lane) {
  const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int kk = 0; kk < BK / 2; ++kk) {
    const int krow = 2 * kk + lh;
    float a[TM], b[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) a[tm] = As[krow * LDA + (wm * TM + tm) * 32 + l31];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) b[tn] = Bs[krow * LDB + (wn * TN + tn) * 32 + l31];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
  }
}

// m-major A image (A/B build only, -DLIP_MMAJOR_A): rows of the 16 k-values of one m, padded to LDK = 20 floats — one
// ds_write_b128 per gathered float4 instead of a 4-dword transposing scatter, 2-way bank-conflicted operand reads.
// Measured in the same box call: 0.6 % SLOWER than the k-major image, so it is not the default.
template <int WM, int WN, int TM, int TN, int LDK, int LDB>
__device__ __forceinline__ void mfma_sweep_mmajor(const float* __restrict__ As, const float* __restrict__ Bs,
                                                  f32x16 (&acc)[TM][TN], int wm, int wn, int lane) {
  const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int kk = 0; kk < BK / 2; ++kk) {
    const int krow = 2 * kk + lh;
    float a[TM], b[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) a[tm] = As[((wm * TM + tm) * 32 + l31) * LDK + krow];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) b[tn] = Bs[krow * LDB + (wn * TN + tn) * 32 + l31];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
  }
}

// ------------------------------------------------------------------------------------------
// Split-precision ("bf16x3") operand path: every f32 operand x is stored in LDS as two bf16 planes,
// hi = bf16(x), lo = bf16(x - hi) (|x - hi - lo| <= 2^-18 |x|), and a 16-deep K-tile of a 32x32 tile is
//   acc += a_lo*b_hi + a_hi*b_lo + a_hi*b_hi        (3 x v_mfma_f32_32x32x16_bf16, f32 accumulate)
// i.e. 96 matrix-pipe cycles instead of the 512 of eight v_mfma_f32_32x32x2_f32; the dropped lo*lo term and the
// roundings bound the error of a product by ~3 * 2^-18 |a b| (1.1e-5).  LDS rows hold the 16 k-values of one
// m (or n) as 32 B of bf16 padded to 48 B: the 16-byte fragment of lane l (row l&31, k-half l>>5) is one
// conflict-free ds_read_b128 (slot = 3*row mod 16 is a bijection on every b128 lane group).
// ------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int SROW = 12;                       // floats per LDS row of the split layout (48 B)

__device__ __forceinline__ void split_store4(float* plane_hi, float* plane_lo, int row, int k4, float x0, float x1,
                                             float x2, float x3) {
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  bf16x4 hi, lo;
  hi[0] = (__bf16)x0; hi[1] = (__bf16)x1; hi[2] = (__bf16)x2; hi[3] = (__bf16)x3;
  lo[0] = (__bf16)(x0 - (float)hi[0]); lo[1] = (__bf16)(x1 - (float)hi[1]);
  lo[2] = (__bf16)(x2 - (float)hi[2]); lo[3] = (__bf16)(x3 - (float)hi[3]);
  *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(plane_hi + row * SROW) + k4) = hi;
  *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(plane_lo + row * SROW) + k4) = lo;
}

__device__ __forceinline__ void split_store1(float* plane_hi, float* plane_lo, int row, int k, float x) {
  const __bf16 hi = (__bf16)x;
  const __bf16 lo = (__bf16)(x - (float)hi);
  reinterpret_cast<__bf16*>(plane_hi + row * SROW)[k] = hi;
  reinterpret_cast<__bf16*>(plane_lo + row * SROW)[k] = lo;
}

// two k-adjacent values of one row (k even): the weight-gradient loaders hold rows (r, r+1) of the reduction axis
__device__ __forceinline__ void split_store_pair(float* plane_hi, float* plane_lo, int row, int k, float x0, float x1) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 hi, lo;
  hi[0] = (__bf16)x0; hi[1] = (__bf16)x1;
  lo[0] = (__bf16)(x0 - (float)hi[0]); lo[1] = (__bf16)(x1 - (float)hi[1]);
  *reinterpret_cast<bf16x2*>(reinterpret_cast<__bf16*>(plane_hi + row * SROW) + k) = hi;
  *reinterpret_cast<bf16x2*>(reinterpret_cast<__bf16*>(plane_lo + row * SROW) + k) = lo;
}

// One 16-deep K-tile on the split LDS images: As = [hi plane BM rows | lo plane BM rows], Bs likewise (BN rows).
template <int WM, int WN, int TM, int TN>
__device__ __forceinline__ void mfma_sweep_split(const float* __restrict__ As, const float* __restrict__ Bs,
                                                 f32x16 (&acc)[TM][TN], int wm, int wn, int lane) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const int l31 = lane & 31, lh = lane >> 5;
  bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int row = (wm * TM + tm) * 32 + l31;
    ah[tm] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(As + row * SROW) + 8 * lh);
    al[tm] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(As + (BM + row) * SROW) + 8 * lh);
  }
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int row = (wn * TN + tn) * 32 + l31;
    bh[tn] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(Bs + row * SROW) + 8 * lh);
    bl[tn] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(Bs + (BN + row) * SROW) + 8 * lh);
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
      acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
      acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
    }
}

// Software pipeline shared by the fast kernels: LDS is double buffered and two register tile sets are
// in flight, so the global loads of K-tile t+2 are issued before the MFMA sweep of tile t and only
// waited for after the sweep of tile t+1 (a counted vmcnt: the loads are branch-free — masked rows read
// a device zero page — so the steady-state body is one basic block); one barrier per K-tile.
//   load(ar, br)            : advance-independent: issue the global loads of the tile under the cursor
//   store(ar, br, As, Bs)   : write a register tile into one LDS buffer
//   advance()               : move the cursor to the next tile (only called while tiles remain)
//   sweep(As, Bs)           : MFMA sweep over one LDS buffer
template <int AE, int BE, int ASZ, int BSZ, class Load, class Store, class Advance, class Sweep>
__device__ __forceinline__ void pipelined_k_loop(int T, float* As, float* Bs, Load load, Store store,
                                                 Advance advance, Sweep sweep) {
  float aA[AE], bA[BE], aB[AE], bB[BE];
  load(aA, bA);                                   // tile 0
  store(aA, bA, As, Bs);
  if (T > 1) { advance(); load(aA, bA); }         // tile 1 in flight in set A
  __syncthreads();
  int t = 0;
  // steady state, two tiles per trip (static register sets): buffer 0 holds tile t on entry
  while (t + 3 < T) {
    advance(); load(aB, bB);                      // tile t+2
    sweep(As, Bs);                                // tile t
    store(aA, bA, As + ASZ, Bs + BSZ);            // tile t+1
    __syncthreads();
    advance(); load(aA, bA);                      // tile t+3
    sweep(As + ASZ, Bs + BSZ);                    // tile t+1
    store(aB, bB, As, Bs);                        // tile t+2
    __syncthreads();
    t += 2;
  }
  // tail: 1..3 tiles left; set A holds tile t+1 when it exists
  bool more1 = t + 1 < T;
  int buf = 0;
  while (true) {
    const bool more2 = t + 2 < T;
    if (more2) { advance(); load(aB, bB); }
    sweep(As + buf * ASZ, Bs + buf * BSZ);
    if (more1) store(aA, bA, As + (buf ^ 1) * ASZ, Bs + (buf ^ 1) * BSZ);
    __syncthreads();
    if (!more1) break;
    buf ^= 1; ++t;
    sweep(As + buf * ASZ, Bs + buf * BSZ);
    if (more2) store(aB, bB, As + (buf ^ 1) * ASZ, Bs + (buf ^ 1) * BSZ);
    __syncthreads();
    if (!more2) break;
    buf ^= 1; ++t;
    more1 = false;                                // at most 3 tiles in the tail
  }
}

// Fused epilogue shared by the generic and the fast implicit-GEMM kernels.
// The accumulators are drained in phases of 8 rows of one 32x32 tile: first ALL operand loads of the phase
// (xhat, residual, act') are issued into registers, then the results are computed and stored — a load never sits
// behind a store that might alias it, so a phase costs one memory round trip instead of eight.
// Measured on MI355X and rejected (no gain, more VGPRs): software-pipelining the phases with two register sets,
// and issuing phase 0's loads before the K loop.
// PAR: the block's rows are class-local indices of the parity class (ph, pw) (see IgemmP); `row_of` maps them back
// to rows of the output tensor.
// This translation unit is built with -fno-slp-vectorize (Makefile): with the SLP vectoriser on, the two-phase form
// (PH = 8) of igemm_fast_kernel<4,1,1,2,SPLIT> — the 64-column tile of the bf16x3 mode — returned a wrong and
// run-to-run different red1 (sum of v * x-hat, the BN-scale cotangent) while `out` and red0 were right: 3e-4 of the
// GGN-vp at the bench geometry (round-2 record).  Localised op by op with scripts/split_localise.py / split_detail.py;
// not the LDS reduction (direct global atomics: same), not the waits (s_waitcnt 0 before the compute: same), not the
// scheduler strategy; PH = 4 or 16, or SLP off, give 4.5e-6 and reproducible sums.  Packed-f32 code (v_pk_mul/add/fma,
// v_pk_mov with op_sel) is what the vectoriser adds; f32 throughput is unchanged without it (1706 vs 1703 GGN-vp/s).
#ifndef LIP_EPI_PHASE
#define LIP_EPI_PHASE 8
#endif
// PAR == 2 (Winograd): the block's 32 accumulator rows are tiles; `rowtab[i]` (LDS) holds the tensor row of pixel (0, 0)
// of tile i or -1, the wave writes pixel (ph, pw) of every tile, `tab_full` says that no tile of the block is masked.
template <int WM, int WN, int TM, int TN, int PAR = 0>
__device__ __forceinline__ void igemm_epilogue(const IgemmP& prm, f32x16 (&acc)[TM][TN], float* redbuf, int p, int r0,
                                               int n0, int wm, int wn, int lane, int tid, int ph = 0, int pw = 0,
                                               const int* rowtab = nullptr, bool tab_full = false) {
  using T = Tile<WM, WN, TM, TN>;
  constexpr int NT = T::NT, BN = T::BN;
  const int N = prm.N, R = PAR == 1 ? prm.Rc : prm.R;
  const int tab_shift = ph * prm.OW + pw;
  auto row_of = [&](int r) -> int {
    if (PAR == 2) return rowtab[r] + tab_shift;
    if (!PAR) return r;
    const int i = prm.dOHW2.div(r), rem = r - i * prm.OHW2;
    const int a = prm.dOW2.div(rem), b = rem - a * prm.OW2;
    return i * prm.OHW + (2 * a + ph) * prm.OW + 2 * b + pw;
  };
  const int l31 = lane & 31, lh = lane >> 5;
  const bool do_red = (prm.red0 != nullptr) || (prm.red1 != nullptr);
  const float* __restrict__ xhat = prm.xhat;
  const float* __restrict__ xhat2 = prm.xhat2;
  const float* __restrict__ dphi = prm.dphi;
  const float* __restrict__ res = prm.res ? prm.res + (long long)p * prm.res_ps : nullptr;
  float* __restrict__ out = prm.out + (long long)p * prm.out_ps;
  const bool has_e1 = prm.e1 != nullptr, has_r1 = prm.red1 != nullptr;
  constexpr int PH = LIP_EPI_PHASE;                   // accumulator rows drained per phase
  const bool full_tile = PAR == 2 ? (tab_full && (n0 + BN <= N)) : ((r0 + WM * TM * 32 <= R) && (n0 + BN <= N));      // uniform: no row / column of the block is masked
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int cl = (wn * TN + tn) * 32 + l31;
    const int col = n0 + cl;
    const bool cv = col < N;
    const float sc = (prm.scale && cv) ? prm.scale[col] : 1.f;
    const float e0v = (prm.e0 && cv) ? prm.e0[(long long)p * prm.e0_ps + col] : 0.f;
    const float e1v = (has_e1 && cv) ? prm.e1[(long long)p * prm.e1_ps + col] : 0.f;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int rbase = r0 + (wm * TM + tm) * 32 + 4 * lh;
#pragma unroll
      for (int h = 0; h < 16 / PH; ++h) {
        // operand presence is uniform over the launch: ONE scalar branch per operand and phase around its eight loads,
        // and one around the guarded / unguarded form of the eight stores (a per-element test of each pointer made the
        // epilogue ~130 taken branches per 32 x 32 tile: 28 k cycles per block for a plain store, timeline study r2)
        float xv[PH], rv[PH], dv[PH], x2[PH];
        unsigned idx[PH];
        bool ok[PH];
#pragma unroll
        for (int q = 0; q < PH; ++q) {
          const int reg = PH * h + q;
          const int r = rbase + (reg & 3) + 8 * (reg >> 2);
          ok[q] = cv && (PAR == 2 ? rowtab[r] >= 0 : r < R);
          idx[q] = ok[q] ? (unsigned)(row_of(r) * N + col) : 0u;               // clamped: loads stay unconditional
        }
        if (has_e1) {
#pragma unroll
          for (int q = 0; q < PH; ++q) xv[q] = xhat[idx[q]];
        } else {
#pragma unroll
          for (int q = 0; q < PH; ++q) xv[q] = 0.f;
        }
        if (res) {
#pragma unroll
          for (int q = 0; q < PH; ++q) rv[q] = res[idx[q]];
        } else {
#pragma unroll
          for (int q = 0; q < PH; ++q) rv[q] = 0.f;
        }
        if (dphi) {
#pragma unroll
          for (int q = 0; q < PH; ++q) dv[q] = dphi[idx[q]];
        } else {
#pragma unroll
          for (int q = 0; q < PH; ++q) dv[q] = 1.f;
        }
        if (has_r1) {
#pragma unroll
          for (int q = 0; q < PH; ++q) x2[q] = xhat2[idx[q]];
        } else {
#pragma unroll
          for (int q = 0; q < PH; ++q) x2[q] = 0.f;
        }
        if (full_tile) {
#pragma unroll
          for (int q = 0; q < PH; ++q) {
            const float v = (acc[tm][tn][PH * h + q] * sc + e0v + e1v * xv[q] + rv[q]) * dv[q];
            out[idx[q]] = v;
            s0 += v;
            s1 += v * x2[q];
          }
        } else {
#pragma unroll
          for (int q = 0; q < PH; ++q) {
            if (ok[q]) {
              const float v = (acc[tm][tn][PH * h + q] * sc + e0v + e1v * xv[q] + rv[q]) * dv[q];
              out[idx[q]] = v;
              s0 += v;
              s1 += v * x2[q];
            }
          }
        }
      }
    }
    if (do_red) {
      s0 += __shfl_xor(s0, 32, 64);
      s1 += __shfl_xor(s1, 32, 64);
      if (lh == 0) {
        atomicAdd(&redbuf[cl], s0);
        atomicAdd(&redbuf[BN + cl], s1);
      }
    }
  }
  if (do_red) {
    __syncthreads();
    for (int c = tid; c < BN; c += NT) {
      const int col = n0 + c;
      if (col < N) {
        if (prm.red0) atomicAdd(prm.red0 + (long long)p * prm.red0_ps + col, redbuf[c]);
        if (prm.red1) atomicAdd(prm.red1 + (long long)p * prm.red1_ps + col, redbuf[BN + c]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// implicit GEMM
// ------------------------------------------------------------------------------------------
template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(WM * WN * 64) void igemm_kernel(const IgemmP prm) {
  using T = Tile<WM, WN, TM, TN>;
  constexpr int NT = T::NT, BM = T::BM, BN = T::BN, AE = T::AE, AQ = T::AQ, BE = T::BE;
  constexpr int LDA = BM + 2, LDB = BN;
  __shared__ float As[BK * LDA];
  __shared__ float Bs[BK * LDB];
  __shared__ float redbuf[2 * BN];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int N = prm.N, R = prm.R;
  const int tiles_n = (N + BN - 1) / BN;
  const int tile_n = blockIdx.x % tiles_n, tile_m = blockIdx.x / tiles_n;
  const int p = blockIdx.y;
  const int r0 = tile_m * BM, n0 = tile_n * BN;

  for (int i = tid; i < 2 * BN; i += NT) redbuf[i] = 0.f;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

  // rows owned by this thread in the vector A-load path: quad q = tid + j*NT -> (m = q>>2, kq = q&3)
  int vi[AQ], voh[AQ], vow[AQ];
#pragma unroll
  for (int j = 0; j < AQ; ++j) {
    const int m = (tid + j * NT) >> 2;
    const int r = r0 + m;
    if (r < R) {
      const int i = prm.dOHW.div(r), rem = r - i * prm.OHW;
      vi[j] = i; voh[j] = prm.dOW.div(rem); vow[j] = rem - voh[j] * prm.OW;
    } else {
      vi[j] = -1; voh[j] = 0; vow[j] = 0;
    }
  }

  float areg[AE], breg[BE];

  // gathered input coordinate of output coordinate o under kernel tap k (lip_seg_t modes)
  auto gather_coord = [](const SegP& s, int o, int k, int pad, int lim, int& valid) -> int {
    int t;
    if (s.mode == 0) {
      t = o * s.stride + k - pad;
    } else {
      t = o + pad - k;
      if (s.stride == 2) {
        if (t & 1) valid = 0;
        t >>= 1;                       // arithmetic shift keeps negatives negative
      } else if (s.stride != 1) {
        if (t < 0 || (t % s.stride) != 0) { valid = 0; return 0; }
        t /= s.stride;
      }
    }
    if (t < 0 || t >= lim) valid = 0;
    return t;
  };

  // K-tile cursor.  (kh, kw, c0) track the kernel tap of the tile with scalar adds only; they are
  // meaningful when C % 16 == 0 (a BK = 16 tile never straddles a tap), the common case.
  int seg = 0, k0 = 0, kh0 = 0, kw0 = 0, c0 = 0;

  auto load_tile = [&](const SegP& s) {
    const float* abase = s.a + (long long)p * s.a_ps;
    if ((s.C & 3) == 0) {
      const bool tap_uniform = (s.C & 15) == 0;
      const bool one_tap = (s.KH * s.KW) == 1;
#pragma unroll
      for (int j = 0; j < AQ; ++j) {
        const int kq = (tid + j * NT) & 3;
        const int kg = k0 + kq * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (vi[j] >= 0 && kg < s.Ktot) {
          int kh, kw, c;
          if (tap_uniform) { kh = kh0; kw = kw0; c = c0 + kq * 4; }
          else if (one_tap) { kh = 0; kw = 0; c = kg; }
          else {
            const int tap = s.dC.div(kg);
            c = kg - tap * s.C;
            kh = s.dKW.div(tap);
            kw = tap - kh * s.KW;
          }
          int valid = 1;
          const int ih = gather_coord(s, voh[j], kh, s.pad_h, s.IH, valid);
          const int iw = gather_coord(s, vow[j], kw, s.pad_w, s.IW, valid);
          if (valid)
            v = *reinterpret_cast<const float4*>(abase + (unsigned)(((vi[j] * s.IH + ih) * s.IW + iw) * s.C + c));
        }
        areg[4 * j + 0] = v.x; areg[4 * j + 1] = v.y; areg[4 * j + 2] = v.z; areg[4 * j + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < AE; ++j) {
        const int e = tid + j * NT;
        const int m = e >> 4, k = e & 15;
        const int r = r0 + m, kg = k0 + k;
        float v = 0.f;
        if (r < R && kg < s.Ktot) {
          const int i = prm.dOHW.div(r), rem = r - i * prm.OHW;
          const int oh = prm.dOW.div(rem), ow = rem - oh * prm.OW;
          const int tap = s.dC.div(kg), c = kg - tap * s.C;
          const int kh = s.dKW.div(tap), kw = tap - kh * s.KW;
          int valid = 1;
          const int ih = gather_coord(s, oh, kh, s.pad_h, s.IH, valid);
          const int iw = gather_coord(s, ow, kw, s.pad_w, s.IW, valid);
          if (valid) v = abase[(unsigned)(((i * s.IH + ih) * s.IW + iw) * s.C + c)];
        }
        areg[j] = v;
      }
    }
    const float* bbase = s.b + (long long)p * s.b_ps;
#pragma unroll
    for (int j = 0; j < BE; ++j) {
      const int e = tid + j * NT;
      const int k = e / BN, nn = e - k * BN;
      const int kg = k0 + k, col = n0 + nn;
      if (s.b_trans) {                    // B[(tap*C + c)][col] = b[(tap*N + col)*C + c]
        const int tap = s.dC.div(kg), c = kg - tap * s.C;
        breg[j] = (kg < s.Ktot && col < N) ? bbase[(unsigned)((tap * N + col) * s.C + c)] : 0.f;
      } else {
        breg[j] = (kg < s.Ktot && col < N) ? bbase[(unsigned)(kg * N + col)] : 0.f;
      }
    }
  };

  // advance the cursor to the next K-tile; false when all segments are consumed
  auto advance = [&]() -> bool {
    const SegP& s = prm.seg[seg];
    k0 += BK;
    c0 += BK;
    if (c0 >= s.C) { c0 -= s.C; if (++kw0 == s.KW) { kw0 = 0; ++kh0; } }
    if (k0 >= s.Ktot) { ++seg; k0 = 0; c0 = 0; kh0 = 0; kw0 = 0; }
    return seg < prm.nseg;
  };

  auto store_tile = [&](const SegP& s) {
    if ((s.C & 3) == 0) {
#pragma unroll
      for (int j = 0; j < AQ; ++j) {
        const int q = tid + j * NT;
        const int m = q >> 2, kq = q & 3;
#pragma unroll
        for (int t = 0; t < 4; ++t) As[(4 * kq + t) * LDA + m] = areg[4 * j + t];
      }
    } else {
#pragma unroll
      for (int j = 0; j < AE; ++j) {
        const int e = tid + j * NT;
        As[(e & 15) * LDA + (e >> 4)] = areg[j];
      }
    }
#pragma unroll
    for (int j = 0; j < BE; ++j) {
      const int e = tid + j * NT;
      const int k = e / BN, nn = e - k * BN;
      Bs[k * LDB + nn] = breg[j];
    }
  };

  load_tile(prm.seg[0]);
  while (true) {
    __syncthreads();
    store_tile(prm.seg[seg]);
    __syncthreads();
    const bool more = advance();
    if (more) load_tile(prm.seg[seg]);
    mfma_sweep<WM, WN, TM, TN, LDA, LDB>(As, Bs, acc, wm, wn, lane);
    if (!more) break;
  }

  igemm_epilogue<WM, WN, TM, TN>(prm, acc, redbuf, p, r0, n0, wm, wn, lane, tid);
}

// ------------------------------------------------------------------------------------------
// implicit GEMM, specialised straight-line variant for the hot configuration:
//   every segment has C % 16 == 0 (a BK = 16 K-tile never straddles a kernel tap), stride in {1, 2},
//   16-byte aligned activations.  Per kernel tap the gathered row offsets / validity are computed
//   once (branch-free, host-precomputed segment scalars) and reused for the C/16 K-tiles of the tap;
//   B and LDS offsets are loop invariant.  ~8 non-MFMA instructions per MFMA instead of ~30.
// ------------------------------------------------------------------------------------------
// BV: the B operand (weights, [K][N] with N contiguous) is read BW = min(4, floats per thread) channels at a time —
// global_load_dwordx2/x4 need only dword alignment (the per-probe rows of V are not 16-byte aligned), so N % BW == 0
// is the only condition.  The K loop is sensitive to the NUMBER of vector-memory instructions, not their width.
typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float float2u __attribute__((ext_vector_type(2), aligned(4)));

template <int WM, int WN, int TM, int TN, bool SPLIT = false, bool PAR = false, bool BV = false, bool KS = false>
__global__ __launch_bounds__(WM * WN * 64) void igemm_fast_kernel(const IgemmP prm) {
  static_assert(!KS || (!PAR && !SPLIT), "split-K: plain row order, exact f32");
  using T = Tile<WM, WN, TM, TN>;
  constexpr int NT = T::NT, BM = T::BM, BN = T::BN, AE = T::AE, AQ = T::AQ;
  constexpr int BE = T::BE;                                  // B floats per thread per K-tile (2, 4 or 8)
  constexpr int BW = BV ? (BE >= 4 ? 4 : BE) : 1;            // floats per B load instruction
  constexpr int NB = BE / BW;                                // B load instructions per thread per K-tile
  static_assert(!T::BPART && BE % BW == 0 && BN % BW == 0, "B vector loads need an even split");
  constexpr int LDA = BM + 2, LDB = BN;
  constexpr int LDK = 20;
#ifdef LIP_MMAJOR_A
  constexpr bool MMAJ = !SPLIT;                               // A/B build (-DLIP_MMAJOR_A): m-major A image
#else
  constexpr bool MMAJ = false;     // same-box A/B on MI355X: k-major 1566 vs m-major 1557 GGN-vp/s -> k-major stays
#endif
  constexpr int ASZ = SPLIT ? 2 * BM * SROW : (MMAJ ? BM * LDK : BK * LDA);      // floats per LDS buffer
  constexpr int BSZ = SPLIT ? 2 * BN * SROW : BK * LDB;
  __shared__ __attribute__((aligned(16))) float As[2 * ASZ];
  __shared__ __attribute__((aligned(16))) float Bs[2 * BSZ];
  __shared__ float redbuf[2 * BN];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int N = prm.N, R = PAR ? prm.Rc : prm.R;
  const int tiles_n = (N + BN - 1) / BN;
  // Workgroups go round-robin over the 8 XCDs (each with its own L2): give every XCD a CONTIGUOUS run of row tiles, so
  // the halo rows two neighbouring tiles both gather come out of one L2.  Measured on the 32-column tile (400 tiles per
  // probe): HBM reads 6.62 -> 5.18 GB per launch, time -1 %.  Not for the parity-class grid (its order is by class:
  // 4.72 -> 6.5 GB and +50 % time when remapped).
  int bid = blockIdx.x, byp = blockIdx.y;
  if (!PAR) {
    const int gx = (int)gridDim.x;
    if (gx >= 64) {
      // many row tiles per probe (large feature maps): an XCD keeps the SAME tile range for every probe, so the
      // primal operands of those rows (activations, x-hat, act') stay in its L2 across the probes
      const int g8 = gx & ~7;
      if (bid < g8) bid = (bid & 7) * (g8 >> 3) + (bid >> 3);
    } else {
      // few row tiles per probe (small feature maps, large per-probe weight slices): an XCD works through whole
      // probes, so a probe's weight slice is fetched into one L2 only (128-column tile: 3.3 -> 2.4 GB per launch;
      // on the 32- and 64-column tiles this order was measured worse: 6.9 -> 8.4 and 3.7 -> 4.5 GB)
      const int g8 = (gx * (int)gridDim.y) & ~7, lin = bid + gx * byp;
      if (lin < g8) {
        const int w = (lin & 7) * (g8 >> 3) + (lin >> 3);
        byp = w / gx; bid = w - byp * gx;
      }
    }
  }
  const int tile_n = bid % tiles_n;
  int tile_m = bid / tiles_n, ph = 0, pw = 0;
  if (PAR) {                    // 4 parity classes x tiles-per-class row tiles
    const int tpc = (prm.Rc + BM - 1) / BM, cls = tile_m / tpc;
    tile_m -= cls * tpc; ph = cls >> 1; pw = cls & 1;
  }
  const int p = byp;
  const int r0 = tile_m * BM, n0 = tile_n * BN;

  for (int i = tid; i < 2 * BN; i += NT) redbuf[i] = 0.f;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

  // thread -> (row m_j, k-quad kq): quad q = tid + j*NT, m_j = q >> 2, kq = tid & 3 (NT % 4 == 0)
  const int kq4 = (tid & 3) * 4;
  int vi[AQ], voh[AQ], vow[AQ];
#pragma unroll
  for (int j = 0; j < AQ; ++j) {
    const int r = r0 + ((tid + j * NT) >> 2);
    if (r < R) {
      if (PAR) {
        const int i = prm.dOHW2.div(r), rem = r - i * prm.OHW2;
        const int a = prm.dOW2.div(rem);
        vi[j] = i; voh[j] = 2 * a + ph; vow[j] = 2 * (rem - a * prm.OW2) + pw;
      } else {
        const int i = prm.dOHW.div(r), rem = r - i * prm.OHW;
        vi[j] = i; voh[j] = prm.dOW.div(rem); vow[j] = rem - voh[j] * prm.OW;
      }
    } else {
      vi[j] = -1; voh[j] = 0; vow[j] = 0;
    }
  }
  // loop-invariant B element offsets: e = tid + j*NT -> (k = e / BN, nn = e % BN)
  unsigned bidx[NB];
  bool bok[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int e = tid + j * NT;
    if (BW > 1) {
      const int k = e / (BN / BW), nq = e - k * (BN / BW);
      bok[j] = (n0 + BW * nq) < N;
      bidx[j] = (unsigned)(k * N + n0 + BW * nq);
    } else {
      const int k = e / BN, nn = e - k * BN;
      bok[j] = (n0 + nn) < N;
      bidx[j] = (unsigned)(k * N + n0 + nn);
    }
  }

  int rowoff[AQ];
  bool rowok[AQ];

  // segment scalars + K cursor
  int seg = 0, kh = 0, kw = 0, c0 = 0;
  const float* abase = nullptr;
  const float* bbase = nullptr;
  int sIH = 0, sIW = 0, sC = 0, sKH = 0, sKW = 0, smul = 0, ssgn = 0, soffh = 0, soffw = 0, smask = 0, ssh = 0;
  int kw0 = 0, kstep = 1;                 // PAR: first matching tap column and tap step of the segment
  const float* bseg = nullptr;

  // PAR: taps of a stride-2 transposed segment that can match the class: k = (parity + pad) & 1, step 2
  auto first_tap = [&](const SegP& s, int par, int off) { return s.mask ? ((par + off) & 1) : 0; };
  auto begin_segment = [&]() {
    if (PAR) {
      while (seg < prm.nseg - 1 && (first_tap(prm.seg[seg], ph, prm.seg[seg].off_h) >= prm.seg[seg].KH ||
                                    first_tap(prm.seg[seg], pw, prm.seg[seg].off_w) >= prm.seg[seg].KW)) ++seg;
    }
    const SegP& s = prm.seg[seg];
    abase = s.a + (long long)p * s.a_ps;
    bbase = s.b + (long long)p * s.b_ps;
    sIH = s.IH; sIW = s.IW; sC = s.C; sKH = s.KH; sKW = s.KW;
    smul = s.mul; ssgn = s.sgn; soffh = s.off_h; soffw = s.off_w; smask = s.mask; ssh = s.sh;
    kh = 0; kw = 0; c0 = 0;
    if (PAR) {
      kstep = s.mask ? 2 : 1;
      kh = first_tap(s, ph, s.off_h); kw0 = first_tap(s, pw, s.off_w); kw = kw0;
      bseg = bbase;
      bbase = bseg + (long long)((kh * sKW + kw) * sC) * N;
    }
  };
  auto set_tap = [&]() {
    const int th = ssgn * kh + soffh, tw = ssgn * kw + soffw;      // scalar
#pragma unroll
    for (int j = 0; j < AQ; ++j) {
      const int t0h = voh[j] * smul + th, t0w = vow[j] * smul + tw;
      const int ih = t0h >> ssh, iw = t0w >> ssh;
      rowok[j] = (vi[j] >= 0) && (((t0h | t0w) & smask) == 0) && ((unsigned)ih < (unsigned)sIH) &&
                 ((unsigned)iw < (unsigned)sIW);
      rowoff[j] = rowok[j] ? ((vi[j] * sIH + ih) * sIW + iw) * sC + kq4 : 0;
    }
  };
  auto load_tile = [&](float (&areg)[AE], float (&breg)[BE]) {
    const float* ap = abase + c0;
#pragma unroll
    for (int j = 0; j < AQ; ++j) {
      // masked rows read the zero page: no exec branch around the load, so vmcnt waits can be counted
      const float* src = rowok[j] ? (ap + (unsigned)rowoff[j]) : prm.zeros;
      const float4 v = *reinterpret_cast<const float4*>(src);
      areg[4 * j + 0] = v.x; areg[4 * j + 1] = v.y; areg[4 * j + 2] = v.z; areg[4 * j + 3] = v.w;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const float* src = bok[j] ? (bbase + bidx[j]) : prm.zeros;
      if (BW == 4) {
        const float4u v = *reinterpret_cast<const float4u*>(src);
        breg[4 * j + 0] = v[0]; breg[4 * j + 1] = v[1]; breg[4 * j + 2] = v[2]; breg[4 * j + 3] = v[3];
      } else if (BW == 2) {
        const float2u v = *reinterpret_cast<const float2u*>(src);
        breg[2 * j + 0] = v[0]; breg[2 * j + 1] = v[1];
      } else {
        breg[j] = *src;
      }
    }
  };
  auto store_tile = [&](const float (&areg)[AE], const float (&breg)[BE], float* Asb, float* Bsb) {
    if (SPLIT) {
#pragma unroll
      for (int j = 0; j < AQ; ++j) {
        const int m = (tid + j * NT) >> 2;
        split_store4(Asb, Asb + BM * SROW, m, kq4, areg[4 * j], areg[4 * j + 1], areg[4 * j + 2], areg[4 * j + 3]);
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int e = tid + j * NT;
        if (BW > 1) {
          const int k = e / (BN / BW), nq = e - k * (BN / BW);
#pragma unroll
          for (int t = 0; t < BW; ++t) split_store1(Bsb, Bsb + BN * SROW, BW * nq + t, k, breg[BW * j + t]);
        } else {
          const int k = e / BN, nn = e - k * BN;
          split_store1(Bsb, Bsb + BN * SROW, nn, k, breg[j]);
        }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < AQ; ++j) {
      const int m = (tid + j * NT) >> 2;
      if (MMAJ) {
        *reinterpret_cast<float4*>(&Asb[m * LDK + kq4]) = make_float4(areg[4 * j], areg[4 * j + 1], areg[4 * j + 2], areg[4 * j + 3]);
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) Asb[(kq4 + t) * LDA + m] = areg[4 * j + t];
      }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int e = tid + j * NT;
      if (BW == 4) {
        const int k = e / (BN / 4), nq = e - k * (BN / 4);
        *reinterpret_cast<float4*>(&Bsb[k * LDB + 4 * nq]) =
            make_float4(breg[4 * j + 0], breg[4 * j + 1], breg[4 * j + 2], breg[4 * j + 3]);
      } else if (BW == 2) {
        const int k = e / (BN / 2), nq = e - k * (BN / 2);
        *reinterpret_cast<float2*>(&Bsb[k * LDB + 2 * nq]) = make_float2(breg[2 * j + 0], breg[2 * j + 1]);
      } else {
        const int k = e / BN, nn = e - k * BN;
        Bsb[k * LDB + nn] = breg[j];
      }
    }
  };
  // move to the next K-tile (callers never advance past the last tile)
  auto advance = [&]() {
    c0 += BK;
    bbase += BK * N;
    if (c0 == sC) {
      c0 = 0;
      if (PAR) {
        kw += kstep;
        if (kw >= sKW) { kw = kw0; kh += kstep; }
        if (kh >= sKH) { ++seg; begin_segment(); }
        else bbase = bseg + (long long)((kh * sKW + kw) * sC) * N;
      } else {
        if (++kw == sKW) { kw = 0; ++kh; }
        if (kh == sKH) { ++seg; begin_segment(); }
      }
      set_tap();
    }
  };

  int ktiles = 0;
  for (int q = 0; q < prm.nseg; ++q) {
    const SegP& s = prm.seg[q];
    if (PAR) {
      const int st = s.mask ? 2 : 1, h0 = first_tap(s, ph, s.off_h), w0 = first_tap(s, pw, s.off_w);
      const int nh = h0 < s.KH ? (s.KH - h0 + st - 1) / st : 0, nw = w0 < s.KW ? (s.KW - w0 + st - 1) / st : 0;
      ktiles += nh * nw * (s.C / BK);
    } else {
      ktiles += s.Ktot / BK;
    }
  }
  unsigned long long t_start = 0, t_loop_end = 0;
  const unsigned dbg_lin = blockIdx.x + blockIdx.y * gridDim.x;
  const bool dbg_on = prm.dbg != nullptr && dbg_lin < 8192 && tid == 0;
  if (dbg_on) t_start = __builtin_amdgcn_s_memtime();
  if (KS) {                               // this block's share of the K-tiles: [t0, t0 + ktiles)
    const int z = (int)blockIdx.z, ks = (int)gridDim.z;
    const int t0 = (int)((long long)ktiles * z / ks), t1 = (int)((long long)ktiles * (z + 1) / ks);
    int t = t0;
    while (seg < prm.nseg - 1 && t >= prm.seg[seg].Ktot / BK) { t -= prm.seg[seg].Ktot / BK; ++seg; }
    begin_segment();
    const int tpt = sC / BK, tap = t / tpt;
    c0 = (t - tap * tpt) * BK; kh = tap / sKW; kw = tap - kh * sKW;
    bbase += (long long)t * BK * N;
    ktiles = t1 - t0;
    set_tap();
    pipelined_k_loop<AE, BE, ASZ, BSZ>(
        ktiles, As, Bs, load_tile, store_tile, advance,
        [&](const float* Asb, const float* Bsb) { mfma_sweep<WM, WN, TM, TN, LDA, LDB>(Asb, Bsb, acc, wm, wn, lane); });
    IgemmP raw = prm;                       // raw sums: no epilogue operands, rows of this share's plane
    raw.out = prm.partial + (long long)z * prm.partial_zs; raw.out_ps = (long long)prm.R * N;
    raw.scale = nullptr; raw.e0 = nullptr; raw.e1 = nullptr; raw.xhat = nullptr; raw.res = nullptr; raw.dphi = nullptr;
    raw.red0 = nullptr; raw.red1 = nullptr; raw.xhat2 = nullptr;
    igemm_epilogue<WM, WN, TM, TN, false>(raw, acc, redbuf, p, r0, n0, wm, wn, lane, tid);
    return;
  }
  if (!PAR || ktiles > 0) {               // PAR: a class no tap can reach (lone 1x1 stride-2) is all zeros
    begin_segment();
    set_tap();
    pipelined_k_loop<AE, BE, ASZ, BSZ>(
        ktiles, As, Bs, load_tile, store_tile, advance,
        [&](const float* Asb, const float* Bsb) {
          if (SPLIT) mfma_sweep_split<WM, WN, TM, TN>(Asb, Bsb, acc, wm, wn, lane);
          else if (MMAJ) mfma_sweep_mmajor<WM, WN, TM, TN, LDK, LDB>(Asb, Bsb, acc, wm, wn, lane);
          else mfma_sweep<WM, WN, TM, TN, LDA, LDB>(Asb, Bsb, acc, wm, wn, lane);
        });
  } else {
    __syncthreads();                      // redbuf zeroing visible before the epilogue's atomics
  }
  if (dbg_on) t_loop_end = __builtin_amdgcn_s_memtime();
  igemm_epilogue<WM, WN, TM, TN, PAR>(prm, acc, redbuf, p, r0, n0, wm, wn, lane, tid, ph, pw);
  if (dbg_on) {
    __builtin_amdgcn_s_waitcnt(0);                 // include the drain of this wave's stores
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    prm.dbg[3 * dbg_lin + 0] = t_start;
    prm.dbg[3 * dbg_lin + 1] = t_loop_end;
    prm.dbg[3 * dbg_lin + 2] = t_end;
  }
}

// second pass of a split-K launch: the shares' raw sums back into the accumulator layout, then the fused epilogue
template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(WM * WN * 64) void igemm_finish_kernel(const IgemmP prm, int ks) {
  using T = Tile<WM, WN, TM, TN>;
  constexpr int NT = T::NT, BM = T::BM, BN = T::BN;
  __shared__ float redbuf[2 * BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lh = lane >> 5;
  const int N = prm.N, R = prm.R;
  const int tiles_n = (N + BN - 1) / BN;
  const int tile_n = (int)blockIdx.x % tiles_n, tile_m = (int)blockIdx.x / tiles_n;
  const int p = (int)blockIdx.y;
  const int r0 = tile_m * BM, n0 = tile_n * BN;
  for (int i = tid; i < 2 * BN; i += NT) redbuf[i] = 0.f;
  f32x16 acc[TM][TN];
  const float* __restrict__ part = prm.partial + (long long)p * R * N;
  unsigned off[TM][TN][16];                // clamped element offsets: every load of a share is issued unconditionally
  bool okk[TM][TN][16];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = n0 + (wn * TN + tn) * 32 + l31;
      const int rbase = r0 + (wm * TM + tm) * 32 + 4 * lh;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int r = rbase + (reg & 3) + 8 * (reg >> 2);
        okk[tm][tn][reg] = col < N && r < R;
        off[tm][tn][reg] = okk[tm][tn][reg] ? (unsigned)(r * N + col) : 0u;
        acc[tm][tn][reg] = 0.f;
      }
    }
  for (int z = 0; z < ks; ++z) {           // one memory round trip per share: 16 TM TN independent loads in flight
    const float* __restrict__ src = part + (long long)z * prm.partial_zs;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const float v = src[off[tm][tn][reg]];
          acc[tm][tn][reg] += okk[tm][tn][reg] ? v : 0.f;
        }
  }
  __syncthreads();
  igemm_epilogue<WM, WN, TM, TN, false>(prm, acc, redbuf, p, r0, n0, wm, wn, lane, tid);
}

// ------------------------------------------------------------------------------------------
// implicit GEMM with the A operand straight from global memory into the MFMA operand registers (tiles whose waves
// own disjoint rows: four waves stacked along M, 32 or 64 columns).  In igemm_fast_kernel the gathered activation
// tile goes registers -> LDS (transposing store) -> registers although no wave ever reads another wave's rows: LDS is
// only a transposer there.  v_mfma_f32_32x32x2_f32 takes A[i = lane & 31][k = lane >> 5]; which two k's of the 16-deep
// K-tile an instruction consumes is free as long as B follows, so lane-half h takes the CONTIGUOUS channels
// 8h .. 8h+7 of its row: two 16-byte loads per K-tile ARE the eight A operands of the sweep (k-step kk multiplies
// A[row][8h + kk] with B[8h + kk][n]).  Only B (16 x BN, shared by the four waves) still goes through LDS.
// Per K-tile and lane: 2 global loads + BE/BW B loads, BE/BW LDS stores, 8 TN LDS reads, 8 TN MFMAs — the fast kernel
// issues 8 more LDS stores and 8 more LDS reads.  Register tiles are three deep (tile t+3 requested after the sweep
// of tile t), B rides along in the same cadence through a double LDS buffer.
// ------------------------------------------------------------------------------------------
template <int TM, int TN, bool BV>
__global__ __launch_bounds__(256) void igemm_adirect_kernel(const IgemmP prm) {
  constexpr int NT = 256, BM = 128 * TM, BN = 32 * TN;
  constexpr int AR = 8 * TM;                                  // A operand registers per K-tile: 8 channels of TM rows
  constexpr int BE = BN * BK / NT;                            // B floats per thread per K-tile (2 or 4)
  constexpr int BW = BV ? (BE >= 4 ? 4 : BE) : 1;
  constexpr int NB = BE / BW;
  constexpr int LDB = BN, BSZ = BK * LDB;
  __shared__ __attribute__((aligned(16))) float Bs[2 * BSZ];
  __shared__ float redbuf[2 * BN];

  const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int N = prm.N, R = prm.R;
  const int tiles_n = (N + BN - 1) / BN;
  int bid = blockIdx.x, byp = blockIdx.y;
  {   // XCD-contiguous order, as in igemm_fast_kernel
    const int gx = (int)gridDim.x;
    if (gx >= 64) {
      const int g8 = gx & ~7;
      if (bid < g8) bid = (bid & 7) * (g8 >> 3) + (bid >> 3);
    } else {
      const int g8 = (gx * (int)gridDim.y) & ~7, lin = bid + gx * byp;
      if (lin < g8) {
        const int w = (lin & 7) * (g8 >> 3) + (lin >> 3);
        byp = w / gx; bid = w - byp * gx;
      }
    }
  }
  const int tile_n = bid % tiles_n, tile_m = bid / tiles_n;
  const int p = byp;
  const int r0 = tile_m * BM, n0 = tile_n * BN;

  for (int i = tid; i < 2 * BN; i += NT) redbuf[i] = 0.f;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

  // this lane's TM rows and its 8-channel half
  int vi[TM], voh[TM], vow[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int r = r0 + (wm * TM + tm) * 32 + l31;
    if (r < R) {
      const int i = prm.dOHW.div(r), rem = r - i * prm.OHW;
      vi[tm] = i; voh[tm] = prm.dOW.div(rem); vow[tm] = rem - voh[tm] * prm.OW;
    } else {
      vi[tm] = -1; voh[tm] = 0; vow[tm] = 0;
    }
  }
  unsigned bidx[NB];
  bool bok[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int e = tid + j * NT;
    if (BW > 1) {
      const int k = e / (BN / BW), nq = e - k * (BN / BW);
      bok[j] = (n0 + BW * nq) < N;
      bidx[j] = (unsigned)(k * N + n0 + BW * nq);
    } else {
      const int k = e / BN, nn = e - k * BN;
      bok[j] = (n0 + nn) < N;
      bidx[j] = (unsigned)(k * N + n0 + nn);
    }
  }

  int rowoff[TM];
  bool rowok[TM];
  int seg = 0, kh = 0, kw = 0, c0 = 0;
  const float* abase = nullptr;
  const float* bbase = nullptr;
  int sIH = 0, sIW = 0, sC = 0, sKH = 0, sKW = 0, smul = 0, ssgn = 0, soffh = 0, soffw = 0, smask = 0, ssh = 0;

  auto begin_segment = [&]() __attribute__((always_inline)) {
    const SegP& s = prm.seg[seg];
    abase = s.a + (long long)p * s.a_ps;
    bbase = s.b + (long long)p * s.b_ps;
    sIH = s.IH; sIW = s.IW; sC = s.C; sKH = s.KH; sKW = s.KW;
    smul = s.mul; ssgn = s.sgn; soffh = s.off_h; soffw = s.off_w; smask = s.mask; ssh = s.sh;
    kh = 0; kw = 0; c0 = 0;
  };
  auto set_tap = [&]() __attribute__((always_inline)) {
    const int th = ssgn * kh + soffh, tw = ssgn * kw + soffw;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int t0h = voh[tm] * smul + th, t0w = vow[tm] * smul + tw;
      const int ih = t0h >> ssh, iw = t0w >> ssh;
      rowok[tm] = (vi[tm] >= 0) && (((t0h | t0w) & smask) == 0) && ((unsigned)ih < (unsigned)sIH) && ((unsigned)iw < (unsigned)sIW);
      rowoff[tm] = rowok[tm] ? ((vi[tm] * sIH + ih) * sIW + iw) * sC + 8 * lh : 0;
    }
  };
  auto load_tile = [&](float (&areg)[AR], float (&breg)[BE]) __attribute__((always_inline)) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const float* src = rowok[tm] ? (abase + c0 + (unsigned)rowoff[tm]) : prm.zeros;      // masked rows read the zero page
      const float4 v0 = *reinterpret_cast<const float4*>(src);
      const float4 v1 = *reinterpret_cast<const float4*>(src + 4);
      areg[8 * tm + 0] = v0.x; areg[8 * tm + 1] = v0.y; areg[8 * tm + 2] = v0.z; areg[8 * tm + 3] = v0.w;
      areg[8 * tm + 4] = v1.x; areg[8 * tm + 5] = v1.y; areg[8 * tm + 6] = v1.z; areg[8 * tm + 7] = v1.w;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const float* bs = bok[j] ? (bbase + bidx[j]) : prm.zeros;
      if (BW == 4) {
        const float4u v = *reinterpret_cast<const float4u*>(bs);
        breg[4 * j + 0] = v[0]; breg[4 * j + 1] = v[1]; breg[4 * j + 2] = v[2]; breg[4 * j + 3] = v[3];
      } else if (BW == 2) {
        const float2u v = *reinterpret_cast<const float2u*>(bs);
        breg[2 * j + 0] = v[0]; breg[2 * j + 1] = v[1];
      } else {
        breg[j] = *bs;
      }
    }
  };
  auto store_b = [&](const float (&breg)[BE], float* Bsb) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int e = tid + j * NT;
      if (BW == 4) {
        const int k = e / (BN / 4), nq = e - k * (BN / 4);
        *reinterpret_cast<float4*>(&Bsb[k * LDB + 4 * nq]) = make_float4(breg[4 * j + 0], breg[4 * j + 1], breg[4 * j + 2], breg[4 * j + 3]);
      } else if (BW == 2) {
        const int k = e / (BN / 2), nq = e - k * (BN / 2);
        *reinterpret_cast<float2*>(&Bsb[k * LDB + 2 * nq]) = make_float2(breg[2 * j + 0], breg[2 * j + 1]);
      } else {
        const int k = e / BN, nn = e - k * BN;
        Bsb[k * LDB + nn] = breg[j];
      }
    }
  };
  auto advance = [&]() __attribute__((always_inline)) {
    c0 += BK;
    bbase += BK * N;
    if (c0 == sC) {
      c0 = 0;
      if (++kw == sKW) { kw = 0; ++kh; }
      if (kh == sKH) { ++seg; begin_segment(); }
      set_tap();
    }
  };
  auto sweep = [&](const float (&areg)[AR], const float* Bsb) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      float b[TN];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) b[tn] = Bsb[(8 * lh + kk) * LDB + tn * 32 + l31];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[8 * tm + kk], b[tn], acc[tm][tn], 0, 0, 0);
    }
  };

  int T = 0;
  for (int q = 0; q < prm.nseg; ++q) T += prm.seg[q].Ktot / BK;

  float a0[AR], a1[AR], a2[AR], b0[BE], b1[BE], b2[BE];
  begin_segment();
  set_tap();
  load_tile(a0, b0);
  store_b(b0, Bs);
  if (T > 1) { advance(); load_tile(a1, b1); }
  if (T > 2) { advance(); load_tile(a2, b2); }
  __syncthreads();
  // iteration t: A of tile t in set t % 3, B of tile t in LDS buffer t & 1, B of tile t+1 in register set (t+1) % 3
#define LIP_AD_ITER(CUR, NXT, AC, BC, BNX, DO_STORE, DO_LOAD)                     \
  do {                                                                            \
    if (DO_STORE) store_b(BNX, Bs + (NXT) * BSZ);                                 \
    sweep(AC, Bs + (CUR) * BSZ);                                                  \
    if (DO_LOAD) { advance(); load_tile(AC, BC); }                                \
    __syncthreads();                                                              \
  } while (0)
  int t = 0;
  while (t + 8 < T) {
    LIP_AD_ITER(0, 1, a0, b0, b1, true, true);
    LIP_AD_ITER(1, 0, a1, b1, b2, true, true);
    LIP_AD_ITER(0, 1, a2, b2, b0, true, true);
    LIP_AD_ITER(1, 0, a0, b0, b1, true, true);
    LIP_AD_ITER(0, 1, a1, b1, b2, true, true);
    LIP_AD_ITER(1, 0, a2, b2, b0, true, true);
    t += 6;
  }
  while (t < T) {
    LIP_AD_ITER(0, 1, a0, b0, b1, t + 1 < T, t + 3 < T); if (++t >= T) break;
    LIP_AD_ITER(1, 0, a1, b1, b2, t + 1 < T, t + 3 < T); if (++t >= T) break;
    LIP_AD_ITER(0, 1, a2, b2, b0, t + 1 < T, t + 3 < T); if (++t >= T) break;
    LIP_AD_ITER(1, 0, a0, b0, b1, t + 1 < T, t + 3 < T); if (++t >= T) break;
    LIP_AD_ITER(0, 1, a1, b1, b2, t + 1 < T, t + 3 < T); if (++t >= T) break;
    LIP_AD_ITER(1, 0, a2, b2, b0, t + 1 < T, t + 3 < T); ++t;
  }
#undef LIP_AD_ITER
  igemm_epilogue<4, 1, TM, TN, false>(prm, acc, redbuf, p, r0, n0, wm, 0, lane, tid);
}

// ------------------------------------------------------------------------------------------
// Winograd F(2x2, 3x3) implicit GEMM (round 3) for the 3x3 / stride 1 / pad 1 layers of the tangent and backward tapes
// — 95 % of a CIFAR-net sweep's FLOPs.  Y = A^T [ (G w G^T) (.) (B^T d B) ] A per 2x2 output patch: 16 positions
// xi = (a, b), each a GEMM over channels  M_xi[tile][n] = sum_c V_xi[tile][c] U_xi[c][n]  — 4 multiplications per
// output pixel and (c, n) where the direct form needs 9, on the same v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate;
// the transforms add 8 f32 additions per operand and move a layer's result by ~2e-7 relative, scripts/micro/
// wino_probe.hip).  One block = 32 tiles (128 output pixels: NI images x BH x BW tiles, a rectangle so that the input
// footprint is small) x 32 columns; wave a owns row a of the 4x4 transformed patch:
//   * A operand: the block's input footprint (NI x (2 BH + 2) x (2 BW + 2) pixels, 32 channels per chunk) goes
//     global -> registers -> LDS in WHOLE 128-byte lines (8 lanes per pixel; out-of-image pixels are dropped by the range
//     check of the buffer descriptor and arrive as zeros), pixel slots padded to 144 bytes; lane (tile i, half h) reads
//     the 2 x 4 pixels of rows r1(a), r2(a) of its tile's patch, 4 channels each (ds_read_b128), and forms V[a][0..3] in
//     registers = the A registers of 16 MFMAs (k-step (g, j): lane half h <-> channel 8g + 4h + j).  The first version
//     fetched those 16-byte pieces straight from global memory — 32 different lines per lane-instruction: its speed was
//     set by line fills and load latency, not by the matrix pipe (timing ablation: every lane reading pixel 0 ran
//     1.4 - 2.0x faster); the LDS-staged form: 2.31 -> 1.72 / 1.54 -> 1.32 / 1.29 -> 1.14 ms on the three CIFAR stages;
//   * B operand: the transformed weights U, stored [xi][C/4][N][4] by wino_weight_transform_kernel so that lane (n, h)
//     reads its 4 channels of U_xi[.][n] as ONE dwordx4 straight into the MFMA registers (512 contiguous bytes per
//     half-wave), requested one group ahead (a scheduling barrier keeps the compiler from sinking the loads to their
//     uses: with them there a group of 16 MFMAs took 5 k cycles);
//   * output transform: over b inside the wave, over a across the four waves through LDS; wave w then owns output
//     pixel (w >> 1, w & 1) of every tile and runs the SAME fused epilogue as the direct kernels (row table in LDS).
// Several K-segments (tangent: conv(da, W) + conv(a, dW_p)) accumulate in the transformed domain: one output transform.
// Mode-1 segments (data gradient, stride 1) are the same correlation with the kernel flipped — the weight transform does it.
// Never used for the primal tape (ReLU gates / pooling arg-maxima are taken from the direct sums).
// ------------------------------------------------------------------------------------------
typedef float f32x4v __attribute__((ext_vector_type(4)));
struct WinoX {
  unsigned a_bytes[3]; unsigned u_bytes[3];
  int BWs, BHs, NI, FR, FC, nbx, nby, NS, n_img, TH, TW;
  float inv_frfc, inv_fc;
  FastDiv dnb, dbxy, dnbx;              // N / 32, nbx * nby, nbx
};

// U[xi = 4a + b][c / 4][n][c % 4] = (G w G^T)[a][b] of the 3x3 kernel w[kh][kw][c][n]  (flip: w[2-kh][2-kw]).
// Thread (c quad, n): 9 x 4 coalesced dword reads, 16 float4 writes (consecutive n -> consecutive 16 bytes).
__global__ __launch_bounds__(256) void wino_weight_transform_kernel(const float* __restrict__ w, long long w_ps, float* __restrict__ u,
                                                                     long long u_ps, int C, int N, int flip) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int c4 = C >> 2;
  if (e >= c4 * N) return;
  const int cq = e / N, n = e - cq * N;
  const float* wp = w + (long long)blockIdx.y * w_ps;
  float* up = u + (long long)blockIdx.y * u_ps;
  f32x4v uu[16];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = 4 * cq + j;
    float g[3][3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int sh = flip ? 2 - kh : kh, sw = flip ? 2 - kw : kw;
        g[kh][kw] = wp[((long long)(sh * 3 + sw) * C + c) * N + n];
      }
    float t[4][3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      t[0][kw] = g[0][kw];
      t[1][kw] = 0.5f * (g[0][kw] + g[1][kw] + g[2][kw]);
      t[2][kw] = 0.5f * (g[0][kw] - g[1][kw] + g[2][kw]);
      t[3][kw] = g[2][kw];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      uu[4 * a + 0][j] = t[a][0];
      uu[4 * a + 1][j] = 0.5f * (t[a][0] + t[a][1] + t[a][2]);
      uu[4 * a + 2][j] = 0.5f * (t[a][0] - t[a][1] + t[a][2]);
      uu[4 * a + 3][j] = t[a][2];
    }
  }
#pragma unroll
  for (int xi = 0; xi < 16; ++xi) *reinterpret_cast<f32x4v*>(&up[(((long long)xi * c4 + cq) * N + n) * 4]) = uu[xi];
}

// Fused epilogue of the Winograd kernel with 16-byte accesses (same arithmetic and operand meaning as igemm_epilogue):
// after the cross-wave exchange a wave owns pixel (po, qo) of the block's 32 tiles x 32 columns; lane (ti = lane >> 3,
// cq = lane & 7) takes columns 4 cq .. 4 cq + 3 of tiles ti, ti + 8, ti + 16, ti + 24 — four consecutive columns of one
// tile are four consecutive floats of the exchange area ([a][q][reg][lane], lane = column) and of every operand tensor,
// so the three exchange reads, the operand loads and the store of a tile are one ds_read_b128 / dwordx4 each (20 memory
// instructions per lane where the accumulator-layout epilogue issues 80).  Used when every operand is 16-byte aligned.
__device__ __forceinline__ void wino_epilogue_v4(const IgemmP& prm, const float* xch, float* redbuf, const int* rowtab, int p, int n0,
                                                 int lane, int tid, int a, bool blk_full) {
  const int po = a >> 1, qo = a & 1;
  const int cq = lane & 7, ti = lane >> 3;
  const int N = prm.N;
  const int col = n0 + 4 * cq;
  const int shift = po * prm.OW + qo;
  const bool has_e1 = prm.e1 != nullptr, has_r1 = prm.red1 != nullptr;
  const bool do_red = (prm.red0 != nullptr) || (prm.red1 != nullptr);
  const float* __restrict__ res = prm.res ? prm.res + (long long)p * prm.res_ps : nullptr;
  float* __restrict__ out = prm.out + (long long)p * prm.out_ps;
  float sc[4], e0v[4], e1v[4], s0[4], s1[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    sc[c] = prm.scale ? prm.scale[col + c] : 1.f;
    e0v[c] = prm.e0 ? prm.e0[(long long)p * prm.e0_ps + col + c] : 0.f;
    e1v[c] = has_e1 ? prm.e1[(long long)p * prm.e1_ps + col + c] : 0.f;
    s0[c] = 0.f; s1[c] = 0.f;
  }
  f32x4v y[4], xv[4], rv[4], dv[4], x2[4];
  unsigned idx[4];
  bool ok[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = ti + 8 * k;
    const int base = ((i & 3) + 4 * (i >> 3)) * 64 + ((i >> 2) & 1) * 32 + 4 * cq;      // accumulator row i, columns 4 cq ..
    const f32x4v t1 = *reinterpret_cast<const f32x4v*>(&xch[(1 * 2 + qo) * 1024 + base]);
    const f32x4v t2 = *reinterpret_cast<const f32x4v*>(&xch[(2 * 2 + qo) * 1024 + base]);
    const f32x4v t03 = *reinterpret_cast<const f32x4v*>(&xch[((po ? 3 : 0) * 2 + qo) * 1024 + base]);
    y[k] = po ? (t1 - t2 - t03) : (t03 + t1 + t2);
    const int rt = rowtab[i];
    ok[k] = rt >= 0;
    idx[k] = ok[k] ? (unsigned)((rt + shift) * N + col) : 0u;                             // clamped: loads stay unconditional
  }
  const f32x4v zero4 = {0.f, 0.f, 0.f, 0.f}, one4 = {1.f, 1.f, 1.f, 1.f};
  if (has_e1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) xv[k] = *reinterpret_cast<const f32x4v*>(prm.xhat + idx[k]);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) xv[k] = zero4;
  }
  if (res) {
#pragma unroll
    for (int k = 0; k < 4; ++k) rv[k] = *reinterpret_cast<const f32x4v*>(res + idx[k]);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) rv[k] = zero4;
  }
  if (prm.dphi) {
#pragma unroll
    for (int k = 0; k < 4; ++k) dv[k] = *reinterpret_cast<const f32x4v*>(prm.dphi + idx[k]);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) dv[k] = one4;
  }
  if (has_r1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) x2[k] = *reinterpret_cast<const f32x4v*>(prm.xhat2 + idx[k]);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) x2[k] = zero4;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f32x4v v;
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = (y[k][c] * sc[c] + e0v[c] + e1v[c] * xv[k][c] + rv[k][c]) * dv[k][c];
    if (blk_full || ok[k]) {
      *reinterpret_cast<f32x4v*>(out + idx[k]) = v;
#pragma unroll
      for (int c = 0; c < 4; ++c) { s0[c] += v[c]; s1[c] += v[c] * x2[k][c]; }
    }
  }
  if (do_red) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int off = 8; off < 64; off <<= 1) {
        s0[c] += __shfl_xor(s0[c], off, 64);
        s1[c] += __shfl_xor(s1[c], off, 64);
      }
    }
    if (ti == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        atomicAdd(&redbuf[4 * cq + c], s0[c]);
        atomicAdd(&redbuf[32 + 4 * cq + c], s1[c]);
      }
    }
    __syncthreads();
    if (tid < 32) {
      if (prm.red0) atomicAdd(prm.red0 + (long long)p * prm.red0_ps + n0 + tid, redbuf[tid]);
      if (prm.red1) atomicAdd(prm.red1 + (long long)p * prm.red1_ps + n0 + tid, redbuf[32 + tid]);
    }
  }
}

constexpr int WINO_SLOTS = 224;     // LDS pixel slots of the staged footprint (9 x 16 bytes each)

template <bool VEPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void igemm_wino_kernel(const IgemmP prm, const WinoX wx) {
  // stage [WINO_SLOTS][36] floats, later the exchange [4 a][2 q][16 reg][64 lane] (aliased); then rowtab[32], redbuf[64]
  extern __shared__ __attribute__((aligned(16))) float wino_lds[];
  constexpr int BN = 32;
  int* rowtab = reinterpret_cast<int*>(wino_lds + 8192);
  float* redbuf = wino_lds + 8192 + 32;
  f32x4v* lds4 = reinterpret_cast<f32x4v*>(wino_lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int a = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int N = prm.N;
  const int nb = N / BN;
  int bid = blockIdx.x, byp = blockIdx.y;
  {   // XCD-contiguous order, as in igemm_fast_kernel
    const int gx = (int)gridDim.x;
    if (gx >= 64) {
      const int g8 = gx & ~7;
      if (bid < g8) bid = (bid & 7) * (g8 >> 3) + (bid >> 3);
    } else {
      const int g8 = (gx * (int)gridDim.y) & ~7, lin = bid + gx * byp;
      if (lin < g8) {
        const int w = (lin & 7) * (g8 >> 3) + (lin >> 3);
        byp = w / gx; bid = w - byp * gx;
      }
    }
  }
  const int tbid = wx.dnb.div(bid), cb = bid - tbid * nb;
  const int bxy = wx.nbx * wx.nby;
  const int bi = wx.dbxy.div(tbid), brem = tbid - bi * bxy;
  const int by = wx.dnbx.div(brem), bx = brem - by * wx.nbx;
  const int p = byp;
  const int n0 = cb * BN;
  if (tid < 2 * BN) redbuf[tid] = 0.f;

  const int W = prm.OW, H = prm.OHW / prm.OW;
  const int BW = 1 << wx.BWs, BH = 1 << wx.BHs, FR = wx.FR, FC = wx.FC;
  // this lane's tile
  const int ni = l31 >> (wx.BWs + wx.BHs), dy = (l31 >> wx.BWs) & (BH - 1), dx = l31 & (BW - 1);
  {
    const int img = bi * wx.NI + ni, ty = by * BH + dy, tx = bx * BW + dx;
    const bool tv = img < wx.n_img && ty < wx.TH && tx < wx.TW;
    if (tid < 32) rowtab[tid] = tv ? (img * H + 2 * ty) * W + 2 * tx : -1;
  }
  const bool blk_full = (bi * wx.NI + wx.NI <= wx.n_img) && (by * BH + BH <= wx.TH) && (bx * BW + BW <= wx.TW);
  // rows of the 4x4 patch this wave's transform row needs:  e = d[r1] + sg d[r2]   (B^T rows: d0-d2, d1+d2, d2-d1, d1-d3)
  const int r1 = (a == 0) ? 0 : (a == 2 ? 2 : 1);
  const int r2 = (a == 0) ? 2 : (a == 1 ? 2 : (a == 2 ? 1 : 3));
  const float sg = (a == 1) ? 1.f : -1.f;
  const int rq1 = ((ni * FR + 2 * dy + r1) * FC + 2 * dx) * 9 + h;          // LDS read bases in 16-byte units
  const int rq2 = ((ni * FR + 2 * dy + r2) * FC + 2 * dx) * 9 + h;
  // staging: thread -> (pixel slot, 16-byte part) x 7
  const int part = tid & 7;
  int spix[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int slot = (tid >> 3) + 32 * j;
    const int sni = (int)(((float)slot + 0.5f) * wx.inv_frfc), srem = slot - sni * FR * FC;     // exact: slot < 256
    const int fr = (int)(((float)srem + 0.5f) * wx.inv_fc), fc = srem - fr * FC;
    const int simg = bi * wx.NI + sni, ih = 2 * by * BH - 1 + fr, iw = 2 * bx * BW - 1 + fc;
    const bool ok = slot < wx.NS && simg < wx.n_img && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    spix[j] = ok ? (simg * H + ih) * W + iw : -1;
  }

  f32x16 acc[4];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

  f32x4v sreg[7], bq[2][4];
  // (buffer loads are cast to float vectors whole: element access through __builtin_bit_cast(float, v[j]) on the
  //  unsigned vector the builtin returns is miscompiled by hipcc 7.2 — only element 0 survives)
  auto stage_load = [&](int seg, int k) __attribute__((always_inline)) {
    const SegP& s = prm.seg[seg];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s.a + (long long)p * s.a_ps), 0, wx.a_bytes[seg], 0x00020000);
    const int C = s.C;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const unsigned vo = spix[j] >= 0 ? (unsigned)((spix[j] * C + 4 * part) * 4) : 0x80000000u;
      sreg[j] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, k * 128, 0));
    }
  };
  auto stage_store = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int slot = (tid >> 3) + 32 * j;
      if (slot < WINO_SLOTS) lds4[slot * 9 + part] = sreg[j];
    }
  };
  auto load_b = [&](int seg, int g, f32x4v (&dst)[4]) __attribute__((always_inline)) {
    const SegP& s = prm.seg[seg];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s.b + (long long)p * s.b_ps), 0, wx.u_bytes[seg], 0x00020000);
    const int c4 = s.C >> 2;
    const unsigned uvoff = (unsigned)((((4 * a) * c4 + h) * N + n0 + l31) * 16);
    const unsigned ub_stride = (unsigned)(c4 * N * 16), ug_stride = (unsigned)(2 * N * 16);
#pragma unroll
    for (int b = 0; b < 4; ++b) dst[b] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs, uvoff, b * ub_stride + g * ug_stride, 0));
  };
  // one 8-channel group: patch pixels from LDS, the NEXT group's B requested, then (behind a scheduling barrier) the
  // input transform and the 16 MFMAs
  auto group = [&](int g, int seg_n, int g_n, const f32x4v (&bc)[4], f32x4v (&bn)[4]) __attribute__((always_inline)) {
    float v[4][4];
    f32x4v raw[8];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      raw[cc] = lds4[rq1 + cc * 9 + 2 * g];
      raw[4 + cc] = lds4[rq2 + cc * 9 + 2 * g];
    }
    load_b(seg_n, g_n, bn);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float e[4];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) e[cc] = fmaf(sg, raw[4 + cc][j], raw[cc][j]);
      v[0][j] = e[0] - e[2]; v[1][j] = e[1] + e[2]; v[2][j] = e[2] - e[1]; v[3][j] = e[1] - e[3];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[b][j], bc[b][j], acc[b], 0, 0, 0);
  };

  stage_load(0, 0);
  load_b(0, 0, bq[0]);
  const int nseg = prm.nseg;
  for (int seg = 0; seg < nseg; ++seg) {
    const int chunks = prm.seg[seg].C >> 5, GT = prm.seg[seg].C >> 3;
    const bool more_seg = seg + 1 < nseg;
    for (int k = 0; k < chunks; ++k) {
      stage_store();
      __syncthreads();
      const int g0 = 4 * k;
      const bool last = k + 1 == chunks;
      // what follows this chunk: the next chunk of the segment, the first of the next segment, or (at the very end) this
      // chunk again — a redundant request instead of a branch in the loop
      const int seg_n = (last && more_seg) ? seg + 1 : seg;
      const int k_n = last ? (more_seg ? 0 : k) : k + 1;
      const int g_n = last ? (more_seg ? 0 : GT - 1) : g0 + 4;
      __builtin_amdgcn_sched_barrier(0);
      group(0, seg, g0 + 1, bq[0], bq[1]);
      group(1, seg, g0 + 2, bq[1], bq[0]);
      group(2, seg, g0 + 3, bq[0], bq[1]);
      group(3, seg_n, g_n, bq[1], bq[0]);
      __builtin_amdgcn_sched_barrier(0);
      stage_load(seg_n, k_n);
      __syncthreads();
    }
  }

  // output transform, in-wave part (A^T rows: m0+m1+m2, m1-m2-m3 over b) ...
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float m0 = acc[0][r], m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r];
    wino_lds[((a * 2 + 0) * 16 + r) * 64 + lane] = m0 + m1 + m2;
    wino_lds[((a * 2 + 1) * 16 + r) * 64 + lane] = m1 - m2 - m3;
  }
  __syncthreads();
  // ... and across the waves over a: wave w owns output pixel (po, qo) of every tile
  if (VEPI) {
    wino_epilogue_v4(prm, wino_lds, redbuf, rowtab, p, n0, lane, tid, a, blk_full);
    return;
  }
  const int po = a >> 1, qo = a & 1;
  f32x16 y[1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float t1 = wino_lds[((1 * 2 + qo) * 16 + r) * 64 + lane];
    const float t2 = wino_lds[((2 * 2 + qo) * 16 + r) * 64 + lane];
    const float t03 = wino_lds[(((po ? 3 : 0) * 2 + qo) * 16 + r) * 64 + lane];
    y[0][0][r] = po ? (t1 - t2 - t03) : (t03 + t1 + t2);
  }
  igemm_epilogue<1, 1, 1, 1, 2>(prm, y, redbuf, p, 0, n0, 0, 0, lane, tid, po, qo, rowtab, blk_full);
}

// ------------------------------------------------------------------------------------------
// first layer of a conv net (round 3): ONE K-segment with a short reduction (K = KH KW C <= 64: 3 x 3 x 3 = 27 for the
// CIFAR nets), the A operand the PRIMAL input — shared by all probes — and 32 output columns.  The generic kernel ran it
// at 23 TFLOP/s (0.98 ms per 256-probe block): per probe it re-gathers the same 27-deep im2col rows.  Here a wave
// gathers the im2col rows of its 64 output rows ONCE into MFMA operand registers (lane (i = row, k = lane >> 5) of
// k-step kk holds element 2 kk + k; zero past K), keeps the epilogue's probe-independent operands (x-hat, act') of
// its 64 x 32 tile in registers too, and walks over the probes of its probe group: per probe 14 coalesced dword
// loads of the weight tangent (the B operand straight into the MFMA registers), 2 x 14 MFMAs, one fused
// BN-tangent / act' epilogue, 32 row-segment stores.  No LDS, no barrier; output-write bound.
// ------------------------------------------------------------------------------------------
template <int KK>
__global__ __launch_bounds__(256) void igemm_first_kernel(const IgemmP prm, int P, int pgroups) {
  constexpr int TM = 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const SegP& sg = prm.seg[0];
  const int R = prm.R, K = sg.Ktot;
  const int rgroup = (int)blockIdx.x * 4 + wave;               // 64 output rows per wave
  const int r0 = rgroup * 64;
  if (r0 >= R) return;
  // ---- A: im2col rows of this lane's TM rows, gathered once (rows >= R and taps outside the image read element 0 and are zeroed)
  float areg[TM][KK];
  const int KWC = sg.KW * sg.C;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int r = r0 + 32 * tm + l31;
    const bool rv = r < R;
    const int rc = rv ? r : 0;
    const int i = prm.dOHW.div(rc), rem = rc - i * prm.OHW;
    const int oh = prm.dOW.div(rem), ow = rem - oh * prm.OW;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const int k = 2 * kk + lh;
      const int kh = k / KWC, kw = (k - kh * KWC) / sg.C, c = k - kh * KWC - kw * sg.C;
      const int ih = oh * sg.stride + kh - sg.pad_h, iw = ow * sg.stride + kw - sg.pad_w;
      const bool ok = rv && k < K && (unsigned)ih < (unsigned)sg.IH && (unsigned)iw < (unsigned)sg.IW;
      const float t = sg.a[ok ? (unsigned)(((i * sg.IH + ih) * sg.IW + iw) * sg.C + c) : 0u];
      areg[tm][kk] = ok ? t : 0.f;
    }
  }
  // ---- probe-independent epilogue operands of the wave's two 32 x 32 tiles, in the accumulator layout
  const bool has_e1 = prm.e1 != nullptr, has_d = prm.dphi != nullptr;
  float xh[TM][16], dv[TM][16];
  unsigned eoff[TM];                                           // element offset of accumulator register 0
  bool full = r0 + 64 <= R;                                    // uniform
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    eoff[tm] = (unsigned)((r0 + 32 * tm + 4 * lh) * 32 + l31);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int r = r0 + 32 * tm + 4 * lh + (q & 3) + 8 * (q >> 2);
      const unsigned e = (unsigned)(min(r, R - 1) * 32 + l31);
      xh[tm][q] = has_e1 ? prm.xhat[e] : 0.f;
      dv[tm][q] = has_d ? prm.dphi[e] : 1.f;
    }
  }
  const float sc = prm.scale ? prm.scale[l31] : 1.f;
  // ---- the probes of this wave's group
  const int pg = (int)blockIdx.y;
  const int per = (P + pgroups - 1) / pgroups;
  const int p_end = min(P, (pg + 1) * per);
  const unsigned bl = (unsigned)(lh * 32 + l31);               // lane part of the B offset: row (2 kk + lh), column l31
  auto load_b = [&](int p, float (&b)[KK]) {
    const float* __restrict__ bp = sg.b + (long long)p * sg.b_ps;          // wave-uniform
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const bool pair = 2 * kk + 1 < K;                                     // uniform; the last k-step of an odd K reads row K - 1 (A is zero there)
      b[kk] = (bp + (pair ? 2 * kk * 32 : (2 * kk < K ? (K - 1) * 32 : 0)))[pair ? bl : (unsigned)l31];
    }
  };
  int p = pg * per;
  if (p >= p_end) return;
  float b[KK], bn[KK];
  load_b(p, b);
  for (; p < p_end; ++p) {
    load_b(min(p + 1, p_end - 1), bn);
    const float e0v = prm.e0 ? prm.e0[(long long)p * prm.e0_ps + l31] : 0.f;
    const float e1v = has_e1 ? prm.e1[(long long)p * prm.e1_ps + l31] : 0.f;
    f32x16 acc[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[tm][q] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) acc[tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[tm][kk], b[kk], acc[tm], 0, 0, 0);
    float* __restrict__ out = prm.out + (long long)p * prm.out_ps;          // wave-uniform
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float v = (acc[tm][q] * sc + e0v + e1v * xh[tm][q]) * dv[tm][q];
        if (full || r0 + 32 * tm + 4 * lh + (q & 3) + 8 * (q >> 2) < R) (out + ((q & 3) + 8 * (q >> 2)) * 32)[eoff[tm]] = v;
      }
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) b[kk] = bn[kk];
  }
}

static bool igemm_first_ok(const IgemmP& p, int P) {
  static const bool off = getenv("LIP_NOFIRST") != nullptr || getenv("LIP_GENERIC") != nullptr;        // A/B switch
  if (off || precision_mode() != 0 || p.nseg != 1 || P < 8) return false;
  const SegP& s = p.seg[0];
  return s.mode == 0 && s.Ktot <= 64 && (s.C & 15) != 0 && p.N == 32 && s.a_ps == 0 && !s.b_trans && !p.res && !p.red0 &&
         !p.red1 && (!p.e1 || p.xhat) && (long long)p.R * 32 < (1ll << 31);
}

static hipError_t run_igemm_first(const IgemmP& p, int P, hipStream_t st) {
  const int rblocks = (p.R + 255) / 256;                                     // 4 waves x 64 rows
  int pgroups = (int)((8ll * 256 * 4 / 4 + rblocks - 1) / rblocks);          // ~8 waves per SIMD's worth of wave items over the chip
  if (pgroups > P / 8) pgroups = P / 8;                                      // >= 8 probes per wave amortise its A / x-hat / act' gathers
  if (pgroups < 1) pgroups = 1;
  dim3 grid((unsigned)rblocks, (unsigned)pgroups);
  if (p.seg[0].Ktot <= 28) hipLaunchKernelGGL((igemm_first_kernel<14>), grid, dim3(256), 0, st, p, P, pgroups);
  else hipLaunchKernelGGL((igemm_first_kernel<32>), grid, dim3(256), 0, st, p, P, pgroups);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------
template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(WM * WN * 64) void wgrad_kernel(const WgradP prm) {
  using T = Tile<WM, WN, TM, TN>;
  constexpr int NT = T::NT, BM = T::BM, BN = T::BN, AE = T::AE, AQ = T::AQ, BE = T::BE;
  constexpr int LDA = BM + 4, LDB = BN;
  constexpr int QPR = BM / 4;              // float4 per LDS row
  __shared__ __attribute__((aligned(16))) float As[BK * LDA];
  __shared__ float Bs[BK * LDB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int N = prm.N, M = prm.M;
  const int tiles_n = (N + BN - 1) / BN;
  const int tile_n = blockIdx.x % tiles_n, tile_m = blockIdx.x / tiles_n;
  const int p = blockIdx.y;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  int rows_per = (prm.R + prm.ksplit - 1) / prm.ksplit;
  rows_per = (rows_per + BK - 1) / BK * BK;
  if (prm.seg_rows) rows_per = prm.seg_rows;           // per-example rows: block z reduces example z only
  const int rbeg = blockIdx.z * rows_per;
  const int rend = min(prm.R, rbeg + rows_per);
  if (rbeg >= rend) return;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

  const bool vec = (prm.C & 3) == 0;
  // this thread's m (fixed across the K loop): vector path m = m0 + 4*(tid % QPR); scalar m0 + tid % BM
  const int my_m = vec ? (m0 + 4 * (tid % QPR)) : (m0 + (tid % BM));
  int kh = 0, kw = 0, ci = 0;
  const bool mvalid = my_m < M;
  if (mvalid) {
    const int tap = my_m / prm.C;
    ci = my_m - tap * prm.C;
    kh = tap / prm.KW;
    kw = tap - kh * prm.KW;
  }

  float areg[AE], breg[BE];
  const float* gbase = prm.g + (long long)p * prm.g_ps;

  auto load_tile = [&](int rk0) {
    if (vec) {
#pragma unroll
      for (int j = 0; j < AQ; ++j) {
        const int k = (tid + j * NT) / QPR;
        const int r = rk0 + k;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mvalid && r < rend) {
          const int i = prm.dOHW.div(r), rem = r - i * prm.OHW;
          const int oh = prm.dOW.div(rem), ow = rem - oh * prm.OW;
          const int ih = oh * prm.stride + kh - prm.pad_h, iw = ow * prm.stride + kw - prm.pad_w;
          if (ih >= 0 && ih < prm.IH && iw >= 0 && iw < prm.IW)
            v = *reinterpret_cast<const float4*>(prm.a + (unsigned)(((i * prm.IH + ih) * prm.IW + iw) * prm.C + ci));
        }
        areg[4 * j + 0] = v.x; areg[4 * j + 1] = v.y; areg[4 * j + 2] = v.z; areg[4 * j + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < AE; ++j) {
        const int k = (tid + j * NT) / BM;
        const int r = rk0 + k;
        float v = 0.f;
        if (mvalid && r < rend) {
          const int i = prm.dOHW.div(r), rem = r - i * prm.OHW;
          const int oh = prm.dOW.div(rem), ow = rem - oh * prm.OW;
          const int ih = oh * prm.stride + kh - prm.pad_h, iw = ow * prm.stride + kw - prm.pad_w;
          if (ih >= 0 && ih < prm.IH && iw >= 0 && iw < prm.IW)
            v = prm.a[(unsigned)(((i * prm.IH + ih) * prm.IW + iw) * prm.C + ci)];
        }
        areg[j] = v;
      }
    }
#pragma unroll
    for (int j = 0; j < BE; ++j) {
      const int e = tid + j * NT;
      const int k = e / BN, nn = e - k * BN;
      const int r = rk0 + k, col = n0 + nn;
      breg[j] = (r < rend && col < N) ? gbase[(unsigned)(r * N + col)] : 0.f;
    }
  };

  auto store_tile = [&]() {
    if (vec) {
#pragma unroll
      for (int j = 0; j < AQ; ++j) {
        const int q = tid + j * NT;
        const int k = q / QPR, mq = q - k * QPR;
        *reinterpret_cast<float4*>(&As[k * LDA + 4 * mq]) =
            make_float4(areg[4 * j + 0], areg[4 * j + 1], areg[4 * j + 2], areg[4 * j + 3]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < AE; ++j) {
        const int e = tid + j * NT;
        const int k = e / BM, mm = e - k * BM;
        As[k * LDA + mm] = areg[j];
      }
    }
#pragma unroll
    for (int j = 0; j < BE; ++j) {
      const int e = tid + j * NT;
      const int k = e / BN, nn = e - k * BN;
      Bs[k * LDB + nn] = breg[j];
    }
  };

  int rk = rbeg;
  load_tile(rk);
  while (true) {
    __syncthreads();
    store_tile();
    __syncthreads();
    rk += BK;
    const bool more = rk < rend;
    if (more) load_tile(rk);
    mfma_sweep<WM, WN, TM, TN, LDA, LDB>(As, Bs, acc, wm, wn, lane);
    if (!more) break;
  }

  const int l31 = lane & 31, lh = lane >> 5;
  float* ybase = prm.y + (long long)p * prm.y_ps + (prm.seg_rows ? (long long)blockIdx.z * prm.seg_ys : 0);
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = n0 + (wn * TN + tn) * 32 + l31;
    if (col >= N) continue;
    const float sc = prm.scale ? prm.scale[col] : 1.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int mb = m0 + (wm * TM + tm) * 32 + 4 * lh;
      if (prm.ksplit > 1) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int m = mb + (reg & 3) + 8 * (reg >> 2);
          if (m < M) atomicAdd(ybase + (unsigned)(m * N + col), acc[tm][tn][reg] * sc);
        }
      } else {
        float old[16];                      // all loads first, then the stores (no load behind a store)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int m = mb + (reg & 3) + 8 * (reg >> 2);
          old[reg] = (m < M) ? ybase[(unsigned)(m * N + col)] : 0.f;
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int m = mb + (reg & 3) + 8 * (reg >> 2);
          if (m < M) ybase[(unsigned)(m * N + col)] = old[reg] + acc[tm][tn][reg] * sc;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// weight gradient, specialised variant: C % 4 == 0 (float4 gathers), straight-line K loop.
// Thread-invariant: its kernel tap (kh, kw, ci) and LDS / B offsets.  Per 16-row K-step each float4
// gather costs two magic-number divisions + a branch-free bounds test.
// ------------------------------------------------------------------------------------------
template <int WM, int WN, int TM, int TN, bool PB = false, bool BV4 = PB, bool SPLIT = false>
__global__ __launch_bounds__(WM * WN * 64) void wgrad_fast_kernel(const WgradP prm) {
  using T = Tile<WM, WN, TM, TN>;
  constexpr int NT = T::NT, BM = T::BM, BN = T::BN;
  // A quads (four channels of one row) of a K-tile over the threads: quad q = tid + j NT -> (row q / QPR, m-quad q % QPR).
  // When NT is a multiple of the quads per row every thread keeps ONE m-quad (the original mapping); the four-wave
  // 96-row tile <1,4,3,1> has 384 quads on 256 threads: two quads per thread, the second only for tid < 128 (AGEN)
  constexpr int QTOT = BM * BK / 4, AQ = (QTOT + NT - 1) / NT, AE = 4 * AQ;
  constexpr bool AGEN = (NT % (BM / 4)) != 0 || (QTOT % NT) != 0;
  static_assert(!SPLIT || !AGEN, "split precision keeps the one-m-quad-per-thread mapping");
  static_assert(!PB || BV4, "the probe-batched variant reads the cotangent as float4");
  static_assert(!SPLIT || (BV4 && AQ == 2), "split precision: float4 cotangent rows, two A quads per thread");
  // SPLIT (bf16x3): both operands are k-contiguous bf16 rows ([m][16 k], [n][16 k]; hi and lo planes), but global
  // memory runs along m resp. n at fixed r — each thread therefore owns the row PAIR (2 kp, 2 kp + 1) of its four
  // channels and stores k-adjacent pairs (one ds_write_b32 per channel and plane).
  constexpr int BQ = (BN * BK / 4 + NT - 1) / NT;          // BV4: B float4 per thread per K-tile
  constexpr bool BQPART = (BN * BK / 4) % NT != 0;
  constexpr int BE = SPLIT ? 8 * ((2 * T::BN + T::NT - 1) / T::NT) : (BV4 ? 4 * BQ : T::BE);
  constexpr int LDA = BM + 4, LDB = BN;
  constexpr int QPR = BM / 4;
  constexpr int ASZ = SPLIT ? 2 * BM * SROW : BK * LDA;      // floats per LDS buffer
  constexpr int BSZ = SPLIT ? 2 * BN * SROW : BK * LDB;
  constexpr int BU = SPLIT ? (2 * BN + NT - 1) / NT : 0;     // SPLIT: B units (2 rows x 4 channels) per thread
  __shared__ __attribute__((aligned(16))) float As[2 * ASZ];
  __shared__ __attribute__((aligned(16))) float Bs[2 * BSZ];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // PB (probe-batched): the probes are laid side by side along the GEMM's N axis — column j = (probe j / N, channel
  // j % N) — so one 128-column tile serves 128 / N probes and the gathered activation tile is shared by all of them
  const int N = prm.N, M = prm.M;
  const int NC = PB ? prm.P * N : N;           // GEMM columns of this launch
  const int tiles_n = (NC + BN - 1) / BN;
  // Workgroups go round-robin over the 8 XCDs: remap the (tile, probe) index so that every XCD works through a
  // CONTIGUOUS run of it — the row tiles of one probe (which all stream the same cotangent rows) and the column tiles
  // over one activation slab then share an L2 instead of fetching the operand once per XCD.
  int bx = blockIdx.x, by = PB ? 0 : (int)blockIdx.y;
  {
    const int gx = (int)gridDim.x, tot = gx * (PB ? 1 : (int)gridDim.y), g8 = tot & ~7;
    const int lin = bx + gx * by;
    if (lin < g8) {
      const int w = (lin & 7) * (g8 >> 3) + (lin >> 3);
      by = w / gx; bx = w - by * gx;
    }
  }
  // PB: row tile fastest, so the (three) row tiles of one column tile — same cotangent columns — are neighbours
  const int tiles_m = (int)gridDim.x / tiles_n;
  const int tile_n = PB ? bx / tiles_m : bx % tiles_n, tile_m = PB ? bx - tile_n * tiles_m : bx / tiles_n;
  const int p = by;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  int rows_per = (prm.R + prm.ksplit - 1) / prm.ksplit;
  rows_per = (rows_per + BK - 1) / BK * BK;
  if (prm.seg_rows) rows_per = prm.seg_rows;           // per-example rows: block z reduces example z only
  const int rbeg = blockIdx.z * rows_per;
  const int rend = min(prm.R, rbeg + rows_per);
  if (rbeg >= rend) return;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

  // quad j of this thread: m = m0 + 4 * (q % QPR) -> tap (kh, kw) and channel ci; row q / QPR of the K-step.  Without
  // AGEN all AQ quads share the m-quad (the compiler folds the copies), rows krow0 + j * NT / QPR
  bool mvalid[AQ];
  int thh[AQ], tww[AQ], ci[AQ], krow[AQ];
#pragma unroll
  for (int j = 0; j < AQ; ++j) {
    const int q = tid + j * NT;
    const int kq = q / QPR, mq = q - kq * QPR;
    const int my_m = m0 + 4 * mq;
    mvalid[j] = my_m < M && (!AGEN || q < QTOT);
    thh[j] = 0; tww[j] = 0; ci[j] = 0;
    krow[j] = SPLIT ? 2 * (tid / QPR) + j : kq;
    if (mvalid[j]) {
      const int tap = my_m / prm.C;
      ci[j] = my_m - tap * prm.C;
      const int kh = tap / prm.KW, kw = tap - kh * prm.KW;
      thh[j] = kh - prm.pad_h;
      tww[j] = kw - prm.pad_w;
    }
  }
  const int krow0 = SPLIT ? 2 * (tid / QPR) : tid / QPR;   // SPLIT: first row of this thread's row pair
  constexpr int NB = SPLIT ? BU : (BV4 ? BQ : BE);   // B units (SPLIT: 2 rows x 4 channels) / load instructions per thread
  unsigned bidx[NB];
  int bk[NB];
  bool bok[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int e = tid + j * NT;
    if (SPLIT) {                                  // unit e -> (row pair kp, channel quad nq)
      const int kp = e / (BN / 4), nq = e - kp * (BN / 4);
      bk[j] = 2 * kp;
      bok[j] = (n0 + 4 * nq) < NC && e < 2 * BN;
      const int jc = bok[j] ? n0 + 4 * nq : 0, pj = PB ? jc / N : 0;
      bidx[j] = (unsigned)((long long)pj * prm.g_ps + 2 * kp * N + (jc - pj * N));
    } else if (BV4) {                                    // float4 along the channels of one probe (N % 4 == 0)
      const int k = e / (BN / 4), nq = e - k * (BN / 4);
      bk[j] = k;
      bok[j] = (n0 + 4 * nq) < NC && (!BQPART || e < BN * BK / 4);
      const int jc = bok[j] ? n0 + 4 * nq : 0, pj = PB ? jc / N : 0;
      bidx[j] = (unsigned)((long long)pj * prm.g_ps + k * N + (jc - pj * N));     // < 2^32 floats (host-checked)
    } else {
      const int k = e / BN, nn = e - k * BN;
      bk[j] = k;
      bok[j] = (n0 + nn) < NC && (!T::BPART || e < BN * BK);
      bidx[j] = (unsigned)(k * N + n0 + nn);
    }
  }

  const float* gp = prm.g + (long long)p * prm.g_ps + (long long)rbeg * N;
  const int IH = prm.IH, IW = prm.IW, C = prm.C, stride = prm.stride, OHW = prm.OHW, OW = prm.OW;

  int rk0 = rbeg;
  auto load_tile = [&](float (&areg)[AE], float (&breg)[BE]) {
#pragma unroll
    for (int j = 0; j < AQ; ++j) {
      const int r = rk0 + krow[j];
      const int i = prm.dOHW.div(r), rem = r - i * OHW;
      const int oh = prm.dOW.div(rem), ow = rem - oh * OW;
      const int ih = oh * stride + thh[j], iw = ow * stride + tww[j];
      const bool ok = mvalid[j] && (r < rend) && ((unsigned)ih < (unsigned)IH) && ((unsigned)iw < (unsigned)IW);
      const float* src = ok ? (prm.a + (unsigned)(((i * IH + ih) * IW + iw) * C + ci[j])) : prm.zeros;
      const float4 v = *reinterpret_cast<const float4*>(src);
      areg[4 * j + 0] = v.x; areg[4 * j + 1] = v.y; areg[4 * j + 2] = v.z; areg[4 * j + 3] = v.w;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (SPLIT) {
        const float* s0 = (bok[j] && (rk0 + bk[j]) < rend) ? (gp + bidx[j]) : prm.zeros;
        const float* s1 = (bok[j] && (rk0 + bk[j] + 1) < rend) ? (gp + bidx[j] + N) : prm.zeros;
        const float4 v0 = *reinterpret_cast<const float4*>(s0);
        const float4 v1 = *reinterpret_cast<const float4*>(s1);
        breg[8 * j + 0] = v0.x; breg[8 * j + 1] = v0.y; breg[8 * j + 2] = v0.z; breg[8 * j + 3] = v0.w;
        breg[8 * j + 4] = v1.x; breg[8 * j + 5] = v1.y; breg[8 * j + 6] = v1.z; breg[8 * j + 7] = v1.w;
        continue;
      }
      const float* src = (bok[j] && (rk0 + bk[j]) < rend) ? (gp + bidx[j]) : prm.zeros;
      if (BV4) {
        const float4 v = *reinterpret_cast<const float4*>(src);
        breg[4 * j + 0] = v.x; breg[4 * j + 1] = v.y; breg[4 * j + 2] = v.z; breg[4 * j + 3] = v.w;
      } else {
        breg[j] = *src;
      }
    }
  };
  auto store_tile = [&](const float (&areg)[AE], const float (&breg)[BE], float* Asb, float* Bsb) {
    if (SPLIT) {
      const int mq = tid % QPR;
#pragma unroll
      for (int t = 0; t < 4; ++t) split_store_pair(Asb, Asb + BM * SROW, 4 * mq + t, krow0, areg[t], areg[4 + t]);
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int e = tid + j * NT;
        const int kp = e / (BN / 4), nq = e - kp * (BN / 4);
        if (e < 2 * BN) {
#pragma unroll
          for (int t = 0; t < 4; ++t)
            split_store_pair(Bsb, Bsb + BN * SROW, 4 * nq + t, 2 * kp, breg[8 * j + t], breg[8 * j + 4 + t]);
        }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < AQ; ++j) {
      const int q = tid + j * NT;
      const int k = q / QPR, mq = q - k * QPR;
      if (!AGEN || q < QTOT)
        *reinterpret_cast<float4*>(&Asb[k * LDA + 4 * mq]) =
            make_float4(areg[4 * j + 0], areg[4 * j + 1], areg[4 * j + 2], areg[4 * j + 3]);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int e = tid + j * NT;
      if (BV4) {
        const int k = e / (BN / 4), nq = e - k * (BN / 4);
        if (!BQPART || e < BN * BK / 4)
          *reinterpret_cast<float4*>(&Bsb[k * LDB + 4 * nq]) =
              make_float4(breg[4 * j + 0], breg[4 * j + 1], breg[4 * j + 2], breg[4 * j + 3]);
      } else {
        const int k = e / BN, nn = e - k * BN;
        if (!T::BPART || e < BN * BK) Bsb[k * LDB + nn] = breg[j];
      }
    }
  };

  auto advance = [&]() {
    rk0 += BK;
    gp += BK * N;
  };
  const int ktiles = (rend - rbeg + BK - 1) / BK;
  pipelined_k_loop<AE, BE, ASZ, BSZ>(
      ktiles, As, Bs, load_tile, store_tile, advance,
      [&](const float* Asb, const float* Bsb) {
        if (SPLIT) mfma_sweep_split<WM, WN, TM, TN>(Asb, Bsb, acc, wm, wn, lane);
        else mfma_sweep<WM, WN, TM, TN, LDA, LDB>(Asb, Bsb, acc, wm, wn, lane);
      });

  const int l31 = lane & 31, lh = lane >> 5;
  const long long yseg = prm.seg_rows ? (long long)blockIdx.z * prm.seg_ys : 0;
  float* ybase = prm.y + (long long)p * prm.y_ps + yseg;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    int col = n0 + (wn * TN + tn) * 32 + l31;
    if (col >= NC) continue;
    if (PB) {
      const int pj = col / N;
      col -= pj * N;
      ybase = prm.y + (long long)pj * prm.y_ps + yseg;
    }
    const float sc = prm.scale ? prm.scale[col] : 1.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int mb = m0 + (wm * TM + tm) * 32 + 4 * lh;
      if (prm.ksplit > 1) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int m = mb + (reg & 3) + 8 * (reg >> 2);
          if (m < M) atomicAdd(ybase + (unsigned)(m * N + col), acc[tm][tn][reg] * sc);
        }
      } else {
        float old[16];                      // all loads first, then the stores (no load behind a store)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int m = mb + (reg & 3) + 8 * (reg >> 2);
          old[reg] = (m < M) ? ybase[(unsigned)(m * N + col)] : 0.f;
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int m = mb + (reg & 3) + 8 * (reg >> 2);
          if (m < M) ybase[(unsigned)(m * N + col)] = old[reg] + acc[tm][tn][reg] * sc;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// weight gradient of a DENSE layer with a short reduction (R = n examples <= 64: the layers of an MLP — BASELINE
// configs[2] — and the final Dense of every net):  dW_p (M x N) = s * a^T g_p  is outer-product shaped, 2 R FLOP per
// output element against 4 B written (+ 4 B of alpha V read): at R = 50 it sits on the machine balance, and what the
// tiled kernels above spend per block (LDS staging, barriers, a read-modify-write epilogue behind a 4-tile K loop)
// left it at 1.8 TB/s.  Here nothing is staged: a wave keeps the A operand of its TM x 32 rows of M for ALL R rows in
// registers (a[r][m]: lane (i = m, k = r & 1) of k-step r >> 1 — one coalesced dword per k-step, loaded once per wave,
// the activations are shared by the probes), walks over (probe, 32-column tile) items, reads the B operand
// g_p[r][n] straight into the MFMA operand registers (coalesced 128-B rows), and writes  y = s acc + alpha v  (overwrite)
// or  y += s acc  (accumulate) with 128-B row segments.  No LDS, no barrier, R need not be padded beyond a multiple of 2.
// ------------------------------------------------------------------------------------------
// FULL: the interior — every (m-group, column tile) item is complete, all indices are affine in (lane, element), no
// clamps, unconditional stores, software pipelined.  !FULL: the same items of the partial last m-group / last column
// tile with clamped loads and masked stores (the interior items are skipped there).
template <int TM, int KK, bool FULL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_skinny_kernel(const WgradP prm) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: the item loop below is wave-uniform (as a VGPR value it
                                                                 // made the loop divergent: both store paths under exec masks, 64-bit per-lane addresses)
  const int l31 = lane & 31, lh = lane >> 5;
  const int N = prm.N, M = prm.M, R = prm.R;
  const int m0 = ((int)blockIdx.y + prm.sk_mg0) * (32 * TM);
  // Loads are unconditional on CLAMPED indices where an index can leave its array (a conditional load costs a branch
  // and a full s_waitcnt per element: 64 serial round trips per item in the first version of this kernel).  Row i of
  // the output depends on row i of A only and column j on column j of B only, so rows >= M and columns >= N may hold
  // anything (they are never stored); only the reduction rows r >= R must not contribute: A is zeroed there, B reads a
  // valid row instead.
  float areg[TM][KK];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int m = min(m0 + 32 * tm + l31, M - 1);
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const int r = 2 * kk + lh;
      const float t = prm.a[(unsigned)(min(r, R - 1) * M + m)];
      areg[tm][kk] = r < R ? t : 0.f;
    }
  }
  const int tiles_n = prm.sk_tiles;                             // column tiles per probe this launch walks: [sk_tn0, sk_tn0 + sk_tiles)
  const int total = prm.P * tiles_n;                            // (probe, column tile) items; < 2^31 (host-checked)
  const bool over = prm.overwrite != 0;
  const float sa = over ? prm.alpha : 1.f;
  const unsigned last = (unsigned)(M * N - 1);
  const int stride = (int)gridDim.x * 4;
  // B operand of k-step kk: rows (2 kk, 2 kk + 1) for the two lane halves.  Addressed as a wave-uniform row pointer
  // (scalar registers) plus ONE per-lane offset — 26 per-lane row offsets would cost 26 VGPRs of a budget that has
  // none to spare.  A k-step whose second row is past R reads row R - 1 (its A operand is zero there), one that is
  // past R altogether reads row 0.
  const unsigned lhN = (unsigned)(lh * N);
  auto load_b = [&](int ct, float (&b)[KK]) {
    const int p = ct / tiles_n;
    const unsigned colc = (unsigned)min((prm.sk_tn0 + ct - p * tiles_n) * 32 + l31, N - 1);
    const unsigned pairc = colc + lhN;
    const float* __restrict__ gp = prm.g + (long long)p * prm.g_ps;        // wave-uniform base
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const bool pair = 2 * kk + 1 < R;                                    // uniform
      const float* __restrict__ rowp = gp + (pair ? (long long)(2 * kk) * N : (2 * kk < R ? (long long)(R - 1) * N : 0ll));
      b[kk] = rowp[pair ? pairc : colc];
    }
  };
  unsigned ebase[TM];                                           // element offset of accumulator register 0 (column l31 of tile 0)
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) ebase[tm] = (unsigned)((m0 + 32 * tm + 4 * lh) * N + l31);
  int ct = (int)blockIdx.x * 4 + wave;
  if (ct >= total) return;
  float b[KK], bn[KK];
  load_b(ct, b);
  for (; ct < total; ct += stride) {
    // software pipeline: the next item's B operand and this item's epilogue operand are requested BEFORE the MFMA
    // block, so with two waves per SIMD the matrix pipe never waits for memory
    load_b(min(ct + stride, total - 1), bn);
    const int p = ct / tiles_n;
    {
      const int n0 = (prm.sk_tn0 + ct - p * tiles_n) * 32;
      const float sc = prm.scale ? prm.scale[min(n0 + l31, N - 1)] : 1.f;
      float* __restrict__ yb = prm.y + (long long)p * prm.y_ps;
      // what is added to s * acc: alpha * v (overwrite; v may be absent) or the block's current content (accumulate)
      const float* __restrict__ src = over ? (prm.v ? prm.v + (long long)p * prm.v_ps : nullptr) : yb;
      float add[TM][16];
      if (src) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            if (FULL) {
              add[tm][q] = (src + (long long)((q & 3) + 8 * (q >> 2)) * N)[ebase[tm] + (unsigned)n0];    // uniform row pointer + lane offset
            } else {
              const unsigned e = ebase[tm] + (unsigned)n0 + (unsigned)(((q & 3) + 8 * (q >> 2)) * N);
              add[tm][q] = src[min(e, last)];
            }
          }
      } else {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int q = 0; q < 16; ++q) add[tm][q] = 0.f;
      }
      f32x16 acc[TM];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[tm][q] = 0.f;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) acc[tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[tm][kk], b[kk], acc[tm], 0, 0, 0);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          if (FULL) {
            (yb + (long long)((q & 3) + 8 * (q >> 2)) * N)[ebase[tm] + (unsigned)n0] = acc[tm][q] * sc + sa * add[tm][q];
          } else {
            const unsigned e = ebase[tm] + (unsigned)n0 + (unsigned)(((q & 3) + 8 * (q >> 2)) * N);
            if (n0 + l31 < N && m0 + 32 * tm + 4 * lh + (q & 3) + 8 * (q >> 2) < M) yb[e] = acc[tm][q] * sc + sa * add[tm][q];
          }
        }
    }
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) b[kk] = bn[kk];
  }
}

static int cu_count();

static bool wgrad_skinny_ok(const WgradP& p) {
  static const bool off = getenv("LIP_NOSKINNY") != nullptr || getenv("LIP_GENERIC") != nullptr;       // A/B switch
  return !off && precision_mode() == 0 && p.seg_rows == 0 && p.ksplit <= 1 && p.R <= 64 && p.OHW == 1 && p.pad_h == 0 &&
         p.pad_w == 0 && p.KH == p.IH && p.KW == p.IW && (long long)p.R * p.M < (1ll << 31) && p.M >= 32 && p.N <= (1 << 20);
}

static bool wgrad_wino_ok(const WgradP& p, int P);
static int wgrad_wino_splits(const WgradP& p, int P);
bool wgrad_will_overwrite(const WgradP& p, int P) { return wgrad_skinny_ok(p) || (wgrad_wino_ok(p, P) && wgrad_wino_splits(p, P) == 1); }

template <int TM, int KK>
static hipError_t run_wgrad_skinny(const WgradP& p0, int P, hipStream_t st) {
  WgradP p = p0;
  p.P = P;
  const int mgroups = (p.M + 32 * TM - 1) / (32 * TM);
  // interior (complete m-groups x complete column tiles), then the two strips with masked stores: the partial last
  // m-group over all column tiles, and the partial last column tile of the complete m-groups (disjoint outputs)
  const int mg_full = p.M / (32 * TM), tn_full = p.N / 32, tn_all = (p.N + 31) / 32;
  // ONE round of resident waves per launch (2 per SIMD = 2 blocks per CU): the items are split statically over the
  // waves, so a grid of 1.1 rounds takes two (measured: 2304 waves on 2048 slots, 0.195 ms; the 16-m-group layer at
  // exactly 2048: 0.085 -> 0.070 with this rule)
  auto grid_x = [&](long long its, int mg) {
    long long g = 2ll * cu_count() / (mg > 0 ? mg : 1);
    g = g >= 8 ? g / 8 * 8 : g;                                // the m-groups of one column range share an XCD (their B rows an L2)
    if (g > (its + 3) / 4) g = (its + 3) / 4;
    return (unsigned)(g < 1 ? 1 : g);
  };
  if (mg_full > 0 && tn_full > 0) {
    p.sk_mg0 = 0; p.sk_tiles = tn_full; p.sk_tn0 = 0;
    hipLaunchKernelGGL((wgrad_skinny_kernel<TM, KK, true>), dim3(grid_x((long long)P * tn_full, mg_full), (unsigned)mg_full), dim3(256), 0, st, p);
  }
  if (mg_full < mgroups) {
    p.sk_mg0 = mg_full; p.sk_tiles = tn_all; p.sk_tn0 = 0;
    hipLaunchKernelGGL((wgrad_skinny_kernel<TM, KK, false>), dim3(grid_x((long long)P * tn_all, 1), 1u), dim3(256), 0, st, p);
  }
  if (tn_full < tn_all && mg_full > 0) {
    p.sk_mg0 = 0; p.sk_tiles = 1; p.sk_tn0 = tn_full;
    hipLaunchKernelGGL((wgrad_skinny_kernel<TM, KK, false>), dim3(grid_x((long long)P, mg_full), (unsigned)mg_full), dim3(256), 0, st, p);
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// weight gradient of the first layer (round 3): M = KH KW C <= 32 rows (27 for the CIFAR nets), 32 output columns, the
// reduction over ALL R = n OH OW rows.  The generic kernel ran it at 18 TFLOP/s (1.25 ms per 256-probe block).  A wave
// owns one probe and one share of the rows and sweeps it in chunks of 2 KC rows: the A operand is the TRANSPOSED
// im2col block — lane (i = tap m, k = row parity) gathers a[pixel(row), tap m] — the B operand the cotangent rows
// g_p[row][n] (coalesced 128-byte rows straight into the MFMA registers); KC MFMAs per chunk into one 32 x 32
// accumulator, the next chunk's operands requested before the sweep; at the end the 27 x 32 valid entries are added
// to Y with float atomics (one per row share).  No LDS, no barrier; bound by the read of g (R N floats per probe).
// ------------------------------------------------------------------------------------------
template <int KC>
__global__ __launch_bounds__(256) void wgrad_first_kernel(const WgradP prm, int rsplit) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int M = prm.M, N = prm.N, R = prm.R;
  const int item = (int)blockIdx.x * 4 + wave;                 // (probe, row share)
  const int p = item / rsplit, share = item - p * rsplit;
  if (p >= prm.P) return;
  int rows_per = (R + rsplit - 1) / rsplit;
  rows_per = (rows_per + 2 * KC - 1) / (2 * KC) * (2 * KC);
  const int rbeg = share * rows_per, rend = min(R, rbeg + rows_per);
  if (rbeg >= rend) return;
  // this lane's tap (row m = l31 of the output): (kh, kw, c); taps >= M contribute zeros
  const bool mv = l31 < M;
  const int mm = mv ? l31 : 0;
  const int tap = mm / prm.C, c = mm - tap * prm.C, kh = tap / prm.KW, kw = tap - kh * prm.KW;
  const int th = kh - prm.pad_h, tw = kw - prm.pad_w;
  const float* __restrict__ gp = prm.g + (long long)p * prm.g_ps;            // wave-uniform
  const unsigned col = (unsigned)min(l31, N - 1);
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  auto load = [&](int r0, float (&a)[KC], float (&b)[KC]) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) {
      const int r = r0 + 2 * kk + lh;
      const int rc = min(r, R - 1);
      const int i = prm.dOHW.div(rc), rem = rc - i * prm.OHW;
      const int oh = prm.dOW.div(rem), ow = rem - oh * prm.OW;
      const int ih = oh * prm.stride + th, iw = ow * prm.stride + tw;
      const bool ok = mv && r < rend && (unsigned)ih < (unsigned)prm.IH && (unsigned)iw < (unsigned)prm.IW;
      const float t = prm.a[ok ? (unsigned)(((i * prm.IH + ih) * prm.IW + iw) * prm.C + c) : 0u];
      a[kk] = ok ? t : 0.f;                                                  // rows past the share are zero in A: B may read any valid row
      b[kk] = gp[(unsigned)rc * (unsigned)N + col];
    }
  };
  float a0[KC], b0[KC], a1[KC], b1[KC];
  load(rbeg, a0, b0);
  for (int r0 = rbeg; r0 < rend; r0 += 4 * KC) {
    if (r0 + 2 * KC < rend) load(r0 + 2 * KC, a1, b1);
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[kk], b0[kk], acc, 0, 0, 0);
    if (r0 + 2 * KC >= rend) break;
    if (r0 + 4 * KC < rend) load(r0 + 4 * KC, a0, b0);
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[kk], b1[kk], acc, 0, 0, 0);
  }
  const float sc = prm.scale ? prm.scale[col] : 1.f;
  float* __restrict__ yb = prm.y + (long long)p * prm.y_ps;
  if (l31 < N) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = 4 * lh + (q & 3) + 8 * (q >> 2);
      if (m < M) atomicAdd(yb + (unsigned)(m * N + l31), acc[q] * sc);
    }
  }
}

static bool wgrad_first_ok(const WgradP& p, int P) {
  static const bool off = getenv("LIP_NOFIRST") != nullptr || getenv("LIP_GENERIC") != nullptr;        // A/B switch
  return !off && precision_mode() == 0 && p.seg_rows == 0 && p.ksplit <= 0 && p.M <= 32 && p.N <= 32 && (p.C & 3) != 0 &&
         p.R >= 2048 && P >= 8 && (long long)p.R * p.N < (1ll << 31);
}

static hipError_t run_wgrad_first(const WgradP& p0, int P, hipStream_t st) {
  WgradP p = p0;
  p.P = P;
  constexpr int KC = 32;
  int rsplit = (2 * 4 * cu_count() + P - 1) / P;                              // ~2 waves per SIMD
  const int maxsplit = p.R / (8 * 2 * KC) > 0 ? p.R / (8 * 2 * KC) : 1;       // >= 8 chunks per share
  if (rsplit > maxsplit) rsplit = maxsplit;
  if (rsplit < 1) rsplit = 1;
  const long long items = (long long)P * rsplit;
  hipLaunchKernelGGL((wgrad_first_kernel<KC>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, p, rsplit);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Winograd F(2x2, 3x3) WEIGHT GRADIENT (round 3) of the 3x3 / stride 1 / pad 1 layers:
//   dW = G^T [ sum_tiles (B^T d B) (.) (A dY A^T) ] G  — 16 positions xi, each a GEMM with the reduction over TILES,
//   dU_xi[c][n] = sum_t V_xi[t][c] Gh_xi[t][n]: 4 multiplications per output pixel and (c, n) instead of 9.
//   * V = B^T a B of the PRIMAL activations is shared by all probes: wino_input_transform_kernel writes it once per
//     launch as Vt[xi][t/4][c][4], so lane (c, half h) reads four consecutive tiles of its channel as ONE dwordx4 (512
//     contiguous bytes per half-wave) — the A registers of 4 k-steps (k-step j multiplies tile 8m + 4h + j);
//   * Gh = A g A^T of the probe's cotangent is formed on the fly: lane (n, h) loads the 2 x 2 pixels of its tile (dwords,
//     128 contiguous bytes per half-wave), 2 + 3 VALU operations -> the B registers of its wave's four positions;
//   * wave a owns row a of the 4x4 position grid (64 accumulator registers); dW = G^T dU G in-wave over b, across the
//     waves over a through LDS; the fused output y = s acc + alpha v when one block reduces all tiles, float atomics
//     into the initialised block when the tiles are split over blocks (few (c, n) tiles per probe: the 32-channel stage).
// scripts/micro/wino_wgrad_probe.hip: 1.15 - 1.21 ms on the three CIFAR stages at 256 probes (direct kernels: 2.03 - 2.13).
// ------------------------------------------------------------------------------------------
struct WgWinoP {
  const float* vt; unsigned vt_bytes;
  const float* g; long long g_ps; unsigned g_bytes;
  float* y; long long y_ps;
  const float* scale; const float* v; long long v_ps; float alpha; int overwrite;
  int H, W, C, N, TW, T, TQ, S, gps, tpi;
  FastDiv dTPI, dTW;
};

// thread (tq, c): the 4x4 patches of tiles 4 tq .. 4 tq + 3, channel c -> 16 float4 (zeros past the last tile)
__global__ __launch_bounds__(256) void wino_input_transform_kernel(const float* __restrict__ x, float* __restrict__ vt, int H, int W, int C,
                                                                    int TH, int TW, int T, int TQ) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)TQ * C) return;
  const int c = (int)(e % C), tq = (int)(e / C);
  float v[16][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int t = 4 * tq + j;
    float d[4][4];
    const int img = t / (TH * TW), rem = t - img * TH * TW, ty = rem / TW, tx = rem - ty * TW;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ih = 2 * ty - 1 + r, iw = 2 * tx - 1 + q;
        d[r][q] = (t < T && ih >= 0 && ih < H && iw >= 0 && iw < W) ? x[((long long)(img * H + ih) * W + iw) * C + c] : 0.f;
      }
    float e4[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      e4[0][q] = d[0][q] - d[2][q]; e4[1][q] = d[1][q] + d[2][q]; e4[2][q] = d[2][q] - d[1][q]; e4[3][q] = d[1][q] - d[3][q];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      v[4 * a + 0][j] = e4[a][0] - e4[a][2]; v[4 * a + 1][j] = e4[a][1] + e4[a][2];
      v[4 * a + 2][j] = e4[a][2] - e4[a][1]; v[4 * a + 3][j] = e4[a][1] - e4[a][3];
    }
  }
#pragma unroll
  for (int xi = 0; xi < 16; ++xi)
    *reinterpret_cast<f32x4v*>(&vt[(((long long)xi * TQ + tq) * C + c) * 4]) = f32x4v{v[xi][0], v[xi][1], v[xi][2], v[xi][3]};
}

template <bool ROWQ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void wgrad_wino_kernel(const WgWinoP prm) {
  extern __shared__ __attribute__((aligned(16))) float wgw_lds[];          // [4 a][3 kw][16 reg][64 lane]
  const int tid = threadIdx.x, lane = tid & 63;
  const int a = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int C = prm.C, N = prm.N, W = prm.W, H = prm.H;
  const int ctn = C >> 5, ntn = N >> 5;
  int bid = blockIdx.x, byp = blockIdx.y;
  {   // XCD-contiguous order: the blocks of one probe (they stream the same cotangent) share an L2
    const int gx = (int)gridDim.x;
    if (gx >= 64) {
      const int g8 = gx & ~7;
      if (bid < g8) bid = (bid & 7) * (g8 >> 3) + (bid >> 3);
    } else {
      const int g8 = (gx * (int)gridDim.y) & ~7, lin = bid + gx * byp;
      if (lin < g8) {
        const int w = (lin & 7) * (g8 >> 3) + (lin >> 3);
        byp = w / gx; bid = w - byp * gx;
      }
    }
  }
  const int nt = bid % ntn; bid /= ntn;
  const int ct = bid % ctn;
  const int z = bid / ctn;                          // share of the tiles
  const int p = byp;
  const int c = ct * 32 + l31, n = nt * 32 + l31;
  const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(prm.vt), 0, prm.vt_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(prm.g + (long long)p * prm.g_ps), 0, prm.g_bytes, 0x00020000);
  // rows of the 2x2 cotangent tile and their coefficients:  x_q = k0 g[0][q] + k1 g[1][q]   (A rows: g0, g0+g1, g0-g1, -g1)
  const float k0 = (a == 3) ? 0.f : 1.f;
  const float k1 = (a == 0) ? 0.f : (a == 1 ? 1.f : -1.f);
  const unsigned avoff = (unsigned)((h * C + c) * 16);
  const unsigned a_xi = (unsigned)prm.TQ * C * 16;            // bytes per position plane
  const unsigned row_bytes = (unsigned)(W * N * 4);

  f32x16 acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  const int m0 = z * prm.gps, m1 = m0 + prm.gps;
  f32x4v areg[2][4];
  float graw[2][4][4];                 // [buffer][tile j][r * 2 + q]
  auto load_group = [&](int m, f32x4v (&ar)[4], float (&gr)[4][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      ar[q] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(vrs, avoff, (4 * a + q) * a_xi + (unsigned)(2 * m) * C * 16, 0));
    const int t = 8 * m + 4 * h;
    if (ROWQ) {                                       // TW % 4 == 0: the lane's four tiles share a tile row
      const int img = prm.dTPI.div(t), rem = t - img * prm.tpi, ty = prm.dTW.div(rem), tx = rem - ty * prm.TW;
      const unsigned gv = t < prm.T ? (unsigned)((((img * H + 2 * ty) * W + 2 * tx) * N + n) * 4) : 0x80000000u;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int q = 0; q < 2; ++q)
            gr[j][2 * r + q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grs, gv + (unsigned)((2 * j + q) * N * 4), r * row_bytes, 0));
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int tj = t + j;
        const int img = prm.dTPI.div(tj), rem = tj - img * prm.tpi, ty = prm.dTW.div(rem), tx = rem - ty * prm.TW;
        const unsigned gv = tj < prm.T ? (unsigned)((((img * H + 2 * ty) * W + 2 * tx) * N + n) * 4) : 0x80000000u;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int q = 0; q < 2; ++q)
            gr[j][2 * r + q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grs, gv + (unsigned)(q * N * 4), r * row_bytes, 0));
      }
    }
  };
  auto compute = [&](const f32x4v (&ar)[4], const float (&gr)[4][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x0 = k0 * gr[j][0] + k1 * gr[j][2], x1 = k0 * gr[j][1] + k1 * gr[j][3];
      const float bv[4] = {x0, x0 + x1, x0 - x1, -x1};
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[q][j], bv[q], acc[q], 0, 0, 0);
    }
  };
  // (gps is even: two groups per iteration, the last iteration re-requests the final group instead of branching)
  load_group(m0, areg[0], graw[0]);
  for (int m = m0; m < m1; m += 2) {
    load_group(m + 1, areg[1], graw[1]);
    __builtin_amdgcn_sched_barrier(0);
    compute(areg[0], graw[0]);
    __builtin_amdgcn_sched_barrier(0);
    load_group(m + 2 < m1 ? m + 2 : m1 - 1, areg[0], graw[0]);
    __builtin_amdgcn_sched_barrier(0);
    compute(areg[1], graw[1]);
  }

  // dW = G^T dU G: over b in the wave (X[kw]), over a across the waves
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float u0 = acc[0][r], u1 = acc[1][r], u2 = acc[2][r], u3 = acc[3][r];
    wgw_lds[((a * 3 + 0) * 16 + r) * 64 + lane] = u0 + 0.5f * (u1 + u2);
    wgw_lds[((a * 3 + 1) * 16 + r) * 64 + lane] = 0.5f * (u1 - u2);
    wgw_lds[((a * 3 + 2) * 16 + r) * 64 + lane] = 0.5f * (u1 + u2) + u3;
  }
  __syncthreads();
  float* yp = prm.y + (long long)p * prm.y_ps;
  const float* vp = prm.v ? prm.v + (long long)p * prm.v_ps : nullptr;
  const float sc = prm.scale ? prm.scale[n] : 1.f;
  for (int o = a; o < 9; o += 4) {
    const int kh = o / 3, kw = o - 3 * kh;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float x0 = wgw_lds[((0 * 3 + kw) * 16 + r) * 64 + lane], x1 = wgw_lds[((1 * 3 + kw) * 16 + r) * 64 + lane];
      const float x2 = wgw_lds[((2 * 3 + kw) * 16 + r) * 64 + lane], x3 = wgw_lds[((3 * 3 + kw) * 16 + r) * 64 + lane];
      const float w = sc * (kh == 0 ? (x0 + 0.5f * (x1 + x2)) : (kh == 1 ? 0.5f * (x1 - x2) : (0.5f * (x1 + x2) + x3)));
      const int ci = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const long long idx = ((long long)(o * C + ci)) * N + n;
      if (prm.S > 1) atomicAdd(yp + idx, w);
      else if (prm.overwrite) yp[idx] = w + (vp ? prm.alpha * vp[idx] : 0.f);
      else yp[idx] += w;
    }
  }
}

static int cu_count();
static float* ksplit_scratch(size_t floats, hipStream_t st);
int wino_mode();

// shares of the tiles per (probe, c tile, n tile): 1 when the launch fills the chip without splitting
static int wgrad_wino_splits(const WgradP& p, int P) {
  const long long blocks = (long long)(p.C / 32) * (p.N / 32) * P;
  const long long want = 4ll * cu_count();
  const int groups = (p.R / 4 + 7) / 8;
  long long S = blocks >= want ? 1 : (want + blocks - 1) / blocks;
  if (S > groups / 16) S = groups / 16;
  if (S > 64) S = 64;
  return (int)(S < 1 ? 1 : S);
}

static bool wgrad_wino_ok(const WgradP& p, int P) {
  if (wino_mode() == 0 || precision_mode() != 0 || p.seg_rows > 0) return false;
  if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad_h != 1 || p.pad_w != 1) return false;
  if ((p.C & 31) != 0 || (p.N & 31) != 0 || p.M != 9 * p.C) return false;
  const int OH = p.OHW / p.OW;
  if (OH != p.IH || p.OW != p.IW || (OH & 1) || (p.OW & 1) != 0) return false;
  if ((long long)p.R * p.N * 4 >= (1ll << 31)) return false;
  {   // ONE (probe, c tile, n tile) block (a single product on the 32-channel stage): its 12 800 tiles would be split 64 ways
    // and added with 64-fold contended float atomics — the direct kernel's own row split is faster there (47 vs 31 us;
    // single product 1.82 -> 1.77 ms).  Two blocks already favour this kernel (2.17 vs 2.60 ms for two products).
    static const int minb = getenv("LIP_WGW_MINBLOCKS") ? atoi(getenv("LIP_WGW_MINBLOCKS")) : 2;     // A/B switch
    if ((long long)(p.C / 32) * (p.N / 32) * P < minb) return false;
  }
  const long long T = p.R / 4;
  if (16ll * (T + 16ll * 64) * p.C * 4 >= (1ll << 31)) return false;
  return true;
}

static hipError_t run_wgrad_wino(const WgradP& p, int P, hipStream_t st) {
  WgWinoP q;
  const int OH = p.OHW / p.OW, TH = OH / 2, TW = p.OW / 2;
  const int T = p.R / 4, groups = (T + 7) / 8;
  const int S = wgrad_wino_splits(p, P);
  int gps = (groups + S - 1) / S; gps += gps & 1;
  q.S = S; q.gps = gps; q.TQ = 2 * gps * S; q.T = T; q.TW = TW; q.tpi = TH * TW;
  q.dTPI = FastDiv((unsigned)q.tpi); q.dTW = FastDiv((unsigned)TW);
  q.H = OH; q.W = p.OW; q.C = p.C; q.N = p.N;
  const size_t vfloats = (size_t)16 * q.TQ * p.C * 4;
  float* vt = ksplit_scratch(vfloats, st);
  if (!vt) return hipErrorOutOfMemory;
  q.vt = vt; q.vt_bytes = (unsigned)(vfloats * 4);
  q.g = p.g; q.g_ps = p.g_ps; q.g_bytes = (unsigned)((long long)p.R * p.N * 4);
  q.y = p.y; q.y_ps = p.y_ps; q.scale = p.scale;
  q.overwrite = (p.overwrite && S == 1) ? 1 : 0; q.v = p.v; q.v_ps = p.v_ps; q.alpha = p.alpha;
  if (p.overwrite && S != 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(wino_input_transform_kernel, dim3((unsigned)(((long long)q.TQ * p.C + 255) / 256)), dim3(256), 0, st, p.a, vt, OH, p.OW, p.C, TH, TW, T, q.TQ);
  const size_t shmem = 12 * 1024 * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)wgrad_wino_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)wgrad_wino_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const dim3 grid((unsigned)((p.C / 32) * (p.N / 32) * S), (unsigned)P);
  if ((TW & 3) == 0) hipLaunchKernelGGL(wgrad_wino_kernel<true>, grid, dim3(256), shmem, st, q);
  else hipLaunchKernelGGL(wgrad_wino_kernel<false>, grid, dim3(256), shmem, st, q);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// C (m, n) += A B^T for A (m, K), B (n, K) float32 with K-contiguous rows and K >> m, n (K = D ~ 1e6): the tall-skinny
// products of the posterior engine on a materialised factor — W^T applied to a block of draws (src/sample.py:130-139),
// the first GEMM of the factor-mode GGN-vp.  Both operands run along K in memory, i.e. both need the transposing
// (k-major) LDS store the implicit-GEMM A operand uses; the reduction axis is split over the grid (each block owns a
// 128 x 128 tile x one K-range, f32 MFMA, pipelined K loop) and the partial tiles meet through float atomics in C
// (zeroed by the launcher).  hipBLASLt reaches 39 TFLOP/s on (256 x D)(450 x D)^T; this kernel is launched instead.
// ------------------------------------------------------------------------------------------
struct GemmNtP {
  const float* a; long long lda; int m;
  const float* b; long long ldb; int n;
  long long K, kper;
  float* c;
};

__global__ __launch_bounds__(256) void gemm_nt_kernel(const GemmNtP prm) {
  using T = Tile<2, 2, 2, 2>;
  constexpr int NT = T::NT, BM = T::BM, BN = T::BN, AE = T::AE, AQ = T::AQ;      // 256 threads, 128 x 128, 2 float4 per operand
  constexpr int LDA = BM + 2, LDB = BN + 2;
  constexpr int ASZ = BK * LDA, BSZ = BK * LDB;
  __shared__ __attribute__((aligned(16))) float As[2 * ASZ];
  __shared__ __attribute__((aligned(16))) float Bs[2 * BSZ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (prm.n + BN - 1) / BN;
  const int m0 = ((int)blockIdx.x / tiles_n) * BM, n0 = ((int)blockIdx.x % tiles_n) * BN;
  const long long kb = (long long)blockIdx.y * prm.kper, ke = (kb + prm.kper < prm.K) ? kb + prm.kper : prm.K;
  const int ktiles = (int)((ke - kb + BK - 1) / BK);
  if (ktiles <= 0) return;

  f32x16 acc[2][2];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

  // quad q = tid + j * NT -> (row = q >> 2, k-quad kq4 = 4 * (tid & 3)), j = 0, 1
  const int kq4 = (tid & 3) * 4;
  const float* arow[AQ];
  const float* brow[AQ];
  bool aok[AQ], bok[AQ];
#pragma unroll
  for (int j = 0; j < AQ; ++j) {
    const int row = (tid + j * NT) >> 2;
    aok[j] = m0 + row < prm.m; bok[j] = n0 + row < prm.n;
    arow[j] = prm.a + (long long)(aok[j] ? m0 + row : 0) * prm.lda + kq4;
    brow[j] = prm.b + (long long)(bok[j] ? n0 + row : 0) * prm.ldb + kq4;
  }
  long long k0 = kb;
  auto fetch = [&](const float* src, bool ok, float* dst) {
    const long long k = k0 + kq4;
    dst[0] = dst[1] = dst[2] = dst[3] = 0.f;
    if (ok && k < ke) {
      if (k + 3 < ke) { const float4u v = *reinterpret_cast<const float4u*>(src + k0); dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3]; }
      else { dst[0] = src[k0]; if (k + 1 < ke) dst[1] = src[k0 + 1]; if (k + 2 < ke) dst[2] = src[k0 + 2]; }
    }
  };
  auto load_tile = [&](float (&areg)[AE], float (&breg)[AE]) {
#pragma unroll
    for (int j = 0; j < AQ; ++j) { fetch(arow[j], aok[j], &areg[4 * j]); fetch(brow[j], bok[j], &breg[4 * j]); }
  };
  auto store_tile = [&](const float (&areg)[AE], const float (&breg)[AE], float* Asb, float* Bsb) {
#pragma unroll
    for (int j = 0; j < AQ; ++j) {
      const int row = (tid + j * NT) >> 2;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        Asb[(kq4 + t) * LDA + row] = areg[4 * j + t];
        Bsb[(kq4 + t) * LDB + row] = breg[4 * j + t];
      }
    }
  };
  auto advance = [&]() { k0 += BK; };
  pipelined_k_loop<AE, AE, ASZ, BSZ>(ktiles, As, Bs, load_tile, store_tile, advance,
                                     [&](const float* Asb, const float* Bsb) { mfma_sweep<2, 2, 2, 2, LDA, LDB>(Asb, Bsb, acc, wm, wn, lane); });

  const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const int col = n0 + (wn * 2 + tn) * 32 + l31;
    if (col >= prm.n) continue;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const int rb = m0 + (wm * 2 + tm) * 32 + 4 * lh;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int r = rb + (reg & 3) + 8 * (reg >> 2);
        if (r < prm.m) atomicAdd(prm.c + (long long)r * prm.n + col, acc[tm][tn][reg]);
      }
    }
  }
}

// (Round 3, measured and removed: the same product with BOTH operands straight from global memory into the MFMA operand
// registers — lane-half h holding the contiguous k's 8h .. 8h+7 of its row of A and of B, no LDS, no barrier, 2 x 2 waves
// of 64 x 64 per block, register tiles three deep: 3.90 ms = 64 TFLOP/s on (256 x D)(450 x D)^T against 2.98 ms = 84 of
// the LDS-staged kernel above.  Every lane's 16-byte load is a request of its own at the L1 — rows are D floats apart —
// and with two direct operands a CU issues one request per cycle: the tag pipeline, not the matrix pipe, sets the pace.
// The 16 x 16 x 4 shape, where lane (i, q) loads the four k's 4q .. 4q+3 of its row — 16 rows x 64 contiguous bytes, a
// quarter of the requests per instruction — with 64 x 64 per wave: 3.57 ms = 70 TFLOP/s, 252 registers.  Also removed.)
hipError_t launch_gemm_nt(const float* A, long long lda, int m, const float* B, long long ldb, int n, long long K, float* C,
                          hipStream_t st) {
  GemmNtP p;
  p.a = A; p.lda = lda; p.m = m; p.b = B; p.ldb = ldb; p.n = n; p.K = K; p.c = C;
  const long long tiles = (long long)((m + 127) / 128) * ((n + 127) / 128);
  long long ktl = (K + BK - 1) / BK, ks = (1024 + tiles - 1) / tiles;          // ~2 blocks per CU and slot
  if (ks > ktl / 64) ks = ktl / 64;                                             // >= 64 K-tiles per block
  if (ks < 1) ks = 1;
  if (ks > 65535) ks = 65535;
  p.kper = (ktl + ks - 1) / ks * BK;
  ks = (K + p.kper - 1) / p.kper;
  hipError_t e = hipMemsetAsync(C, 0, sizeof(float) * (size_t)m * n, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(gemm_nt_kernel, dim3((unsigned)tiles, (unsigned)ks), dim3(256), 0, st, p);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Out (m, N) = T (m, k) B (k, N) + beta V (m, N) for a SHORT reduction k (the d <= ~1000 rows of a materialised factor) and
// a very long N (= D ~ 1e6): the second pass of every posterior draw,  x = alpha^(-1/2) eps + (coefficients) Qm
// (src/sample.py:139-143 collapsed, see sample.py), and the second product of the factor-mode GGN-vp.  T is tiny and
// K-contiguous (transposing k-major LDS store, as the implicit-GEMM A operand), B rows run along N (k-major as stored:
// float4 rows), 128 x 128 tile, f32 MFMA, the same pipelined K loop; the addend rides in the epilogue and may alias
// Out (every element is read before it is written, by the thread that writes it).  Replaces torch.addmm (hipBLASLt).
// ------------------------------------------------------------------------------------------
struct GemmNnP {
  const float* t; long long ldt; int m, k;
  const float* b; long long ldb; long long N;
  const float* v; long long ldv; float beta;
  float* out; long long ldo;
};

__global__ __launch_bounds__(256) void gemm_nn_axpy_kernel(const GemmNnP prm) {
  using T = Tile<2, 2, 2, 2>;
  constexpr int NT = T::NT, BM = T::BM, BN = T::BN, AE = T::AE, AQ = T::AQ;      // 256 threads, 128 x 128
  constexpr int BE = BN * BK / NT, BQ = BE / 4;                                   // 8 floats = 2 float4 of B per thread and K-tile
  constexpr int LDA = BM + 2, LDB = BN;
  constexpr int ASZ = BK * LDA, BSZ = BK * LDB;
  __shared__ __attribute__((aligned(16))) float As[2 * ASZ];
  __shared__ __attribute__((aligned(16))) float Bs[2 * BSZ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // block -> (column tile, row tile), row tile fastest, and every XCD a contiguous run of blocks: the row tiles of a
  // column tile read the same 128-column slab of B and should find it in one L2
  const int tiles_m = (prm.m + BM - 1) / BM;
  long long bid = blockIdx.x;
  {
    const long long g8 = (long long)gridDim.x & ~7ll;
    if (bid < g8) bid = (bid & 7) * (g8 >> 3) + (bid >> 3);
  }
  const int m0 = (int)(bid % tiles_m) * BM;
  const long long n0 = (bid / tiles_m) * BN;
  const int ktiles = (prm.k + BK - 1) / BK;
  const bool cols_full = n0 + BN <= prm.N;                                       // uniform over the block

  f32x16 acc[2][2];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

  // All loads are unconditional on clamped addresses (a conditional load costs a branch and a full wait per element:
  // the first version of this kernel, 3.13 ms against 2.97 now and hipBLASLt's 2.84 on (256 x 450)(450 x 1.08 M)).
  // A: quad q = tid + j * NT -> (row = q >> 2, k-quad kq4 = 4 * (tid & 3)); rows >= m read row 0 (their outputs are
  // never stored), the quads of the last K-tile that reach past k read the last full quad and are shifted / zeroed by
  // selects — the reduction rows past k must not contribute, so B may then read any valid row.
  const int kq4 = (tid & 3) * 4;
  const float* arow[AQ];
#pragma unroll
  for (int j = 0; j < AQ; ++j) {
    const int row = m0 + ((tid + j * NT) >> 2);
    arow[j] = prm.t + (long long)(row < prm.m ? row : 0) * prm.ldt;
  }
  const int klast4 = prm.k - 4;                                                  // k >= 4 (host-checked)
  // B: float4 e = tid + j * NT -> (k = e / 32, column quad nq = e % 32)
  int bk[BQ];
  long long bcol[BQ];
#pragma unroll
  for (int j = 0; j < BQ; ++j) {
    const int e = tid + j * NT;
    bk[j] = e / (BN / 4);
    bcol[j] = n0 + 4 * (e % (BN / 4));
  }
  int k0 = 0;
  auto load_tile = [&](float (&areg)[AE], float (&breg)[BE]) {
    const int ka = k0 + kq4;
    const bool tail = k0 + BK > prm.k;                                           // uniform: only the last K-tile
#pragma unroll
    for (int j = 0; j < AQ; ++j) {
      const float4u v = *reinterpret_cast<const float4u*>(arow[j] + (tail ? min(ka, klast4) : ka));
      if (!tail) {
        areg[4 * j + 0] = v[0]; areg[4 * j + 1] = v[1]; areg[4 * j + 2] = v[2]; areg[4 * j + 3] = v[3];
      } else {
        const int sh = ka - min(ka, klast4);                                     // the quad was read sh elements early
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float x = (t + sh == 0) ? v[0] : (t + sh == 1) ? v[1] : (t + sh == 2) ? v[2] : (t + sh == 3) ? v[3] : 0.f;
          areg[4 * j + t] = (ka + t < prm.k) ? x : 0.f;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < BQ; ++j) {
      const float* brow = prm.b + (long long)min(k0 + bk[j], prm.k - 1) * prm.ldb;
      if (cols_full) {
        const float4u v = *reinterpret_cast<const float4u*>(brow + bcol[j]);
        breg[4 * j + 0] = v[0]; breg[4 * j + 1] = v[1]; breg[4 * j + 2] = v[2]; breg[4 * j + 3] = v[3];
      } else {                                            // the last column tile: element loads, columns clamped (never stored)
#pragma unroll
        for (int t = 0; t < 4; ++t) breg[4 * j + t] = brow[min(bcol[j] + t, prm.N - 1)];
      }
    }
  };
  auto store_tile = [&](const float (&areg)[AE], const float (&breg)[BE], float* Asb, float* Bsb) {
#pragma unroll
    for (int j = 0; j < AQ; ++j) {
      const int row = (tid + j * NT) >> 2;
#pragma unroll
      for (int t = 0; t < 4; ++t) Asb[(kq4 + t) * LDA + row] = areg[4 * j + t];
    }
#pragma unroll
    for (int j = 0; j < BQ; ++j) {
      const int e = tid + j * NT;
      *reinterpret_cast<float4*>(&Bsb[(e / (BN / 4)) * LDB + 4 * (e % (BN / 4))]) =
          make_float4(breg[4 * j + 0], breg[4 * j + 1], breg[4 * j + 2], breg[4 * j + 3]);
    }
  };
  auto advance = [&]() { k0 += BK; };
  pipelined_k_loop<AE, BE, ASZ, BSZ>(ktiles, As, Bs, load_tile, store_tile, advance,
                                     [&](const float* Asb, const float* Bsb) { mfma_sweep<2, 2, 2, 2, LDA, LDB>(Asb, Bsb, acc, wm, wn, lane); });

  const int l31 = lane & 31, lh = lane >> 5;
  const float beta = prm.v ? prm.beta : 0.f;
  const bool full = cols_full && m0 + BM <= prm.m;                               // uniform
  // (requesting the addend of the whole 64 x 64 wave tile in one round trip — 64 registers — was measured: 4.38 ms)
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const long long col = n0 + (wn * 2 + tn) * 32 + l31;
    const bool cv = col < prm.N;
    const long long colc = cv ? col : prm.N - 1;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const int rb = m0 + (wm * 2 + tm) * 32 + 4 * lh;
      float add[16];
      if (prm.v) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {                     // all loads of the phase first, then its stores
          const int r = min(rb + (q & 3) + 8 * (q >> 2), prm.m - 1);
          add[q] = prm.v[(long long)r * prm.ldv + colc];
        }
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) add[q] = 0.f;
      }
      if (full) {
#pragma unroll
        for (int q = 0; q < 16; ++q) prm.out[(long long)(rb + (q & 3) + 8 * (q >> 2)) * prm.ldo + col] = acc[tm][tn][q] + beta * add[q];
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int r = rb + (q & 3) + 8 * (q >> 2);
          if (cv && r < prm.m) prm.out[(long long)r * prm.ldo + col] = acc[tm][tn][q] + beta * add[q];
        }
      }
    }
  }
}

hipError_t launch_gemm_nn_axpy(const float* T, long long ldt, int m, int k, const float* B, long long ldb, long long N,
                               const float* V, long long ldv, float beta, float* Out, long long ldo, hipStream_t st) {
  GemmNnP p;
  p.t = T; p.ldt = ldt; p.m = m; p.k = k; p.b = B; p.ldb = ldb; p.N = N; p.v = V; p.ldv = ldv; p.beta = beta; p.out = Out; p.ldo = ldo;
  const long long blocks = (long long)((m + 127) / 128) * ((N + 127) / 128);
  hipLaunchKernelGGL(gemm_nn_axpy_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// launchers: pick the tile shape from the problem shape
// ------------------------------------------------------------------------------------------
// 256 bytes of device zeros (per device): the source of masked gather rows in the fast kernels.
static const float* zero_page() {
  static float* pages[64] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!pages[dev]) {
    float* ptr = nullptr;
    if (hipMalloc((void**)&ptr, 256) != hipSuccess) return nullptr;
    if (hipMemset(ptr, 0, 256) != hipSuccess) return nullptr;
    pages[dev] = ptr;
  }
  return pages[dev];
}

// 0: exact f32 MFMA (default); 1: split-precision bf16x3 operands (lip_set_precision / LIP_PRECISION=bf16x3)
static int g_precision = -1;
int precision_mode() {
  if (g_precision < 0) {
    const char* e = getenv("LIP_PRECISION");
    g_precision = (e && (e[0] == 'b' || e[0] == '1')) ? 1 : 0;
  }
  return g_precision;
}
void set_precision_mode(int m) { g_precision = m ? 1 : 0; }

// split-K of under-filled implicit GEMMs: -1 = not set (environment LIP_NOKSPLIT decides), 0 = off, 1 = on
static int g_split_k = -1;
void set_split_k_mode(int on) { g_split_k = on ? 1 : 0; }
static bool split_k_enabled() {
  if (g_split_k < 0) g_split_k = getenv("LIP_NOKSPLIT") ? 0 : 1;
  return g_split_k == 1;
}

static bool igemm_fast_ok(const IgemmP& p) {
  for (int s = 0; s < p.nseg; ++s) {
    const SegP& q = p.seg[s];
    if ((q.C & 15) != 0 || (q.stride != 1 && q.stride != 2) || q.b_trans) return false;
    if ((((uintptr_t)q.a) & 15) || (q.a_ps & 3)) return false;
  }
  return true;
}

static int cu_count();

// Scratch planes of the split-K launches, one buffer per (device, stream) — kernels of one stream are ordered, so the
// shares of launch i are consumed by its finishing pass before launch i+1 overwrites them; a second stream or device
// gets its own buffer.  Grown on demand (after draining that stream), kept for the life of the process.  Returns null
// when the table of 16 entries is full or the allocation fails (the caller then launches unsplit).
static float* ksplit_scratch(size_t floats, hipStream_t st) {
  struct Entry { int dev; hipStream_t st; float* buf; size_t cap; };
  static Entry table[16];
  static int used = 0;
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  Entry* e = nullptr;
  for (int i = 0; i < used; ++i)
    if (table[i].dev == dev && table[i].st == st) { e = &table[i]; break; }
  if (!e) {
    if (used == 16) return nullptr;
    e = &table[used++];
    e->dev = dev; e->st = st; e->buf = nullptr; e->cap = 0;
  }
  if (e->cap < floats) {
    if (e->buf) { (void)hipStreamSynchronize(st); (void)hipFree(e->buf); e->buf = nullptr; e->cap = 0; }
    if (hipMalloc((void**)&e->buf, floats * sizeof(float)) != hipSuccess) return nullptr;
    e->cap = floats;
  }
  return e->buf;
}

template <int WM, int WN, int TM, int TN>
static hipError_t run_igemm(const IgemmP& p, int P, hipStream_t st) {
  using T = Tile<WM, WN, TM, TN>;
  const long long tiles = (long long)((p.R + T::BM - 1) / T::BM) * ((p.N + T::BN - 1) / T::BN);
  dim3 grid((unsigned)tiles, (unsigned)P, 1);
  static const bool force_generic = getenv("LIP_GENERIC") != nullptr;     // A/B switch
  if (!force_generic && igemm_fast_ok(p)) {
    IgemmP q = p;
    q.zeros = zero_page();
    if (!q.zeros) return hipErrorOutOfMemory;
    static const bool dbg = getenv("LIP_DBG") != nullptr;       // diagnostic stamps (never in a timed run)
    static unsigned long long* dbgbuf = nullptr;
    q.dbg = nullptr;

    if (dbg) {
      if (!dbgbuf && hipMalloc((void**)&dbgbuf, 3 * 8192 * sizeof(unsigned long long)) != hipSuccess) return hipErrorOutOfMemory;
      (void)hipMemsetAsync(dbgbuf, 0, 3 * 8192 * sizeof(unsigned long long), st);
      q.dbg = dbgbuf;
    }
    // stride-2 data gradient on an even grid: parity-class row order, unreachable taps skipped (A/B: LIP_NOPAR)
    static const bool nopar = getenv("LIP_NOPAR") != nullptr;
    const int OH = p.OHW / p.OW;
    bool par = !nopar && (OH % 2 == 0) && (p.OW % 2 == 0), any_s2 = false;
    for (int s = 0; s < p.nseg; ++s) {
      par = par && p.seg[s].mode == 1 && (p.seg[s].stride == 1 || p.seg[s].stride == 2);
      any_s2 = any_s2 || (p.seg[s].mode == 1 && p.seg[s].stride == 2);
    }
    par = par && any_s2;
    // B as dwordx4 when N % 4 == 0, on the tiles of 64+ columns (A/B switch LIP_NOBV4; measured per op at P = 256:
    // N = 128 forward 117.5 -> 128.1 TF, N = 64 forward 111.5 -> 119, backward +2..7 %; the 32-column tile, where only
    // half the threads would carry a B load, lost 5 % and keeps dword loads)
    static const bool nobv4 = getenv("LIP_NOBV4") != nullptr;
    const bool split = precision_mode() == 1;      // (split mode: 2081 -> 1820 GGN-vp/s with dwordx4 B loads — off)
    const bool bv4 = !nobv4 && !split && (p.N & 3) == 0;
    if (par) {
      q.OW2 = p.OW / 2; q.OHW2 = (OH / 2) * q.OW2; q.Rc = (p.R / p.OHW) * q.OHW2;
      q.dOHW2 = FastDiv((unsigned)q.OHW2); q.dOW2 = FastDiv((unsigned)q.OW2);
      grid.x = (unsigned)(4ll * ((q.Rc + T::BM - 1) / T::BM) * ((p.N + T::BN - 1) / T::BN));
    }
    // few probes: fewer blocks than the chip holds and a long K loop — split the K-tiles over gridDim.z (each share
    // >= 12 K-tiles, <= 4 shares; only where the chip is at most half full: at 200 blocks — the 64-column stage at one probe —
    // the second pass costs more than the shorter K loop saves, 37 -> 40 us per launch), raw sums to a scratch plane per share, igemm_finish_kernel adds them
    // and runs the fused epilogue (a fix-up inside the kernel by the share that arrives last at a tile counter needs
    // agent-scope fences: 38 -> 80 us per launch, measured).  A/B switch LIP_NOKSPLIT.
    if constexpr (WM == 2 && TM == 1 && TN == 1) {
      const bool noks = !split_k_enabled();
      int kt = 0;
      for (int s = 0; s < p.nseg; ++s) kt += p.seg[s].Ktot / BK;
      const long long blocks = tiles * P;
      long long ks = (4ll * cu_count()) / (blocks > 0 ? blocks : 1);
      if (ks > kt / 12) ks = kt / 12;
      if (ks > 4) ks = 4;
      const size_t plane = (size_t)P * p.R * p.N;
      if (!noks && !p.no_ksplit && !split && !par && !dbg && ks >= 2 && 2 * blocks <= cu_count() && plane * ks * sizeof(float) <= ((size_t)256 << 20)) {
        // per (device, stream): launches of one stream are ordered.  Split-K is only an optimisation: when the scratch
        // table is full (a process rotating through many streams) or the allocation fails, the launch below runs unsplit
        float* scratch = ksplit_scratch(plane * ks, st);
        if (scratch) {
          q.partial = scratch; q.partial_zs = (long long)plane;
          dim3 g3((unsigned)tiles, (unsigned)P, (unsigned)ks);
          if (bv4) hipLaunchKernelGGL((igemm_fast_kernel<WM, WN, TM, TN, false, false, true, true>), g3, dim3(T::NT), 0, st, q);
          else hipLaunchKernelGGL((igemm_fast_kernel<WM, WN, TM, TN, false, false, false, true>), g3, dim3(T::NT), 0, st, q);
          hipLaunchKernelGGL((igemm_finish_kernel<WM, WN, TM, TN>), grid, dim3(T::NT), 0, st, q, (int)ks);
          return hipGetLastError();
        }
      }
    }
    if constexpr (WM == 4 && WN == 1) {
      // A operand straight into the MFMA registers (A/B switch LIP_NOADIRECT)
      static const bool noad = getenv("LIP_NOADIRECT") != nullptr;
      if (!noad && !split && !par && !dbg) {
        if (bv4) hipLaunchKernelGGL((igemm_adirect_kernel<TM, TN, true>), grid, dim3(256), 0, st, q);
        else hipLaunchKernelGGL((igemm_adirect_kernel<TM, TN, false>), grid, dim3(256), 0, st, q);
        return hipGetLastError();
      }
    }
#define LIP_LAUNCH_IGEMM(S_, P_, V_) hipLaunchKernelGGL((igemm_fast_kernel<WM, WN, TM, TN, S_, P_, V_>), grid, dim3(T::NT), 0, st, q)
    if (split) {
      if (par) LIP_LAUNCH_IGEMM(true, true, false); else LIP_LAUNCH_IGEMM(true, false, false);
    } else {
      if (par) { if (bv4) LIP_LAUNCH_IGEMM(false, true, true); else LIP_LAUNCH_IGEMM(false, true, false); }
      else { if (bv4) LIP_LAUNCH_IGEMM(false, false, true); else LIP_LAUNCH_IGEMM(false, false, false); }
    }
#undef LIP_LAUNCH_IGEMM
    if (dbg) {
      static int reports = 0;
      if (reports < 60) {
        (void)hipStreamSynchronize(st);
        static unsigned long long host[3 * 8192];
        (void)hipMemcpy(host, dbgbuf, sizeof(host), hipMemcpyDeviceToHost);
        const long long nb = (long long)tiles * P < 8192 ? (long long)tiles * P : 8192;
        double loop = 0, epi = 0; long long cnt = 0;
        for (long long b = 0; b < nb; ++b) {
          if (!host[3 * b + 2]) continue;
          loop += (double)(host[3 * b + 1] - host[3 * b]); epi += (double)(host[3 * b + 2] - host[3 * b + 1]);
          ++cnt;
        }
        if (cnt) fprintf(stderr, "[lip dbg] igemm<%d,%d,%d,%d> split=%d blocks=%lld (of %lld) nseg=%d N=%d: K-loop %.0f cycles, epilogue+drain %.0f cycles per block\n",
                         WM, WN, TM, TN, precision_mode(), cnt, (long long)tiles * P, p.nseg, p.N, loop / cnt, epi / cnt);
        ++reports;
      }
    }
  }
  else
    hipLaunchKernelGGL((igemm_kernel<WM, WN, TM, TN>), grid, dim3(T::NT), 0, st, p);
  return hipGetLastError();
}

static int tile_override() {
  static int v = -2;
  if (v == -2) { const char* e = getenv("LIP_TILE"); v = e ? atoi(e) : -1; }
  return v;
}

static int cu_count() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t pr;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
            ? pr.multiProcessorCount : 256;
  }
  return n;
}

// Winograd route of the 3x3 / stride-1 layers: -1 = not set (environment: LIP_NOWINO -> off), 0 = off, 1 = on (default:
// every eligible launch), 2 = on (kept for the tests that force the route; same launches as 1 since the fill rule went)
static int g_wino = -1;
void set_wino_mode(int m) { g_wino = (m < 0 || m > 2) ? 1 : m; }
int wino_mode() {
  if (g_wino < 0) {
    const char* f = getenv("LIP_WINO");
    g_wino = getenv("LIP_NOWINO") ? 0 : ((f && f[0] == 'f') ? 2 : 1);
  }
  return g_wino;
}

struct WinoGeom { int BWs, BHs, NI, FR, FC, nbx, nby, nbi, NS, TH, TW; };
static WinoGeom wino_geom(int OH, int OW, long long n_img) {
  WinoGeom g;
  g.TH = OH / 2; g.TW = OW / 2;
  g.BWs = 0; while ((1 << g.BWs) < g.TW && g.BWs < 4) ++g.BWs;
  g.BHs = 0; while ((1 << g.BHs) < g.TH && g.BWs + g.BHs < 5) ++g.BHs;
  g.NI = 32 >> (g.BWs + g.BHs);
  g.FR = 2 * (1 << g.BHs) + 2; g.FC = 2 * (1 << g.BWs) + 2; g.NS = g.NI * g.FR * g.FC;
  g.nbx = (g.TW + (1 << g.BWs) - 1) >> g.BWs; g.nby = (g.TH + (1 << g.BHs) - 1) >> g.BHs;
  g.nbi = (int)((n_img + g.NI - 1) / g.NI);
  return g;
}

static bool igemm_wino_ok(const IgemmP& p, int P) {
  const int mode = wino_mode();
  if (mode == 0 || precision_mode() != 0 || p.no_ksplit) return false;
  if (p.nseg < 1 || p.N < 32 || (p.N & 31) != 0) return false;
  const int OH = p.OHW / p.OW;
  if ((OH & 1) || (p.OW & 1) || OH * p.OW != p.OHW) return false;
  const long long n_img = p.R / p.OHW;
  for (int s = 0; s < p.nseg; ++s) {
    const SegP& q = p.seg[s];
    if (q.KH != 3 || q.KW != 3 || q.stride != 1 || q.pad_h != 1 || q.pad_w != 1 || q.b_trans) return false;
    if (q.IH != OH || q.IW != p.OW || (q.C & 31) != 0 || (q.mode != 0 && q.mode != 1)) return false;
    if ((((uintptr_t)q.a) & 15) || (q.a_ps & 3) || (((uintptr_t)q.b) & 3)) return false;
    if (n_img * p.OHW * q.C * 4 >= (1ll << 31) || 16ll * q.C * p.N * 4 >= (1ll << 31)) return false;
  }
  const WinoGeom g = wino_geom(OH, p.OW, n_img);
  if (g.NS > WINO_SLOTS) return false;                 // (maps smaller than 8 x 8: the direct kernels)
  // (a rule "only launches that fill the chip" was measured and dropped: with the route on every eligible launch a
  //  single product takes 1.85 instead of 2.02 ms, two 2.21 instead of 3.43, four 2.99 instead of 4.66, eight 4.65
  //  instead of 5.15 — an under-filled Winograd launch lasts one short block, an under-filled direct launch one long one)
  (void)P;
  return true;
}

static hipError_t run_igemm_wino(const IgemmP& p, int P, hipStream_t st) {
  IgemmP q = p;
  WinoX wx;
  const int OH = p.OHW / p.OW;
  const long long n_img = p.R / p.OHW;
  size_t need = 0;
  for (int s = 0; s < p.nseg; ++s) need += (size_t)(p.seg[s].b_ps ? P : 1) * 16 * p.seg[s].C * p.N;
  float* scratch = ksplit_scratch(need, st);
  if (!scratch) return hipErrorOutOfMemory;
  size_t off = 0;
  for (int s = 0; s < p.nseg; ++s) {
    const SegP& g = p.seg[s];
    const long long un = 16ll * g.C * p.N;
    const int PW = g.b_ps ? P : 1;
    hipLaunchKernelGGL(wino_weight_transform_kernel, dim3((unsigned)(((g.C / 4) * p.N + 255) / 256), (unsigned)PW), dim3(256), 0, st,
                       g.b, g.b_ps, scratch + off, un, g.C, p.N, g.mode == 1 ? 1 : 0);
    q.seg[s].b = scratch + off;
    q.seg[s].b_ps = g.b_ps ? un : 0;
    wx.a_bytes[s] = (unsigned)(n_img * p.OHW * g.C * 4);
    wx.u_bytes[s] = (unsigned)(un * 4);
    off += (size_t)PW * un;
  }
  for (int s = p.nseg; s < 3; ++s) { wx.a_bytes[s] = 0; wx.u_bytes[s] = 0; }
  const WinoGeom g = wino_geom(OH, p.OW, n_img);
  wx.BWs = g.BWs; wx.BHs = g.BHs; wx.NI = g.NI; wx.FR = g.FR; wx.FC = g.FC; wx.nbx = g.nbx; wx.nby = g.nby; wx.NS = g.NS;
  wx.n_img = (int)n_img; wx.TH = g.TH; wx.TW = g.TW;
  wx.inv_frfc = 1.f / (float)(g.FR * g.FC); wx.inv_fc = 1.f / (float)g.FC;
  wx.dnb = FastDiv((unsigned)(p.N / 32)); wx.dbxy = FastDiv((unsigned)(g.nbx * g.nby)); wx.dnbx = FastDiv((unsigned)g.nbx);
  q.zeros = nullptr; q.dbg = nullptr; q.partial = nullptr;
  const size_t shmem = (size_t)(8192 + 32 + 64) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)igemm_wino_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)igemm_wino_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  // 16-byte epilogue accesses when every operand tensor allows them (A/B switch LIP_WINO_NOVEPI)
  auto al16 = [](const void* ptr, long long ps) { return ptr == nullptr || ((((uintptr_t)ptr) & 15) == 0 && (ps & 3) == 0); };
  static const bool novepi = getenv("LIP_WINO_NOVEPI") != nullptr;
  const bool vepi = !novepi && al16(p.out, p.out_ps) && al16(p.res, p.res_ps) && al16(p.xhat, 0) && al16(p.dphi, 0) && al16(p.xhat2, 0);
  dim3 grid((unsigned)((long long)g.nbx * g.nby * g.nbi * (p.N / 32)), (unsigned)P, 1);
  if (vepi) hipLaunchKernelGGL(igemm_wino_kernel<true>, grid, dim3(256), shmem, st, q, wx);
  else hipLaunchKernelGGL(igemm_wino_kernel<false>, grid, dim3(256), shmem, st, q, wx);
  return hipGetLastError();
}

hipError_t launch_igemm(const IgemmP& p, int P, hipStream_t st) {
  if (igemm_first_ok(p, P)) return run_igemm_first(p, P, st);
  if (igemm_wino_ok(p, P)) {
    // (no scratch for the transformed weights — a 17th stream, or the allocation failed: the direct kernels below)
    const hipError_t e = run_igemm_wino(p, P, st);
    if (e != hipErrorOutOfMemory) return e;
    (void)hipGetLastError();
  }
  // Few probes (single-vector Krylov loops): with fewer 128-row blocks than CUs the launch time is ONE block's K loop,
  // so take the tiles with the least work per wave (32x32 per wave, 64-row blocks) — at P = 1 the 128-channel layers
  // of the CIFAR net run 25 blocks of 144 K-tiles otherwise (measured 177 us per launch).  A/B switch LIP_NOSMALLP.
  static const bool nosmallp = getenv("LIP_NOSMALLP") != nullptr;
  const long long blocks128 = (long long)((p.R + 127) / 128) * ((p.N + (p.N > 64 ? 127 : (p.N > 32 ? 63 : 31))) / (p.N > 64 ? 128 : (p.N > 32 ? 64 : 32))) * P;
  static const int smallp_factor = getenv("LIP_SMALLP_FACTOR") ? atoi(getenv("LIP_SMALLP_FACTOR")) : 2;   // 1..4 measured: +-3 %
  if (!nosmallp && p.R > 64 && blocks128 < (long long)smallp_factor * cu_count()) {
    if (p.N > 32) return run_igemm<2, 2, 1, 1>(p, P, st);
    return run_igemm<2, 1, 1, 1>(p, P, st);
  }
  // experiment switches (A/B only): LIP_TILE=2 one-wave blocks (32 x 32 / 32 x 64 per block, no cross-wave barrier),
  // LIP_TILE=3 one wave with two row tiles (64 x 32), LIP_TILE=4 two-wave 64-row blocks
  if (tile_override() == 2 && p.N <= 64) return p.N > 32 ? run_igemm<1, 1, 1, 2>(p, P, st) : run_igemm<1, 1, 1, 1>(p, P, st);
  if (tile_override() == 3 && p.N <= 32) return run_igemm<1, 1, 2, 1>(p, P, st);
  if (tile_override() == 4 && p.N <= 32) return run_igemm<2, 1, 1, 1>(p, P, st);
  const bool small_m = p.R <= 64;
  const bool big_m = p.R >= 4096 && tile_override() == 1;    // LIP_TILE=1: 256-row tiles (A/B: slower on every shape, r1 and r2)
  if (p.N > 64) return small_m ? run_igemm<2, 2, 1, 2>(p, P, st) : run_igemm<2, 2, 2, 2>(p, P, st);
  if (p.N > 32) return small_m ? run_igemm<2, 2, 1, 1>(p, P, st) : (big_m ? run_igemm<4, 1, 2, 2>(p, P, st) : run_igemm<4, 1, 1, 2>(p, P, st));
  return small_m ? run_igemm<2, 1, 1, 1>(p, P, st) : (big_m ? run_igemm<4, 1, 2, 1>(p, P, st) : run_igemm<4, 1, 1, 1>(p, P, st));
}


// Row split (float atomics into Y) of a weight-gradient launch with `blocks` blocks before splitting and `occ`
// resident blocks per CU.  Blocks are long (R/16 K-tiles) and, at small probe counts, few: 640 blocks on 768 slots
// leave a sixth of the chip idle, 1152 on 512 run a third round a quarter full.  Fill the chip when under-filled
// and otherwise prefer a split that ends on a whole number of rounds, but only while a launch has few rounds (at
// 256 probes the same rule measured +-0).  >= 64 K-tiles stay in every split.
static int auto_ksplit(long long blocks, int occ, int R, int tile_elems, int mfma_per_wave_tile) {
  const double slots = (double)cu_count() * occ;
  const int maxks = R / (64 * BK) > 0 ? R / (64 * BK) : 1;
  if ((double)blocks < slots) {
    // under-filled (few probes): splitting shortens every block's K loop (~700 + 64 * MFMAs cycles per 16-row K-tile
    // for a lone wave) but each split adds tile_elems float atomics per block, which the L2 retires at ~128 per clock
    // chip-wide: minimise  ktiles * cycles_per_tile / ks  +  ks * blocks * tile_elems / 128  — e.g. ks = 22 rather
    // than "fill the chip" (50) for the 1152 x 128 weight gradient at one probe (measured 50 -> 21 us).
    const double ktiles = (double)R / BK, per_tile = 700.0 + 64.0 * mfma_per_wave_tile;
    double ks_opt = sqrt(ktiles * per_tile / ((double)blocks * tile_elems / 128.0));
    long long ks = (long long)(ks_opt + 0.5);
    const long long fill = (long long)((2.0 * slots + blocks - 1) / blocks), cap = R / 64 > 0 ? R / 64 : 1;
    if (ks > fill) ks = fill;
    if (ks > cap) ks = cap;
    if (ks > 1024) ks = 1024;
    return (int)(ks < 1 ? 1 : ks);
  }
  int best = 1;
  double best_eff = 0.0;
  for (int ks = 1; ks <= 4 && ks <= maxks; ++ks) {
    const double rounds = (double)blocks * ks / slots;
    if (ks > 1 && rounds > 8.0) break;
    const double eff = rounds / (double)(long long)(rounds + 0.999999);
    if (eff > best_eff + 0.03) { best_eff = eff; best = ks; }
  }
  return best;
}

template <int WM, int WN, int TM, int TN>
static hipError_t run_wgrad(const WgradP& p0, int P, hipStream_t st) {
  using T = Tile<WM, WN, TM, TN>;
  const long long tiles = (long long)((p0.M + T::BM - 1) / T::BM) * ((p0.N + T::BN - 1) / T::BN);
  WgradP p = p0;
  if (p.ksplit <= 0) p.ksplit = auto_ksplit(tiles * P, TM * TN >= 4 ? 2 : (TM * TN == 2 ? 3 : 4), p.R, T::BM * T::BN, 8 * TM * TN);
  dim3 grid((unsigned)tiles, (unsigned)P, (unsigned)p.ksplit);
  static const bool force_generic = getenv("LIP_GENERIC") != nullptr;     // A/B switch
  if (!force_generic && (p.C & 3) == 0 && (((uintptr_t)p.a) & 15) == 0) {
    WgradP q = p;
    q.zeros = zero_page();
    if (!q.zeros) return hipErrorOutOfMemory;
    // cotangent rows as float4 on the 64+ column tiles (N % 4 == 0, 16-byte aligned slot; A/B switch LIP_NOBV4)
    static const bool nobv4 = getenv("LIP_NOBV4") != nullptr;
    const bool v4 = (p.N & 3) == 0 && (p.g_ps & 3) == 0 && (((uintptr_t)p.g) & 15) == 0;
    if constexpr (T::AQ == 2) {
      if (precision_mode() == 1 && v4) {              // split precision (bf16x3 operands), every tile width
        hipLaunchKernelGGL((wgrad_fast_kernel<WM, WN, TM, TN, false, true, true>), grid, dim3(T::NT), 0, st, q);
        return hipGetLastError();
      }
    }
    if (!nobv4 && T::BN >= 64 && v4)
      hipLaunchKernelGGL((wgrad_fast_kernel<WM, WN, TM, TN, false, true>), grid, dim3(T::NT), 0, st, q);
    else
      hipLaunchKernelGGL((wgrad_fast_kernel<WM, WN, TM, TN, false, false>), grid, dim3(T::NT), 0, st, q);
  }
  else
    hipLaunchKernelGGL((wgrad_kernel<WM, WN, TM, TN>), grid, dim3(T::NT), 0, st, p);
  return hipGetLastError();
}

// Probe-batched weight gradient (N <= 64): columns = P*N, 128-column tiles; 96-row tiles (three waves) when
// M = KH*KW*C is a multiple of 96 but not of 128 (288, 576).  The row reduction is split (float atomics) until
// the launch has ~6 blocks per CU.
template <int WM, int WN, int TM, int TN>
static hipError_t run_wgrad_pb(const WgradP& p, int P, hipStream_t st) {
  using T = Tile<WM, WN, TM, TN>;
  const long long tiles = (long long)((p.M + T::BM - 1) / T::BM) * (((long long)P * p.N + T::BN - 1) / T::BN);
  WgradP q = p;
  q.zeros = zero_page();
  if (!q.zeros) return hipErrorOutOfMemory;
  q.P = P;
  if (p.ksplit <= 0) {
    long long ks = (1536 + tiles - 1) / tiles;
    const long long maxks = tiles >= 64 ? (p.R + 64 * BK - 1) / (64 * BK)        // >= 64 K-tiles per split ...
                                        : (p.R / 64 > 0 ? p.R / 64 : 1);          // ... 4 when there are few probes
    if (ks > maxks) ks = maxks;
    if (ks > 1024) ks = 1024;
    q.ksplit = ks < 1 ? 1 : (int)ks;
  }
  dim3 grid((unsigned)tiles, 1, (unsigned)q.ksplit);
  if constexpr (T::NT % (T::BM / 4) == 0) {        // (the split-precision loader needs one m-quad per thread)
    if (precision_mode() == 1) {
      hipLaunchKernelGGL((wgrad_fast_kernel<WM, WN, TM, TN, true, true, true>), grid, dim3(T::NT), 0, st, q);
      return hipGetLastError();
    }
  }
  hipLaunchKernelGGL((wgrad_fast_kernel<WM, WN, TM, TN, true>), grid, dim3(T::NT), 0, st, q);
  return hipGetLastError();
}

hipError_t launch_wgrad(const WgradP& p, int P, hipStream_t st) {
  if (wgrad_first_ok(p, P)) return run_wgrad_first(p, P, st);
  if (wgrad_skinny_ok(p)) {
    if (p.R <= 16) return run_wgrad_skinny<2, 8>(p, P, st);
    if (p.R <= 52) return run_wgrad_skinny<2, 26>(p, P, st);
    return run_wgrad_skinny<2, 32>(p, P, st);
  }
  if (wgrad_wino_ok(p, P)) {
    const hipError_t e = run_wgrad_wino(p, P, st);
    if (e != hipErrorOutOfMemory) return e;             // (no scratch for the transformed activations: the direct kernels)
    (void)hipGetLastError();
  }
  if (p.overwrite) return hipErrorInvalidValue;       // the engine asks for it only where wgrad_will_overwrite() holds
  static const bool nopb = getenv("LIP_NOPB") != nullptr || getenv("LIP_GENERIC") != nullptr;   // A/B switch
  const bool pb_ok = !nopb && P > 1 && p.N <= 64 && (p.N & 3) == 0 && p.M >= 96 && (p.g_ps & 3) == 0 && (((uintptr_t)p.g) & 15) == 0 && (p.C & 3) == 0 && (((uintptr_t)p.a) & 15) == 0 &&
                     (long long)(P - 1) * p.g_ps + (long long)p.R * p.N < (1ll << 32);
  // measured on MI355X (CIFAR ResNet1M, P = 256): N = 32, M = 288: 3.72 -> 2.67 ms; N = 64, M = 288: 1.58 -> 1.36 ms;
  // N = 64, M = 576 with 128-row tiles (11 % padded rows): 2.61 -> 2.68 ms — that case takes the 96-row tile below
  const int waste128 = (p.M + 127) / 128 * 128 - p.M;
  static const bool pb96 = getenv("LIP_NOPB96") == nullptr;     // A/B: 96-row probe-batched tile also for N = 64, M = 576
  if (pb_ok && (p.N <= 32 || 5 * waste128 >= p.M || (pb96 && p.M % 96 == 0 && p.M % 128 != 0))) {
    // 96-row tiles for M = 288 / 576: four waves side by side along the probes' columns (each wave 96 x 32 = one
    // probe's channels) — the three-wave form <3,1,1,4> (208 registers: two blocks of three waves per CU) put 2-2-1-1
    // waves on the four SIMDs; it stays for the split-precision mode, whose row-pair loader needs one m-quad per thread
    static const bool w3 = getenv("LIP_WGRAD3") != nullptr;          // A/B switch
    if (p.M % 96 == 0 && p.M % 128 != 0)
      return (precision_mode() == 1 || w3) ? run_wgrad_pb<3, 1, 1, 4>(p, P, st) : run_wgrad_pb<1, 4, 3, 1>(p, P, st);
    return run_wgrad_pb<2, 2, 2, 2>(p, P, st);
  }
  // (64-row per-probe tiles for M = 288 were measured slower than 128-row ones: 54.6 vs 52.4 ms per step — removed;
  //  a 96 x 256 probe-batched tile <1,4,3,2> — the transposed im2col gather shared by eight probes instead of four — ran
  //  the M = 288 / 576 launches at 105 instead of 113 TFLOP/s: removed)
  const bool small_m = p.M <= 64;
  if (p.N > 64) return small_m ? run_wgrad<2, 2, 1, 2>(p, P, st) : run_wgrad<2, 2, 2, 2>(p, P, st);
  if (p.N > 32) return small_m ? run_wgrad<2, 2, 1, 1>(p, P, st) : run_wgrad<4, 1, 1, 2>(p, P, st);
  return small_m ? run_wgrad<2, 1, 1, 1>(p, P, st) : run_wgrad<4, 1, 1, 1>(p, P, st);
}

}  // namespace lip
