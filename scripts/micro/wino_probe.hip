// Winograd F(2x2, 3x3) implicit GEMM on the f32 matrix pipe — feasibility probe for the 3x3 stride-1 layers
// (tangent-forward and data-gradient ops of the CIFAR net: 95 % of a sweep's FLOPs).
//   hipcc --offload-arch=gfx950 -O3 wino_probe.hip -o wino_probe && ./wino_probe
// Y = A^T [ (G w G^T) (.) (B^T d B) ] A per 2x2 output patch; 16 positions xi = (a, b), each a GEMM over channels:
// M_xi[tile][n] = sum_c V_xi[tile][c] U_xi[c][n]  — 4 multiplications per output pixel and (c, n) instead of 9.
// One block = 32 tiles (128 output pixels) x 32 TN columns; wave a owns row a of the 4x4 transformed patch:
//   * A operand: lane (tile i, half h) loads the 2 x 4 pixels rows r1(a), r2(a) of its tile's 4x4 input patch, 4 channels
//     each (buffer_load_dwordx4, out-of-image pixels dropped by the range check), forms V[a][0..3] in registers
//     (8 adds per channel) = the A registers of 16 MFMAs (k-step (g, j): lane half h <-> channel 8g + 4h + j);
//   * B operand: U stored [xi][C/4][N][4] so that lane (n, h) reads its 4 channels of U_xi[.][n] as one dwordx4,
//     straight into the MFMA registers; no LDS and no barrier in the channel loop;
//   * output transform: in-wave over b, across the four waves over a through LDS.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// (element access through __builtin_bit_cast(float, v[j]) on the u32x4 a buffer load returns is miscompiled by this
//  hipcc — only element 0 survives — so the loads are cast to f32x4 as whole vectors)

struct WinoP {
  const float* a; long long a_ps;     // [P][n][H][W][C]
  const float* u; long long u_ps;     // [P or 1][16][C/4][N][4]
  float* out; long long out_ps;       // [P][n][H][W][N]
  int n_img, H, W, C, N, TH, TW, ntiles;
  unsigned a_bytes, u_bytes;
  // v2: a block covers NI images x BH x BW tiles (NI BH BW = 32), its input footprint NI x FR x FC pixels is staged in LDS
  int BWs, BHs, NI, FR, FC, nbx, nby, nbi, NS;
  unsigned long long* stamps;   // [blocks][8] s_memtime stamps of wave 0 (diagnostic) or null
};

__global__ void wino_weight_transform(const float* __restrict__ w, long long w_ps, float* __restrict__ u, long long u_ps,
                                      int C, int N, int flip) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= C * N) return;
  const int c = e / N, n = e - c * N;
  const float* wp = w + (long long)blockIdx.y * w_ps;
  float* up = u + (long long)blockIdx.y * u_ps;
  float g[3][3];
  for (int kh = 0; kh < 3; ++kh)
    for (int kw = 0; kw < 3; ++kw) {
      const int sh = flip ? 2 - kh : kh, sw = flip ? 2 - kw : kw;
      g[kh][kw] = wp[((long long)(sh * 3 + sw) * C + c) * N + n];
    }
  float t[4][3];                         // G w
  for (int kw = 0; kw < 3; ++kw) {
    t[0][kw] = g[0][kw];
    t[1][kw] = 0.5f * (g[0][kw] + g[1][kw] + g[2][kw]);
    t[2][kw] = 0.5f * (g[0][kw] - g[1][kw] + g[2][kw]);
    t[3][kw] = g[2][kw];
  }
  for (int a = 0; a < 4; ++a) {
    const float v0 = t[a][0], v1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]), v2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]), v3 = t[a][2];
    const float vv[4] = {v0, v1, v2, v3};
    for (int b = 0; b < 4; ++b) up[(((long long)(4 * a + b) * (C / 4) + (c >> 2)) * N + n) * 4 + (c & 3)] = vv[b];
  }
}

// DIAG (timing-only ablations): 1 = every lane reads pixel 0 (perfect L1 locality), 2 = B loaded once (no B loads in the
// loop), 3 = no cross-wave exchange (each wave stores its own sums), 4 = A loaded once
template <int TN, int DIAG = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void wino_kernel(const WinoP prm) {
  extern __shared__ float lds[];                        // [4 a][2 q][TN][16 reg][64 lane] floats + 32 tile bases
  const int tid = threadIdx.x, lane = tid & 63;
  const int a = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int N = prm.N, C = prm.C, H = prm.H, W = prm.W;
  const int nb = N / (32 * TN);
  const int tb = blockIdx.x / nb, cb = blockIdx.x - tb * nb;
  const int p = blockIdx.y;
  const int n0 = cb * 32 * TN;
  int* tbase = reinterpret_cast<int*>(lds + 4 * 2 * TN * 1024);

  const int t = tb * 32 + l31;
  const bool tv = t < prm.ntiles;
  const int tpi = prm.TH * prm.TW;
  const int img = t / tpi, rem = t - img * tpi;
  const int ty = rem / prm.TW, tx = rem - ty * prm.TW;
  if (tid < 32) tbase[tid] = tv ? ((img * H + 2 * ty) * W + 2 * tx) * N : -1;

  // rows of the 4x4 patch this wave's transform row needs:  e = d[r1] + sg d[r2]
  const int r1 = (a == 0) ? 0 : (a == 2 ? 2 : 1);
  const int r2 = (a == 0) ? 2 : (a == 1 ? 2 : (a == 2 ? 1 : 3));
  const float sg = (a == 1) ? 1.f : -1.f;
  unsigned voff[8];
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const int ih = 2 * ty - 1 + (rr ? r2 : r1);
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const int iw = 2 * tx - 1 + cc;
      const bool ok = tv && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
      voff[4 * rr + cc] = ok ? (unsigned)((((img * H + ih) * W + iw) * C + 4 * h) * 4) : 0x80000000u;
      if (DIAG == 1 || DIAG == 6) voff[4 * rr + cc] = (unsigned)(4 * h * 4);
    }
  }
  const float* ap = prm.a + (long long)p * prm.a_ps;
  const float* up = prm.u + (long long)p * prm.u_ps;
  const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ap), 0, prm.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(up), 0, prm.u_bytes, 0x00020000);
  const int c4 = C >> 2;
  // lane part of the U address: xi = 4a, channel quad h, column n0 + l31
  const unsigned uvoff = (unsigned)((((4 * a) * c4 + h) * N + n0 + l31) * 16);
  const unsigned ub_stride = (unsigned)(c4 * N * 16);          // next b
  const unsigned ug_stride = (unsigned)(2 * N * 16);           // next 8-channel group

  f32x16 acc[4][TN];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][tn][r] = 0.f;

  const int G = C >> 3;
  f32x4 raw[8];
  f32x4 bq[2][4][TN];
  auto load_a = [&](int g) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 8; ++q) raw[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, voff[q], g * 32, 0));
  };
  auto load_b = [&](int g, f32x4 (&dst)[4][TN]) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        dst[b][tn] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ursrc, uvoff, b * ub_stride + g * ug_stride + tn * 512, 0));
  };
  auto transform = [&](float (&v)[4][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float e[4];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) e[cc] = fmaf(sg, raw[4 + cc][j], raw[cc][j]);
      v[0][j] = e[0] - e[2]; v[1][j] = e[1] + e[2]; v[2][j] = e[2] - e[1]; v[3][j] = e[1] - e[3];
    }
  };
  auto sweep = [&](const float (&v)[4][4], const f32x4 (&bc)[4][TN]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[b][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[b][j], bc[b][tn][j], acc[b][tn], 0, 0, 0);
  };
  // branch-free channel loop (C % 16 == 0: an even number of 8-channel groups); the last iteration re-requests the
  // final group instead of testing for the end
  load_a(0);
  load_b(0, bq[0]);
  for (int g = 0; g < G; g += 2) {
    float v[4][4];
    transform(v);
    if (DIAG != 4) load_a(g + 1);
    if (DIAG != 2) load_b(g + 1, bq[1]);
    sweep(v, bq[0]);
    transform(v);
    const int gn = (g + 2 < G) ? g + 2 : G - 1;
    if (DIAG != 4) load_a(gn);
    if (DIAG != 2) load_b(gn, bq[0]);
    sweep(v, DIAG == 2 ? bq[0] : bq[1]);
  }

  // output transform, in-wave part: T[q] over b
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float m0 = acc[0][tn][r], m1 = acc[1][tn][r], m2 = acc[2][tn][r], m3 = acc[3][tn][r];
      if (DIAG == 3) continue;
      lds[(((a * 2 + 0) * TN + tn) * 16 + r) * 64 + lane] = m0 + m1 + m2;
      lds[(((a * 2 + 1) * TN + tn) * 16 + r) * 64 + lane] = m1 - m2 - m3;
    }
  }
  if (DIAG != 3) __syncthreads();
  // wave w writes output pixel (po, qo) of every tile: Y = T_0 + T_1 + T_2 (po = 0) or T_1 - T_2 - T_3 (po = 1)
  const int po = a >> 1, qo = a & 1;
  float* outp = prm.out + (long long)p * prm.out_ps + (po * W + qo) * N + n0 + l31;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (DIAG == 3) { const int i = (r & 3) + 8 * (r >> 2) + 4 * h; const int base = tbase[i]; if (base >= 0) outp[base + tn * 32] = acc[0][tn][r] + acc[1][tn][r] + acc[2][tn][r] + acc[3][tn][r]; continue; }
      const float t1 = lds[(((1 * 2 + qo) * TN + tn) * 16 + r) * 64 + lane];
      const float t2 = lds[(((2 * 2 + qo) * TN + tn) * 16 + r) * 64 + lane];
      const float tx3 = lds[((((po ? 3 : 0) * 2 + qo) * TN + tn) * 16 + r) * 64 + lane];
      const float y = po ? (t1 - t2 - tx3) : (tx3 + t1 + t2);
      const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
      const int base = tbase[i];
      if (base >= 0 && ((DIAG != 5 && DIAG != 6) || y != y)) outp[base + tn * 32] = y;
    }
  }
}


// ---- v2: the block's input footprint goes global -> registers -> LDS in whole 128-byte lines (8 lanes per pixel, 32 channels
// per chunk), the waves read their patch pixels from LDS (ds_read_b128, pixel slots padded to 144 bytes).  v1 fetched 16 bytes
// of 32 different lines per lane-instruction: the L1 fill rate, not the matrix pipe, set its speed (timing ablation: every
// lane reading pixel 0 ran 1.4 - 2.0x faster).
template <int TN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void wino2_kernel(const WinoP prm) {
  const unsigned long long t_start = __builtin_amdgcn_s_memtime();
  extern __shared__ __attribute__((aligned(16))) float lds[];   // stage [NS][36] floats  |  exchange [4][2][TN][16][64]   (aliased), then tbase[32]
  constexpr int SLOT = 36;
  const int tid = threadIdx.x, lane = tid & 63;
  const int a = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int N = prm.N, H = prm.H, W = prm.W;
  const int nb = N / (32 * TN);
  const int cb = blockIdx.x % nb, tbid = blockIdx.x / nb;
  const int bx = tbid % prm.nbx, by = (tbid / prm.nbx) % prm.nby, bi = tbid / (prm.nbx * prm.nby);
  const int p = blockIdx.y;
  const int n0 = cb * 32 * TN;
  constexpr int XF = 8 * TN * 1024;
  int* tbase = reinterpret_cast<int*>(lds + (XF > 224 * SLOT ? XF : 224 * SLOT));
  const int BW = 1 << prm.BWs, BH = 1 << prm.BHs, FR = prm.FR, FC = prm.FC;
  // this lane's tile
  const int ni = l31 >> (prm.BWs + prm.BHs), dy = (l31 >> prm.BWs) & (BH - 1), dx = l31 & (BW - 1);
  const int img = bi * prm.NI + ni, ty = by * BH + dy, tx = bx * BW + dx;
  const bool tv = img < prm.n_img && ty < prm.TH && tx < prm.TW;
  if (tid < 32) tbase[tid] = tv ? ((img * H + 2 * ty) * W + 2 * tx) * N : -1;
  const int r1 = (a == 0) ? 0 : (a == 2 ? 2 : 1);
  const int r2 = (a == 0) ? 2 : (a == 1 ? 2 : (a == 2 ? 1 : 3));
  const float sg = (a == 1) ? 1.f : -1.f;
  const int rb1 = ((ni * FR + 2 * dy + r1) * FC + 2 * dx) * SLOT + 4 * h;
  const int rb2 = ((ni * FR + 2 * dy + r2) * FC + 2 * dx) * SLOT + 4 * h;
  // staging: thread -> (slot, 16-byte part) x 7
  const int part = tid & 7;
  const float inv_frfc = 1.f / (float)(FR * FC), inv_fc = 1.f / (float)FC;
  int spix[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int slot = (tid >> 3) + 32 * j;
    const int sni = (int)(((float)slot + 0.5f) * inv_frfc), srem = slot - sni * FR * FC;     // exact: slot < 256
    const int fr = (int)(((float)srem + 0.5f) * inv_fc), fc = srem - fr * FC;
    const int simg = bi * prm.NI + sni, ih = 2 * by * BH - 1 + fr, iw = 2 * bx * BW - 1 + fc;
    const bool ok = slot < prm.NS && simg < prm.n_img && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    spix[j] = ok ? (simg * H + ih) * W + iw : -1;
  }
  const int C = prm.C, c4 = C >> 2, chunks = C >> 5, GT = C >> 3;
  const float* ap = prm.a + (long long)p * prm.a_ps;
  const float* up = prm.u + (long long)p * prm.u_ps;
  const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ap), 0, prm.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(up), 0, prm.u_bytes, 0x00020000);
  unsigned svoff[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) svoff[j] = spix[j] >= 0 ? (unsigned)((spix[j] * C + 4 * part) * 4) : 0x80000000u;
  const unsigned uvoff = (unsigned)((((4 * a) * c4 + h) * N + n0 + l31) * 16);
  const unsigned ub_stride = (unsigned)(c4 * N * 16), ug_stride = (unsigned)(2 * N * 16);

  f32x16 acc[4][TN];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][tn][r] = 0.f;

  f32x4 sreg[7], bq[2][4][TN];
  auto stage_load = [&](int k) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 7; ++j) sreg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, svoff[j], k * 128, 0));
  };
  auto stage_store = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int slot = (tid >> 3) + 32 * j;
      if (slot < 224) reinterpret_cast<f32x4*>(lds)[slot * 9 + part] = sreg[j];
    }
  };
  auto load_b = [&](int g, f32x4 (&dst)[4][TN]) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        dst[b][tn] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ursrc, uvoff, b * ub_stride + g * ug_stride + tn * 512, 0));
  };
  const f32x4* lds4 = reinterpret_cast<const f32x4*>(lds);
  const int rq1 = rb1 >> 2, rq2 = rb2 >> 2;          // in 16-byte units (SLOT = 9 units)
  auto group = [&](int g, int gnext, const f32x4 (&bc)[4][TN], f32x4 (&bn)[4][TN]) __attribute__((always_inline)) {   // g: group inside the chunk
    float v[4][4];
    f32x4 raw[8];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      raw[cc] = lds4[rq1 + cc * 9 + 2 * g];
      raw[4 + cc] = lds4[rq2 + cc * 9 + 2 * g];
    }
    load_b(gnext, bn);
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
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[b][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[b][j], bc[b][tn][j], acc[b][tn], 0, 0, 0);
  };
  const unsigned long long t_setup = __builtin_amdgcn_s_memtime();
  unsigned long long t_arrive = 0, t_staged = 0;
  stage_load(0);
  load_b(0, bq[0]);
  for (int k = 0; k < chunks; ++k) {
    if (prm.stamps && k == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t_arrive = __builtin_amdgcn_s_memtime(); }
    stage_store();
    __syncthreads();
    if (prm.stamps && k == 0) t_staged = __builtin_amdgcn_s_memtime();
    const int g0 = 4 * k;
    __builtin_amdgcn_sched_barrier(0);
    group(0, g0 + 1, bq[0], bq[1]);
    group(1, g0 + 2, bq[1], bq[0]);
    group(2, g0 + 3, bq[0], bq[1]);
    group(3, (g0 + 4 < GT) ? g0 + 4 : GT - 1, bq[1], bq[0]);
    __builtin_amdgcn_sched_barrier(0);
    stage_load((k + 1 < chunks) ? k + 1 : k);      // (the last one is redundant: no branch in the loop)
    __syncthreads();
  }
  const unsigned long long t_loop = __builtin_amdgcn_s_memtime();

#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float m0 = acc[0][tn][r], m1 = acc[1][tn][r], m2 = acc[2][tn][r], m3 = acc[3][tn][r];
      lds[(((a * 2 + 0) * TN + tn) * 16 + r) * 64 + lane] = m0 + m1 + m2;
      lds[(((a * 2 + 1) * TN + tn) * 16 + r) * 64 + lane] = m1 - m2 - m3;
    }
  }
  __syncthreads();
  const unsigned long long t_xch = __builtin_amdgcn_s_memtime();
  const int po = a >> 1, qo = a & 1;
  float* outp = prm.out + (long long)p * prm.out_ps + (po * W + qo) * N + n0 + l31;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float t1 = lds[(((1 * 2 + qo) * TN + tn) * 16 + r) * 64 + lane];
      const float t2 = lds[(((2 * 2 + qo) * TN + tn) * 16 + r) * 64 + lane];
      const float t03 = lds[((((po ? 3 : 0) * 2 + qo) * TN + tn) * 16 + r) * 64 + lane];
      const float y = po ? (t1 - t2 - t03) : (t03 + t1 + t2);
      const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
      const int base = tbase[i];
      if (base >= 0) outp[base + tn * 32] = y;
    }
  }
  if (prm.stamps && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    unsigned long long* d = prm.stamps + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 8;
    d[0] = t_start; d[1] = t_setup; d[2] = t_arrive; d[3] = t_staged; d[4] = t_loop; d[5] = t_xch; d[6] = t_end;
    unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); d[7] = hw;
  }
}

// naive reference: out[p][i][oh][ow][n] = sum a[p][i][oh+kh-1][ow+kw-1][c] w[kh][kw][c][n]   (flip: w[2-kh][2-kw])
__global__ void ref_conv(const float* a, long long a_ps, const float* w, long long w_ps, float* out, long long out_ps,
                         int n_img, int H, int W, int C, int N, int flip) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long tot = (long long)n_img * H * W * N;
  if (e >= tot) return;
  const int p = blockIdx.y;
  const int n = (int)(e % N);
  long long r = e / N;
  const int ow = (int)(r % W); r /= W;
  const int oh = (int)(r % H);
  const int i = (int)(r / H);
  const float* ap = a + p * a_ps;
  const float* wp = w + p * w_ps;
  double s = 0.0;
  for (int kh = 0; kh < 3; ++kh)
    for (int kw = 0; kw < 3; ++kw) {
      const int ih = oh + kh - 1, iw = ow + kw - 1;
      if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
      const int sh = flip ? 2 - kh : kh, sw = flip ? 2 - kw : kw;
      for (int c = 0; c < C; ++c) s += (double)ap[((long long)(i * H + ih) * W + iw) * C + c] * (double)wp[((long long)(sh * 3 + sw) * C + c) * N + n];
    }
  out[p * out_ps + e] = (float)s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int TN, int DIAG = 0, int VER = 1>
static float run_case(int n_img, int H, int C, int N, int P, bool per_probe_w, int flip, bool check) {
  const int W = H;
  const long long act = (long long)n_img * H * W * C, outn = (long long)n_img * H * W * N, wn = 9ll * C * N, un = 16ll * C * N;
  const int PW = per_probe_w ? P : 1;
  float *a, *w, *u, *out, *ref = nullptr;
  CK(hipMalloc(&a, act * P * 4)); CK(hipMalloc(&w, wn * PW * 4)); CK(hipMalloc(&u, un * PW * 4)); CK(hipMalloc(&out, outn * P * 4));
  {
    std::vector<float> ha((size_t)act * P), hw((size_t)wn * PW);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.f - 1.f; };
    for (auto& x : ha) x = rnd();
    for (auto& x : hw) x = rnd() * 0.1f;
    CK(hipMemcpy(a, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  }
  CK(hipMemset(out, 0xFF, outn * P * 4));
  WinoP prm;
  prm.a = a; prm.a_ps = act; prm.u = u; prm.u_ps = per_probe_w ? un : 0; prm.out = out; prm.out_ps = outn;
  prm.n_img = n_img; prm.H = H; prm.W = W; prm.C = C; prm.N = N; prm.TH = H / 2; prm.TW = W / 2;
  prm.ntiles = n_img * prm.TH * prm.TW;
  prm.a_bytes = (unsigned)(act * 4); prm.u_bytes = (unsigned)(un * 4);
  {
    int bws = 0; while ((1 << bws) < prm.TW && bws < 4) ++bws;
    int bhs = 0; while ((1 << bhs) < prm.TH && bws + bhs < 5) ++bhs;
    prm.BWs = bws; prm.BHs = bhs; prm.NI = 32 >> (bws + bhs);
    prm.FR = 2 * (1 << bhs) + 2; prm.FC = 2 * (1 << bws) + 2; prm.NS = prm.NI * prm.FR * prm.FC;
    prm.nbx = (prm.TW + (1 << bws) - 1) >> bws; prm.nby = (prm.TH + (1 << bhs) - 1) >> bhs; prm.nbi = (n_img + prm.NI - 1) / prm.NI;
    if (prm.NS > 224) { printf("footprint too large\n"); exit(1); }
  }
  const int nb = N / (32 * TN);
  const int tbs = VER == 2 ? prm.nbx * prm.nby * prm.nbi : (prm.ntiles + 31) / 32;
  const size_t xf = (size_t)8 * TN * 1024, sf = 224 * 36;
  const size_t shmem = VER == 2 ? ((xf > sf ? xf : sf) + 32) * 4 : (size_t)(4 * 2 * TN * 1024 + 32) * 4;
  if (VER == 2) CK(hipFuncSetAttribute((const void*)wino2_kernel<TN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  else CK(hipFuncSetAttribute((const void*)wino_kernel<TN, DIAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  unsigned long long* stamps = nullptr;
  const long long nblk = (long long)tbs * nb * P;
  prm.stamps = nullptr;
  if (VER == 2 && getenv("WINO_STAMPS")) { CK(hipMalloc(&stamps, nblk * 64)); CK(hipMemset(stamps, 0, nblk * 64)); prm.stamps = stamps; }
  hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  float best = 1e30f, best_t = 0.f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(wino_weight_transform, dim3((C * N + 255) / 256, PW), dim3(256), 0, 0, w, wn, u, un, C, N, flip);
    CK(hipEventRecord(e1));
    if (VER == 2) hipLaunchKernelGGL((wino2_kernel<TN>), dim3(tbs * nb, P), dim3(256), shmem, 0, prm);
    else hipLaunchKernelGGL((wino_kernel<TN, DIAG>), dim3(tbs * nb, P), dim3(256), shmem, 0, prm);
    CK(hipEventRecord(e2));
    CK(hipEventSynchronize(e2));
    float t0, t1; CK(hipEventElapsedTime(&t0, e0, e1)); CK(hipEventElapsedTime(&t1, e1, e2));
    if (t1 < best) { best = t1; best_t = t0; }
  }
  const double flops = 2.0 * n_img * H * W * 9.0 * C * N * P;
  printf("[v%d diag %d] n=%d %dx%d C=%d N=%d P=%d TN=%d w/probe=%d flip=%d: main %.3f ms (%.1f TF direct-equivalent), weight transform %.3f ms", VER, DIAG, n_img, H, W, C, N, P,
         TN, (int)per_probe_w, flip, best, flops / best / 1e9, best_t);
  if (stamps) {
    std::vector<unsigned long long> hs((size_t)nblk * 8);
    CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
    double d[6] = {0, 0, 0, 0, 0, 0}; unsigned long long tmin = ~0ull, tmax = 0;
    for (long long b = 0; b < nblk; ++b) {
      for (int q = 0; q < 6; ++q) d[q] += (double)(hs[b * 8 + q + 1] - hs[b * 8 + q]);
      if (hs[b * 8] < tmin) tmin = hs[b * 8];
      if (hs[b * 8 + 6] > tmax) tmax = hs[b * 8 + 6];
    }
    printf("\n   stamps (s_memtime ticks, mean per block): setup %.0f, first loads %.0f, LDS store+barrier %.0f, K loop %.0f, exchange %.0f, epilogue+drain %.0f; kernel span %.0f ticks = %.3f ms -> %.1f ticks/us",
           d[0] / nblk, d[1] / nblk, d[2] / nblk, d[3] / nblk, d[4] / nblk, d[5] / nblk, (double)(tmax - tmin), best, (double)(tmax - tmin) / best / 1e3);
    CK(hipFree(stamps));
  }
  if (check) {
    CK(hipMalloc(&ref, outn * P * 4));
    hipLaunchKernelGGL(ref_conv, dim3((unsigned)((outn + 255) / 256), P), dim3(256), 0, 0, a, act, w, per_probe_w ? wn : 0, ref, outn, n_img, H, W, C, N, flip);
    CK(hipDeviceSynchronize());
    std::vector<float> ho((size_t)outn * P), hr((size_t)outn * P);
    CK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hr.data(), ref, hr.size() * 4, hipMemcpyDeviceToHost));
    double md = 0, mr = 0;
    for (size_t i = 0; i < ho.size(); ++i) { md = fmax(md, fabs((double)ho[i] - hr[i])); mr = fmax(mr, fabs((double)hr[i])); }
    printf("  | max abs diff %.3e (max |ref| %.3e, rel %.2e)", md, mr, md / mr);
    CK(hipFree(ref));
  }
  printf("\n");
  fflush(stdout);
  CK(hipFree(a)); CK(hipFree(w)); CK(hipFree(u)); CK(hipFree(out));
  return best;
}

int main() {
  run_case<1, 0, 2>(2, 32, 32, 32, 2, true, 0, true);
  run_case<1, 0, 2>(3, 14, 64, 32, 2, true, 1, true);
  run_case<1, 0, 2>(50, 32, 32, 32, 256, false, 0, false);
  run_case<1, 0, 2>(50, 16, 64, 64, 256, false, 0, false);
  run_case<2, 0, 2>(50, 16, 64, 64, 256, false, 0, false);
  run_case<1, 0, 2>(50, 8, 128, 128, 256, false, 0, false);
  run_case<2, 0, 2>(50, 8, 128, 128, 256, false, 0, false);
  return 0;
}
