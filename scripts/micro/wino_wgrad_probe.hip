// Winograd F(2x2, 3x3) WEIGHT GRADIENT on the f32 matrix pipe — feasibility probe (3x3 / stride 1 / pad 1 layers).
//   hipcc --offload-arch=gfx950 -O3 wino_wgrad_probe.hip -o wino_wgrad_probe && ./wino_wgrad_probe
// dW = G^T [ sum_tiles (B^T d B) (.) (A dY A^T) ] G : 16 positions xi, each a GEMM with the reduction over TILES:
//   dU_xi[c][n] = sum_t V_xi[t][c] Gh_xi[t][n]   — 4 multiplications per output pixel and (c, n) instead of 9.
//   * V = B^T x B of the PRIMAL activations is shared by all probes: transformed once per launch into
//     Vt[xi][t/4][c][4] so that lane (c, half h) reads four consecutive tiles of its channel as ONE dwordx4 — the A
//     registers of 4 k-steps (k-step j multiplies tile 8m + 4h + j);
//   * Gh = A g A^T of the per-probe cotangent is formed on the fly: lane (n, h) loads the 2 x 2 pixels of its tile
//     (dwords, 128 contiguous bytes per half-wave), 2 + 3 VALU ops -> the B registers of its wave's four positions;
//   * wave a owns row a of the 4x4 position grid (64 accumulator registers); dW = G^T dU G in-wave over b, across the
//     waves over a through LDS.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WgP {
  const float* vt;                    // [16][TQ][C][4]
  const float* g; long long g_ps;     // [P][n][H][W][N]
  float* y; long long y_ps;           // [P][9][C][N]
  int n_img, H, W, C, N, TH, TW, T, TQ, S, groups_per_split;
  unsigned vt_bytes, g_bytes;
};

// thread (tq, c): the 4x4 patches of tiles 4 tq .. 4 tq + 3, channel c -> 16 float4
__global__ __launch_bounds__(256) void wino_input_transform(const float* __restrict__ x, float* __restrict__ vt, int n_img, int H, int W, int C,
                                                             int TH, int TW, int T, int TQ) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)TQ * C) return;
  const int c = (int)(e % C), tq = (int)(e / C);
  float v[16][4];
  for (int j = 0; j < 4; ++j) {
    const int t = 4 * tq + j;
    float d[4][4];
    const int img = t / (TH * TW), rem = t - img * TH * TW, ty = rem / TW, tx = rem - ty * TW;
    for (int r = 0; r < 4; ++r)
      for (int q = 0; q < 4; ++q) {
        const int ih = 2 * ty - 1 + r, iw = 2 * tx - 1 + q;
        d[r][q] = (t < T && ih >= 0 && ih < H && iw >= 0 && iw < W) ? x[((long long)(img * H + ih) * W + iw) * C + c] : 0.f;
      }
    float e4[4][4];
    for (int q = 0; q < 4; ++q) {
      e4[0][q] = d[0][q] - d[2][q]; e4[1][q] = d[1][q] + d[2][q]; e4[2][q] = d[2][q] - d[1][q]; e4[3][q] = d[1][q] - d[3][q];
    }
    for (int a = 0; a < 4; ++a) {
      v[4 * a + 0][j] = e4[a][0] - e4[a][2]; v[4 * a + 1][j] = e4[a][1] + e4[a][2];
      v[4 * a + 2][j] = e4[a][2] - e4[a][1]; v[4 * a + 3][j] = e4[a][1] - e4[a][3];
    }
  }
  for (int xi = 0; xi < 16; ++xi)
    *reinterpret_cast<f32x4*>(&vt[(((long long)xi * TQ + tq) * C + c) * 4]) = f32x4{v[xi][0], v[xi][1], v[xi][2], v[xi][3]};
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void wino_wgrad_kernel(const WgP prm) {
  extern __shared__ __attribute__((aligned(16))) float lds[];          // [4 a][3 kw][16 reg][64 lane]
  const int tid = threadIdx.x, lane = tid & 63;
  const int a = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int C = prm.C, N = prm.N, W = prm.W, H = prm.H;
  const int ctn = C >> 5, ntn = N >> 5;
  int b = blockIdx.x;
  const int nt = b % ntn; b /= ntn;
  const int ct = b % ctn; b /= ctn;
  const int z = b;                                  // K split
  const int p = blockIdx.y;
  const int c = ct * 32 + l31, n = nt * 32 + l31;
  const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(prm.vt), 0, prm.vt_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(prm.g + (long long)p * prm.g_ps), 0, prm.g_bytes, 0x00020000);
  // rows of the 2x2 cotangent tile and their coefficients:  x_q = k0 g[0][q] + k1 g[1][q]   (A rows: g0, g0+g1, g0-g1, -g1)
  const float k0 = (a == 3) ? 0.f : 1.f;
  const float k1 = (a == 0) ? 0.f : (a == 1 ? 1.f : -1.f);
  const unsigned avoff = (unsigned)((h * C + c) * 16);
  const unsigned a_xi = (unsigned)prm.TQ * C * 16;            // bytes per position plane
  const unsigned row_bytes = (unsigned)(W * N * 4);
  const int tpi = prm.TH * prm.TW;

  f32x16 acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  const int m0 = z * prm.groups_per_split;
  int m1 = m0 + prm.groups_per_split;
  if (m1 > prm.TQ / 2) m1 = prm.TQ / 2;
  f32x4 areg[2][4];
  float graw[2][4][4];                 // [buffer][tile j][r * 2 + q]
  auto load_group = [&](int m, f32x4 (&ar)[4], float (&gr)[4][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      ar[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(vrs, avoff, (4 * a + q) * a_xi + (unsigned)(2 * m) * C * 16, 0));
    const int t = 8 * m + 4 * h;
    const int img = t / tpi, rem = t - img * tpi, ty = rem / prm.TW, tx = rem - ty * prm.TW;
    const unsigned gv = t < prm.T ? (unsigned)((((img * H + 2 * ty) * W + 2 * tx) * N + n) * 4) : 0x80000000u;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          gr[j][2 * r + q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grs, gv + (unsigned)((2 * j + q) * N * 4), r * row_bytes, 0));
  };
  auto compute = [&](const f32x4 (&ar)[4], const float (&gr)[4][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x0 = k0 * gr[j][0] + k1 * gr[j][2], x1 = k0 * gr[j][1] + k1 * gr[j][3];
      const float bv[4] = {x0, x0 + x1, x0 - x1, -x1};
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[q][j], bv[q], acc[q], 0, 0, 0);
    }
  };
  // (groups_per_split is even: two groups per iteration, the last iteration re-requests the final group)
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
    lds[((a * 3 + 0) * 16 + r) * 64 + lane] = u0 + 0.5f * (u1 + u2);
    lds[((a * 3 + 1) * 16 + r) * 64 + lane] = 0.5f * (u1 - u2);
    lds[((a * 3 + 2) * 16 + r) * 64 + lane] = 0.5f * (u1 + u2) + u3;
  }
  __syncthreads();
  float* yp = prm.y + (long long)p * prm.y_ps;
  for (int o = a; o < 9; o += 4) {
    const int kh = o / 3, kw = o - 3 * kh;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float x0 = lds[((0 * 3 + kw) * 16 + r) * 64 + lane], x1 = lds[((1 * 3 + kw) * 16 + r) * 64 + lane];
      const float x2 = lds[((2 * 3 + kw) * 16 + r) * 64 + lane], x3 = lds[((3 * 3 + kw) * 16 + r) * 64 + lane];
      const float w = kh == 0 ? (x0 + 0.5f * (x1 + x2)) : (kh == 1 ? 0.5f * (x1 - x2) : (0.5f * (x1 + x2) + x3));
      const int ci = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      float* dst = yp + ((long long)(o * C + ci)) * N + n;
      if (prm.S > 1) atomicAdd(dst, w); else *dst = w;
    }
  }
}

// naive reference: dW[p][kh][kw][c][n] = sum_{i,oh,ow} x[i][oh+kh-1][ow+kw-1][c] g[p][i][oh][ow][n]
__global__ void ref_wgrad(const float* x, const float* g, long long g_ps, float* y, long long y_ps, int n_img, int H, int W, int C, int N) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 9 * C * N) return;
  const int p = blockIdx.y;
  const int n = e % N, c = (e / N) % C, o = e / (N * C);
  const int kh = o / 3, kw = o % 3;
  double s = 0.0;
  for (int i = 0; i < n_img; ++i)
    for (int oh = 0; oh < H; ++oh)
      for (int ow = 0; ow < W; ++ow) {
        const int ih = oh + kh - 1, iw = ow + kw - 1;
        if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
        s += (double)x[((long long)(i * H + ih) * W + iw) * C + c] * (double)g[p * g_ps + ((long long)(i * H + oh) * W + ow) * N + n];
      }
  y[p * y_ps + e] = (float)s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static void run_case(int n_img, int H, int C, int N, int P, int S, bool check) {
  const int W = H;
  const long long act = (long long)n_img * H * W * C, gn = (long long)n_img * H * W * N, yn = 9ll * C * N;
  WgP prm;
  prm.n_img = n_img; prm.H = H; prm.W = W; prm.C = C; prm.N = N; prm.TH = H / 2; prm.TW = W / 2;
  prm.T = n_img * prm.TH * prm.TW;
  const int groups = (prm.T + 7) / 8;
  int gps = (groups + S - 1) / S; gps += gps & 1;             // even
  prm.S = S; prm.groups_per_split = gps;
  prm.TQ = 2 * gps * S;                                        // padded: every split reads whole groups
  float *x, *g, *vt, *y, *ref = nullptr;
  CK(hipMalloc(&x, act * 4)); CK(hipMalloc(&g, gn * P * 4)); CK(hipMalloc(&vt, 16ll * prm.TQ * C * 4 * 4)); CK(hipMalloc(&y, yn * P * 4));
  {
    std::vector<float> hx((size_t)act), hg((size_t)gn * P);
    unsigned s = 777u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.f - 1.f; };
    for (auto& v : hx) v = rnd();
    for (auto& v : hg) v = rnd();
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(g, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
  }
  prm.vt = vt; prm.g = g; prm.g_ps = gn; prm.y = y; prm.y_ps = yn;
  prm.vt_bytes = (unsigned)(16ll * prm.TQ * C * 16); prm.g_bytes = (unsigned)(gn * 4);
  const size_t shmem = 12 * 1024 * 4;
  CK(hipFuncSetAttribute((const void*)wino_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  float best = 1e30f, best_t = 0.f;
  for (int rep = 0; rep < 4; ++rep) {
    if (S > 1) CK(hipMemsetAsync(y, 0, yn * P * 4, 0));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(wino_input_transform, dim3((unsigned)(((long long)prm.TQ * C + 255) / 256)), dim3(256), 0, 0, x, vt, n_img, H, W, C, prm.TH, prm.TW, prm.T, prm.TQ);
    CK(hipEventRecord(e1));
    hipLaunchKernelGGL(wino_wgrad_kernel, dim3((C / 32) * (N / 32) * S, P), dim3(256), shmem, 0, prm);
    CK(hipEventRecord(e2));
    CK(hipEventSynchronize(e2));
    float t0, t1; CK(hipEventElapsedTime(&t0, e0, e1)); CK(hipEventElapsedTime(&t1, e1, e2));
    if (t1 < best) { best = t1; best_t = t0; }
  }
  const double flops = 2.0 * n_img * H * W * 9.0 * C * N * P;
  printf("n=%d %dx%d C=%d N=%d P=%d S=%d: main %.3f ms (%.1f TF direct-equivalent), input transform %.3f ms", n_img, H, W, C, N, P, S, best,
         flops / best / 1e9, best_t);
  if (check) {
    CK(hipMalloc(&ref, yn * P * 4));
    hipLaunchKernelGGL(ref_wgrad, dim3((unsigned)((yn + 255) / 256), P), dim3(256), 0, 0, x, g, gn, ref, yn, n_img, H, W, C, N);
    CK(hipDeviceSynchronize());
    std::vector<float> ho((size_t)yn * P), hr((size_t)yn * P);
    CK(hipMemcpy(ho.data(), y, ho.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hr.data(), ref, hr.size() * 4, hipMemcpyDeviceToHost));
    double md = 0, mr = 0;
    for (size_t i = 0; i < ho.size(); ++i) { md = fmax(md, fabs((double)ho[i] - hr[i])); mr = fmax(mr, fabs((double)hr[i])); }
    printf("  | max abs diff %.3e (max |ref| %.3e, rel %.2e)", md, mr, md / mr);
    CK(hipFree(ref));
  }
  printf("\n");
  fflush(stdout);
  CK(hipFree(x)); CK(hipFree(g)); CK(hipFree(vt)); CK(hipFree(y));
}

int main() {
  run_case(3, 8, 32, 32, 2, 1, true);
  run_case(3, 8, 32, 64, 2, 2, true);
  run_case(5, 16, 64, 32, 3, 3, true);
  // direct kernels at P = 256: 288 x 32: 2.12 ms, 576 x 64: 2.13 ms, 1152 x 128: 2.03 ms
  run_case(50, 32, 32, 32, 256, 4, false);
  run_case(50, 32, 32, 32, 256, 8, false);
  run_case(50, 16, 64, 64, 256, 1, false);
  run_case(50, 16, 64, 64, 256, 2, false);
  run_case(50, 8, 128, 128, 256, 1, false);
  return 0;
}
