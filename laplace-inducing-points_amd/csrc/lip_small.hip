// Non-GEMM kernels of the linearised-network engine: all HBM-bound, coalesced along the channel
// (innermost NHWC) dimension, wavefront reductions where a sum is needed.
#include "lip_internal.h"

namespace lip {

// ---- per-channel column sums: red0[p][c] += sum_r g[p][r][c] ; red1[p][c] += sum_r g*xhat --------
// (bias / BN-parameter cotangents that could not be fused into an igemm epilogue)
// Blocks take `rpb` rows (128, fewer when the grid would not fill the chip).  Whenever N % 4 == 0 and the rows are 16-byte
// aligned a thread owns one column quad: float4 loads, private sums, four LDS atomics at the end.  (The first version
// did one LDS atomic and a 64-bit modulo per ELEMENT once N > 256: 0.3 - 0.4 TB/s on ResNet-50's 512 - 2048-channel
// layers, 4.6 ms of its 173 ms sweep.)
__global__ __launch_bounds__(256) void reduce_kernel(const ReduceP prm) {
  extern __shared__ float sm[];           // [2*N]
  const int N = prm.N, p = blockIdx.y;
  float* s0 = sm; float* s1 = sm + N;
  for (int i = threadIdx.x; i < 2 * N; i += 256) sm[i] = 0.f;
  __syncthreads();
  const int rbeg = blockIdx.x * prm.rpb;
  const int rows = min(prm.rpb, prm.R - rbeg);
  const long long seg0 = (long long)blockIdx.z * prm.R * N;      // per-example rows: segment z of R rows
  const float* g = prm.g + (long long)p * prm.g_ps + seg0 + (long long)rbeg * N;
  const float* xh = prm.xhat ? prm.xhat + seg0 + (long long)rbeg * N : nullptr;
  const long long cnt = (long long)rows * N;
  const bool quads = (N & 3) == 0 && (((uintptr_t)g | (uintptr_t)xh) & 15) == 0;
  if (quads) {
    const int nq = N >> 2;
    const int groups = nq <= 256 ? 256 / nq : 1;                 // row groups working side by side
    const int rg = nq <= 256 ? (int)threadIdx.x / nq : 0;
    for (int cq = nq <= 256 ? (int)threadIdx.x - rg * nq : (int)threadIdx.x; cq < nq && rg < groups; cq += 256) {
      float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
#pragma unroll 4
      for (int r = rg; r < rows; r += groups) {
        const float4 v = *reinterpret_cast<const float4*>(g + (long long)r * N + 4 * cq);
        a0.x += v.x; a0.y += v.y; a0.z += v.z; a0.w += v.w;
        if (xh) {
          const float4 x = *reinterpret_cast<const float4*>(xh + (long long)r * N + 4 * cq);
          a1.x += v.x * x.x; a1.y += v.y * x.y; a1.z += v.z * x.z; a1.w += v.w * x.w;
        }
      }
      atomicAdd(&s0[4 * cq], a0.x); atomicAdd(&s0[4 * cq + 1], a0.y); atomicAdd(&s0[4 * cq + 2], a0.z); atomicAdd(&s0[4 * cq + 3], a0.w);
      if (xh) { atomicAdd(&s1[4 * cq], a1.x); atomicAdd(&s1[4 * cq + 1], a1.y); atomicAdd(&s1[4 * cq + 2], a1.z); atomicAdd(&s1[4 * cq + 3], a1.w); }
    }
  } else if (N <= 256 && (256 % N) == 0) {
    // fixed channel per thread: accumulate privately, one LDS atomic per thread
    const int c = threadIdx.x % N;
    float a0 = 0.f, a1 = 0.f;
    for (long long idx = threadIdx.x; idx < cnt; idx += 256) {
      const float v = g[idx];
      a0 += v;
      if (xh) a1 += v * xh[idx];
    }
    atomicAdd(&s0[c], a0);
    if (xh) atomicAdd(&s1[c], a1);
  } else {
    for (long long idx = threadIdx.x; idx < cnt; idx += 256) {
      const int c = (int)(idx % N);
      const float v = g[idx];
      atomicAdd(&s0[c], v);
      if (xh) atomicAdd(&s1[c], v * xh[idx]);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < N; c += 256) {
    if (prm.red0) atomicAdd(prm.red0 + (long long)p * prm.red0_ps + blockIdx.z * prm.red_seg + c, s0[c]);
    if (prm.red1) atomicAdd(prm.red1 + (long long)p * prm.red1_ps + blockIdx.z * prm.red_seg + c, s1[c]);
  }
}

hipError_t launch_reduce(const ReduceP& p0, int P, hipStream_t st) {
  ReduceP p = p0;
  const long long segs = (long long)P * (p.nseg > 0 ? p.nseg : 1);
  p.rpb = 128;
  while (p.rpb > 16 && (long long)((p.R + p.rpb - 1) / p.rpb) * segs < 2048) p.rpb >>= 1;
  dim3 grid((p.R + p.rpb - 1) / p.rpb, P, p.nseg > 0 ? p.nseg : 1);
  hipLaunchKernelGGL(reduce_kernel, grid, dim3(256), 2 * p.N * sizeof(float), st, p);
  return hipGetLastError();
}

// ---- mean pool over pixels: out[p][i][c] = inv * sum_pix in[p][i][pix][c]  (jnp.mean(x,(1,2))) ------
__global__ __launch_bounds__(256) void pool_fwd_kernel(const PoolP prm) {
  extern __shared__ float sm[];           // [C]
  const int C = prm.C, i = blockIdx.x, p = blockIdx.y;
  for (int c = threadIdx.x; c < C; c += 256) sm[c] = 0.f;
  __syncthreads();
  const float* in = prm.in + (long long)p * prm.in_ps + (long long)i * prm.HW * C;
  const long long cnt = (long long)prm.HW * C;
  if ((C & 3) == 0 && ((uintptr_t)in & 15) == 0 && !(C <= 256 && (256 % C) == 0)) {
    // a thread owns whole channel quads: float4 loads down the pixels, no LDS traffic (wide layers: the per-element
    // LDS atomic + modulo of the general branch ran ResNet-50's 2048-channel pool at 0.3 TB/s)
    for (int cq = threadIdx.x; cq < (C >> 2); cq += 256) {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
      for (int px = 0; px < prm.HW; ++px) {
        const float4 v = *reinterpret_cast<const float4*>(in + (long long)px * C + 4 * cq);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
      sm[4 * cq] = a.x; sm[4 * cq + 1] = a.y; sm[4 * cq + 2] = a.z; sm[4 * cq + 3] = a.w;
    }
  } else if (C <= 256 && (256 % C) == 0) {
    const int c = threadIdx.x % C;
    float a = 0.f;
    for (long long idx = threadIdx.x; idx < cnt; idx += 256) a += in[idx];
    atomicAdd(&sm[c], a);
  } else {
    for (long long idx = threadIdx.x; idx < cnt; idx += 256) atomicAdd(&sm[idx % C], in[idx]);
  }
  __syncthreads();
  float* out = prm.out + (long long)p * prm.out_ps + (long long)i * C;
  for (int c = threadIdx.x; c < C; c += 256) out[c] = sm[c] * prm.inv;
}

hipError_t launch_pool_fwd(const PoolP& p, int P, hipStream_t st) {
  hipLaunchKernelGGL(pool_fwd_kernel, dim3(p.n, P, 1), dim3(256), p.C * sizeof(float), st, p);
  return hipGetLastError();
}

// ---- pool backward: out[p][i][pix][c] = dphi[i][pix][c] * inv * in[p][i][c], plus reductions ------------
constexpr int PB_PIX = 16;
__global__ __launch_bounds__(256) void pool_bwd_kernel(const PoolP prm) {
  extern __shared__ float sm[];           // [2*C]
  const int C = prm.C, i = blockIdx.y, p = blockIdx.z;
  float* s0 = sm; float* s1 = sm + C;
  const bool red = prm.red0 || prm.red1;
  if (red) {
    for (int c = threadIdx.x; c < 2 * C; c += 256) sm[c] = 0.f;
    __syncthreads();
  }
  const int pbeg = blockIdx.x * PB_PIX;
  const int npix = min(PB_PIX, prm.HW - pbeg);
  const float* in = prm.in + (long long)p * prm.in_ps + (long long)i * C;
  const long long base = ((long long)i * prm.HW + pbeg) * C;
  float* out = prm.out + (long long)p * prm.out_ps + base;
  const long long cnt = (long long)npix * C;
  // 256 % C == 0: a thread keeps ONE channel for all its pixels, so the column sums accumulate in registers and
  // reach LDS once per thread (an LDS atomic per element made this broadcast kernel run at 0.4 TB/s)
  const bool fixed_c = (256 % C) == 0;
  float a0 = 0.f, a1 = 0.f;
  if (!fixed_c && (C & 3) == 0 && (((uintptr_t)in | (uintptr_t)out | (uintptr_t)(prm.dphi ? prm.dphi + base : nullptr) |
                                    (uintptr_t)(prm.xhat ? prm.xhat + base : nullptr)) & 15) == 0) {
    // wide layers: a thread owns whole channel quads for all the block's pixels; sums stay in registers and go to the
    // per-probe totals directly (no LDS, no per-element atomics)
    for (int cq = threadIdx.x; cq < (C >> 2); cq += 256) {
      const float4 u = *reinterpret_cast<const float4*>(in + 4 * cq);
      float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
      for (int px = 0; px < npix; ++px) {
        const long long o = (long long)px * C + 4 * cq;
        float4 v = make_float4(u.x * prm.inv, u.y * prm.inv, u.z * prm.inv, u.w * prm.inv);
        if (prm.dphi) {
          const float4 d = *reinterpret_cast<const float4*>(prm.dphi + base + o);
          v.x *= d.x; v.y *= d.y; v.z *= d.z; v.w *= d.w;
        }
        *reinterpret_cast<float4*>(out + o) = v;
        if (red) {
          r0.x += v.x; r0.y += v.y; r0.z += v.z; r0.w += v.w;
          if (prm.red1) {
            const float4 x = *reinterpret_cast<const float4*>(prm.xhat + base + o);
            r1.x += v.x * x.x; r1.y += v.y * x.y; r1.z += v.z * x.z; r1.w += v.w * x.w;
          }
        }
      }
      if (prm.red0) { float* d0 = prm.red0 + (long long)p * prm.red0_ps + 4 * cq;
        atomicAdd(d0, r0.x); atomicAdd(d0 + 1, r0.y); atomicAdd(d0 + 2, r0.z); atomicAdd(d0 + 3, r0.w); }
      if (prm.red1) { float* d1 = prm.red1 + (long long)p * prm.red1_ps + 4 * cq;
        atomicAdd(d1, r1.x); atomicAdd(d1 + 1, r1.y); atomicAdd(d1 + 2, r1.z); atomicAdd(d1 + 3, r1.w); }
    }
    return;
  }
  for (long long idx = threadIdx.x; idx < cnt; idx += 256) {
    const int c = (int)(idx % C);
    float v = in[c] * prm.inv;
    if (prm.dphi) v *= prm.dphi[base + idx];
    out[idx] = v;
    if (red) {
      const float x = prm.red1 ? v * prm.xhat[base + idx] : 0.f;
      if (fixed_c) { a0 += v; a1 += x; }
      else {
        atomicAdd(&s0[c], v);
        if (prm.red1) atomicAdd(&s1[c], x);
      }
    }
  }
  if (red && fixed_c && threadIdx.x < cnt) {
    const int c = threadIdx.x % C;
    atomicAdd(&s0[c], a0);
    if (prm.red1) atomicAdd(&s1[c], a1);
  }
  if (red) {
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      if (prm.red0) atomicAdd(prm.red0 + (long long)p * prm.red0_ps + c, s0[c]);
      if (prm.red1) atomicAdd(prm.red1 + (long long)p * prm.red1_ps + c, s1[c]);
    }
  }
}

hipError_t launch_pool_bwd(const PoolP& p, int P, hipStream_t st) {
  dim3 grid((p.HW + PB_PIX - 1) / PB_PIX, p.n, P);
  hipLaunchKernelGGL(pool_bwd_kernel, grid, dim3(256), 2 * p.C * sizeof(float), st, p);
  return hipGetLastError();
}

// ---- window pools (max / average) ----------------------------------------------------------------------------
// A null argmax buffer selects the window AVERAGE (sum / (KH*KW), padding counted: flax.linen.avg_pool's default).
// primal: out = max over the window, argmax cached as the linear pixel index of the winning input (first max wins)
__global__ __launch_bounds__(256) void maxpool_primal_kernel(const MaxPoolP prm) {
  const long long total = (long long)prm.n * prm.OH * prm.OW * prm.C;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % prm.C);
    long long t = idx / prm.C;
    const int ow = (int)(t % prm.OW); t /= prm.OW;
    const int oh = (int)(t % prm.OH);
    const int i = (int)(t / prm.OH);
    const bool avg = prm.amax_w == nullptr;
    float best = avg ? 0.f : -3.0e38f; int bi = -1;
    for (int kh = 0; kh < prm.KH; ++kh) {
      const int ih = oh * prm.stride + kh - prm.pad_h;
      if (ih < 0 || ih >= prm.IH) continue;
      for (int kw = 0; kw < prm.KW; ++kw) {
        const int iw = ow * prm.stride + kw - prm.pad_w;
        if (iw < 0 || iw >= prm.IW) continue;
        const float v = prm.in[(((long long)i * prm.IH + ih) * prm.IW + iw) * prm.C + c];
        if (avg) best += v;
        else if (v > best) { best = v; bi = ih * prm.IW + iw; }
      }
    }
    if (avg) { prm.out[idx] = best * (1.f / (float)(prm.KH * prm.KW)); continue; }
    prm.out[idx] = best;
    prm.amax_w[idx] = (float)bi;
  }
}

hipError_t launch_maxpool_primal(const MaxPoolP& p, hipStream_t st) {
  const long long total = (long long)p.n * p.OH * p.OW * p.C;
  const long long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(maxpool_primal_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, st, p);
  return hipGetLastError();
}

// tangent: out[p][i][oh][ow][c] = in[p][i][argmax][c]
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const MaxPoolP prm) {
  const int p = blockIdx.y;
  const long long per_img_out = (long long)prm.OH * prm.OW * prm.C, per_img_in = (long long)prm.IH * prm.IW * prm.C;
  const long long total = prm.n * per_img_out;
  const float* in = prm.in + (long long)p * prm.in_ps;
  float* out = prm.out + (long long)p * prm.out_ps;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % prm.C);
    const long long i = idx / per_img_out;
    if (!prm.amax) {
      long long t = (idx % per_img_out) / prm.C;
      const int ow = (int)(t % prm.OW), oh = (int)(t / prm.OW);
      float acc = 0.f;
      for (int kh = 0; kh < prm.KH; ++kh) {
        const int ih = oh * prm.stride + kh - prm.pad_h;
        if (ih < 0 || ih >= prm.IH) continue;
        for (int kw = 0; kw < prm.KW; ++kw) {
          const int iw = ow * prm.stride + kw - prm.pad_w;
          if (iw < 0 || iw >= prm.IW) continue;
          acc += in[i * per_img_in + ((long long)ih * prm.IW + iw) * prm.C + c];
        }
      }
      out[idx] = acc * (1.f / (float)(prm.KH * prm.KW));
      continue;
    }
    const int pix = (int)prm.amax[idx];
    out[idx] = pix >= 0 ? in[i * per_img_in + (long long)pix * prm.C + c] : 0.f;
  }
}

hipError_t launch_maxpool_fwd(const MaxPoolP& p, int P, hipStream_t st) {
  const long long total = (long long)p.n * p.OH * p.OW * p.C;
  const long long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096), P), dim3(256), 0, st, p);
  return hipGetLastError();
}

// cotangent: out[p][i][ih][iw][c] = dphi * sum over windows (oh, ow) covering (ih, iw) whose argmax is (ih, iw)
// of in[p][i][oh][ow][c]; plus the reductions of LIP_OP_POOL_BWD.  A gather, so no atomics on the tensor.
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const MaxPoolP prm) {
  extern __shared__ float sm[];           // [2*C]
  const int C = prm.C, p = blockIdx.y;
  float* s0 = sm; float* s1 = sm + C;
  const bool red = prm.red0 || prm.red1;
  if (red) {
    for (int c = threadIdx.x; c < 2 * C; c += 256) sm[c] = 0.f;
    __syncthreads();
  }
  const long long per_img_in = (long long)prm.IH * prm.IW * C, per_img_g = (long long)prm.OH * prm.OW * C;
  const long long total = prm.n * per_img_in;
  const float* g = prm.in + (long long)p * prm.in_ps;
  float* out = prm.out + (long long)p * prm.out_ps;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % C);
    long long t = idx / C;
    const int iw = (int)(t % prm.IW); t /= prm.IW;
    const int ih = (int)(t % prm.IH);
    const long long i = t / prm.IH;
    const float me = (float)(ih * prm.IW + iw);
    float v = 0.f;
    for (int kh = 0; kh < prm.KH; ++kh) {
      const int th = ih + prm.pad_h - kh;
      if (th < 0 || (th % prm.stride) != 0) continue;
      const int oh = th / prm.stride;
      if (oh >= prm.OH) continue;
      for (int kw = 0; kw < prm.KW; ++kw) {
        const int tw = iw + prm.pad_w - kw;
        if (tw < 0 || (tw % prm.stride) != 0) continue;
        const int ow = tw / prm.stride;
        if (ow >= prm.OW) continue;
        const long long o = i * per_img_g + ((long long)oh * prm.OW + ow) * C + c;
        if (!prm.amax || prm.amax[o] == me) v += g[o];
      }
    }
    if (!prm.amax) v *= 1.f / (float)(prm.KH * prm.KW);
    if (prm.dphi) v *= prm.dphi[idx];
    out[idx] = v;
    if (red) {
      atomicAdd(&s0[c], v);
      if (prm.red1) atomicAdd(&s1[c], v * prm.xhat[idx]);
    }
  }
  if (red) {
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      if (prm.red0) atomicAdd(prm.red0 + (long long)p * prm.red0_ps + c, s0[c]);
      if (prm.red1) atomicAdd(prm.red1 + (long long)p * prm.red1_ps + c, s1[c]);
    }
  }
}

// Same gather, one channel QUAD per thread (C = 4, 8, ..., 1024: the grid stride is a multiple of C / 4, so a thread
// keeps its channels): 32-bit index arithmetic once per quad, float4 loads / stores, the reduction sums private in
// registers.  The scalar kernel above spends a 64-bit div / mod chain and two LDS atomics per element: 4.2 ms for
// ResNet-50's stem pool at 8 images x 64 probes (0.5 TB/s); it stays for the other channel counts.
__global__ __launch_bounds__(256) void maxpool_bwd_quad_kernel(const MaxPoolP prm) {
  extern __shared__ float sm[];           // [2*C]
  const int C = prm.C, nq = C >> 2, p = blockIdx.y;
  float* s0 = sm; float* s1 = sm + C;
  const bool red = prm.red0 || prm.red1;
  if (red) {
    for (int c = threadIdx.x; c < 2 * C; c += 256) sm[c] = 0.f;
    __syncthreads();
  }
  const unsigned total = (unsigned)prm.n * prm.IH * prm.IW * nq;
  const unsigned per_img_g = (unsigned)prm.OH * prm.OW * C;
  const float* g = prm.in + (long long)p * prm.in_ps;
  float* out = prm.out + (long long)p * prm.out_ps;
  const unsigned first = blockIdx.x * 256u + threadIdx.x;
  const int c = (int)(first % (unsigned)nq) * 4;
  float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
  const float scale = prm.amax ? 1.f : 1.f / (float)(prm.KH * prm.KW);
  for (unsigned q = first; q < total; q += gridDim.x * 256u) {
    unsigned t = q / (unsigned)nq;
    const int iw = (int)(t % (unsigned)prm.IW); t /= (unsigned)prm.IW;
    const int ih = (int)(t % (unsigned)prm.IH);
    const unsigned i = t / (unsigned)prm.IH;
    const float me = (float)(ih * prm.IW + iw);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int kh = 0; kh < prm.KH; ++kh) {
      const int th = ih + prm.pad_h - kh;
      if (th < 0 || (th % prm.stride) != 0) continue;
      const int oh = th / prm.stride;
      if (oh >= prm.OH) continue;
      for (int kw = 0; kw < prm.KW; ++kw) {
        const int tw = iw + prm.pad_w - kw;
        if (tw < 0 || (tw % prm.stride) != 0) continue;
        const int ow = tw / prm.stride;
        if (ow >= prm.OW) continue;
        const unsigned o = i * per_img_g + (unsigned)(oh * prm.OW + ow) * C + c;
        const float4 gv = *reinterpret_cast<const float4*>(g + o);
        if (prm.amax) {
          const float4 am = *reinterpret_cast<const float4*>(prm.amax + o);
          v.x += am.x == me ? gv.x : 0.f; v.y += am.y == me ? gv.y : 0.f;
          v.z += am.z == me ? gv.z : 0.f; v.w += am.w == me ? gv.w : 0.f;
        } else {
          v.x += gv.x; v.y += gv.y; v.z += gv.z; v.w += gv.w;
        }
      }
    }
    v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
    const long long idx = (long long)q * 4;
    if (prm.dphi) {
      const float4 d = *reinterpret_cast<const float4*>(prm.dphi + idx);
      v.x *= d.x; v.y *= d.y; v.z *= d.z; v.w *= d.w;
    }
    *reinterpret_cast<float4*>(out + idx) = v;
    if (red) {
      r0.x += v.x; r0.y += v.y; r0.z += v.z; r0.w += v.w;
      if (prm.red1) {
        const float4 x = *reinterpret_cast<const float4*>(prm.xhat + idx);
        r1.x += v.x * x.x; r1.y += v.y * x.y; r1.z += v.z * x.z; r1.w += v.w * x.w;
      }
    }
  }
  if (red) {
    atomicAdd(&s0[c], r0.x); atomicAdd(&s0[c + 1], r0.y); atomicAdd(&s0[c + 2], r0.z); atomicAdd(&s0[c + 3], r0.w);
    if (prm.red1) { atomicAdd(&s1[c], r1.x); atomicAdd(&s1[c + 1], r1.y); atomicAdd(&s1[c + 2], r1.z); atomicAdd(&s1[c + 3], r1.w); }
    __syncthreads();
    for (int cc = threadIdx.x; cc < C; cc += 256) {
      if (prm.red0) atomicAdd(prm.red0 + (long long)p * prm.red0_ps + cc, s0[cc]);
      if (prm.red1) atomicAdd(prm.red1 + (long long)p * prm.red1_ps + cc, s1[cc]);
    }
  }
}

hipError_t launch_maxpool_bwd(const MaxPoolP& p, int P, hipStream_t st) {
  const long long total = (long long)p.n * p.IH * p.IW * p.C;
  const int nq = p.C >> 2;
  auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  const bool quad = (p.C & 3) == 0 && nq <= 256 && 256 % nq == 0 && total < (1ll << 31) && (long long)p.n * p.OH * p.OW * p.C < (1ll << 31) &&
                    al16(p.in) && al16(p.out) && al16(p.amax) && al16(p.dphi) && al16(p.xhat) && (p.in_ps & 3) == 0 && (p.out_ps & 3) == 0;
  if (quad) {
    // few, long-lived blocks per probe: every block ends in 2 C global atomics on the same per-probe sums (4 096 blocks
    // per probe: 33 M contended atomics, 3.97 ms; 128 per probe: 0.4 ms)
    const long long blocks = (total / 4 + 255) / 256, want = 8192 / (P > 0 ? P : 1) > 8 ? 8192 / (P > 0 ? P : 1) : 8;
    hipLaunchKernelGGL(maxpool_bwd_quad_kernel, dim3((unsigned)(blocks < want ? blocks : want), P), dim3(256),
                       2 * p.C * sizeof(float), st, p);
    return hipGetLastError();
  }
  const long long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048), P), dim3(256),
                     2 * p.C * sizeof(float), st, p);
  return hipGetLastError();
}

// ---- primal post-processing: z -> (xhat, a = act(y), dphi = act'(y)),  y = BN(z + bias) + res --------
__device__ __forceinline__ void act_eval(int act, float y, float& a, float& d) {
  if (act == 1) { a = y > 0.f ? y : 0.f; d = y > 0.f ? 1.f : 0.f; }
  else if (act == 2) { const float t = tanhf(y); a = t; d = 1.f - t * t; }
  else if (act == 3) {  // tanh-form GELU (flax.linen.gelu default approximate=True; src/toymodels.py:11)
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float u = k0 * (y + k1 * y * y * y);
    const float t = tanhf(u);
    a = 0.5f * y * (1.f + t);
    d = 0.5f * (1.f + t) + 0.5f * y * (1.f - t * t) * k0 * (1.f + 3.f * k1 * y * y);
  } else { a = y; d = 1.f; }
}

__global__ __launch_bounds__(256) void primal_post_kernel(const PrimalPostP prm) {
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < prm.count;
       idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % prm.N);
    float y = prm.z[idx];
    if (prm.bias) y += prm.bias[c];
    if (prm.gamma) {
      const float xh = (y - prm.mean[c]) * prm.rstd[c];
      if (prm.xhat) prm.xhat[idx] = xh;
      y = xh * prm.gamma[c] + prm.beta[c];
    }
    if (prm.res) y += prm.res[idx];
    float a, d;
    act_eval(prm.act, y, a, d);
    prm.a[idx] = a;
    if (prm.dphi) prm.dphi[idx] = d;
  }
}

hipError_t launch_primal_post(const PrimalPostP& p, hipStream_t st) {
  const long long blocks = (p.count + 255) / 256;
  hipLaunchKernelGGL(primal_post_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, p);
  return hipGetLastError();
}

// ---- softmax of the primal logits (one wave per example): p and sqrt(p)  (src/ggn.py:23-24,127) ------
__global__ __launch_bounds__(64) void softmax_kernel(const float* logits, float* prob, float* sqrtp, int n, int K) {
  const int i = blockIdx.x, lane = threadIdx.x;
  const float* f = logits + (long long)i * K;
  float mx = -3.0e38f;
  for (int k = lane; k < K; k += 64) mx = fmaxf(mx, f[k]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s += expf(f[k] - mx);
  s = wave_sum(s);
  const float inv = 1.f / s;
  for (int k = lane; k < K; k += 64) {
    const float pk = expf(f[k] - mx) * inv;
    prob[(long long)i * K + k] = pk;
    sqrtp[(long long)i * K + k] = sqrtf(pk);
  }
}

hipError_t launch_softmax(const float* logits, float* prob, float* sqrtp, int n, int K, hipStream_t st) {
  hipLaunchKernelGGL(softmax_kernel, dim3(n), dim3(64), 0, st, logits, prob, sqrtp, n, K);
  return hipGetLastError();
}

// ---- output-space head: Hessian / square-root-factor action per (probe, example) ----------------------
//   GGN : g = c (p.u - p (p^T u))                 src/ggn.py:125-131
//   LT  : U = c (s.u - (p^T u) s)                 src/ggn.py:29-39
//   L   : g = c (s.u - (s^T u) p)                 src/ggn.py:16-27
//   regressor: every mode multiplies by c (c carries exp(-logvar) or its square root, src/ggn.py:17-19,112-113)
__global__ __launch_bounds__(64) void head_kernel(const HeadP prm) {
  const int i = blockIdx.x, p = blockIdx.y, lane = threadIdx.x, K = prm.K;
  const float* u = prm.in + (long long)p * prm.in_ps + (long long)i * K;
  float* o = prm.out + (long long)p * prm.out_ps + (long long)i * K;
  if (!prm.classifier || prm.mode == LIP_HEAD_OUT || prm.mode == LIP_HEAD_IN) {
    for (int k = lane; k < K; k += 64) o[k] = prm.c * u[k];
    return;
  }
  const float* pp = prm.p + (long long)i * K;
  const float* ss = prm.s + (long long)i * K;
  float dot = 0.f;
  if (prm.mode == LIP_HEAD_L) { for (int k = lane; k < K; k += 64) dot += ss[k] * u[k]; }
  else                        { for (int k = lane; k < K; k += 64) dot += pp[k] * u[k]; }
  dot = wave_sum(dot);
  for (int k = lane; k < K; k += 64) {
    float v;
    if (prm.mode == LIP_HEAD_GGN)      v = pp[k] * (u[k] - dot);
    else if (prm.mode == LIP_HEAD_LT)  v = ss[k] * (u[k] - dot);
    else                               v = ss[k] * u[k] - dot * pp[k];
    o[k] = prm.c * v;
  }
}

hipError_t launch_head(const HeadP& p, int P, hipStream_t st) {
  hipLaunchKernelGGL(head_kernel, dim3(p.n, P, 1), dim3(64), 0, st, p);
  return hipGetLastError();
}

// ---- y = a * x (initialises the output block with the prior-precision term alpha*V, src/lla.py:21-22) --
__global__ __launch_bounds__(256) void scale_copy_kernel(float* y, const float* x, float a, long long count) {
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < count; idx += (long long)gridDim.x * 256)
    y[idx] = a * x[idx];
}

__global__ void scale_copy_range_kernel(float* __restrict__ y, const float* __restrict__ x, float a, long long off, long long len,
                                        long long ld) {
  const long long base = (long long)blockIdx.y * ld + off;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (long long)gridDim.x * blockDim.x)
    y[base + i] = x ? a * x[base + i] : 0.f;
}

hipError_t launch_scale_copy_range(float* y, const float* x, float a, long long off, long long len, int P, long long ld, hipStream_t st) {
  if (len <= 0 || P <= 0) return hipSuccess;
  const long long blocks = (len + 255) / 256;
  hipLaunchKernelGGL(scale_copy_range_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024), (unsigned)P), dim3(256), 0, st, y, x, a, off, len, ld);
  return hipGetLastError();
}

hipError_t launch_scale_copy(float* y, const float* x, float a, long long count, hipStream_t st) {
  const long long blocks = (count + 255) / 256;
  hipLaunchKernelGGL(scale_copy_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, st, y, x, a, count);
  return hipGetLastError();
}

}  // namespace lip
