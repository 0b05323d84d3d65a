// Pure MFMA streams on gfx950: how fast does v_mfma_f32_32x32x2_f32 issue when a wave's MFMAs form ONE dependent
// accumulator chain, or 2 / 4 independent chains?   hipcc --offload-arch=gfx950 -O3 mfma_stream.hip -o mfma_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH>
__global__ __launch_bounds__(256) void stream(float* out, int iters, float a0, float b0) {
  f32x16 acc[CH];
  for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16 / CH; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CH>
static void run(int blocks_per_cu) {
  const int blocks = 256 * blocks_per_cu, iters = 4000;
  float* out; hipMalloc(&out, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((stream<CH>), dim3(blocks), dim3(256), 0, 0, out, 10, 1.f, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((stream<CH>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 /*waves*/ * iters * 16 * 4096.0;
  printf("chains/wave %d, waves/SIMD %d: %.1f TFLOP/s\n", CH, blocks_per_cu, flops / ms / 1e9);
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 4}) { run<1>(w); run<2>(w); run<4>(w); }
  return 0;
}
