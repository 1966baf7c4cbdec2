// Power ceiling of the matrix pipe alone: every wave runs back-to-back v_mfma_f32_16x16x32_bf16 on register operands (random bf16
// values, 16 independent accumulator tiles, no LDS, no memory traffic) for a few seconds; prints the sustained TFLOP/s.  Run
// `rocm-smi --showpower --showclocks` beside it for the clock and package power it holds.
//   hipcc --offload-arch=gfx950 -O3 probe/mfma_power.hip -o probe/mfma_power && probe/mfma_power [waves_per_simd=2] [seconds=4] [32x32x16=0]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(256) void mfma_loop(const unsigned* __restrict__ seed, float* __restrict__ out, int iters) {
  unsigned s = seed[threadIdx.x + blockIdx.x * 256];
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      s = s * 1664525u + 1013904223u; a[i][j] = (__bf16)(((int)(s >> 8) % 4096 - 2048) / 1024.0f);
      s = s * 1664525u + 1013904223u; b[i][j] = (__bf16)(((int)(s >> 8) % 4096 - 2048) / 1024.0f);
    }
  f32x4 acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  float t = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[threadIdx.x + blockIdx.x * 256] = t;
}

typedef __attribute__((ext_vector_type(16))) float f32x16;
// the same FLOPs per iteration with v_mfma_f32_32x32x16_bf16: 8 MFMAs on 2 x 4 accumulator tiles of 32 x 32
__global__ __launch_bounds__(256) void mfma_loop32(const unsigned* __restrict__ seed, float* __restrict__ out, int iters) {
  unsigned s = seed[threadIdx.x + blockIdx.x * 256];
  bf16x8 a[2], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      s = s * 1664525u + 1013904223u; if (i < 2) a[i][j] = (__bf16)(((int)(s >> 8) % 4096 - 2048) / 1024.0f);
      s = s * 1664525u + 1013904223u; b[i][j] = (__bf16)(((int)(s >> 8) % 4096 - 2048) / 1024.0f);
    }
  f32x16 acc[2][4];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int t = 0; t < 16; ++t) acc[i][j][t] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  float t = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int k = 0; k < 16; ++k) t += acc[i][j][k];
  out[threadIdx.x + blockIdx.x * 256] = t;
}

int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 2;
  const double secs = argc > 2 ? atof(argv[2]) : 4.0;
  const int shape32 = argc > 3 ? atoi(argv[3]) : 0;       // 1: v_mfma_f32_32x32x16_bf16 (same FLOPs per iteration)
  int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  const int blocks = ncu * wps;                  // 4 waves per block -> wps waves per SIMD
  unsigned* seed; float* out;
  hipMalloc(&seed, blocks * 256 * 4); hipMalloc(&out, blocks * 256 * 4);
  unsigned* h = (unsigned*)malloc(blocks * 256 * 4);
  for (int i = 0; i < blocks * 256; ++i) h[i] = 12345u + 7919u * i;
  hipMemcpy(seed, h, blocks * 256 * 4, hipMemcpyHostToDevice);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto go = [&]() { if (shape32) hipLaunchKernelGGL(mfma_loop32, dim3(blocks), dim3(256), 0, 0, seed, out, iters); else hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, seed, out, iters); };
  go(); hipDeviceSynchronize();
  double total_ms = 0; long launches = 0;
  while (total_ms < secs * 1e3) {
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) go();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); total_ms += ms; launches += 10;
  }
  const double flops = (double)launches * blocks * 4 /*waves*/ * iters * 16 /*mfma*/ * 2.0 * 16 * 16 * 32;
  printf("%s waves/SIMD %d: %.1f TFLOP/s sustained over %.1f s (%d CUs)\n", shape32 ? "32x32x16" : "16x16x32", wps, flops / (total_ms * 1e-3) / 1e12, total_ms * 1e-3, ncu);
  return 0;
}
