// Probe: what bounds the GEMM epilogue's store tail?  One 512-thread workgroup per CU writes 256x256 bf16 tiles of a
// [M][N] matrix with the epilogue's access shape; timed for different numbers of storing CUs and access shapes.
//   mode 0: 8 lanes x 16 B per 128-B row segment, 8 rows per wave instruction (the shipped strip epilogue)
//   mode 1: 32 lanes x 16 B = one full 512-B tile row per half wave (2 rows per instruction)
//   mode 2: mode 0 with nontemporal stores
//   mode 3: mode 1 with nontemporal stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) float f4;
template <int MODE>
__global__ __launch_bounds__(512) void store_kernel(unsigned short* C, int M, int N, int ntiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 2, wn = wave & 3;
  const int tiles_n = N / 256;
  f4 v = {1.f, 2.f, 3.f, (float)lane};
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int m0 = (t / tiles_n) * 256, n0 = (t % tiles_n) * 256;
    if (MODE == 0 || MODE == 2) {
      // wave owns rows wm*128 .. +128, cols wn*64 .. +64 : 8 strips of 16 rows, 2 instructions per strip
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int m = m0 + wm * 128 + i * 16 + it * 8 + (lane >> 3), n = n0 + wn * 64 + (lane & 7) * 8;
          f4* p = (f4*)(C + (size_t)m * N + n);
          if (MODE == 2) __builtin_nontemporal_store(v, p); else *p = v;
        }
    } else {
      // wave owns 32 full tile rows: 16 instructions of 2 rows x 512 B
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = m0 + wave * 32 + i * 2 + (lane >> 5), n = n0 + (lane & 31) * 8;
        f4* p = (f4*)(C + (size_t)m * N + n);
        if (MODE == 3) __builtin_nontemporal_store(v, p); else *p = v;
      }
    }
  }
}
int main() {
  const int M = 262144, N = 2304;
  unsigned short* C;
  hipMalloc(&C, (size_t)M * N * 2);
  hipMemset(C, 0, (size_t)M * N * 2);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int ntiles_full = (M / 256) * (N / 256);
  for (int mode = 0; mode < 4; ++mode)
    for (int grid : {256, 128, 64, 32, 8}) {
      // every workgroup writes the same number of tiles (36) whatever the grid, so per-CU time is comparable
      const int ntiles = grid * 36;
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(a);
        switch (mode) {
          case 0: hipLaunchKernelGGL(store_kernel<0>, dim3(grid), dim3(512), 0, 0, C, M, N, ntiles); break;
          case 1: hipLaunchKernelGGL(store_kernel<1>, dim3(grid), dim3(512), 0, 0, C, M, N, ntiles); break;
          case 2: hipLaunchKernelGGL(store_kernel<2>, dim3(grid), dim3(512), 0, 0, C, M, N, ntiles); break;
          default: hipLaunchKernelGGL(store_kernel<3>, dim3(grid), dim3(512), 0, 0, C, M, N, ntiles); break;
        }
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
      }
      const double bytes = (double)ntiles * 256 * 256 * 2;
      printf("mode %d grid %3d: %.3f ms  %.2f TB/s total  %.1f GB/s per CU  %.2f us per 128-KiB tile\n", mode, grid, best,
             bytes / best / 1e9, bytes / best / 1e6 / grid, best * 1e3 / 36);
    }
  (void)ntiles_full;
  return 0;
}
