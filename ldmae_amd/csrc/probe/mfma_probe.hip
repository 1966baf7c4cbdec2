// gfx950 layout self-test: prints PASS/FAIL for every MFMA / LDS-transpose-read
// lane map the kernels in ../ rely on.  Exact small-integer data, host reference.
//   hipcc --offload-arch=gfx950 -O2 mfma_probe.hip -o mfma_probe && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(2);} } while (0)

__device__ inline __bf16 f2bf(float f) { return (__bf16)f; }

// ---- 1. ds_read_b64_tr_b16 semantics: dump what each lane receives
__global__ void k_tr(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short T[16][64];   // 128-B rows
  for (int i = threadIdx.x; i < 16 * 64; i += 64) T[i / 64][i % 64] = (unsigned short)((i / 64) * 256 + (i % 64));
  __syncthreads();
  int l = threadIdx.x, g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
  // group g reads block rows 4g..4g+3, cols 16..31 ; lane 4q+p supplies row q, cols 4p..4p+3
  const unsigned short* addr = &T[4 * g + q][16 + 4 * p];
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = (unsigned short)v[e];
}

// ---- 2. MFMA 32x32x16 bf16 : A[i][k] lane (r=l&31,h=l>>5) elem j = A[r][8h+j]; B elem j = B[8h+j][r]
__global__ void k_mfma32(const float* A, const float* B, float* C) {   // A[32][16], B[16][32], C[32][32]
  int l = threadIdx.x, r = l & 31, h = l >> 5;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = f2bf(A[r * 16 + 8 * h + j]); b[j] = f2bf(B[(8 * h + j) * 32 + r]); }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int reg = 0; reg < 16; ++reg) { int row = (reg & 3) + 8 * (reg >> 2) + 4 * h; C[row * 32 + r] = c[reg]; }
}
// ---- 3. MFMA 16x16x32 bf16 : lane l: A[l&15][8(l>>4)+j], B[8(l>>4)+j][l&15]; C col=l&15,row=(l>>4)*4+reg
__global__ void k_mfma16(const float* A, const float* B, float* C) {   // A[16][32], B[32][16], C[16][16]
  int l = threadIdx.x, r = l & 15, g = l >> 4;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = f2bf(A[r * 32 + 8 * g + j]); b[j] = f2bf(B[(8 * g + j) * 16 + r]); }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int reg = 0; reg < 4; ++reg) C[(g * 4 + reg) * 16 + r] = c[reg];
}
// ---- 4. f32 MFMA 32x32x2 : A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]
__global__ void k_mfma32f(const float* A, const float* B, float* C) {  // A[32][2], B[2][32]
  int l = threadIdx.x, r = l & 31, h = l >> 5;
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * 2 + h], B[h * 32 + r], c, 0, 0, 0);
  for (int reg = 0; reg < 16; ++reg) { int row = (reg & 3) + 8 * (reg >> 2) + 4 * h; C[row * 32 + r] = c[reg]; }
}
// ---- 5. f32 MFMA 16x16x4 : A[l&15][k=l>>4], B[k=l>>4][l&15]
__global__ void k_mfma16f(const float* A, const float* B, float* C) {  // A[16][4], B[4][16]
  int l = threadIdx.x, r = l & 15, g = l >> 4;
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * 4 + g], B[g * 16 + r], c, 0, 0, 0);
  for (int reg = 0; reg < 4; ++reg) C[(g * 4 + reg) * 16 + r] = c[reg];
}
// ---- 6. accumulator tile X[32][32] (col on lane) as B operand of the next 32x32x16: Y = A2 . X
// k-step s uses regs 8s..8s+7; element j of lane half h is row 16s + 8(j>>2) + 4h + (j&3) of X.
__global__ void k_acc_as_b(const float* A1, const float* B1, const float* A2, float* Y) { // A1[32][16],B1[16][32],A2[32][32]
  int l = threadIdx.x, r = l & 31, h = l >> 5;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = f2bf(A1[r * 16 + 8 * h + j]); b[j] = f2bf(B1[(8 * h + j) * 32 + r]); }
  f32x16 x = {0};
  x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, x, 0, 0, 0);
  f32x16 y = {0};
  for (int s = 0; s < 2; ++s) {
    bf16x8 xb, a2;
    for (int j = 0; j < 8; ++j) {
      xb[j] = f2bf(x[8 * s + j]);
      int krow = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
      a2[j] = f2bf(A2[r * 32 + krow]);
    }
    y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, xb, y, 0, 0, 0);
  }
  for (int reg = 0; reg < 16; ++reg) { int row = (reg & 3) + 8 * (reg >> 2) + 4 * h; Y[row * 32 + r] = y[reg]; }
}
// ---- 7. A operand of 32x32x16 built by tr-reads from a k-major LDS tile T[k][row] (the attention-V / dW pattern)
// lane (r,h) group g=l>>4: r = 16(g&1)+i, h = g>>1 ; two tr reads: k rows 8h..8h+3 and 8h+4..8h+7, cols 16(g&1)..+15
__global__ void k_tr_a_operand(const float* At /*[16 k][32 rows]*/, const float* B /*[16][32]*/, float* C) {
  __shared__ __attribute__((aligned(16))) __bf16 T[16][32];   // 64-B rows
  for (int i = threadIdx.x; i < 16 * 32; i += 64) T[i / 32][i % 32] = f2bf(At[i]);
  __syncthreads();
  int l = threadIdx.x, r = l & 31, h = l >> 5, g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
  const __bf16* a0 = &T[8 * h + q][16 * (g & 1) + 4 * p];
  const __bf16* a1 = &T[8 * h + 4 + q][16 * (g & 1) + 4 * p];
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
  union { bf16x8 v; short s[8]; } a;
  for (int j = 0; j < 4; ++j) { a.s[j] = lo[j]; a.s[4 + j] = hi[j]; }
  bf16x8 b;
  for (int j = 0; j < 8; ++j) b[j] = f2bf(B[(8 * h + j) * 32 + r]);
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b, c, 0, 0, 0);
  for (int reg = 0; reg < 16; ++reg) { int row = (reg & 3) + 8 * (reg >> 2) + 4 * h; C[row * 32 + r] = c[reg]; }
}
// ---- 8. same for 16x16x32: lane l: row = l&15, k = 8(l>>4)+j ; tr reads: rows(k) 8g..8g+3 / +4, cols 0..15
__global__ void k_tr_a_operand16(const float* At /*[32 k][16 rows]*/, const float* B /*[32][16]*/, float* C) {
  __shared__ __attribute__((aligned(16))) __bf16 T[32][16];   // 32-B rows
  for (int i = threadIdx.x; i < 32 * 16; i += 64) T[i / 16][i % 16] = f2bf(At[i]);
  __syncthreads();
  int l = threadIdx.x, r = l & 15, g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)&T[8 * g + q][4 * p]);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)&T[8 * g + 4 + q][4 * p]);
  union { bf16x8 v; short s[8]; } a;
  for (int j = 0; j < 4; ++j) { a.s[j] = lo[j]; a.s[4 + j] = hi[j]; }
  bf16x8 b;
  for (int j = 0; j < 8; ++j) b[j] = f2bf(B[(8 * g + j) * 16 + r]);
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b, c, 0, 0, 0);
  for (int reg = 0; reg < 4; ++reg) C[(g * 4 + reg) * 16 + r] = c[reg];
}
// ---- 9. global_load_lds 16B: LDS dest = wave-uniform base + lane*16
__global__ void k_glds(const unsigned* src, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned L[512];
  for (int i = threadIdx.x; i < 512; i += 64) L[i] = 0xdeadbeef;
  __syncthreads();
  // lane l fetches 16 B from src + (63-l)*4 dwords (reversed) -> should land at L[4*l .. 4*l+3]
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (63 - threadIdx.x) * 4),
                                   (__attribute__((address_space(3))) void*)(L + 64), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = L[i];
}
// ---- 10. permlane32_swap semantics
__global__ void k_swap(unsigned* out) {
  unsigned a = 1000 + threadIdx.x, b = 2000 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[threadIdx.x * 2] = r[0]; out[threadIdx.x * 2 + 1] = r[1];
}

static void matmul(const float* A, const float* B, float* C, int M, int N, int K) {
  for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) { float s = 0; for (int k = 0; k < K; ++k) s += A[i * K + k] * B[k * N + j]; C[i * N + j] = s; }
}
static int cmp(const char* name, const float* got, const float* ref, int n) {
  int bad = 0; for (int i = 0; i < n; ++i) if (got[i] != ref[i]) ++bad;
  printf("%-28s %s (%d/%d mismatches)\n", name, bad ? "FAIL" : "PASS", bad, n);
  return bad;
}
static void fill(std::vector<float>& v, int seed) { unsigned s = seed * 2654435761u + 12345; for (auto& x : v) { s = s * 1664525u + 1013904223u; x = (float)((int)((s >> 16) % 7) - 3); } }

int main() {
  int fails = 0;
  float *dA, *dB, *dC, *dA2; CK(hipMalloc(&dA, 4096 * 4)); CK(hipMalloc(&dB, 4096 * 4)); CK(hipMalloc(&dC, 4096 * 4)); CK(hipMalloc(&dA2, 4096 * 4));
  std::vector<float> A(512), B(512), C(1024), R(1024), A2(1024);
  { // 1
    unsigned short* d; CK(hipMalloc(&d, 512)); k_tr<<<1, 64>>>(d); std::vector<unsigned short> o(256); CK(hipMemcpy(o.data(), d, 512, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) { int g = l >> 4, i = l & 15; for (int e = 0; e < 4; ++e) if (o[l * 4 + e] != (4 * g + e) * 256 + 16 + i) ++bad; }
    printf("%-28s %s\n", "tr16_b64 (lane i<-col i, elem e<-row e)", bad ? "FAIL" : "PASS");
    if (bad) { for (int l = 0; l < 64; ++l) { printf("  lane %2d:", l); for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", o[l * 4 + e] >> 8, o[l * 4 + e] & 255); printf("\n"); } }
    fails += bad;
  }
  auto up = [&](float* d, std::vector<float>& v) { CK(hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice)); };
  auto down = [&](int n) { CK(hipMemcpy(C.data(), dC, n * 4, hipMemcpyDeviceToHost)); };
  fill(A, 1); fill(B, 2); up(dA, A); up(dB, B);
  k_mfma32<<<1, 64>>>(dA, dB, dC); down(1024); matmul(A.data(), B.data(), R.data(), 32, 32, 16); fails += cmp("mfma 32x32x16 bf16", C.data(), R.data(), 1024);
  k_mfma16<<<1, 64>>>(dA, dB, dC); down(256); matmul(A.data(), B.data(), R.data(), 16, 16, 32); fails += cmp("mfma 16x16x32 bf16", C.data(), R.data(), 256);
  k_mfma32f<<<1, 64>>>(dA, dB, dC); down(1024); matmul(A.data(), B.data(), R.data(), 32, 32, 2); fails += cmp("mfma 32x32x2 f32", C.data(), R.data(), 1024);
  k_mfma16f<<<1, 64>>>(dA, dB, dC); down(256); matmul(A.data(), B.data(), R.data(), 16, 16, 4); fails += cmp("mfma 16x16x4 f32", C.data(), R.data(), 256);
  { fill(A2, 3); up(dA2, A2); k_acc_as_b<<<1, 64>>>(dA, dB, dA2, dC); down(1024);
    std::vector<float> X(1024); matmul(A.data(), B.data(), X.data(), 32, 32, 16); matmul(A2.data(), X.data(), R.data(), 32, 32, 32);
    fails += cmp("acc-as-B-operand k order", C.data(), R.data(), 1024); }
  { // 7: At[16][32] -> A[row][k] = At[k][row]
    std::vector<float> At(512), Am(512); fill(At, 5); for (int k = 0; k < 16; ++k) for (int r = 0; r < 32; ++r) Am[r * 16 + k] = At[k * 32 + r];
    up(dA, At); k_tr_a_operand<<<1, 64>>>(dA, dB, dC); down(1024); matmul(Am.data(), B.data(), R.data(), 32, 32, 16); fails += cmp("tr-read A operand 32x32x16", C.data(), R.data(), 1024); }
  { std::vector<float> At(512), Am(512); fill(At, 6); for (int k = 0; k < 32; ++k) for (int r = 0; r < 16; ++r) Am[r * 32 + k] = At[k * 16 + r];
    up(dA, At); k_tr_a_operand16<<<1, 64>>>(dA, dB, dC); down(256); matmul(Am.data(), B.data(), R.data(), 16, 16, 32); fails += cmp("tr-read A operand 16x16x32", C.data(), R.data(), 256); }
  { unsigned *ds, *dout2; CK(hipMalloc(&ds, 1024)); CK(hipMalloc(&dout2, 2048)); std::vector<unsigned> s(256), o(512); for (int i = 0; i < 256; ++i) s[i] = i;
    CK(hipMemcpy(ds, s.data(), 1024, hipMemcpyHostToDevice)); k_glds<<<1, 64>>>(ds, dout2); CK(hipMemcpy(o.data(), dout2, 2048, hipMemcpyDeviceToHost));
    int bad = 0; for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) if (o[64 + 4 * l + e] != (unsigned)((63 - l) * 4 + e)) ++bad;
    for (int i = 0; i < 64; ++i) if (o[i] != 0xdeadbeef) ++bad;
    printf("%-28s %s\n", "global_load_lds x16 (base+lane*16)", bad ? "FAIL" : "PASS"); fails += bad; }
  { unsigned* d; CK(hipMalloc(&d, 512)); k_swap<<<1, 64>>>(d); std::vector<unsigned> o(128); CK(hipMemcpy(o.data(), d, 512, hipMemcpyDeviceToHost));
    // expected (guide T21): lanes 32-63 of vdst(a) swap with lanes 0-31 of src(b)
    int bad = 0; for (int l = 0; l < 64; ++l) { unsigned ea = l < 32 ? 1000 + l : 2000 + (l - 32), eb = l < 32 ? 1000 + l + 32 : 2000 + l; if (o[2 * l] != ea || o[2 * l + 1] != eb) ++bad; }
    printf("%-28s %s\n", "permlane32_swap", bad ? "FAIL" : "PASS");
    if (bad) for (int l = 0; l < 64; l += 8) printf("  lane %2d: r0=%u r1=%u\n", l, o[2 * l], o[2 * l + 1]);
  }
  CK(hipDeviceSynchronize());
  printf("probe done, fails=%d\n", fails);
  return 0;
}
