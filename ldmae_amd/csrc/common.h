// Shared device/host helpers for the gfx950 (CDNA4, wave64) kernels of libldmae_hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ldmae_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
// fp16 activations (round 5): the TF32-CLASS forward path and VMAE pre-training under fp16 autocast.  fp16 has TF32's 10-bit mantissa; the operands of
// the GEMMs / attention products it is used for (LayerNorm outputs, q / k / v, softmax probabilities, GELU outputs) are O(1), so the 5-bit exponent is
// enough; the FORWARD GEMM epilogues saturate at +-65504 instead of producing infinities (EpiArgs::f16_max), gradient outputs overflow to infinity as
// under torch's fp16 autocast (the loss scaler skips the step).  v_mfma_f32_*_f16 runs at the bf16 rate with f32 accumulation.
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// global -> LDS DMA: every lane brings 16 B (or 4 B) from its own global address to LDS byte (lds + lane * 16) (or * 4); `lds`
// must be wave-uniform.  Issued as inline asm ON PURPOSE: when hipcc sees the builtin it may decide that later ds_reads alias
// the DMA destination and put `s_waitcnt vmcnt(0)` before the first fragment read of every K-step (seen in 2 of 6 epilogue
// instantiations of one GEMM kernel: the whole ring drains each step, -35 %).  Hidden from the compiler, the only vmcnt waits
// in a main loop are the counted ones written there; compiler-counted waits for ordinary loads can then only over-wait.
// The kernels order DMA against LDS reads themselves (counted s_waitcnt vmcnt + s_barrier).
__device__ __forceinline__ void glds16(const void* g, const void* lds) {
  const unsigned a = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(void, lds));
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(a) : "memory", "m0");
}
__device__ __forceinline__ void glds4(const void* g, const void* lds) {
  const unsigned a = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(void, lds));
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(g), "s"(a) : "memory", "m0");
}

// SGPR-base forms: the 64-bit base and the LDS address are wave-uniform (scalar registers, advanced per tile by scalar adds), the lane
// supplies only a 32-bit byte offset that is computed once per kernel: no per-piece 64-bit VALU address arithmetic in the main loops.
__device__ __forceinline__ void glds16_s(const void* sbase, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ void glds4_s(const void* sbase, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}

// ---------------------------------------------------------------- error plumbing
void ldmae_set_error(const char* fmt, ...);
#define LDMAE_FAIL(code, ...) do { ldmae_set_error(__VA_ARGS__); return (code); } while (0)
#define LDMAE_REQUIRE(cond, ...) do { if (!(cond)) LDMAE_FAIL(LDMAE_ERR_INVALID, __VA_ARGS__); } while (0)
#define LDMAE_CHECK_LAUNCH(name) do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) LDMAE_FAIL(LDMAE_ERR_HIP, "%s: %s", name, hipGetErrorString(e_)); } while (0)

// ---------------------------------------------------------------- scalar conversions
template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16>(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float v) { return (bf16)v; }   // v_cvt_pk_bf16_f32, RNE, NaN-safe
template <> __device__ __forceinline__ float to_f<f16>(f16 v) { return (float)v; }
template <> __device__ __forceinline__ f16 from_f<f16>(float v) { return (f16)v; }   // IEEE: beyond +-65504 -> infinity (what a loss scaler watches for)
// forward outputs of the fp16 GEMM epilogues SATURATE instead (mx = 65504; mx = infinity: no-op, the gradient GEMMs); NaN kept
__device__ __forceinline__ float sat_f16(float v, float mx) { return v == v ? __builtin_amdgcn_fmed3f(v, -mx, mx) : v; }
__device__ __forceinline__ float4 sat_f16(float4 v, float mx) { return make_float4(sat_f16(v.x, mx), sat_f16(v.y, mx), sat_f16(v.z, mx), sat_f16(v.w, mx)); }
// 8- / 4-element register vectors of a 2-byte activation type
template <typename T> struct Pack;
template <> struct Pack<bf16> { typedef bf16x8 v8; typedef bf16x4 v4; };
template <> struct Pack<f16> { typedef f16x8 v8; typedef f16x4 v4; };

// 8-element vector load/store as floats (16 B for bf16, 2 x 16 B for f32); pointers 16-B aligned
// Exact-erf GELU (timm Mlp act, models_mae.py:172).  f32 activations: libm's erff.  bf16 activations: erf by Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7, three orders below the bf16 rounding of the result): 2 transcendental + ~12 plain instructions, no branches, against
// ~40 instructions with three branches -- the GELU epilogue of the fc1 GEMM and the GELU backward pass were bound by that arithmetic.
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  return copysignf(fmaf(-poly, __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f), 1.f), x);
}
template <typename T> __device__ __forceinline__ float erf_act(float x) { return erff(x); }
template <> __device__ __forceinline__ float erf_act<bf16>(float x) { return erf_as(x); }
template <> __device__ __forceinline__ float erf_act<f16>(float x) { return erf_as(x); }      // 1.5e-7 against fp16's 5e-4 rounding
template <typename T> __device__ __forceinline__ float gelu_act(float y) { return 0.5f * y * (1.f + erf_act<T>(y * 0.70710678118654752f)); }

template <typename T> struct Vec8;
template <> struct Vec8<float> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
    float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
    *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
};
template <> struct Vec8<bf16> {
  static __device__ __forceinline__ void load(const bf16* p, float (&v)[8]) {
    bf16x8 a = *(const bf16x8*)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
  }
  static __device__ __forceinline__ void store(bf16* p, const float (&v)[8]) {
    bf16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (bf16)v[i];
    *(bf16x8*)p = a;
  }
};

template <> struct Vec8<f16> {
  static __device__ __forceinline__ void load(const f16* p, float (&v)[8]) {
    f16x8 a = *(const f16x8*)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
  }
  static __device__ __forceinline__ void store(f16* p, const float (&v)[8]) {
    f16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = from_f<f16>(v[i]);
    *(f16x8*)p = a;
  }
};

// ---------------------------------------------------------------- wave / block reductions (wave = 64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// The same sum without LDS round trips: four DPP adds leave every 16-lane row holding its row sum, four v_readlane + three adds join the
// rows (wave-uniform result).  __shfl_xor is a ds_bpermute: six DEPENDENT LDS round trips per reduction (~700 cycles) that sit in the
// middle of every row of the row-wise kernels; this form is ~50 cycles.  Different association than wave_sum (not bitwise interchangeable).
template <int CTRL> __device__ __forceinline__ float dpp_add_f(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v = dpp_add_f<0xB1>(v);            // quad_perm [1,0,3,2]
  v = dpp_add_f<0x4E>(v);            // quad_perm [2,3,0,1]
  v = dpp_add_f<0x141>(v);           // row_half_mirror
  v = dpp_add_f<0x140>(v);           // row_mirror
  const int i = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48));
  return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// sum over a power-of-two sub-group of lanes (width <= 64)
template <int W> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// RoPE on two interleaved pairs (VisionRotaryEmbeddingFast.forward, pos_embed.py:135: t * cos + rotate_half(t) * sin, rotate_half: (x0, x1) -> (-x1, x0)).
// One definition for the QK-norm / RoPE kernels of elementwise.hip and the fused qkv epilogue of gemm_nt_common.h: they agree bit for bit.
__device__ __forceinline__ float4 rope_apply(float4 t, float4 c, float4 s) {
  return make_float4(t.x * c.x - t.y * s.x, t.y * c.y + t.x * s.y, t.z * c.z - t.w * s.z, t.w * c.w + t.z * s.w);
}

// XCD-aware bijective block remap (8 XCDs, blocks dealt round-robin): consecutive
// logical tiles land on one XCD so neighbours share that XCD's L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// A/B knobs and the tile-timeline stamps exist only in the DIAGNOSTIC build (`make diag` -> libldmae_hip_diag.so, -DLDMAE_DIAG,
// declared in probe/ldmae_diag.h).  In the product library every knob reads 0 (the shipped behaviour) at compile time.
#ifdef LDMAE_DIAG
int ldmae_tune_get(int key);
#else
static inline constexpr int ldmae_tune_get(int) { return 0; }
#endif
// launch counts by kernel family (core.hip; LDMAE_COUNT_* of the public header)
void ldmae_count(int family);
// timing hook (core.hip)
bool ldmae_prof_is_on();
long ldmae_prof_begin(hipStream_t st, double flops);
void ldmae_prof_end(long idx, hipStream_t st);

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }
static inline unsigned cdiv(long a, long b) { return (unsigned)((a + b - 1) / b); }

// sigmoid on the fast-math units: v_exp_f32 + v_rcp_f32 (1 ulp each) instead of the 12-instruction IEEE division; every SwiGLU
// site (fused GEMM epilogues and the standalone kernels) uses this one definition, so fused and unfused paths stay bit-identical.
__device__ __forceinline__ float fast_sigmoid(float a) { return __builtin_amdgcn_rcpf(1.f + __expf(-a)); }
