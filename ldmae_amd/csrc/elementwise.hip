// HBM-bound row / elementwise kernels of the LightningDiT block (gfx950, wave64).
// One wave per row for the norm kernels (12 elements per lane at D = 768, shuffles for the
// row reduce); per-sample and per-column gradient sums are two-stage (per-workgroup partial
// rows, then a fixed-order reduce) so results are bitwise reproducible run to run.
#include "common.h"

// ------------------------------------------------------------------ small vector helpers
template <typename T> __device__ __forceinline__ float4 load4(const T* p);
template <> __device__ __forceinline__ float4 load4<float>(const float* p) { const f32x4 v = __builtin_nontemporal_load((const f32x4*)p); return make_float4(v[0], v[1], v[2], v[3]); }
template <> __device__ __forceinline__ float4 load4<bf16>(const bf16* p) {
  bf16x4 v = __builtin_nontemporal_load((const bf16x4*)p);
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
template <> __device__ __forceinline__ float4 load4<f16>(const f16* p) {
  const f16x4 a = *(const f16x4*)p;
  return make_float4((float)a[0], (float)a[1], (float)a[2], (float)a[3]);
}
template <typename T> __device__ __forceinline__ void store4(T* p, float4 v);
template <> __device__ __forceinline__ void store4<float>(float* p, float4 v) { f32x4 o = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(o, (f32x4*)p); }
template <> __device__ __forceinline__ void store4<bf16>(bf16* p, float4 v) {
  bf16x4 o; o[0] = (bf16)v.x; o[1] = (bf16)v.y; o[2] = (bf16)v.z; o[3] = (bf16)v.w;
  __builtin_nontemporal_store(o, (bf16x4*)p);
}
__device__ __forceinline__ float4 f4(float a) { return make_float4(a, a, a, a); }
__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator-(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 operator*(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 operator*(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float hsum(float4 a) { return (a.x + a.y) + (a.z + a.w); }

// out[o*out_ld + c] = beta*out + sum_{r<group} P[(o*group + r)*p_ld + c]; 8 row-lanes per column stride the group,
// partial sums are combined in a fixed order (bitwise reproducible).
template <int RL>
__global__ __launch_bounds__(32 * RL) void group_reduce_kernel(const float* __restrict__ P, int p_ld, int nout, int cols, int group,
                                                              float* __restrict__ out, int out_ld, float beta,
                                                              const float* __restrict__ P2 = nullptr, float* __restrict__ out2 = nullptr,
                                                              float beta2 = 0.f) {
  __shared__ float red[RL][33];
  if (blockIdx.z == 1) { P = P2; out = out2; beta = beta2; }        // a second reduction of the same shape in the same launch
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl, o = blockIdx.y;
  float s = 0.f;
  if (c < cols) {
    const float* p = P + (size_t)o * group * p_ld + c;
    for (int r = rl; r < group; r += RL) s += p[(size_t)r * p_ld];
  }
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < cols) {
    float t = red[0][cl];
#pragma unroll
    for (int k = 1; k < RL; ++k) t += red[k][cl];
    float* q = out + (size_t)o * out_ld + c;
    *q = (beta != 0.f ? beta * *q : 0.f) + t;
  }
}
static void group_reduce(const float* P, int p_ld, int nout, int cols, int group, float* out, int out_ld, float beta, hipStream_t st,
                         const float* P2 = nullptr, float* out2 = nullptr, float beta2 = 0.f) {
  // long groups (thousands of per-workgroup partial rows into one output row): 32 row lanes, otherwise 8.  P2 / out2: a second reduction
  // of the same shape rides in the same launch (grid z = 2), with exactly the arithmetic it would have on its own
  const unsigned gz = P2 ? 2 : 1;
  if (group >= 256)
    hipLaunchKernelGGL(group_reduce_kernel<32>, dim3(cdiv(cols, 32), nout, gz), dim3(1024), 0, st, P, p_ld, nout, cols, group, out, out_ld, beta, P2, out2, beta2);
  else
    hipLaunchKernelGGL(group_reduce_kernel<8>, dim3(cdiv(cols, 32), nout, gz), dim3(256), 0, st, P, p_ld, nout, cols, group, out, out_ld, beta, P2, out2, beta2);
}

// rows per workgroup for the per-sample reductions: largest of 64/32/16/8/4 dividing rows_per_batch
static int pick_rows_per_wg(int rows_per_batch) {
  const int cap = ldmae_tune_get(10) > 0 ? ldmae_tune_get(10) : 64;      // tune key 10: A/B knob
  for (int r = cap; r >= 1; r >>= 1) if (rows_per_batch % r == 0) return r;      // (odd token counts -- a 5 x 5 grid -- end at 1 row per workgroup: correct, slow)
  return 0;
}

// a product that is never contracted into a following add (HIP's __fmul_rn is a plain `*`): its consumers sum the value AS STORED
__device__ __forceinline__ float4 mul_rn(float4 a, float4 b) {
#pragma clang fp contract(off)
  return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
}

// ------------------------------------------------------------------ RMSNorm + modulate
// FULL (D % 256 == 0, M % 16 == 0, rows_per_batch % 16 == 0: the workgroup's 16 rows belong to one sample): no per-chunk guards, and the
// per-sample (1 + scale) / shift vectors are loaded ONCE per wave before the row loop.  The guarded form re-loads them per row and chunk
// behind a branch and a full `s_waitcnt vmcnt(0)` each: six serialized L2 round trips in the middle of every row.
// (A variant with all four rows of a wave requested up front -- 12 loads in flight per lane, 100 VGPRs -- measured 3 % SLOWER than one row
// at a time: here the waves in flight carry the bandwidth.)
template <int NCH, typename OutT, bool FULL = false>
__global__ __launch_bounds__(256) void rmsnorm_mod_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ shift, const float* __restrict__ scale,
                                                              int mod_ld, OutT* __restrict__ out, float* __restrict__ rstd, int M, int D,
                                                              int rpb, float eps, int center = 0) {
  // center (guarded form only): LayerNorm WITHOUT affine parameters (the use_rmsnorm=False blocks, lightningdit.py:200-201) = the RMS norm of
  // the centred row, w = NULL -> 1: y = (x - mean) * rsqrt(mean((x - mean)^2) + eps) * (1 + scale) + shift
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nch = D >> 2;
  float4 wv[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) { const int c = lane + 64 * i; wv[i] = c < nch ? (w ? *(const float4*)(w + 4 * c) : f4(1.f)) : f4(0.f); }
  if constexpr (FULL) {
    const int m0 = (blockIdx.x * 4 + wave) * 4, b = m0 / rpb;
    float4 sc1[NCH], sh[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      sc1[i] = scale ? f4(1.f) + *(const float4*)(scale + (size_t)b * mod_ld + 4 * lane + 256 * i) : f4(1.f);
      sh[i] = shift ? *(const float4*)(shift + (size_t)b * mod_ld + 4 * lane + 256 * i) : f4(0.f);
    }
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + r;
      float4 xv[NCH];
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) xv[i] = *(const float4*)(x + (size_t)m * D + 4 * lane + 256 * i);
#pragma unroll
      for (int i = 0; i < NCH; ++i) ss += hsum(xv[i] * xv[i]);
      ss = wave_sum(ss);
      const float rs = rsqrtf(ss / (float)D + eps);
      if (lane == 0 && rstd) rstd[m] = rs;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        float4 y = (xv[i] * rs) * wv[i];
        if (scale) y = y * sc1[i];
        if (shift) y = y + sh[i];
        store4<OutT>(out + (size_t)m * D + 4 * lane + 256 * i, y);
      }
    }
    return;
  }
  for (int r = 0; r < 4; ++r) {
    const int m = (blockIdx.x * 4 + wave) * 4 + r;
    if (m >= M) return;
    const int b = m / rpb;
    float4 xv[NCH];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      xv[i] = c < nch ? *(const float4*)(x + (size_t)m * D + 4 * c) : f4(0.f);
      ss += hsum(xv[i] * xv[i]);
    }
    if (center) {
      float s1 = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) s1 += hsum(xv[i]);
      const float mean = wave_sum(s1) / (float)D;
      ss = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        if (lane + 64 * i < nch) xv[i] = xv[i] - f4(mean);
        ss += hsum(xv[i] * xv[i]);
      }
    }
    ss = wave_sum(ss);
    const float rs = rsqrtf(ss / (float)D + eps);
    if (lane == 0 && rstd) rstd[m] = rs;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        float4 y = (xv[i] * rs) * wv[i];
        if (scale) y = y * (f4(1.f) + *(const float4*)(scale + (size_t)b * mod_ld + 4 * c));
        if (shift) y = y + *(const float4*)(shift + (size_t)b * mod_ld + 4 * c);
        store4<OutT>(out + (size_t)m * D + 4 * c, y);
      }
    }
  }
}

// partials P[wg][3][D] = {sum dout, sum dout*y, sum dy*n} over the workgroup's rows (one sample)
// GATE: the gated-residual backward of the branch BELOW this norm (the attention branch under norm2) rides along: the updated residual
// gradient is consumed while it is in registers -- dy = dx_new * gate[b] (rounded to T), dgate partials sum dx_new * y, bias-gradient
// partials sum dy -- instead of being re-read by a separate gate_bwd pass (805 MB per block).  Same arithmetic, same partial layout
// and same summation order as gate_bwd_kernel.
struct GateBwdArgs { const void* y; const float* gate; int gate_ld; void* dy; float* Pg; float* Pb; int dx_overwrite; int center; };
template <int NCH, typename T, bool GATE, bool FULL = false>
__global__ __launch_bounds__(256) void rmsnorm_mod_bwd_kernel(const T* __restrict__ dout, const float* __restrict__ x,
                                                              const float* __restrict__ w, const float* __restrict__ scale, int mod_ld,
                                                              const float* __restrict__ rstd, float* __restrict__ dx, float* __restrict__ P,
                                                              int M, int D, int rpb, int rows_per_wg, GateBwdArgs ga) {
  extern __shared__ float red[];   // [4 waves][3][D]; GATE: the first 3*D floats hold the per-column constants during the row loop
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nch = D >> 2;
  const int m_base = blockIdx.x * rows_per_wg, b = m_base / rpb;
  float4 wv[GATE ? 1 : NCH], sc1[GATE ? 1 : NCH], a_sh[NCH], a_sc[NCH], a_w[NCH];
  float4 a_g[GATE ? NCH : 1], a_b[GATE ? NCH : 1];
  const T* gy = (const T*)ga.y;
  T* gdy = (T*)ga.dy;
  // GATE: five accumulator sets live in registers, so the three per-column constants (norm weight, 1 + scale, gate) are kept in LDS
  // and re-read per row instead (36 registers: 174 -> 3 waves per SIMD instead of 2 for a kernel that lives on bytes in flight)
  float* cw = red;
  float* cs = red + D;
  float* cg = red + 2 * D;
  if constexpr (GATE) {
    for (int c = threadIdx.x; c < nch; c += 256) {
      *(float4*)(cw + 4 * c) = w ? *(const float4*)(w + 4 * c) : f4(1.f);
      *(float4*)(cs + 4 * c) = scale ? f4(1.f) + *(const float4*)(scale + (size_t)b * mod_ld + 4 * c) : f4(1.f);
      *(float4*)(cg + 4 * c) = *(const float4*)(ga.gate + (size_t)b * ga.gate_ld + 4 * c);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if constexpr (!GATE) {
      wv[i] = c < nch ? (w ? *(const float4*)(w + 4 * c) : f4(1.f)) : f4(0.f);
      sc1[i] = (c < nch && scale) ? f4(1.f) + *(const float4*)(scale + (size_t)b * mod_ld + 4 * c) : f4(1.f);
    }
    a_sh[i] = a_sc[i] = a_w[i] = f4(0.f);
    if constexpr (GATE) a_g[i] = a_b[i] = f4(0.f);
  }
  if constexpr (FULL) {
    // D == NCH * 256 (host): no per-chunk guards, and everything a row needs from HBM -- dout, x, the old dx and (GATE) y -- is requested
    // before the first use: 4 * NCH loads in flight per lane.  (The guarded form below compiles to one branch + `s_waitcnt vmcnt(0)` per
    // chunk: 2 loads in flight.)  Per-element arithmetic and summation order are those of the guarded form.
    for (int r = wave; r < rows_per_wg; r += 4) {
      const int m = m_base + r;
      if (m >= M) break;
      const size_t ro = (size_t)m * D + 4 * lane;
      float4 g[NCH], nv[NCH], dold[NCH], yv[GATE ? NCH : 1];
#pragma unroll
      for (int i = 0; i < NCH; ++i) { g[i] = load4<T>(dout + ro + 256 * i); nv[i] = *(const float4*)(x + ro + 256 * i); }
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        dold[i] = ga.dx_overwrite ? f4(0.f) : *(const float4*)(dx + ro + 256 * i);
        if constexpr (GATE) yv[i] = load4<T>(gy + ro + 256 * i);
      }
      const float rs = rstd[m];
      float4 dn[NCH];
      float dot = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        const float4 wc = GATE ? *(const float4*)(cw + 4 * c) : wv[i], sc = GATE ? *(const float4*)(cs + 4 * c) : sc1[i];
        nv[i] = nv[i] * rs;
        const float4 dy = g[i] * sc;
        a_sh[i] = a_sh[i] + g[i];
        a_sc[i] = a_sc[i] + g[i] * (nv[i] * wc);
        a_w[i] = a_w[i] + dy * nv[i];
        dn[i] = dy * wc;
        dot += hsum(dn[i] * nv[i]);
      }
      dot = wave_sum_dpp(dot) / (float)D;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        const float4 d0 = (dn[i] - nv[i] * dot) * rs;
        const float4 gn = ga.dx_overwrite ? d0 : dold[i] + d0;
        *(float4*)(dx + ro + 256 * i) = gn;
        if constexpr (GATE) {
          a_g[i] = a_g[i] + gn * yv[i];
          const float4 d = mul_rn(gn, *(const float4*)(cg + 4 * c));
          store4<T>(gdy + ro + 256 * i, d);
          a_b[i] = a_b[i] + make_float4(to_f<T>(from_f<T>(d.x)), to_f<T>(from_f<T>(d.y)), to_f<T>(from_f<T>(d.z)), to_f<T>(from_f<T>(d.w)));
        }
      }
    }
  } else
  for (int r = wave; r < rows_per_wg; r += 4) {
    const int m = m_base + r;
    if (m >= M) break;
    const float rs = rstd[m];
    float4 nv[NCH], dn[NCH];
    float dot = 0.f, mean = 0.f;
    // center (LayerNorm without affine parameters: the norm of the CENTRED row): the row mean is recomputed from x (a second, cache-resident
    // read), n = (x - mean) * rstd, and the centring's own backward takes the mean of dn out of the result below.  mean = msum = 0 otherwise:
    // x - 0 and d - 0 leave every bit of the RMS form as it was.
    if (ga.center) {
      float s1 = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) { const int c = lane + 64 * i; if (c < nch) s1 += hsum(*(const float4*)(x + (size_t)m * D + 4 * c)); }
      mean = wave_sum(s1) / (float)D;
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        const float4 g = load4<T>(dout + (size_t)m * D + 4 * c);
        const float4 wc = GATE ? *(const float4*)(cw + 4 * c) : wv[i], sc = GATE ? *(const float4*)(cs + 4 * c) : sc1[i];
        nv[i] = (*(const float4*)(x + (size_t)m * D + 4 * c) - f4(mean)) * rs;
        const float4 dy = g * sc;
        a_sh[i] = a_sh[i] + g;
        a_sc[i] = a_sc[i] + g * (nv[i] * wc);
        a_w[i] = a_w[i] + dy * nv[i];
        dn[i] = dy * wc;
        dot += hsum(dn[i] * nv[i]);
      } else { nv[i] = dn[i] = f4(0.f); }
    }
    dot = wave_sum(dot) / (float)D;
    float msum = 0.f;
    if (ga.center) {
      float s1 = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) s1 += hsum(dn[i]);
      msum = wave_sum(s1) / (float)D;
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        float* p = dx + (size_t)m * D + 4 * c;
        const float4 d0 = ((dn[i] - nv[i] * dot) - f4(msum)) * rs;
        const float4 g = ga.dx_overwrite ? d0 : *(const float4*)p + d0;        // beta_x = 0: dx is written, not read (no memset by the caller)
        *(float4*)p = g;
        if constexpr (GATE) {
          a_g[i] = a_g[i] + g * load4<T>(gy + (size_t)m * D + 4 * c);
          const float4 d = mul_rn(g, *(const float4*)(cg + 4 * c));
          store4<T>(gdy + (size_t)m * D + 4 * c, d);
          a_b[i] = a_b[i] + make_float4(to_f<T>(from_f<T>(d.x)), to_f<T>(from_f<T>(d.y)), to_f<T>(from_f<T>(d.z)), to_f<T>(from_f<T>(d.w)));
        }
      }
    }
  }
  if constexpr (GATE) __syncthreads();            // every wave is done with the constants: the region becomes reduction scratch
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      *(float4*)(red + (wave * 3 + 0) * D + 4 * c) = a_sh[i];
      *(float4*)(red + (wave * 3 + 1) * D + 4 * c) = a_sc[i];
      *(float4*)(red + (wave * 3 + 2) * D + 4 * c) = a_w[i];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * D; i += 256)
    P[(size_t)blockIdx.x * 3 * D + i] = (red[i] + red[3 * D + i]) + (red[6 * D + i] + red[9 * D + i]);
  if constexpr (GATE) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) { *(float4*)(red + (wave * 2 + 0) * D + 4 * c) = a_g[i]; *(float4*)(red + (wave * 2 + 1) * D + 4 * c) = a_b[i]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < D; i += 256) {
      ga.Pg[(size_t)blockIdx.x * D + i] = (red[i] + red[2 * D + i]) + (red[4 * D + i] + red[6 * D + i]);
      ga.Pb[(size_t)blockIdx.x * D + i] = (red[D + i] + red[3 * D + i]) + (red[5 * D + i] + red[7 * D + i]);
    }
  }
}

// per-sample reduce of the per-workgroup partials, every set optional: P [G][3][D] -> dshift / dscale [B, dmod_ld] and dwb [B, D] (the norm
// weight gradient per sample); Pg [G][D] -> dgate [B, dgate_ld]; Pb [G][D] -> dbb [B, D] (the bias gradient per sample).  One launch for
// what used to be a reduce kernel + a grouped reduce + the first stage of a column sum; dwb and dbb are then summed over the samples by
// ONE group_reduce launch (grid z = 2).
__global__ void mod_partials_reduce_kernel(const float* __restrict__ P, int D, int gps, float* __restrict__ dshift,
                                           float* __restrict__ dscale, int dmod_ld, float* __restrict__ dwb,
                                           const float* __restrict__ Pg = nullptr, float* __restrict__ dgate = nullptr, int dgate_ld = 0,
                                           const float* __restrict__ Pb = nullptr, float* __restrict__ dbb = nullptr) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (d >= D) return;
  // The partials of a sample are added in their fixed order (bitwise equal to the stand-alone gate_bwd path), but LOADED eight at a time: the
  // rolled loop had one load in flight per accumulator and the launch was a chain of dependent L2 round trips (29 us for 31 MB)
  auto sum = [&](const float* p, size_t stride) {
    float s = 0.f;
    int g = 0;
    for (; g + 8 <= gps; g += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = p[(size_t)(g + j) * stride];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; g < gps; ++g) s += p[(size_t)g * stride];
    return s;
  };
  if (P) {
    const float* p = P + (size_t)b * gps * 3 * D + d;
    const float s0 = sum(p, (size_t)3 * D), s1 = sum(p + D, (size_t)3 * D), s2 = sum(p + 2 * D, (size_t)3 * D);
    if (dshift) dshift[(size_t)b * dmod_ld + d] = s0;
    if (dscale) dscale[(size_t)b * dmod_ld + d] = s1;
    dwb[(size_t)b * D + d] = s2;
  }
  if (Pg) dgate[(size_t)b * dgate_ld + d] = sum(Pg + (size_t)b * gps * D + d, (size_t)D);
  if (Pb) dbb[(size_t)b * D + d] = sum(Pb + (size_t)b * gps * D + d, (size_t)D);
}

#define DISPATCH_NCH(D, CALL)                                   \
  switch (((D) / 4 + 63) / 64) {                                \
    case 1: { constexpr int NCH = 1; CALL; } break;             \
    case 2: { constexpr int NCH = 2; CALL; } break;             \
    case 3: { constexpr int NCH = 3; CALL; } break;             \
    case 4: { constexpr int NCH = 4; CALL; } break;             \
    case 5: { constexpr int NCH = 5; CALL; } break;             \
    case 6: { constexpr int NCH = 6; CALL; } break;             \
    case 7: { constexpr int NCH = 7; CALL; } break;             \
    case 8: { constexpr int NCH = 8; CALL; } break;             \
    default: LDMAE_FAIL(LDMAE_ERR_INVALID, "row width D=%d > 2048 unsupported", (D)); }

static int norm_modulate_fwd(int out_dtype, const float* x, const float* w, const float* shift, const float* scale,
                             int mod_ld, void* out, float* rstd, int M, int D, int rows_per_batch, float eps, int center, void* stream) {
  LDMAE_REQUIRE(x && (w || center) && out && M > 0 && D > 0, "rmsnorm_modulate_fwd: null pointer or empty");
  LDMAE_REQUIRE(D % 4 == 0 && (mod_ld % 4 == 0 || (!shift && !scale)), "rmsnorm_modulate_fwd: D=%d mod_ld=%d must be multiples of 4", D, mod_ld);
  LDMAE_REQUIRE(rows_per_batch > 0 && M % rows_per_batch == 0, "rmsnorm_modulate_fwd: M=%d %% rows_per_batch=%d != 0", M, rows_per_batch);
  hipStream_t st = as_stream(stream);
  const unsigned grid = cdiv(M, 16);
  if (out_dtype == LDMAE_BF16 && D % 256 == 0 && M % 16 == 0 && rows_per_batch % 16 == 0 && !center) {
    DISPATCH_NCH(D, hipLaunchKernelGGL((rmsnorm_mod_fwd_kernel<NCH, bf16, true>), dim3(grid), dim3(256), 0, st, x, w, shift, scale, mod_ld, (bf16*)out, rstd, M, D, rows_per_batch, eps, 0));
  } else if (out_dtype == LDMAE_BF16) {
    DISPATCH_NCH(D, hipLaunchKernelGGL((rmsnorm_mod_fwd_kernel<NCH, bf16>), dim3(grid), dim3(256), 0, st, x, w, shift, scale, mod_ld, (bf16*)out, rstd, M, D, rows_per_batch, eps, center));
  } else {
    DISPATCH_NCH(D, hipLaunchKernelGGL((rmsnorm_mod_fwd_kernel<NCH, float>), dim3(grid), dim3(256), 0, st, x, w, shift, scale, mod_ld, (float*)out, rstd, M, D, rows_per_batch, eps, center));
  }
  LDMAE_CHECK_LAUNCH("rmsnorm_modulate_fwd");
  return LDMAE_OK;
}
extern "C" int ldmae_rmsnorm_modulate_fwd(int out_dtype, const float* x, const float* w, const float* shift, const float* scale,
                                          int mod_ld, void* out, float* rstd, int M, int D, int rows_per_batch, float eps, void* stream) {
  return norm_modulate_fwd(out_dtype, x, w, shift, scale, mod_ld, out, rstd, M, D, rows_per_batch, eps, 0, stream);
}
// LayerNorm(elementwise_affine=False) + modulate: the norm of the blocks built with use_rmsnorm=False (lightningdit.py:200-201,257; eps 1e-6).
// rstd [M] = rsqrt(var + eps) (the backward recomputes the row mean from x).
extern "C" int ldmae_layernorm_modulate_fwd(int out_dtype, const float* x, const float* shift, const float* scale, int mod_ld, void* out,
                                            float* rstd, int M, int D, int rows_per_batch, float eps, void* stream) {
  return norm_modulate_fwd(out_dtype, x, nullptr, shift, scale, mod_ld, out, rstd, M, D, rows_per_batch, eps, 1, stream);
}

extern "C" long ldmae_rmsnorm_modulate_bwd_workspace_bytes(int M, int D, int rows_per_batch) {
  const int rw = pick_rows_per_wg(rows_per_batch);
  if (rw == 0) return -1;
  return ((long)(M / rw) * 3 * D + (long)(M / rows_per_batch) * D) * 4;
}

extern "C" long ldmae_colsum_workspace_bytes(int M, int N);
extern "C" int ldmae_colsum(int dtype, const void* X, int ldx, int M, int N, float* out, float beta, float* workspace, void* stream);
extern "C" long ldmae_gate_bwd_workspace_bytes(int M, int D, int rows_per_batch);

static int rmsnorm_modulate_bwd_core(int dtype, const void* dout, const float* x, const float* w, const float* scale, int mod_ld,
                                     const float* rstd, float* dx_accum, float beta_x, float* dshift, float* dscale, int dmod_ld, float* dw,
                                     float beta_w, int M, int D, int rows_per_batch, float* workspace, const GateBwdArgs* gate,
                                     float* dgate, int dgate_ld, float* dbias, void* stream, int center = 0) {
  LDMAE_REQUIRE(dout && x && (w || center) && rstd && dx_accum && (dw || center) && workspace, "rmsnorm_modulate_bwd: null pointer");
  LDMAE_REQUIRE(D % 4 == 0 && M > 0 && rows_per_batch > 0 && M % rows_per_batch == 0, "rmsnorm_modulate_bwd: bad shape M=%d D=%d rpb=%d", M, D, rows_per_batch);
  const int rw = pick_rows_per_wg(rows_per_batch);
  LDMAE_REQUIRE(rw > 0, "rmsnorm_modulate_bwd: rows_per_batch=%d must be a multiple of 4", rows_per_batch);
  hipStream_t st = as_stream(stream);
  const int G = M / rw, B = M / rows_per_batch, gps = rows_per_batch / rw;
  float* P = workspace;
  float* dwb = workspace + (size_t)G * 3 * D;
  float* gws = dwb + (size_t)B * D;                     // gate partials (fused form): [G][D] dgate, [G][D] bias, colsum scratch
  const size_t lds = (size_t)4 * 3 * D * sizeof(float);
  LDMAE_REQUIRE(beta_x == 0.f || beta_x == 1.f, "rmsnorm_modulate_bwd: beta_x must be 0 (write dx) or 1 (accumulate into dx)");
  GateBwdArgs ga{nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, 0};
  if (gate) { ga = *gate; ga.Pg = gws; ga.Pb = gws + (size_t)G * D; }
  ga.dx_overwrite = beta_x == 0.f;
  ga.center = center;
  if (center && !dw) dw = P;             // LayerNorm without affine parameters: no weight gradient; the last reduce (stream-ordered behind the consumers of P) writes its [D] sums there
#define LAUNCH(T, GATE) DISPATCH_NCH(D, hipLaunchKernelGGL((rmsnorm_mod_bwd_kernel<NCH, T, GATE>), dim3(G), dim3(256), lds, st, (const T*)dout, x, w, scale, mod_ld, rstd, dx_accum, P, M, D, rows_per_batch, rw, ga))
#define LAUNCH_FULL(T, GATE) DISPATCH_NCH(D, hipLaunchKernelGGL((rmsnorm_mod_bwd_kernel<NCH, T, GATE, true>), dim3(G), dim3(256), lds, st, (const T*)dout, x, w, scale, mod_ld, rstd, dx_accum, P, M, D, rows_per_batch, rw, ga))
  if (dtype == LDMAE_BF16 && D % 256 == 0 && !center) { if (gate) { LAUNCH_FULL(bf16, true); } else { LAUNCH_FULL(bf16, false); } }
  else if (dtype == LDMAE_BF16) { if (gate) { LAUNCH(bf16, true); } else { LAUNCH(bf16, false); } }
  else { if (gate) { LAUNCH(float, true); } else { LAUNCH(float, false); } }
#undef LAUNCH
#undef LAUNCH_FULL
  LDMAE_CHECK_LAUNCH("rmsnorm_modulate_bwd");
  if (gate) {        // two launches: every per-sample sum, then dw and dbias over the samples together
    float* dbb = ga.Pb + (size_t)G * D;                 // [B][D], behind the gate partials (ldmae_gate_bwd_workspace_bytes reserves it)
    hipLaunchKernelGGL(mod_partials_reduce_kernel, dim3(cdiv(D, 256), B), dim3(256), 0, st, P, D, gps, dshift, dscale, dmod_ld, dwb,
                       (const float*)ga.Pg, dgate, dgate_ld, (const float*)ga.Pb, dbb);
    group_reduce(dwb, D, 1, D, B, dw, D, beta_w, st, dbb, dbias, 0.f);
    LDMAE_CHECK_LAUNCH("rmsnorm_modulate_bwd_gate reduce");
    return LDMAE_OK;
  }
  hipLaunchKernelGGL(mod_partials_reduce_kernel, dim3(cdiv(D, 256), B), dim3(256), 0, st, P, D, gps, dshift, dscale, dmod_ld, dwb);
  group_reduce(dwb, D, 1, D, B, dw, D, beta_w, st);
  LDMAE_CHECK_LAUNCH("rmsnorm_modulate_bwd reduce");
  return LDMAE_OK;
}

extern "C" int ldmae_rmsnorm_modulate_bwd(int dtype, const void* dout, const float* x, const float* w, const float* scale, int mod_ld,
                                          const float* rstd, float* dx_accum, float beta_x, float* dshift, float* dscale, int dmod_ld, float* dw,
                                          float beta_w, int M, int D, int rows_per_batch, float* workspace, void* stream) {
  return rmsnorm_modulate_bwd_core(dtype, dout, x, w, scale, mod_ld, rstd, dx_accum, beta_x, dshift, dscale, dmod_ld, dw, beta_w, M, D, rows_per_batch,
                                   workspace, nullptr, nullptr, 0, nullptr, stream);
}

extern "C" long ldmae_rmsnorm_modulate_bwd_gate_workspace_bytes(int M, int D, int rows_per_batch) {
  const long a = ldmae_rmsnorm_modulate_bwd_workspace_bytes(M, D, rows_per_batch), b = ldmae_gate_bwd_workspace_bytes(M, D, rows_per_batch);
  return a < 0 ? a : a + b;
}
// rmsnorm_modulate_bwd followed by gate_bwd of the updated dx_accum (dy = dx_accum * gate[b] in `dtype`; dgate [B, dgate_ld] = sum_n
// dx_accum * y; dbias [D] = column sums of dy) in one pass over the rows.
extern "C" int ldmae_rmsnorm_modulate_bwd_gate(int dtype, const void* dout, const float* x, const float* w, const float* scale, int mod_ld,
                                               const float* rstd, float* dx_accum, float beta_x, float* dshift, float* dscale, int dmod_ld, float* dw,
                                               float beta_w, const void* y, const float* gate, int gate_ld, void* dy, float* dgate,
                                               int dgate_ld, float* dbias, int M, int D, int rows_per_batch, float* workspace, void* stream) {
  LDMAE_REQUIRE(y && gate && dy && dgate && dbias, "rmsnorm_modulate_bwd_gate: null pointer");
  LDMAE_REQUIRE(gate_ld % 4 == 0, "rmsnorm_modulate_bwd_gate: gate_ld=%d must be a multiple of 4", gate_ld);
  const GateBwdArgs ga{y, gate, gate_ld, dy, nullptr, nullptr, 0, 0};
  return rmsnorm_modulate_bwd_core(dtype, dout, x, w, scale, mod_ld, rstd, dx_accum, beta_x, dshift, dscale, dmod_ld, dw, beta_w, M, D, rows_per_batch,
                                   workspace, &ga, dgate, dgate_ld, dbias, stream);
}
// The LayerNorm(elementwise_affine=False) forms of the two entry points above (use_rmsnorm=False blocks): same arguments without w / dw, same
// workspaces (ldmae_rmsnorm_modulate_bwd(_gate)_workspace_bytes).
extern "C" int ldmae_layernorm_modulate_bwd(int dtype, const void* dout, const float* x, const float* scale, int mod_ld, const float* rstd,
                                            float* dx_accum, float beta_x, float* dshift, float* dscale, int dmod_ld, int M, int D, int rows_per_batch,
                                            float* workspace, void* stream) {
  return rmsnorm_modulate_bwd_core(dtype, dout, x, nullptr, scale, mod_ld, rstd, dx_accum, beta_x, dshift, dscale, dmod_ld, nullptr, 0.f, M, D, rows_per_batch,
                                   workspace, nullptr, nullptr, 0, nullptr, stream, 1);
}
extern "C" int ldmae_layernorm_modulate_bwd_gate(int dtype, const void* dout, const float* x, const float* scale, int mod_ld, const float* rstd,
                                                 float* dx_accum, float beta_x, float* dshift, float* dscale, int dmod_ld, const void* y, const float* gate,
                                                 int gate_ld, void* dy, float* dgate, int dgate_ld, float* dbias, int M, int D, int rows_per_batch,
                                                 float* workspace, void* stream) {
  LDMAE_REQUIRE(y && gate && dy && dgate && dbias, "layernorm_modulate_bwd_gate: null pointer");
  LDMAE_REQUIRE(gate_ld % 4 == 0, "layernorm_modulate_bwd_gate: gate_ld=%d must be a multiple of 4", gate_ld);
  const GateBwdArgs ga{y, gate, gate_ld, dy, nullptr, nullptr, 0, 1};
  return rmsnorm_modulate_bwd_core(dtype, dout, x, nullptr, scale, mod_ld, rstd, dx_accum, beta_x, dshift, dscale, dmod_ld, nullptr, 0.f, M, D, rows_per_batch,
                                   workspace, &ga, dgate, dgate_ld, dbias, stream, 1);
}

// ------------------------------------------------------------------ QK-RMSNorm + RoPE + head-major relayout
// item = (b, n, h); LPR lanes per item, 4 elements per lane.
// (rope_apply: common.h -- shared with the fused qkv epilogue of gemm_nt_common.h)
__device__ __forceinline__ float4 rope_apply_bwd(float4 g, float4 c, float4 s) {
  return make_float4(g.x * c.x + g.y * s.y, g.y * c.y - g.x * s.x, g.z * c.z + g.w * s.w, g.w * c.w - g.z * s.z);
}

template <int LPR, typename T>
__global__ __launch_bounds__(256) void qknorm_rope_fwd_kernel(const T* __restrict__ qkv, const float* __restrict__ wq,
                                                              const float* __restrict__ wk, const float* __restrict__ cosT,
                                                              const float* __restrict__ sinT, T* __restrict__ q, T* __restrict__ k,
                                                              T* __restrict__ v, int B, int N, int H, int hd, float eps) {
  const int sub = threadIdx.x % LPR, c4 = sub * 4;
  const bool act = c4 < hd;
  const long items = (long)B * N * H;
  const long gid = ((long)blockIdx.x * 256 + threadIdx.x) / LPR, gstride = (long)gridDim.x * 256 / LPR;
  const bool plain = cosT == nullptr;    // VMAE attention: head-major relayout only (models_mae.py:133-134)
  const bool norm = wq != nullptr;       // false with tables: RoPE only (use_qknorm=False: q_norm = k_norm = nn.Identity, lightningdit.py:60-61)
  const float4 wqv = (act && norm) ? *(const float4*)(wq + c4) : f4(1.f), wkv = (act && norm) ? *(const float4*)(wk + c4) : f4(1.f);
  for (long it = gid; it < items; it += gstride) {
    const int h = it % H, n = (it / H) % N, b = it / ((long)H * N);
    const T* src = qkv + ((size_t)(b * N + n) * 3 * H + h) * hd + c4;
    const size_t dst = ((size_t)(b * H + h) * N + n) * hd + c4;
    float4 qv = f4(0.f), kv = f4(0.f), vv = f4(0.f), cs = f4(0.f), sn = f4(0.f);
    if (act) {
      qv = load4<T>(src); kv = load4<T>(src + (size_t)H * hd);
      if (v) vv = load4<T>(src + (size_t)2 * H * hd);
      if (!plain) { cs = *(const float4*)(cosT + (size_t)n * hd + c4); sn = *(const float4*)(sinT + (size_t)n * hd + c4); }
    }
    if (plain) {
      if (act) { store4<T>(q + dst, qv); store4<T>(k + dst, kv); store4<T>(v + dst, vv); }
      continue;
    }
    const float rq = norm ? rsqrtf(group_sum<LPR>(hsum(qv * qv)) / (float)hd + eps) : 1.f;
    const float rk = norm ? rsqrtf(group_sum<LPR>(hsum(kv * kv)) / (float)hd + eps) : 1.f;
    if (act) {
      store4<T>(q + dst, rope_apply((qv * rq) * wqv, cs, sn));
      store4<T>(k + dst, rope_apply((kv * rk) * wkv, cs, sn));
      if (v) store4<T>(v + dst, vv);            // v == NULL: attention reads v from the packed qkv itself (ldmae_attention_fwd_pv)
    }
  }
}

// bf16, head dims 64 / 128: 8 elements = 16 B per lane and access (the 4-element form above moves 8 B per lane for bf16: half-width
// requests at 0.54-0.70x the 16-B rate, MI355X_MICROARCH), 8 / 16 lanes per item, two items in flight per lane group.
template <int LPR>
__global__ __launch_bounds__(256) void qknorm_rope_fwd8_kernel(const bf16* __restrict__ qkv, const float* __restrict__ wq, const float* __restrict__ wk,
                                                               const float* __restrict__ cosT, const float* __restrict__ sinT, bf16* __restrict__ q,
                                                               bf16* __restrict__ k, int B, int N, int H, float eps) {
  constexpr int hd = LPR * 8;
  const int sub = threadIdx.x % LPR, c8 = sub * 8;
  const long items = (long)B * N * H;
  const long gid = ((long)blockIdx.x * 256 + threadIdx.x) / LPR, gstride = (long)gridDim.x * 256 / LPR;
  float wqv[8], wkv[8];
  const bool norm = wq != nullptr;       // false: RoPE only (use_qknorm=False); x * 1 * 1 is exact
#pragma unroll
  for (int j = 0; j < 8; ++j) { wqv[j] = norm ? wq[c8 + j] : 1.f; wkv[j] = norm ? wk[c8 + j] : 1.f; }
  auto one = [&](long it, bf16x8 qi, bf16x8 ki) {
    const int h = it % H, n = (it / H) % N, b = it / ((long)H * N);
    const float4 c0 = *(const float4*)(cosT + (size_t)n * hd + c8), c1 = *(const float4*)(cosT + (size_t)n * hd + c8 + 4);
    const float4 s0 = *(const float4*)(sinT + (size_t)n * hd + c8), s1 = *(const float4*)(sinT + (size_t)n * hd + c8 + 4);
    float qv[8], kv[8], sq = 0.f, sk = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { qv[j] = (float)qi[j]; kv[j] = (float)ki[j]; }
    // same association as the 4-element form: (x0^2 + x1^2) + (x2^2 + x3^2) per 4-chunk, chunks summed by the lane-group butterfly
    sq = ((qv[0] * qv[0] + qv[1] * qv[1]) + (qv[2] * qv[2] + qv[3] * qv[3])) + ((qv[4] * qv[4] + qv[5] * qv[5]) + (qv[6] * qv[6] + qv[7] * qv[7]));
    sk = ((kv[0] * kv[0] + kv[1] * kv[1]) + (kv[2] * kv[2] + kv[3] * kv[3])) + ((kv[4] * kv[4] + kv[5] * kv[5]) + (kv[6] * kv[6] + kv[7] * kv[7]));
    const float rq = norm ? rsqrtf(group_sum<LPR>(sq) / (float)hd + eps) : 1.f, rk = norm ? rsqrtf(group_sum<LPR>(sk) / (float)hd + eps) : 1.f;
    const float4 a0 = rope_apply(make_float4(qv[0] * rq * wqv[0], qv[1] * rq * wqv[1], qv[2] * rq * wqv[2], qv[3] * rq * wqv[3]), c0, s0);
    const float4 a1 = rope_apply(make_float4(qv[4] * rq * wqv[4], qv[5] * rq * wqv[5], qv[6] * rq * wqv[6], qv[7] * rq * wqv[7]), c1, s1);
    const float4 b0 = rope_apply(make_float4(kv[0] * rk * wkv[0], kv[1] * rk * wkv[1], kv[2] * rk * wkv[2], kv[3] * rk * wkv[3]), c0, s0);
    const float4 b1 = rope_apply(make_float4(kv[4] * rk * wkv[4], kv[5] * rk * wkv[5], kv[6] * rk * wkv[6], kv[7] * rk * wkv[7]), c1, s1);
    bf16x8 qo, ko;
    qo[0] = (bf16)a0.x; qo[1] = (bf16)a0.y; qo[2] = (bf16)a0.z; qo[3] = (bf16)a0.w; qo[4] = (bf16)a1.x; qo[5] = (bf16)a1.y; qo[6] = (bf16)a1.z; qo[7] = (bf16)a1.w;
    ko[0] = (bf16)b0.x; ko[1] = (bf16)b0.y; ko[2] = (bf16)b0.z; ko[3] = (bf16)b0.w; ko[4] = (bf16)b1.x; ko[5] = (bf16)b1.y; ko[6] = (bf16)b1.z; ko[7] = (bf16)b1.w;
    const size_t dst = ((size_t)(b * H + h) * N + n) * hd + c8;
    __builtin_nontemporal_store(qo, (bf16x8*)(q + dst));
    __builtin_nontemporal_store(ko, (bf16x8*)(k + dst));
  };
  auto src_of = [&](long it) { const int h = it % H; const long bn = it / H; return qkv + ((size_t)bn * 3 * H + h) * hd + c8; };
  long it = gid;
  for (; it + gstride < items; it += 2 * gstride) {          // two items in flight
    const bf16* p0 = src_of(it);
    const bf16* p1 = src_of(it + gstride);
    const bf16x8 q0 = __builtin_nontemporal_load((const bf16x8*)p0), k0 = __builtin_nontemporal_load((const bf16x8*)(p0 + (size_t)H * hd));
    const bf16x8 q1 = __builtin_nontemporal_load((const bf16x8*)p1), k1 = __builtin_nontemporal_load((const bf16x8*)(p1 + (size_t)H * hd));
    one(it, q0, k0);
    one(it + gstride, q1, k1);
  }
  if (it < items) {
    const bf16* p0 = src_of(it);
    one(it, *(const bf16x8*)p0, *(const bf16x8*)(p0 + (size_t)H * hd));
  }
}

// Head dims whose 4-element chunk count is not a power of two (LightningDiT-XL: 72 -> 18 chunks): the lane-group form above rounds the
// group up to 32 lanes and leaves 14 of them idle (2.0 TB/s at hd = 72).  Here the workgroup's threads are packed densely -- thread t
// serves chunk t % cpi of item t / cpi, 14 items x 18 chunks = 252 of 256 threads at hd = 72, consecutive threads on consecutive
// bytes -- and the two row sums go through LDS (every thread adds its item's cpi partials in the same order).  Same arithmetic per element.
template <typename T>
__global__ __launch_bounds__(256) void qknorm_rope_fwd_dense_kernel(const T* __restrict__ qkv, const float* __restrict__ wq,
                                                                    const float* __restrict__ wk, const float* __restrict__ cosT,
                                                                    const float* __restrict__ sinT, T* __restrict__ q, T* __restrict__ k,
                                                                    T* __restrict__ v, int B, int N, int H, int hd, float eps) {
  __shared__ float2 red[256];
  const int cpi = hd >> 2, ipw = 256 / cpi, t = threadIdx.x, li = t / cpi, c4 = (t % cpi) * 4;
  const bool lane_ok = li < ipw;
  const long items = (long)B * N * H;
  const float4 wqv = *(const float4*)(wq + c4), wkv = *(const float4*)(wk + c4);
  for (long base = (long)blockIdx.x * ipw; base < items; base += (long)gridDim.x * ipw) {
    const long it = base + li;
    const bool act = lane_ok && it < items;
    // item order: tiles of 8 tokens x H heads, token fastest inside a head -- consecutive items then write consecutive rows of ONE head
    // (8 rows of 144 B = 9 whole lines at hd = 72; with the head fastest every 144-B row is a partial-line write of its own)
    const long tile = it / (8 * H);
    const int rr = (int)(it % (8 * H)), h = rr >> 3;
    const long tok = tile * 8 + (rr & 7);
    const int n = (int)(tok % N), b = (int)(tok / N);
    const T* src = qkv + ((size_t)(b * N + n) * 3 * H + h) * hd + c4;
    const size_t dst = ((size_t)(b * H + h) * N + n) * hd + c4;
    float4 qv = f4(0.f), kv = f4(0.f), vv = f4(0.f), cs = f4(0.f), sn = f4(0.f);
    if (act) {
      qv = load4<T>(src); kv = load4<T>(src + (size_t)H * hd);
      if (v) vv = load4<T>(src + (size_t)2 * H * hd);
      cs = *(const float4*)(cosT + (size_t)n * hd + c4); sn = *(const float4*)(sinT + (size_t)n * hd + c4);
    }
    red[t] = make_float2(hsum(qv * qv), hsum(kv * kv));
    __syncthreads();
    float sq = 0.f, sk = 0.f;
    if (lane_ok) {
      for (int j = 0; j < cpi; ++j) { const float2 p = red[li * cpi + j]; sq += p.x; sk += p.y; }
    }
    __syncthreads();
    if (act) {
      const float rq = rsqrtf(sq / (float)hd + eps), rk = rsqrtf(sk / (float)hd + eps);
      store4<T>(q + dst, rope_apply((qv * rq) * wqv, cs, sn));
      store4<T>(k + dst, rope_apply((kv * rk) * wkv, cs, sn));
      if (v) store4<T>(v + dst, vv);
    }
  }
}

// The bf16 form of the dense kernel with 16-B accesses (8 elements per thread; hd = 72: 9 chunks per item, 28 items x 9 = 252 of 256 threads):
// the 4-element form above moves 8 B per lane -- half-width requests, 3.4 TB/s on the XL/1 forward (4.8 GB per launch at CFG batch 512).
// Same arithmetic and the same association of the row sums (per 4-element chunk, chunks added in order): bitwise the same outputs.
__global__ __launch_bounds__(256) void qknorm_rope_fwd_dense8_kernel(const bf16* __restrict__ qkv, const float* __restrict__ wq,
                                                                     const float* __restrict__ wk, const float* __restrict__ cosT,
                                                                     const float* __restrict__ sinT, bf16* __restrict__ q, bf16* __restrict__ k,
                                                                     bf16* __restrict__ v, int B, int N, int H, int hd, float eps) {
  __shared__ float2 red[512];
  const int cpi = hd >> 3, ipw = 256 / cpi, t = threadIdx.x, li = t / cpi, ci = t % cpi, c8 = ci * 8;
  const bool lane_ok = li < ipw;
  const long items = (long)B * N * H;
  float wqv[8], wkv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { wqv[j] = wq[c8 + j]; wkv[j] = wk[c8 + j]; }
  for (long base = (long)blockIdx.x * ipw; base < items; base += (long)gridDim.x * ipw) {
    const long it = base + li;
    const bool act = lane_ok && it < items;
    const long tile = it / (8 * H);                       // item order as in the 4-element form: 8 tokens x H heads, token fastest inside a head
    const int rr = (int)(it % (8 * H)), h = rr >> 3;
    const long tok = tile * 8 + (rr & 7);
    const int n = (int)(tok % N), b = (int)(tok / N);
    const bf16* src = qkv + ((size_t)(b * N + n) * 3 * H + h) * hd + c8;
    const size_t dst = ((size_t)(b * H + h) * N + n) * hd + c8;
    float qv[8], kv[8];
    bf16x8 vi;
    float4 c0 = f4(0.f), c1 = f4(0.f), s0 = f4(0.f), s1 = f4(0.f);
#pragma unroll
    for (int j = 0; j < 8; ++j) { qv[j] = 0.f; kv[j] = 0.f; }
    if (act) {
      const bf16x8 qi = __builtin_nontemporal_load((const bf16x8*)src), ki = __builtin_nontemporal_load((const bf16x8*)(src + (size_t)H * hd));
      if (v) vi = __builtin_nontemporal_load((const bf16x8*)(src + (size_t)2 * H * hd));
#pragma unroll
      for (int j = 0; j < 8; ++j) { qv[j] = (float)qi[j]; kv[j] = (float)ki[j]; }
      c0 = *(const float4*)(cosT + (size_t)n * hd + c8); c1 = *(const float4*)(cosT + (size_t)n * hd + c8 + 4);
      s0 = *(const float4*)(sinT + (size_t)n * hd + c8); s1 = *(const float4*)(sinT + (size_t)n * hd + c8 + 4);
    }
    red[2 * t] = make_float2((qv[0] * qv[0] + qv[1] * qv[1]) + (qv[2] * qv[2] + qv[3] * qv[3]), (kv[0] * kv[0] + kv[1] * kv[1]) + (kv[2] * kv[2] + kv[3] * kv[3]));
    red[2 * t + 1] = make_float2((qv[4] * qv[4] + qv[5] * qv[5]) + (qv[6] * qv[6] + qv[7] * qv[7]), (kv[4] * kv[4] + kv[5] * kv[5]) + (kv[6] * kv[6] + kv[7] * kv[7]));
    __syncthreads();
    float sq = 0.f, sk = 0.f;
    if (lane_ok) {
      for (int j = 0; j < 2 * cpi; ++j) { const float2 p = red[2 * li * cpi + j]; sq += p.x; sk += p.y; }
    }
    __syncthreads();
    if (act) {
      const float rq = rsqrtf(sq / (float)hd + eps), rk = rsqrtf(sk / (float)hd + eps);
      const float4 a0 = rope_apply(make_float4(qv[0] * rq * wqv[0], qv[1] * rq * wqv[1], qv[2] * rq * wqv[2], qv[3] * rq * wqv[3]), c0, s0);
      const float4 a1 = rope_apply(make_float4(qv[4] * rq * wqv[4], qv[5] * rq * wqv[5], qv[6] * rq * wqv[6], qv[7] * rq * wqv[7]), c1, s1);
      const float4 b0 = rope_apply(make_float4(kv[0] * rk * wkv[0], kv[1] * rk * wkv[1], kv[2] * rk * wkv[2], kv[3] * rk * wkv[3]), c0, s0);
      const float4 b1 = rope_apply(make_float4(kv[4] * rk * wkv[4], kv[5] * rk * wkv[5], kv[6] * rk * wkv[6], kv[7] * rk * wkv[7]), c1, s1);
      bf16x8 qo, ko;
      qo[0] = (bf16)a0.x; qo[1] = (bf16)a0.y; qo[2] = (bf16)a0.z; qo[3] = (bf16)a0.w; qo[4] = (bf16)a1.x; qo[5] = (bf16)a1.y; qo[6] = (bf16)a1.z; qo[7] = (bf16)a1.w;
      ko[0] = (bf16)b0.x; ko[1] = (bf16)b0.y; ko[2] = (bf16)b0.z; ko[3] = (bf16)b0.w; ko[4] = (bf16)b1.x; ko[5] = (bf16)b1.y; ko[6] = (bf16)b1.z; ko[7] = (bf16)b1.w;
      __builtin_nontemporal_store(qo, (bf16x8*)(q + dst));
      __builtin_nontemporal_store(ko, (bf16x8*)(k + dst));
      if (v) __builtin_nontemporal_store(vi, (bf16x8*)(v + dst));
    }
  }
}

template <int LPR, typename T>
__global__ __launch_bounds__(256) void qknorm_rope_bwd_kernel(const T* __restrict__ dq, const T* __restrict__ dk, const T* __restrict__ dv,
                                                              const T* __restrict__ qkv, const float* __restrict__ wq,
                                                              const float* __restrict__ wk, const float* __restrict__ cosT,
                                                              const float* __restrict__ sinT, T* __restrict__ dqkv, float* __restrict__ P,
                                                              float* __restrict__ Pb, int B, int N, int H, int hd, float eps) {
  // Pb (optional): bias gradient of the qkv Linear = column sums of dqkv AS STORED.  The host makes the group stride a multiple of
  // H, so a lane group keeps one head for all its items and its lanes own fixed columns: three float4 running sums per lane,
  // written as row `gid` of Pb[groups][3*hd] (q | k | v of head gid % H); the host sums the rows of each head.
  __shared__ float4 red[256][2];
  const int sub = threadIdx.x % LPR, c4 = sub * 4;
  const bool act = c4 < hd;
  const long items = (long)B * N * H;
  const long gid = ((long)blockIdx.x * 256 + threadIdx.x) / LPR, gstride = (long)gridDim.x * 256 / LPR;
  const bool plain = cosT == nullptr;
  const bool norm = wq != nullptr;       // false with tables: RoPE adjoint only (use_qknorm=False); the pre-norm rows are then not read
  const float4 wqv = (act && norm) ? *(const float4*)(wq + c4) : f4(1.f), wkv = (act && norm) ? *(const float4*)(wk + c4) : f4(1.f);
  float4 awq = f4(0.f), awk = f4(0.f), bq = f4(0.f), bk = f4(0.f), bv = f4(0.f);
  auto rnd = [](float4 v) { return make_float4(to_f<T>(from_f<T>(v.x)), to_f<T>(from_f<T>(v.y)), to_f<T>(from_f<T>(v.z)), to_f<T>(from_f<T>(v.w))); };
  for (long it = gid; it < items; it += gstride) {
    const int h = it % H, n = (it / H) % N, b = it / ((long)H * N);
    const size_t so = ((size_t)(b * N + n) * 3 * H + h) * hd + c4;
    const size_t go = ((size_t)(b * H + h) * N + n) * hd + c4;
    if (plain) {
      if (act) {
        const float4 a = load4<T>(dq + go), bb = load4<T>(dk + go), c = load4<T>(dv + go);
        store4<T>(dqkv + so, a);
        store4<T>(dqkv + so + (size_t)H * hd, bb);
        store4<T>(dqkv + so + (size_t)2 * H * hd, c);
        if (Pb) { bq = bq + a; bk = bk + bb; bv = bv + c; }
      }
      continue;
    }
    float4 qv = f4(0.f), kv = f4(0.f), gq = f4(0.f), gk = f4(0.f), gv = f4(0.f), cs = f4(0.f), sn = f4(0.f);
    if (act) {
      if (norm) { qv = load4<T>(qkv + so); kv = load4<T>(qkv + so + (size_t)H * hd); }
      gq = load4<T>(dq + go); gk = load4<T>(dk + go);
      // dv == NULL: the attention backward has already written dv into the v slot of dqkv (ldmae_attention_bwd_pv); it is only
      // read back for the bias-gradient column sums
      if (dv) gv = load4<T>(dv + go); else if (Pb) gv = load4<T>(dqkv + so + (size_t)2 * H * hd);
      cs = *(const float4*)(cosT + (size_t)n * hd + c4); sn = *(const float4*)(sinT + (size_t)n * hd + c4);
    }
    const float rq = norm ? rsqrtf(group_sum<LPR>(hsum(qv * qv)) / (float)hd + eps) : 1.f;      // RoPE only: n = 0, m = 0, r = 1 -> the
    const float rk = norm ? rsqrtf(group_sum<LPR>(hsum(kv * kv)) / (float)hd + eps) : 1.f;      // stored rows are exactly rope^T(g)
    const float4 nq = qv * rq, nk = kv * rk;
    const float4 tq = rope_apply_bwd(gq, cs, sn), tk = rope_apply_bwd(gk, cs, sn);   // grad wrt (n * w)
    awq = awq + tq * nq; awk = awk + tk * nk;
    const float4 dnq = tq * wqv, dnk = tk * wkv;
    const float mq = group_sum<LPR>(hsum(dnq * nq)) / (float)hd, mk = group_sum<LPR>(hsum(dnk * nk)) / (float)hd;
    if (act) {
      const float4 oq = (dnq - nq * mq) * rq, ok = (dnk - nk * mk) * rk;
      store4<T>(dqkv + so, oq);
      store4<T>(dqkv + so + (size_t)H * hd, ok);
      if (dv) store4<T>(dqkv + so + (size_t)2 * H * hd, gv);
      if (Pb) { bq = bq + rnd(oq); bk = bk + rnd(ok); bv = bv + gv; }
    }
  }
  if (Pb && act) {
    float* row = Pb + (size_t)gid * 3 * hd + c4;
    *(float4*)row = bq; *(float4*)(row + hd) = bk; *(float4*)(row + 2 * hd) = bv;
  }
  if (!norm) return;
  red[threadIdx.x][0] = awq; red[threadIdx.x][1] = awk;
  __syncthreads();
  if (threadIdx.x < LPR && act) {
    float4 sq = f4(0.f), sk = f4(0.f);
    for (int t = threadIdx.x; t < 256; t += LPR) { sq = sq + red[t][0]; sk = sk + red[t][1]; }
    *(float4*)(P + (size_t)blockIdx.x * 2 * hd + c4) = sq;
    *(float4*)(P + (size_t)blockIdx.x * 2 * hd + hd + c4) = sk;
  }
}

static unsigned qk_grid(long items, int lpr) {
  long wg = (items * lpr + 255) / 256;
  return (unsigned)(wg < 2048 ? (wg > 0 ? wg : 1) : 2048);
}

extern "C" int ldmae_qknorm_rope_fwd(int dtype, const void* qkv, const float* wq, const float* wk, const float* cos, const float* sin,
                                     void* q, void* k, void* v, int B, int N, int H, int hd, float eps, void* stream) {
  LDMAE_REQUIRE(qkv && q && k && (v || cos), "qknorm_rope_fwd: null pointer (v may be NULL only with RoPE: v then stays in the packed qkv)");
  LDMAE_REQUIRE(!wq == !wk && !cos == !sin && (!wq || cos), "qknorm_rope_fwd: pass wq/wk/cos/sin (QK-norm + RoPE), cos/sin alone (RoPE only: use_qknorm=False), or none (plain head-major relayout)");
  LDMAE_REQUIRE(hd % 8 == 0 && hd <= 128 && B > 0 && N > 0 && H > 0, "qknorm_rope_fwd: head_dim=%d must be a multiple of 8 and <= 128", hd);
  hipStream_t st = as_stream(stream);
  const long items = (long)B * N * H;
#define QK_FWD(LPR, T) hipLaunchKernelGGL((qknorm_rope_fwd_kernel<LPR, T>), dim3(qk_grid(items, LPR)), dim3(256), 0, st, (const T*)qkv, wq, wk, cos, sin, (T*)q, (T*)k, (T*)v, B, N, H, hd, eps)
  const int cpi = hd / 4;
  if (wq && hd > 64 && (cpi & (cpi - 1)) != 0 && N % 8 == 0) {      // e.g. hd = 72: densely packed threads instead of 32-lane groups with 18 busy lanes
    const long wgs = (items + 256 / cpi - 1) / (256 / cpi);
    const unsigned grid = (unsigned)(wgs < 4096 ? wgs : 4096);
    if (dtype == LDMAE_BF16) {
      const int cpi8 = hd / 8;
      const long wgs8 = (items + 256 / cpi8 - 1) / (256 / cpi8);
      hipLaunchKernelGGL(qknorm_rope_fwd_dense8_kernel, dim3((unsigned)(wgs8 < 4096 ? wgs8 : 4096)), dim3(256), 0, st, (const bf16*)qkv, wq, wk, cos, sin, (bf16*)q, (bf16*)k, (bf16*)v, B, N, H, hd, eps);
    }
    else hipLaunchKernelGGL(qknorm_rope_fwd_dense_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)qkv, wq, wk, cos, sin, (float*)q, (float*)k, (float*)v, B, N, H, hd, eps);
  }
  else if (dtype == LDMAE_BF16 && cos && !v && (hd == 64 || hd == 128) && items % (256 / (hd / 8)) == 0) {
    // the LightningDiT block's form (v stays in the packed qkv): 16-B accesses, whole lane groups
    const unsigned grid = qk_grid(items, hd / 8);
    if (hd == 64) hipLaunchKernelGGL(qknorm_rope_fwd8_kernel<8>, dim3(grid), dim3(256), 0, st, (const bf16*)qkv, wq, wk, cos, sin, (bf16*)q, (bf16*)k, B, N, H, eps);
    else hipLaunchKernelGGL(qknorm_rope_fwd8_kernel<16>, dim3(grid), dim3(256), 0, st, (const bf16*)qkv, wq, wk, cos, sin, (bf16*)q, (bf16*)k, B, N, H, eps);
  }
  else if (hd <= 64) { if (dtype == LDMAE_BF16) QK_FWD(16, bf16); else QK_FWD(16, float); }
  else { if (dtype == LDMAE_BF16) QK_FWD(32, bf16); else QK_FWD(32, float); }
#undef QK_FWD
  LDMAE_CHECK_LAUNCH("qknorm_rope_fwd");
  return LDMAE_OK;
}

// backward grid: like qk_grid, rounded down so that the lane-group stride (grid * 256 / lpr) is a multiple of H -- every lane group
// then serves ONE head (needed for the fused bias-gradient sums)
static unsigned qk_bwd_grid(long items, int lpr, int H) {
  unsigned g = qk_grid(items, lpr);
  const int gpw = 256 / lpr;
  int a = H, bb = gpw;
  while (bb) { const int t = a % bb; a = bb; bb = t; }       // gcd(H, groups per workgroup)
  const unsigned m = (unsigned)(H / a);
  return g >= m ? g / m * m : m;
}
extern "C" long ldmae_colsum_workspace_bytes(int M, int N);
extern "C" int ldmae_colsum(int dtype, const void* X, int ldx, int M, int N, float* out, float beta, float* workspace, void* stream);
extern "C" long ldmae_qknorm_rope_bwd_workspace_bytes(int B, int N, int H, int hd) {
  const int lpr = hd <= 64 ? 16 : 32;
  const long grid = qk_bwd_grid((long)B * N * H, lpr, H), groups = grid * (256 / lpr);
  // dwq/dwk partials + bias partial rows [groups][3*hd] + their column-sum scratch
  return grid * 2 * hd * 4 + groups * 3 * hd * 4 + ldmae_colsum_workspace_bytes((int)(groups / H), 3 * H * hd);
}

extern "C" int ldmae_qknorm_rope_bwd(int dtype, const void* dq, const void* dk, const void* dv, const void* qkv, const float* wq,
                                     const float* wk, const float* cos, const float* sin, void* dqkv, float* dwq, float* dwk, float beta_w,
                                     float* dbias_hqd, int B, int N, int H, int hd, float eps, float* workspace, void* stream) {
  LDMAE_REQUIRE(dq && dk && dqkv && (dv || cos), "qknorm_rope_bwd: null pointer (dv may be NULL only with RoPE: dv is then already in dqkv)");
  LDMAE_REQUIRE(!wq == !wk && !cos == !sin && (!wq || (qkv && cos && dwq && dwk)) && (!cos || workspace),
                "qknorm_rope_bwd: pass all norm/rope arguments, cos/sin without wq/wk (RoPE adjoint only: use_qknorm=False), or none of them (plain relayout)");
  LDMAE_REQUIRE(!dbias_hqd || workspace, "qknorm_rope_bwd: dbias requested without workspace");
  LDMAE_REQUIRE(hd % 8 == 0 && hd <= 128 && B > 0 && N > 0 && H > 0, "qknorm_rope_bwd: head_dim=%d must be a multiple of 8 and <= 128", hd);
  hipStream_t st = as_stream(stream);
  const long items = (long)B * N * H;
  const int lpr = hd <= 64 ? 16 : 32;
  const unsigned grid = qk_bwd_grid(items, lpr, H);
  const long groups = (long)grid * (256 / lpr);
  float* Pb = dbias_hqd ? workspace + (size_t)grid * 2 * hd : nullptr;
#define QK_BWD(LPR, T) hipLaunchKernelGGL((qknorm_rope_bwd_kernel<LPR, T>), dim3(grid), dim3(256), 0, st, (const T*)dq, (const T*)dk, (const T*)dv, (const T*)qkv, wq, wk, cos, sin, (T*)dqkv, workspace, Pb, B, N, H, hd, eps)
  if (hd <= 64) { if (dtype == LDMAE_BF16) QK_BWD(16, bf16); else QK_BWD(16, float); }
  else { if (dtype == LDMAE_BF16) QK_BWD(32, bf16); else QK_BWD(32, float); }
#undef QK_BWD
  LDMAE_CHECK_LAUNCH("qknorm_rope_bwd");
  if (wq) {
    group_reduce(workspace, 2 * hd, 1, hd, grid, dwq, hd, beta_w, st);
    group_reduce(workspace + hd, 2 * hd, 1, hd, grid, dwk, hd, beta_w, st);
    LDMAE_CHECK_LAUNCH("qknorm_rope_bwd reduce");
  }
  // rows g of Pb belong to head g % H: as a [groups/H][H*3*hd] matrix its column sums are the bias gradient in (head, q|k|v, d) order
  if (dbias_hqd) return ldmae_colsum(LDMAE_F32, Pb, 3 * H * hd, (int)(groups / H), 3 * H * hd, dbias_hqd, 0.f, Pb + (size_t)groups * 3 * hd, stream);
  return LDMAE_OK;
}

// ------------------------------------------------------------------ standalone RoPE (VisionRotaryEmbeddingFast.forward, pos_embed.py:135)
// out[r, :] = t[r, :] * cos[r % N, :] + rotate_half(t[r, :]) * sin[r % N, :]  (rotate_half: (x0, x1) -> (-x1, x0) per pair);
// transposed = 1: the adjoint (its backward).  The block path never comes here (RoPE is fused into ldmae_qknorm_rope_*).
template <typename T>
__global__ void rope_kernel(const T* __restrict__ t, const float* __restrict__ cosT, const float* __restrict__ sinT, T* __restrict__ out,
                            long rows, int N, int hd, int transposed) {
  const long n4 = rows * (hd / 4), per = hd / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const long r = i / per;
    const int c4 = (int)(i % per) * 4, n = (int)(r % N);
    const float4 v = load4<T>(t + r * hd + c4);
    const float4 cs = *(const float4*)(cosT + (size_t)n * hd + c4), sn = *(const float4*)(sinT + (size_t)n * hd + c4);
    store4<T>(out + r * hd + c4, transposed ? rope_apply_bwd(v, cs, sn) : rope_apply(v, cs, sn));
  }
}

extern "C" int ldmae_rope(int dtype, const void* t, const float* cos, const float* sin, void* out, long rows, int N, int hd, int transposed,
                          void* stream) {
  LDMAE_REQUIRE(t && cos && sin && out && rows > 0 && N > 0, "rope: null pointer or empty input");
  LDMAE_REQUIRE(hd % 4 == 0 && rows % N == 0, "rope: head_dim=%d must be a multiple of 4 and rows=%ld a multiple of N=%d", hd, rows, N);
  LDMAE_REQUIRE(dtype == LDMAE_F32 || dtype == LDMAE_BF16, "rope: bad dtype %d", dtype);
  const long n4 = rows * (hd / 4);
  const unsigned grid = (unsigned)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
  if (dtype == LDMAE_BF16) hipLaunchKernelGGL(rope_kernel<bf16>, dim3(grid), dim3(256), 0, as_stream(stream), (const bf16*)t, cos, sin, (bf16*)out, rows, N, hd, transposed);
  else hipLaunchKernelGGL(rope_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), (const float*)t, cos, sin, (float*)out, rows, N, hd, transposed);
  LDMAE_CHECK_LAUNCH("rope");
  return LDMAE_OK;
}

// ------------------------------------------------------------------ SwiGLU
template <typename T>
__global__ void swiglu_fwd_kernel(const T* __restrict__ h12, T* __restrict__ hid, long M, int Hs) {
  const long n8 = M * Hs / 8, per_row = Hs / 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const long m = i / per_row, c = (i % per_row) * 8;
    float a[8], b[8], o[8];
    Vec8<T>::load(h12 + m * 2 * Hs + c, a);
    Vec8<T>::load(h12 + m * 2 * Hs + Hs + c, b);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = a[j] * fast_sigmoid(a[j]) * b[j];
    Vec8<T>::store(hid + m * Hs + c, o);
  }
}
template <typename T>
__global__ void swiglu_bwd_kernel(const T* __restrict__ dhid, const T* __restrict__ h12, T* __restrict__ dh12, long M, int Hs) {
  const long n8 = M * Hs / 8, per_row = Hs / 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const long m = i / per_row, c = (i % per_row) * 8;
    float a[8], b[8], g[8], da[8], db[8];
    Vec8<T>::load(h12 + m * 2 * Hs + c, a);
    Vec8<T>::load(h12 + m * 2 * Hs + Hs + c, b);
    Vec8<T>::load(dhid + m * Hs + c, g);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float s = fast_sigmoid(a[j]);
      da[j] = g[j] * b[j] * s * (1.f + a[j] * (1.f - s));
      db[j] = g[j] * a[j] * s;
    }
    Vec8<T>::store(dh12 + m * 2 * Hs + c, da);
    Vec8<T>::store(dh12 + m * 2 * Hs + Hs + c, db);
  }
}
static unsigned ew_grid(long n) { long g = (n + 255) / 256; return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g)); }

extern "C" int ldmae_swiglu_fwd(int dtype, const void* h12, void* hid, int M, int Hs, void* stream) {
  LDMAE_REQUIRE(h12 && hid && M > 0 && Hs > 0 && Hs % 8 == 0, "swiglu_fwd: bad arguments (Hs=%d must be a multiple of 8)", Hs);
  const unsigned grid = ew_grid((long)M * Hs / 8);
  if (dtype == LDMAE_BF16) hipLaunchKernelGGL(swiglu_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, as_stream(stream), (const bf16*)h12, (bf16*)hid, (long)M, Hs);
  else hipLaunchKernelGGL(swiglu_fwd_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), (const float*)h12, (float*)hid, (long)M, Hs);
  LDMAE_CHECK_LAUNCH("swiglu_fwd");
  return LDMAE_OK;
}
extern "C" int ldmae_swiglu_bwd(int dtype, const void* dhid, const void* h12, void* dh12, int M, int Hs, void* stream) {
  LDMAE_REQUIRE(dhid && h12 && dh12 && M > 0 && Hs > 0 && Hs % 8 == 0, "swiglu_bwd: bad arguments (Hs=%d must be a multiple of 8)", Hs);
  const unsigned grid = ew_grid((long)M * Hs / 8);
  if (dtype == LDMAE_BF16) hipLaunchKernelGGL(swiglu_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, as_stream(stream), (const bf16*)dhid, (const bf16*)h12, (bf16*)dh12, (long)M, Hs);
  else hipLaunchKernelGGL(swiglu_bwd_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), (const float*)dhid, (const float*)h12, (float*)dh12, (long)M, Hs);
  LDMAE_CHECK_LAUNCH("swiglu_bwd");
  return LDMAE_OK;
}

// ------------------------------------------------------------------ gated residual backward
template <int NCH, typename T>
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ dxo, const T* __restrict__ y, const float* __restrict__ gate,
                                                       int gate_ld, T* __restrict__ dy, float* __restrict__ P, float* __restrict__ Pb,
                                                       int M, int D, int rpb, int rows_per_wg) {
  extern __shared__ float red[];   // [4][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nch = D >> 2;
  const int m_base = blockIdx.x * rows_per_wg, b = m_base / rpb;
  float4 gv[NCH], acc[NCH], accb[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    gv[i] = (c < nch && gate) ? *(const float4*)(gate + (size_t)b * gate_ld + 4 * c) : f4(1.f);
    acc[i] = f4(0.f); accb[i] = f4(0.f);
  }
  for (int r = wave; r < rows_per_wg; r += 4) {
    const int m = m_base + r;
    if (m >= M) break;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        const float4 g = *(const float4*)(dxo + (size_t)m * D + 4 * c);
        if (P) acc[i] = acc[i] + g * load4<T>(y + (size_t)m * D + 4 * c);
        const float4 d = mul_rn(g, gv[i]);       // never contracted into the bias sum below: that sum is over the values AS STORED
        store4<T>(dy + (size_t)m * D + 4 * c, d);
        // bias gradient of the Linear that produced the branch = column sums of dy AS STORED (rounded to T: what the weight-
        // gradient GEMM reads), formed here while the values are in registers instead of inside the TN GEMM
        if (Pb) accb[i] = accb[i] + make_float4(to_f<T>(from_f<T>(d.x)), to_f<T>(from_f<T>(d.y)), to_f<T>(from_f<T>(d.z)), to_f<T>(from_f<T>(d.w)));
      }
    }
  }
  if (P) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) { const int c = lane + 64 * i; if (c < nch) *(float4*)(red + wave * D + 4 * c) = acc[i]; }
    __syncthreads();
    for (int i = threadIdx.x; i < D; i += 256) P[(size_t)blockIdx.x * D + i] = (red[i] + red[D + i]) + (red[2 * D + i] + red[3 * D + i]);
    __syncthreads();
  }
  if (Pb) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) { const int c = lane + 64 * i; if (c < nch) *(float4*)(red + wave * D + 4 * c) = accb[i]; }
    __syncthreads();
    for (int i = threadIdx.x; i < D; i += 256) Pb[(size_t)blockIdx.x * D + i] = (red[i] + red[D + i]) + (red[2 * D + i] + red[3 * D + i]);
  }
}
extern "C" long ldmae_colsum_workspace_bytes(int M, int N);
extern "C" int ldmae_colsum(int dtype, const void* X, int ldx, int M, int N, float* out, float beta, float* workspace, void* stream);
extern "C" long ldmae_gate_bwd_workspace_bytes(int M, int D, int rows_per_batch) {
  const int rw = pick_rows_per_wg(rows_per_batch);
  if (rw <= 0) return 0;
  const int G = M / rw;                                // dgate partials + bias-gradient partials + max(column-sum scratch, per-sample bias sums)
  const long tail = ldmae_colsum_workspace_bytes(G, D), dbb = (long)(M / rows_per_batch) * D * 4;
  return 2L * G * D * 4 + (tail > dbb ? tail : dbb);
}
extern "C" int ldmae_gate_bwd(int dtype, const float* dxout, const void* y, const float* gate, int gate_ld, void* dy, float* dgate,
                              int dgate_ld, float* dbias, int M, int D, int rows_per_batch, float* workspace, void* stream) {
  LDMAE_REQUIRE(dxout && dy && M > 0 && D % 4 == 0, "gate_bwd: null pointer or D=%d not a multiple of 4", D);
  LDMAE_REQUIRE(rows_per_batch > 0 && M % rows_per_batch == 0, "gate_bwd: M=%d %% rows_per_batch=%d != 0", M, rows_per_batch);
  LDMAE_REQUIRE(!dgate || (y && gate && workspace), "gate_bwd: dgate requested without y/gate/workspace");
  LDMAE_REQUIRE(!dbias || workspace, "gate_bwd: dbias requested without workspace");
  const int rw = pick_rows_per_wg(rows_per_batch);
  LDMAE_REQUIRE(rw > 0, "gate_bwd: rows_per_batch=%d must be a multiple of 4", rows_per_batch);
  hipStream_t st = as_stream(stream);
  const int G = M / rw;
  float* P = dgate ? workspace : nullptr;
  float* Pb = dbias ? workspace + (size_t)G * D : nullptr;
  const size_t lds = (size_t)4 * D * sizeof(float);
  if (dtype == LDMAE_BF16) {
    DISPATCH_NCH(D, hipLaunchKernelGGL((gate_bwd_kernel<NCH, bf16>), dim3(G), dim3(256), lds, st, dxout, (const bf16*)y, gate, gate_ld, (bf16*)dy, P, Pb, M, D, rows_per_batch, rw));
  } else {
    DISPATCH_NCH(D, hipLaunchKernelGGL((gate_bwd_kernel<NCH, float>), dim3(G), dim3(256), lds, st, dxout, (const float*)y, gate, gate_ld, (float*)dy, P, Pb, M, D, rows_per_batch, rw));
  }
  LDMAE_CHECK_LAUNCH("gate_bwd");
  if (dgate && dbias) {      // the same two reduce launches, in the same order of summation, as the fused rmsnorm_modulate_bwd_gate (bitwise equal)
    const int B = M / rows_per_batch;
    float* dbb = Pb + (size_t)G * D;
    hipLaunchKernelGGL(mod_partials_reduce_kernel, dim3(cdiv(D, 256), B), dim3(256), 0, st, (const float*)nullptr, D, rows_per_batch / rw,
                       (float*)nullptr, (float*)nullptr, 0, (float*)nullptr, (const float*)P, dgate, dgate_ld, (const float*)Pb, dbb);
    group_reduce(dbb, D, 1, D, B, dbias, D, 0.f, st);
    LDMAE_CHECK_LAUNCH("gate_bwd reduce");
    return LDMAE_OK;
  }
  if (dgate) group_reduce(P, D, M / rows_per_batch, D, rows_per_batch / rw, dgate, dgate_ld, 0.f, st);
  if (dbias) return ldmae_colsum(LDMAE_F32, Pb, D, G, D, dbias, 0.f, Pb + (size_t)G * D, stream);   // two-stage column sum of the G partial rows
  LDMAE_CHECK_LAUNCH("gate_bwd reduce");
  return LDMAE_OK;
}

// ------------------------------------------------------------------ column sums (bias gradients)
// rows summed by one workgroup: enough row groups for ~512 workgroups (the adaLN bias gradient is 256 rows x 4608 columns:
// with a fixed 256 rows it ran on 5 workgroups, 65 us for 4.7 MB)
static int colsum_rows(int M, int N) {
  const long colblocks = cdiv(N, 1024);
  long groups = 512 / colblocks; if (groups < 1) groups = 1;
  long rows = (M + groups - 1) / groups;
  int r = 8; while (r < rows && r < 256) r <<= 1;
  return r;
}
// a column block narrower than 1024 columns leaves threads without a column: they take interleaved rows of the same columns instead
// (N = 128: 8 row streams per workgroup instead of 32 busy threads) and the streams are summed in fixed order through LDS
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, int ldx, int M, int N, int rows, float* __restrict__ P) {
  __shared__ float4 red[256];
  const int ncol4 = min(N / 4 - (int)blockIdx.x * 256, 256), nsub = 256 / ncol4;
  const int ci = threadIdx.x % ncol4, rsub = threadIdx.x / ncol4, c = (blockIdx.x * 256 + ci) * 4;
  const int m0 = blockIdx.y * rows, m1 = min(M, m0 + rows);
  float4 s = f4(0.f);
  if (rsub < nsub)
    for (int m = m0 + rsub; m < m1; m += nsub) s = s + load4<T>(X + (size_t)m * ldx + c);
  if (nsub > 1) {
    red[threadIdx.x] = s;
    __syncthreads();
    if (rsub == 0)
      for (int k = 1; k < nsub; ++k) s = s + red[k * ncol4 + ci];
  }
  if (rsub == 0) *(float4*)(P + (size_t)blockIdx.y * N + c) = s;
}
extern "C" long ldmae_colsum_workspace_bytes(int M, int N) { return (long)cdiv(M, colsum_rows(M, N)) * N * 4; }
extern "C" int ldmae_colsum(int dtype, const void* X, int ldx, int M, int N, float* out, float beta, float* workspace, void* stream) {
  LDMAE_REQUIRE(X && out && workspace && M > 0 && N > 0 && N % 4 == 0 && ldx % 4 == 0, "colsum: bad arguments (N=%d ldx=%d multiples of 4)", N, ldx);
  hipStream_t st = as_stream(stream);
  const int rows = colsum_rows(M, N), G = cdiv(M, rows);
  if (dtype == LDMAE_F16) hipLaunchKernelGGL(colsum_kernel<f16>, dim3(cdiv(N, 1024), G), dim3(256), 0, st, (const f16*)X, ldx, M, N, rows, workspace);
  else if (dtype == LDMAE_BF16) hipLaunchKernelGGL(colsum_kernel<bf16>, dim3(cdiv(N, 1024), G), dim3(256), 0, st, (const bf16*)X, ldx, M, N, rows, workspace);
  else hipLaunchKernelGGL(colsum_kernel<float>, dim3(cdiv(N, 1024), G), dim3(256), 0, st, (const float*)X, ldx, M, N, rows, workspace);
  group_reduce(workspace, N, 1, N, G, out, N, beta, st);
  LDMAE_CHECK_LAUNCH("colsum");
  return LDMAE_OK;
}

// ------------------------------------------------------------------ casts
template <typename S, typename Dt>
__global__ void cast_kernel(const S* __restrict__ s, Dt* __restrict__ d, long n) {
  const long n8 = n / 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float v[8];
    Vec8<S>::load(s + i * 8, v);
    Vec8<Dt>::store(d + i * 8, v);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) d[n8 * 8 + threadIdx.x] = from_f<Dt>(to_f<S>(s[n8 * 8 + threadIdx.x]));
}
extern "C" int ldmae_cast(int src_dtype, int dst_dtype, const void* src, void* dst, long n, void* stream) {
  LDMAE_REQUIRE(src && dst && n > 0, "cast: null pointer or empty");
  LDMAE_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "cast: pointers must be 16-B aligned");
  hipStream_t st = as_stream(stream);
  const unsigned grid = ew_grid(n / 8 + 1);
  if (src_dtype == LDMAE_F32 && dst_dtype == LDMAE_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16>), dim3(grid), dim3(256), 0, st, (const float*)src, (bf16*)dst, n);
  else if (src_dtype == LDMAE_BF16 && dst_dtype == LDMAE_F32) hipLaunchKernelGGL((cast_kernel<bf16, float>), dim3(grid), dim3(256), 0, st, (const bf16*)src, (float*)dst, n);
  else if (src_dtype == LDMAE_F32 && dst_dtype == LDMAE_F16) hipLaunchKernelGGL((cast_kernel<float, f16>), dim3(grid), dim3(256), 0, st, (const float*)src, (f16*)dst, n);
  else if (src_dtype == LDMAE_F16 && dst_dtype == LDMAE_F32) hipLaunchKernelGGL((cast_kernel<f16, float>), dim3(grid), dim3(256), 0, st, (const f16*)src, (float*)dst, n);
  else if (src_dtype == LDMAE_F32 && dst_dtype == LDMAE_F32) hipLaunchKernelGGL((cast_kernel<float, float>), dim3(grid), dim3(256), 0, st, (const float*)src, (float*)dst, n);
  else LDMAE_FAIL(LDMAE_ERR_INVALID, "cast: unsupported %d -> %d", src_dtype, dst_dtype);
  LDMAE_CHECK_LAUNCH("cast");
  return LDMAE_OK;
}

// ------------------------------------------------------------------ thin GEMMs: contraction length K <= 32 over M = batch * tokens rows
// The DiT's PatchEmbed at patch 1 (lightningdit.py:309, 402: K = C * p * p = 16) and the dX of the FinalLayer's 16-column Linear (:270)
// are [M, 768] <-> [M, 16] products at M = 262144: 0.02 FLOP per byte, pure HBM streaming.  As 64x64-tile MFMA GEMMs they ran at
// 2.2 TB/s (358 us forward, 366 + 34 + 146 us for weight gradient + split reduce + bias column sum: two passes over the 805-MB
// gradient).  Here the K-vectors of a workgroup's rows sit in LDS (broadcast reads) and the weights / partial sums live in registers: one
// pass over the big tensor.
//   thin_nt: out[m, n] = sum_k T[m, k] W[n, k] + bias[n] (+ pos[m % rows_per_batch, n])
//   thin_tn: dW[n, k] = sum_m G[m, n] T[m, k], dbias[n] = sum_m G[m, n]   (per-chunk partials, summed in fixed order by splitk_reduce)
template <int K, typename OutT, bool POS>
__global__ __launch_bounds__(256) void thin_nt_kernel(const float* __restrict__ T, const float* __restrict__ W, const float* __restrict__ bias,
                                                      const float* __restrict__ pos, OutT* __restrict__ out, int M, int N, int rows_per_batch,
                                                      int rows_per_wg) {
  // the workgroup's rows of T (128 x K floats) are staged in LDS once (coalesced 16-B loads) and read back at a wave-uniform address
  // (LDS broadcast); a thread owns FOUR adjacent columns, so a wave stores whole 1-KiB runs (one column per thread moved 4 B per lane:
  // 2.5 TB/s); 128 rows per workgroup keep ~6 waves per SIMD in flight
  __shared__ __attribute__((aligned(16))) float ts[128 * K];
  const int m0 = blockIdx.x * rows_per_wg, m1 = min(M, m0 + rows_per_wg);
  for (int i = threadIdx.x * 4; i < (m1 - m0) * K; i += 256 * 4) *(float4*)(ts + i) = *(const float4*)(T + (size_t)m0 * K + i);
  __syncthreads();
  const int n = (blockIdx.y * 256 + threadIdx.x) * 4;
  if (n >= N) return;
  float w[4][K];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < K; k += 4) {
      const float4 v = *(const float4*)(W + (size_t)(n + j) * K + k);
      w[j][k] = v.x; w[j][k + 1] = v.y; w[j][k + 2] = v.z; w[j][k + 3] = v.w;
    }
  const float4 b = bias ? *(const float4*)(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 2
  for (int m = m0; m < m1; ++m) {
    const float* t = ts + (m - m0) * K;
    float a0 = b.x, a1 = b.y, a2 = b.z, a3 = b.w;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float tk = t[k];
      a0 = fmaf(tk, w[0][k], a0); a1 = fmaf(tk, w[1][k], a1); a2 = fmaf(tk, w[2][k], a2); a3 = fmaf(tk, w[3][k], a3);
    }
    if (POS) {
      const float4 q = *(const float4*)(pos + (size_t)(m % rows_per_batch) * N + n);
      a0 += q.x; a1 += q.y; a2 += q.z; a3 += q.w;
    }
    if constexpr (sizeof(OutT) == 4) *(float4*)((float*)out + (size_t)m * N + n) = make_float4(a0, a1, a2, a3);
    else { bf16x4 o; o[0] = (bf16)a0; o[1] = (bf16)a1; o[2] = (bf16)a2; o[3] = (bf16)a3; *(bf16x4*)((bf16*)out + (size_t)m * N + n) = o; }
  }
}

template <int K>
__global__ __launch_bounds__(256) void thin_tn_kernel(const float* __restrict__ G, const float* __restrict__ T, float* __restrict__ P,
                                                      float* __restrict__ Pb, int M, int N, int rows_per_wg) {
  __shared__ __attribute__((aligned(16))) float ts[512 * K];
  const int m0 = blockIdx.x * rows_per_wg, m1 = min(M, m0 + rows_per_wg);
  for (int i = threadIdx.x * 4; i < (m1 - m0) * K; i += 256 * 4) *(float4*)(ts + i) = *(const float4*)(T + (size_t)m0 * K + i);
  __syncthreads();
  const int n = blockIdx.y * 256 + threadIdx.x;
  if (n >= N) return;
  float acc[K], accb = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) acc[k] = 0.f;
#pragma unroll 4
  for (int m = m0; m < m1; ++m) {
    const float g = G[(size_t)m * N + n];
    const float* t = ts + (m - m0) * K;                 // wave-uniform LDS address: broadcast reads
    accb += g;
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = fmaf(g, t[k], acc[k]);
  }
  float* p = P + ((size_t)blockIdx.x * N + n) * K;
#pragma unroll
  for (int k = 0; k < K; k += 4) *(float4*)(p + k) = make_float4(acc[k], acc[k + 1], acc[k + 2], acc[k + 3]);
  if (Pb) Pb[(size_t)blockIdx.x * N + n] = accb;
}
// out[i] = beta * out[i] + sum_c P[c][i], i < n, in a FIXED order (deterministic): one wave per 4 consecutive outputs, lane l sums chunks
// l, l + 64, ... and the 64 lane sums are folded by a butterfly -- few outputs (12288) x many chunks (512) needs its parallelism across chunks
__global__ __launch_bounds__(256) void thin_reduce_kernel(const float* __restrict__ P, float* __restrict__ out, long n, int chunks, float beta) {
  const int lane = threadIdx.x & 63;
  const long i = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
  if (i >= n) return;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int c = lane; c < chunks; c += 64) {
    const float4 t = *(const float4*)(P + (size_t)c * n + i);
    s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s.x += __shfl_xor(s.x, o, 64); s.y += __shfl_xor(s.y, o, 64); s.z += __shfl_xor(s.z, o, 64); s.w += __shfl_xor(s.w, o, 64);
  }
  if (lane == 0) {
    if (beta != 0.f) { const float4 o = *(const float4*)(out + i); s.x += beta * o.x; s.y += beta * o.y; s.z += beta * o.z; s.w += beta * o.w; }
    *(float4*)(out + i) = s;
  }
}
constexpr int THIN_ROWS = 512;      // rows of M per workgroup (the kernels' LDS image of T is sized for it)
static bool thin_ok(int N, int K) { return (K == 16 || K == 32) && N % 4 == 0 && N >= 4; }

extern "C" int ldmae_thin_nt(int out_dtype, const float* T, const float* W, const float* bias, const float* pos, void* out, int M, int N, int K,
                             int rows_per_batch, void* stream) {
  LDMAE_REQUIRE(T && W && out && M > 0, "thin_nt: null pointer or empty input");
  LDMAE_REQUIRE(thin_ok(N, K), "thin_nt: K=%d must be 16 or 32 and N=%d a multiple of 4 (use ldmae_gemm_nt otherwise)", K, N);
  LDMAE_REQUIRE(!pos || rows_per_batch > 0, "thin_nt: pos needs rows_per_batch");
  LDMAE_REQUIRE(out_dtype == LDMAE_F32 || out_dtype == LDMAE_BF16, "thin_nt: bad out dtype %d", out_dtype);
  const dim3 grid(cdiv(M, 128), cdiv(N / 4, 256));
  hipStream_t st = as_stream(stream);
#define THIN_NT(KK, OT, PP) hipLaunchKernelGGL((thin_nt_kernel<KK, OT, PP>), grid, dim3(256), 0, st, T, W, bias, pos, (OT*)out, M, N, rows_per_batch, 128)
#define THIN_NT_K(KK)                                                                                 \
  if (out_dtype == LDMAE_F32) { if (pos) THIN_NT(KK, float, true); else THIN_NT(KK, float, false); }  \
  else { if (pos) THIN_NT(KK, bf16, true); else THIN_NT(KK, bf16, false); }
  if (K == 16) { THIN_NT_K(16) } else { THIN_NT_K(32) }
#undef THIN_NT_K
#undef THIN_NT
  LDMAE_CHECK_LAUNCH("thin_nt");
  return LDMAE_OK;
}
extern "C" long ldmae_thin_tn_workspace_bytes(int M, int N, int K) { return (long)cdiv(M, THIN_ROWS) * ((long)N * K + N) * 4; }
extern "C" int ldmae_thin_tn(const float* G, const float* T, float* dW, float* dbias, int M, int N, int K, float beta, float* workspace,
                             long workspace_bytes, void* stream) {
  LDMAE_REQUIRE(G && T && dW && M > 0, "thin_tn: null pointer or empty input");
  LDMAE_REQUIRE(thin_ok(N, K), "thin_tn: K=%d must be 16 or 32 and N=%d a multiple of 4 (use ldmae_gemm_tn otherwise)", K, N);
  LDMAE_REQUIRE(beta == 0.f || beta == 1.f, "thin_tn: beta must be 0 or 1");
  LDMAE_REQUIRE(workspace && workspace_bytes >= ldmae_thin_tn_workspace_bytes(M, N, K), "thin_tn: workspace too small");
  const int chunks = cdiv(M, THIN_ROWS);
  float* P = workspace;
  float* Pb = dbias ? workspace + (size_t)chunks * N * K : nullptr;
  const dim3 grid(chunks, cdiv(N, 256));
  hipStream_t st = as_stream(stream);
  if (K == 16) hipLaunchKernelGGL(thin_tn_kernel<16>, grid, dim3(256), 0, st, G, T, P, Pb, M, N, THIN_ROWS);
  else hipLaunchKernelGGL(thin_tn_kernel<32>, grid, dim3(256), 0, st, G, T, P, Pb, M, N, THIN_ROWS);
  const long n = (long)N * K;
  hipLaunchKernelGGL(thin_reduce_kernel, dim3((unsigned)((n / 4 + 3) / 4)), dim3(256), 0, st, P, dW, n, chunks, beta);
  if (dbias) hipLaunchKernelGGL(thin_reduce_kernel, dim3(cdiv(N / 4, 4)), dim3(256), 0, st, Pb, dbias, (long)N, chunks, beta);
  LDMAE_CHECK_LAUNCH("thin_tn");
  return LDMAE_OK;
}

// dst[i] += src[i] for up to 32 (dst, src, n) f32 triples in ONE launch: the small parameter gradients of a block (norm weights, biases, QK-norm
// weights) added into their .grad views of the optimizer's slab -- as separate AccumulateGrad adds they were 8 launches of ~5 us per block.
constexpr int MULTI_ADD_MAX = 32;
struct MultiAddArgs { float* dst[MULTI_ADD_MAX]; const float* src[MULTI_ADD_MAX]; long n[MULTI_ADD_MAX]; };
__global__ void multi_add_kernel(MultiAddArgs a) {
  float* d = a.dst[blockIdx.y];
  const float* s = a.src[blockIdx.y];
  const long n = a.n[blockIdx.y];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) d[i] += s[i];
}
extern "C" int ldmae_multi_add(int count, void* const* dst, const void* const* src, const long* n, void* stream) {
  LDMAE_REQUIRE(dst && src && n && count > 0 && count <= MULTI_ADD_MAX, "multi_add: count=%d must be 1..%d", count, MULTI_ADD_MAX);
  MultiAddArgs a{};
  long nmax = 0;
  for (int i = 0; i < count; ++i) {
    LDMAE_REQUIRE(dst[i] && src[i] && n[i] > 0, "multi_add: entry %d is null or empty", i);
    a.dst[i] = (float*)dst[i]; a.src[i] = (const float*)src[i]; a.n[i] = n[i];
    nmax = n[i] > nmax ? n[i] : nmax;
  }
  const unsigned gx = (unsigned)min((long)256, (nmax + 255) / 256);
  hipLaunchKernelGGL(multi_add_kernel, dim3(gx, count), dim3(256), 0, as_stream(stream), a);
  LDMAE_CHECK_LAUNCH("multi_add");
  return LDMAE_OK;
}

// `count` equally sized f32 tensors -> one stacked tensor in the activation type with ONE launch (the adaLN weights of all blocks become the
// [depth * 6D, D] operand of a single GEMM: models/lightningdit.py:_AdaLNAllFn).  The source pointers travel by value in the kernel arguments.
constexpr int CAST_STACK_MAX = 64;
struct CastStackArgs { const float* src[CAST_STACK_MAX]; };
template <typename Dt>
__global__ void cast_stack_kernel(CastStackArgs a, Dt* __restrict__ dst, long n_each) {
  const float* s = a.src[blockIdx.y];
  Dt* d = dst + (size_t)blockIdx.y * n_each;
  const long n8 = n_each / 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float v[8];
    Vec8<float>::load(s + i * 8, v);
    Vec8<Dt>::store(d + i * 8, v);
  }
}
extern "C" int ldmae_cast_stack(int dst_dtype, const void* const* srcs, int count, long n_each, void* dst, void* stream) {
  LDMAE_REQUIRE(srcs && dst && count > 0 && count <= CAST_STACK_MAX, "cast_stack: count=%d must be 1..%d", count, CAST_STACK_MAX);
  LDMAE_REQUIRE(n_each > 0 && n_each % 8 == 0, "cast_stack: n_each=%ld must be a positive multiple of 8", n_each);
  LDMAE_REQUIRE(((uintptr_t)dst & 15) == 0, "cast_stack: dst must be 16-B aligned");
  CastStackArgs a{};
  for (int i = 0; i < count; ++i) {
    LDMAE_REQUIRE(srcs[i] && ((uintptr_t)srcs[i] & 15) == 0, "cast_stack: source %d is null or not 16-B aligned", i);
    a.src[i] = (const float*)srcs[i];
  }
  const unsigned gx = (unsigned)min((long)1024, (n_each / 8 + 255) / 256);
  if (dst_dtype == LDMAE_BF16) hipLaunchKernelGGL(cast_stack_kernel<bf16>, dim3(gx, count), dim3(256), 0, as_stream(stream), a, (bf16*)dst, n_each);
  else if (dst_dtype == LDMAE_F32) hipLaunchKernelGGL(cast_stack_kernel<float>, dim3(gx, count), dim3(256), 0, as_stream(stream), a, (float*)dst, n_each);
  else LDMAE_FAIL(LDMAE_ERR_INVALID, "cast_stack: unsupported dst dtype %d", dst_dtype);
  LDMAE_CHECK_LAUNCH("cast_stack");
  return LDMAE_OK;
}

// f32 [R,C] -> T [R,C] and T [C,R] through a 32x33 LDS tile
template <typename T>
__global__ __launch_bounds__(256) void cast_weight_kernel(const float* __restrict__ src, T* __restrict__ dst, T* __restrict__ dstT, int R, int C) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int j = ty; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + tx;
    float v = 0.f;
    if (r < R && c < C) { v = src[(size_t)r * C + c]; if (dst) dst[(size_t)r * C + c] = from_f<T>(v); }
    tile[j][tx] = v;
  }
  __syncthreads();
  if (!dstT) return;
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + tx;
    if (r < R && c < C) dstT[(size_t)c * R + r] = from_f<T>(tile[tx][j]);
  }
}
// The same for shapes whose sides are multiples of 64 (every Linear weight of the shipped models): 64x64 tiles, 16-B loads, 8-B stores of the
// straight copy and 32-B runs of the transposed one (the 4-byte-per-lane form above ran at ~2 TB/s: 48 launches = 0.34 ms per train step)
template <typename T>
__global__ __launch_bounds__(256) void cast_weight64_kernel(const float* __restrict__ src, T* __restrict__ dst, T* __restrict__ dstT, int R, int C) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int lr = threadIdx.x >> 4, lc = (threadIdx.x & 15) * 4;          // 16 rows x 16 float4 per pass
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int r = p * 16 + lr;
    const float4 v = *(const float4*)(src + (size_t)(r0 + r) * C + c0 + lc);
    if (dst) {
      T* d = dst + (size_t)(r0 + r) * C + c0 + lc;
      if constexpr (sizeof(T) == 4) *(float4*)d = v;
      else { typename Pack<T>::v4 o; o[0] = from_f<T>(v.x); o[1] = from_f<T>(v.y); o[2] = from_f<T>(v.z); o[3] = from_f<T>(v.w); *(typename Pack<T>::v4*)d = o; }
    }
    tile[r][lc] = v.x; tile[r][lc + 1] = v.y; tile[r][lc + 2] = v.z; tile[r][lc + 3] = v.w;
  }
  __syncthreads();
  if (!dstT) return;
  const int c = threadIdx.x >> 2, rq = (threadIdx.x & 3) * 16;            // output row c of the transposed tile, 16 of its 64 values
  T* d = dstT + (size_t)(c0 + c) * R + r0 + rq;
#pragma unroll
  for (int k = 0; k < 16; k += 4) {
    const float a0 = tile[rq + k][c], a1 = tile[rq + k + 1][c], a2 = tile[rq + k + 2][c], a3 = tile[rq + k + 3][c];
    if constexpr (sizeof(T) == 4) *(float4*)(d + k) = make_float4(a0, a1, a2, a3);
    else { typename Pack<T>::v4 o; o[0] = from_f<T>(a0); o[1] = from_f<T>(a1); o[2] = from_f<T>(a2); o[3] = from_f<T>(a3); *(typename Pack<T>::v4*)(d + k) = o; }
  }
}
extern "C" int ldmae_cast_weight(int dst_dtype, const float* src, void* dst, void* dstT, int R, int C, void* stream) {
  LDMAE_REQUIRE(src && (dst || dstT) && R > 0 && C > 0, "cast_weight: null pointer or empty");
  const bool al = ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0 && ((uintptr_t)dstT & 15) == 0;
  if (R % 64 == 0 && C % 64 == 0 && al) {
    dim3 grid(C / 64, R / 64);
    if (dst_dtype == LDMAE_F16) hipLaunchKernelGGL(cast_weight64_kernel<f16>, grid, dim3(256), 0, as_stream(stream), src, (f16*)dst, (f16*)dstT, R, C);
    else if (dst_dtype == LDMAE_BF16) hipLaunchKernelGGL(cast_weight64_kernel<bf16>, grid, dim3(256), 0, as_stream(stream), src, (bf16*)dst, (bf16*)dstT, R, C);
    else hipLaunchKernelGGL(cast_weight64_kernel<float>, grid, dim3(256), 0, as_stream(stream), src, (float*)dst, (float*)dstT, R, C);
    LDMAE_CHECK_LAUNCH("cast_weight");
    return LDMAE_OK;
  }
  dim3 grid(cdiv(C, 32), cdiv(R, 32));
  if (dst_dtype == LDMAE_F16) hipLaunchKernelGGL(cast_weight_kernel<f16>, grid, dim3(256), 0, as_stream(stream), src, (f16*)dst, (f16*)dstT, R, C);
  else if (dst_dtype == LDMAE_BF16) hipLaunchKernelGGL(cast_weight_kernel<bf16>, grid, dim3(256), 0, as_stream(stream), src, (bf16*)dst, (bf16*)dstT, R, C);
  else hipLaunchKernelGGL(cast_weight_kernel<float>, grid, dim3(256), 0, as_stream(stream), src, (float*)dst, (float*)dstT, R, C);
  LDMAE_CHECK_LAUNCH("cast_weight");
  return LDMAE_OK;
}

// ------------------------------------------------------------------ embedders
__global__ void timestep_embedding_kernel(const float* __restrict__ t, float* __restrict__ out, int B, int dim, float max_period) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  const int b = i / half, j = i % half;
  // lightningdit.py:124-129: freqs = exp(-log(max_period) * j / half); args = t * freqs; [cos | sin]
  const float freq = expf(-logf(max_period) * (float)j / (float)half);
  const float a = t[b] * freq;
  out[(size_t)b * dim + j] = cosf(a);
  out[(size_t)b * dim + half + j] = sinf(a);
  if ((dim & 1) && j == 0) out[(size_t)b * dim + dim - 1] = 0.f;
}
extern "C" int ldmae_timestep_embedding(const float* t, float* out, int B, int dim, float max_period, void* stream) {
  LDMAE_REQUIRE(t && out && B > 0 && dim >= 2, "timestep_embedding: bad arguments");
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3(cdiv((long)B * (dim / 2), 256)), dim3(256), 0, as_stream(stream), t, out, B, dim, max_period);
  LDMAE_CHECK_LAUNCH("timestep_embedding");
  return LDMAE_OK;
}

template <typename OutT>
__global__ void silu_fwd_kernel(const float* __restrict__ x, OutT* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = x[i];
    out[i] = from_f<OutT>(v / (1.f + expf(-v)));
  }
}
__global__ void silu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = x[i], s = 1.f / (1.f + expf(-v));
    dx[i] = dy[i] * s * (1.f + v * (1.f - s));
  }
}
extern "C" int ldmae_silu_fwd(int out_dtype, const float* x, void* out, long n, void* stream) {
  LDMAE_REQUIRE(x && out && n > 0, "silu_fwd: bad arguments");
  if (out_dtype == LDMAE_BF16) hipLaunchKernelGGL(silu_fwd_kernel<bf16>, dim3(ew_grid(n)), dim3(256), 0, as_stream(stream), x, (bf16*)out, n);
  else hipLaunchKernelGGL(silu_fwd_kernel<float>, dim3(ew_grid(n)), dim3(256), 0, as_stream(stream), x, (float*)out, n);
  LDMAE_CHECK_LAUNCH("silu_fwd");
  return LDMAE_OK;
}
extern "C" int ldmae_silu_bwd(const float* dy, const float* x, float* dx, long n, void* stream) {
  LDMAE_REQUIRE(dy && x && dx && n > 0, "silu_bwd: bad arguments");
  hipLaunchKernelGGL(silu_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, as_stream(stream), dy, x, dx, n);
  LDMAE_CHECK_LAUNCH("silu_bwd");
  return LDMAE_OK;
}

__global__ void label_embed_fwd_kernel(const float* __restrict__ table, const long long* __restrict__ y, const unsigned char* __restrict__ drop,
                                       float* __restrict__ out, int B, int D, int num_classes) {
  const int b = blockIdx.x;
  const long long row = (drop && drop[b]) ? num_classes : y[b];
  for (int d = threadIdx.x; d < D; d += blockDim.x) out[(size_t)b * D + d] = table[(size_t)row * D + d];
}
__global__ void label_embed_bwd_kernel(const float* __restrict__ dout, const long long* __restrict__ y, const unsigned char* __restrict__ drop,
                                       float* __restrict__ dtable, int B, int D, int num_classes) {
  // one workgroup per table row; the samples that hit the row are found cooperatively (one sample per thread and pass: a ballot per wave) and
  // summed in ascending sample order (deterministic) by walking the set bits.  (Every thread used to scan all B labels itself: 140 us for a
  // 3 MB result; then a 256-step loop over a hit array in LDS: 84 us.)
  __shared__ unsigned long long mask[4];
  const int row = blockIdx.x, wave = threadIdx.x >> 6;
  for (int b0 = 0; b0 < B; b0 += blockDim.x) {
    const int b = b0 + threadIdx.x;
    const bool mine = b < B && ((drop && drop[b]) ? num_classes : y[b]) == row;
    const unsigned long long m = __ballot(mine);
    if ((threadIdx.x & 63) == 0) mask[wave] = m;
    __syncthreads();
    if ((mask[0] | mask[1] | mask[2] | mask[3]) != 0ull) {
      for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float s = 0.f;
        for (int w = 0; w < 4; ++w)
          for (unsigned long long mm = mask[w]; mm; mm &= mm - 1) s += dout[(size_t)(b0 + w * 64 + __builtin_ctzll(mm)) * D + d];
        dtable[(size_t)row * D + d] += s;
      }
    }
    __syncthreads();
  }
}
extern "C" int ldmae_label_embed_fwd(const float* table, const long long* y, const unsigned char* drop, float* out, int B, int D,
                                     int num_classes, void* stream) {
  LDMAE_REQUIRE(table && y && out && B > 0 && D > 0, "label_embed_fwd: bad arguments");
  hipLaunchKernelGGL(label_embed_fwd_kernel, dim3(B), dim3(256), 0, as_stream(stream), table, y, drop, out, B, D, num_classes);
  LDMAE_CHECK_LAUNCH("label_embed_fwd");
  return LDMAE_OK;
}
extern "C" int ldmae_label_embed_bwd(const float* dout, const long long* y, const unsigned char* drop, float* dtable, int B, int D,
                                     int num_classes, int rows, void* stream) {
  LDMAE_REQUIRE(dout && y && dtable && B > 0 && D > 0 && rows > 0, "label_embed_bwd: bad arguments");
  hipLaunchKernelGGL(label_embed_bwd_kernel, dim3(rows), dim3(256), 0, as_stream(stream), dout, y, drop, dtable, B, D, num_classes);
  LDMAE_CHECK_LAUNCH("label_embed_bwd");
  return LDMAE_OK;
}

// ------------------------------------------------------------------ fused AdamW + EMA over flat f32 buffers
// Scalars are prepared on the host exactly as torch does (python doubles rounded to f32 at the point of use):
// torch/optim/adamw.py _single_tensor_adamw (train_accum.py:121) then ema.mul_(d).add_(p, alpha=1-d) (:336-347).
struct AdamArgs { float decay_mul, w1, beta2, w2, bc2_sqrt, eps, neg_step, ema_d, ema_a, gscale; };
__global__ void adamw_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                 float* __restrict__ ema, long n, AdamArgs a) {
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long)gridDim.x * blockDim.x * 4) {
    float4 pv = *(float4*)(p + i), gv = *(const float4*)(g + i), mv = *(float4*)(m + i), vv = *(float4*)(v + i);
    float pa[4] = {pv.x, pv.y, pv.z, pv.w}, ga[4] = {gv.x, gv.y, gv.z, gv.w}, ma[4] = {mv.x, mv.y, mv.z, mv.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
    float ea[4] = {0, 0, 0, 0};
    if (ema) { float4 e = *(float4*)(ema + i); ea[0] = e.x; ea[1] = e.y; ea[2] = e.z; ea[3] = e.w; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gj = a.gscale == 1.f ? ga[j] : ga[j] * a.gscale;
      pa[j] = pa[j] * a.decay_mul;                                  // p.mul_(1 - lr*wd)
      ma[j] = ma[j] + a.w1 * (gj - ma[j]);                          // exp_avg.lerp_(grad, 1-beta1)
      va[j] = va[j] * a.beta2 + a.w2 * (gj * gj);                   // exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
      const float denom = sqrtf(va[j]) / a.bc2_sqrt + a.eps;        // (sqrt(v)/bc2_sqrt).add_(eps)
      pa[j] = pa[j] + a.neg_step * (ma[j] / denom);                 // p.addcdiv_(m, denom, value=-step_size)
      ea[j] = ea[j] * a.ema_d + a.ema_a * pa[j];
    }
    *(float4*)(p + i) = make_float4(pa[0], pa[1], pa[2], pa[3]);
    *(float4*)(m + i) = make_float4(ma[0], ma[1], ma[2], ma[3]);
    *(float4*)(v + i) = make_float4(va[0], va[1], va[2], va[3]);
    if (ema) *(float4*)(ema + i) = make_float4(ea[0], ea[1], ea[2], ea[3]);
  }
}
extern "C" int ldmae_adamw_ema(float* p, const float* g, float* m, float* v, float* ema, long n, int step, double lr, double beta1,
                               double beta2, double eps, double weight_decay, double ema_decay, double grad_scale, void* stream) {
  LDMAE_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adamw_ema: bad arguments");
  LDMAE_REQUIRE(n % 4 == 0, "adamw_ema: flat length %ld must be padded to a multiple of 4", n);
  const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
  AdamArgs a;
  a.decay_mul = (float)(1.0 - lr * weight_decay);
  a.w1 = (float)(1.0 - beta1);
  a.beta2 = (float)beta2;
  a.w2 = (float)(1.0 - beta2);
  a.bc2_sqrt = (float)sqrt(bc2);
  a.eps = (float)eps;
  a.neg_step = (float)(-(lr / bc1));
  a.ema_d = (float)ema_decay;
  a.ema_a = (float)(1.0 - ema_decay);
  a.gscale = (float)grad_scale;
  hipLaunchKernelGGL(adamw_ema_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, as_stream(stream), p, g, m, v, ema, n, a);
  LDMAE_CHECK_LAUNCH("adamw_ema");
  return LDMAE_OK;
}
__global__ void ema_only_kernel(float* __restrict__ ema, const float* __restrict__ p, long n, float d, float a) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) ema[i] = ema[i] * d + a * p[i];
}
extern "C" int ldmae_ema_only(float* ema, const float* p, long n, double ema_decay, void* stream) {
  LDMAE_REQUIRE(ema && p && n > 0, "ema_only: bad arguments");
  hipLaunchKernelGGL(ema_only_kernel, dim3(ew_grid(n)), dim3(256), 0, as_stream(stream), ema, p, n, (float)ema_decay, (float)(1.0 - ema_decay));
  LDMAE_CHECK_LAUNCH("ema_only");
  return LDMAE_OK;
}
