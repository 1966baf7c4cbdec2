// Shared pieces of the bf16 NT GEMM kernels (gemm.hip): epilogue arguments, scalar / vector epilogues, the LDS stage
// swizzle and the strip epilogue.
#pragma once
#include "common.h"

// ------------------------------------------------------------------------------------------------
// Epilogue shared by all NT kernels.  Accumulator tile layout (16x16 MFMA C/D map):
//   col = lane & 15 ; row = (lane >> 4) * 4 + reg
// ------------------------------------------------------------------------------------------------
struct EpiArgs {
  void* C;              // output [M,N] (dtype out_dt)      -- EPI_BIAS: result; EPI_GATE_RES: y (may be null)
  void* C2;             // EPI_BIAS_GELU: pre-activation copy (may be null)
  const float* bias;    // [N] or null
  // EPI_GATE_RES: xout[m,n] = xin[m,n] + gate[m / rows_per_batch, n] * (acc + bias)
  const float* xin;     // [M,N] f32
  float* xout;          // [M,N] f32 (may alias xin)
  const float* gate;    // [B, gate_ld] f32 (a column slice of the adaLN output)
  int gate_ld;
  int rows_per_batch;
  int ldc;              // row stride of C in elements
  float beta;           // EPI_BIAS with f32 out: C = acc + bias + beta*C   (beta 0 or 1: gradient accumulation)
  float f16_max;        // fp16 outputs: clamp to +-f16_max before the conversion (65504: forward GEMMs saturate; infinity: gradient GEMMs overflow to inf)
  // EPI_QKV_ROPE: C = packed qkv [M, 3*heads*64]; q2 / k2 head-major [B, heads, rows_per_batch, 64]; QK-norm weights (NULL: RoPE only), tables [rows_per_batch, 64]
  void* q2;
  void* k2;
  const float* wq;
  const float* wk;
  const float* cosT;
  const float* sinT;
  int heads;
  int store_raw_qk;     // 0: forward-only call, the pre-norm q / k thirds of C are not written
  float eps;
};

// gemm_nt_lines.hip: the whole-line form of the persistent bf16 NT kernel (128-B LDS rows, seamless ring of five half-block slots).  Returns 0
// when the shape / alignment is outside what it covers; the caller then launches gemm_nt_persist_kernel.
int ldmae_launch_nt_lines(int epi, int out_bf16, const void* A, const void* B, int M, int N, int K, int lda, int ldb, const struct EpiArgs& e,
                          int grid, int ntiles, hipStream_t st);
// the same kernel on fp16 operands (v_mfma_f32_16x16x32_f16; the TF32-class forward path): epilogues BIAS / GATE_RES / BIAS_POS / BIAS_GELU,
// output fp16 (out_f16) or f32.  Returns 0 for anything else.
int ldmae_launch_nt_lines_f16(int epi, int out_f16, const void* A, const void* B, int M, int N, int K, int lda, int ldb, const struct EpiArgs& e,
                              int grid, int ntiles, hipStream_t st);

#ifdef LDMAE_DIAG
// probe/gemm_nt_defer.hip (diagnostic build): persistent bf16 NT kernel whose fused epilogue (gated residual / SwiGLU) runs inside the NEXT
// tile's main loop.  Returns 0 when the shape or the arguments are outside what it covers.
int ldmae_launch_nt_defer(int epi, const void* A, const void* B, int M, int N, int K, int lda, int ldb, const EpiArgs& e, int grid, int ntiles,
                          hipStream_t st);
// probe/gemm_nt_wl.hip (diagnostic build): the same tile with a whole-line ring DMA (128-B LDS rows); bf16 outputs, epilogues 0 / 1 / 4 / 5
int ldmae_launch_nt_wl(int epi, const void* A, const void* B, int M, int N, int K, int lda, int ldb, const EpiArgs& e, int grid, int ntiles,
                       hipStream_t st);
// probe/gemm_w4.hip (diagnostic build): experimental NT kernels; return 0 if they have no instantiation for `epi`
int ldmae_launch_nt_w4(int epi, int out_bf16, const void* A, const void* B, int M, int N, int K, int lda, int ldb, const EpiArgs& e, int grid,
                       int ntiles, hipStream_t st);
int ldmae_launch_nt_p8(int epi, int out_bf16, const void* A, const void* B, int M, int N, int K, int lda, int ldb, const EpiArgs& e, int grid,
                       int ntiles, hipStream_t st);
#endif

// The gated residual adds gate * y with y AS STORED in the activation type (the reference's autocast Linear returns bf16:
// lightningdit.py:248-249); every kernel that forms it -- the fused epilogues here, the deferred units of gemm_nt_defer.hip --
// uses this one definition, so they agree bit for bit.
template <typename OutT> __device__ __forceinline__ float act_round(float y) { return to_f<OutT>(from_f<OutT>(y)); }
__device__ __forceinline__ float4 gate_res4(float4 x, float4 g, float4 y) {
  return make_float4(fmaf(g.x, y.x, x.x), fmaf(g.y, y.y, x.y), fmaf(g.z, y.z, x.z), fmaf(g.w, y.w, x.w));
}
template <typename OutT> __device__ __forceinline__ float4 act_round4(float4 y) {
  return make_float4(act_round<OutT>(y.x), act_round<OutT>(y.y), act_round<OutT>(y.z), act_round<OutT>(y.w));
}

// d/dx of the exact-erf GELU times the incoming gradient AS STORED (vmae.hip: gelu_bwd_kernel computes the same expression on the stored
// fc2 input gradient, so the fused epilogue and GEMM + ldmae_gelu_bwd agree bit for bit)
template <typename OutT> __device__ __forceinline__ float gelu_bwd_val(float acc, float v) {
  const float cdf = 0.5f * (1.f + erf_act<OutT>(v * 0.70710678118654752f)), pdf = 0.3989422804014327f * __expf(-0.5f * v * v);
  return act_round<OutT>(acc) * (cdf + v * pdf);
}

// scalar-path output conversion: fp16 outputs clamp to +-e.f16_max first (forward: 65504 = saturate; gradient GEMMs: infinity = plain cast)
template <typename OutT> __device__ __forceinline__ OutT cvt_out(const EpiArgs& e, float v) {
  if constexpr (__is_same(OutT, f16)) return from_f<OutT>(sat_f16(v, e.f16_max));
  else return from_f<OutT>(v);
}
template <int EPI, typename OutT>
__device__ __forceinline__ void epi_store(const EpiArgs& e, int m, int n, int M, int N, float acc) {
  if (m >= M || n >= N) return;
  float y = acc + (e.bias ? e.bias[n] : 0.f);
  if (EPI == LDMAE_EPI_BIAS) {
    OutT* C = (OutT*)e.C;
    if (e.beta != 0.f) y += e.beta * to_f<OutT>(C[(size_t)m * e.ldc + n]);
    C[(size_t)m * e.ldc + n] = cvt_out<OutT>(e, y);
  } else if (EPI == LDMAE_EPI_BIAS_POS) {
    y += e.xin[(size_t)(m % e.rows_per_batch) * N + n];
    ((OutT*)e.C)[(size_t)m * e.ldc + n] = cvt_out<OutT>(e, y);
  } else if (EPI == LDMAE_EPI_BIAS_GELU) {
    if (e.C2) ((OutT*)e.C2)[(size_t)m * e.ldc + n] = cvt_out<OutT>(e, y);
    ((OutT*)e.C)[(size_t)m * e.ldc + n] = cvt_out<OutT>(e, gelu_act<OutT>(y));
  } else if (EPI == LDMAE_EPI_GELU_BWD) {
    const size_t o = (size_t)m * e.ldc + n;
    ((OutT*)e.C)[o] = cvt_out<OutT>(e, gelu_bwd_val<OutT>(y, to_f<OutT>(((const OutT*)e.xin)[o])));
  } else {  // LDMAE_EPI_GATE_RES
    if (e.C) ((OutT*)e.C)[(size_t)m * e.ldc + n] = cvt_out<OutT>(e, y);
    const size_t o = (size_t)m * N + n;
    const float gt = e.gate ? e.gate[(size_t)(m / e.rows_per_batch) * e.gate_ld + n] : 1.f;
    e.xout[o] = fmaf(gt, act_round<OutT>(y), e.xin[o]);
  }
}

// ------------------------------------------------------------------------------------------------
// bf16 NT GEMM building blocks.  256 x 256 output tile, 8 waves (2 x 4, 128 x 64 per wave = 8 x 4 MFMA
// tiles), BK = 32, a 3-deep LDS ring filled by global_load_lds that stays in flight ACROSS the barriers
// (raw s_barrier + counted vmcnt, never __syncthreads in the loop).
// LDS stage image: [256 A rows + 256 B rows][32 k] bf16 = 64-B rows; 16-B chunk c of row r sits at position
// c ^ F[(r >> 2) & 3], F = {0,2,3,1}: conflict-free for the ds_read_b128 lane groups of the 16x16x32 operands.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int ring_f(int g) { return (0x78 >> (2 * g)) & 3; }

// Vector epilogue for a 64-row x 4-column strip: per-column constants (bias, gate) are loaded once, then
// one 16-B access per row.  `fast` = the whole strip is in range and (for the gated form) inside one sample.
template <int EPI, typename OutT> struct Epi4 {
  const EpiArgs& e;
  int n, N;
  float4 bias, gate;
  __device__ __forceinline__ Epi4(const EpiArgs& e_, int m_first, int n_, int N_) : e(e_), n(n_), N(N_) {
    bias = e.bias ? *(const float4*)(e.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    gate = make_float4(1.f, 1.f, 1.f, 1.f);
    if (EPI == LDMAE_EPI_GATE_RES && e.gate) gate = *(const float4*)(e.gate + (size_t)(m_first / e.rows_per_batch) * e.gate_ld + n);
  }
  __device__ __forceinline__ void put(void* base, size_t off, float4 v) const {
    const float e_f16_max = e.f16_max;
    OutT* p = (OutT*)base + off;
    if constexpr (sizeof(OutT) == 4) *(float4*)p = v;
    else { if constexpr (__is_same(OutT, f16)) v = sat_f16(v, e_f16_max);
      typename Pack<OutT>::v4 o; o[0] = from_f<OutT>(v.x); o[1] = from_f<OutT>(v.y); o[2] = from_f<OutT>(v.z); o[3] = from_f<OutT>(v.w); *(typename Pack<OutT>::v4*)p = o; }
  }
  // value destined for C (and C2 for GELU) without storing them; side outputs (xout) are stored here
  __device__ __forceinline__ void compute(int m, float4 a, float4& c, float4& c2) const {
    a.x += bias.x; a.y += bias.y; a.z += bias.z; a.w += bias.w;
    c = a; c2 = a;
    if (EPI == LDMAE_EPI_BIAS) {
      if (e.beta != 0.f) {
        const OutT* p = (const OutT*)e.C + (size_t)m * e.ldc + n;
        c.x += e.beta * to_f<OutT>(p[0]); c.y += e.beta * to_f<OutT>(p[1]); c.z += e.beta * to_f<OutT>(p[2]); c.w += e.beta * to_f<OutT>(p[3]);
      }
    } else if (EPI == LDMAE_EPI_BIAS_POS) {
      const float4 q = *(const float4*)(e.xin + (size_t)(m % e.rows_per_batch) * N + n);
      c = make_float4(a.x + q.x, a.y + q.y, a.z + q.z, a.w + q.w);
    } else if (EPI == LDMAE_EPI_BIAS_GELU) {
      auto g = [](float y) { return gelu_act<OutT>(y); };
      c = make_float4(g(a.x), g(a.y), g(a.z), g(a.w));
    } else if (EPI == LDMAE_EPI_GELU_BWD) {
      const OutT* p = (const OutT*)e.xin + (size_t)m * e.ldc + n;
      c = make_float4(gelu_bwd_val<OutT>(a.x, to_f<OutT>(p[0])), gelu_bwd_val<OutT>(a.y, to_f<OutT>(p[1])), gelu_bwd_val<OutT>(a.z, to_f<OutT>(p[2])),
                      gelu_bwd_val<OutT>(a.w, to_f<OutT>(p[3])));
    } else if (EPI == LDMAE_EPI_GATE_RES) {
      const size_t o = (size_t)m * N + n;
      const float4 xi = *(const float4*)(e.xin + o);
      *(float4*)(e.xout + o) = gate_res4(xi, gate, act_round4<OutT>(a));
    }
  }
  __device__ __forceinline__ void apply(int m, float4 a) const {
    a.x += bias.x; a.y += bias.y; a.z += bias.z; a.w += bias.w;
    const size_t oc = (size_t)m * e.ldc + n;
    if (EPI == LDMAE_EPI_BIAS) {
      if (e.beta != 0.f) {
        const OutT* p = (const OutT*)e.C + oc;
        a.x += e.beta * to_f<OutT>(p[0]); a.y += e.beta * to_f<OutT>(p[1]); a.z += e.beta * to_f<OutT>(p[2]); a.w += e.beta * to_f<OutT>(p[3]);
      }
      put(e.C, oc, a);
    } else if (EPI == LDMAE_EPI_BIAS_POS) {
      const float4 q = *(const float4*)(e.xin + (size_t)(m % e.rows_per_batch) * N + n);
      put(e.C, oc, make_float4(a.x + q.x, a.y + q.y, a.z + q.z, a.w + q.w));
    } else if (EPI == LDMAE_EPI_BIAS_GELU) {
      if (e.C2) put(e.C2, oc, a);
      auto g = [](float y) { return gelu_act<OutT>(y); };
      put(e.C, oc, make_float4(g(a.x), g(a.y), g(a.z), g(a.w)));
    } else if (EPI == LDMAE_EPI_GELU_BWD) {
      const OutT* p = (const OutT*)e.xin + oc;
      put(e.C, oc, make_float4(gelu_bwd_val<OutT>(a.x, to_f<OutT>(p[0])), gelu_bwd_val<OutT>(a.y, to_f<OutT>(p[1])),
                               gelu_bwd_val<OutT>(a.z, to_f<OutT>(p[2])), gelu_bwd_val<OutT>(a.w, to_f<OutT>(p[3]))));
    } else {
      if (e.C) put(e.C, oc, a);
      const size_t o = (size_t)m * N + n;
      const float4 xi = *(const float4*)(e.xin + o);
      *(float4*)(e.xout + o) = gate_res4(xi, gate, act_round4<OutT>(a));
    }
  }
};

// Epilogue shared by the NT kernels: accumulators -> per-wave f32 LDS strip [16 rows][68] (`ew`, private to the wave, so no
// workgroup barrier is needed) -> row-contiguous 16-B global accesses with the fused bias / gated residual / pos / GELU /
// SwiGLU forms.
struct NoHook { __device__ __forceinline__ void operator()(int, int) const {} };
// `consumed(i, cblk)`: called once accumulator tiles acc[i][4*cblk .. 4*cblk+3] have been copied to the strip (they are dead from then
// on): the one-wave-per-SIMD kernel re-zeroes them there on the idle matrix pipe instead of in a pass of its own.
template <int EPI, typename OutT, int TM, int TNn, int MI, int NI, typename Hook = NoHook>
__device__ __forceinline__ void nt_epilogue(f32x4 (&acc)[MI][NI], float* ew, float* ex, const EpiArgs& e, int m0, int n0, int wm, int wn,
                                            int lane, int M, int N, Hook consumed = Hook()) {   // ew: strip [16][68]; ex: 1024 more private floats
  constexpr int ELD = 68;
  auto fill = [&](int i, int cblk) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) ew[((lane >> 4) * 4 + r) * ELD + j * 16 + (lane & 15)] = acc[i][cblk * 4 + j][r];
    consumed(i, cblk);
  };
  if constexpr (EPI == LDMAE_EPI_QKV_ROPE) {
    // The qkv Linear of the DiT block with q_norm / k_norm / RoPE in the epilogue.  TNn == 64: a wave's column slice is ONE head of q, k or v, and the
    // strip lane map (8 lanes x 8 columns per row) is the lane group of qknorm_rope_fwd8_kernel (elementwise.hip) -- the same arithmetic in the same
    // order on the same bf16-rounded values, so q2 / k2 are bitwise what GEMM + that kernel produce, without its 806-MB re-read at bs 256.
    // Host contract (ldmae_gemm_nt_qkv_rope_ok): M % 256 == 0, rows_per_batch % 128 == 0 (a wave's 128 rows lie in one sample), N == 3 * heads * 64.
    static_assert(TNn == 64 && sizeof(OutT) == 2, "EPI_QKV_ROPE: 64-column wave slices, bf16 outputs");
    const int nb = n0 + wn * TNn, c8 = (lane & 7) * 8, mw = m0 + wm * TM;
    const int which = nb / (e.heads * 64), hh = (nb >> 6) % e.heads;                      // wave-uniform: 0 q, 1 k, 2 v
    const int bsmp = mw / e.rows_per_batch, tok0 = mw - bsmp * e.rows_per_batch;
    const float4 b0 = e.bias ? *(const float4*)(e.bias + nb + c8) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 b1 = e.bias ? *(const float4*)(e.bias + nb + c8 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned coff = ((unsigned)(lane >> 3) * (unsigned)e.ldc + (unsigned)(nb + c8)) * 2u;       // raw qkv: wave-uniform row base + this lane offset
    auto raw8 = [&](int i, int it, const bf16x8& o) {
      char* rb = (char*)e.C + (size_t)(mw + i * 16 + it * 8) * (size_t)e.ldc * 2;
      __builtin_nontemporal_store(o, (bf16x8*)(rb + coff));
    };
    auto round8 = [&](const float4& u, const float4& v) {
      bf16x8 o;
      o[0] = (bf16)(u.x + b0.x); o[1] = (bf16)(u.y + b0.y); o[2] = (bf16)(u.z + b0.z); o[3] = (bf16)(u.w + b0.w);
      o[4] = (bf16)(v.x + b1.x); o[5] = (bf16)(v.y + b1.y); o[6] = (bf16)(v.z + b1.z); o[7] = (bf16)(v.w + b1.w);
      return o;
    };
    if (which == 2) {                                                                       // v: the plain bias epilogue
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        fill(i, 0);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int row = it * 8 + (lane >> 3);
          raw8(i, it, round8(*(const float4*)(ew + row * ELD + c8), *(const float4*)(ew + row * ELD + c8 + 4)));
        }
      }
      return;
    }
    const bool norm = e.wq != nullptr;
    const float* wsel = which == 0 ? e.wq : e.wk;
    float wv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wv[j] = norm ? wsel[c8 + j] : 1.f;
    // this wave's 128 rows of head hh are 128 consecutive 128-B rows of q2 / k2
    char* const dstb = (char*)(which == 0 ? e.q2 : e.k2) + (((size_t)bsmp * e.heads + hh) * e.rows_per_batch + tok0) * 128 + (size_t)(lane >> 3) * 128 + c8 * 2;
    const unsigned toff = ((unsigned)(tok0 + (lane >> 3)) * 64u + (unsigned)c8) * 4u;      // tables: row tok0 + strip row, this lane's 8 columns
    float4 cs[2 * MI][2], sn[2 * MI][2];                                                    // fully unrolled: only the look-ahead's worth is live
    // table rows of half-strip s = 2 i + it, requested one HALF-strip ahead and BEFORE the previous half's stores (vmcnt retires in issue order: a
    // load issued behind a store cannot be waited for without draining that store); a whole strip ahead costs 32 more live registers and spilled
    auto ldt = [&](int s2) {
      const unsigned o = toff + (unsigned)(s2 * 8) * 256u;
      cs[s2][0] = *(const float4*)((const char*)e.cosT + o); cs[s2][1] = *(const float4*)((const char*)e.cosT + o + 16);
      sn[s2][0] = *(const float4*)((const char*)e.sinT + o); sn[s2][1] = *(const float4*)((const char*)e.sinT + o + 16);
    };
    ldt(0);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      fill(i, 0);
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int s2 = 2 * i + it;
        if (s2 + 1 < 2 * MI) ldt(s2 + 1);
        const int row = it * 8 + (lane >> 3);
        const bf16x8 x = round8(*(const float4*)(ew + row * ELD + c8), *(const float4*)(ew + row * ELD + c8 + 4));
        if (e.store_raw_qk) raw8(i, it, x);
        float xv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = (float)x[j];
        // the association of qknorm_rope_fwd8_kernel: pairs, 4-chunks, the 8-lane butterfly
        const float sq = ((xv[0] * xv[0] + xv[1] * xv[1]) + (xv[2] * xv[2] + xv[3] * xv[3])) + ((xv[4] * xv[4] + xv[5] * xv[5]) + (xv[6] * xv[6] + xv[7] * xv[7]));
        const float r = norm ? rsqrtf(group_sum<8>(sq) / 64.f + e.eps) : 1.f;
        const float4 a0 = rope_apply(make_float4(xv[0] * r * wv[0], xv[1] * r * wv[1], xv[2] * r * wv[2], xv[3] * r * wv[3]), cs[s2][0], sn[s2][0]);
        const float4 a1 = rope_apply(make_float4(xv[4] * r * wv[4], xv[5] * r * wv[5], xv[6] * r * wv[6], xv[7] * r * wv[7]), cs[s2][1], sn[s2][1]);
        bf16x8 o;
        o[0] = (bf16)a0.x; o[1] = (bf16)a0.y; o[2] = (bf16)a0.z; o[3] = (bf16)a0.w; o[4] = (bf16)a1.x; o[5] = (bf16)a1.y; o[6] = (bf16)a1.z; o[7] = (bf16)a1.w;
        __builtin_nontemporal_store(o, (bf16x8*)(dstb + (size_t)(s2 * 8) * 128));
      }
    }
    return;
  }
  if constexpr (EPI == LDMAE_EPI_SWIGLU) {
    // strip cols 0..31 = x1 (hid columns hc..), 32..63 = x2.  Values are rounded to bf16 BEFORE silu so the result
    // matches the unfused ldmae_swiglu_fwd on the stored h12 (same formula; last-bit FMA-contraction differences are possible).
    // a wave slice of TNn columns is TNn / 64 groups of [x1 32 | x2 32]
    bf16* h12 = (bf16*)e.C;
    bf16* hid = (bf16*)e.xout;
    const int Hs = N >> 1;
#pragma unroll
    for (int cblk = 0; cblk < TNn / 64; ++cblk) {
    const int hb = (n0 >> 1) + (wn * (TNn / 64) + cblk) * 32, hc = hb + (lane & 7) * 4;
    const int hcl = min(hc, Hs - 4);
    const float4 b1 = e.bias ? *(const float4*)(e.bias + hcl) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 b2 = e.bias ? *(const float4*)(e.bias + Hs + hcl) : make_float4(0.f, 0.f, 0.f, 0.f);
    // wave-uniform in-range test: the straight-line form lets the compiler count vmcnt over the stores (a per-lane guard
    // around them made it drain every store before the next strip: 8 us per tile instead of 4)
    const bool whole = m0 + wm * TM + TM <= M && hb + 32 <= Hs;
    auto strip = [&](int i, bool guard) {
      fill(i, cblk);
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int row = it * 8 + (lane >> 3), m = m0 + wm * TM + i * 16 + row;
        const float4 u = *(const float4*)(ew + row * ELD + (lane & 7) * 4), v = *(const float4*)(ew + row * ELD + 32 + (lane & 7) * 4);
        if (!guard || (m < M && hc < Hs)) {
          bf16x4 x1, x2, ho;
          x1[0] = (bf16)(u.x + b1.x); x1[1] = (bf16)(u.y + b1.y); x1[2] = (bf16)(u.z + b1.z); x1[3] = (bf16)(u.w + b1.w);
          x2[0] = (bf16)(v.x + b2.x); x2[1] = (bf16)(v.y + b2.y); x2[2] = (bf16)(v.z + b2.z); x2[3] = (bf16)(v.w + b2.w);
#pragma unroll
          for (int j = 0; j < 4; ++j) { const float a = (float)x1[j]; ho[j] = (bf16)(a * fast_sigmoid(a) * (float)x2[j]); }
          if (h12) {                                   // wave-uniform; NULL = forward-only call: h12 is only read by the backward pass
            *(bf16x4*)(h12 + (size_t)m * N + hc) = x1;
            *(bf16x4*)(h12 + (size_t)m * N + Hs + hc) = x2;
          }
          *(bf16x4*)(hid + (size_t)m * Hs + hc) = ho;
        }
      }
    };
    if (whole) {
#pragma unroll
      for (int i = 0; i < MI; ++i) strip(i, false);
    } else {
#pragma unroll
      for (int i = 0; i < MI; ++i) strip(i, true);
    }
    }
    return;
  }
  if constexpr (EPI == LDMAE_EPI_SWIGLU_BWD) {
    // acc = dhid (N = Hs columns); a,b = h12[:, n], h12[:, Hs+n]; dh12 = (g*b*s*(1+a(1-s)), g*a*s) with g rounded to bf16 first.
    // Every global access is 16 B per lane: a lane owns 8 columns of one row, 8 lanes a 128-B line, one wave instruction 8 whole rows
    // (the 8-B form, 4 rows x 128 B per instruction, issued twice the loads and stores for the same lines: the epilogue is bound by
    // the number of VMEM instructions it has to issue and retire in order, not by bytes).  Hs % 8 == 0 (host check).
    const int Hs = N;
    const bf16* h12 = (const bf16*)e.xin;
    bf16* dh12 = (bf16*)e.C;
#pragma unroll
    for (int cblk = 0; cblk < TNn / 64; ++cblk) {
      const int n = n0 + wn * TNn + cblk * 64 + (lane & 7) * 8;
      const int mw = m0 + wm * TM;
      const bool inr = n < Hs;
      // h12 rows of strip i+1 are requested before the stores of strip i (see the note on vmcnt order below)
#ifndef NT_HPF
#define NT_HPF 1
#endif
      constexpr int HPF = NT_HPF;                   // h12 rows are requested HPF strips ahead of their use
      bf16x8 hv[MI][2][2];
      const bool whole = mw + TM <= M && n0 + wn * TNn + cblk * 64 + 64 <= Hs;     // wave-uniform: straight-line stores (see SwiGLU fwd)
      // whole tiles: wave-uniform row base (SGPR pair) + one 32-bit lane offset shared by every access of the epilogue -- the loads and
      // stores take the saddr form and no 64-bit address lives in VGPRs (the per-access v_lshl_add_u64 / v_mad_i64 chains and their
      // registers were what kept the h12 look-ahead at one strip)
      const unsigned loff = (unsigned)(lane >> 3) * (unsigned)(4 * Hs) + (unsigned)(n0 + wn * TNn + cblk * 64 + (lane & 7) * 8) * 2u;
      auto ldh = [&](int i) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          if (whole) {
            const char* rb = (const char*)h12 + (size_t)(mw + i * 16 + it * 8) * (size_t)(4 * Hs);
            hv[i][it][0] = *(const bf16x8*)(rb + loff); hv[i][it][1] = *(const bf16x8*)(rb + 2 * Hs + loff);
          } else {
            const int m = min(mw + i * 16 + it * 8 + (lane >> 3), M - 1);
            const bf16* hp = h12 + (size_t)m * 2 * Hs + (inr ? n : 0);
            hv[i][it][0] = *(const bf16x8*)hp; hv[i][it][1] = *(const bf16x8*)(hp + Hs);
          }
        }
      };
      // bias gradient of w12 = column sums of dh12 AS STORED (bf16), formed here while the values are in registers: per wave the
      // sums over its 128 rows go to e.xout[(m0 / 128 + wm)][2 * Hs] (one partial row per 128 output rows; summed by the caller).
      // Kept in the wave's LDS scratch ex[row group q = lane >> 3][da 64 | db 64] (sixteen more live registers spilled the kernel).
      float* exq = ex + (lane >> 3) * 128 + (lane & 7) * 8;
      const bool sums = e.xout != nullptr;
      if (sums) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        *(float4*)exq = z; *(float4*)(exq + 4) = z; *(float4*)(exq + 64) = z; *(float4*)(exq + 68) = z;
      }
      auto strip = [&](int i, bool guard) {
        fill(i, cblk);
        // look-ahead grows as accumulator registers die: one strip ahead while strips 0 / 1 still hold 112 / 96 of them, two from then on
        if (HPF == 1) { if (i + 1 < MI) ldh(i + 1); }
        else if (i == 0) ldh(1);
        else if (i == 1) { ldh(2); ldh(3); }
        else if (i + 2 < MI) ldh(i + 2);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int row = it * 8 + (lane >> 3), m = mw + i * 16 + row;
          const float4 g0 = *(const float4*)(ew + row * ELD + (lane & 7) * 8), g1 = *(const float4*)(ew + row * ELD + (lane & 7) * 8 + 4);
          if (!guard || (m < M && inr)) {
            const bf16x8 av = hv[i][it][0], bv = hv[i][it][1];
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            bf16x8 da, db;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float g = (float)(bf16)gg[j], a = (float)av[j], b = (float)bv[j], sg = fast_sigmoid(a);
              da[j] = (bf16)(g * b * sg * (1.f + a * (1.f - sg)));
              db[j] = (bf16)(g * a * sg);
            }
            if (sums) {
              float4 ua = *(float4*)exq, ua2 = *(float4*)(exq + 4), ub = *(float4*)(exq + 64), ub2 = *(float4*)(exq + 68);
              ua.x += (float)da[0]; ua.y += (float)da[1]; ua.z += (float)da[2]; ua.w += (float)da[3];
              ua2.x += (float)da[4]; ua2.y += (float)da[5]; ua2.z += (float)da[6]; ua2.w += (float)da[7];
              ub.x += (float)db[0]; ub.y += (float)db[1]; ub.z += (float)db[2]; ub.w += (float)db[3];
              ub2.x += (float)db[4]; ub2.y += (float)db[5]; ub2.z += (float)db[6]; ub2.w += (float)db[7];
              *(float4*)exq = ua; *(float4*)(exq + 4) = ua2; *(float4*)(exq + 64) = ub; *(float4*)(exq + 68) = ub2;
            }
            if (!guard) {
              char* wb = (char*)dh12 + (size_t)(mw + i * 16 + it * 8) * (size_t)(4 * Hs);
              __builtin_nontemporal_store(da, (bf16x8*)(wb + loff));
              __builtin_nontemporal_store(db, (bf16x8*)(wb + 2 * Hs + loff));
            } else {
              __builtin_nontemporal_store(da, (bf16x8*)(dh12 + (size_t)m * 2 * Hs + n));
              __builtin_nontemporal_store(db, (bf16x8*)(dh12 + (size_t)m * 2 * Hs + Hs + n));
            }
          }
        }
      };
      ldh(0);
      if (whole) {
#pragma unroll
        for (int i = 0; i < MI; ++i) strip(i, false);
      } else {
#pragma unroll
        for (int i = 0; i < MI; ++i) strip(i, true);
      }
      if (sums && lane < 16 && inr && mw < M) {          // lanes 0-7 sum the da halves of the 8 row groups, lanes 8-15 the db halves
        const float* src = ex + (lane >> 3) * 64 + (lane & 7) * 8;
        float4 t = *(const float4*)src, t2 = *(const float4*)(src + 4);
#pragma unroll
        for (int qg = 1; qg < 8; ++qg) {
          const float4 u = *(const float4*)(src + qg * 128), u2 = *(const float4*)(src + qg * 128 + 4);
          t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; t2.x += u2.x; t2.y += u2.y; t2.z += u2.z; t2.w += u2.w;
        }
        float* dst = e.xout + (size_t)(mw / TM) * 2 * Hs + (lane >> 3) * Hs + n;
        *(float4*)dst = t; *(float4*)(dst + 4) = t2;
      }
    }
    return;
  }
  const bool nfast = (N % 8 == 0) && (e.ldc % 8 == 0) && (EPI != LDMAE_EPI_GATE_RES || e.rows_per_batch % 16 == 0);
  auto put8 = [&](void* base, size_t oc, float4 a, float4 b) {
    if constexpr (sizeof(OutT) == 4) { *(float4*)((float*)base + oc) = a; *(float4*)((float*)base + oc + 4) = b; }
    else {
      if constexpr (__is_same(OutT, f16)) { a = sat_f16(a, e.f16_max); b = sat_f16(b, e.f16_max); }
      typename Pack<OutT>::v8 o;
      o[0] = from_f<OutT>(a.x); o[1] = from_f<OutT>(a.y); o[2] = from_f<OutT>(a.z); o[3] = from_f<OutT>(a.w);
      o[4] = from_f<OutT>(b.x); o[5] = from_f<OutT>(b.y); o[6] = from_f<OutT>(b.z); o[7] = from_f<OutT>(b.w);
      __builtin_nontemporal_store(o, (typename Pack<OutT>::v8*)((OutT*)base + oc));      // streamed output: keeps the B tiles in L2 (+0.5..2 %)
    }
  };
#pragma unroll
  for (int cblk = 0; cblk < TNn / 64; ++cblk) {
    const int nb = n0 + wn * TNn + cblk * 64, col = (lane & 15) * 4, c8 = (lane & 7) * 8;
    const int mw = m0 + wm * TM;
    // Whole wave slice in range (and, gated form, inside one sample): per-column constants are loaded ONCE and the only loads
    // between the stores are the next strip's residual rows, issued BEFORE the current strip's stores.  vmcnt retires in
    // issue order, so a load issued after a store cannot be waited for without draining that store: a per-strip bias load
    // used to serialise the whole store tail (5.6 us per 256x256 tile instead of ~2).
    bool hoist = nfast && mw + TM <= M && nb + 64 <= N && EPI != LDMAE_EPI_BIAS_POS && (EPI != LDMAE_EPI_BIAS || e.beta == 0.f);
    if (EPI == LDMAE_EPI_GATE_RES && hoist) hoist = (mw / e.rows_per_batch) == ((mw + TM - 1) / e.rows_per_batch);
    if (hoist) {
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f), o4 = make_float4(1.f, 1.f, 1.f, 1.f);
      const float4 b0 = e.bias ? *(const float4*)(e.bias + nb + c8) : z4, b1 = e.bias ? *(const float4*)(e.bias + nb + c8 + 4) : z4;
      float4 g0 = o4, g1 = o4;
      if (EPI == LDMAE_EPI_GATE_RES && e.gate) {
        const float* gp = e.gate + (size_t)(mw / e.rows_per_batch) * e.gate_ld + nb + c8;
        g0 = *(const float4*)gp; g1 = *(const float4*)(gp + 4);
      }
#ifndef NT_XPF
#define NT_XPF 1
#endif
      constexpr int XPF = NT_XPF;                    // residual rows are requested XPF strips ahead of their use (2: one ahead for strips 0 / 1)
      float4 xi[MI][2][2];                           // fully unrolled: only the look-ahead's worth is live at a time
      // wave-uniform row base + one 32-bit lane offset for every residual access (saddr form: no 64-bit addresses in VGPRs)
      const unsigned xoff = ((unsigned)(lane >> 3) * (unsigned)N + (unsigned)(nb + c8)) * 4u;
      auto ldx = [&](int i) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const char* rb = (const char*)e.xin + (size_t)(mw + i * 16 + it * 8) * (size_t)N * 4;
          xi[i][it][0] = *(const float4*)(rb + xoff); xi[i][it][1] = *(const float4*)(rb + xoff + 16);
        }
      };
      // GELU backward: the pre-activation rows of a strip (8 values per lane and row) are requested one strip ahead, like the residual rows
      float pv[MI][2][8];
      const unsigned poff = ((unsigned)(lane >> 3) * (unsigned)e.ldc + (unsigned)(nb + c8)) * (unsigned)sizeof(OutT);
      auto ldp = [&](int i) {
#pragma unroll
        for (int it = 0; it < 2; ++it)
          Vec8<OutT>::load((const OutT*)((const char*)e.xin + (size_t)(mw + i * 16 + it * 8) * (size_t)e.ldc * sizeof(OutT) + poff), pv[i][it]);
      };
      if constexpr (EPI == LDMAE_EPI_GATE_RES) ldx(0);
      if constexpr (EPI == LDMAE_EPI_GELU_BWD) ldp(0);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        fill(i, cblk);
        float4 a[2][2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int row = it * 8 + (lane >> 3);
          float4 u = *(const float4*)(ew + row * ELD + c8), v = *(const float4*)(ew + row * ELD + c8 + 4);
          a[it][0] = make_float4(u.x + b0.x, u.y + b0.y, u.z + b0.z, u.w + b0.w);
          a[it][1] = make_float4(v.x + b1.x, v.y + b1.y, v.z + b1.z, v.w + b1.w);
        }
        if constexpr (EPI == LDMAE_EPI_GATE_RES) {
          if (XPF == 1) { if (i + 1 < MI) ldx(i + 1); }
          else if (i == 0) ldx(1);
          else if (i == 1) { ldx(2); ldx(3); }
          else if (i + 2 < MI) ldx(i + 2);
        }
        if constexpr (EPI == LDMAE_EPI_GELU_BWD) { if (i + 1 < MI) ldp(i + 1); }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int m = mw + i * 16 + it * 8 + (lane >> 3);
          const size_t oc = (size_t)m * e.ldc + nb + c8;
          const float4 p = a[it][0], q = a[it][1];
          if constexpr (EPI == LDMAE_EPI_GATE_RES) {
            char* wb = (char*)e.xout + (size_t)(mw + i * 16 + it * 8) * (size_t)N * 4;
            const float4 x0 = xi[i][it][0], x1 = xi[i][it][1];
            *(float4*)(wb + xoff) = gate_res4(x0, g0, act_round4<OutT>(p));
            *(float4*)(wb + xoff + 16) = gate_res4(x1, g1, act_round4<OutT>(q));
            if (e.C) put8(e.C, oc, p, q);
          } else if constexpr (EPI == LDMAE_EPI_BIAS_GELU) {
            auto g = [](float y) { return gelu_act<OutT>(y); };
            put8(e.C, oc, make_float4(g(p.x), g(p.y), g(p.z), g(p.w)), make_float4(g(q.x), g(q.y), g(q.z), g(q.w)));
            if (e.C2) put8(e.C2, oc, p, q);
          } else if constexpr (EPI == LDMAE_EPI_GELU_BWD) {
            const float* v = pv[i][it];
            put8(e.C, oc, make_float4(gelu_bwd_val<OutT>(p.x, v[0]), gelu_bwd_val<OutT>(p.y, v[1]), gelu_bwd_val<OutT>(p.z, v[2]), gelu_bwd_val<OutT>(p.w, v[3])),
                 make_float4(gelu_bwd_val<OutT>(q.x, v[4]), gelu_bwd_val<OutT>(q.y, v[5]), gelu_bwd_val<OutT>(q.z, v[6]), gelu_bwd_val<OutT>(q.w, v[7])));
          } else {
            put8(e.C, oc, p, q);
          }
        }
      }
      continue;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      fill(i, cblk);
      const int mb = mw + i * 16;
      if (nfast && mb + 16 <= M && nb + 64 <= N) {
        const Epi4<EPI, OutT> ep0(e, mb, nb + c8, N), ep1(e, mb, nb + c8 + 4, N);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int row = it * 8 + (lane >> 3), m = mb + row;
          float4 y0, y1, z0, z1;
          ep0.compute(m, *(const float4*)(ew + row * ELD + c8), y0, z0);
          ep1.compute(m, *(const float4*)(ew + row * ELD + c8 + 4), y1, z1);
          const size_t oc = (size_t)m * e.ldc + nb + c8;
          if (e.C) put8(e.C, oc, y0, y1);
          if (EPI == LDMAE_EPI_BIAS_GELU && e.C2) put8(e.C2, oc, z0, z1);
        }
      } else {
        for (int it = 0; it < 4; ++it) {
          const int row = it * 4 + (lane >> 4);
          const float4 v = *(const float4*)(ew + row * ELD + col);
          const float vv[4] = {v.x, v.y, v.z, v.w};
          for (int j = 0; j < 4; ++j) epi_store<EPI, OutT>(e, mb + row, nb + col + j, M, N, vv[j]);
        }
      }
    }
  }
}

