// The VMAE masked-token ENCODER as one kernel (inference, bf16 MFMA, f32 residual stream): tokenizer/models_mae.py:499-523 after the
// gather -- depth x Block (:149-187: LN -> qkv -> softmax attention -> proj -> +x ; LN -> fc1 -> exact GELU -> fc2 -> +x) -> LayerNorm
// (:369) -- for the shipped tokenizer geometry at mask_ratio 0.75: 256 kept tokens per image, width 192, 12 heads of 16, hidden 768.
//
// Why one kernel: as separate launches the block is 7 kernels whose activations (x f32, LN output, qkv, attention output, hidden) make
// 800 MB of HBM traffic per block at batch 256, ~1.9 ms per encoder pass before a single FLOP -- and every GEMM is a K = 192 tile that is
// all prologue and epilogue (profiles/r03_vmae_kernel_stats.csv: 3.4 ms per pass).  256 kept tokens are exactly what one workgroup can own:
//
//   * one 512-thread workgroup per IMAGE; wave w owns tokens 32w .. 32w+31 for the whole encoder.  The residual stream lives in registers
//     as x^T accumulator blocks (lane = token, 96 f32 per lane): it is read from HBM once and the result written once -- nothing else
//     of the activations ever leaves the CU.
//   * every Linear is computed TRANSPOSED, Y^T[f, tok] = W[f, :] . X^T[:, tok] (v32x32x16 bf16 MFMA, A = weight rows, B = activations), so
//       - LayerNorm is a per-lane reduction over the lane's 96 values + one cross-half shuffle,
//       - an accumulator block converts in registers into the B operand of the NEXT product (LN output -> qkv / fc1, GELU(fc1) -> fc2,
//         attention output -> proj): no activation goes through LDS except K and V, which every wave needs from every other wave.
//       The k-order a C/D block has when re-used as an operand is absorbed by permuting the weight columns once on the host.
//   * weights stream through a 4-slot LDS ring by LDS-DMA (global_load_lds, 1 KiB per wave instruction) in the exact order of use, stored by
//     the host as ready-made MFMA A-operand fragments (lane-linear 1-KiB images: conflict-free ds_read_b128, no address arithmetic);
//     the small f32 vectors (LayerNorm weights, biases) ride in the same slots.  One s_barrier per step of 24 MFMAs per wave, counted vmcnt.
//   * attention per pair of heads: q^T, k^T from the transposed product (a lane's 8 values ARE its QK^T operand fragment), v from the
//     un-transposed product (MFMA(A = x, B = Wv): lane = head-dim column, accumulator rows = tokens = the PV operand k-order); K and V^T go
//     to LDS as fragment images, S^T = K . Q^T keeps a query on a lane (row max / sum without shuffles), online softmax over 64-key steps,
//     P^T re-used from the accumulator as the B operand of O^T += V^T . P^T.
#include "common.h"

#include <type_traits>

// F16 (the TF32-class docking calls, MODE 1 / 2 / 3 only): fp16 operands = TF32's 10-bit mantissa; fragments keep the bf16x8 REGISTER type, the
// bits are fp16 (the blob is packed in fp16, ldmae_amd/tokenizer/fused_encoder.py); conversions saturate at +-65504 like the forward GEMMs
#define MFMA(a, b, c) vf_mfma<F16>(a, b, c)
template <bool F16> __device__ __forceinline__ f32x16 vf_mfma(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <bool F16> __device__ __forceinline__ bf16 vf_cv(float v) {            // f32 -> the operand type (as bits in a bf16 register slot)
  if constexpr (F16) return __builtin_bit_cast(bf16, from_f<f16>(sat_f16(v, 65504.f)));
  else return (bf16)v;
}
template <bool F16> __device__ __forceinline__ float vf_rnd(float v) {          // v as the operand type stores it
  if constexpr (F16) return (float)from_f<f16>(sat_f16(v, 65504.f));
  else return (float)(bf16)v;
}

#ifndef VF_TL
#define VF_TL 0
#endif
#ifndef VF_DBG
#define VF_DBG 0      // ablation builds for tools/ only: 1 = no GELU arithmetic, 2 = no softmax / PV, 3 = no LDS-DMA waits (stale weights), 4 = no MLP
#endif
namespace {
constexpr int VD = 192, VHID = 768, VH = 12, VTOK = 256;
constexpr int FRAG = 1024, PANEL = 12 * FRAG, VECB = 2048;         // a panel = 12 operand fragments; 2 KiB of f32 vectors per slot
constexpr int SLOTB = 2 * PANEL + VECB, NSLOT = 4, PIECES = SLOTB / 1024;   // 26 KiB per step, 26 DMA pieces; 4 slots: slot g - 1 stays readable during step g
constexpr int STEPS = 36;                                          // per block: 6 head pairs x 2 + 24 MLP chunks
constexpr int KVB = 32768;                                         // K [2 heads][8 key blocks][1 KiB] | V^T [8 key blocks][2 k-steps][1 KiB]
constexpr int LDS_BYTES = NSLOT * SLOTB + KVB;                     // 139,264 B

template <bool F16 = false> __device__ __forceinline__ bf16x8 cvt8(const f32x16& x, int s, float mul = 1.f) {
  bf16x8 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = vf_cv<F16>(x[8 * s + j] * mul);
  return f;
}
__device__ __forceinline__ f32x16 zero16() {
  f32x16 o;
#pragma unroll
  for (int t = 0; t < 16; ++t) o[t] = 0.f;
  return o;
}
// exact-erf GELU with erf by Abramowitz-Stegun 7.1.26 (common.h: erf_as -- the bf16 per-layer kernels use the same function)
__device__ __forceinline__ float gelu_erf(float y) { return gelu_act<bf16>(y); }
}  // namespace

// Sequences LONGER than one workgroup's 256 tokens (the docking encoder `_encode` runs all 1024 patches of an image, models_mae.py:819-833)
// keep the same machinery -- tokens on the lanes, residual stream in registers, weights through the ring from the SAME blob -- but a block
// is cut where the tiles have to meet, because every token needs every other tile's K and V:
//   MODE 1 (per 256-token tile): LayerNorm -> q^T, k^T, v^T of the block (the blob's twelve attention steps), staged through LDS and
//           written as whole 64-B segments into the packed token-major qkv [tokens][3][12][16] that the flash kernel reads
//           (ldmae_attention_fwd_qkv: attention.hip, any N);
//   MODE 2 (per tile): attention output rows -> operand fragments straight from global memory, proj (+ bias) into the residual stream,
//           LayerNorm -> fc1 -> GELU -> fc2 exactly as MODE 0, x written back (or the closing LayerNorm after the last block).
//   MODE 3 = MODE 2 followed, with the residual stream still in registers, by MODE 1 for the NEXT block (its twelve steps are simply the
//           ring's next slots): per block one launch of it and one of the flash kernel; MODE 1 alone only opens the stack, MODE 2 closes it.
// Per block the activations that reach HBM are x (f32, read once, written once), qkv and the attention output: 1.2 GB at 256 images
// against 3.2 GB for the per-layer kernels, in 2 launches instead of 7.
constexpr int STG_PITCH = 144;                                     // bytes per token row of the MODE 1 staging image (64 features + pad)
constexpr int LDS_BYTES_QKV = NSLOT * SLOTB + VTOK * STG_PITCH;    // 143,360 B

// MODE 0: x, out: [B, 256, 192] f32, nblk = number of blocks.  MODE 1 / 2 / 3: x, out [tiles * 256, 192] f32, nblk = INDEX of the block.
// blob: 36 slots of SLOTB bytes per block (ldmae_amd/tokenizer/fused_encoder.py packs it; layout in the header).
template <int MODE, bool F16 = false>
__global__ __launch_bounds__(512) void vmae_encoder_kernel(const float* __restrict__ x, float* __restrict__ out, const char* __restrict__ blob,
                                                           int nblk, float eps, float qscale, bf16* __restrict__ qkv, const bf16* __restrict__ oatt, int last,
                                                           unsigned* __restrict__ qkmax, int tpi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* kv = smem + NSLOT * SLOTB;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, h = lane >> 5;
  const int total = MODE == 0 ? nblk * STEPS : (MODE == 1 ? 12 : (MODE == 2 ? 30 : 42));
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(void, smem));
  const bool extra = wave < PIECES - 24;                            // waves 0, 1 carry the two vector pieces
  // ring step g -> slot of the blob: MODE 1 walks the block's attention steps, MODE 2 the proj halves (odd attention steps) and the MLP,
  // MODE 3 then runs on into the next block's attention steps (g + 6 >= 36)
  auto slot_of = [&](int g) { return MODE == 0 ? g : nblk * STEPS + (MODE == 1 ? g : (g < 6 ? 2 * g + 1 : g + 6)); };
  auto issue = [&](int g) {
    const char* src = blob + (size_t)slot_of(g) * SLOTB;
    const unsigned dst = lds0 + (g % NSLOT) * SLOTB;
#pragma unroll
    for (int i = 0; i < 3; ++i) glds16_s(src, (unsigned)((wave + 8 * i) * 1024 + lane * 16), dst + (wave + 8 * i) * 1024);
    if (extra) glds16_s(src, (unsigned)((24 + wave) * 1024 + lane * 16), dst + (24 + wave) * 1024);
  };
  issue(0);
  if (total > 1) issue(1);

  // residual stream: xacc[d][4g + j] = x[token][32 d + 8 g + 4 h + j]   (C/D block d: rows = features 32d .. 32d+31, column = token)
#if VF_TL
  unsigned long long tl[20] = {};
#endif
  f32x16 xacc[6];
  {
    const float* xr = x + ((size_t)blockIdx.x * VTOK + wave * 32 + r) * VD;
#pragma unroll
    for (int d = 0; d < 6; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 v = *(const float4*)(xr + 32 * d + 8 * g + 4 * h);
        xacc[d][4 * g] = v.x; xacc[d][4 * g + 1] = v.y; xacc[d][4 * g + 2] = v.z; xacc[d][4 * g + 3] = v.w;
      }
  }
  bf16x8 of2[MODE >= 2 ? 6 : 1][2];            // MODE 2 / 3: the attention output rows of this lane's token as proj operand fragments (k-order: see `of` below)
  if constexpr (MODE >= 2) {
    const bf16* orow = oatt + ((size_t)blockIdx.x * VTOK + wave * 32 + r) * VD + 4 * h;
#pragma unroll
    for (int hp = 0; hp < 6; ++hp)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        union { bf16x8 v; bf16x4 q[2]; } u;
        u.q[0] = *(const bf16x4*)(orow + (2 * hp + e) * 16);
        u.q[1] = *(const bf16x4*)(orow + (2 * hp + e) * 16 + 8);
        of2[hp][e] = u.v;
      }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);          // the x loads have landed (and with them the two prologue slots): no compiler-counted wait inside the loop

  bf16x8 xf[12];                               // LayerNorm output as operand fragments: k-step ks = features 16 ks .. 16 ks + 15
  auto layernorm = [&](const float* w, const float* b, auto&& emit) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < 6; ++d)
#pragma unroll
      for (int t = 0; t < 16; ++t) s += xacc[d][t];
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.f / VD);
    float v = 0.f;
#pragma unroll
    for (int d = 0; d < 6; ++d)
#pragma unroll
      for (int t = 0; t < 16; ++t) { const float c = xacc[d][t] - mean; v += c * c; }
    v += __shfl_xor(v, 32, 64);
    const float rstd = rsqrtf(v * (1.f / VD) + eps);
#pragma unroll
    for (int d = 0; d < 6; ++d) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 w4 = *(const float4*)(w + 32 * d + 8 * g + 4 * h), b4 = *(const float4*)(b + 32 * d + 8 * g + 4 * h);
        emit(d, g, make_float4((xacc[d][4 * g] - mean) * rstd * w4.x + b4.x, (xacc[d][4 * g + 1] - mean) * rstd * w4.y + b4.y,
                               (xacc[d][4 * g + 2] - mean) * rstd * w4.z + b4.z, (xacc[d][4 * g + 3] - mean) * rstd * w4.w + b4.w));
      }
      __builtin_amdgcn_sched_barrier(0);       // keep the 48 weight / bias reads from being hoisted in one batch (192 registers)
    }
  };
  auto to_frags = [&](int d, int g, float4 y) {      // element 4g + j of block d -> fragment 2d + (g >> 1), slot 4 (g & 1) + j
    const int f = 2 * d + (g >> 1), o = 4 * (g & 1);
    xf[f][o] = vf_cv<F16>(y.x); xf[f][o + 1] = vf_cv<F16>(y.y); xf[f][o + 2] = vf_cv<F16>(y.z); xf[f][o + 3] = vf_cv<F16>(y.w);
  };
  auto add_rowvec = [&](const float* v) {             // xacc += v[feature] (proj / fc2 bias)
#pragma unroll
    for (int d = 0; d < 6; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = *(const float4*)(v + 32 * d + 8 * g + 4 * h);
        xacc[d][4 * g] += b4.x; xacc[d][4 * g + 1] += b4.y; xacc[d][4 * g + 2] += b4.z; xacc[d][4 * g + 3] += b4.w;
      }
  };
  auto add_rows32 = [&](f32x16& a, const float* v) {   // a[4g + j] += v[8 g + 4 h + j]: per-row bias of one 32-row block
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 b4 = *(const float4*)(v + 8 * g + 4 * h);
      a[4 * g] += b4.x; a[4 * g + 1] += b4.y; a[4 * g + 2] += b4.z; a[4 * g + 3] += b4.w;
    }
  };

  // 12 weight fragments of a panel against the LayerNorm output, software-pipelined in groups of four: the next group's ds_read_b128 are in
  // flight under the current group's MFMAs (a plain loop exposed one LDS round trip per group: 40 % of the wave cycles were parked).
  // SWAP: un-transposed product (A = activations, B = weights).
  auto mm12 = [&](f32x16 acc, const char* p, auto SWAP) -> f32x16 {
    constexpr bool swap = decltype(SWAP)::value;
    bf16x8 fa[4], fb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) fa[j] = *(const bf16x8*)(p + j * FRAG);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = *(const bf16x8*)(p + (4 + j) * FRAG);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = swap ? MFMA(xf[j], fa[j], acc) : MFMA(fa[j], xf[j], acc);
#pragma unroll
    for (int j = 0; j < 4; ++j) fa[j] = *(const bf16x8*)(p + (8 + j) * FRAG);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = swap ? MFMA(xf[4 + j], fb[j], acc) : MFMA(fb[j], xf[4 + j], acc);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = swap ? MFMA(xf[8 + j], fa[j], acc) : MFMA(fa[j], xf[8 + j], acc);
    return acc;
  };
  // 6 feature blocks x 2 k-steps of a [192 x 32] panel into the residual stream, same pipelining
  auto mm_out = [&](const char* p, bf16x8 b0, bf16x8 b1) {
    bf16x8 fa[4], fb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) fa[j] = *(const bf16x8*)(p + j * FRAG);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = *(const bf16x8*)(p + (4 + j) * FRAG);
    __builtin_amdgcn_sched_barrier(0);
    xacc[0] = MFMA(fa[0], b0, xacc[0]); xacc[0] = MFMA(fa[1], b1, xacc[0]); xacc[1] = MFMA(fa[2], b0, xacc[1]); xacc[1] = MFMA(fa[3], b1, xacc[1]);
#pragma unroll
    for (int j = 0; j < 4; ++j) fa[j] = *(const bf16x8*)(p + (8 + j) * FRAG);
    __builtin_amdgcn_sched_barrier(0);
    xacc[2] = MFMA(fb[0], b0, xacc[2]); xacc[2] = MFMA(fb[1], b1, xacc[2]); xacc[3] = MFMA(fb[2], b0, xacc[3]); xacc[3] = MFMA(fb[3], b1, xacc[3]);
    __builtin_amdgcn_sched_barrier(0);
    xacc[4] = MFMA(fa[0], b0, xacc[4]); xacc[4] = MFMA(fa[1], b1, xacc[4]); xacc[5] = MFMA(fa[2], b0, xacc[5]); xacc[5] = MFMA(fa[3], b1, xacc[5]);
  };
  // one ring step: own pieces of slot g have landed (slot g + 1's may stay in flight); the barrier then makes everyone's visible and
  // tells that every wave is done with slot g - 1, which the DMA of slot g + 2 overwrites.  Returns this lane's fragment base in slot g.
  int g = 0;
  auto next_slot = [&]() -> const char* {
    constexpr int Q0 = MODE == 3 ? 30 : 0;       // first q | k | v step of MODE 1 / 3
    if (MODE == 1 || (MODE == 3 && g >= Q0)) {
      // behind slot g's pieces in the (in-order) vector-memory queue: the vector-memory instructions of step g - 2, the pieces of slot
      // g + 1, those of step g - 1 (a q | k step: 4 norm-maximum atomics + 4 qkv stores per wave, a v step: 2 stores: 10 per pair of steps)
      if (g + 1 < total) {
        if (g >= Q0 + 2) { if (extra) asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); }
        else if (g == Q0 + 1) { if (extra) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); }
        else { if (extra) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
      } else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    } else if (g + 1 < total) {
      if (extra) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (g + 2 < total) issue(g + 2);
    const char* slot = smem + (g % NSLOT) * SLOTB;
    ++g;
    return slot;
  };
  // ================= q | k | v of a block for this tile's 256 tokens -> packed qkv (natural feature order: the weight rows are not permuted)
  auto qkv_phase = [&]() {
    char* const stg = kv;
    auto stage_t = [&](const f32x16& a, int byteoff) {        // element 4g + j = feature 8g + 4h + j of the 32-row block, this lane's token
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        bf16x4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = vf_cv<F16>(a[4 * gq + j]);
        *(bf16x4*)(stg + (wave * 32 + r) * STG_PITCH + byteoff + (8 * gq + 4 * h) * 2) = w;
      }
    };
    bf16* const qrow = qkv + (size_t)blockIdx.x * VTOK * (3 * VD);
    const char* slot0 = next_slot();
    layernorm((const float*)(slot0 + 2 * PANEL), (const float*)(slot0 + 2 * PANEL) + VD, to_frags);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int hp = 0; hp < 6; ++hp) {
      {
        const char* slot = hp ? next_slot() : slot0;
        const char* p0 = slot + lane * 16;
        const char* p1 = p0 + PANEL;
        const float* vec = (const float*)(slot + 2 * PANEL);
        f32x16 qT = mm12(zero16(), p0, std::false_type{});
        add_rows32(qT, vec + 384);
        stage_t(qT, 0);
        f32x16 kT = mm12(zero16(), p1, std::false_type{});
        add_rows32(kT, vec + 416);
        stage_t(kT, 64);
        // max |q|^2 and max |k|^2 over this tile's tokens, per head, as stored (bf16): they bound every score of the (image, head), and the
        // flash kernel then needs no running maximum (attention.hip: ldmae_attention_fwd_qkv_bounded).  Non-negative floats order like their bits.
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float qs = 0.f, ks = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float a = vf_rnd<F16>(qT[8 * e + j]), b = vf_rnd<F16>(kT[8 * e + j]);
            qs += a * a; ks += b * b;
          }
          qs += __shfl_xor(qs, 32, 64); ks += __shfl_xor(ks, 32, 64);
#pragma unroll
          for (int o = 16; o > 0; o >>= 1) { qs = fmaxf(qs, __shfl_xor(qs, o, 64)); ks = fmaxf(ks, __shfl_xor(ks, o, 64)); }
          if (lane == 0) {
            unsigned* dst = qkmax + ((size_t)(blockIdx.x / tpi) * VH + 2 * hp + e) * 2;
            atomicMax(dst, __float_as_uint(qs));
            atomicMax(dst + 1, __float_as_uint(ks));
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) {                          // 256 rows x (4 chunks of q | 4 of k): 64-B segments of the token's q and k slots
          const int idx = i * 512 + threadIdx.x, row = idx >> 3, ch = idx & 7;
          *(bf16x8*)(qrow + (size_t)row * (3 * VD) + (ch >> 2) * VD + hp * 32 + (ch & 3) * 8) = *(const bf16x8*)(stg + row * STG_PITCH + ch * 16);
        }
      }
      {
        const char* slot = next_slot();                        // (its barrier: every wave has read the q | k image)
        const char* p0 = slot + lane * 16;
        const float* vec = (const float*)(slot + 2 * PANEL);
        f32x16 vT = mm12(zero16(), p0, std::false_type{});
        add_rows32(vT, vec + 384);
        stage_t(vT, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int idx = i * 512 + threadIdx.x, row = idx >> 2, ch = idx & 3;
          *(bf16x8*)(qrow + (size_t)row * (3 * VD) + 2 * VD + hp * 32 + ch * 8) = *(const bf16x8*)(stg + row * STG_PITCH + ch * 16);
        }
      }
    }
  };
  if constexpr (MODE == 1) { qkv_phase(); return; }
  for (int blk = 0; blk < (MODE == 0 ? nblk : 1); ++blk) {
    // ================= attention branch: six pairs of heads, two ring steps each
    if constexpr (MODE >= 2) {
      // the attention ran as its own launch: proj of its output rows, one pair of heads per ring step (the blob's odd attention steps)
      for (int hp = 0; hp < 6; ++hp) {
        const char* slot = next_slot();
        const float* vec = (const float*)(slot + 2 * PANEL);
        mm_out(slot + lane * 16 + PANEL, of2[hp][0], of2[hp][1]);
        if (hp == 5) add_rowvec(vec);                          // + proj bias
      }
    }
    for (int hp = 0; hp < (MODE == 0 ? 6 : 0); ++hp) {
      bf16x8 qf[2];                            // this wave's queries of the head pair (operand fragments, scale * log2(e) folded in)
      float qn2[2];                            // |q|^2 of this lane's query per head (of the scaled operand), for the score bound
      const float* kmx = nullptr;              // [2 heads][8 waves] max |k|^2 over a wave's tokens, in step A's slot
      {
        // ---- step A: q^T and k^T of heads 2hp, 2hp+1 (panel rows = Wqkv rows 32hp.. and 192 + 32hp..)
        const char* slot = next_slot();
        const char* p0 = slot + lane * 16;             // fragment i of panel 0: p0 + i * 1024
        const char* p1 = p0 + PANEL;
        const float* vec = (const float*)(slot + 2 * PANEL);   // [0..191] row vector #1, [192..383] #2, [384..511] misc
        if (hp == 0) layernorm(vec, vec + VD, to_frags);
        f32x16 qT = mm12(zero16(), p0, std::false_type{});
        add_rows32(qT, vec + 384);
        // a lane's elements 8e .. 8e+7 are head e's 8 head-dim values of its half: exactly the 16-B operand fragment of S^T = K . Q^T
        // (both operands in the same implicit order of the head dim, which a dot product does not care about)
        qf[0] = cvt8<F16>(qT, 0, qscale); qf[1] = cvt8<F16>(qT, 1, qscale);
        f32x16 kT = mm12(zero16(), p1, std::false_type{});
        add_rows32(kT, vec + 416);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const bf16x8 kf = cvt8<F16>(kT, e);
          *(bf16x8*)(kv + (e * 8 + wave) * FRAG + lane * 16) = kf;
          // |k|^2 of this lane's token (as stored), maximum over the wave's 32 tokens -> the slot's spare vector floats (readable through
          // the next step): with |q| it bounds every score of the head, and the softmax below needs no maximum pass (see `bq`)
          float ks = 0.f, qs = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) { ks += (float)kf[j] * (float)kf[j]; qs += (float)qf[e][j] * (float)qf[e][j]; }
          ks += __shfl_xor(ks, 32, 64);
          qn2[e] = qs + __shfl_xor(qs, 32, 64);
#pragma unroll
          for (int o = 16; o > 0; o >>= 1) ks = fmaxf(ks, __shfl_xor(ks, o, 64));
          if (lane == 0) ((float*)vec)[448 + e * 8 + wave] = ks;
        }
        kmx = vec + 448;
      }
      {
        // ---- step B: v (un-transposed: rows = tokens, columns = 2 heads x 16), attention of both heads, proj into the residual stream
        const char* slot = next_slot();
        const char* p0 = slot + lane * 16;
        const char* p1 = p0 + PANEL;
        const float* vec = (const float*)(slot + 2 * PANEL);
        f32x16 v = mm12(zero16(), p0, std::true_type{});
        const float bv = vec[384 + r];
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) v[tt] += bv;
        // V^T image: [key block = wave][k-step s][lane = (h, column d)]: elements 8s .. 8s+7 are tokens in the accumulator k-order
        *(bf16x8*)(kv + 16384 + (wave * 2 + 0) * FRAG + lane * 16) = cvt8<F16>(v, 0);
        *(bf16x8*)(kv + 16384 + (wave * 2 + 1) * FRAG + lane * 16) = cvt8<F16>(v, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // K (written in step A) and V^T of all 256 tokens are in LDS
        bf16x8 of[2];                                        // o^T of the two heads as proj operand fragments
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          if (VF_DBG == 2) { of[e] = qf[e]; continue; }
          // The shift of the softmax: bq = |q| max_j |k_j| (Cauchy-Schwarz; q carries scale * log2(e)) bounds every score of this query,
          // so with it every exponent lies in [-2 bq, 0] -- nothing overflows, and up to bq = 50 nothing flushes to zero: exact, and no
          // pass over the keys for the maximum.  A wave with a larger bound takes that pass (a score product is ONE MFMA at head dim 16,
          // cheaper than an online rescale of O).  Either way the score chains start from -shift and p = exp2(score) is one instruction.
          const float4 k0 = *(const float4*)(kmx + e * 8), k1 = *(const float4*)(kmx + e * 8 + 4);
          const float km2 = fmaxf(fmaxf(fmaxf(k0.x, k0.y), fmaxf(k0.z, k0.w)), fmaxf(fmaxf(k1.x, k1.y), fmaxf(k1.z, k1.w)));
          float mx = sqrtf(qn2[e] * km2) * 1.002f + 0.01f;
          if (__builtin_amdgcn_ballot_w64(!(mx <= 50.f)) != 0) {
            f32x16 mv = MFMA(*(const bf16x8*)(kv + (e * 8) * FRAG + lane * 16), qf[e], zero16());
#pragma unroll
            for (int kb = 1; kb < 8; ++kb) {
              const f32x16 sc = MFMA(*(const bf16x8*)(kv + (e * 8 + kb) * FRAG + lane * 16), qf[e], zero16());
#pragma unroll
              for (int tt = 0; tt < 16; ++tt) mv[tt] = fmaxf(mv[tt], sc[tt]);
            }
            mx = mv[0];
#pragma unroll
            for (int tt = 1; tt < 16; ++tt) mx = fmaxf(mx, mv[tt]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
          }
          f32x16 nm;
#pragma unroll
          for (int tt = 0; tt < 16; ++tt) nm[tt] = -mx;
          f32x16 o = zero16();
          float l = 0.f;
#pragma unroll
          for (int kb = 0; kb < 8; ++kb) {
            f32x16 sc = MFMA(*(const bf16x8*)(kv + (e * 8 + kb) * FRAG + lane * 16), qf[e], nm);
#pragma unroll
            for (int tt = 0; tt < 16; ++tt) { sc[tt] = __builtin_amdgcn_exp2f(sc[tt]); l += sc[tt]; }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
              o = MFMA(*(const bf16x8*)(kv + 16384 + (kb * 2 + s2) * FRAG + lane * 16), cvt8<F16>(sc, s2), o);
            __builtin_amdgcn_sched_barrier(0);
          }
          l += __shfl_xor(l, 32, 64);
          // rows 16e .. 16e+15 of o^T (elements 8e .. 8e+7) belong to head e; the other half multiplied the other head's V: dropped
          of[e] = cvt8<F16>(o, e, 1.f / l);
        }
        // proj: x^T += Wp[:, 32hp .. 32hp+31] . o_pair^T   (panel 1: 6 feature blocks x 2 k-steps; k-step e = head e of the pair)
        mm_out(p1, of[0], of[1]);
        if (hp == 5) add_rowvec(vec);                        // + proj bias, once per block
      }
    }
    // ================= MLP: 24 chunks of 32 hidden units (panel 0 = fc1 rows, panel 1 = fc2 columns of the same units), skewed by one
    // chunk inside the wave: step c runs fc1 of chunk c with the GELU of chunk c - 1 spread between its MFMAs (the GELU is ~20 vector
    // instructions per element and was the longest phase of the step, with the matrix pipe idle), then fc2 of chunk c - 1 out of the
    // PREVIOUS slot's panel 1 -- which the 4-slot ring keeps readable for one more step.
    {
#if VF_TL
#define TLS(k) if (blk == 1 && c >= 4 && c < 8) tl[(c - 4) * 5 + (k)] = __builtin_amdgcn_s_memtime()
#else
#define TLS(k)
#endif
      f32x16 hprev = zero16();
      const char* p1prev = nullptr;
      const float* vec = nullptr;
      auto gelu4 = [&](f32x16& hh, int q) {
#pragma unroll
        for (int tt = 4 * q; tt < 4 * q + 4; ++tt)
          if (VF_DBG != 1) hh[tt] = gelu_erf(hh[tt]);
      };
      for (int c = 0; c < 24; ++c) {
        TLS(0);
        const char* slot = next_slot();
        TLS(1);
        if (VF_DBG == 4) continue;
        const char* p0 = slot + lane * 16;
        vec = (const float*)(slot + 2 * PANEL);
        if (c == 0) layernorm(vec, vec + VD, to_frags);
        f32x16 hT = zero16();
        bf16x8 fa[3], fb[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) fa[j] = *(const bf16x8*)(p0 + j * FRAG);
#pragma unroll
        for (int q = 0; q < 4; ++q) {            // 3 fc1 MFMAs + 4 GELUs of the previous chunk per group; next group's fragments in flight
          if (q < 3) {
#pragma unroll
            for (int j = 0; j < 3; ++j) (q & 1 ? fa : fb)[j] = *(const bf16x8*)(p0 + (3 * (q + 1) + j) * FRAG);
          }
          // the fc1 MFMAs form ONE dependency chain (same accumulator): an in-order wave that meets the next MFMA before the previous one
          // has finished just waits, so a GELU sits BETWEEN every two of them (scheduling fences keep the compiler from regrouping)
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            hT = MFMA((q & 1 ? fb : fa)[j], xf[3 * q + j], hT);
            if (c > 0 && VF_DBG != 1) hprev[4 * q + j] = gelu_erf(hprev[4 * q + j]);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (c > 0 && VF_DBG != 1) hprev[4 * q + 3] = gelu_erf(hprev[4 * q + 3]);
        }
        TLS(2);
        add_rows32(hT, vec + 384);
        const bf16x8 h0 = cvt8<F16>(hprev, 0), h1 = cvt8<F16>(hprev, 1);
        TLS(3);
        if (c > 0) mm_out(p1prev, h0, h1);
        TLS(4);
        hprev = hT;
        p1prev = p0 + PANEL;
      }
      if (VF_DBG != 4) {                         // drain: GELU and fc2 of the last chunk, + fc2 bias
#pragma unroll
        for (int q = 0; q < 4; ++q) { gelu4(hprev, q); __builtin_amdgcn_sched_barrier(0); }
        mm_out(p1prev, cvt8<F16>(hprev, 0), cvt8<F16>(hprev, 1));
        add_rowvec(vec);
      }
    }
  }
  // closing LayerNorm (models_mae.py:369, 521): its weight / bias ride in the second-to-last slot, which nothing has overwritten
  const float* nv = (const float*)(smem + ((total - 2) % NSLOT) * SLOTB + 2 * PANEL);
  float* orow = out + ((size_t)blockIdx.x * VTOK + wave * 32 + r) * VD;
  if constexpr (MODE == 3) qkv_phase();        // q | k | v of the next block while x is here
  if (MODE == 3 || (MODE == 2 && !last)) {     // not the last block: the residual stream goes back as it is
#pragma unroll
    for (int d = 0; d < 6; ++d)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        *(float4*)(orow + 32 * d + 8 * g4 + 4 * h) = make_float4(xacc[d][4 * g4], xacc[d][4 * g4 + 1], xacc[d][4 * g4 + 2], xacc[d][4 * g4 + 3]);
    return;
  }
  layernorm(nv, nv + VD, [&](int d, int g, float4 y) { *(float4*)(orow + 32 * d + 8 * g + 4 * h) = y; });
#if VF_TL
  if (lane == 0 && (wave == 0 || wave == 4)) {          // timing build only: the stamps overwrite the head of this wave's output rows
    unsigned long long* o = (unsigned long long*)orow;
    for (int i = 0; i < 20; ++i) o[i] = tl[i];
  }
#endif
}

extern "C" long ldmae_vmae_encoder_blob_bytes(int nblocks) { return (long)nblocks * STEPS * SLOTB; }

extern "C" int ldmae_vmae_encoder_fwd(const float* x, float* out, const void* blob, int B, int tokens, int dim, int heads, int hidden, int nblocks,
                                      float eps, void* stream) {
  LDMAE_REQUIRE(x && out && blob && B > 0, "vmae_encoder_fwd: null pointer or empty batch");
  LDMAE_REQUIRE(tokens == VTOK && dim == VD && heads == VH && hidden == VHID,
                "vmae_encoder_fwd: fused kernel is built for %d tokens x width %d, %d heads, hidden %d (got %d, %d, %d, %d): use the per-layer entry points",
                VTOK, VD, VH, VHID, tokens, dim, heads, hidden);
  LDMAE_REQUIRE(nblocks >= 1, "vmae_encoder_fwd: nblocks=%d", nblocks);
  LDMAE_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)blob & 15) == 0, "vmae_encoder_fwd: pointers must be 16-B aligned");
  hipFuncSetAttribute((const void*)vmae_encoder_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  const float qscale = 0.25f * 1.4426950408889634f;          // head_dim^-0.5 (models_mae.py:123) * log2(e)
  hipLaunchKernelGGL(vmae_encoder_kernel<0>, dim3(B), dim3(512), LDS_BYTES, as_stream(stream), x, out, (const char*)blob, nblocks, eps, qscale, (bf16*)nullptr,
                     (const bf16*)nullptr, 1, (unsigned*)nullptr, 1);
  LDMAE_CHECK_LAUNCH("vmae_encoder_fwd");
  return LDMAE_OK;
}

// ---- the same encoder on sequences of several 256-token tiles (docking `_encode`: all 1024 patches): per block qkv kernel -> flash attention
// on the packed qkv -> proj / MLP kernel.  workspace: qkv [B*tokens][576] bf16 | attention output [B*tokens][192] bf16 | lse [B][12][tokens] f32 |
// per block the norm maxima [B][12][2] f32 that give the flash kernel its static softmax shift.
extern "C" int ldmae_attention_fwd_qkv_bounded(int dtype, const void* qkv, void* o, float* lse, const float* qk_max2, int B, int H, int N, int hd,
                                               float scale, void* stream);
extern "C" long ldmae_vmae_encoder_fwd_tiled_workspace_bytes(int B, int tokens) {
  return (long)B * tokens * (3 * VD + VD) * 2 + (long)B * VH * tokens * 4 + ((long)B * VH * 2 * 4 + 15) / 16 * 16 * 64;      // (up to 64 blocks)
}
extern "C" int ldmae_attention_fwd_qkv(int dtype, const void* qkv, void* o, float* lse, int B, int H, int N, int hd, float scale, void* stream);
template <bool F16>
static int encoder_fwd_tiled(const float* x, float* out, const void* blob, void* workspace, int B, int tokens, int dim, int heads,
                             int hidden, int nblocks, float eps, void* stream) {
  LDMAE_REQUIRE(x && out && blob && workspace && B > 0, "vmae_encoder_fwd_tiled: null pointer or empty batch");
  LDMAE_REQUIRE(tokens > 0 && tokens % VTOK == 0 && dim == VD && heads == VH && hidden == VHID,
                "vmae_encoder_fwd_tiled: built for whole %d-token tiles of width %d, %d heads, hidden %d (got %d, %d, %d, %d): use the per-layer entry points",
                VTOK, VD, VH, VHID, tokens, dim, heads, hidden);
  LDMAE_REQUIRE(nblocks >= 1 && nblocks <= 64, "vmae_encoder_fwd_tiled: nblocks=%d (1 .. 64)", nblocks);
  LDMAE_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)blob & 15) == 0 && ((uintptr_t)workspace & 15) == 0,
                "vmae_encoder_fwd_tiled: pointers must be 16-B aligned");
  LDMAE_REQUIRE((long)B * tokens / VTOK < (1L << 31), "vmae_encoder_fwd_tiled: too many tiles");
  hipFuncSetAttribute((const void*)vmae_encoder_kernel<1, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_QKV);
  hipFuncSetAttribute((const void*)vmae_encoder_kernel<2, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipFuncSetAttribute((const void*)vmae_encoder_kernel<3, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_QKV);
  bf16* qkv = (bf16*)workspace;
  bf16* oatt = qkv + (size_t)B * tokens * 3 * VD;
  float* lse = (float*)(oatt + (size_t)B * tokens * VD);
  const size_t mxs = ((size_t)B * VH * 2 * 4 + 15) / 16 * 16;      // bytes of one block's norm maxima
  char* mx = (char*)(lse + (size_t)B * VH * tokens);
  LDMAE_REQUIRE(hipMemsetAsync(mx, 0, mxs * nblocks, as_stream(stream)) == hipSuccess, "vmae_encoder_fwd_tiled: hipMemsetAsync of the q / k norm maxima failed");
  const int tpi = tokens / VTOK;
  const unsigned tiles = (unsigned)((long)B * tokens / VTOK);
  hipLaunchKernelGGL((vmae_encoder_kernel<1, F16>), dim3(tiles), dim3(512), LDS_BYTES_QKV, as_stream(stream), x, (float*)nullptr, (const char*)blob, 0, eps, 0.f, qkv,
                     (const bf16*)nullptr, 0, (unsigned*)mx, tpi);
  LDMAE_CHECK_LAUNCH("vmae_encoder_fwd_tiled(qkv)");
  for (int blk = 0; blk < nblocks; ++blk) {
    const float* xin = blk == 0 ? x : out;                    // the residual stream lives in `out` from the first block on (a tile rewrites only its own rows)
    // scale = head_dim^-0.5 (models_mae.py:123).  bf16: the static softmax shift from the norm maxima the q | k | v kernel left behind.  fp16
    // keeps the running maximum: with a static shift the probabilities of a row whose scores sit far below the bound leave fp16's range.
    if (int e = F16 ? ldmae_attention_fwd_qkv(LDMAE_F16, qkv, oatt, lse, B, VH, tokens, VD / VH, 0.25f, stream)
                    : ldmae_attention_fwd_qkv_bounded(LDMAE_BF16, qkv, oatt, lse, (const float*)(mx + mxs * blk), B, VH, tokens, VD / VH, 0.25f, stream))
      return e;
    if (blk + 1 < nblocks)      // proj / MLP of this block, then q | k | v of the next one (the attention has consumed this block's)
      hipLaunchKernelGGL((vmae_encoder_kernel<3, F16>), dim3(tiles), dim3(512), LDS_BYTES_QKV, as_stream(stream), xin, out, (const char*)blob, blk, eps, 0.f, qkv,
                         (const bf16*)oatt, 0, (unsigned*)(mx + mxs * (blk + 1)), tpi);
    else
      hipLaunchKernelGGL((vmae_encoder_kernel<2, F16>), dim3(tiles), dim3(512), LDS_BYTES, as_stream(stream), xin, out, (const char*)blob, blk, eps, 0.f, (bf16*)nullptr,
                         (const bf16*)oatt, 1, (unsigned*)nullptr, tpi);
    LDMAE_CHECK_LAUNCH("vmae_encoder_fwd_tiled(post)");
  }
  return LDMAE_OK;
}
extern "C" int ldmae_vmae_encoder_fwd_tiled(const float* x, float* out, const void* blob, void* workspace, int B, int tokens, int dim, int heads,
                                            int hidden, int nblocks, float eps, void* stream) {
  return encoder_fwd_tiled<false>(x, out, blob, workspace, B, tokens, dim, heads, hidden, nblocks, eps, stream);
}
// The TF32-class form (f32 docking calls under torch.backends.cuda.matmul.allow_tf32): the same kernels on fp16 operands -- the blob packed in fp16
// (same layout, same size), qkv / attention output in fp16, f32 residual stream, statistics and accumulation.
extern "C" int ldmae_vmae_encoder_fwd_tiled_f16(const float* x, float* out, const void* blob, void* workspace, int B, int tokens, int dim, int heads,
                                                int hidden, int nblocks, float eps, void* stream) {
  return encoder_fwd_tiled<true>(x, out, blob, workspace, B, tokens, dim, heads, hidden, nblocks, eps, stream);
}
