// Flash-style attention for the LightningDiT block (hd = 64) on gfx950: softmax(q k^T * scale) v,
// non-causal, no mask (lightningdit.py:76-80).  q,k,v are head-major [B,H,N,hd]; o / do are
// token-major [B,N,H*hd] (the layout the proj GEMM consumes).
//
// bf16 path: mfma_f32_32x32x16_bf16.  Scores are computed TRANSPOSED (S^T = K . Q^T) so a lane owns one
// query column: the row max / row sum are in-register plus one cross-half shuffle, and the f32
// accumulator tile is re-used directly as the B operand of the next product (P^T for O^T = V^T . P^T)
// with the k-order of the 32x32 C/D map (verified by csrc/probe/mfma_probe.hip).  K/V tiles sit in LDS
// in ONE image that serves both row reads (ds_read_b128) and transposed reads (ds_read_b64_tr_b16):
// 8-row x 32-col sub-tiles of 512 B with a 2-bit XOR on the 16-B chunk; filled by global_load_lds with
// the permutation applied to the per-lane SOURCE address.
//
// The loops are bound by vector-instruction ISSUE, not by the matrix pipe (hd = 64: 16 MFMAs against 32 exponentials per 64-key tile
// and wave), so everything around the exponential is folded away: the stationary operand carries scale*log2(e), the score chains start
// from accumulators that already hold -(running max) / -lse / -delta, and tile addresses are scalar (SGPR-base LDS-DMA).
//
// Backward = two passes without atomics (bitwise reproducible):
//   dKdV: a wave owns 32 keys (key on the lane), sweeps 64-query tiles: S, dP, then dV^T += dO^T P,
//         dK^T += Q^T dS with Q / dO read both by rows and transposed from the same LDS image.
//   dQ  : a wave owns 32 queries (query on the lane), sweeps 64-key tiles: S^T, dP^T, dQ^T += K^T dS^T.
// f32 path (parity contract): same structure on mfma_f32_32x32x2f32 with padded f32 LDS tiles.
#include "common.h"

#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define MFMA_F32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

// byte offset of 16-B chunk `ch` of row `row` in the dual-use LDS image of a [rows][HD] bf16 tile
template <int HD> __device__ __forceinline__ int tile_off(int row, int ch) {
  return (HD * 16) * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
}
// inverse: LDS byte offset (multiple of 16) -> (row, ch)
template <int HD> __device__ __forceinline__ void tile_inv(int off, int& row, int& ch) {
  const int grp = off / (HD * 16), within = off % (HD * 16);
  const int sub = within >> 9, rem = within & 511, r7 = rem >> 6, pos = (rem & 63) >> 4;
  row = grp * 8 + r7;
  ch = sub * 4 + (pos ^ ((row >> 2) & 3));
}
// Head dims that are not a multiple of 32 (LightningDiT-XL: 72, VMAE: 16) are tiled with the LDS images and the register
// fragments zero-padded to HDP = the next multiple of 32; global memory keeps the true HD (no padded copies in HBM).  Zero q/k
// columns add 0 to every score, zero v / dO columns produce output columns that are never stored.  Padded 16-B chunks of an
// LDS tile are fetched from this 16 zero bytes (the DMA source address is per lane).
__device__ __attribute__((aligned(16))) unsigned g_zero16[4] = {0u, 0u, 0u, 0u};
// 1.0 (bf16) in the first of eight columns: as the first PADDED chunk of a V row it makes column HD of the V image a column of ones, and
// the P.V product then delivers the softmax row sums in accumulator row HD of the (padded) output -- for free, on the matrix pipe
__device__ __attribute__((aligned(16))) unsigned g_one16[4] = {0x00003F80u, 0u, 0u, 0u};
__device__ __attribute__((aligned(16))) unsigned g_one16h[4] = {0x00003C00u, 0u, 0u, 0u};     // the same column of ones in fp16
constexpr int hd_pad(int hd) { return (hd + 31) / 32 * 32; }

// stage a [ROWS][HD] bf16 tile (global row stride ld elements) into the [ROWS][HDP] LDS image with global_load_lds; 256 threads
template <int HD, int ROWS, bool ONES = false, bool F16 = false>      // ONES: the first padded chunk of every row comes from g_one16 (the forward kernel's V tile)
__device__ __forceinline__ void stage_tile(const bf16* __restrict__ g, long ld, int row_limit, char* lds, int wave, int lane) {
  constexpr int HDP = hd_pad(HD), PIECES = ROWS * HDP * 2 / 1024;
  static_assert(PIECES % 4 == 0 || PIECES == 2 || PIECES == 1, "tile too small");
#pragma unroll
  for (int i = 0; i < (PIECES + 3) / 4; ++i) {
    const int pi = wave * ((PIECES + 3) / 4) + i;
    if (pi < PIECES) {
      int row, ch;
      tile_inv<HDP>(pi * 1024 + lane * 16, row, ch);
      row = min(row, row_limit);
      const void* src = (HD == HDP || ch * 8 < HD) ? (const void*)(g + (long)row * ld + ch * 8)
                                                    : ((ONES && ch * 8 == HD) ? (const void*)(F16 ? g_one16h : g_one16) : (const void*)g_zero16);
      glds16(src, lds + pi * 1024);
    }
  }
}
// The same staging with the addressing hoisted out of the tile loop: per-lane byte offsets of this wave's pieces, computed once; a tile is
// then (uniform base pointer, uniform LDS address) + one LDS-DMA instruction per piece.  Padded head dims (16, 72): the lanes whose chunk
// lies past the true head dim take NO part in the DMA (the instruction runs under an EXEC mask: the hardware writes LDS bytes base + 16 *
// lane for active lanes only), and their sixteen bytes of every LDS image are written ONCE, by `prefill`, before the first tile: zeros, or
// (ONES) the ones column of the forward kernel's V tile.  Round 3 fetched those chunks from 16 zero bytes with a per-lane POINTER, which
// put the whole address computation (~70 vector instructions per tile) back into the tile loop of kernels that are bound by vector issue.
// Every piece has real lanes (checked at compile time for the instantiated head dims), so every wave still issues the same number of
// DMA instructions per tile and the counted vmcnt waits hold.
template <int HD, int ROWS> struct TileMap {
  static constexpr int HDP = hd_pad(HD), PIECES = ROWS * HDP * 2 / 1024, NPW = (PIECES + 3) / 4;
  unsigned voff[NPW];
  unsigned real;                 // bit i: this lane's chunk of piece i lies inside the true head dim
  __device__ __forceinline__ void init(long ld, int wave, int lane) {
    real = 0;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      int row, ch;
      tile_inv<HDP>((wave * NPW + i) * 1024 + lane * 16, row, ch);
      voff[i] = (unsigned)((row * ld + ch * 8) * 2);
      if (ch * 8 < HD) real |= 1u << i;
    }
  }
  // the padded chunks of one LDS tile image (call once per image, then lgkmcnt(0) + a barrier before the first read)
  template <bool ONES = false, bool F16 = false> __device__ __forceinline__ void prefill(char* lds, int wave, int lane) const {
    if constexpr (HD != HDP) {
#pragma unroll
      for (int i = 0; i < NPW; ++i)
        if (wave * NPW + i < PIECES && !((real >> i) & 1u)) {
          int row, ch;
          tile_inv<HDP>((wave * NPW + i) * 1024 + lane * 16, row, ch);
          *(uint4*)(lds + (wave * NPW + i) * 1024 + lane * 16) = make_uint4((ONES && ch * 8 == HD) ? (F16 ? 0x00003C00u : 0x00003F80u) : 0u, 0u, 0u, 0u);
        }
    }
  }
  // g: first element of the tile (wave-uniform); lds_addr: LDS byte address of the tile's image (wave-uniform)
  __device__ __forceinline__ void issue(const bf16* g, long, char*, unsigned lds_addr, int wave, int) const {
#pragma unroll
    for (int i = 0; i < NPW; ++i)
      if (wave * NPW + i < PIECES) {
        if constexpr (HD == HDP) glds16_s(g, voff[i], lds_addr + (wave * NPW + i) * 1024);
        else if ((real >> i) & 1u) glds16_s(g, voff[i], lds_addr + (wave * NPW + i) * 1024);       // EXEC-masked
      }
  }
};
__device__ __forceinline__ unsigned lds_addr_of(const void* p) { return __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(void, p)); }
// register fragment times a scalar, rounded back to bf16 (the stationary operand of a score product carries scale * log2(e))
__device__ __forceinline__ bf16x8 frag_scale(const bf16x8& a, float c) {
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)a[j] * c);
  return o;
}
// the fp16 forms (the forward kernel's TF32-class instantiation: fragments keep the bf16x8 REGISTER type, the bits are fp16)
__device__ __forceinline__ bf16x8 frag_scale_h(const bf16x8& a, float c) {
  const f16x8 x = __builtin_bit_cast(f16x8, a);
  f16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = from_f<f16>((float)x[j] * c);
  return __builtin_bit_cast(bf16x8, o);
}
template <bool F16> __device__ __forceinline__ f32x16 mfma_att(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <bool F16> __device__ __forceinline__ bf16x8 frag_scale_t(const bf16x8& a, float c) {
  if constexpr (F16) return frag_scale_h(a, c); else return frag_scale(a, c);
}
template <bool F16> struct ActT { typedef bf16 t; };
template <> struct ActT<true> { typedef f16 t; };
// dot product of two 8-element register fragments in f32
template <bool F16> __device__ __forceinline__ float frag_dot(const bf16x8& a, const bf16x8& b) {
  float d = 0.f;
  if constexpr (F16) {
    const f16x8 x = __builtin_bit_cast(f16x8, a), y = __builtin_bit_cast(f16x8, b);
#pragma unroll
    for (int j = 0; j < 8; ++j) d += (float)x[j] * (float)y[j];
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) d += (float)a[j] * (float)b[j];
  }
  return d;
}
__device__ __forceinline__ f32x16 splat16(float v) {
  f32x16 o;
#pragma unroll
  for (int t = 0; t < 16; ++t) o[t] = v;
  return o;
}
// 8 bf16 of a row held in global memory at column c (a register fragment); zero past the true head dim
template <int HD> __device__ __forceinline__ bf16x8 gfrag(const bf16* row, int c) {
  if (HD % 16 == 0 || c < HD) return *(const bf16x8*)(row + c);
  bf16x8 z;
#pragma unroll
  for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
  return z;
}
// A-operand fragment for a product that contracts over the tile's ROW index (transposed read):
// rows r0 + {4h.. , 8+4h..} in the acc-as-operand k order, 32 columns starting at c0 (lane r = column)
template <int HD> __device__ __forceinline__ bf16x8 frag_tr(const char* tile, int r0, int c0, int lane) {
  const int h = lane >> 5, g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int row = r0 + 4 * h + q, ch = (c0 >> 3) + 2 * (g & 1) + (p >> 1);
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + tile_off<HD>(row, ch) + ((p & 1) << 3)));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + tile_off<HD>(row + 8, ch) + ((p & 1) << 3)));
  union { bf16x8 v; s16x4 h2[2]; } u;
  u.h2[0] = lo; u.h2[1] = hi;
  return u.v;
}
// A-operand fragment by rows: row r0 + (lane&31), k chunk (2*ks + h)
template <int HD> __device__ __forceinline__ bf16x8 frag_row(const char* tile, int r0, int ks, int lane) {
  return *(const bf16x8*)(tile + tile_off<HD>(r0 + (lane & 31), 2 * ks + (lane >> 5)));
}
__device__ __forceinline__ bf16x8 acc_frag(const f32x16& x, int s) {
  bf16x8 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (bf16)x[8 * s + j];
  return f;
}
__device__ __forceinline__ bf16x8 acc_frag_h(const f32x16& x, int s) {       // p in [0, 1]: no saturation needed
  f16x8 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (f16)x[8 * s + j];
  return __builtin_bit_cast(bf16x8, f);
}
template <bool F16> __device__ __forceinline__ bf16x8 acc_frag_t(const f32x16& x, int s) {
  if constexpr (F16) return acc_frag_h(x, s); else return acc_frag(x, s);
}
__device__ __forceinline__ int acc_row(int t, int h) { return (t & 3) + 8 * (t >> 2) + 4 * h; }

// Sequence lengths that are not a multiple of 64 (VMAE: int(L * (1 - mask_ratio)) kept tokens for any ratio, models_mae.py:472-497): the
// LAST tile of a sweep is ragged.  Its missing rows are staged as copies of row N-1 (finite data, in bounds) and the score-chain
// accumulator rows that belong to them are set to -inf before the exponential: p = exp2(-inf) = 0, so they add nothing to the row sums,
// to O, or to any gradient.  `base` = first row of the 32-row block the accumulator covers; element t of half h is row acc_row(t, h).
__device__ __forceinline__ void mask_rows_past(f32x16& x, int base, int h, int N) {
#pragma unroll
  for (int t = 0; t < 16; ++t)
    if (base + (t & 3) + 8 * (t >> 2) + 4 * h >= N) x[t] = -__builtin_inff();
}

// K/V (Q/dO) tiles travel through a STAGES-deep LDS ring: global_load_lds stays in flight across the per-tile
// barrier (raw s_barrier + counted vmcnt), `ahead` = number of later stages allowed to be still in flight.
template <int PPW, int STAGES> __device__ __forceinline__ void ring_wait(int ahead) {
  if (ahead >= 2) { if constexpr (STAGES >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory"); }
  else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#define EXP2(x) __builtin_amdgcn_exp2f(x)
constexpr int ATT_STAGES = 3;
template <int I> struct IC { static constexpr int value = I; };
// f(IC<0>{}), f(IC<1>{}), ... f(IC<N-1>{}): a loop whose index is a compile-time constant in the body
template <int N, int I = 0, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(IC<I>{}); static_for<N, I + 1>(f); }
}
// running-max update threshold (log2 units): the O / l rescale is skipped while no query of the wave saw its
// maximum grow by more than this (P then stays <= 2^RESCALE_THR; exact softmax either way after the final 1/l)
constexpr float RESCALE_THR = 6.0f;

// ================================================================================================ forward, bf16
// Epilogue store of a wave's [32 rows][HD] result held as (result)^T accumulators (lane = row r + 32 h; element t of block d is
// column d*32 + 8*(t/4) + 4h + t%4).  Per-lane stores would put 8 B into 32 different rows per instruction (the store tail of a
// workgroup then costs ~9k cycles, MI355X_MICROARCH 'attention epilogue store tail'); instead the tile goes through a wave-private
// LDS image (144-B row pitch) and leaves as whole 128-B rows, 16 B per lane, 8 rows per instruction.  `mul` is per lane (= per row).
template <int HD, typename T = bf16>
__device__ __forceinline__ void store_rows_t(const f32x16 (&acc)[hd_pad(HD) / 32], float mul, char* lds_wave, bf16* gbase, long gstride, int lane,
                                             int nrows = 32) {      // nrows: rows of the wave's 32 that exist (ragged last block when N % 32 != 0)
  constexpr int PITCH = hd_pad(HD) * 2 + 16, CPR = HD / 8;    // bytes per LDS row; 16-B chunks per (true) row
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int d = 0; d < hd_pad(HD) / 32; ++d)
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      typename Pack<T>::v4 w;
#pragma unroll
      for (int j = 0; j < 4; ++j) w[j] = from_f<T>(acc[d][4 * t4 + j] * mul);
      *(typename Pack<T>::v4*)(lds_wave + r * PITCH + (d * 32 + 8 * t4 + 4 * h) * 2) = w;
    }
#pragma unroll
  for (int it = 0; it < (32 * CPR + 63) / 64; ++it) {
    const int idx = it * 64 + lane, row = idx / CPR, ch = idx % CPR;
    if (((32 * CPR) % 64 == 0 || idx < 32 * CPR) && row < nrows) {
      const bf16x8 v = *(const bf16x8*)(lds_wave + row * PITCH + ch * 16);
      *(bf16x8*)(gbase + (long)row * gstride + ch * 8) = v;
    }
  }
}

// ---- QK-RMSNorm + RoPE backward folded into the dQ / dK epilogues (LightningDiT block, head dims with 8 or 16 chunks per row).
// The unfused chain wrote dq / dk head-major, then elementwise.hip's qknorm_rope_bwd read them back with the pre-norm q / k to produce
// the packed dqkv (1.6 GB per layer at bs 256).  Here the dq (dk) tile goes through the same bf16 row image as store_rows_t and every
// (row, 16-B chunk) lane finishes the job: rope^T, RMSNorm backward against the pre-norm row it loads from the packed qkv, the result
// stored into the q (k) slot of dqkv as a whole 128-B line per row.  Same per-element formulas as qknorm_rope_bwd_kernel.
// Partial sums for the norm-weight and qkv-bias gradients are left per workgroup (batch, 128-row block, head): Pw [B*NB*H][2*hd] (q | k),
// Pb [B*NB][3*H*hd] (q | k | v) with NB = ceil(N / 128), summed over rows by the host (ldmae_colsum).
struct QkNormBwd {
  const bf16* qkv;
  const float* wq; const float* wk; const float* cos; const float* sin;
  bf16* dqkv;
  float* Pw; float* Pb;
  float eps;
};
// sum over the CPR (8 or 16) consecutive lanes that hold one row, by DPP moves (a few cycles each; __shfl_xor is a ds_bpermute round
// trip, and two of these reductions sit in the dependent chain of every row): quad_perm xor 1, xor 2, then row_half_mirror joins the
// two quads of an 8-lane group and row_mirror the two halves of a 16-lane row (the partial sums are uniform inside a group by then)
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CPR> __device__ __forceinline__ float row_sum(float v) {
  v = dpp_add<0xB1>(v);            // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);            // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);           // row_half_mirror
  if constexpr (CPR == 16) v = dpp_add<0x140>(v);      // row_mirror
  return v;
}
template <int CPR> __device__ __forceinline__ float col_sum(float v) {      // over the 64 / CPR lanes that hold one chunk column
#pragma unroll
  for (int o = CPR; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// everything the epilogue reads from global memory (pre-norm rows: HBM; cos / sin rows: L2-resident tables) is requested BEFORE the
// end-of-loop barrier, all iterations at once, so the latency hides under the barrier and the staging (the main loop's registers are dead)
template <int HD> struct QkPref { bf16x8 x[32 * (HD / 8) / 64]; float4 cs[32 * (HD / 8) / 64][2], sn[32 * (HD / 8) / 64][2]; };
template <int HD>
__device__ __forceinline__ void qk_prefetch_cs(QkPref<HD>& p, const QkNormBwd& a, int n0, int lane) {
  constexpr int CPR = HD / 8;
#pragma unroll
  for (int it = 0; it < 32 * CPR / 64; ++it) {
    const size_t to = (size_t)(n0 + (it * 64 + lane) / CPR) * HD + (lane % CPR) * 8;
    p.cs[it][0] = *(const float4*)(a.cos + to); p.cs[it][1] = *(const float4*)(a.cos + to + 4);
    p.sn[it][0] = *(const float4*)(a.sin + to); p.sn[it][1] = *(const float4*)(a.sin + to + 4);
  }
}
template <int HD>
__device__ __forceinline__ void qk_prefetch(QkPref<HD>& p, const QkNormBwd& a, int which, int b, int hh, int H, int N, int n0, int lane) {
  constexpr int CPR = HD / 8;
#pragma unroll
  for (int it = 0; it < 32 * CPR / 64; ++it) {
    const int row = (it * 64 + lane) / CPR, ch = lane % CPR;
    if (a.wq) p.x[it] = *(const bf16x8*)(a.qkv + (((size_t)b * N + n0 + row) * 3 + which) * H * HD + (size_t)hh * HD + ch * 8);
    else {                                   // RoPE only (use_qknorm=False): the pre-norm rows are not needed; zeros keep the row math finite
#pragma unroll
      for (int j = 0; j < 8; ++j) p.x[it][j] = (bf16)0.f;
    }
  }
}
// The row math of the fused QK-norm / RoPE backward: lane = (row, 16-B chunk) of a wave's 32 rows, `grad(it)` = the lane's 8 incoming
// gradients (already rounded to bf16) of iteration `it`.  aw / ab: this wave's column sums (norm-weight gradient, bias gradient) for
// columns ch*8 .. +7, valid on lanes < CPR
template <int HD, typename GRAD>
__device__ __forceinline__ void qknorm_rows_math(GRAD&& grad, int lane, const QkNormBwd& a, int which, int b, int hh, int H, int N, int n0,
                                                 const QkPref<HD>& pf, float (&aw)[8], float (&ab)[8]) {
  static_assert(HD == 64 || HD == 128, "fused QK-norm backward: 8 or 16 chunks per row");
  constexpr int CPR = HD / 8, NIT = 32 * CPR / 64;
  const int ch = lane % CPR;                      // 64 % CPR == 0: a lane keeps its chunk column in every iteration
  const bool norm = a.wq != nullptr;              // false: q_norm = k_norm = nn.Identity (lightningdit.py:60-61): w = 1, rstd = 1, n = 0 -> the stored row is rope^T(g)
  const float* wsrc = (which ? a.wk : a.wq) + ch * 8;
  float w8[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { w8[j] = norm ? wsrc[j] : 1.f; aw[j] = 0.f; ab[j] = 0.f; }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int row = (it * 64 + lane) / CPR, n = n0 + row;
    const bf16x8 gv = grad(it);
    const size_t so = (((size_t)b * N + n) * 3 + which) * H * HD + (size_t)hh * HD + ch * 8;
    const float c8[8] = {pf.cs[it][0].x, pf.cs[it][0].y, pf.cs[it][0].z, pf.cs[it][0].w, pf.cs[it][1].x, pf.cs[it][1].y, pf.cs[it][1].z, pf.cs[it][1].w};
    const float s8[8] = {pf.sn[it][0].x, pf.sn[it][0].y, pf.sn[it][0].z, pf.sn[it][0].w, pf.sn[it][1].x, pf.sn[it][1].y, pf.sn[it][1].z, pf.sn[it][1].w};
    float x[8], g[8], ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { x[j] = (float)pf.x[it][j]; g[j] = (float)gv[j]; ss += x[j] * x[j]; }
    const float rs = norm ? rsqrtf(row_sum<CPR>(ss) / (float)HD + a.eps) : 1.f;
    float nq[8], dn[8], dot = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {              // rope^T on the pair (j, j+1): elementwise.hip rope_apply_bwd
      const float t0 = g[j] * c8[j] + g[j + 1] * s8[j + 1], t1 = g[j + 1] * c8[j + 1] - g[j] * s8[j];
      nq[j] = x[j] * rs; nq[j + 1] = x[j + 1] * rs;
      aw[j] += t0 * nq[j]; aw[j + 1] += t1 * nq[j + 1];
      dn[j] = t0 * w8[j]; dn[j + 1] = t1 * w8[j + 1];
      dot += dn[j] * nq[j] + dn[j + 1] * nq[j + 1];
    }
    const float m = row_sum<CPR>(dot) / (float)HD;
    bf16x8 ov;
#pragma unroll
    for (int j = 0; j < 8; ++j) { ov[j] = (bf16)((dn[j] - nq[j] * m) * rs); ab[j] += (float)ov[j]; }
    *(bf16x8*)(a.dqkv + so) = ov;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { aw[j] = col_sum<CPR>(aw[j]); ab[j] = col_sum<CPR>(ab[j]); }
}
// the dq (dk) tile held as (result)^T accumulators: through the wave's bf16 row image (as store_rows_t), then the row math
template <int HD>
__device__ __forceinline__ void qknorm_rows_bwd(const f32x16 (&acc)[HD / 32], float mul, char* lds_wave, int lane, const QkNormBwd& a, int which,
                                                int b, int hh, int H, int N, int n0, const QkPref<HD>& pf, float (&aw)[8], float (&ab)[8]) {
  constexpr int PITCH = HD * 2 + 16, CPR = HD / 8;
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int d = 0; d < HD / 32; ++d)
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      bf16x4 w;
#pragma unroll
      for (int j = 0; j < 4; ++j) w[j] = (bf16)(acc[d][4 * t4 + j] * mul);
      *(bf16x4*)(lds_wave + r * PITCH + (d * 32 + 8 * t4 + 4 * h) * 2) = w;
    }
  qknorm_rows_math<HD>([&](int it) { return *(const bf16x8*)(lds_wave + ((it * 64 + lane) / CPR) * PITCH + (lane % CPR) * 16); },
                       lane, a, which, b, hh, H, N, n0, pf, aw, ab);
}
// store_rows_t that also returns the column sums of the rows AS STORED (the v part of the qkv bias gradient): ab, valid on lanes < CPR
template <int HD>
__device__ __forceinline__ void store_rows_t_colsum(const f32x16 (&acc)[HD / 32], float mul, char* lds_wave, bf16* gbase, long gstride, int lane,
                                                    float (&ab)[8]) {
  static_assert(HD == 64 || HD == 128, "8 or 16 chunks per row");
  constexpr int PITCH = HD * 2 + 16, CPR = HD / 8;
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int d = 0; d < HD / 32; ++d)
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      bf16x4 w;
#pragma unroll
      for (int j = 0; j < 4; ++j) w[j] = (bf16)(acc[d][4 * t4 + j] * mul);
      *(bf16x4*)(lds_wave + r * PITCH + (d * 32 + 8 * t4 + 4 * h) * 2) = w;
    }
  const int ch = lane % CPR;
#pragma unroll
  for (int j = 0; j < 8; ++j) ab[j] = 0.f;
#pragma unroll
  for (int it = 0; it < 32 * CPR / 64; ++it) {
    const int row = (it * 64 + lane) / CPR;
    const bf16x8 v = *(const bf16x8*)(lds_wave + row * PITCH + ch * 16);
    *(bf16x8*)(gbase + (long)row * gstride + ch * 8) = v;
#pragma unroll
    for (int j = 0; j < 8; ++j) ab[j] += (float)v[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) ab[j] = col_sum<CPR>(ab[j]);
}
// the four waves' column sums (NV vectors of HD floats each, lanes < CPR hold 8 columns) -> one row per workgroup, summed in wave order
template <int HD, int NV>
__device__ __forceinline__ void wg_colsums(const float (&v)[NV][8], float* red, int wave, int lane, bool active, int nact, float* const (&dst)[NV]) {
  constexpr int CPR = HD / 8;
  if (active && lane < CPR) {
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) red[(wave * NV + k) * HD + lane * 8 + j] = v[k][j];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < NV * HD; t += 256) {
    float s = red[t];
    for (int w = 1; w < nact; ++w) s += red[w * NV * HD + t];
    dst[t / HD][t % HD] = s;
  }
}

// where q / k / v (and dq / dk / dv) live: element offsets of (batch b, head h, row n) = b * sb + h * sh + n * ld.
// Head-major [B,H,N,hd]: sb = H*N*hd, sh = N*hd, ld = hd.  Packed token-major qkv [B,N,3,H,hd] (what the qkv Linear writes, used as is
// by the VMAE blocks, which have no QK-norm / RoPE between the Linear and the attention): sb = N*3*H*hd, sh = hd, ld = 3*H*hd.
struct QkvLayout { long sb, sh, ld; };

// VALU diet (the loop is bound by vector ISSUE, not by the matrix pipe: ~260 VALU instructions per 16 MFMAs before, profiles/r02):
// (i) q carries scale*log2(e) (rounded once more to bf16 in registers) and every score chain starts from the accumulator block
// nm = -(running max): the MFMA result is already the exponent, p = exp2(s) is ONE instruction per score; (ii) the running max only
// moves in the (rare, wave-uniform) rescale branch, which shifts the tile's scores and refreshes nm; (iii) K / V tile addresses are scalar.
// F16: the operands are fp16 (same bytes per element; v_mfma_f32_32x32x16_f16): the TF32-class forward path of the VMAE docking calls
template <int HD, bool RAGGED = false, bool F16 = false>      // RAGGED: N % 64 != 0 (its own instantiation: the masking costs the hot shapes no registers)
__global__ __launch_bounds__(256) void attn_fwd_bf16_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K, const bf16* __restrict__ V,
                                                            bf16* __restrict__ O, float* __restrict__ LSE, int H, int N, float c, QkvLayout L, QkvLayout Lv,
                                                            const float* __restrict__ SB, int sb_heads) {
  constexpr int HDP = hd_pad(HD), KS = (HD + 15) / 16, DB = HDP / 32, TB = 64 * HDP * 2;
  constexpr bool BATCH = HDP <= 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [STAGES][K tile | V tile]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, h = lane >> 5;
  const int qblocks = (N + 127) / 128;
  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = lid / qblocks, q0 = (lid % qblocks) * 128 + wave * 32;
  const bool active = q0 < N;
  const size_t hb = (size_t)(bh / H) * L.sb + (size_t)(bh % H) * L.sh;
  const bf16* qp = Q + hb;
  const bf16* kp = K + hb;
  const bf16* vp = V + (size_t)(bh / H) * Lv.sb + (size_t)(bh % H) * Lv.sh;
  const long ld = L.ld, ldv = Lv.ld;
  bf16x8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const bf16x8 qraw = gfrag<HD>(qp + (size_t)min(q0 + r, N - 1) * ld, ks * 16 + 8 * h);
    qf[ks] = F16 ? frag_scale_h(qraw, c) : frag_scale(qraw, c);
  }
  f32x16 oacc[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d) oacc[d] = splat16(0.f);
  f32x16 nm = splat16(0.f);                // -ms in every element: the C operand of the first MFMA of a score chain
  float ms = 0.f, l = 0.f;                 // ms is set from the first tile (rescale branch), so 0 is never used as a maximum
  const int nt = (N + 63) / 64;
  constexpr bool ragged = RAGGED;          // the last key tile has N % 64 real rows (see mask_rows_past)
  constexpr int PPW = 2 * (TB / 1024) / 4;
  TileMap<HD, 64> mk, mv;
  mk.init(ld, wave, lane);
  mv.init(ldv, wave, lane);
  // padded head dims: the row sums come out of the P.V product (ones column in the V image, see g_one16) instead of 32 vector adds per tile
  constexpr bool LSUM = HD != HDP;
  constexpr int LROW = HD % 32, LREG = (LROW & 3) + 4 * (LROW >> 3), LHALF = (LROW >> 2) & 1;
  if constexpr (HD != HDP) {
#pragma unroll
    for (int st = 0; st < ATT_STAGES; ++st) {
      mk.prefill(smem + st * 2 * TB, wave, lane);
      mv.template prefill<true, F16>(smem + st * 2 * TB + TB, wave, lane);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): in LDS before this wave reaches the first tile barrier
  }
  // A STATIC shift instead of the running maximum, when the caller can bound the scores (SB: one float on the device, an upper bound of
  // |score * scale * log2(e)| over all queries and keys -- for the QK-normalised DiT heads it follows from the norm weights alone:
  // ldmae_qk_score_bound).  With the shift bq every exponent s - bq lies in [-2 bq, 0]: no p overflows, and while 2 bq <= 100 none flushes
  // to zero, so the softmax is exact without the per-tile 31-deep max chain, cross-half shuffle, ballot and rescale branch (a fifth of the
  // loop's vector instructions: -8 % at head dim 64, -12 % at 16, profiles/r04_attn_static_shift.txt).  No bound, or one above 50: the
  // tracked form below, as before.
  // sb_heads: SB is [B*H][2] = (max_i |q_i|^2, max_j |k_j|^2) of each (batch, head) as STORED (unscaled): |q_i . k_j| <= |q_i| |k_j|
  // (the fused VMAE q | k | v kernel leaves these maxima behind for free: csrc/vmae_fused.hip).
  // A first value of 0 means "no maximum over the queries": every lane then uses the norm of its own (scaled) query (ldmae_k_norm_max).
  float bq = 0.f;
  if (SB && !F16) {                         // (the fp16 instantiation keeps the tracked form: its callers pass no bound)
    if (!sb_heads) bq = *SB;
    else if (SB[2 * bh] > 0.f) bq = sqrtf(SB[2 * bh] * SB[2 * bh + 1]) * c * 1.02f;
    else {
      float qn = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) qn += (float)qf[ks][j] * (float)qf[ks][j];
      qn += __shfl_xor(qn, 32, 64);
      bq = sqrtf(qn * SB[2 * bh + 1]) * 1.02f + 0.01f;
    }
  }
  const bool stat = SB != nullptr && !F16 && __builtin_amdgcn_ballot_w64(!(bq <= 50.f)) == 0;       // wave-uniform
  const unsigned lds0 = lds_addr_of(smem);
  auto stage = [&](int kt) {
    const int so = (kt % ATT_STAGES) * 2 * TB;
    if (ragged && kt == nt - 1) {          // same number of LDS-DMA instructions per wave as the hoisted form: the vmcnt counts hold
      stage_tile<HD, 64>(kp + (size_t)kt * 64 * ld, ld, N - 1 - kt * 64, smem + so, wave, lane);
      stage_tile<HD, 64, LSUM, F16>(vp + (size_t)kt * 64 * ldv, ldv, N - 1 - kt * 64, smem + so + TB, wave, lane);
      return;
    }
    mk.issue(kp + (size_t)kt * 64 * ld, ld, smem + so, lds0 + so, wave, lane);
    mv.issue(vp + (size_t)kt * 64 * ldv, ldv, smem + so + TB, lds0 + so + TB, wave, lane);
  };
#pragma unroll
  for (int st = 0; st < ATT_STAGES - 1; ++st)
    if (st < nt) stage(st);
  // vmcnt(0) through the BUILTIN (visible to the compiler's waitcnt pass): the register fragments above are then known to have
  // landed, so no compiler-counted wait for them ends up inside the tile loop, where -- the ring DMA being hidden in asm -- it
  // would drain the ring at every tile.  The prologue stages are in flight beside those loads, so this costs one round trip.
  __builtin_amdgcn_s_waitcnt(0x0F70);
  auto body = [&](auto ST, auto TRK, int kt) {
    constexpr int st = decltype(ST)::value;
    constexpr bool track = decltype(TRK)::value != 0;
    ring_wait<PPW, ATT_STAGES>(min(ATT_STAGES - 2, nt - 1 - kt));
    __builtin_amdgcn_s_barrier();
    if (kt + ATT_STAGES - 1 < nt) stage(kt + ATT_STAGES - 1);
    const char* Kt = smem + st * 2 * TB;
    const char* Vt = Kt + TB;
    // LDS reads are batched ahead of the MFMAs that consume them (left to itself the compiler keeps ONE fragment in flight and waits
    // for it before every MFMA: 16 exposed LDS round trips per tile); the two 32-key chains are interleaved
    // (head dims up to 64; wider ones would not fit the register file that way and read each fragment where it is used)
    f32x16 s[2];
    bf16x8 kfr[KS][2];
    if constexpr (BATCH) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) kfr[ks][kb] = frag_row<HDP>(Kt, kb * 32, ks, lane);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        s[kb] = mfma_att<F16>(BATCH ? kfr[ks][kb] : frag_row<HDP>(Kt, kb * 32, ks, lane), qf[ks], ks == 0 ? nm : s[kb]);
    bf16x8 vfr[2][2][DB];                              // V^T fragments: in flight under the softmax arithmetic
    if constexpr (BATCH) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int d = 0; d < DB; ++d) vfr[kb][s2][d] = frag_tr<HDP>(Vt, kb * 32 + 16 * s2, d * 32, lane);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ragged && kt == nt - 1) { mask_rows_past(s[0], kt * 64, h, N); mask_rows_past(s[1], kt * 64 + 32, h, N); }
    // s = score * scale * log2(e) - ms
    if constexpr (track) {
      float mx = fmaxf(s[0][0], s[1][0]);
  #pragma unroll
      for (int t = 1; t < 16; ++t) mx = fmaxf(mx, fmaxf(s[0][t], s[1][t]));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const bool first = kt == 0;
      if (first || __builtin_amdgcn_ballot_w64(mx > RESCALE_THR) != 0) {      // wave-uniform: first tile, or some query's max grew a lot
        const float d = first ? mx : fmaxf(mx, 0.f);
        ms += d;
        if (!first) {
          const float alpha = EXP2(-d);
          l *= alpha;
  #pragma unroll
          for (int dd = 0; dd < DB; ++dd)
  #pragma unroll
            for (int t = 0; t < 16; ++t) oacc[dd][t] *= alpha;
        }
  #pragma unroll
        for (int t = 0; t < 16; ++t) { s[0][t] -= d; s[1][t] -= d; }
        nm = splat16(-ms);
      }
    }
    float rs = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int t = 0; t < 16; ++t) { const float p = EXP2(s[kb][t]); s[kb][t] = p; if constexpr (!LSUM) rs += p; }
    l += rs;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pf = F16 ? acc_frag_h(s[kb], s2) : acc_frag(s[kb], s2);
#pragma unroll
        for (int d = 0; d < DB; ++d)
          oacc[d] = mfma_att<F16>(BATCH ? vfr[kb][s2][d] : frag_tr<HDP>(Vt, kb * 32 + 16 * s2, d * 32, lane), pf, oacc[d]);
      }
  };
  if (stat) {                               // the shift is the bound: no maximum to track
    ms = bq;
    nm = splat16(-bq);
    for (int kt = 0; kt < nt; kt += ATT_STAGES) {
      body(IC<0>{}, IC<0>{}, kt);
      if (kt + 1 < nt) body(IC<1>{}, IC<0>{}, kt + 1);
      if (kt + 2 < nt) body(IC<2>{}, IC<0>{}, kt + 2);
    }
  } else {
    for (int kt = 0; kt < nt; kt += ATT_STAGES) {
      body(IC<0>{}, IC<1>{}, kt);
      if (kt + 1 < nt) body(IC<1>{}, IC<1>{}, kt + 1);
      if (kt + 2 < nt) body(IC<2>{}, IC<1>{}, kt + 2);
    }
  }
  if constexpr (LSUM) l = __shfl(oacc[HD / 32][LREG], (lane & 31) + 32 * LHALF, 64);      // row HD of O^T = sum over the keys of P (as rounded to bf16)
  else l += __shfl_xor(l, 32, 64);
  __syncthreads();                                   // every wave is past its last K/V read: the ring becomes store scratch
  if (!active) return;
  const float inv = 1.f / l;
  const int b = bh / H, hh = bh % H;
  if constexpr (F16) store_rows_t<HD, f16>(oacc, inv, smem + wave * 32 * (HDP * 2 + 16), O + ((size_t)(b * N + q0) * H + hh) * HD, (long)H * HD, lane, N - q0);
  else store_rows_t<HD>(oacc, inv, smem + wave * 32 * (HDP * 2 + 16), O + ((size_t)(b * N + q0) * H + hh) * HD, (long)H * HD, lane, N - q0);
  if (h == 0 && q0 + r < N) LSE[(size_t)bh * N + q0 + r] = (ms + log2f(l)) * 0.6931471805599453f;
}

// delta[b,h,n] = sum_d o[b,n,h,d] * do[b,n,h,d]
template <typename T>
__global__ void attn_delta_kernel(const T* __restrict__ O, const T* __restrict__ dO, float* __restrict__ delta, int B, int H, int N, int hd) {
  const long items = (long)B * N * H;
  const int lpr = 8;                     // lanes per (b,n,h) row
  const long gid = ((long)blockIdx.x * 256 + threadIdx.x) / lpr;
  const int sub = threadIdx.x % lpr;
  for (long it = gid; it < items; it += (long)gridDim.x * 256 / lpr) {
    const int hh = it % H, n = (it / H) % N, b = it / ((long)H * N);
    const T* o = O + (size_t)it * hd;
    const T* g = dO + (size_t)it * hd;
    float s = 0.f;
    for (int d = sub * 8; d < hd; d += lpr * 8) {
      float a[8], e[8];
      Vec8<T>::load(o + d, a); Vec8<T>::load(g + d, e);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += a[j] * e[j];
    }
    s = group_sum<8>(s);
    if (sub == 0) delta[((size_t)b * H + hh) * N + n] = s;
  }
}

// ================================================================================================ backward dK/dV, bf16
// ROWC = [2][B*H*N] f32 written by the dQ kernel (which runs first): slot 0 = -delta, slot 1 = -lse * log2(e).  They are the INITIAL
// ACCUMULATORS of the dP and S chains (a query row = an accumulator element here, so they come from the LDS copy of the tile's 64 values
// by ds_read_b128), k carries scale*log2(e): p = exp2(S) and dS = p * dP are one instruction per element each.
// PF (round 6): LDS operand fragments are requested PF MFMAs ahead of the one that consumes them (a ring of PF + 1 fragments in registers, the
// reads pinned in place against the scheduler), instead of right in front of it behind an `s_waitcnt lgkmcnt(0)` each
// PIPE (round 6, diagnostic A/B only; needs PF > 0): the two 32-row halves of a tile run software-pipelined -- order [S / dP of half 0] [S / dP of half 1]
// [dV / dK of half 0] [dV / dK of half 1], so the exponentials / conversions of one half have the other half's MFMAs to run beside (verdict r05 item 3b); both
// halves' score blocks are live at once: +32 VGPRs, 2 waves per SIMD
template <int HD, bool QKN = false, bool RAGGED = false, bool F16 = false, int PF = 0, int PIPE = 0>      // F16: fp16 operands / outputs (VMAE pre-training under fp16 autocast), QKN = false only
__global__ __launch_bounds__(256) void attn_bwd_dkdv_bf16_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K, const bf16* __restrict__ V,
                                                                 const bf16* __restrict__ dO, const float* __restrict__ ROWC, long rc_stride,
                                                                 bf16* __restrict__ dK, bf16* __restrict__ dV,
                                                                 int H, int N, float scale, QkvLayout L, QkvLayout Lv, QkNormBwd qn) {
  constexpr int HDP = hd_pad(HD), KS = (HD + 15) / 16, DB = HDP / 32, TB = 64 * HDP * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [STAGES][Q tile | dO tile | -lse2[64] -delta[64] scratch[128]]
  constexpr int BUF = 2 * TB + 1024;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, h = lane >> 5;
  const float c = scale * 1.4426950408889634f;
  const int kblocks = (N + 127) / 128;
  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = lid / kblocks, k0 = (lid % kblocks) * 128 + wave * 32;
  const bool active = k0 < N;
  const int b = bh / H, hh = bh % H;
  const size_t hb = (size_t)b * L.sb + (size_t)hh * L.sh, hbv = (size_t)b * Lv.sb + (size_t)hh * Lv.sh;
  const long ld = L.ld;
  const bf16* qp = Q + hb;
  const bf16* dop = dO + ((size_t)b * N * H + hh) * HD;     // row stride H*HD
  const long dold = (long)H * HD;
  bf16x8 kf[KS], vf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    kf[ks] = frag_scale_t<F16>(gfrag<HD>(K + hb + (size_t)min(k0 + r, N - 1) * ld, ks * 16 + 8 * h), c);
    vf[ks] = gfrag<HD>(V + hbv + (size_t)min(k0 + r, N - 1) * Lv.ld, ks * 16 + 8 * h);
  }
  f32x16 dkacc[DB], dvacc[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d) { dkacc[d] = splat16(0.f); dvacc[d] = splat16(0.f); }
  const int nt = (N + 63) / 64, NP = nt * 64;      // ROWC rows are padded to whole tiles: the dQ kernel fills the pad with (-inf, 0)
  constexpr bool ragged = RAGGED;
  constexpr int PPW = 2 * (TB / 1024) / 4 + 1;
  const float* rowc = ROWC + (size_t)bh * NP + ((wave & 1) ? 0 : rc_stride);     // wave 0/2: -lse2, wave 1/3: -delta
  TileMap<HD, 64> mq, mo;
  mq.init(ld, wave, lane);
  mo.init(dold, wave, lane);
  if constexpr (HD != HDP) {                   // zero padding of every Q / dO image, once (TileMap)
#pragma unroll
    for (int st = 0; st < ATT_STAGES; ++st) { mq.prefill(smem + st * BUF, wave, lane); mo.prefill(smem + st * BUF + TB, wave, lane); }
    __builtin_amdgcn_s_waitcnt(0xC07F);
  }
  const unsigned lds0 = lds_addr_of(smem);
  auto stage = [&](int qt) {
    const int so = (qt % ATT_STAGES) * BUF;
    if (ragged && qt == nt - 1) {          // missing query rows: copies of row N-1; their -lse2 = -inf in ROWC zeroes p (and with it dS)
      stage_tile<HD, 64>(qp + (size_t)qt * 64 * ld, ld, N - 1 - qt * 64, smem + so, wave, lane);
      stage_tile<HD, 64>(dop + (size_t)qt * 64 * dold, dold, N - 1 - qt * 64, smem + so + TB, wave, lane);
    } else {
    mq.issue(qp + (size_t)qt * 64 * ld, ld, smem + so, lds0 + so, wave, lane);
    mo.issue(dop + (size_t)qt * 64 * dold, dold, smem + so + TB, lds0 + so + TB, wave, lane);
    }
    // 64 floats of -lse2 (slot 0) / -delta (slot 1); waves 2,3 fill scratch slots so every wave issues PPW loads
    glds4_s(rowc + qt * 64, lane * 4, lds0 + so + 2 * TB + wave * 256);
  };
#pragma unroll
  for (int st = 0; st < ATT_STAGES - 1; ++st)
    if (st < nt) stage(st);
  // vmcnt(0) through the BUILTIN (visible to the compiler's waitcnt pass): the register fragments above are then known to have
  // landed, so no compiler-counted wait for them ends up inside the tile loop, where -- the ring DMA being hidden in asm -- it
  // would drain the ring at every tile.  The prologue stages are in flight beside those loads, so this costs one round trip.
  __builtin_amdgcn_s_waitcnt(0x0F70);
  auto body = [&](auto ST, int qt) {
    constexpr int st = decltype(ST)::value;
    ring_wait<PPW, ATT_STAGES>(min(ATT_STAGES - 2, nt - 1 - qt));
    __builtin_amdgcn_s_barrier();
    if (qt + ATT_STAGES - 1 < nt) stage(qt + ATT_STAGES - 1);
    const char* Qt = smem + st * BUF;
    const char* dOt = Qt + TB;
    const float* nl = (const float*)(Qt + 2 * TB);
    const float* nd = nl + 64;
    if constexpr (PF > 0) {
      // the 2 x (2 KS + 4 DB) MFMAs of the tile in their order, each with ONE operand fragment from LDS: fragment i + PF is requested before MFMA i
      constexpr int N1 = 2 * KS, N2 = 4 * DB, NQ = N1 + N2, NF = 2 * NQ;
      // sequence position i -> (half qb, step j of that half's N1 + N2 MFMAs)
      constexpr auto half_of = [](int i) { return PIPE ? (i < 2 * N1 ? i / N1 : (i - 2 * N1) / N2) : i / NQ; };
      constexpr auto step_of = [](int i) { return PIPE ? (i < 2 * N1 ? i % N1 : N1 + (i - 2 * N1) % N2) : i % NQ; };
      auto frag = [&](auto I) -> bf16x8 {
        constexpr int i = decltype(I)::value, qb = half_of(i), j = step_of(i);
        if constexpr (j < N1) {
          if constexpr ((j & 1) != 0) return frag_row<HDP>(dOt, qb * 32, j / 2, lane); else return frag_row<HDP>(Qt, qb * 32, j / 2, lane);
        } else {
          constexpr int t = j - N1, s2 = t / (2 * DB), d = (t % (2 * DB)) / 2;
          if constexpr ((t & 1) != 0) return frag_tr<HDP>(Qt, qb * 32 + 16 * s2, d * 32, lane); else return frag_tr<HDP>(dOt, qb * 32 + 16 * s2, d * 32, lane);
        }
      };
      bf16x8 win[PF + 1];
      static_for<PF>([&](auto I) { win[decltype(I)::value] = frag(I); });
      f32x16 sx[PIPE ? 2 : 1], dpx[PIPE ? 2 : 1];
      bf16x8 pf, dsf;
      static_for<NF>([&](auto I) {
        constexpr int i = decltype(I)::value, qb = half_of(i), j = step_of(i);
        f32x16& s = sx[PIPE ? qb : 0];
        f32x16& dp = dpx[PIPE ? qb : 0];
        if constexpr (i + PF < NF) win[(i + PF) % (PF + 1)] = frag(IC<i + PF>{});
        if constexpr (j == 0) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 a = *(const f32x4*)(nl + qb * 32 + 8 * g + 4 * h), e = *(const f32x4*)(nd + qb * 32 + 8 * g + 4 * h);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) { s[4 * g + jj] = a[jj]; dp[4 * g + jj] = e[jj]; }
          }
        }
        __builtin_amdgcn_sched_barrier(0x6);          // vector / scalar ALU may move across; LDS reads and MFMAs keep this order
        // the exponentials of a half: written where its score chains end (PIPE: in front of the OTHER half's next section, beside whose MFMAs they can run)
        if constexpr (PIPE ? (i == N1 || i == 2 * N1) : j == N1) {
          constexpr int qe = PIPE ? (i == N1 ? 0 : 1) : 0;
          f32x16& se = sx[PIPE ? qe : 0];
          f32x16& de = dpx[PIPE ? qe : 0];
#pragma unroll
          for (int t = 0; t < 16; ++t) { const float p = EXP2(se[t]); se[t] = p; de[t] *= p; }
        }
        if constexpr (j >= N1 && (j - N1) % (2 * DB) == 0) { pf = acc_frag_t<F16>(s, (j - N1) / (2 * DB)); dsf = acc_frag_t<F16>(dp, (j - N1) / (2 * DB)); }
        const bf16x8 a = win[i % (PF + 1)];
        if constexpr (j < N1) {
          if constexpr ((j & 1) != 0) dp = mfma_att<F16>(a, vf[j / 2], dp); else s = mfma_att<F16>(a, kf[j / 2], s);
        } else {
          constexpr int d = ((j - N1) % (2 * DB)) / 2;
          if constexpr (((j - N1) & 1) != 0) dkacc[d] = mfma_att<F16>(a, dsf, dkacc[d]); else dvacc[d] = mfma_att<F16>(a, pf, dvacc[d]);
        }
      });
    } else
    // (fragments are read where they are used: batching them ALL ahead, as the forward kernel does, costs this kernel its third wave per SIMD
    // and measured slower: 1.64 vs 1.56 ms)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x16 s, dp;
#pragma unroll
      for (int g = 0; g < 4; ++g) {              // accumulator element 4g+j <-> query row qb*32 + 8g + 4h + j
        const f32x4 a = *(const f32x4*)(nl + qb * 32 + 8 * g + 4 * h), e = *(const f32x4*)(nd + qb * 32 + 8 * g + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) { s[4 * g + j] = a[j]; dp[4 * g + j] = e[j]; }
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        s = mfma_att<F16>(frag_row<HDP>(Qt, qb * 32, ks, lane), kf[ks], s);
        dp = mfma_att<F16>(frag_row<HDP>(dOt, qb * 32, ks, lane), vf[ks], dp);
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const float p = EXP2(s[t]);
        s[t] = p;
        dp[t] *= p;
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pf = acc_frag_t<F16>(s, s2), dsf = acc_frag_t<F16>(dp, s2);
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          dvacc[d] = mfma_att<F16>(frag_tr<HDP>(dOt, qb * 32 + 16 * s2, d * 32, lane), pf, dvacc[d]);
          dkacc[d] = mfma_att<F16>(frag_tr<HDP>(Qt, qb * 32 + 16 * s2, d * 32, lane), dsf, dkacc[d]);
        }
      }
    }
  };
  for (int qt = 0; qt < nt; qt += ATT_STAGES) {
    body(IC<0>{}, qt);
    if (qt + 1 < nt) body(IC<1>{}, qt + 1);
    if (qt + 2 < nt) body(IC<2>{}, qt + 2);
  }
  if constexpr (QKN) {       // dk -> k slot of dqkv through the QK-norm / RoPE backward; dv as before, plus its column sums for the qkv bias gradient
    QkPref<HD> pf;
    if (active) qk_prefetch<HD>(pf, qn, 1, b, hh, H, N, k0, lane);
    __syncthreads();                                 // ring -> store scratch
    char* sw = smem + wave * 32 * (HDP * 2 + 16);
    float cs3[3][8];
    if (active) {
      qk_prefetch_cs<HD>(pf, qn, k0, lane);          // (two accumulator sets are still live before the barrier: no room there)
      qknorm_rows_bwd<HD>(dkacc, scale, sw, lane, qn, 1, b, hh, H, N, k0, pf, cs3[0], cs3[1]);
      store_rows_t_colsum<HD>(dvacc, 1.f, sw, dV + hbv + (size_t)k0 * Lv.ld, Lv.ld, lane, cs3[2]);
    }
    const int kb = lid % kblocks;
    const size_t blk = (size_t)b * kblocks + kb;
    float* const dst[3] = {qn.Pw + (blk * H + hh) * (2 * HD) + HD, qn.Pb + blk * (3 * H * HD) + ((size_t)H + hh) * HD,
                           qn.Pb + blk * (3 * H * HD) + ((size_t)2 * H + hh) * HD};
    wg_colsums<HD, 3>(cs3, (float*)(smem + 4 * 32 * (HDP * 2 + 16)), wave, lane, active, min(4, (N - kb * 128) / 32), dst);
  } else {
    __syncthreads();                                   // ring -> store scratch
    if (!active) return;
    char* sw = smem + wave * 32 * (HDP * 2 + 16);
    store_rows_t<HD, typename ActT<F16>::t>(dkacc, scale, sw, dK + hb + (size_t)k0 * ld, ld, lane, N - k0);
    store_rows_t<HD, typename ActT<F16>::t>(dvacc, 1.f, sw, dV + hbv + (size_t)k0 * Lv.ld, Lv.ld, lane, N - k0);       // same wave, same scratch: LDS ops stay in order
  }
}

// ================================================================================================ backward dQ, bf16
template <int HD, bool QKN = false, bool RAGGED = false, bool F16 = false, int PF = 0>      // F16: fp16 operands / outputs (VMAE pre-training under fp16 autocast), QKN = false only; PF: see the dK/dV kernel
__global__ __launch_bounds__(256) void attn_bwd_dq_bf16_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K, const bf16* __restrict__ V,
                                                               const bf16* __restrict__ O, const bf16* __restrict__ dO,
                                                               const float* __restrict__ LSE, float* __restrict__ ROWC, long rc_stride,
                                                               bf16* __restrict__ dQ, int H, int N, float scale, QkvLayout L, QkvLayout Lv, QkNormBwd qn) {
  constexpr int HDP = hd_pad(HD), KS = (HD + 15) / 16, DB = HDP / 32, TB = 64 * HDP * 2;
  constexpr bool BTR = HDP <= 64;     // transposed K fragments read ahead, under the exp / multiply arithmetic (fits 3 waves per SIMD)
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [STAGES][K tile | V tile]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, h = lane >> 5;
  const float c = scale * 1.4426950408889634f;
  const int qblocks = (N + 127) / 128;
  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = lid / qblocks, q0 = (lid % qblocks) * 128 + wave * 32;
  const bool active = q0 < N;
  const int b = bh / H, hh = bh % H;
  const int qrow = min(q0 + r, N - 1);
  const size_t hb = (size_t)b * L.sb + (size_t)hh * L.sh;
  const long ld = L.ld, ldv = Lv.ld;
  const bf16* kp = K + hb;
  const bf16* vp = V + (size_t)b * Lv.sb + (size_t)hh * Lv.sh;
  bf16x8 qf[KS], dof[KS];
  // delta_i = sum_d dO[i,d] * O[i,d] is formed here from the dO fragments this lane holds anyway (its half of the row; the other
  // half sits on lane ^ 32) and published (negated, with -lse*log2(e) beside it) for the dK/dV kernel, which runs after this one:
  // no separate pass over O and dO.  Here the query sits on the lane, so both are lane constants: splat into the accumulator blocks
  // the S and dP chains start from; q carries scale*log2(e).
  float dpart = 0.f;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    qf[ks] = frag_scale_t<F16>(gfrag<HD>(Q + hb + (size_t)qrow * ld, ks * 16 + 8 * h), c);
    const size_t oo = (((size_t)b * N + qrow) * H + hh) * HD;
    dof[ks] = gfrag<HD>(dO + oo, ks * 16 + 8 * h);
    const bf16x8 of = gfrag<HD>(O + oo, ks * 16 + 8 * h);
    dpart += frag_dot<F16>(dof[ks], of);
  }
  const float dl = dpart + __shfl_xor(dpart, 32);
  const float lse2 = LSE[(size_t)bh * N + qrow] * 1.4426950408889634f;
  const int NP = (N + 63) / 64 * 64;        // ROWC rows padded to whole tiles; pad rows (-delta, -lse2) = (0, -inf): p = dS = 0 in the dK/dV kernel
  if (h == 0 && q0 + r < NP) {
    const bool real = q0 + r < N;
    ROWC[(size_t)bh * NP + q0 + r] = real ? -dl : 0.f;
    ROWC[rc_stride + (size_t)bh * NP + q0 + r] = real ? -lse2 : -__builtin_inff();
  }
  const f32x16 nl = splat16(-lse2), nd = splat16(-dl);
  f32x16 dqacc[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d) dqacc[d] = splat16(0.f);
  const int nt = (N + 63) / 64;
  constexpr bool ragged = RAGGED;
  constexpr int PPW = 2 * (TB / 1024) / 4;
  TileMap<HD, 64> mk, mv;
  mk.init(ld, wave, lane);
  mv.init(ldv, wave, lane);
  if constexpr (HD != HDP) {                   // zero padding of every K / V image, once (TileMap)
#pragma unroll
    for (int st = 0; st < ATT_STAGES; ++st) { mk.prefill(smem + st * 2 * TB, wave, lane); mv.prefill(smem + st * 2 * TB + TB, wave, lane); }
    __builtin_amdgcn_s_waitcnt(0xC07F);
  }
  const unsigned lds0 = lds_addr_of(smem);
  auto stage = [&](int kt) {
    const int so = (kt % ATT_STAGES) * 2 * TB;
    if (ragged && kt == nt - 1) {
      stage_tile<HD, 64>(kp + (size_t)kt * 64 * ld, ld, N - 1 - kt * 64, smem + so, wave, lane);
      stage_tile<HD, 64>(vp + (size_t)kt * 64 * ldv, ldv, N - 1 - kt * 64, smem + so + TB, wave, lane);
      return;
    }
    mk.issue(kp + (size_t)kt * 64 * ld, ld, smem + so, lds0 + so, wave, lane);
    mv.issue(vp + (size_t)kt * 64 * ldv, ldv, smem + so + TB, lds0 + so + TB, wave, lane);
  };
#pragma unroll
  for (int st = 0; st < ATT_STAGES - 1; ++st)
    if (st < nt) stage(st);
  // vmcnt(0) through the BUILTIN (visible to the compiler's waitcnt pass): the register fragments above are then known to have
  // landed, so no compiler-counted wait for them ends up inside the tile loop, where -- the ring DMA being hidden in asm -- it
  // would drain the ring at every tile.  The prologue stages are in flight beside those loads, so this costs one round trip.
  __builtin_amdgcn_s_waitcnt(0x0F70);
  auto body = [&](auto ST, int kt) {
    constexpr int st = decltype(ST)::value;
    ring_wait<PPW, ATT_STAGES>(min(ATT_STAGES - 2, nt - 1 - kt));
    __builtin_amdgcn_s_barrier();
    if (kt + ATT_STAGES - 1 < nt) stage(kt + ATT_STAGES - 1);
    const char* Kt = smem + st * 2 * TB;
    const char* Vt = Kt + TB;
    if constexpr (PF > 0) {
      // the 2 x (2 KS + 2 DB) MFMAs of the tile in their order, each with ONE operand fragment from LDS (K / V rows for the score chains, transposed
      // K for dQ): fragment i + PF is requested before MFMA i; replaces the batch of transposed reads (BTR) of the PF = 0 form
      constexpr int N1 = 2 * KS, N2 = 2 * DB, NQ = N1 + N2, NF = 2 * NQ;
      auto frag = [&](auto I) -> bf16x8 {
        constexpr int i = decltype(I)::value, kb = i / NQ, j = i % NQ;
        if constexpr (j < N1) {
          if constexpr ((j & 1) != 0) return frag_row<HDP>(Vt, kb * 32, j / 2, lane); else return frag_row<HDP>(Kt, kb * 32, j / 2, lane);
        } else return frag_tr<HDP>(Kt, kb * 32 + 16 * ((j - N1) / DB), ((j - N1) % DB) * 32, lane);
      };
      bf16x8 win[PF + 1];
      static_for<PF>([&](auto I) { win[decltype(I)::value] = frag(I); });
      f32x16 s, dp;
      bf16x8 dsf;
      static_for<NF>([&](auto I) {
        constexpr int i = decltype(I)::value, kb = i / NQ, j = i % NQ;
        if constexpr (i + PF < NF) win[(i + PF) % (PF + 1)] = frag(IC<i + PF>{});
        __builtin_amdgcn_sched_barrier(0x6);          // vector / scalar ALU may move across; LDS reads and MFMAs keep this order
        if constexpr (j == N1) {
          if (ragged && kt == nt - 1) mask_rows_past(s, kt * 64 + kb * 32, h, N);
#pragma unroll
          for (int t = 0; t < 16; ++t) dp[t] *= EXP2(s[t]);
        }
        if constexpr (j >= N1 && (j - N1) % DB == 0) dsf = acc_frag_t<F16>(dp, (j - N1) / DB);
        const bf16x8 a = win[i % (PF + 1)];
        if constexpr (j < N1) {
          if constexpr ((j & 1) != 0) dp = mfma_att<F16>(a, dof[j / 2], j == 1 ? nd : dp); else s = mfma_att<F16>(a, qf[j / 2], j == 0 ? nl : s);
        } else dqacc[(j - N1) % DB] = mfma_att<F16>(a, dsf, dqacc[(j - N1) % DB]);
      });
    } else
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 s, dp;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        s = mfma_att<F16>(frag_row<HDP>(Kt, kb * 32, ks, lane), qf[ks], ks == 0 ? nl : s);
        dp = mfma_att<F16>(frag_row<HDP>(Vt, kb * 32, ks, lane), dof[ks], ks == 0 ? nd : dp);
      }
      if (ragged && kt == nt - 1) mask_rows_past(s, kt * 64 + kb * 32, h, N);       // keys past N: p = 0, dS = 0
      bf16x8 ktr[2][DB];
      if constexpr (BTR) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int d = 0; d < DB; ++d) ktr[s2][d] = frag_tr<HDP>(Kt, kb * 32 + 16 * s2, d * 32, lane);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) dp[t] *= EXP2(s[t]);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 dsf = acc_frag_t<F16>(dp, s2);
#pragma unroll
        for (int d = 0; d < DB; ++d) dqacc[d] = mfma_att<F16>(BTR ? ktr[s2][d] : frag_tr<HDP>(Kt, kb * 32 + 16 * s2, d * 32, lane), dsf, dqacc[d]);
      }
    }
  };
  for (int kt = 0; kt < nt; kt += ATT_STAGES) {
    body(IC<0>{}, kt);
    if (kt + 1 < nt) body(IC<1>{}, kt + 1);
    if (kt + 2 < nt) body(IC<2>{}, kt + 2);
  }
  if constexpr (QKN) {
    QkPref<HD> pf;
    if (active) { qk_prefetch<HD>(pf, qn, 0, b, hh, H, N, q0, lane); qk_prefetch_cs<HD>(pf, qn, q0, lane); }
    __syncthreads();                                 // ring -> store scratch
    float cs2[2][8];
    if (active) qknorm_rows_bwd<HD>(dqacc, scale, smem + wave * 32 * (HDP * 2 + 16), lane, qn, 0, b, hh, H, N, q0, pf, cs2[0], cs2[1]);
    const int qb = lid % qblocks;
    const size_t blk = (size_t)b * qblocks + qb;
    float* const dst[2] = {qn.Pw + (blk * H + hh) * (2 * HD), qn.Pb + blk * (3 * H * HD) + (size_t)hh * HD};
    wg_colsums<HD, 2>(cs2, (float*)(smem + 4 * 32 * (HDP * 2 + 16)), wave, lane, active, min(4, (N - qb * 128) / 32), dst);
  } else {
    __syncthreads();                                   // ring -> store scratch
    if (!active) return;
    store_rows_t<HD, typename ActT<F16>::t>(dqacc, scale, smem + wave * 32 * (HDP * 2 + 16), dQ + hb + (size_t)q0 * ld, ld, lane, N - q0);
  }
}

#ifdef LDMAE_DIAG
// ================================================================================================ backward in one kernel, head_dim 16
// DIAGNOSTIC BUILD ONLY (tune key 21 = 1; ablations: tune key 22; A/B: tools/bench_attn16.py, profiles/r05_attn16_onepass.txt).  Built to
// test whether the VMAE heads' backward (two kernels, each bound by vector issue + LDS, the exponential paid twice) runs faster as one
// kernel -- and it does NOT: 1.31 ms against 0.70 + 0.54 ms at 256 x 12 heads x 1024 tokens.  Per (batch, head) the kernel below needs a
// whole workgroup (the dQ sum across key blocks must stay on chip to avoid the float atomics that sank the head-dim-64 one-pass kernel),
// which pins the CU at 2 waves per SIMD with 222 registers: with nothing else resident, every LDS round trip and MFMA result of the
// score -> exp2 -> dV / dK -> dS image -> dQ chain is exposed (removing the exponentials or the accumulator-initialising LDS reads changes
// nothing: ablations 4, 8; the stripped S / dP / dV / dK loop alone takes 1.13 ms where the 3-waves-per-SIMD dK/dV kernel does all of it in 0.70).
// The VMAE heads (models_mae.py:103-147: 12 heads of 16 over 256 .. 1024 tokens).  At head_dim 16 a score costs two MFMA k-steps and SIX
// vector-instruction slots (exp2 = 4, the dS multiply, two packed conversions): the two-kernel backward above is bound by vector issue and
// pays the exponential twice (profiles/r05_vmae_pretrain: dK/dV 0.70 + dQ 0.54 ms per 256 x 12 x 1024^2 scores against 0.56 ms of
// exponentials + multiplies for ONE pass).  Here a whole (batch, head) belongs to one 512-thread workgroup, which is what makes one pass
// possible without atomics: the dQ rows of a 64-query tile are the sum over the eight waves' key blocks, formed in LDS in a fixed order.
//   * wave w owns KSETS x 32 keys per SWEEP (key on the lane, dK^T / dV^T stationary in registers as in the dK/dV kernel); a sweep walks
//     all N queries in 64-row tiles through the 3-deep LDS-DMA ring (waves 0-3 fetch the Q image, 4-7 the dO image: one piece each);
//     N = 256 KSETS keys per sweep; longer sequences take N / (256 KSETS) sweeps over the queries, the partial dQ waiting in LDS (f32);
//   * S, dP, p = exp2(S), dS = p dP, dV^T += dO^T p, dK^T += Q^T dS exactly as in the dK/dV kernel; then the dS block [32 keys][32
//     queries] goes through a wave-private 2-KiB LDS image (written as the accumulator holds it, read back transposed by
//     ds_read_b64_tr_b16) and dQ^T[d, q] += K^T[d, keys] . dS^T[keys, q] with the K^T fragments stationary;
//   * per tile the eight partial dQ^T blocks [64 q][16 d] f32 meet in LDS, one barrier, and thread (q, d pair) adds them in wave order:
//     bitwise reproducible.  -delta and -lse log2(e) of all N rows are formed once per workgroup from O, dO and LSE (no separate pass).
template <int KSETS, bool F16>
__global__ __launch_bounds__(512) void attn_bwd_fused16_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K, const bf16* __restrict__ V,
                                                               const bf16* __restrict__ O, const bf16* __restrict__ dO, const float* __restrict__ LSE,
                                                               bf16* __restrict__ dQ, bf16* __restrict__ dK, bf16* __restrict__ dV,
                                                               int H, int N, float scale, QkvLayout L, QkvLayout Lv, int dbg) {
  constexpr int HD = 16, HDP = 32, TB = 64 * HDP * 2, STG = 2 * TB;          // one ring stage = Q image | dO image
  typedef typename ActT<F16>::t T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [3 stages][Q | dO]  |  -lse2[N] -delta[N]  |  dS images [8 waves][2 KiB]  |  partial dQ [8 waves][64 q][16 d] f32  |  dQ sums [N][16] f32
  char* const ring = smem;
  float* const nl = (float*)(smem + ATT_STAGES * STG);
  float* const nd = nl + N;
  char* const dsimg = (char*)(nd + N);
  float* const slab = (float*)(dsimg + 8 * 2048);
  float* const dqsum = slab + 8 * 64 * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  const float c = scale * 1.4426950408889634f;
  const unsigned bh = xcd_remap(blockIdx.x, gridDim.x);
  const int b = bh / H, hh = bh % H;
  const size_t hb = (size_t)b * L.sb + (size_t)hh * L.sh, hbv = (size_t)b * Lv.sb + (size_t)hh * Lv.sh;
  const long ld = L.ld, dold = (long)H * HD;
  const bf16* qp = Q + hb;
  const bf16* dop = dO + ((size_t)b * N * H + hh) * HD;
  const int nt = N / 64, sweeps = N / (256 * KSETS), total = nt * sweeps;

  // ---- ring: wave w < 4 fetches piece w of the Q image, wave w >= 4 piece w - 4 of the dO image
  const bool isq = wave < 4;
  TileMap<HD, 64> tm;
  tm.init(isq ? ld : dold, wave & 3, lane);
#pragma unroll
  for (int st = 0; st < ATT_STAGES; ++st) tm.prefill(ring + st * STG + (isq ? 0 : TB), wave & 3, lane);
  const unsigned lds0 = lds_addr_of(smem);
  const bf16* const gsrc = isq ? qp : dop;
  const long gstep = (isq ? ld : dold) * 64;
  auto stage = [&](int j) {                 // j = running tile number over all sweeps
    const int qt = j % nt;
    tm.issue(gsrc + (size_t)qt * gstep, 0, nullptr, lds0 + (j % ATT_STAGES) * STG + (isq ? 0 : TB), wave & 3, lane);
  };
  stage(0);
  if (total > 1) stage(1);

  // ---- -lse log2(e) and -delta = -sum_d dO O of every query row of the head
  for (int n = tid; n < N; n += 512) {
    const size_t oo = (((size_t)b * N + n) * H + hh) * HD;
    float d = 0.f;
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) d += frag_dot<F16>(*(const bf16x8*)(dO + oo + 8 * ch), *(const bf16x8*)(O + oo + 8 * ch));
    nl[n] = -LSE[(size_t)bh * N + n] * 1.4426950408889634f;
    nd[n] = -d;
  }

  f32x16 dkacc[KSETS], dvacc[KSETS];
  bf16x8 kf[KSETS], vf[KSETS], ktr[KSETS][2];
  char* const myimg = dsimg + wave * 2048;
  float* const myslab = slab + wave * 64 * 16;
  int j = 0;
#pragma unroll 1
  for (int sw = 0; sw < sweeps; ++sw) {
    // ---- this sweep's keys: operand fragments from global memory; K^T fragments through the (idle) slab as a dual-use image
    const int kbase = sw * 256 * KSETS + wave * 32;                            // + 256 s for set s
    __syncthreads();                                                           // previous sweep's reduce has read the slabs
#pragma unroll
    for (int s = 0; s < KSETS; ++s) {
      const bf16x8 kraw = *(const bf16x8*)(K + hb + (size_t)(kbase + 256 * s + r) * ld + 8 * h);
      vf[s] = *(const bf16x8*)(V + hbv + (size_t)(kbase + 256 * s + r) * Lv.ld + 8 * h);
      kf[s] = frag_scale_t<F16>(kraw, c);
      char* img = (char*)myslab + s * 2048;
      bf16x8 z;
#pragma unroll
      for (int i = 0; i < 8; ++i) z[i] = (bf16)0.f;
      *(bf16x8*)(img + tile_off<HDP>(r, h)) = kraw;
      *(bf16x8*)(img + tile_off<HDP>(r, 2 + h)) = z;
      dkacc[s] = splat16(0.f); dvacc[s] = splat16(0.f);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0) through the builtin: the fragments (and the ring stages in flight) have landed: no compiler-counted wait in the tile loop
    __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): the image writes
#pragma unroll
    for (int s = 0; s < KSETS; ++s)
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) ktr[s][k2] = frag_tr<HDP>((const char*)myslab + s * 2048, 16 * k2, 0, lane);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    const bool first = sw == 0, last = sw == sweeps - 1;
#pragma unroll 1
    for (int qt = 0; qt < nt; ++qt, ++j) {
      // stage j has landed: behind it in the (in-order) vector-memory queue are stage j + 1 and, in the last sweep, the dQ stores of the two tiles before
      if (j + 1 < total) { if (last && qt >= 2) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else if (last && qt == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); }
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();          // (also: the previous tile's reduce has read the slabs; sweep 0, tile 0: nl / nd and the prefill are visible)
      __builtin_amdgcn_sched_barrier(0);
      if (j + 2 < total) stage(j + 2);
      const char* Qt = ring + (j % ATT_STAGES) * STG;
      const char* dOt = Qt + TB;
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        f32x16 dq = splat16(0.f);
        const bf16x8 qrow = frag_row<HDP>(Qt, qb * 32, 0, lane), dorow = frag_row<HDP>(dOt, qb * 32, 0, lane);
        bf16x8 qtr[2], dotr[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) { qtr[s2] = frag_tr<HDP>(Qt, qb * 32 + 16 * s2, 0, lane); dotr[s2] = frag_tr<HDP>(dOt, qb * 32 + 16 * s2, 0, lane); }
#pragma unroll
        for (int s = 0; s < KSETS; ++s) {
          f32x16 sc, dp;
          if (dbg & 8) { sc = splat16(-10.f); dp = splat16(0.5f); } else
#pragma unroll
          for (int g = 0; g < 4; ++g) {          // accumulator element 4g + j <-> query row qb*32 + 8g + 4h + j
            const f32x4 a = *(const f32x4*)(nl + qt * 64 + qb * 32 + 8 * g + 4 * h), e = *(const f32x4*)(nd + qt * 64 + qb * 32 + 8 * g + 4 * h);
#pragma unroll
            for (int i = 0; i < 4; ++i) { sc[4 * g + i] = a[i]; dp[4 * g + i] = e[i]; }
          }
          sc = mfma_att<F16>(qrow, kf[s], sc);
          dp = mfma_att<F16>(dorow, vf[s], dp);
#pragma unroll
          for (int t = 0; t < 16; ++t) {
            const float p = (dbg & 4) ? sc[t] * 0.01f : EXP2(sc[t]);
            sc[t] = p;
            dp[t] *= p;
          }
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pf = acc_frag_t<F16>(sc, s2), dsf = acc_frag_t<F16>(dp, s2);
            dvacc[s] = mfma_att<F16>(dotr[s2], pf, dvacc[s]);
            dkacc[s] = mfma_att<F16>(qtr[s2], dsf, dkacc[s]);
            if (dbg & 2) continue;
            // dS^T image: row = key (this lane's), columns = queries; elements 8 s2 + 4 gg + i are queries 8 (2 s2 + gg) + 4h + i
            union { bf16x8 v; bf16x4 q[2]; } u;
            u.v = dsf;
            *(bf16x4*)(myimg + tile_off<HDP>(r, 2 * s2) + 8 * h) = u.q[0];
            *(bf16x4*)(myimg + tile_off<HDP>(r, 2 * s2 + 1) + 8 * h) = u.q[1];
          }
          // (no wait: the LDS operations of one wave execute in order, the transposed reads see the image just written)
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2) if (!(dbg & 2)) dq = mfma_att<F16>(ktr[s][k2], frag_tr<HDP>(myimg, 16 * k2, 0, lane), dq);
        }
        // rows of dq = d (elements 0..7 are d = 4h + i and 8 + 4h + i), lane column = query qb*32 + r
        float* dst = myslab + (qb * 32 + r) * 16 + 4 * h;
        if (dbg & 1) continue;
        *(f32x4*)dst = f32x4{dq[0], dq[1], dq[2], dq[3]};
        *(f32x4*)(dst + 8) = f32x4{dq[4], dq[5], dq[6], dq[7]};
      }
      if (dbg & 1) continue;
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      {
        const int q = tid >> 3, d2 = (tid & 7) * 2;
        float2 a = *(const float2*)(slab + q * 16 + d2);
#pragma unroll
        for (int w = 1; w < 8; ++w) { const float2 e = *(const float2*)(slab + (w * 64 + q) * 16 + d2); a.x += e.x; a.y += e.y; }
        float2* acc = (float2*)(dqsum + (size_t)(qt * 64 + q) * 16 + d2);
        if (!first) { const float2 e = *acc; a.x += e.x; a.y += e.y; }
        if (!last) *acc = a;
        else {
          typedef T T2 __attribute__((ext_vector_type(2)));
          T2 o2;
          o2[0] = from_f<T>(a.x * scale); o2[1] = from_f<T>(a.y * scale);
          *(T2*)(dQ + hb + (size_t)(qt * 64 + q) * ld + d2) = o2;
        }
      }
    }
    // ---- this sweep's dK / dV rows (the slabs double as the store images once every wave is through its reduce)
    __syncthreads();
    char* sw_img = (char*)myslab;
#pragma unroll
    for (int s = 0; s < KSETS; ++s) {
      store_rows_t<HD, T>(*(const f32x16(*)[1])&dkacc[s], scale, sw_img, dK + hb + (size_t)(kbase + 256 * s) * ld, ld, lane);
      store_rows_t<HD, T>(*(const f32x16(*)[1])&dvacc[s], 1.f, sw_img, dV + hbv + (size_t)(kbase + 256 * s) * Lv.ld, Lv.ld, lane);
    }
  }
}

#endif

#ifdef LDMAE_DIAG
// ================================================================================================ backward in ONE pass, bf16
// DIAGNOSTIC BUILD ONLY (tune key 17 = 1; A/B and ablations: tools/bench_attn.py --fused, profiles/r04_attn_onepass_ab.txt).  Built to
// settle the question the two-kernel backward leaves open -- it executes 7 products for the 5 the backward has -- and it LOSES on this
// shape: 4.41 ms against 1.74 + 1.35 ms.  Without its dQ stores / atomics the kernel runs 2.38 ms (the two saved products are worth
// 0.7 ms), but the dQ sum over the N / 128 key blocks is 3.2 GB of f32 adds per layer at bs 256: float atomics execute at the memory
// side at ~1.3 TB/s chip-wide (MI355X_MICROARCH 'Global float atomics'; measured here: +2.0 ms), and every store-based alternative
// (f32 read-modify-write by the owning wave, bf16 slabs + a reduce pass) moves at least as many bytes through the fabric, because stores
// never stay in L2.  With 256 keys per workgroup the sum still costs >= 1 ms -- more than the 0.7 ms the products save.
// Shape: (LightningDiT block shape: hd 64, N a multiple of 128, QK-norm / RoPE backward fused.)  The two-pass form executes 7 products for the 5 the
// backward has (S and dP are formed in both kernels).  Here a workgroup owns one (batch, head) and walks its key blocks of 128 (wave = 32 keys,
// dK / dV stationary in registers, as in the dK/dV kernel); for every 64-query tile the dS block [128 keys][64 queries] goes through LDS once
// (bf16, the dual-use image: written by rows, read transposed) and each wave forms one 32 x 32 block of dQ[q, d] = dS[q, :] . K[:, d] over all
// 128 keys, with its K^T fragments stationary in registers.  dQ is accumulated in an f32 scratch [B*H][N][64]: plain stores in the first
// key block, f32 atomic adds at L2 in the others -- every element is only ever touched by ONE wave (same (tile, block) -> same wave in every
// key block), in program order, so the sum order is fixed and the result bitwise reproducible.  attn_dq_finish_kernel then runs the
// QK-norm / RoPE backward on the finished rows.  -delta / -lse*log2(e) come from attn_rowc_kernel (the dQ kernel that used to publish them is gone).
template <int PPW> __device__ __forceinline__ void onepass_wait(int ahead, int batches) {
  // in flight behind the stage that must have landed: `ahead` later stages (PPW DMA pieces each) and `batches` x 16 dQ stores / atomics
  if (ahead) {
    if (batches >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + 32) : "memory");
    else if (batches == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + 16) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
  } else {
    if (batches >= 2) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else if (batches == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}
template <int HD, int DBG = 0>       // DBG (diagnostic build, timing only): 1 = no dQ stores / atomics, 2 = no dQ product and no second barrier, 3 = no dS image either
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_onepass_bf16_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K, const bf16* __restrict__ V,
                                                                    const bf16* __restrict__ dO, const float* __restrict__ ROWC, long rc_stride,
                                                                    bf16* __restrict__ dV, float* __restrict__ DQ,
                                                                    int H, int N, float scale, QkvLayout L, QkvLayout Lv, QkNormBwd qn) {
  static_assert(HD == 64, "one-pass backward: head dim 64");
  constexpr int KS = HD / 16, DB = HD / 32, TB = 64 * HD * 2, BUF = 2 * TB + 1024;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [STAGES][Q tile | dO tile | -lse2[64] -delta[64] scratch[128]] | dS image [128][64]
  char* const dsimg = smem + ATT_STAGES * BUF;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, h = lane >> 5;
  const float c = scale * 1.4426950408889634f;
  const int kblocks = N / 128, nt = N / 64;
  const int bh = xcd_remap(blockIdx.x, gridDim.x);
  const int b = bh / H, hh = bh % H;
  const size_t hb = (size_t)b * L.sb + (size_t)hh * L.sh, hbv = (size_t)b * Lv.sb + (size_t)hh * Lv.sh;
  const long ld = L.ld;
  const bf16* qp = Q + hb;
  const bf16* dop = dO + ((size_t)b * N * H + hh) * HD;     // row stride H*HD
  const long dold = (long)H * HD;
  constexpr int PPW = 2 * (TB / 1024) / 4 + 1;
  const float* rowc = ROWC + (size_t)bh * N + ((wave & 1) ? 0 : rc_stride);     // wave 0/2: -lse2, wave 1/3: -delta
  TileMap<HD, 64> mq, mo;
  mq.init(ld, wave, lane);
  mo.init(dold, wave, lane);
  const unsigned lds0 = lds_addr_of(smem);
  auto stage = [&](int qt) {
    const int so = (qt % ATT_STAGES) * BUF;
    mq.issue(qp + (size_t)qt * 64 * ld, ld, smem + so, lds0 + so, wave, lane);
    mo.issue(dop + (size_t)qt * 64 * dold, dold, smem + so + TB, lds0 + so + TB, wave, lane);
    glds4_s(rowc + qt * 64, lane * 4, lds0 + so + 2 * TB + wave * 256);
  };
  // this lane's 8-byte slots in the dS image: row = its key, chunk (4 qb + c) of the row at dsw + 512 qb + wx[c] (tile_off, + the half h)
  const int krow = wave * 32 + r;
  char* const dsw = dsimg + (HD * 16) * (krow >> 3) + 64 * (krow & 7) + 8 * h;
  int wx[4];
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) wx[cc] = 16 * (cc ^ ((krow >> 2) & 3));
  const int qbi = wave >> 1, dbi = wave & 1;       // this wave's block of the tile's dQ: queries 32 qbi.., columns 32 dbi..
  // dQ scratch: element t of the block is row acc_row(t, h) = (t & 3) + 8 (t >> 2) + 4 h, this lane's column is 32 dbi + r
  float* const dqw = DQ + ((size_t)bh * N + qbi * 32 + 4 * h) * HD + dbi * 32 + r;

  for (int kb = 0; kb < kblocks; ++kb) {
    const int k0 = kb * 128 + wave * 32;
    if (kb) __syncthreads();                       // the epilogue's scratch (and its column-sum rows) -> ring
    // the key block's 128 rows into the first ring stage (two 64-row images), for the transposed fragments of the dQ product
    stage_tile<HD, 64>(K + hb + (size_t)(kb * 128) * ld, ld, 63, smem, wave, lane);
    stage_tile<HD, 64>(K + hb + (size_t)(kb * 128 + 64) * ld, ld, 63, smem + TB, wave, lane);
    bf16x8 kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[ks] = frag_scale(gfrag<HD>(K + hb + (size_t)(k0 + r) * ld, ks * 16 + 8 * h), c);
      vf[ks] = gfrag<HD>(V + hbv + (size_t)(k0 + r) * Lv.ld, ks * 16 + 8 * h);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0), visible to the compiler (see the dK/dV kernel)
    __syncthreads();
    bf16x8 kT[8];                                  // K^T fragments: keys 16 s .. 16 s + 15 of the block x this wave's 32 columns
#pragma unroll
    for (int s = 0; s < 8; ++s) kT[s] = frag_tr<HD>(smem + (s >> 2) * TB, 16 * (s & 3), dbi * 32, lane);
    __syncthreads();                               // every wave has its fragments: the ring may overwrite the images
    f32x16 dkacc[DB], dvacc[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d) { dkacc[d] = splat16(0.f); dvacc[d] = splat16(0.f); }
#pragma unroll
    for (int st = 0; st < ATT_STAGES - 1; ++st) stage(st);
    auto body = [&](auto ST, int qt) {
      constexpr int st = decltype(ST)::value;
      onepass_wait<PPW>(qt + 1 < nt ? 1 : 0, DBG ? 0 : min(qt, 2));
      __builtin_amdgcn_s_barrier();
      if (qt + ATT_STAGES - 1 < nt) stage(qt + ATT_STAGES - 1);
      const char* Qt = smem + st * BUF;
      const char* dOt = Qt + TB;
      const float* nl = (const float*)(Qt + 2 * TB);
      const float* nd = nl + 64;
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        f32x16 s, dp;
#pragma unroll
        for (int g = 0; g < 4; ++g) {              // accumulator element 4g+j <-> query row qb*32 + 8g + 4h + j
          const f32x4 a = *(const f32x4*)(nl + qb * 32 + 8 * g + 4 * h), e = *(const f32x4*)(nd + qb * 32 + 8 * g + 4 * h);
#pragma unroll
          for (int j = 0; j < 4; ++j) { s[4 * g + j] = a[j]; dp[4 * g + j] = e[j]; }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          s = MFMA_BF16(frag_row<HD>(Qt, qb * 32, ks, lane), kf[ks], s);
          dp = MFMA_BF16(frag_row<HD>(dOt, qb * 32, ks, lane), vf[ks], dp);
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const float p = EXP2(s[t]);
          s[t] = p;
          dp[t] *= p;
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 pf = acc_frag(s, s2), dsf = acc_frag(dp, s2);
          union { bf16x8 v; uint2 u[2]; } w;
          w.v = dsf;                               // queries qb*32 + 16 s2 + 4h + 0..3 | + 8 + 4h + 0..3 of this lane's key
          if (DBG < 3) {
            *(uint2*)(dsw + 512 * qb + wx[2 * s2]) = w.u[0];
            *(uint2*)(dsw + 512 * qb + wx[2 * s2 + 1]) = w.u[1];
          }
#pragma unroll
          for (int d = 0; d < DB; ++d) {
            dvacc[d] = MFMA_BF16(frag_tr<HD>(dOt, qb * 32 + 16 * s2, d * 32, lane), pf, dvacc[d]);
            dkacc[d] = MFMA_BF16(frag_tr<HD>(Qt, qb * 32 + 16 * s2, d * 32, lane), dsf, dkacc[d]);
          }
        }
      }
      if (DBG >= 2) return;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                // dS of all 128 keys is in LDS (the next tile's first barrier keeps it until every wave has read it)
      f32x16 dq = splat16(0.f);
#pragma unroll
      for (int s = 0; s < 8; ++s) dq = MFMA_BF16(frag_tr<HD>(dsimg, 16 * s, qbi * 32, lane), kT[s], dq);
      float* const dst = dqw + (size_t)qt * 64 * HD;
      if (DBG == 1) {
        if (dq[0] + dq[5] + dq[10] + dq[15] == 123.456f) dst[0] = dq[3];      // keeps the product alive
      } else if (kb == 0) {
#pragma unroll
        for (int t = 0; t < 16; ++t) dst[((t & 3) + 8 * (t >> 2)) * HD] = dq[t] * scale;
      } else {
#pragma unroll
        for (int t = 0; t < 16; ++t)
          __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)(dst + ((t & 3) + 8 * (t >> 2)) * HD), dq[t] * scale);
      }
    };
    for (int qt = 0; qt < nt; qt += ATT_STAGES) {
      body(IC<0>{}, qt);
      if (qt + 1 < nt) body(IC<1>{}, qt + 1);
      if (qt + 2 < nt) body(IC<2>{}, qt + 2);
    }
    // dk -> k slot of dqkv through the QK-norm / RoPE backward; dv + its column sums (qkv bias gradient): as the dK/dV kernel
    QkPref<HD> pf;
    qk_prefetch<HD>(pf, qn, 1, b, hh, H, N, k0, lane);
    __syncthreads();                               // ring -> store scratch
    char* sw = smem + wave * 32 * (HD * 2 + 16);
    float cs3[3][8];
    qk_prefetch_cs<HD>(pf, qn, k0, lane);
    qknorm_rows_bwd<HD>(dkacc, scale, sw, lane, qn, 1, b, hh, H, N, k0, pf, cs3[0], cs3[1]);
    store_rows_t_colsum<HD>(dvacc, 1.f, sw, dV + hbv + (size_t)k0 * Lv.ld, Lv.ld, lane, cs3[2]);
    const size_t blk = (size_t)b * kblocks + kb;
    float* const dst[3] = {qn.Pw + (blk * H + hh) * (2 * HD) + HD, qn.Pb + blk * (3 * H * HD) + ((size_t)H + hh) * HD,
                           qn.Pb + blk * (3 * H * HD) + ((size_t)2 * H + hh) * HD};
    wg_colsums<HD, 3>(cs3, (float*)(smem + 4 * 32 * (HD * 2 + 16)), wave, lane, true, 4, dst);
  }
}

// ROWC for the one-pass kernel: [0] = -delta = -rowsum(dO * O), [1] = -lse * log2(e); o / do token-major [B,N,H,hd], rows of ROWC head-major
template <int HD>
__global__ __launch_bounds__(256) void attn_rowc_kernel(const bf16* __restrict__ O, const bf16* __restrict__ dO, const float* __restrict__ LSE,
                                                        float* __restrict__ ROWC, long rc_stride, int H, int N, long items) {
  constexpr int LPR = HD / 8;                      // lanes per (b, n, h) row, 16 B each
  const long it = ((long)blockIdx.x * 256 + threadIdx.x) / LPR;
  const int sub = threadIdx.x % LPR;
  if (it >= items) return;
  const int hh = it % H, n = (it / H) % N;
  const long b = it / ((long)H * N);
  const bf16x8 o = *(const bf16x8*)(O + (size_t)it * HD + sub * 8), g = *(const bf16x8*)(dO + (size_t)it * HD + sub * 8);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += (float)g[j] * (float)o[j];
  s = group_sum<LPR>(s);
  if (sub == 0) {
    const size_t dst = ((size_t)b * H + hh) * N + n;
    ROWC[dst] = -s;
    ROWC[rc_stride + dst] = -LSE[dst] * 1.4426950408889634f;
  }
}

// dQ scratch (f32, already scaled) -> q slot of dqkv through the QK-norm / RoPE backward, + the per-block column sums (as the dQ kernel's epilogue)
template <int HD>
__global__ __launch_bounds__(256) void attn_dq_finish_kernel(const float* __restrict__ DQ, int H, int N, QkNormBwd qn) {
  constexpr int CPR = HD / 8, NIT = 32 * CPR / 64;
  __shared__ float red[4 * 2 * HD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qblocks = N / 128;
  const int bh = blockIdx.x / qblocks, qb = blockIdx.x % qblocks, q0 = qb * 128 + wave * 32;
  const int b = bh / H, hh = bh % H;
  QkPref<HD> pf;
  qk_prefetch<HD>(pf, qn, 0, b, hh, H, N, q0, lane);
  qk_prefetch_cs<HD>(pf, qn, q0, lane);
  float4 g4[NIT][2];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const float* src = DQ + ((size_t)bh * N + q0 + (it * 64 + lane) / CPR) * HD + (lane % CPR) * 8;
    g4[it][0] = *(const float4*)src; g4[it][1] = *(const float4*)(src + 4);
  }
  float cs2[2][8];
  qknorm_rows_math<HD>([&](int it) {
    bf16x8 v;
    v[0] = (bf16)g4[it][0].x; v[1] = (bf16)g4[it][0].y; v[2] = (bf16)g4[it][0].z; v[3] = (bf16)g4[it][0].w;
    v[4] = (bf16)g4[it][1].x; v[5] = (bf16)g4[it][1].y; v[6] = (bf16)g4[it][1].z; v[7] = (bf16)g4[it][1].w;
    return v; }, lane, qn, 0, b, hh, H, N, q0, pf, cs2[0], cs2[1]);
  const size_t blk = (size_t)b * qblocks + qb;
  float* const dst[2] = {qn.Pw + (blk * H + hh) * (2 * HD), qn.Pb + blk * (3 * H * HD) + (size_t)hh * HD};
  wg_colsums<HD, 2>(cs2, red, wave, lane, true, 4, dst);
}
#endif  // LDMAE_DIAG

// ================================================================================================ f32 path (parity)
// Tiles are f32 [rows][LD] with LD = HD + 1 (odd stride: conflict-free row reads); operands are single
// floats: A[i = lane&31][k = lane>>5], B[k = lane>>5][j = lane&31] per 32x32x2 step.
template <int HD, int ROWS>
__device__ __forceinline__ void stage_tile_f32(const float* __restrict__ g, long ld, int nrows_valid, float* lds) {
  constexpr int LD = HD + 1;
  for (int i = threadIdx.x; i < ROWS * HD; i += 256) {
    const int row = i / HD, d = i % HD;
    lds[row * LD + d] = row < nrows_valid ? g[(long)row * ld + d] : 0.f;
  }
}

template <int HD>
__global__ __launch_bounds__(256) void attn_fwd_f32_kernel(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
                                                           float* __restrict__ O, float* __restrict__ LSE, int H, int N, float c) {
  constexpr int LD = HD + 1, DB = (HD + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Qs = (float*)smem;            // [128][LD]
  float* Ks = Qs + 128 * LD;           // [64][LD]
  float* Vs = Ks + 64 * LD;            // [64][LD]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int qblocks = (N + 127) / 128;
  const int bh = blockIdx.x / qblocks, qb0 = (blockIdx.x % qblocks) * 128, q0 = qb0 + wave * 32;
  const bool active = q0 < N;
  stage_tile_f32<HD, 128>(Q + ((size_t)bh * N + qb0) * HD, HD, N - qb0, Qs);
  f32x16 oacc[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int t = 0; t < 16; ++t) oacc[d][t] = 0.f;
  float ms = -1e30f, l = 0.f;
  for (int kt = 0; kt < (N + 63) / 64; ++kt) {
    __syncthreads();
    stage_tile_f32<HD, 64>(K + ((size_t)bh * N + kt * 64) * HD, HD, N - kt * 64, Ks);      // rows past N: zeros, masked below
    stage_tile_f32<HD, 64>(V + ((size_t)bh * N + kt * 64) * HD, HD, N - kt * 64, Vs);
    __syncthreads();
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int t = 0; t < 16; ++t) s[kb][t] = 0.f;
#pragma unroll 8
      for (int kk = 0; kk < HD; kk += 2) s[kb] = MFMA_F32(Ks[(kb * 32 + r) * LD + kk + h], Qs[(wave * 32 + r) * LD + kk + h], s[kb]);
      if (kt * 64 + 64 > N) mask_rows_past(s[kb], kt * 64 + kb * 32, h, N);
    }
    float mx = s[0][0];
#pragma unroll
    for (int t = 0; t < 16; ++t) mx = fmaxf(mx, fmaxf(s[0][t], s[1][t]));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float ms_new = fmaxf(ms, mx * c), alpha = exp2f(ms - ms_new);
    ms = ms_new;
    float rs = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int t = 0; t < 16; ++t) { const float p = exp2f(s[kb][t] * c - ms); s[kb][t] = p; rs += p; }
    l = l * alpha + rs;
#pragma unroll
    for (int d = 0; d < DB; ++d) {
#pragma unroll
      for (int t = 0; t < 16; ++t) oacc[d][t] *= alpha;
      const bool dv = d * 32 + r < HD;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const float a = dv ? Vs[(kb * 32 + acc_row(t, h)) * LD + d * 32 + r] : 0.f;
          oacc[d] = MFMA_F32(a, s[kb][t], oacc[d]);
        }
    }
  }
  l += __shfl_xor(l, 32, 64);
  if (!active) return;
  const float inv = 1.f / l;
  const int b = bh / H, hh = bh % H;
  float* op = O + ((size_t)(b * N + q0 + r) * H + hh) * HD;
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int dd = d * 32 + acc_row(t, h);
      if (dd < HD && q0 + r < N) op[dd] = oacc[d][t] * inv;
    }
  if (h == 0 && q0 + r < N) LSE[(size_t)bh * N + q0 + r] = (ms + log2f(l)) * 0.6931471805599453f;
}

// Head dim 16 (the VMAE heads: what `_encode` / `decode_to_images` run in the reference's f32): the 32x32x2 form above pads the P.V product to 32
// output rows (a third of its MFMA cycles wasted) and stages its tiles synchronously.  Here everything is 16x16x4 f32 MFMAs: S^T blocks
// [16 keys][16 queries] = K . Q^T (4 MFMAs per block, the 16 dims), and -- the accumulator-as-operand trick of the bf16 kernels in its 16x16
// form -- element r of a score block's accumulator IS the B operand of step r of O^T[16 dims][16 queries] += V^T . P with contraction slot
// g = key 4 g + r of the block; the A operand reads V[key][dim] from LDS.  A wave owns 32 queries (two column blocks that share every K / V
// read), tiles of 64 keys, the next tile's rows in flight in registers under the current tile's arithmetic.  Running maximum per tile
// as in the general kernel; the exponentials are the bare v_exp_f32 (arguments <= 0: what it flushes to zero lies below 2^-126 of the row maximum).
__global__ __launch_bounds__(256) void attn_fwd_f32_hd16_kernel(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
                                                                float* __restrict__ O, float* __restrict__ LSE, int H, int N, float c, QkvLayout L) {
  constexpr int HD = 16, LD = 20;              // LDS row pitch in floats: 16-B aligned rows; both operand read patterns conflict-free (banks 20 key + g, 80 g + d)
  __shared__ __attribute__((aligned(16))) float Ks[2][64 * LD], Vs[2][64 * LD];      // double-buffered: one barrier per tile
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  const int qblocks = (N + 127) / 128;
  const int bh = blockIdx.x / qblocks, q0 = (blockIdx.x % qblocks) * 128 + wave * 32;
  const size_t hb = (size_t)(bh / H) * L.sb + (size_t)(bh % H) * L.sh;      // head-major [B,H,N,16] or the packed token-major qkv (QkvLayout)
  const long ld = L.ld;
  const float* kb = K + hb;
  const float* vb = V + hb;
  float qv[2][4];                              // Q[q0 + 16 qb + j][4 s + g] * c: the B operand of the score products
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) qv[qb][s4] = Q[hb + (size_t)min(q0 + 16 * qb + j, N - 1) * ld + 4 * s4 + g] * c;
  f32x4 oacc[2];
  float ms[2], l[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) { oacc[qb] = (f32x4){0.f, 0.f, 0.f, 0.f}; ms[qb] = -1e30f; l[qb] = 0.f; }
  const int nt = (N + 63) / 64;
  const int srow = threadIdx.x >> 2, sch = (threadIdx.x & 3) * 4;            // this thread's 16 B of every K and V tile
  auto fetch = [&](int kt, float4& kr, float4& vr) {
    const int row = kt * 64 + srow;
    if (row < N) { kr = *(const float4*)(kb + (size_t)row * ld + sch); vr = *(const float4*)(vb + (size_t)row * ld + sch); }
    else { kr = make_float4(0.f, 0.f, 0.f, 0.f); vr = kr; }                  // rows past N: zeros, masked below
  };
  float4 kr, vr;
  fetch(0, kr, vr);
  *(float4*)&Ks[0][srow * LD + sch] = kr;
  *(float4*)&Vs[0][srow * LD + sch] = vr;
  __syncthreads();
  if (nt > 1) fetch(1, kr, vr);                // tile kt + 1 waits in registers while tile kt is computed
  for (int kt = 0; kt < nt; ++kt) {
    const float* Kt = Ks[kt & 1];
    const float* Vt = Vs[kt & 1];
    f32x4 sc[2][4];                            // [query block][key block]: rows = keys 16 kb + 4 g + r, column = query j
#pragma unroll
    for (int kbk = 0; kbk < 4; ++kbk) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) sc[qb][kbk] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const float a = Kt[(16 * kbk + j) * LD + 4 * s4 + g];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) sc[qb][kbk] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, qv[qb][s4], sc[qb][kbk], 0, 0, 0);
      }
    }
    if (kt * 64 + 64 > N) {                    // ragged last tile: keys past N drop out of the softmax
#pragma unroll
      for (int kbk = 0; kbk < 4; ++kbk)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kt * 64 + 16 * kbk + 4 * g + r >= N) { sc[0][kbk][r] = -__builtin_inff(); sc[1][kbk][r] = -__builtin_inff(); }
    }
    float alpha[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float mx = sc[qb][0][0];
#pragma unroll
      for (int kbk = 0; kbk < 4; ++kbk)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[qb][kbk][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float ms_new = fmaxf(ms[qb], mx);
      alpha[qb] = EXP2(ms[qb] - ms_new);
      ms[qb] = ms_new;
      float rs = 0.f;
#pragma unroll
      for (int kbk = 0; kbk < 4; ++kbk)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float pp = EXP2(sc[qb][kbk][r] - ms_new); sc[qb][kbk][r] = pp; rs += pp; }
      l[qb] = l[qb] * alpha[qb] + rs;
#pragma unroll
      for (int r = 0; r < 4; ++r) oacc[qb][r] *= alpha[qb];
    }
#pragma unroll
    for (int kbk = 0; kbk < 4; ++kbk)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = Vt[(16 * kbk + 4 * g + r) * LD + j];                  // V^T[dim j][slot g] of step r
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) oacc[qb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, sc[qb][kbk][r], oacc[qb], 0, 0, 0);
      }
    if (kt + 1 < nt) {                         // the other buffer was last read one tile ago, before the previous barrier
      *(float4*)&Ks[(kt + 1) & 1][srow * LD + sch] = kr;
      *(float4*)&Vs[(kt + 1) & 1][srow * LD + sch] = vr;
      __syncthreads();
      if (kt + 2 < nt) fetch(kt + 2, kr, vr);
    }
  }
  const int b = bh / H, hh = bh % H;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float lt = l[qb];
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
    const int q = q0 + 16 * qb + j;
    if (q < N) {
      const float inv = 1.f / lt;              // rows of O^T in this lane: dims 4 g .. 4 g + 3 of query q
      *(float4*)(O + ((size_t)(b * N + q) * H + hh) * HD + 4 * g) = make_float4(oacc[qb][0] * inv, oacc[qb][1] * inv, oacc[qb][2] * inv, oacc[qb][3] * inv);
      if (g == 0) LSE[(size_t)bh * N + q] = (ms[qb] + log2f(lt)) * 0.6931471805599453f;
    }
  }
}

template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_dkdv_f32_kernel(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
                                                                const float* __restrict__ dO, const float* __restrict__ LSE,
                                                                const float* __restrict__ DELTA, float* __restrict__ dK, float* __restrict__ dV,
                                                                int H, int N, float scale) {
  constexpr int LD = HD + 1, DB = (HD + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Ks = (float*)smem;            // [128][LD]
  float* Vs = Ks + 128 * LD;           // [128][LD]
  float* Qs = Vs + 128 * LD;           // [64][LD]
  float* dOs = Qs + 64 * LD;           // [64][LD]
  float* lse2 = dOs + 64 * LD;         // [64]
  float* dl = lse2 + 64;               // [64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const float c = scale * 1.4426950408889634f;
  const int kblocks = (N + 127) / 128;
  const int bh = blockIdx.x / kblocks, kb0 = (blockIdx.x % kblocks) * 128, k0 = kb0 + wave * 32;
  const bool active = k0 < N;
  const int b = bh / H, hh = bh % H;
  stage_tile_f32<HD, 128>(K + ((size_t)bh * N + kb0) * HD, HD, N - kb0, Ks);
  stage_tile_f32<HD, 128>(V + ((size_t)bh * N + kb0) * HD, HD, N - kb0, Vs);
  f32x16 dkacc[DB], dvacc[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int t = 0; t < 16; ++t) { dkacc[d][t] = 0.f; dvacc[d][t] = 0.f; }
  for (int qt = 0; qt < (N + 63) / 64; ++qt) {
    __syncthreads();
    stage_tile_f32<HD, 64>(Q + ((size_t)bh * N + qt * 64) * HD, HD, N - qt * 64, Qs);
    stage_tile_f32<HD, 64>(dO + (((size_t)b * N + qt * 64) * H + hh) * HD, (long)H * HD, N - qt * 64, dOs);
    // query rows past N: lse = +inf -> p = exp2(0 - inf) = 0, so they add nothing to dV or dK
    if (threadIdx.x < 64) lse2[threadIdx.x] = qt * 64 + (int)threadIdx.x < N ? LSE[(size_t)bh * N + qt * 64 + threadIdx.x] * 1.4426950408889634f : __builtin_inff();
    else if (threadIdx.x < 128) dl[threadIdx.x - 64] = qt * 64 + (int)threadIdx.x - 64 < N ? DELTA[(size_t)bh * N + qt * 64 + threadIdx.x - 64] : 0.f;
    __syncthreads();
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x16 s, dp;
#pragma unroll
      for (int t = 0; t < 16; ++t) { s[t] = 0.f; dp[t] = 0.f; }
#pragma unroll 8
      for (int kk = 0; kk < HD; kk += 2) {
        s = MFMA_F32(Qs[(qb * 32 + r) * LD + kk + h], Ks[(wave * 32 + r) * LD + kk + h], s);
        dp = MFMA_F32(dOs[(qb * 32 + r) * LD + kk + h], Vs[(wave * 32 + r) * LD + kk + h], dp);
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int row = qb * 32 + acc_row(t, h);
        const float p = exp2f(s[t] * c - lse2[row]);
        s[t] = p;
        dp[t] = p * (dp[t] - dl[row]);
      }
#pragma unroll
      for (int d = 0; d < DB; ++d) {
        const bool dv = d * 32 + r < HD;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int row = qb * 32 + acc_row(t, h);
          dvacc[d] = MFMA_F32(dv ? dOs[row * LD + d * 32 + r] : 0.f, s[t], dvacc[d]);
          dkacc[d] = MFMA_F32(dv ? Qs[row * LD + d * 32 + r] : 0.f, dp[t], dkacc[d]);
        }
      }
    }
  }
  if (!active || k0 + r >= N) return;
  float* dkp = dK + ((size_t)bh * N + k0 + r) * HD;
  float* dvp = dV + ((size_t)bh * N + k0 + r) * HD;
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int dd = d * 32 + acc_row(t, h);
      if (dd < HD) { dkp[dd] = dkacc[d][t] * scale; dvp[dd] = dvacc[d][t]; }
    }
}

template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_dq_f32_kernel(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
                                                              const float* __restrict__ dO, const float* __restrict__ LSE,
                                                              const float* __restrict__ DELTA, float* __restrict__ dQ, int H, int N, float scale) {
  constexpr int LD = HD + 1, DB = (HD + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Qs = (float*)smem;            // [128][LD]
  float* dOs = Qs + 128 * LD;          // [128][LD]
  float* Ks = dOs + 128 * LD;          // [64][LD]
  float* Vs = Ks + 64 * LD;            // [64][LD]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const float c = scale * 1.4426950408889634f;
  const int qblocks = (N + 127) / 128;
  const int bh = blockIdx.x / qblocks, qb0 = (blockIdx.x % qblocks) * 128, q0 = qb0 + wave * 32;
  const bool active = q0 < N;
  const int b = bh / H, hh = bh % H;
  const int qrow = min(q0 + r, N - 1);
  stage_tile_f32<HD, 128>(Q + ((size_t)bh * N + qb0) * HD, HD, N - qb0, Qs);
  stage_tile_f32<HD, 128>(dO + (((size_t)b * N + qb0) * H + hh) * HD, (long)H * HD, N - qb0, dOs);
  const float lse2 = LSE[(size_t)bh * N + qrow] * 1.4426950408889634f, dl = DELTA[(size_t)bh * N + qrow];
  f32x16 dqacc[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int t = 0; t < 16; ++t) dqacc[d][t] = 0.f;
  for (int kt = 0; kt < (N + 63) / 64; ++kt) {
    __syncthreads();
    stage_tile_f32<HD, 64>(K + ((size_t)bh * N + kt * 64) * HD, HD, N - kt * 64, Ks);
    stage_tile_f32<HD, 64>(V + ((size_t)bh * N + kt * 64) * HD, HD, N - kt * 64, Vs);
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 s, dp;
#pragma unroll
      for (int t = 0; t < 16; ++t) { s[t] = 0.f; dp[t] = 0.f; }
#pragma unroll 8
      for (int kk = 0; kk < HD; kk += 2) {
        s = MFMA_F32(Ks[(kb * 32 + r) * LD + kk + h], Qs[(wave * 32 + r) * LD + kk + h], s);
        dp = MFMA_F32(Vs[(kb * 32 + r) * LD + kk + h], dOs[(wave * 32 + r) * LD + kk + h], dp);
      }
      if (kt * 64 + 64 > N) mask_rows_past(s, kt * 64 + kb * 32, h, N);       // keys past N: p = 0
#pragma unroll
      for (int t = 0; t < 16; ++t) dp[t] = exp2f(s[t] * c - lse2) * (dp[t] - dl);
#pragma unroll
      for (int d = 0; d < DB; ++d) {
        const bool dv = d * 32 + r < HD;
#pragma unroll
        for (int t = 0; t < 16; ++t)
          dqacc[d] = MFMA_F32(dv ? Ks[(kb * 32 + acc_row(t, h)) * LD + d * 32 + r] : 0.f, dp[t], dqacc[d]);
      }
    }
  }
  if (!active || q0 + r >= N) return;
  float* dqp = dQ + ((size_t)bh * N + q0 + r) * HD;
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int dd = d * 32 + acc_row(t, h);
      if (dd < HD) dqp[dd] = dqacc[d][t] * scale;
    }
}

// ================================================================================================ C ABI
#define ATTN_HD_DISPATCH(hd, MACRO) \
  switch (hd) { case 16: MACRO(16); break; case 32: MACRO(32); break; case 64: MACRO(64); break; case 72: MACRO(72); break; case 128: MACRO(128); break; \
    default: LDMAE_FAIL(LDMAE_ERR_INVALID, "attention(bf16): head_dim %d unsupported (16, 32, 64, 72, 128)", hd); }
#define ATTN_HD_DISPATCH_F32(hd, MACRO) \
  switch (hd) { case 16: MACRO(16); break; case 32: MACRO(32); break; case 64: MACRO(64); break; case 72: MACRO(72); break; case 80: MACRO(80); break; \
    case 96: MACRO(96); break; case 128: MACRO(128); break; /* 80 / 96: mae_vit_huge's heads of 80 and anything up to 96 -- the f32 BACKWARD's LDS tiles do not fit at 128 */ \
    default: LDMAE_FAIL(LDMAE_ERR_INVALID, "attention(f32): head_dim %d unsupported (16, 32, 64, 72, 80, 96, 128)", hd); }

// dynamic LDS of the bf16 kernels: the K/V (Q/dO) ring, and at least the 4 per-wave store images that re-use it at the end
static int attn_lds(int hd, int extra) {
  const int hdp = hd_pad(hd), ring = ATT_STAGES * (2 * 64 * hdp * 2 + extra), scratch = 4 * 32 * (hdp * 2 + 16);
  return ring > scratch ? ring : scratch;
}

static int attn_check(const char* who, int dtype, int B, int H, int N, int hd) {
  ldmae_count(dtype == LDMAE_F16 ? LDMAE_COUNT_ATTN_F16 : (dtype == LDMAE_BF16 ? LDMAE_COUNT_ATTN_BF16 : LDMAE_COUNT_ATTN_F32));
  LDMAE_REQUIRE(dtype == LDMAE_F32 || dtype == LDMAE_BF16 || dtype == LDMAE_F16, "%s: bad dtype %d", who, dtype);
  LDMAE_REQUIRE(B > 0 && H > 0 && N > 0 && hd > 0, "%s: empty problem", who);
  return LDMAE_OK;      // any N: the last 64-row tile of a sweep may be ragged (mask_rows_past)
}

static int attention_fwd_core(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int N, int hd,
                              float scale, QkvLayout Lq, QkvLayout Lv, hipStream_t st, const float* score_bound = nullptr, int sb_heads = 0) {
  const unsigned grid = (unsigned)B * H * ((N + 127) / 128);
  const float c = scale * 1.4426950408889634f;
  if (dtype == LDMAE_F16) {
    // the TF32-class forward (fp16 operands): head_dim 16, the VMAE heads
    LDMAE_REQUIRE(hd == 16 && score_bound == nullptr, "attention_fwd(fp16): head_dim 16 only (the VMAE heads), no static bound");
#define LRH(R) { hipFuncSetAttribute((const void*)attn_fwd_bf16_kernel<16, R, true>, hipFuncAttributeMaxDynamicSharedMemorySize, attn_lds(16, 0)); \
    hipLaunchKernelGGL((attn_fwd_bf16_kernel<16, R, true>), dim3(grid), dim3(256), attn_lds(16, 0), st, (const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)o, lse, H, N, c, Lq, Lv, (const float*)nullptr, 0); }
    if (N % 64 == 0) LRH(false) else LRH(true)
#undef LRH
  } else if (dtype == LDMAE_BF16) {
#define LR(HD, R) { hipFuncSetAttribute((const void*)attn_fwd_bf16_kernel<HD, R>, hipFuncAttributeMaxDynamicSharedMemorySize, attn_lds(HD, 0)); \
    hipLaunchKernelGGL((attn_fwd_bf16_kernel<HD, R>), dim3(grid), dim3(256), attn_lds(HD, 0), st, (const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)o, lse, H, N, c, Lq, Lv, score_bound, sb_heads); }
#define L(HD) if (N % 64 == 0) LR(HD, false) else LR(HD, true)
    ATTN_HD_DISPATCH(hd, L);
#undef L
#undef LR
  } else {
    LDMAE_REQUIRE((hd == 16 && Lq.ld == Lv.ld && Lq.sh == Lv.sh && Lq.sb == Lv.sb) ||
                  (Lq.ld == hd && Lq.sh == (long)N * hd && Lv.ld == hd && Lv.sh == (long)N * hd),
                  "attention_fwd(f32): head-major q/k/v only (head_dim 16: also the packed qkv)");
#define L(HD) { const size_t lds = (size_t)(128 + 64 + 64) * (HD + 1) * 4; \
    hipFuncSetAttribute((const void*)attn_fwd_f32_kernel<HD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(attn_fwd_f32_kernel<HD>, dim3(grid), dim3(256), lds, st, (const float*)q, (const float*)k, (const float*)v, (float*)o, lse, H, N, c); }
    if (hd == 16) hipLaunchKernelGGL(attn_fwd_f32_hd16_kernel, dim3(grid), dim3(256), 0, st, (const float*)q, (const float*)k, (const float*)v, (float*)o, lse, H, N, c, Lq);
    else ATTN_HD_DISPATCH_F32(hd, L);
#undef L
  }
  LDMAE_CHECK_LAUNCH("attention_fwd");
  return LDMAE_OK;
}

extern "C" int ldmae_attention_fwd(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int N, int hd,
                                   float scale, void* stream) {
  LDMAE_REQUIRE(q && k && v && o && lse, "attention_fwd: null pointer");
  if (int e = attn_check("attention_fwd", dtype, B, H, N, hd)) return e;
  const QkvLayout hm{(long)H * N * hd, (long)N * hd, (long)hd};
  return attention_fwd_core(dtype, q, k, v, o, lse, B, H, N, hd, scale, hm, hm, as_stream(stream));
}

extern "C" int ldmae_attention_fwd_qkv(int dtype, const void* qkv, void* o, float* lse, int B, int H, int N, int hd, float scale, void* stream) {
  LDMAE_REQUIRE(qkv && o && lse, "attention_fwd_qkv: null pointer");
  LDMAE_REQUIRE(dtype == LDMAE_BF16 || ((dtype == LDMAE_F32 || dtype == LDMAE_F16) && hd == 16),
                "attention_fwd_qkv: bf16, or f32 / fp16 at head_dim 16 (other f32 head dims take head-major q/k/v)");
  if (int e = attn_check("attention_fwd_qkv", dtype, B, H, N, hd)) return e;
  LDMAE_REQUIRE(hd % 8 == 0, "attention_fwd_qkv: head_dim %d must be a multiple of 8", hd);
  const long hw = (long)H * hd;
  const QkvLayout pk{(long)N * 3 * hw, (long)hd, 3 * hw};
  if (dtype == LDMAE_F32) {
    const float* pf = (const float*)qkv;
    return attention_fwd_core(dtype, pf, pf + hw, pf + 2 * hw, o, lse, B, H, N, hd, scale, pk, pk, as_stream(stream));
  }
  const bf16* p = (const bf16*)qkv;
  return attention_fwd_core(dtype, p, p + hw, p + 2 * hw, o, lse, B, H, N, hd, scale, pk, pk, as_stream(stream));
}

// max_j |k_j|^2 per (batch, head) of a packed token-major qkv [B*N][3][H][hd] (bf16) -> out [B*H][2] = (0, max): the input of
// ldmae_attention_fwd_qkv_bounded when nothing upstream knows the norms (one pass over the k slots; worth it for long sequences only).
__global__ __launch_bounds__(256) void k_norm_max_kernel(const bf16* __restrict__ qkv, unsigned* __restrict__ out, int N, int H, int hd) {
  __shared__ unsigned hmax[64];
  const int b = blockIdx.y, t0 = blockIdx.x * 64;
  if (threadIdx.x < 64) hmax[threadIdx.x] = 0u;
  __syncthreads();
  const int pairs = min(64, N - t0) * H;
  for (int p = threadIdx.x; p < pairs; p += 256) {
    const int tok = t0 + p / H, hh = p % H;
    const bf16* kr = qkv + (((size_t)b * N + tok) * 3 + 1) * H * hd + (size_t)hh * hd;
    float ss = 0.f;
    for (int ch = 0; ch < hd / 8; ++ch) {
      const bf16x8 v = *(const bf16x8*)(kr + ch * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += (float)v[j] * (float)v[j];
    }
    atomicMax(&hmax[hh], __float_as_uint(ss));           // non-negative floats order like their bits
  }
  __syncthreads();
  if (threadIdx.x < H) atomicMax(out + ((size_t)b * H + threadIdx.x) * 2 + 1, hmax[threadIdx.x]);
}
extern "C" int ldmae_k_norm_max(const void* qkv, float* qk_max2, int B, int N, int H, int hd, void* stream) {
  LDMAE_REQUIRE(qkv && qk_max2 && B > 0 && N > 0, "k_norm_max: null pointer or empty problem");
  LDMAE_REQUIRE(H >= 1 && H <= 64 && hd % 8 == 0, "k_norm_max: %d heads (1 .. 64) of %d (multiple of 8)", H, hd);
  // the atomic-max buffer MUST start at zero (a smaller-than-true maximum makes the static-shift softmax wrong, silently)
  LDMAE_REQUIRE(hipMemsetAsync(qk_max2, 0, (size_t)B * H * 2 * sizeof(float), as_stream(stream)) == hipSuccess, "k_norm_max: hipMemsetAsync of the norm buffer failed");
  hipLaunchKernelGGL(k_norm_max_kernel, dim3((N + 63) / 64, B), dim3(256), 0, as_stream(stream), (const bf16*)qkv, (unsigned*)qk_max2, N, H, hd);
  LDMAE_CHECK_LAUNCH("k_norm_max");
  return LDMAE_OK;
}

// ldmae_attention_fwd_qkv with the static softmax shift from per-(batch, head) norm maxima: qk_max2 [B*H][2] f32 on the device =
// (max_i |q_i|^2, max_j |k_j|^2) of the q / k rows as stored in the packed qkv.  Heads whose bound |q|max |k|max scale log2(e) * 1.02 exceeds 50
// keep the running maximum.  The maxima must be true maxima (or larger): a smaller value makes the result wrong.
extern "C" int ldmae_attention_fwd_qkv_bounded(int dtype, const void* qkv, void* o, float* lse, const float* qk_max2, int B, int H, int N, int hd,
                                               float scale, void* stream) {
  LDMAE_REQUIRE(qkv && o && lse && qk_max2, "attention_fwd_qkv_bounded: null pointer");
  LDMAE_REQUIRE(dtype == LDMAE_BF16, "attention_fwd_qkv_bounded: bf16 only");
  if (int e = attn_check("attention_fwd_qkv_bounded", dtype, B, H, N, hd)) return e;
  LDMAE_REQUIRE(hd % 8 == 0, "attention_fwd_qkv_bounded: head_dim %d must be a multiple of 8", hd);
  const bf16* p = (const bf16*)qkv;
  const long hw = (long)H * hd;
  const QkvLayout pk{(long)N * 3 * hw, (long)hd, 3 * hw};
  return attention_fwd_core(dtype, p, p + hw, p + 2 * hw, o, lse, B, H, N, hd, scale, pk, pk, as_stream(stream), qk_max2, 1);
}

// operand-fragment prefetch depth (template parameter PF of the backward kernels) of the plain (no QK-norm epilogue) pair per head dim, whole tiles
// only: 1 at head_dim 64 (-2.4 % as in the fused pair); head_dim 16 (the VMAE blocks' backward, bound by vector issue) measured NEUTRAL with it
// (1.225 vs 1.221 ms at 256 x 12 x 1024^2, profiles/r06_attn_prefetch_ab.txt) and keeps PF 0, like the head dims that were not measured.
// Diagnostic build: tune key 25 = 1 forces PF 1, 3 forces PF 0 (tools/bench_attn16.py --prefetch).
#define ATTN_PF_PLAIN(HD) ((HD) == 64 ? 1 : 0)
static int attention_bwd_core(int dtype, const void* q, const void* k, const void* v, const void* o, const void* do_, const float* lse,
                              void* dq, void* dk, void* dv, float* delta, int B, int H, int N, int hd, float scale, QkvLayout Lq, QkvLayout Lv, hipStream_t st) {
  const unsigned grid = (unsigned)B * H * ((N + 127) / 128);
  // delta = [2][B*H*NP] f32 workspace, NP = N rounded up to 64 (bf16: -delta | -lse*log2e, rows padded to whole tiles; f32: slot 0 = delta, unpadded)
  const long items = (long)B * N * H, rcs = (long)B * H * ((N + 63) / 64 * 64);
  const unsigned dgrid = (unsigned)((items * 8 + 255) / 256 < 8192 ? (items * 8 + 255) / 256 : 8192);
#ifdef LDMAE_DIAG
  if (hd == 16 && (dtype == LDMAE_BF16 || dtype == LDMAE_F16) && ldmae_tune_get(21) == 1 && N % 256 == 0 && N <= 1024) {
    // diagnostic A/B: the VMAE heads on whole 256-token blocks in one kernel, one workgroup per (batch, head) (attn_bwd_fused16_kernel); `delta` stays unused
    const int ksets = N % 512 == 0 ? 2 : 1, sweeps = N / (256 * ksets);
    const int dbg = ldmae_tune_get(22);      // ablations (timing only): 1 = no dQ slabs / reduce, 2 = no dS image and dQ product, 4 = no exponential, 8 = no accumulator-initialising LDS reads
    const int lds = ATT_STAGES * 2 * 64 * 32 * 2 + 8 * N + 8 * 2048 + 8 * 64 * 16 * 4 + (sweeps > 1 ? 64 * N : 0);
#define LF(KS, F) { hipFuncSetAttribute((const void*)attn_bwd_fused16_kernel<KS, F>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
    hipLaunchKernelGGL((attn_bwd_fused16_kernel<KS, F>), dim3((unsigned)B * H), dim3(512), lds, st, (const bf16*)q, (const bf16*)k, (const bf16*)v, (const bf16*)o, (const bf16*)do_, lse, (bf16*)dq, (bf16*)dk, (bf16*)dv, H, N, scale, Lq, Lv, dbg); }
    if (dtype == LDMAE_F16) { if (ksets == 2) LF(2, true) else LF(1, true) }
    else { if (ksets == 2) LF(2, false) else LF(1, false) }
#undef LF
    LDMAE_CHECK_LAUNCH("attention_bwd(fused, head_dim 16)");
    return LDMAE_OK;
  }
#endif
  if (dtype == LDMAE_F16) {
    // fp16 operands / gradients (VMAE pre-training under fp16 autocast; head_dim 16): the bf16 kernels with the f16 MFMAs and conversions
    LDMAE_REQUIRE(hd == 16, "attention_bwd(fp16): head_dim 16 only (the VMAE heads)");
#define LRHP(R, P) { \
    hipFuncSetAttribute((const void*)attn_bwd_dq_bf16_kernel<16, false, R, true, P>, hipFuncAttributeMaxDynamicSharedMemorySize, attn_lds(16, 0)); \
    hipFuncSetAttribute((const void*)attn_bwd_dkdv_bf16_kernel<16, false, R, true, P>, hipFuncAttributeMaxDynamicSharedMemorySize, attn_lds(16, 1024)); \
    hipLaunchKernelGGL((attn_bwd_dq_bf16_kernel<16, false, R, true, P>), dim3(grid), dim3(256), attn_lds(16, 0), st, (const bf16*)q, (const bf16*)k, (const bf16*)v, (const bf16*)o, (const bf16*)do_, lse, delta, rcs, (bf16*)dq, H, N, scale, Lq, Lv, QkNormBwd{}); \
    hipLaunchKernelGGL((attn_bwd_dkdv_bf16_kernel<16, false, R, true, P>), dim3(grid), dim3(256), attn_lds(16, 1024), st, (const bf16*)q, (const bf16*)k, (const bf16*)v, (const bf16*)do_, delta, rcs, (bf16*)dk, (bf16*)dv, H, N, scale, Lq, Lv, QkNormBwd{}); }
#ifdef LDMAE_DIAG
#define LRH(R) { const int k25 = ldmae_tune_get(25); if (k25 == 3) LRHP(R, 0) else if (k25 == 1) LRHP(R, 1) else LRHP(R, ATTN_PF_PLAIN(16)) }
#else
#define LRH(R) LRHP(R, ATTN_PF_PLAIN(16))
#endif
    if (N % 64 == 0) LRH(false) else LRHP(true, 0)
#undef LRH
#undef LRHP
  } else if (dtype == LDMAE_BF16) {
    // dQ first: it forms delta = rowsum(dO * O) from its own fragments and publishes it for the dK/dV kernel
#define LRP(HD, R, P) { \
    hipFuncSetAttribute((const void*)attn_bwd_dq_bf16_kernel<HD, false, R, false, P>, hipFuncAttributeMaxDynamicSharedMemorySize, attn_lds(HD, 0)); \
    hipFuncSetAttribute((const void*)attn_bwd_dkdv_bf16_kernel<HD, false, R, false, P>, hipFuncAttributeMaxDynamicSharedMemorySize, attn_lds(HD, 1024)); \
    hipLaunchKernelGGL((attn_bwd_dq_bf16_kernel<HD, false, R, false, P>), dim3(grid), dim3(256), attn_lds(HD, 0), st, (const bf16*)q, (const bf16*)k, (const bf16*)v, (const bf16*)o, (const bf16*)do_, lse, delta, rcs, (bf16*)dq, H, N, scale, Lq, Lv, QkNormBwd{}); \
    hipLaunchKernelGGL((attn_bwd_dkdv_bf16_kernel<HD, false, R, false, P>), dim3(grid), dim3(256), attn_lds(HD, 1024), st, (const bf16*)q, (const bf16*)k, (const bf16*)v, (const bf16*)do_, delta, rcs, (bf16*)dk, (bf16*)dv, H, N, scale, Lq, Lv, QkNormBwd{}); }
#ifdef LDMAE_DIAG
#define LR(HD, R) { const int k25 = ldmae_tune_get(25); if (k25 == 3) LRP(HD, R, 0) else if (k25 == 1) LRP(HD, R, 1) else LRP(HD, R, ATTN_PF_PLAIN(HD)) }
#else
#define LR(HD, R) LRP(HD, R, ATTN_PF_PLAIN(HD))
#endif
#define L(HD) if (N % 64 == 0) LR(HD, false) else LRP(HD, true, 0)
    ATTN_HD_DISPATCH(hd, L);
#undef L
#undef LR
#undef LRP
  } else {
    LDMAE_REQUIRE(Lq.ld == hd && Lq.sh == (long)N * hd && Lv.ld == hd && Lv.sh == (long)N * hd, "attention_bwd(f32): head-major q/k/v only");
    LDMAE_REQUIRE((size_t)(128 + 128 + 64 + 64) * (hd + 1) * 4 + 512 <= 160 * 1024,
                  "attention_bwd(f32): head_dim %d needs more than the 160 KiB of LDS (f32 backward: head dims up to 96; bf16 covers 128)", hd);
    hipLaunchKernelGGL(attn_delta_kernel<float>, dim3(dgrid), dim3(256), 0, st, (const float*)o, (const float*)do_, delta, B, H, N, hd);
#define L(HD) { const size_t l1 = (size_t)(128 + 128 + 64 + 64) * (HD + 1) * 4 + 512; \
    hipFuncSetAttribute((const void*)attn_bwd_dkdv_f32_kernel<HD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l1); \
    hipFuncSetAttribute((const void*)attn_bwd_dq_f32_kernel<HD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l1); \
    hipLaunchKernelGGL(attn_bwd_dkdv_f32_kernel<HD>, dim3(grid), dim3(256), l1, st, (const float*)q, (const float*)k, (const float*)v, (const float*)do_, lse, delta, (float*)dk, (float*)dv, H, N, scale); \
    hipLaunchKernelGGL(attn_bwd_dq_f32_kernel<HD>, dim3(grid), dim3(256), l1, st, (const float*)q, (const float*)k, (const float*)v, (const float*)do_, lse, delta, (float*)dq, H, N, scale); }
    ATTN_HD_DISPATCH_F32(hd, L);
#undef L
  }
  LDMAE_CHECK_LAUNCH("attention_bwd");
  return LDMAE_OK;
}

extern "C" int ldmae_attention_bwd(int dtype, const void* q, const void* k, const void* v, const void* o, const void* do_, const float* lse,
                                   void* dq, void* dk, void* dv, float* delta, int B, int H, int N, int hd, float scale, void* stream) {
  LDMAE_REQUIRE(q && k && v && o && do_ && lse && dq && dk && dv && delta, "attention_bwd: null pointer");
  if (int e = attn_check("attention_bwd", dtype, B, H, N, hd)) return e;
  LDMAE_REQUIRE(hd % 8 == 0, "attention_bwd: head_dim %d must be a multiple of 8", hd);
  const QkvLayout hm{(long)H * N * hd, (long)N * hd, (long)hd};
  return attention_bwd_core(dtype, q, k, v, o, do_, lse, dq, dk, dv, delta, B, H, N, hd, scale, hm, hm, as_stream(stream));
}

extern "C" int ldmae_attention_bwd_qkv(int dtype, const void* qkv, const void* o, const void* do_, const float* lse, void* dqkv, float* delta,
                                       int B, int H, int N, int hd, float scale, void* stream) {
  LDMAE_REQUIRE(qkv && o && do_ && lse && dqkv && delta, "attention_bwd_qkv: null pointer");
  LDMAE_REQUIRE(dtype == LDMAE_BF16 || (dtype == LDMAE_F16 && hd == 16), "attention_bwd_qkv: bf16, or fp16 at head_dim 16");
  if (int e = attn_check("attention_bwd_qkv", dtype, B, H, N, hd)) return e;
  LDMAE_REQUIRE(hd % 8 == 0, "attention_bwd_qkv: head_dim %d must be a multiple of 8", hd);
  const bf16* p = (const bf16*)qkv;
  bf16* g = (bf16*)dqkv;
  const long hw = (long)H * hd;
  const QkvLayout pk{(long)N * 3 * hw, (long)hd, 3 * hw};
  return attention_bwd_core(dtype, p, p + hw, p + 2 * hw, o, do_, lse, g, g + hw, g + 2 * hw, delta, B, H, N, hd, scale, pk, pk, as_stream(stream));
}

// q, k (dq, dk) head-major [B,H,N,hd]; v read from -- and dv written into -- the v slot of a packed token-major [B,N,3,H,hd] buffer
// (the LightningDiT block: QK-norm + RoPE produce new q / k, v is used exactly as the qkv Linear wrote it)
extern "C" int ldmae_attention_fwd_pv(int dtype, const void* q, const void* k, const void* qkv, void* o, float* lse, int B, int H, int N, int hd,
                                      float scale, void* stream) {
  LDMAE_REQUIRE(q && k && qkv && o && lse, "attention_fwd_pv: null pointer");
  LDMAE_REQUIRE(dtype == LDMAE_BF16, "attention_fwd_pv: bf16 only");
  if (int e = attn_check("attention_fwd_pv", dtype, B, H, N, hd)) return e;
  LDMAE_REQUIRE(hd % 8 == 0, "attention_fwd_pv: head_dim %d must be a multiple of 8", hd);
  const long hw = (long)H * hd;
  const QkvLayout hm{(long)H * N * hd, (long)N * hd, (long)hd}, pk{(long)N * 3 * hw, (long)hd, 3 * hw};
  return attention_fwd_core(dtype, q, k, (const bf16*)qkv + 2 * hw, o, lse, B, H, N, hd, scale, hm, pk, as_stream(stream));
}
// ldmae_attention_fwd_pv with a caller-supplied bound on |score * scale * log2(e)| (one float on the device; see the kernel): the softmax
// then runs with that static shift.  A bound that does not hold makes the result wrong (overflow), so it has to be a proven one.
extern "C" int ldmae_attention_fwd_pv_bounded(int dtype, const void* q, const void* k, const void* qkv, void* o, float* lse, const float* score_bound,
                                              int B, int H, int N, int hd, float scale, void* stream) {
  LDMAE_REQUIRE(q && k && qkv && o && lse && score_bound, "attention_fwd_pv_bounded: null pointer");
  LDMAE_REQUIRE(dtype == LDMAE_BF16, "attention_fwd_pv_bounded: bf16 only");
  if (int e = attn_check("attention_fwd_pv_bounded", dtype, B, H, N, hd)) return e;
  LDMAE_REQUIRE(hd % 8 == 0, "attention_fwd_pv_bounded: head_dim %d must be a multiple of 8", hd);
  const long hw = (long)H * hd;
  const QkvLayout hm{(long)H * N * hd, (long)N * hd, (long)hd}, pk{(long)N * 3 * hw, (long)hd, 3 * hw};
  return attention_fwd_core(dtype, q, k, (const bf16*)qkv + 2 * hw, o, lse, B, H, N, hd, scale, hm, pk, as_stream(stream), score_bound);
}
// The bound for heads that went through QK-RMSNorm + RoPE (lightningdit.py:66-80; elementwise.hip: q = x * rsqrt(mean(x^2) + eps) * w, then a
// rotation): |q|^2 = sum w_i^2 xhat_i^2 <= max|w|^2 * hd, the same for k, RoPE preserves norms, so
// |q . k| * scale * log2(e) <= hd * max|wq| * max|wk| * scale * log2(e); 2 % on top for the bf16 roundings of q, k and of the scaled q.
__global__ void qk_score_bound_kernel(const float* __restrict__ wq, const float* __restrict__ wk, int hd, float c, float* __restrict__ out) {
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < hd; i += 64) { a = fmaxf(a, fabsf(wq[i])); b = fmaxf(b, fabsf(wk[i])); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a = fmaxf(a, __shfl_xor(a, o, 64)); b = fmaxf(b, __shfl_xor(b, o, 64)); }
  if (threadIdx.x == 0) out[0] = (float)hd * a * b * c * 1.02f;
}
extern "C" int ldmae_qk_score_bound(const float* wq, const float* wk, int hd, float scale, float* out, void* stream) {
  LDMAE_REQUIRE(wq && wk && out && hd > 0, "qk_score_bound: null pointer or empty head");
  hipLaunchKernelGGL(qk_score_bound_kernel, dim3(1), dim3(64), 0, as_stream(stream), wq, wk, hd, scale * 1.4426950408889634f, out);
  LDMAE_CHECK_LAUNCH("qk_score_bound");
  return LDMAE_OK;
}
extern "C" int ldmae_attention_bwd_pv(int dtype, const void* q, const void* k, const void* qkv, const void* o, const void* do_, const float* lse,
                                      void* dq, void* dk, void* dqkv, float* delta, int B, int H, int N, int hd, float scale, void* stream) {
  LDMAE_REQUIRE(q && k && qkv && o && do_ && lse && dq && dk && dqkv && delta, "attention_bwd_pv: null pointer");
  LDMAE_REQUIRE(dtype == LDMAE_BF16, "attention_bwd_pv: bf16 only");
  if (int e = attn_check("attention_bwd_pv", dtype, B, H, N, hd)) return e;
  LDMAE_REQUIRE(hd % 8 == 0, "attention_bwd_pv: head_dim %d must be a multiple of 8", hd);
  const long hw = (long)H * hd;
  const QkvLayout hm{(long)H * N * hd, (long)N * hd, (long)hd}, pk{(long)N * 3 * hw, (long)hd, 3 * hw};
  return attention_bwd_core(dtype, q, k, (const bf16*)qkv + 2 * hw, o, do_, lse, dq, dk, (bf16*)dqkv + 2 * hw, delta, B, H, N, hd, scale, hm, pk,
                            as_stream(stream));
}

// ---- attention backward + QK-RMSNorm / RoPE backward in one (LightningDiT block, bf16): dq / dk never exist head-major; the packed dqkv
// comes out complete (q | k through the norm / rope backward in the epilogues, v as attention_bwd_pv), with the norm-weight gradients
// and the qkv bias gradient.  Replaces ldmae_attention_bwd_pv + ldmae_qknorm_rope_bwd for head dims 64 / 128.
extern "C" long ldmae_colsum_workspace_bytes(int M, int N);
extern "C" int ldmae_colsum(int dtype, const void* X, int ldx, int M, int N, float* out, float beta, float* workspace, void* stream);
// the diagnostic build's one-pass kernel: its shapes (f32 dQ scratch [B*H][N][hd] at the end of the workspace)
#ifdef LDMAE_DIAG
static bool attn_onepass_shape(int N, int hd) { return hd == 64 && N % 128 == 0; }
#else
static constexpr bool attn_onepass_shape(int, int) { return false; }
#endif
extern "C" long ldmae_attention_bwd_pv_qknorm_workspace_bytes(int B, int H, int N, int hd) {
  const long blk = (long)B * ((N + 127) / 128);
  const long cw = ldmae_colsum_workspace_bytes((int)(blk * H), 2 * hd), cb = ldmae_colsum_workspace_bytes((int)blk, 3 * H * hd);
  const long cmax = ((cw > cb ? cw : cb) + 15) / 16 * 16;
  return (2L * B * H * N + blk * H * 2 * hd + blk * 3 * H * hd + 2L * hd) * 4 + cmax + (attn_onepass_shape(N, hd) ? 4L * B * H * N * hd : 0);
}
extern "C" int ldmae_attention_bwd_pv_qknorm(int dtype, const void* q, const void* k, const void* qkv, const void* o, const void* do_,
                                             const float* lse, const float* wq, const float* wk, const float* cos, const float* sin, float eps,
                                             void* dqkv, float* dwq, float* dwk, float* dbias, float* workspace, int B, int H, int N, int hd,
                                             float scale, void* stream) {
  LDMAE_REQUIRE(q && k && qkv && o && do_ && lse && cos && sin && dqkv && dbias && workspace, "attention_bwd_pv_qknorm: null pointer");
  LDMAE_REQUIRE(!wq == !wk && (!wq || (dwq && dwk)), "attention_bwd_pv_qknorm: pass wq, wk, dwq, dwk (QK-norm + RoPE) or none of them (RoPE only: use_qknorm=False)");
  LDMAE_REQUIRE(dtype == LDMAE_BF16, "attention_bwd_pv_qknorm: bf16 only");
  LDMAE_REQUIRE(hd == 64 || hd == 128, "attention_bwd_pv_qknorm: head_dim %d (64 or 128; others: attention_bwd_pv + qknorm_rope_bwd)", hd);
  if (int e = attn_check("attention_bwd_pv_qknorm", dtype, B, H, N, hd)) return e;
  LDMAE_REQUIRE(N % 64 == 0, "attention_bwd_pv_qknorm: N=%d must be a multiple of 64 (ragged N: attention_bwd_pv + qknorm_rope_bwd)", N);
  hipStream_t st = as_stream(stream);
  const long hw = (long)H * hd, items = (long)B * H * N, blk = (long)B * ((N + 127) / 128);
  const QkvLayout hm{(long)H * N * hd, (long)N * hd, (long)hd}, pk{(long)N * 3 * hw, (long)hd, 3 * hw};
  float* rowc = workspace;
  float* Pw = rowc + 2 * items;
  float* Pb = Pw + blk * H * 2 * hd;
  float* dw2 = Pb + blk * 3 * H * hd;              // [2*hd] = dwq | dwk
  float* cws = dw2 + 2 * hd;
  const QkNormBwd qn{(const bf16*)qkv, wq, wk, cos, sin, (bf16*)dqkv, Pw, Pb, eps};
  const unsigned grid = (unsigned)B * H * ((N + 127) / 128);
  const bf16* v = (const bf16*)qkv + 2 * hw;
  bf16* dv = (bf16*)dqkv + 2 * hw;
#ifdef LDMAE_DIAG
  if (attn_onepass_shape(N, hd) && ldmae_tune_get(17) == 1) {
    const long cw = ldmae_colsum_workspace_bytes((int)(blk * H), 2 * hd), cb = ldmae_colsum_workspace_bytes((int)blk, 3 * H * hd);
    float* dqs = (float*)((char*)cws + ((cw > cb ? cw : cb) + 15) / 16 * 16);
    constexpr int LDS1 = ATT_STAGES * (2 * 64 * 64 * 2 + 1024) + 128 * 64 * 2;
    hipLaunchKernelGGL(attn_rowc_kernel<64>, dim3((unsigned)((items * 8 + 255) / 256)), dim3(256), 0, st, (const bf16*)o, (const bf16*)do_, lse, rowc, items, H, N, items);
#define LOP(D) { hipFuncSetAttribute((const void*)attn_bwd_onepass_bf16_kernel<64, D>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS1); \
    hipLaunchKernelGGL((attn_bwd_onepass_bf16_kernel<64, D>), dim3((unsigned)B * H), dim3(256), LDS1, st, (const bf16*)q, (const bf16*)k, v, (const bf16*)do_, rowc, items, \
                       dv, dqs, H, N, scale, hm, pk, qn); }
    switch (ldmae_tune_get(18)) { case 1: LOP(1); break; case 2: LOP(2); break; case 3: LOP(3); break; default: LOP(0); }
#undef LOP
    hipLaunchKernelGGL(attn_dq_finish_kernel<64>, dim3(grid), dim3(256), 0, st, (const float*)dqs, H, N, qn);
  } else
#endif
  // Operand-fragment prefetch depth of the fused backward pair (template parameter PF).  Shipped: 1 at head_dim 64 (the B/1 step; bitwise equal to
  // PF = 0 and 2-3 % faster: profiles/r06_attn_prefetch_ab.txt), 0 at head_dim 128 (one wave per SIMD either way, not measured).  Diagnostic build:
  // tune keys 23 (dK/dV) / 24 (dQ): 0 = the shipped depth, 1 / 2 = that depth, 3 = no prefetch (tools/bench_attn.py --prefetch).
#define ATTN_PF_DEFAULT(HD) ((HD) == 64 ? 1 : 0)
#define ATTN_DKDV_GO(HD, P) { hipFuncSetAttribute((const void*)attn_bwd_dkdv_bf16_kernel<HD, true, false, false, P>, hipFuncAttributeMaxDynamicSharedMemorySize, attn_lds(HD, 1024)); \
    hipLaunchKernelGGL((attn_bwd_dkdv_bf16_kernel<HD, true, false, false, P>), dim3(grid), dim3(256), attn_lds(HD, 1024), st, (const bf16*)q, (const bf16*)k, v, (const bf16*)do_, rowc, items, (bf16*)nullptr, dv, H, N, scale, hm, pk, qn); }
#define ATTN_DQ_GO(HD, P) { hipFuncSetAttribute((const void*)attn_bwd_dq_bf16_kernel<HD, true, false, false, P>, hipFuncAttributeMaxDynamicSharedMemorySize, attn_lds(HD, 0)); \
    hipLaunchKernelGGL((attn_bwd_dq_bf16_kernel<HD, true, false, false, P>), dim3(grid), dim3(256), attn_lds(HD, 0), st, (const bf16*)q, (const bf16*)k, v, (const bf16*)o, (const bf16*)do_, lse, rowc, items, (bf16*)nullptr, H, N, scale, hm, pk, qn); }
#ifdef LDMAE_DIAG
#define ATTN_DKDV_GO_PIPE(HD, P) { hipFuncSetAttribute((const void*)attn_bwd_dkdv_bf16_kernel<HD, true, false, false, P, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, attn_lds(HD, 1024)); \
    hipLaunchKernelGGL((attn_bwd_dkdv_bf16_kernel<HD, true, false, false, P, 1>), dim3(grid), dim3(256), attn_lds(HD, 1024), st, (const bf16*)q, (const bf16*)k, v, (const bf16*)do_, rowc, items, (bf16*)nullptr, dv, H, N, scale, hm, pk, qn); }
#define ATTN_DKDV_QKN(HD) { switch (ldmae_tune_get(23)) { case 1: ATTN_DKDV_GO(HD, 1) break; case 2: ATTN_DKDV_GO(HD, 2) break; case 3: ATTN_DKDV_GO(HD, 0) break; \
    case 4: ATTN_DKDV_GO_PIPE(HD, 1) break; case 5: ATTN_DKDV_GO_PIPE(HD, 2) break; default: ATTN_DKDV_GO(HD, ATTN_PF_DEFAULT(HD)) } }
#define ATTN_DQ_QKN(HD) { switch (ldmae_tune_get(24)) { case 1: ATTN_DQ_GO(HD, 1) break; case 2: ATTN_DQ_GO(HD, 2) break; case 3: ATTN_DQ_GO(HD, 0) break; default: ATTN_DQ_GO(HD, ATTN_PF_DEFAULT(HD)) } }
#else
#define ATTN_DKDV_QKN(HD) ATTN_DKDV_GO(HD, ATTN_PF_DEFAULT(HD))
#define ATTN_DQ_QKN(HD) ATTN_DQ_GO(HD, ATTN_PF_DEFAULT(HD))
#endif
#define L(HD) { \
    ATTN_DQ_QKN(HD) \
    ATTN_DKDV_QKN(HD) }
  if (hd == 64) L(64) else L(128)
#undef L
  LDMAE_CHECK_LAUNCH("attention_bwd_pv_qknorm");
  if (!wq) {}                   // RoPE only: no norm-weight gradients
  else if (dwk == dwq + hd) {        // the caller keeps dwq | dwk adjacent: reduce straight into them
    if (int e = ldmae_colsum(LDMAE_F32, Pw, 2 * hd, (int)(blk * H), 2 * hd, dwq, 0.f, cws, stream)) return e;
  } else {
    if (int e = ldmae_colsum(LDMAE_F32, Pw, 2 * hd, (int)(blk * H), 2 * hd, dw2, 0.f, cws, stream)) return e;
    hipMemcpyAsync(dwq, dw2, hd * sizeof(float), hipMemcpyDeviceToDevice, st);
    hipMemcpyAsync(dwk, dw2 + hd, hd * sizeof(float), hipMemcpyDeviceToDevice, st);
  }
  return ldmae_colsum(LDMAE_F32, Pb, 3 * (int)hw, (int)blk, 3 * (int)hw, dbias, 0.f, cws, stream);
}
