// MFMA GEMMs for the LightningDiT / VMAE linears (gfx950, wave64).
//
//   NT : C[M,N] = A[M,K] . B[N,K]^T   (both operands contraction-contiguous: Linear fwd,
//        and dX = dY . W with the [in,out] transposed weight copy)
//   TN : C[N,K] = A[M,N]^T . B[M,K]   (weight gradients; contraction over the token rows,
//        operands fetched K-major and transposed on the LDS read with ds_read_b64_tr_b16)
//
// bf16: mfma_f32_16x16x32_bf16, LDS ring filled by global_load_lds (16 B/lane, swizzle on
// the SOURCE address, linear LDS image), persistent 256x256-tile kernels.
// f32 : mfma_f32_16x16x4f32 (exact f32 FMA chain), register staged.  The f32 path exists
// for the fp32 parity contract (1e-4 vs the CPU oracle); bf16 is the throughput path.
#include "common.h"

#include "gemm_nt_common.h"

// bf16 or fp16 operand bits in the same bf16x8 registers (F16: VMAE pre-training under fp16 autocast, the LDMAE_F16 family)
template <bool F16> __device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// bf16 NT GEMM kernel: persistent, one 512-thread workgroup per CU walks the 256x256 output tiles t = b, b + grid, ...
// Main loop: the two M wave groups (which share the SIMDs pairwise) run half a K-step apart -- while one group issues its
// global_load_lds + fragment ds_reads, the other runs its 32 MFMAs; a barrier at every phase boundary is the metronome, and
// every wave waits for its own pieces of stage kt+1 one phase before anyone reads them.
// The first STAGES-1 K-steps of the NEXT tile are put in flight before the epilogue of the current one
// (its LDS strips live beside the ring, not in it), so the L2/HBM fill latency and the workgroup relaunch disappear behind
// the store tail, and no workgroup barrier separates main loop and epilogue: the wave group that finishes half a phase
// earlier starts storing while the other still multiplies.  `delay` (units of 8128 clocks) holds back every other workgroup of
// an XCD once, at start: a GEMM whose tiles all take the same time otherwise runs all 256 CUs in lockstep -- everyone
// loads, then everyone multiplies, then everyone stores 32 MiB into HBM at once.
// ------------------------------------------------------------------------------------------------
template <int STAGES, int EPI, typename OutT, bool TL = false>
__global__ __launch_bounds__(512) void gemm_nt_persist_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, int M, int N, int K,
                                                              int lda, int ldb, EpiArgs e, int ntiles, int delay,
                                                              unsigned long long* stamps) {
  constexpr int BM = 256, BN = 256, WM = 2, WN = 4, NW = 8, TM = 128, TNn = 64, MI = 8, NI = 4;
  constexpr int STAGE_BYTES = (BM + BN) * 64, PPW = (BM + BN) / 16 / NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const unsigned tiles_n = (N + BN - 1) / BN;
  // LDS map.  3-deep ring: the epilogue strips behind the ring (96 KiB ..), the SwiGLU-bwd column-sum scratch `ex` (1024 floats per wave) on
  // ring stage 2, which is free during an epilogue (the prologue of the next tile fills stages 0 and 1 only).  4-deep ring (128 KiB): the
  // strips (34 KiB) are ALIASED onto stages 2 and 3 -- free during an epilogue for the same reason: stage 2 of the next tile is requested
  // behind the tile-start barrier, which every wave reaches after its epilogue -- and `ex` sits behind the ring (.. 160 KiB).
  float* ew = (float*)(smem + (STAGES == 3 ? 3 : 2) * STAGE_BYTES) + wave * (16 * 68);
  float* ex = (float*)(smem + (STAGES == 3 ? 2 : 4) * STAGE_BYTES) + wave * 1024;
  static_assert(STAGE_BYTES >= 8 * 1024 * 4, "epilogue scratch does not fit a ring stage");
  // workgroup b sits on XCD b % 8 (256 workgroups, one per CU): give each XCD a contiguous run of 32 tiles per round so the
  // workgroups that share an A row-block (and the whole of B) share an L2
  // (gridDim.x == ntiles: one tile per workgroup, the classic launch -- the hardware then overlaps a finished workgroup's store
  // drain with the next workgroup's fill, which a persistent workgroup cannot do: vmcnt retires loads and stores in one order)
  const bool persistent = (int)gridDim.x != ntiles;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  // persistent: XCD x owns a contiguous range of A row-blocks for the whole launch and deals its tiles (all N tiles of a row-block
  // are consecutive) round-robin to its CUs: a row-block is never split across two L2s, and the next round finds it in L2
  const int rbx = (((M + BM - 1) / BM) + 7) / 8;
  const int first = persistent ? xcd * rbx * (int)tiles_n + slot : (int)xcd_remap(blockIdx.x, gridDim.x);
  const int tend = persistent ? min(ntiles, (xcd + 1) * rbx * (int)tiles_n) : ntiles;
  const int tstride = persistent ? per_xcd : ntiles;
  const int cgrp = delay >> 16;                 // (diagnostic knob, see set_tile)
  delay &= 0xffff;
  if (delay > 0 && (slot & 1) && persistent)
    for (int i = 0; i < delay; ++i) __builtin_amdgcn_s_sleep(127);

  const bf16* src[PPW];
  int m0 = 0, n0 = 0;
  // column-group walk (diagnostic A/B, tune key 16 = g, passed in the high bits of `delay`): an XCD takes its tiles g column tiles at a time
  // over ALL its row-blocks instead of all column tiles of one row-block after the other -- fewer B-tile refetches per round, more A refetches
  auto set_tile = [&](int t) {
    m0 = (t / tiles_n) * BM; n0 = (t % tiles_n) * BN;
    if (persistent && cgrp > 0 && tiles_n % cgrp == 0) {
      const int rb0 = xcd * rbx, rbe = min(rbx, (M + BM - 1) / BM - rb0), u = t - rb0 * (int)tiles_n;
      const int grp = u / (rbe * cgrp), v = u % (rbe * cgrp);
      m0 = (rb0 + v / cgrp) * BM; n0 = (grp * cgrp + v % cgrp) * BN;
    }
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int piece = wave * PPW + i;
      const int row = piece * 16 + (lane >> 2);
      const int c = (lane & 3) ^ ring_f((lane >> 4) & 3);
      int brow = n0 + row - BM;
      if constexpr (EPI == LDMAE_EPI_SWIGLU) {
        const int rl = row - BM;
        brow = ((rl & 32) ? (N >> 1) : 0) + (n0 >> 1) + (rl >> 6) * 32 + (rl & 31);
      }
      src[i] = row < BM ? A + (size_t)min(m0 + row, M - 1) * lda + c * 8 : B + (size_t)min(brow, N - 1) * ldb + c * 8;
    }
  };
  auto issue = [&](int kt) {
    char* base = smem + (kt % STAGES) * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < PPW; ++i)
        glds16(src[i] + kt * 32, base + (wave * PPW + i) * 1024);
  };
  constexpr int PL = PPW;      // all of a wave's pieces of stage kt+STAGES-1 are issued in its load phase
  const int fpos = ((lane >> 4) ^ ring_f((lane >> 2) & 3)) << 4;
  const int a_off = (wm * TM + (lane & 15)) * 64 + fpos, b_off = (BM + wn * TNn + (lane & 15)) * 64 + fpos;
  const int nk = K / 32;
  const bool grpB = wm >= WM / 2;
  static_assert(STAGES == 3 || STAGES == 4, "wait accounting below is written for a 3- or 4-deep ring");
  // own pieces of stage kt+1 landed; the stages requested after it (kt+2 .. kt+STAGES-1, as far as the tile has them) may stay in flight
  auto wait_next = [&](int kt, bool) {
    const int after = nk - kt - 2;                 // stages of this tile behind stage kt+1
    if (after >= STAGES - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * PPW) : "memory");
    else if (STAGES == 4 && after == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
    else if (after == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  constexpr int PRO = 2;            // K-steps put in flight ahead of a tile's epilogue (a 4-deep ring requests its third at the tile start)
  int t = first;
  if (t < tend) {
    set_tile(t);
#pragma unroll
    for (int s = 0; s < PRO; ++s)
      if (s < nk) issue(s);
  }
  int iter = 0;
  auto stamp = [&](int k) {             // diagnostic timeline (tools/gemm_timeline.py); stamps == nullptr in product runs
    if (stamps && lane == 0 && (wave & 3) == 0)
      stamps[(((size_t)iter * gridDim.x + blockIdx.x) * 2 + (wave >> 2)) * 4 + k] = __builtin_amdgcn_s_memrealtime();
  };
  // TL (diagnostic build): per-K-step phase stamps (shader clock) of the second tile, kept in LDS beside the strips
  unsigned long long* tl = (unsigned long long*)(smem + STAGES * STAGE_BYTES + 8 * 16 * 68 * 4);
  auto tstamp = [&](int kt, int k2) {
    if constexpr (TL) {
      if (iter == 1 && (wave & 3) == 0 && lane == 0 && kt < 32) tl[((wave >> 2) * 32 + kt) * 4 + k2] = __builtin_amdgcn_s_memtime();
    }
  };
  while (t < tend) {
    stamp(0);
    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // vmcnt counts in issue order, and the previous tile's epilogue accesses are younger than this tile's prologue loads:
    // the counted waits below can only over-wait (never under-wait) because of them
    {
    // Tile start: vmcnt(0) through the BUILTIN, so that the compiler's waitcnt pass sees it.  It then knows that no load of the
    // previous epilogue (bias, residual rows, h12) is pending when the K loop begins; otherwise it protects those registers,
    // which the loop reuses, with vmcnt(0) waits INSIDE the loop -- and with the ring DMA hidden in asm such a wait drains
    // the whole ring every K-step (seen: SwiGLU instantiations 35 % slower than the plain-bias one).  Cost: stage 1 must
    // have landed too, and the previous tile's stores are drained (~0.2 us per tile).
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_barrier();
    if constexpr (STAGES == 4) { if (PRO < nk) issue(PRO); }       // behind the barrier: every wave has left the strips that alias this stage
    if (grpB) __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
      tstamp(kt, 0);
      const bool more = kt + STAGES - 1 < nk;
      const char* st = smem + (kt % STAGES) * STAGE_BYTES;
      bf16x8 af[MI], bfr[NI];
      __builtin_amdgcn_s_setprio(1);      // load phase at raised priority
#pragma unroll
      for (int j = 0; j < NI; ++j) bfr[j] = *(const bf16x8*)(st + b_off + j * 1024);
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8*)(st + a_off + i * 1024);
      if (more) issue(kt + STAGES - 1);      // after the fragment reads: their LDS latency runs under the DMA issue (+0.5 %)
      if constexpr (TL) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      tstamp(kt, 1);
      __builtin_amdgcn_s_setprio(0);
      if (grpB) wait_next(kt, false);
      __builtin_amdgcn_s_barrier();
      tstamp(kt, 2);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      tstamp(kt, 3);
      if (!grpB) wait_next(kt, true);
      __builtin_amdgcn_s_barrier();
    }
    if (!grpB) __builtin_amdgcn_s_barrier();             // every wave is past its last fragment read: the ring is free
    }
    stamp(1);
    const int em0 = m0, en0 = n0;
    t += tstride;
    if (t < tend) {
      set_tile(t);
#pragma unroll
      for (int s = 0; s < PRO; ++s)
        if (s < nk) issue(s);
    }
    nt_epilogue<EPI, OutT, TM, TNn, MI, NI>(acc, ew, ex, e, em0, en0, wm, wn, lane, M, N);
    if constexpr (TL) {
      if (iter == 1 && stamps) {
        __syncthreads();
        if (tid < 256) stamps[(size_t)gridDim.x * 1024 + (size_t)blockIdx.x * 256 + tid] = tl[tid];
      }
    }
    stamp(2);
    if (stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(3); }
    ++iter;
  }
}

#ifdef LDMAE_DIAG
// launch-gap probes (tools/, diagnostic build): the persistent NT kernel's launch configuration and arguments around an empty body
__global__ __launch_bounds__(512) void gemm_nt_dummy_kernel(const bf16* A, const bf16* B, int M, int N, int K, int lda, int ldb, EpiArgs e, int ntiles,
                                                            int delay, unsigned long long* stamps) {
  if (M < 0) stamps[0] = (unsigned long long)(size_t)A + (size_t)B + N + K + lda + ldb + ntiles + delay + (size_t)e.C;
}
#endif
// ------------------------------------------------------------------------------------------------
// f32 NT GEMM: 64x64 tile, 256 threads (2x2 waves, 32x32 per wave), BK = 16, register staged.
// ------------------------------------------------------------------------------------------------
constexpr int F_BM = 64, F_BN = 64, F_BK = 16, F_LD = 20;   // LDS row stride (floats), 80 B keeps 16-B alignment

template <int EPI, typename OutT, int TW = 2>
__global__ __launch_bounds__(64 * TW * TW) void gemm_nt_f32_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                                  int M, int N, int K, int lda, int ldb, EpiArgs e) {
  // TW x TW waves of 32x32: TW = 2 -> 64x64 tile / 256 threads.  (TW = 1, 32x32 tiles of one wave, was tried for the small-M
  // adaLN GEMMs: 4x the workgroups but 2x slower -- a lone wave cannot overlap its loads with its MFMAs.)
  constexpr int BT = 32 * TW, NT = 64 * TW * TW, RPP = NT / 4, PASSES = BT / RPP;
  __shared__ __attribute__((aligned(16))) float As[2][BT * F_LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BT * F_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / TW, wn = wave % TW;
  const unsigned tiles_n = (N + BT - 1) / BT;
  const int m0 = (blockIdx.x / tiles_n) * BT, n0 = (blockIdx.x % tiles_n) * BT;
  const int lr = tid >> 2, lc = (tid & 3) * 4;
  const float* ap[PASSES];
  const float* bp[PASSES];
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    ap[p] = A + (size_t)min(m0 + lr + p * RPP, M - 1) * lda + lc;
    bp[p] = B + (size_t)min(n0 + lr + p * RPP, N - 1) * ldb + lc;
  }
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nk = K / F_BK;
  float4 ra[PASSES], rb[PASSES];
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    ra[p] = *(const float4*)ap[p]; rb[p] = *(const float4*)bp[p];
    *(float4*)&As[0][(lr + p * RPP) * F_LD + lc] = ra[p];
    *(float4*)&Bs[0][(lr + p * RPP) * F_LD + lc] = rb[p];
  }
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) {
#pragma unroll
      for (int p = 0; p < PASSES; ++p) { ra[p] = *(const float4*)(ap[p] + (kt + 1) * F_BK); rb[p] = *(const float4*)(bp[p] + (kt + 1) * F_BK); }
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = As[cur][(wm * 32 + i * 16 + (lane & 15)) * F_LD + kk * 4 + (lane >> 4)];
        bfr[i] = Bs[cur][(wn * 32 + i * 16 + (lane & 15)) * F_LD + kk * 4 + (lane >> 4)];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) {
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        *(float4*)&As[cur ^ 1][(lr + p * RPP) * F_LD + lc] = ra[p];
        *(float4*)&Bs[cur ^ 1][(lr + p * RPP) * F_LD + lc] = rb[p];
      }
    }
    __syncthreads();
    cur ^= 1;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        epi_store<EPI, OutT>(e, m0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r, n0 + wn * 32 + j * 16 + (lane & 15), M, N, acc[i][j][r]);
}

// ------------------------------------------------------------------------------------------------
// bf16 TN GEMM (weight gradient), fallback form for ragged M: P[s][N,K] = sum over rows m in split s of A[m,n] * B[m,k].
// 128(n) x 128(k) output tile, 256 threads (2x2 waves, 64x64 per wave), 64 rows of M per step.
// LDS image per operand: [64 m][128 cols] bf16, 256-B rows, 16-B chunk ch of row r stored at
// ch ^ sw(r), sw(r) = ((r&3)<<2) | ((r>>2)&3)  -> conflict-free ds_read_b64_tr_b16 (guide T10 (b)).
// ------------------------------------------------------------------------------------------------
constexpr int TN_BN = 128, TN_BK = 128, TN_BM = 64;
constexpr int TN_TILE_BYTES = TN_BM * 128 * 2;   // 16 KiB

__device__ __forceinline__ int tn_sw(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }
// token rows past the end of a split (M % 64 != 0: VMAE keeps int(L * (1 - mask_ratio)) tokens for any ratio) are fetched from these
// 16 zero bytes -- the LDS-DMA source address is per lane -- so they add nothing to the products
__device__ __attribute__((aligned(16))) unsigned g_tn_zero16[4] = {0u, 0u, 0u, 0u};

template <bool F16 = false>
__global__ __launch_bounds__(256) void gemm_tn_bf16_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B,
                                                           float* __restrict__ P, int M, int N, int K, int lda, int ldb,
                                                           int rows_per_split) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int tiles_k = (K + TN_BK - 1) / TN_BK, tiles_n = (N + TN_BN - 1) / TN_BN;
  const int tile = blockIdx.x % (tiles_k * tiles_n), split = blockIdx.x / (tiles_k * tiles_n);
  const int n0 = (tile / tiles_k) * TN_BN, k0 = (tile % tiles_k) * TN_BK;
  const int mbeg = split * rows_per_split, mend = min(M, mbeg + rows_per_split);
  // staging: per operand 16 pieces of 4 rows x 256 B; wave issues 4 + 4.  Column chunks beyond N / K
  // are clamped to the last valid chunk (their products land in output columns that are never stored).
  int arow[4], acol[4], bcol[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave * 4 + i) * 4 + (lane >> 4);
    const int ch = (lane & 15) ^ tn_sw(r);
    arow[i] = r;
    acol[i] = min(n0 + ch * 8, N - 8);
    bcol[i] = min(k0 + ch * 8, K - 8);
  }
  auto stage = [&](int buf, int mt) {
    char* base = smem + buf * 2 * TN_TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const size_t m = (size_t)(mt + arow[i]);
      const bool real = mt + arow[i] < mend;
      glds16(real ? (const void*)(A + m * lda + acol[i]) : (const void*)g_tn_zero16, base + (wave * 4 + i) * 1024);
      glds16(real ? (const void*)(B + m * ldb + bcol[i]) : (const void*)g_tn_zero16, base + TN_TILE_BYTES + (wave * 4 + i) * 1024);
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // transposed fragment read: lane (g = l>>4, i = l&15, q = i>>2, p = i&3) supplies row 8g+q (+4), cols c0+4p..+3
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  auto frag = [&](const char* tile_base, int mstep, int c0) -> bf16x8 {
    const int r0 = mstep * 32 + 8 * g + q, r1 = r0 + 4;
    const int ch = (c0 >> 3) + (p >> 1);
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile_base + r0 * 256 + ((ch ^ tn_sw(r0)) << 4) + ((p & 1) << 3)));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile_base + r1 * 256 + ((ch ^ tn_sw(r1)) << 4) + ((p & 1) << 3)));
    union { bf16x8 v; s16x4 h[2]; } u;
    u.h[0] = lo; u.h[1] = hi;
    return u.v;
  };

  const int nsteps = (mend - mbeg + TN_BM - 1) / TN_BM;
  // rows past mend inside the last step contribute zero (zero DMA source above); rows_per_split % 64 == 0 (host)
  if (nsteps > 0) {
    stage(0, mbeg);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  int cur = 0;
  for (int st = 0; st < nsteps; ++st) {
    if (st + 1 < nsteps) stage(cur ^ 1, mbeg + (st + 1) * TN_BM);
    const char* ta = smem + cur * 2 * TN_TILE_BYTES;
    const char* tb = ta + TN_TILE_BYTES;
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = frag(ta, ms, wn * 64 + i * 16);
        bfr[i] = frag(tb, ms, wk * 64 + i * 16);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<F16>(af[i], bfr[j], acc[i][j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }
  float* out = P + (size_t)split * N * K;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 64 + i * 16 + (lane >> 4) * 4 + r, k = k0 + wk * 64 + j * 16 + (lane & 15);
        if (n < N && k < K) out[(size_t)n * K + k] = acc[i][j][r];
      }
}

// ------------------------------------------------------------------------------------------------
// bf16 TN GEMM, ring-pipelined (the product kernel; the 128x128 one above is the fallback for M % 32 != 0): 256(n) x 256(k)
// output tile, WNn x 4 waves (default 2 x 4: 128 x 64 per wave), 32 token rows per step, STAGES LDS buffers in flight across the
// barriers (same protocol as gemm_nt_persist_kernel; STAG = the two wave groups run half a step apart).
// Stage image: A rows then B rows, [32 m][256 cols] bf16 = 512-B rows; 16-B chunk ch of row r sits at
// ch ^ sw(r), sw(r) = ((r&3)<<1) | (((r>>3)&1)<<3): the 32 lanes of a ds_read_b64_tr_b16 half hit 32 distinct
// 8-B bank pairs.  Optional fused bias gradient: waves of the k-tile-0 column also run one MFMA per A fragment
// against an all-ones B operand (column sums of A on the matrix core, no extra pass over dY).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int tnr_sw(int r) { return ((r & 3) << 1) | (((r >> 3) & 1) << 3); }

template <int STAGES, int WNn = 2, bool STAG = false, bool F16 = false>
__global__ __launch_bounds__(WNn * 4 * 64) void gemm_tn_ring_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, float* __restrict__ P,
                                                           float* __restrict__ Pb, int M, int N, int K, int lda, int ldb,
                                                           int rows_per_split) {
  constexpr int BNn = 256, BKk = 256, WKk = 4, NW = WNn * WKk, MI = 256 / WNn / 16, NI = 4, STEP = 32;
  constexpr int OP_BYTES = STEP * 512, STAGE_BYTES = 2 * OP_BYTES, PPW = (2 * OP_BYTES / 1024) / NW;   // 32 pieces over the waves
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave / WKk, wk = wave % WKk;
  const int tiles_k = (K + BKk - 1) / BKk, tiles_n = (N + BNn - 1) / BNn;
  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile = lid % (tiles_k * tiles_n), split = lid / (tiles_k * tiles_n);
  const int n0 = (tile / tiles_k) * BNn, k0 = (tile % tiles_k) * BKk;
  const int mbeg = split * rows_per_split, mend = min(M, mbeg + rows_per_split);
  const int nsteps = (mend - mbeg) / STEP;               // host guarantees multiples of 32
  const bool want_bias = Pb != nullptr && k0 == 0 && wk == 0;

  const bf16* src[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int piece = wave * PPW + i;                    // 0..15 A, 16..31 B ; piece = 2 rows x 512 B
    const bool isB = piece >= 16;
    const int row = (piece & 15) * 2 + (lane >> 5);
    const int ch = (lane & 31) ^ tnr_sw(row);
    const int col = isB ? min(k0 + ch * 8, K - 8) : min(n0 + ch * 8, N - 8);
    src[i] = (isB ? B + (size_t)row * ldb : A + (size_t)row * lda) + col;
  }
  const size_t astep = (size_t)STEP * lda, bstep = (size_t)STEP * ldb;
  auto issue = [&](int st) {
    char* base = smem + (st % STAGES) * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const bool isB = (wave * PPW + i) >= 16;
      glds16(src[i] + (size_t)mbeg * (isB ? ldb : lda) + st * (isB ? bstep : astep), base + (wave * PPW + i) * 1024);
    }
  };
  f32x4 acc[MI][NI], accb[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // transposed fragment: rows 8g+q (+4), 16 columns starting at c0
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int r0 = 8 * g + q, r1 = r0 + 4;
  auto frag = [&](const char* op, int c0) -> bf16x8 {
    const int ch = (c0 >> 3) + (p >> 1);
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, op + r0 * 512 + ((ch ^ tnr_sw(r0)) << 4) + ((p & 1) << 3)));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, op + r1 * 512 + ((ch ^ tnr_sw(r1)) << 4) + ((p & 1) << 3)));
    union { bf16x8 v; s16x4 h[2]; } u;
    u.h[0] = lo; u.h[1] = hi;
    return u.v;
  };
  bf16x8 ones;
  if constexpr (F16) { f16x8 o; for (int j = 0; j < 8; ++j) o[j] = (f16)1.0f; ones = __builtin_bit_cast(bf16x8, o); }
  else {
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16)1.0f;
  }

#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nsteps) issue(s);
  if constexpr (STAG) {
    // two wave groups half a step apart (as in gemm_nt_persist_kernel): loads + transposed fragment reads of one group
    // run beside the MFMAs of the other on every SIMD
    const bool grpB = wn >= WNn / 2;
    auto wait_next = [&](int st) {
      const int ahead = min(STAGES - 2, nsteps - 2 - st);
      if (ahead >= 2) { if constexpr (STAGES >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory"); }
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
      else if (ahead == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    wait_next(-1);
    __builtin_amdgcn_s_barrier();
    if (grpB) __builtin_amdgcn_s_barrier();
    for (int st = 0; st < nsteps; ++st) {
      if (st + STAGES - 1 < nsteps) issue(st + STAGES - 1);
      const char* ta = smem + (st % STAGES) * STAGE_BYTES;
      const char* tb = ta + OP_BYTES;
      bf16x8 af[MI], bfr[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) bfr[j] = frag(tb, wk * 64 + j * 16);
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = frag(ta, wn * (MI * 16) + i * 16);
      if (grpB) wait_next(st);
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = mfma16<F16>(af[i], bfr[j], acc[i][j]);
      if (want_bias) {
#pragma unroll
        for (int i = 0; i < MI; ++i) accb[i] = mfma16<F16>(af[i], ones, accb[i]);
      }
      if (!grpB) wait_next(st);
      __builtin_amdgcn_s_barrier();
    }
    if (!grpB) __builtin_amdgcn_s_barrier();
  } else {
  for (int st = 0; st < nsteps; ++st) {
    const int ahead = min(STAGES - 2, nsteps - 1 - st);
    if (ahead >= 3) { if constexpr (STAGES >= 5) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PPW) : "memory"); }
    else if (ahead == 2) { if constexpr (STAGES >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory"); }
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (st + STAGES - 1 < nsteps) issue(st + STAGES - 1);
    const char* ta = smem + (st % STAGES) * STAGE_BYTES;
    const char* tb = ta + OP_BYTES;
    bf16x8 af[MI], bfr[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) bfr[j] = frag(tb, wk * 64 + j * 16);
#pragma unroll
    for (int i = 0; i < MI; ++i) af[i] = frag(ta, wn * (MI * 16) + i * 16);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = mfma16<F16>(af[i], bfr[j], acc[i][j]);
    if (want_bias) {
#pragma unroll
      for (int i = 0; i < MI; ++i) accb[i] = mfma16<F16>(af[i], ones, accb[i]);
    }
    __builtin_amdgcn_s_setprio(0);
  }
  }
  // ---- epilogue: f32 partial tile through a per-wave LDS tile for row-contiguous stores (8 waves at a time: 139 KiB)
  __syncthreads();
  constexpr int ELD = 68;
  float* ew = (float*)smem + (wave & 7) * (64 * ELD);
  float* out = P + (size_t)split * N * K;
#pragma unroll
  for (int grp = 0; grp < NW / 8; ++grp) {
    if ((wave >> 3) == grp) {
#pragma unroll
      for (int half = 0; half < MI / 4; ++half) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) ew[(i * 16 + (lane >> 4) * 4 + r) * ELD + j * 16 + (lane & 15)] = acc[half * 4 + i][j][r];
        const int nb = n0 + wn * (MI * 16) + half * 64, kb = k0 + wk * 64, col = (lane & 15) * 4;
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
          const int row = it * 4 + (lane >> 4);
          const float4 v = *(const float4*)(ew + row * ELD + col);
          if (nb + row < N && kb + col < K) *(float4*)(out + (size_t)(nb + row) * K + kb + col) = v;   // K % 4 == 0 (host check)
        }
      }
    }
    if (grp + 1 < NW / 8) __syncthreads();
  }
  if (want_bias && (lane & 15) == 0) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * (MI * 16) + i * 16 + (lane >> 4) * 4 + r;
        if (n < N) Pb[(size_t)split * N + n] = accb[i][r];
      }
  }
}

// f32 TN: 64(n) x 64(k) tile, 16 rows of M per step; A[l&15][k=l>>4] fragments are single floats
constexpr int FT_BN = 64, FT_BK = 64, FT_BM = 16, FT_LD = 68;

__global__ __launch_bounds__(256) void gemm_tn_f32_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                          float* __restrict__ P, int M, int N, int K, int lda, int ldb,
                                                          int rows_per_split) {
  __shared__ __attribute__((aligned(16))) float As[2][FT_BM * FT_LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][FT_BM * FT_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int tiles_k = (K + FT_BK - 1) / FT_BK, tiles_n = (N + FT_BN - 1) / FT_BN;
  const int tile = blockIdx.x % (tiles_k * tiles_n), split = blockIdx.x / (tiles_k * tiles_n);
  const int n0 = (tile / tiles_k) * FT_BN, k0 = (tile % tiles_k) * FT_BK;
  const int mbeg = split * rows_per_split, mend = min(M, mbeg + rows_per_split);
  const int lr = tid >> 4, lc = (tid & 15) * 4;      // 16 rows x 16 float4
  const int an = min(n0 + lc, N - 4), bk = min(k0 + lc, K - 4);
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto ld = [&](int mt, float4& ra, float4& rb) {
    const int m = mt + lr;
    if (m < mend) { ra = *(const float4*)(A + (size_t)m * lda + an); rb = *(const float4*)(B + (size_t)m * ldb + bk); }
    else { ra = make_float4(0, 0, 0, 0); rb = ra; }
  };
  const int nsteps = (mend - mbeg + FT_BM - 1) / FT_BM;
  float4 ra, rb;
  if (nsteps > 0) {
    ld(mbeg, ra, rb);
    *(float4*)&As[0][lr * FT_LD + lc] = ra;
    *(float4*)&Bs[0][lr * FT_LD + lc] = rb;
  }
  __syncthreads();
  int cur = 0;
  for (int st = 0; st < nsteps; ++st) {
    if (st + 1 < nsteps) ld(mbeg + (st + 1) * FT_BM, ra, rb);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = As[cur][(kk * 4 + (lane >> 4)) * FT_LD + wn * 32 + i * 16 + (lane & 15)];
        bfr[i] = Bs[cur][(kk * 4 + (lane >> 4)) * FT_LD + wk * 32 + i * 16 + (lane & 15)];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (st + 1 < nsteps) {
      *(float4*)&As[cur ^ 1][lr * FT_LD + lc] = ra;
      *(float4*)&Bs[cur ^ 1][lr * FT_LD + lc] = rb;
    }
    __syncthreads();
    cur ^= 1;
  }
  float* out = P + (size_t)split * N * K;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 32 + i * 16 + (lane >> 4) * 4 + r, k = k0 + wk * 32 + j * 16 + (lane & 15);
        if (n < N && k < K) out[(size_t)n * K + k] = acc[i][j][r];
      }
}

// sum the split-K partial slabs in fixed order (deterministic): out = beta*out + sum_s P[s].  `bid` of `nblk` workgroups of 256 threads.
__device__ __forceinline__ void splitk_reduce_threads(const float* __restrict__ P, float* __restrict__ out, long n, int splits, float beta,
                                                      unsigned bid, unsigned nblk) {
  for (long i = ((long)bid * 256 + threadIdx.x) * 4; i < n; i += (long)nblk * 256 * 4) {
    float4 s = *(const float4*)(P + i);
    for (int k = 1; k < splits; ++k) {
      float4 t = *(const float4*)(P + (size_t)k * n + i);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    if (beta != 0.f) {
      float4 o = *(const float4*)(out + i);
      s.x += beta * o.x; s.y += beta * o.y; s.z += beta * o.z; s.w += beta * o.w;
    }
    *(float4*)(out + i) = s;
  }
}
// the same sum for MANY splits of a SMALL output (VMAE weight gradients: 192 x 192 outputs from up to 256 row splits): one wave per 4
// adjacent outputs, lane l sums splits l, l + 64, ... and a butterfly folds the lane sums -- fixed order, so still deterministic; the
// per-thread loop above left 36 workgroups walking 100 partial slabs one after the other (24 us, 197 times per VMAE step)
__device__ __forceinline__ void splitk_reduce_waves(const float* __restrict__ P, float* __restrict__ out, long n, int splits, float beta, unsigned bid) {
  const int lane = threadIdx.x & 63;
  const long i = ((long)bid * 4 + (threadIdx.x >> 6)) * 4;
  if (i >= n) return;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = lane; k < splits; k += 64) {
    const float4 t = *(const float4*)(P + (size_t)k * n + i);
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
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ P, float* __restrict__ out, long n, int splits, float beta) {
  splitk_reduce_threads(P, out, n, splits, beta, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(256) void splitk_reduce_wave_kernel(const float* __restrict__ P, float* __restrict__ out, long n, int splits, float beta) {
  splitk_reduce_waves(P, out, n, splits, beta, blockIdx.x);
}
// weight gradient AND bias gradient of one TN GEMM in one launch (they were two: ~200 launches of ~10 us per VMAE pre-training step): the
// first g0 workgroups sum the first output, the rest the second, each in the form (w0 / w1: wave form) it would have had alone -- same bits
struct ReducePart { const float* P; float* out; long n; int wave; };
__global__ __launch_bounds__(256) void splitk_reduce_pair_kernel(ReducePart a, ReducePart b, unsigned g0, int splits, float beta) {
  const bool first = blockIdx.x < g0;
  const ReducePart& p = first ? a : b;
  const unsigned bid = first ? blockIdx.x : blockIdx.x - g0, nblk = first ? g0 : gridDim.x - g0;
  if (p.wave) splitk_reduce_waves(p.P, p.out, p.n, splits, beta, bid);
  else splitk_reduce_threads(p.P, p.out, p.n, splits, beta, bid, nblk);
}
// (wave form for many splits only: with 28 splits of a 768 x 768 output it leaves half its lanes idle on 147K waves -- 40 us against 12)
static bool reduce_wave_form(long n, int splits) { return splits >= 32 && n % 4 == 0 && n <= (1L << 18); }
static unsigned reduce_grid(long n, int splits) {
  return reduce_wave_form(n, splits) ? (unsigned)((n / 4 + 3) / 4) : (unsigned)((n / 4 + 255) / 256 < 4096 ? (n / 4 + 255) / 256 : 4096);
}
static void splitk_reduce(const float* P, float* out, long n, int splits, float beta, hipStream_t st) {
  if (reduce_wave_form(n, splits)) hipLaunchKernelGGL(splitk_reduce_wave_kernel, dim3(reduce_grid(n, splits)), dim3(256), 0, st, P, out, n, splits, beta);
  else hipLaunchKernelGGL(splitk_reduce_kernel, dim3(reduce_grid(n, splits)), dim3(256), 0, st, P, out, n, splits, beta);
}
static void splitk_reduce_pair(const float* P, float* out, long n, const float* Pb, float* outb, long nb, int splits, float beta, hipStream_t st) {
  const unsigned g0 = reduce_grid(n, splits), g1 = reduce_grid(nb, splits);
  hipLaunchKernelGGL(splitk_reduce_pair_kernel, dim3(g0 + g1), dim3(256), 0, st, ReducePart{P, out, n, reduce_wave_form(n, splits) ? 1 : 0},
                     ReducePart{Pb, outb, nb, reduce_wave_form(nb, splits) ? 1 : 0}, g0, splits, beta);
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
#ifdef LDMAE_DIAG
static void* g_nt_stamps = nullptr;
// diagnostic build only: device buffer that receives s_memrealtime stamps of the persistent NT kernel (NULL = off)
extern "C" void ldmae_debug_nt_stamps(void* buf) { g_nt_stamps = buf; }
#else
static constexpr void* g_nt_stamps = nullptr;
#endif
template <typename OutT>
static int launch_nt(int dtype, int epi, bool tile_launch, bool half_lines, const void* A, const void* B, int M, int N, int K, int lda, int ldb, const EpiArgs& e,
                     hipStream_t st) {
  const long pi = (ldmae_prof_is_on() && dtype == LDMAE_BF16) ? ldmae_prof_begin(st, 2.0 * M * N * K) : -1;
  // per call and per device (a process may drive several GPUs; the query is a cached driver attribute, ~100 ns)
  int ncu = 0;
  {
    int dev = 0, n = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    ncu = n >= 8 ? n / 8 * 8 : 8;
  }
  const int ntiles = cdiv(M, 256) * cdiv(N, 256);
#ifdef LDMAE_DIAG
  // tune key 11 (launch-gap probes): 1 = an empty kernel with the same launch configuration, 2 = the real kernel with ntiles = 0 (every
  // workgroup returns at once)
#define NT_DUMMY_BRANCHES(E)                                                                                                       \
    else if (ldmae_tune_get(11) == 1) { hipFuncSetAttribute((const void*)gemm_nt_dummy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds + 2048); hipLaunchKernelGGL(gemm_nt_dummy_kernel, dim3(pgrid), dim3(512), lds + 2048, st, (const bf16*)A, (const bf16*)B, M, N, K, lda, ldb, e, ntiles, 0, (unsigned long long*)nullptr); } \
    else if (ldmae_tune_get(11) == 2) { PERS_ATTR(3, E, OutT); hipLaunchKernelGGL((gemm_nt_persist_kernel<3, E, OutT>), dim3(pgrid), dim3(512), lds + 2048, st, (const bf16*)A, (const bf16*)B, M, N, K, lda, ldb, e, 0, 0, (unsigned long long*)nullptr); }
#else
#define NT_DUMMY_BRANCHES(E)
#endif
  // launch mode, a per-call argument (LDMAE_EPI_TILE_LAUNCH or'ed into `epi` by the caller): one 256x256 tile per workgroup instead of
  // one persistent workgroup per CU -- what a data-parallel caller asks for while RCCL's collective kernels hold some CUs
  // the persistent mapping gives every XCD a contiguous range of A row-blocks: with fewer than 8 row-blocks (the batched adaLN GEMM: M = batch
  // = 256, N = depth * 6D) whole XCDs would idle, so such shapes take the tile launch, which deals tiles to all XCDs
  const bool pers = !tile_launch && ldmae_tune_get(8) != 2 && cdiv(M, 256) >= 8;
  const int pgrid = (pers && ntiles != ncu) ? ncu : ntiles;
#ifdef LDMAE_DIAG
  // tune key 0: 1 / 2 = the experimental kernels of probe/gemm_w4.hip (diagnostic build only)
  if (dtype == LDMAE_BF16 && ldmae_tune_get(0) != 0 &&
      (ldmae_tune_get(0) == 1 ? ldmae_launch_nt_w4(epi, sizeof(OutT) == 2, A, B, M, N, K, lda, ldb, e, pgrid, ntiles, st)
                                : ldmae_launch_nt_p8(epi, sizeof(OutT) == 2, A, B, M, N, K, lda, ldb, e, pgrid, ntiles, st))) {
    if (pi >= 0) ldmae_prof_end(pi, st);
    LDMAE_CHECK_LAUNCH("gemm_nt_w4");
    return LDMAE_OK;
  }
#endif
#ifdef LDMAE_DIAG
  // tune key 12 = 1 (diagnostic build only): the fused epilogue's elementwise part runs inside the next tile's main loop
  // (probe/gemm_nt_defer.hip).  Built, bitwise equal, and 5-9 % SLOWER than the fused epilogues (profiles/r04_defer_ab.txt).
  if (dtype == LDMAE_BF16 && sizeof(OutT) == 2 && pers && pgrid != ntiles && (epi == LDMAE_EPI_GATE_RES || epi == LDMAE_EPI_SWIGLU) &&
      ldmae_tune_get(12) == 1 && ldmae_launch_nt_defer(epi, A, B, M, N, K, lda, ldb, e, pgrid, ntiles, st)) {
    if (pi >= 0) ldmae_prof_end(pi, st);
    LDMAE_CHECK_LAUNCH("gemm_nt_defer");
    return LDMAE_OK;
  }
#endif
#ifdef LDMAE_DIAG
  // tune key 15 = 1 (diagnostic build only): whole-line ring DMA (probe/gemm_nt_wl.hip: 128-B LDS rows, pieces of 8 rows x 128 B).  Built,
  // bitwise equal on every epilogue, 6.5 % SLOWER over the block's eight GEMMs (profiles/r04_wl_ab.txt): the CU's DMA path is limited in
  // BYTES per cycle, not in line requests.
  if (ldmae_tune_get(15) == 1 && dtype == LDMAE_BF16 && sizeof(OutT) == 2 && M % 8 == 0 && N % 8 == 0 && M >= 8 && N >= 8 && K % 64 == 0 && lda % 64 == 0 &&
      ldb % 64 == 0 && ((uintptr_t)A & 127) == 0 && ((uintptr_t)B & 127) == 0 && (epi != LDMAE_EPI_SWIGLU || N % 256 == 0) &&
      ldmae_launch_nt_wl(epi, A, B, M, N, K, lda, ldb, e, pgrid, ntiles, st)) {
    if (pi >= 0) ldmae_prof_end(pi, st);
    LDMAE_CHECK_LAUNCH("gemm_nt_wl");
    return LDMAE_OK;
  }
#endif
  // bf16, whole-line form (gemm_nt_lines.hip; round 5): every shape whose rows start on 128-B lines.  Diagnostic tune key 19 = 1 keeps the
  // half-line kernel below for A/B runs.
  if (dtype == LDMAE_BF16 && !half_lines && ldmae_tune_get(19) == 0 && ldmae_tune_get(7) == 0 && ldmae_tune_get(11) == 0 && ldmae_tune_get(14) == 0 &&
      ldmae_launch_nt_lines(epi, sizeof(OutT) == 2, A, B, M, N, K, lda, ldb, e, pgrid, ntiles, st)) {
    if (pi >= 0) ldmae_prof_end(pi, st);
    LDMAE_CHECK_LAUNCH("gemm_nt_lines");
    return LDMAE_OK;
  }
  // bf16: the persistent ring kernel (one workgroup per CU).  tune key 5 = start delay of every other workgroup (A/B knob),
  // key 7 = diagnostic per-K-step stamp build.
#define PERS_ATTR(...) hipFuncSetAttribute((const void*)gemm_nt_persist_kernel<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, lds + 2048)
#define PERS_GO(...)                                                                                                             \
  hipLaunchKernelGGL((gemm_nt_persist_kernel<__VA_ARGS__>), dim3(pgrid), dim3(512), lds + 2048, st, (const bf16*)A, (const bf16*)B, \
                     M, N, K, lda, ldb, e, ntiles, ldmae_tune_get(5) | (ldmae_tune_get(16) << 16), (unsigned long long*)g_nt_stamps)
  // tune key 14 = 1 (diagnostic build only): 4-deep ring, 128 KiB, strips aliased onto it, the SwiGLU-bwd scratch behind it (160 KiB in all).
  // Measured neutral (profiles/r04_ring4_ab.txt: block total 8.158 -> 8.118 ms): the L2 read latency seen by the CU is ~360 cycles
  // (TCP_TCC_READ_REQ_LATENCY / READ_REQ, profiles/r04_pmc_ta.md), far inside what two stages in flight cover.
#ifdef LDMAE_DIAG
#define PERS_RING4(E)                                                                                                             \
  if (ldmae_tune_get(14) == 1) {                                                                                                  \
    constexpr int lds4 = 4 * 512 * 64 + (E == LDMAE_EPI_SWIGLU_BWD ? 8 * 1024 * 4 : 0);                                            \
    hipFuncSetAttribute((const void*)gemm_nt_persist_kernel<4, E, OutT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds4);       \
    hipLaunchKernelGGL((gemm_nt_persist_kernel<4, E, OutT>), dim3(pgrid), dim3(512), lds4, st, (const bf16*)A, (const bf16*)B,      \
                       M, N, K, lda, ldb, e, ntiles, ldmae_tune_get(5), (unsigned long long*)g_nt_stamps);                       \
  } else
#else
#define PERS_RING4(E)
#endif
#define PERS(E)                                                                                                                  \
  PERS_RING4(E)                                                                                                                   \
  {                                                                                                                               \
    constexpr int lds = 3 * 512 * 64 + 8 * 16 * 68 * 4;                                                                           \
    /* the LDS opt-in is per device and cheap: set it at every launch (no process-wide "done" flag) */                            \
    if (E == LDMAE_EPI_BIAS && ldmae_tune_get(7) == 1) { PERS_ATTR(3, LDMAE_EPI_BIAS, OutT, true); PERS_GO(3, LDMAE_EPI_BIAS, OutT, true); }     \
    else if (E == LDMAE_EPI_SWIGLU && ldmae_tune_get(7) == 1) { PERS_ATTR(3, LDMAE_EPI_SWIGLU, OutT, true); PERS_GO(3, LDMAE_EPI_SWIGLU, OutT, true); } \
    NT_DUMMY_BRANCHES(E)                                                                                                          \
    else { PERS_ATTR(3, E, OutT); PERS_GO(3, E, OutT); }                                                                          \
  }
#define NT_LAUNCH(E)                                                                                                             \
  if (dtype == LDMAE_BF16) PERS(E)                                                                                                \
  else                                                                                                                           \
    hipLaunchKernelGGL((gemm_nt_f32_kernel<E, OutT>), dim3(cdiv(M, F_BM) * cdiv(N, F_BN)), dim3(256), 0, st, (const float*)A,    \
                       (const float*)B, M, N, K, lda, ldb, e)
  switch (epi) {
    case LDMAE_EPI_BIAS: NT_LAUNCH(LDMAE_EPI_BIAS); break;
    case LDMAE_EPI_GATE_RES: NT_LAUNCH(LDMAE_EPI_GATE_RES); break;
    case LDMAE_EPI_BIAS_POS: NT_LAUNCH(LDMAE_EPI_BIAS_POS); break;
    case LDMAE_EPI_SWIGLU: PERS(LDMAE_EPI_SWIGLU); break;
    case LDMAE_EPI_SWIGLU_BWD: PERS(LDMAE_EPI_SWIGLU_BWD); break;
    case LDMAE_EPI_GELU_BWD: NT_LAUNCH(LDMAE_EPI_GELU_BWD); break;
    default: NT_LAUNCH(LDMAE_EPI_BIAS_GELU); break;
  }
#undef NT_LAUNCH
#undef PERS
#undef PERS_GO
#undef PERS_RING4
#undef PERS_ATTR
  if (pi >= 0) ldmae_prof_end(pi, st);
  LDMAE_CHECK_LAUNCH("gemm_nt");
  return LDMAE_OK;
}

extern "C" int ldmae_gemm_nt(int dtype, int out_dtype, int epi, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                             int M, int N, int K, const float* bias, float beta, const float* xin, float* xout,
                             const float* gate, int gate_ld, int rows_per_batch, void* stream) {
  const bool tile_launch = (epi & LDMAE_EPI_TILE_LAUNCH) != 0, half_lines = (epi & LDMAE_EPI_HALF_LINES) != 0, f16_inf = (epi & LDMAE_EPI_F16_INF) != 0;
  epi &= ~(LDMAE_EPI_TILE_LAUNCH | LDMAE_EPI_HALF_LINES | LDMAE_EPI_F16_INF);
  LDMAE_REQUIRE(dtype == LDMAE_F32 || dtype == LDMAE_BF16 || dtype == LDMAE_F16, "gemm_nt: bad dtype %d", dtype);
  LDMAE_REQUIRE(out_dtype == LDMAE_F32 || out_dtype == LDMAE_BF16 || (out_dtype == LDMAE_F16 && dtype == LDMAE_F16), "gemm_nt: bad out_dtype %d", out_dtype);
  LDMAE_REQUIRE(dtype != LDMAE_F16 || out_dtype != LDMAE_BF16, "gemm_nt: fp16 operands give fp16 or f32 outputs");
  LDMAE_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_nt: empty problem M=%d N=%d K=%d", M, N, K);
  LDMAE_REQUIRE(A && B, "gemm_nt: null operand");
  const int kq = dtype != LDMAE_F32 ? 64 : F_BK, al = dtype != LDMAE_F32 ? 8 : 4;
  LDMAE_REQUIRE(K % kq == 0, "gemm_nt: K=%d must be a multiple of %d", K, kq);
  LDMAE_REQUIRE(lda % al == 0 && ldb % al == 0 && lda >= K && ldb >= K, "gemm_nt: lda=%d ldb=%d need 16-B aligned rows >= K", lda, ldb);
  LDMAE_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "gemm_nt: operands must be 16-B aligned");
  EpiArgs e{};
  e.C = C; e.bias = bias; e.ldc = ldc; e.beta = beta;
  e.f16_max = f16_inf ? __builtin_inff() : 65504.f;
  if (epi == LDMAE_EPI_BIAS) {
    LDMAE_REQUIRE(C && ldc >= N, "gemm_nt: C null or ldc < N");
    LDMAE_REQUIRE(beta == 0.f || beta == 1.f, "gemm_nt: beta must be 0 or 1");
  } else if (epi == LDMAE_EPI_GATE_RES) {
    LDMAE_REQUIRE(xin && xout && rows_per_batch > 0 && (!gate || gate_ld >= N), "gemm_nt: gated-residual epilogue needs xin/xout (gate optional)");
    LDMAE_REQUIRE(M % rows_per_batch == 0, "gemm_nt: M=%d not a multiple of rows_per_batch=%d", M, rows_per_batch);
    e.xin = xin; e.xout = xout; e.gate = gate; e.gate_ld = gate_ld; e.rows_per_batch = rows_per_batch;
  } else if (epi == LDMAE_EPI_BIAS_POS) {
    LDMAE_REQUIRE(C && ldc >= N && xin && rows_per_batch > 0, "gemm_nt: pos epilogue needs C and pos table in xin");
    e.xin = xin; e.rows_per_batch = rows_per_batch;
  } else if (epi == LDMAE_EPI_BIAS_GELU) {
    LDMAE_REQUIRE(C && ldc >= N, "gemm_nt: gelu epilogue needs C");
    e.C2 = xout;   /* pre-activation copy, same dtype/ld as C */
  } else if (epi == LDMAE_EPI_GELU_BWD) {
    LDMAE_REQUIRE(C && ldc >= N && xin && out_dtype == dtype, "gemm_nt: gelu-bwd epilogue needs C, the pre-activation in xin (same type and row stride as C) and out_dtype == dtype");
    e.xin = xin;
  } else if (epi == LDMAE_EPI_SWIGLU) {
    LDMAE_REQUIRE(dtype == LDMAE_BF16 && out_dtype == LDMAE_BF16, "gemm_nt: swiglu epilogue is bf16 only");
    LDMAE_REQUIRE(xout && N % 256 == 0 && K % 32 == 0 && (!C || ldc == N), "gemm_nt: swiglu epilogue needs hid (xout), N %% 256 == 0 (N=%d); h12 (C, ldc = N) may be NULL in forward-only calls", N);
    e.xout = xout;
  } else if (epi == LDMAE_EPI_SWIGLU_BWD) {
    LDMAE_REQUIRE(dtype == LDMAE_BF16 && out_dtype == LDMAE_BF16, "gemm_nt: swiglu-bwd epilogue is bf16 only");
    LDMAE_REQUIRE(C && xin && N % 8 == 0 && K % 32 == 0, "gemm_nt: swiglu-bwd epilogue needs dh12 (C), h12 (xin) and Hs = N %% 8 == 0 (N=%d)", N);
    e.xin = xin;
    e.xout = xout;   /* optional: [ceil(M/128)][2*Hs] f32 partial column sums of dh12 (bias gradient of w12) */
  } else {
    LDMAE_FAIL(LDMAE_ERR_INVALID, "gemm_nt: unknown epilogue %d", epi);
  }
  if (dtype == LDMAE_F16) {
    // the TF32-class forward family: whole-line kernel only (every Linear of the VMAE / DiT blocks qualifies), one tile per workgroup or persistent
    LDMAE_REQUIRE(epi == LDMAE_EPI_BIAS || epi == LDMAE_EPI_GATE_RES || epi == LDMAE_EPI_BIAS_POS || epi == LDMAE_EPI_BIAS_GELU || epi == LDMAE_EPI_GELU_BWD,
                  "gemm_nt(fp16): bias, gated residual, pos, gelu, gelu-bwd epilogues (the VMAE blocks), got %d", epi);
    LDMAE_REQUIRE(beta == 0.f, "gemm_nt(fp16): beta must be 0");
    ldmae_count(LDMAE_COUNT_NT_F16);
    int ncu = 0, dev = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    ncu = ncu >= 8 ? ncu / 8 * 8 : 8;
    const int ntiles = cdiv(M, 256) * cdiv(N, 256);
    const bool pers = !tile_launch && cdiv(M, 256) >= 8;
    const int pgrid = (pers && ntiles != ncu) ? ncu : ntiles;
    LDMAE_REQUIRE(ldmae_launch_nt_lines_f16(epi, out_dtype == LDMAE_F16, A, B, M, N, K, lda, ldb, e, pgrid, ntiles, as_stream(stream)),
                  "gemm_nt(fp16): shape outside the whole-line kernel (M=%d N=%d multiples of 8, K=%d lda=%d ldb=%d multiples of 64, 128-B aligned operands)",
                  M, N, K, lda, ldb);
    LDMAE_CHECK_LAUNCH("gemm_nt_lines_f16");
    return LDMAE_OK;
  }
  ldmae_count(dtype == LDMAE_BF16 ? LDMAE_COUNT_NT_BF16 : LDMAE_COUNT_NT_F32);
  return out_dtype == LDMAE_BF16 ? launch_nt<bf16>(dtype, epi, tile_launch, half_lines, A, B, M, N, K, lda, ldb, e, as_stream(stream))
                                 : launch_nt<float>(dtype, epi, tile_launch, half_lines, A, B, M, N, K, lda, ldb, e, as_stream(stream));
}

// ---- the qkv Linear with QK-RMSNorm + RoPE in the epilogue (whole-line kernel only; include/ldmae_hip.h)
extern "C" int ldmae_gemm_nt_qkv_rope_ok(int B, int N, int H, int hd, int K, int lda, int ldb) {
  const long M = (long)B * N;
  return hd == 64 && B > 0 && N > 0 && H > 0 && M % 256 == 0 && M < (1l << 31) && N % 128 == 0 && (3 * H * 64) % 256 == 0 && K > 0 && K % 64 == 0 &&
         lda % 64 == 0 && ldb % 64 == 0 && lda >= K && ldb >= K;
}

extern "C" int ldmae_gemm_nt_qkv_rope(const void* A, int lda, const void* W, int ldb, const float* bias, void* qkv, void* q2, void* k2, const float* wq,
                                      const float* wk, const float* cos, const float* sin, int B, int N, int H, int hd, int K, float eps,
                                      int store_raw_qk, int tile_launch, void* stream) {
  LDMAE_REQUIRE(A && W && qkv && q2 && k2 && cos && sin, "gemm_nt_qkv_rope: null pointer (only bias and wq / wk may be NULL)");
  LDMAE_REQUIRE(!wq == !wk, "gemm_nt_qkv_rope: pass both QK-norm weights or neither (RoPE only)");
  LDMAE_REQUIRE(ldmae_gemm_nt_qkv_rope_ok(B, N, H, hd, K, lda, ldb),
                "gemm_nt_qkv_rope: shape outside the fused kernel (head_dim 64, B*N %% 256 == 0, N %% 128 == 0, 3*H*64 %% 256 == 0, K / lda / ldb %% 64 == 0): "
                "B=%d N=%d H=%d hd=%d K=%d lda=%d ldb=%d -- use ldmae_gemm_nt + ldmae_qknorm_rope_fwd", B, N, H, hd, K, lda, ldb);
  LDMAE_REQUIRE(((uintptr_t)A & 127) == 0 && ((uintptr_t)W & 127) == 0 && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)q2 & 15) == 0 && ((uintptr_t)k2 & 15) == 0 &&
                ((uintptr_t)cos & 15) == 0 && ((uintptr_t)sin & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0) && (!wq || (((uintptr_t)wq | (uintptr_t)wk) & 3) == 0),
                "gemm_nt_qkv_rope: operands must start on 128-B lines, outputs / tables / bias on 16 B");
  const int M = B * N, Nc = 3 * H * 64;
  EpiArgs e{};
  e.C = qkv; e.ldc = Nc; e.bias = bias; e.rows_per_batch = N; e.f16_max = 65504.f;
  e.q2 = q2; e.k2 = k2; e.wq = wq; e.wk = wk; e.cosT = cos; e.sinT = sin; e.heads = H; e.store_raw_qk = store_raw_qk; e.eps = eps;
  hipStream_t st = as_stream(stream);
  const long pi = ldmae_prof_is_on() ? ldmae_prof_begin(st, 2.0 * M * Nc * K) : -1;
  int ncu = 0, dev = 0;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  ncu = ncu >= 8 ? ncu / 8 * 8 : 8;
  const int ntiles = cdiv(M, 256) * cdiv(Nc, 256);
  const bool pers = !tile_launch && cdiv(M, 256) >= 8;
  const int pgrid = (pers && ntiles != ncu) ? ncu : ntiles;
  ldmae_count(LDMAE_COUNT_NT_BF16);
  LDMAE_REQUIRE(ldmae_launch_nt_lines(LDMAE_EPI_QKV_ROPE, 1, A, W, M, Nc, K, lda, ldb, e, pgrid, ntiles, st), "gemm_nt_qkv_rope: the whole-line kernel refused the shape");
  if (pi >= 0) ldmae_prof_end(pi, st);
  LDMAE_CHECK_LAUNCH("gemm_nt_qkv_rope");
  return LDMAE_OK;
}

static int tn_plan(int dtype, int M, int N, int K, int* rows_out) {
  const bool ring = dtype == LDMAE_BF16 && ldmae_tune_get(1) == 0 && M % 32 == 0;
  const int bn = ring ? 256 : (dtype == LDMAE_BF16 ? TN_BN : FT_BN), bk = ring ? 256 : (dtype == LDMAE_BF16 ? TN_BK : FT_BK);
  const long tiles = (long)cdiv(N, bn) * cdiv(K, bk);
  // ring kernel: ONE round of at most 256 workgroups (one per CU; measured 5-10 % faster than two rounds of 512: no second-round
  // tail, longer row splits); tune key 9 overrides the target for A/B runs
  const long target = ring ? (ldmae_tune_get(9) > 0 ? ldmae_tune_get(9) : 256) : 2048;
  long want = target / tiles;
  long maxs = M / (64 * 8) > 0 ? M / (64 * 8) : 1;           // at least 8 steps of 64 rows per split
  long s = want < maxs ? want : maxs;
  if (s < 1) s = 1;
  if (s > 256) s = 256;          // one-tile outputs (VMAE: 192 x 192) take a split per CU (128 left half the chip idle)
  int rows = (int)(((long)M + s - 1) / s);
  rows = (rows + 63) / 64 * 64;
  if (rows_out) *rows_out = rows;
  return (M + rows - 1) / rows;
}

extern "C" int ldmae_gemm_tn_splits(int dtype, int M, int N, int K) { return tn_plan(dtype == LDMAE_F16 ? LDMAE_BF16 : dtype, M, N, K, nullptr); }

extern "C" long ldmae_gemm_tn_workspace_bytes(int dtype, int M, int N, int K) {
  return (long)tn_plan(dtype == LDMAE_F16 ? LDMAE_BF16 : dtype, M, N, K, nullptr) * ((long)N * K + N) * 4;
}

extern "C" int ldmae_gemm_tn(int dtype, const void* A, int lda, const void* B, int ldb, float* C, float* dbias, int M, int N, int K,
                             float beta, float* workspace, long workspace_bytes, void* stream) {
  LDMAE_REQUIRE(dtype == LDMAE_F32 || dtype == LDMAE_BF16 || dtype == LDMAE_F16, "gemm_tn: bad dtype %d", dtype);
  LDMAE_REQUIRE(M > 0 && N > 0 && K > 0 && A && B && C, "gemm_tn: empty problem or null pointer");
  const bool half = dtype == LDMAE_F16;
  if (half) dtype = LDMAE_BF16;                 // same tiling, plan and workspace as bf16; only the MFMA differs
  LDMAE_REQUIRE(beta == 0.f || beta == 1.f, "gemm_tn: beta must be 0 or 1");
  const int al = dtype == LDMAE_BF16 ? 8 : 4;
  LDMAE_REQUIRE(N % al == 0 && K % al == 0 && lda % al == 0 && ldb % al == 0, "gemm_tn: N=%d K=%d lda=%d ldb=%d must be multiples of %d", N, K, lda, ldb, al);
  LDMAE_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 && ((uintptr_t)C & 15) == 0, "gemm_tn: pointers must be 16-B aligned");
  int rows = 0;
  const int splits = tn_plan(dtype, M, N, K, &rows);
  const bool ring = dtype == LDMAE_BF16 && ldmae_tune_get(1) == 0 && M % 32 == 0;
  hipStream_t st = as_stream(stream);
  ldmae_count(half ? LDMAE_COUNT_TN_F16 : (dtype == LDMAE_BF16 ? LDMAE_COUNT_TN_BF16 : LDMAE_COUNT_TN_F32));
  LDMAE_REQUIRE(workspace && workspace_bytes >= (long)splits * ((long)N * K + N) * 4, "gemm_tn: workspace too small (%ld < %ld)",
                workspace_bytes, (long)splits * ((long)N * K + N) * 4);
  // one split and nothing to accumulate into: the GEMM writes C (and the bias gradient) itself -- no partial slab, no reduce pass
  const bool direct = splits == 1 && beta == 0.f;
  float* P = direct ? C : workspace;
  float* Pb = direct ? dbias : workspace + (size_t)splits * N * K;
  if (ring) {
    constexpr int lds = 5 * 32768;   // ring (4 x 32 KiB) and the 139 KiB epilogue region share it
    if (ldmae_tune_get(4) == 3) hipFuncSetAttribute((const void*)gemm_tn_ring_kernel<4, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    else if (ldmae_tune_get(4) == 4) hipFuncSetAttribute((const void*)gemm_tn_ring_kernel<4, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    else hipFuncSetAttribute((const void*)gemm_tn_ring_kernel<4, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const unsigned grid = cdiv(N, 256) * cdiv(K, 256) * splits;
    // default: 8 waves (128x64 each), 4-deep ring, the two wave groups half a step apart (1115-1135 TF/s with the fused bias
    // gradient vs 965-1030 for 16 lock-step waves); tune key 4: 3 = 16 lock-step waves, 4 = 16 staggered waves.
    // The bias-gradient MFMAs stay on the wk == 0 waves: spreading them over all waves (a wave-uniform switch on wk, or one
    // slot per K-tile workgroup) measured 8-15 % SLOWER -- every wave's MFMA phase is on the staggered loop's critical path.
    if (ldmae_tune_get(4) == 3)
      hipLaunchKernelGGL((gemm_tn_ring_kernel<4, 4, false>), dim3(grid), dim3(1024), lds, st, (const bf16*)A, (const bf16*)B, P, dbias ? Pb : nullptr, M, N, K, lda, ldb, rows);
    else if (ldmae_tune_get(4) == 4)
      hipLaunchKernelGGL((gemm_tn_ring_kernel<4, 4, true>), dim3(grid), dim3(1024), lds, st, (const bf16*)A, (const bf16*)B, P, dbias ? Pb : nullptr, M, N, K, lda, ldb, rows);
    else if (half) {
      hipFuncSetAttribute((const void*)gemm_tn_ring_kernel<4, 2, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      hipLaunchKernelGGL((gemm_tn_ring_kernel<4, 2, true, true>), dim3(grid), dim3(512), lds, st, (const bf16*)A, (const bf16*)B, P, dbias ? Pb : nullptr, M, N, K, lda, ldb, rows);
    } else
      hipLaunchKernelGGL((gemm_tn_ring_kernel<4, 2, true>), dim3(grid), dim3(512), lds, st, (const bf16*)A, (const bf16*)B, P, dbias ? Pb : nullptr, M, N, K, lda, ldb, rows);
  } else if (dtype == LDMAE_BF16) {
    const unsigned grid = cdiv(N, TN_BN) * cdiv(K, TN_BK) * splits;
    if (half) hipLaunchKernelGGL(gemm_tn_bf16_kernel<true>, dim3(grid), dim3(256), 4 * TN_TILE_BYTES, st, (const bf16*)A, (const bf16*)B, P, M, N, K, lda, ldb, rows);
    else hipLaunchKernelGGL(gemm_tn_bf16_kernel<false>, dim3(grid), dim3(256), 4 * TN_TILE_BYTES, st, (const bf16*)A, (const bf16*)B, P, M, N, K, lda, ldb, rows);
  } else {
    const unsigned grid = cdiv(N, FT_BN) * cdiv(K, FT_BK) * splits;
    hipLaunchKernelGGL(gemm_tn_f32_kernel, dim3(grid), dim3(256), 0, st, (const float*)A, (const float*)B, P, M, N, K, lda, ldb, rows);
  }
  LDMAE_CHECK_LAUNCH("gemm_tn");
  if (!direct) {
    if (dbias && ring) splitk_reduce_pair(P, C, (long)N * K, Pb, dbias, (long)N, splits, beta, st);
    else splitk_reduce(P, C, (long)N * K, splits, beta, st);
    LDMAE_CHECK_LAUNCH("splitk_reduce");
  }
  if (dbias && !ring) {   // non-ring paths: separate column-sum pass (elementwise.hip), re-using the workspace
    LDMAE_REQUIRE(workspace_bytes >= ldmae_colsum_workspace_bytes(M, N), "gemm_tn: workspace too small for the bias-gradient pass");
    return ldmae_colsum(half ? LDMAE_F16 : dtype, A, lda, M, N, dbias, beta, workspace, stream);
  }
  return LDMAE_OK;
}
