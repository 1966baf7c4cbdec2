// bf16 NT GEMM, WHOLE-LINE form (round 5): C[M,N] = A[M,K] . B[N,K]^T with the fused epilogues of gemm_nt_common.h.
//
// Why it exists (profiles/r05_ingest.md, csrc/probe/ingest_probe.hip).  gemm_nt_persist_kernel stages 32-deep K-steps: its LDS rows are 64 B, so a
// ring-DMA piece is 16 rows x 64 B = sixteen HALF-line requests, and the other half of every line is asked for again one K-step later.  The
// probe shows what that costs: with all 256 CUs loading, an XCD's L2 serves ~13 requests per clock whether they are 64 B or 128 B, so a CU
// takes in 21-28 B/clk in half lines (1170-1450 cycles for the 32 KiB of a 256 x 256 x 32 K-step, against 1024 cycles of MFMA issue) and
// 40-52 B/clk in whole lines.  Round 4's whole-line kernel (probe/gemm_nt_wl.hip) asked for whole lines but kept only ONE 64-KiB block in
// flight behind full vmcnt(0) drains and lost what the request shape won.  Here:
//   * LDS rows hold 64 k = one 128-B line; a piece = 8 rows x 128 B; a line is requested ONCE per tile.
//   * The ring is five HALF-block slots of 32 KiB (the 256 A rows, or the 256 B rows, x 128 B): 160 KiB.  The issue stream is
//     A0 B0 A1 B1 A2 B2 ... (slot = position mod 5) and runs on ACROSS tiles: during block j (K-steps 2j and 2j+1, 32 deep each) every wave
//     issues its 4 pieces of B(j+1) in the first K-step and its 4 pieces of A(j+2) in the second -- 4 pieces per wave and K-step as before,
//     the L2-resident weight rows one block ahead, the streamed activation rows two.  One counted wait per block (vmcnt(4): the A pieces
//     just issued stay in flight).
//   * Tile boundary: the A half-block that would be issued in a tile's LAST K-step is deferred to the next tile's first K-step, so during
//     an epilogue only the next tile's A0 and B0 are in flight and three slots are idle: the per-wave epilogue strips (8 x 4352 B) take two
//     of them, the SwiGLU-backward column-sum scratch (8 x 4 KiB) the third.
//   * Everything else is gemm_nt_persist_kernel: 256 x 256 tile, 8 waves of 128 x 64 (8 x 4 tiles of mfma_f32_16x16x32_bf16), the two wave
//     groups half a K-step apart, XCD-owned row-block ranges, nt_epilogue.  Results are bitwise equal to that kernel's (same products in the
//     same order; tests/test_gpu_kernels.py).
// LDS image: 16-B chunk c (0..7) of row r at position c ^ ((r >> 1) & 7): conflict-free for the ds_read_b128 lane groups of the 16x16x32
// operands.  Requires K % 64 == 0, lda / ldb % 64 == 0 and 128-B aligned operands (rows start on line boundaries), M, N multiples of 8 (a
// piece's eight rows are clamped as a whole at the edges); other shapes keep gemm_nt_persist_kernel.
#include "common.h"

#include "gemm_nt_common.h"

// s_barrier pinned against the instruction scheduler: MFMA builtins have no side effects, and the scheduler was seen hoisting a K-step's
// closing barrier above its 32 MFMAs (csrc/probe/ingest_probe.hip, mode 7 before the pin: 1700 cycles per K-step instead of 1150)
#define LINES_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

// InT: bf16, or f16 (same bytes, v_mfma_f32_16x16x32_f16: the TF32-class forward path; A / B stay typed as 2-byte elements)
template <int EPI, typename OutT, typename InT = bf16>
__global__ __launch_bounds__(512) void gemm_nt_lines_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, int M, int N, int K, int lda,
                                                            int ldb, EpiArgs e, int ntiles) {
  constexpr int BM = 256, BN = 256, WN = 4, TM = 128, TNn = 64, MI = 8, NI = 4, SLOT = 256 * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (N + BN - 1) / BN;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(void, smem));
  // tile ownership as in gemm_nt_persist_kernel
  const bool persistent = (int)gridDim.x != ntiles;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  const int rbx = (((M + BM - 1) / BM) + 7) / 8;
  const int first = persistent ? xcd * rbx * tiles_n + slot : (int)xcd_remap(blockIdx.x, gridDim.x);
  const int tend = persistent ? min(ntiles, (xcd + 1) * rbx * tiles_n) : ntiles;
  const int tstride = persistent ? per_xcd : ntiles;
  const int nb = K / 64;

  // ---- ring DMA.  Piece i (0..3) of a wave = rows 8 * wave + 64 * i of the half-block: all pieces of a wave have the same swizzle phase, so
  // ONE 32-bit lane offset per operand serves them all; the piece bases are wave-uniform (scalar registers), recomputed once per tile.
  const unsigned swz = (unsigned)(((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7)) << 4);
  const unsigned voffA = (unsigned)(lane >> 3) * (unsigned)lda * 2u + swz, voffB = (unsigned)(lane >> 3) * (unsigned)ldb * 2u + swz;
  const char* pa[4];
  const char* pb[4];
  auto rowsA = [&](int t) {
    const int m0 = (t / tiles_n) * BM;
#pragma unroll
    for (int i = 0; i < 4; ++i) pa[i] = (const char*)(A + (size_t)min(m0 + 8 * wave + 64 * i, M - 8) * lda);
  };
  auto rowsB = [&](int t) {
    const int n0 = (t % tiles_n) * BN;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int brow = n0 + 8 * wave + 64 * i;
      // SwiGLU: tile row rl of B is w12 row ((rl & 32) ? Hs : 0) + n0 / 2 + (rl >> 6) * 32 + (rl & 31): x1 and x2 of a hidden unit in one wave
      if constexpr (EPI == LDMAE_EPI_SWIGLU) brow = ((wave & 4) ? (N >> 1) : 0) + (n0 >> 1) + i * 32 + 8 * (wave & 3);
      pb[i] = (const char*)(B + (size_t)min(brow, N - 8) * ldb);
    }
  };
  int tA = first, jA = 0, tB = first, jB = 0;            // next half-block of each stream: (tile, 64-deep block)
  unsigned sA = 0, sB = 1;                                // and the slot it goes to (stream position mod 5; A even positions, B odd)
  auto issueA = [&]() -> bool {
    if (tA >= tend) return false;
    const unsigned la = lds0 + sA * SLOT + wave * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16_s(pa[i] + jA * 128, voffA, la + i * 8192);
    sA = sA >= 3 ? sA - 3 : sA + 2;
    if (++jA == nb) { jA = 0; tA += tstride; if (tA < tend) rowsA(tA); }
    return true;
  };
  auto issueB = [&]() -> bool {
    if (tB >= tend) return false;
    const unsigned la = lds0 + sB * SLOT + wave * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16_s(pb[i] + jB * 128, voffB, la + i * 8192);
    sB = sB >= 3 ? sB - 3 : sB + 2;
    if (++jB == nb) { jB = 0; tB += tstride; if (tB < tend) rowsB(tB); }
    return true;
  };
  // fragment addresses inside a slot (chunk half 0; half 1 = the same with bit 6 flipped)
  const int fsw = ((lane >> 4) ^ ((lane & 15) >> 1)) << 4;
  const int a_off = (wm * TM + (lane & 15)) * 128 + fsw, b_off = (wn * TNn + (lane & 15)) * 128 + fsw;
  const bool grpB = wm >= 1;

  int t = first;
  if (t < tend) {
    rowsA(t); rowsB(t);
    issueA(); issueB();                                   // A0 B0; A1 is the first tile's "deferred" half-block
  }
  unsigned ga = 0, gb = 1;                                // slots of the block being multiplied
  while (t < tend) {
    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // tile start: as in gemm_nt_persist_kernel -- vmcnt(0) through the builtin (the compiler then knows no epilogue load is pending and puts no
    // vmcnt wait of its own into the K loop); A0 and B0 of this tile have landed, every wave is out of the strips
    __builtin_amdgcn_s_waitcnt(0x0F70);
    LINES_BAR();
    if (grpB) LINES_BAR();
    for (int j = 0; j < nb; ++j) {
      const char* sa = smem + ga * SLOT;
      const char* sb = smem + gb * SLOT;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int ao = a_off ^ (h << 6), bo = b_off ^ (h << 6);
        bf16x8 af[MI], bfr[NI];
        __builtin_amdgcn_s_setprio(1);      // load phase at raised priority
#pragma unroll
        for (int jj = 0; jj < NI; ++jj) bfr[jj] = *(const bf16x8*)(sb + bo + jj * 2048);
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8*)(sa + ao + i * 2048);
        bool issued = false;
        if (h == 0) {
          if (j == 0) issueA();                           // the half-block deferred over the tile boundary
          issueB();
        } else if (j + 1 < nb) issued = issueA();         // (a tile's last K-step issues nothing: deferred)
        __builtin_amdgcn_s_setprio(0);
        // block j+1 (A issued a block ago, B in the previous K-step) has to be complete one phase before anyone reads it
        if (grpB && h == 1 && j + 1 < nb) {
          if (issued) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        LINES_BAR();
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int jj = 0; jj < NI; ++jj) {
            if constexpr (sizeof(InT) == 2 && !__is_same(InT, bf16))
              acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[i]), __builtin_bit_cast(f16x8, bfr[jj]), acc[i][jj], 0, 0, 0);
            else acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[jj], acc[i][jj], 0, 0, 0);
          }
        if (!grpB && h == 1 && j + 1 < nb) {
          if (issued) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        LINES_BAR();
      }
      ga = ga >= 3 ? ga - 3 : ga + 2;
      gb = gb >= 3 ? gb - 3 : gb + 2;
    }
    if (!grpB) LINES_BAR();                               // every wave is past its last fragment read
    // epilogue scratch in the three idle slots: next A slot and next B slot (strips of waves 0-3 / 4-7), the A slot after that (`ex`)
    const unsigned sx = sA >= 3 ? sA - 3 : sA + 2;
    float* ew = (float*)(smem + ((wave & 4) ? sB : sA) * SLOT) + (wave & 3) * (16 * 68);
    float* ex = (float*)(smem + sx * SLOT) + wave * 1024;
    const int em0 = (t / tiles_n) * BM, en0 = (t % tiles_n) * BN;
    t += tstride;
    nt_epilogue<EPI, OutT, TM, TNn, MI, NI>(acc, ew, ex, e, em0, en0, wm, wn, lane, M, N);
  }
}

// returns 0 when the shape / arguments are outside what the whole-line kernel covers (the caller then launches gemm_nt_persist_kernel)
template <typename OutT>
static int launch_lines(int epi, const void* A, const void* B, int M, int N, int K, int lda, int ldb, const EpiArgs& e, int grid, int ntiles,
                        hipStream_t st) {
  constexpr int lds = 5 * 256 * 128;
#define LINES_GO(E)                                                                                                                    \
  {                                                                                                                                     \
    hipFuncSetAttribute((const void*)gemm_nt_lines_kernel<E, OutT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);                   \
    hipLaunchKernelGGL((gemm_nt_lines_kernel<E, OutT>), dim3(grid), dim3(512), lds, st, (const bf16*)A, (const bf16*)B, M, N, K, lda, ldb, e, \
                       ntiles);                                                                                                        \
    return 1;                                                                                                                           \
  }
  switch (epi) {
    case LDMAE_EPI_BIAS: LINES_GO(LDMAE_EPI_BIAS);
    case LDMAE_EPI_GATE_RES: LINES_GO(LDMAE_EPI_GATE_RES);
    case LDMAE_EPI_BIAS_POS: LINES_GO(LDMAE_EPI_BIAS_POS);
    case LDMAE_EPI_BIAS_GELU: LINES_GO(LDMAE_EPI_BIAS_GELU);
    case LDMAE_EPI_GELU_BWD: LINES_GO(LDMAE_EPI_GELU_BWD);
    case LDMAE_EPI_SWIGLU: if constexpr (sizeof(OutT) == 2) LINES_GO(LDMAE_EPI_SWIGLU) else return 0;
    case LDMAE_EPI_SWIGLU_BWD: if constexpr (sizeof(OutT) == 2) LINES_GO(LDMAE_EPI_SWIGLU_BWD) else return 0;
    case LDMAE_EPI_QKV_ROPE: if constexpr (sizeof(OutT) == 2) LINES_GO(LDMAE_EPI_QKV_ROPE) else return 0;
    default: return 0;
  }
#undef LINES_GO
}

static bool lines_shape_ok(int epi, const void* A, const void* B, int M, int N, int K, int lda, int ldb) {
  if (M < 8 || N < 8 || M % 8 != 0 || N % 8 != 0 || K % 64 != 0 || lda % 64 != 0 || ldb % 64 != 0) return false;
  if (((uintptr_t)A & 127) != 0 || ((uintptr_t)B & 127) != 0) return false;
  return !(epi == LDMAE_EPI_SWIGLU && N % 256 != 0);
}

template <typename OutT>
static int launch_lines_f16(int epi, const void* A, const void* B, int M, int N, int K, int lda, int ldb, const EpiArgs& e, int grid, int ntiles,
                            hipStream_t st) {
  constexpr int lds = 5 * 256 * 128;
#define LINES_GO(E)                                                                                                                    \
  {                                                                                                                                     \
    hipFuncSetAttribute((const void*)gemm_nt_lines_kernel<E, OutT, f16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);              \
    hipLaunchKernelGGL((gemm_nt_lines_kernel<E, OutT, f16>), dim3(grid), dim3(512), lds, st, (const bf16*)A, (const bf16*)B, M, N, K, lda, ldb, \
                       e, ntiles);                                                                                                     \
    return 1;                                                                                                                           \
  }
  switch (epi) {
    case LDMAE_EPI_BIAS: LINES_GO(LDMAE_EPI_BIAS);
    case LDMAE_EPI_GATE_RES: LINES_GO(LDMAE_EPI_GATE_RES);
    case LDMAE_EPI_BIAS_POS: LINES_GO(LDMAE_EPI_BIAS_POS);
    case LDMAE_EPI_BIAS_GELU: LINES_GO(LDMAE_EPI_BIAS_GELU);
    case LDMAE_EPI_GELU_BWD: if constexpr (sizeof(OutT) == 2) LINES_GO(LDMAE_EPI_GELU_BWD) else return 0;
    default: return 0;
  }
#undef LINES_GO
}

int ldmae_launch_nt_lines_f16(int epi, int out_f16, const void* A, const void* B, int M, int N, int K, int lda, int ldb, const EpiArgs& e, int grid,
                              int ntiles, hipStream_t st) {
  if (!lines_shape_ok(epi, A, B, M, N, K, lda, ldb)) return 0;
  return out_f16 ? launch_lines_f16<f16>(epi, A, B, M, N, K, lda, ldb, e, grid, ntiles, st)
                 : launch_lines_f16<float>(epi, A, B, M, N, K, lda, ldb, e, grid, ntiles, st);
}

int ldmae_launch_nt_lines(int epi, int out_bf16, const void* A, const void* B, int M, int N, int K, int lda, int ldb, const EpiArgs& e, int grid,
                          int ntiles, hipStream_t st) {
  if (!lines_shape_ok(epi, A, B, M, N, K, lda, ldb)) return 0;
  // the lane offsets are 32-bit: 8 rows of the operand must stay below 4 GiB (they do by many orders of magnitude)
  return out_bf16 ? launch_lines<bf16>(epi, A, B, M, N, K, lda, ldb, e, grid, ntiles, st)
                  : launch_lines<float>(epi, A, B, M, N, K, lda, ldb, e, grid, ntiles, st);
}
