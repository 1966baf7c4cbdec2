// Diagnostic build only (tools/bench_nt.py --key 15): the bf16 NT GEMM with a WHOLE-LINE ring DMA.  Negative result of round 4, kept as
// evidence: profiles/r04_wl_ab.txt.
#include "../common.h"
#include "../gemm_nt_common.h"

// ------------------------------------------------------------------------------------------------
// bf16 NT GEMM, whole-line form.  Same tile (256 x 256, 8 waves of 128 x 64), same two wave groups half a K-step apart, same epilogues as
// gemm_nt_persist_kernel; what changes is the SHAPE OF THE RING DMA.  The kernel above is bound by the CU's vector-memory path, not by the
// matrix pipe (profiles/r04_pmc_ta.md: one 64-B L1 -> L2 request per 3 cycles and CU, data-return unit 80-90 % busy, L2 latency 360 cycles):
// its LDS image has 64-B rows (32 k), so a DMA piece is 16 rows x 64 B -- sixteen half-line requests -- and the other half of every line
// is asked for again one K-step later.  Here an LDS row holds 64 k = one whole 128-B line, a piece is 8 rows x 128 B (eight whole-line
// requests for the same KiB), and a line is requested ONCE per tile.
//   LDS: two PAIRS of slots, pair p = [A slot | B slot], a slot = 256 rows x 128 B = 32 KiB.  16-B chunk c (0..7) of row r sits at position
//        c ^ ((r >> 1) & 7): conflict-free for the ds_read_b128 lane groups of the 16x16x32 operands (a group's sixteen rows x one or
//        two chunks cover all sixteen 16-B slots of a 256-B bank row).
//   K-steps stay 32 deep (32 MFMAs per wave and phase): step kt reads chunk half h = kt & 1 of block j = kt >> 1 in pair j & 1.
//   DMA: block j+1 goes into the other pair during steps 2j and 2j+1 -- that pair was last read in step 2j-1.  Waves 0-3 (group A, whose
//        load phase comes first) bring the A rows, four pieces in each of the two steps; waves 4-7 (group B) bring the B rows, all eight
//        pieces in the EVEN step: pieces requested in group B's load phase of the odd step would have to be complete at the end of that very
//        phase.  Every piece has at least one full K-step to land; the waits are vmcnt(0) at the end of the odd step (everything a wave has
//        in flight is needed by then).
//   The epilogue strips alias pair 1, which is idle during an epilogue (the next tile's block 0 goes into pair 0; block 1 is requested in
//   K-step 0, behind the tile-start barrier); the SwiGLU-bwd scratch sits behind the ring (160 KiB in all for that epilogue).
// Requires M, N multiples of 8 (a piece's eight rows are clamped as a whole at the edges), lda / ldb multiples of 64 elements (rows start
// on line boundaries) and K a multiple of 64; other shapes keep gemm_nt_persist_kernel.
// ------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(512) void gemm_nt_wl_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, int M, int N, int K, int lda, int ldb,
                                                         EpiArgs e, int ntiles) {
  constexpr int BM = 256, BN = 256, WN = 4, TM = 128, TNn = 64, MI = 8, NI = 4;
  constexpr int SLOT = 256 * 128, PAIR = 2 * SLOT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const unsigned tiles_n = (N + BN - 1) / BN;
  float* ew = (float*)(smem + PAIR) + wave * (16 * 68);
  float* ex = (float*)(smem + 2 * PAIR) + wave * 1024;
  const bool persistent = (int)gridDim.x != ntiles;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  const int rbx = (((M + BM - 1) / BM) + 7) / 8;
  const int first = persistent ? xcd * rbx * (int)tiles_n + slot : (int)xcd_remap(blockIdx.x, gridDim.x);
  const int tend = persistent ? min(ntiles, (xcd + 1) * rbx * (int)tiles_n) : ntiles;
  const int tstride = persistent ? per_xcd : ntiles;

  // ---- ring DMA: wave-uniform piece bases + ONE 32-bit lane offset (all pieces of a wave have the same row parity of their first row / 8)
  const bool ldA = wave < 4;
  const int w4 = wave & 3;
  const unsigned ld_op = ldA ? (unsigned)lda : (unsigned)ldb;
  const unsigned dma_voff = (unsigned)(lane >> 3) * ld_op * 2u + (unsigned)((((lane & 7) ^ (4 * (w4 & 1) + (lane >> 4)))) << 4);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(void, smem)) + (ldA ? 0 : SLOT) + w4 * 1024;
  const char* pbase[8];          // first row of piece i (rows 8 * w4 + 32 * i of the tile's A or B rows), scalar registers
  int m0 = 0, n0 = 0;
  auto set_tile = [&](int t) {
    m0 = (t / tiles_n) * BM; n0 = (t % tiles_n) * BN;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (ldA) pbase[i] = (const char*)(A + (size_t)min(m0 + 8 * w4 + 32 * i, M - 8) * lda);
      else {
        int brow = n0 + 8 * w4 + 32 * i;
        // SwiGLU: tile rows rl of B are w12 rows ((rl & 32) ? Hs : 0) + n0 / 2 + (rl >> 6) * 32 + (rl & 31) (x1 and x2 of a hidden unit in one wave)
        if constexpr (EPI == LDMAE_EPI_SWIGLU) brow = ((i & 1) ? (N >> 1) : 0) + (n0 >> 1) + (i >> 1) * 32 + 8 * w4;
        pbase[i] = (const char*)(B + (size_t)min(brow, N - 8) * ldb);
      }
    }
  };
  // pieces i0 .. i1-1 of block `blk` into pair `pr`
  auto issue = [&](int blk, int pr, int i0, int i1) {
    const unsigned la = lds0 + pr * PAIR;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i >= i0 && i < i1) glds16_s(pbase[i] + (size_t)blk * 128, dma_voff, la + i * 4096);
  };
  // fragment addresses (chunk half 0; half 1 = the same with bit 6 flipped)
  const int fsw = ((lane >> 4) ^ ((lane & 15) >> 1)) << 4;
  const int a_off = (wm * TM + (lane & 15)) * 128 + fsw, b_off = SLOT + (wn * TNn + (lane & 15)) * 128 + fsw;
  const int nk = K / 32, nb = K / 64;
  const bool grpB = wm >= 1;

  int t = first;
  if (t < tend) {
    set_tile(t);
    issue(0, 0, 0, 8);
  }
  while (t < tend) {
    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // tile start: as in gemm_nt_persist_kernel (block 0 landed, the previous epilogue's accesses retired, every wave out of the strips)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_barrier();
    if (grpB) __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
      const int j = kt >> 1, h = kt & 1;
      const bool more = j + 1 < nb;
      const char* st = smem + (j & 1) * PAIR;
      const int ao = a_off ^ (h << 6), bo = b_off ^ (h << 6);
      bf16x8 af[MI], bfr[NI];
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int jj = 0; jj < NI; ++jj) bfr[jj] = *(const bf16x8*)(st + bo + jj * 2048);
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8*)(st + ao + i * 2048);
      if (more) {
        if (h == 0) { if (ldA) issue(j + 1, (j + 1) & 1, 0, 4); else issue(j + 1, (j + 1) & 1, 0, 8); }
        else if (ldA) issue(j + 1, (j + 1) & 1, 4, 8);
      }
      __builtin_amdgcn_s_setprio(0);
      if (grpB && h == 1 && more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jj = 0; jj < NI; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[jj], acc[i][jj], 0, 0, 0);
      if (!grpB && h == 1 && more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (!grpB) __builtin_amdgcn_s_barrier();             // every wave is past its last fragment read: the ring is free
    const int em0 = m0, en0 = n0;
    t += tstride;
    if (t < tend) {
      set_tile(t);
      issue(0, 0, 0, 8);
    }
    nt_epilogue<EPI, bf16, TM, TNn, MI, NI>(acc, ew, ex, e, em0, en0, wm, wn, lane, M, N);
  }
}


int ldmae_launch_nt_wl(int epi, const void* A, const void* B, int M, int N, int K, int lda, int ldb, const EpiArgs& e, int grid, int ntiles,
                       hipStream_t st) {
#define WL_GO(E)                                                                                                                        \
  {                                                                                                                                      \
    constexpr int ldsw = 4 * 256 * 128 + (E == LDMAE_EPI_SWIGLU_BWD ? 8 * 1024 * 4 : 0);                                                  \
    hipFuncSetAttribute((const void*)gemm_nt_wl_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsw);                             \
    hipLaunchKernelGGL((gemm_nt_wl_kernel<E>), dim3(grid), dim3(512), ldsw, st, (const bf16*)A, (const bf16*)B, M, N, K, lda, ldb, e, ntiles); \
    return 1;                                                                                                                            \
  }
  switch (epi) {
    case LDMAE_EPI_BIAS: WL_GO(LDMAE_EPI_BIAS);
    case LDMAE_EPI_GATE_RES: WL_GO(LDMAE_EPI_GATE_RES);
    case LDMAE_EPI_SWIGLU: WL_GO(LDMAE_EPI_SWIGLU);
    case LDMAE_EPI_SWIGLU_BWD: WL_GO(LDMAE_EPI_SWIGLU_BWD);
    default: return 0;
  }
#undef WL_GO
}
