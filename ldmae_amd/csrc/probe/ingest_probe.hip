// Per-CU operand INGEST probe for the bf16 NT GEMM (round 5, verdict item 1): how many bytes per clock can a CU take in for a 256 x 256 x 32
// K-step (32 KiB of operands: 256 A rows + 256 B rows of 64 B) while its matrix pipes run, and does it depend on WHO issues the loads and
// by WHICH path?  One persistent 512-thread workgroup per CU walks the tiles of a real GEMM shape (A streamed from HBM / L2, B resident in
// L2) exactly as gemm_nt_persist_kernel does; nothing is stored (the accumulators are summed into one float per lane at the end).
//
//   mode 0  the shipped pattern: all 8 waves issue 4 LDS-DMA pieces (16 rows x 64 B) + 12 fragment reads + 32 MFMAs per K-step, the two
//           wave groups half a K-step apart, 3-deep ring
//   mode 1  4 dedicated LOADER waves (8 LDS-DMA pieces each per K-step, nothing else) + 4 CONSUMER waves (one per SIMD, 128 x 128 each:
//           16 fragment reads + 64 MFMAs per K-step), one barrier per K-step, 4-deep ring (three K-steps in flight)
//   mode 2  as mode 1, but the loaders move the bytes through registers (global_load_dwordx4 -> ds_write_b128)
//   mode 3  all 8 waves load + multiply as in mode 0, but through registers (4 global_load_dwordx4 per wave and K-step, written to LDS one
//           K-step later)
//   mode 4  A never touches LDS: a wave owns 32 rows x 256 columns and loads its two A fragments per K-step straight into the MFMA operand
//           registers (16 B per lane = the 16x16x32 A layout); only B goes through LDS-DMA (2 pieces per wave and K-step, 16 B fragments)
//   flags   1 = MFMAs on, 2 = fragment reads on, 4 = `nt` on the A stream
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form probe/ingest_probe.hip -o probe/ingest_probe && probe/ingest_probe
#include "../common.h"

#include <stdlib.h>
#include <type_traits>
#include <algorithm>
#include <vector>

__device__ __forceinline__ int ring_f(int g) { return (0x78 >> (2 * g)) & 3; }

template <bool NT> __device__ __forceinline__ void glds16_p(const void* g, unsigned lds_addr) {
  if constexpr (NT) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(g), "s"(lds_addr) : "memory", "m0");
  else asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_addr) : "memory", "m0");
}
// register-path load hidden from the compiler's waitcnt pass (the waits are counted by hand, like the DMA ring's)
template <bool NT> __device__ __forceinline__ void gload16(f32x4& d, const void* g) {
  if constexpr (NT) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(d) : "v"(g) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(g) : "memory");
}
template <int N> __device__ __forceinline__ void vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// a counted wait that the loaded registers data-depend on: the compiler cannot read (or copy) them above it
template <int N> __device__ __forceinline__ void vmwait_on(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void vmwait_on2(f32x4& a, f32x4& b) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int BM = 256, BN = 256, STAGE_BYTES = (BM + BN) * 64;

__device__ char* g_xbuf = nullptr;         // mode 7 + epilogue-like traffic: streamed reads from here ...
__device__ char* g_ybuf = nullptr;         // ... and streamed writes to here, EL / ES KiB per wave and K-step, issued inside the K loop
__device__ int g_bstream = 0;             // mode 7: 1 = the B rows are STREAMED too, shared like the A rows (the TN GEMM's situation); 2 = streamed, unshared
__device__ int g_wrap_rb = 1 << 30;       // E2: A row-blocks wrap at this count (small = the A stream is L2-resident)
struct Walk {   // tile ownership of gemm_nt_persist_kernel (persistent form)
  int t, tend, tstride, tiles_n;
  __device__ Walk(int M, int N, int ntiles) {
    tiles_n = (N + BN - 1) / BN;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int rbx = (((M + BM - 1) / BM) + 7) / 8;
    t = xcd * rbx * tiles_n + slot;
    tend = min(ntiles, (xcd + 1) * rbx * tiles_n);
    tstride = per_xcd;
  }
};

template <int MODE, int FLAGS>
__global__ __launch_bounds__(512) void ingest_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, int M, int N, int K, int ntiles,
                                                     float* __restrict__ out, unsigned long long* __restrict__ stamps) {
  constexpr bool MF = FLAGS & 1, FR = FLAGS & 2, NTA = (FLAGS & 4) != 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(void, smem));
  const int nk = K / 32, lda = K, ldb = K;
  Walk w(M, N, ntiles);
  unsigned long long c0 = 0, r0 = 0;
  if (tid == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int fpos = ((lane >> 4) ^ ring_f((lane >> 2) & 3)) << 4;
  bf16x8 rz;                                           // register operands when the fragment reads are off
#pragma unroll
  for (int j = 0; j < 8; ++j) rz[j] = (bf16)(0.01f * (float)((lane * 7 + j * 13) % 97) - 0.5f);

  if constexpr (MODE == 0 || MODE == 3) {
    // ---------------------------------------------------------------- all 8 waves load and multiply (the shipped schedule)
    constexpr int STAGES = 3, PPW = 4, WN = 4, TM = 128, TNn = 64, MI = 8, NI = 4;
    const int wm = wave / WN, wn = wave % WN;
    const bool grpB = wm >= 1;
    const int a_off = (wm * TM + (lane & 15)) * 64 + fpos, b_off = (BM + wn * TNn + (lane & 15)) * 64 + fpos;
    const bf16* src[PPW];
    unsigned wpos[PPW];                                // MODE 3: this lane's byte position inside the stage image
    auto set_tile = [&](int t) {
      const int m0 = ((t / w.tiles_n) % g_wrap_rb) * BM, n0 = (t % w.tiles_n) * BN;
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        const int piece = wave * PPW + i, row = piece * 16 + (lane >> 2), c = (lane & 3) ^ ring_f((lane >> 4) & 3);
        src[i] = row < BM ? A + (size_t)min(m0 + row, M - 1) * lda + c * 8 : B + (size_t)min(n0 + row - BM, N - 1) * ldb + c * 8;
        wpos[i] = piece * 1024 + lane * 16;
      }
    };
    f32x4 stg[2][PPW];                                 // MODE 3 staging registers, two K-steps in flight
    auto issue = [&](int kt, int par) {
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        const bool isA = (wave * PPW + i) < 16;        // wave-uniform
        if constexpr (MODE == 0) {
          if (NTA && isA) glds16_p<true>(src[i] + kt * 32, lds0 + (kt % STAGES) * STAGE_BYTES + (wave * PPW + i) * 1024);
          else glds16_p<false>(src[i] + kt * 32, lds0 + (kt % STAGES) * STAGE_BYTES + (wave * PPW + i) * 1024);
        } else {
          if (NTA && isA) gload16<true>(stg[par][i], src[i] + kt * 32);
          else gload16<false>(stg[par][i], src[i] + kt * 32);
        }
      }
    };
    auto land = [&](int kt, int par) {                 // MODE 3: registers of stage kt -> LDS
#pragma unroll
      for (int i = 0; i < PPW; ++i) *(f32x4*)(smem + (kt % STAGES) * STAGE_BYTES + wpos[i]) = stg[par][i];
    };
    if (w.t < w.tend) { set_tile(w.t); issue(0, 0); if (nk > 1) issue(1, 1); }
    while (w.t < w.tend) {
      if constexpr (MODE == 0) { vmwait<0>(); }
      else { vmwait_on<PPW>(stg[0][0], stg[0][1], stg[0][2], stg[0][3]); land(0, 0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
      __builtin_amdgcn_s_barrier();
      if (grpB) __builtin_amdgcn_s_barrier();
      auto kstep = [&](int kt, auto parc) {
        constexpr int par = decltype(parc)::value;     // parity of kt (MODE 3 register sets)
        const bool more = kt + 2 < nk;
        const char* st = smem + (kt % STAGES) * STAGE_BYTES;
        bf16x8 af[MI], bfr[NI];
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < NI; ++j) bfr[j] = FR ? *(const bf16x8*)(st + b_off + j * 1024) : rz;
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = FR ? *(const bf16x8*)(st + a_off + i * 1024) : rz;
        if constexpr (MODE == 3) {
          // stage kt+1 (requested one K-step ago into the other register set) goes to LDS now; then its registers take stage kt+2... no:
          // stage kt+2 goes into THIS parity's set (stage kt left it at the previous step)
          if (kt + 1 < nk) {
            if (more) { issue(kt + 2, par); vmwait_on<PPW>(stg[par ^ 1][0], stg[par ^ 1][1], stg[par ^ 1][2], stg[par ^ 1][3]); }
            else vmwait_on<0>(stg[par ^ 1][0], stg[par ^ 1][1], stg[par ^ 1][2], stg[par ^ 1][3]);
            land(kt + 1, par ^ 1);
          }
        } else if (more) issue(kt + 2, 0);
        __builtin_amdgcn_s_setprio(0);
        if constexpr (MODE == 0) { if (grpB) { if (more) vmwait<PPW>(); else vmwait<0>(); } }
        else if (grpB) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if constexpr (MF) {
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < MI; ++i) acc[i][0][0] += (float)af[i][0] + (float)bfr[i & 3][1];
        }
        if constexpr (MODE == 0) { if (!grpB) { if (more) vmwait<PPW>(); else vmwait<0>(); } }
        else if (!grpB) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      };
      for (int kt = 0; kt < nk; kt += 2) {
        kstep(kt, std::integral_constant<int, 0>());
        if (kt + 1 < nk) kstep(kt + 1, std::integral_constant<int, 1>());
      }
      if (!grpB) __builtin_amdgcn_s_barrier();
      w.t += w.tstride;
      if (w.t < w.tend) { set_tile(w.t); issue(0, 0); if (nk > 1) issue(1, 1); }
    }
  } else if constexpr (MODE == 1 || MODE == 2) {
    // ---------------------------------------------------------------- 4 loader waves + 4 consumer waves, one barrier per K-step
    constexpr int STAGES = 4, AHEAD = (MODE == 1 ? 3 : 2), PPW = 8;      // K-steps in flight beyond the one being multiplied
    const bool loader = wave >= 4;                     // waves w and w + 4 share a SIMD
    if (loader) {
      const int lw = wave - 4;
      const bf16* src[PPW];
      unsigned wpos[PPW];
      auto set_tile = [&](int t) {
        const int m0 = ((t / w.tiles_n) % g_wrap_rb) * BM, n0 = (t % w.tiles_n) * BN;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
          const int piece = lw * PPW + i, row = piece * 16 + (lane >> 2), c = (lane & 3) ^ ring_f((lane >> 4) & 3);
          src[i] = row < BM ? A + (size_t)min(m0 + row, M - 1) * lda + c * 8 : B + (size_t)min(n0 + row - BM, N - 1) * ldb + c * 8;
          wpos[i] = piece * 1024 + lane * 16;
        }
      };
      f32x4 stg[2][PPW];
      auto issue = [&](int kt, int par) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
          const bool isA = lw < 2;
          if constexpr (MODE == 1) {
            if (NTA && isA) glds16_p<true>(src[i] + kt * 32, lds0 + (kt % STAGES) * STAGE_BYTES + (lw * PPW + i) * 1024);
            else glds16_p<false>(src[i] + kt * 32, lds0 + (kt % STAGES) * STAGE_BYTES + (lw * PPW + i) * 1024);
          } else {
            if (NTA && isA) gload16<true>(stg[par][i], src[i] + kt * 32);
            else gload16<false>(stg[par][i], src[i] + kt * 32);
          }
        }
      };
      auto land = [&](int kt, int par) {
        vmwait_on<0>(stg[par][0], stg[par][1], stg[par][2], stg[par][3]);     // (counted form below; this only ties the registers)
        vmwait_on<0>(stg[par][4], stg[par][5], stg[par][6], stg[par][7]);
#pragma unroll
        for (int i = 0; i < PPW; ++i) *(f32x4*)(smem + (kt % STAGES) * STAGE_BYTES + wpos[i]) = stg[par][i];
      };
      while (w.t < w.tend) {
        set_tile(w.t);
        if constexpr (MODE == 1) {
#pragma unroll
          for (int s = 0; s < AHEAD; ++s) if (s < nk) issue(s, 0);
          // stage 0 landed: at most min(AHEAD, nk) - 1 stages behind it
          if (nk >= 3) vmwait<2 * PPW>(); else if (nk == 2) vmwait<PPW>(); else vmwait<0>();
          __builtin_amdgcn_s_barrier();
          for (int kt = 0; kt < nk; ++kt) {
            if (kt + AHEAD < nk) issue(kt + AHEAD, 0);
            const int behind = min(nk - 1, kt + AHEAD) - (kt + 1);       // stages requested after stage kt+1
            if (behind >= 2) vmwait<2 * PPW>(); else if (behind == 1) vmwait<PPW>(); else vmwait<0>();
            __builtin_amdgcn_s_barrier();
          }
        } else {
          // register path: stage s is loaded into set s & 1 and written to LDS one K-step later
          issue(0, 0);
          if (nk > 1) issue(1, 1);
          land(0, 0);                                  // waits for everything (vmcnt(0) tie): stage 1 too, once per tile
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          for (int kt = 0; kt < nk; kt += 2) {
            // even step: stage kt+2 -> set 0 (stage kt is in LDS), stage kt+1 (set 1) -> LDS
            if (kt + 2 < nk) issue(kt + 2, 0);
            if (kt + 1 < nk) {
              if (kt + 2 < nk) { vmwait_on<PPW>(stg[1][0], stg[1][1], stg[1][2], stg[1][3]); vmwait_on<PPW>(stg[1][4], stg[1][5], stg[1][6], stg[1][7]); }
#pragma unroll
              for (int i = 0; i < PPW; ++i) {
                if (kt + 2 >= nk) vmwait_on2<0>(stg[1][i], stg[1][i]);
                *(f32x4*)(smem + ((kt + 1) % STAGES) * STAGE_BYTES + wpos[i]) = stg[1][i];
              }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 1 >= nk) break;
            if (kt + 3 < nk) issue(kt + 3, 1);
            if (kt + 2 < nk) {
              if (kt + 3 < nk) { vmwait_on<PPW>(stg[0][0], stg[0][1], stg[0][2], stg[0][3]); vmwait_on<PPW>(stg[0][4], stg[0][5], stg[0][6], stg[0][7]); }
#pragma unroll
              for (int i = 0; i < PPW; ++i) {
                if (kt + 3 >= nk) vmwait_on2<0>(stg[0][i], stg[0][i]);
                *(f32x4*)(smem + ((kt + 2) % STAGES) * STAGE_BYTES + wpos[i]) = stg[0][i];
              }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
          }
        }
        w.t += w.tstride;
      }
    } else {
      // consumer: 128 x 128 of the tile (2 x 2 consumer waves): 8 A + 8 B fragments, 64 MFMAs on 32 accumulator tiles (each used twice)
      const int wm = wave >> 1, wn = wave & 1;
      const int a_off = (wm * 128 + (lane & 15)) * 64 + fpos, b_off = (BM + wn * 128 + (lane & 15)) * 64 + fpos;
      while (w.t < w.tend) {
        __builtin_amdgcn_s_barrier();                  // stage 0 landed
        for (int kt = 0; kt < nk; ++kt) {
          const char* st = smem + (kt % STAGES) * STAGE_BYTES;
          bf16x8 af[8], bfr[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) bfr[j] = FR ? *(const bf16x8*)(st + b_off + j * 1024) : rz;
#pragma unroll
          for (int i = 0; i < 8; ++i) af[i] = FR ? *(const bf16x8*)(st + a_off + i * 1024) : rz;
          if constexpr (MF) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
              for (int j = 0; j < 8; ++j) acc[i][j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j & 3], 0, 0, 0);
          } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i][0][0] += (float)af[i][0] + (float)bfr[i][1];
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();                // this stage's slot may be refilled; the next stage has landed
        }
        w.t += w.tstride;
      }
    }
  } else {
    // ---------------------------------------------------------------- MODE 4: A straight into operand registers, B by LDS-DMA
    constexpr int STAGES = 3, BST = BN * 64;           // B stage image: 256 rows x 64 B = 16 KiB
    const int b_off = (lane & 15) * 64 + fpos;
    const bf16* bsrc[2];
    const bf16* asrc[2];
    auto set_tile = [&](int t) {
      const int m0 = ((t / w.tiles_n) % g_wrap_rb) * BM, n0 = (t % w.tiles_n) * BN;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int piece = wave * 2 + i, row = piece * 16 + (lane >> 2), c = (lane & 3) ^ ring_f((lane >> 4) & 3);
        bsrc[i] = B + (size_t)min(n0 + row, N - 1) * ldb + c * 8;
        asrc[i] = A + (size_t)min(m0 + wave * 32 + i * 16 + (lane & 15), M - 1) * lda + (lane >> 4) * 8;   // the 16x16x32 A operand layout
      }
    };
    f32x4 areg[3][2];                                  // A fragments of three K-steps
    auto issue = [&](int kt, int set) {                // per K-step and wave: 2 B pieces (DMA) then 2 A fragments (registers): 4 vmcnt units
#pragma unroll
      for (int i = 0; i < 2; ++i) glds16_p<false>(bsrc[i] + kt * 32, lds0 + (kt % STAGES) * BST + (wave * 2 + i) * 1024);
#pragma unroll
      for (int i = 0; i < 2; ++i) gload16<NTA>(areg[set][i], asrc[i] + kt * 32);
    };
    while (w.t < w.tend) {
      set_tile(w.t);
      issue(0, 0);
      if (nk > 1) issue(1, 1);
      auto kstep = [&](int kt, auto setc) {
        constexpr int set = decltype(setc)::value;     // kt % 3
        if (kt + 2 < nk) issue(kt + 2, (set + 2) % 3);
        // stage kt landed (own B pieces + own A registers); up to two younger stages stay in flight
        const int behind = min(nk - 1, kt + 2) - kt;
        if (behind >= 2) vmwait_on2<8>(areg[set][0], areg[set][1]);
        else if (behind == 1) vmwait_on2<4>(areg[set][0], areg[set][1]);
        else vmwait_on2<0>(areg[set][0], areg[set][1]);
        __builtin_amdgcn_s_barrier();                  // everyone's B pieces of stage kt landed; everyone is past the reads of stage kt-1
        const char* st = smem + (kt % STAGES) * BST;
        bf16x8 af[2];
        af[0] = __builtin_bit_cast(bf16x8, areg[set][0]); af[1] = __builtin_bit_cast(bf16x8, areg[set][1]);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const bf16x8 bfr = FR ? *(const bf16x8*)(st + b_off + j * 1024) : rz;
          if constexpr (MF) {
            acc[(j >> 2) * 2][j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bfr, acc[(j >> 2) * 2][j & 3], 0, 0, 0);
            acc[(j >> 2) * 2 + 1][j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bfr, acc[(j >> 2) * 2 + 1][j & 3], 0, 0, 0);
          } else acc[j & 7][0][0] += (float)bfr[0] + (float)af[j & 1][1];
        }
      };
      // a 3-deep ring with ONE barrier per K-step: stage kt+2 is requested before barrier kt into the slot stage kt-1 was read from, and
      // every wave passed barrier kt-1... no: a wave may still be reading stage kt-1 when a faster wave requests stage kt+2 -- close the
      // K-step with a second barrier (cheap: the probe is after the ingest rate, not the last per cent)
      for (int kt = 0; kt < nk; kt += 3) {
        kstep(kt, std::integral_constant<int, 0>()); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier();
        if (kt + 1 < nk) { kstep(kt + 1, std::integral_constant<int, 1>()); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        if (kt + 2 < nk) { kstep(kt + 2, std::integral_constant<int, 2>()); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
      }
      w.t += w.tstride;
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[(size_t)blockIdx.x * 512 + tid] = s;
  __syncthreads();
  if (tid == 0) {
    stamps[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - c0;
    stamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

__global__ void fill_kernel(bf16* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned s = (unsigned)i * 2654435761u + seed; s ^= s >> 15; s *= 2246822519u; s ^= s >> 13;
    p[i] = (bf16)(((int)(s & 4095) - 2048) / 1024.0f);
  }
}



// ---------------------------------------------------------------- mode 5: ingest only, request SHAPE and SOURCE level
// All 8 waves issue 4 LDS-DMA pieces per step (32 KiB per step and CU, 3-deep ring, one barrier per step); a piece = R rows x (1024 / R)
// bytes of a row-major matrix with 2 * K bytes per row.  Tile t covers rows (t % wrap_tiles) * 32 R .. + 32 R: `wrap_tiles` small = every
// CU re-reads an L2-resident set, large = streamed.  BUF: buffer_load_dwordx4 ... lds (raw buffer, offen) instead of global_load_lds.
// WV = number of waves that issue (8: 4 pieces each, 4: 8 pieces each, 2: 16 pieces each).
template <int R, bool BUF, int WV>
__global__ __launch_bounds__(512) void shape_kernel(const bf16* __restrict__ A, int K, int tiles_per_cu, int wrap_tiles, int cu_stride,
                                                    float* __restrict__ out, unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(void, smem));
  constexpr int RB = 1024 / R, LPR = RB / 16, PPW = 32 / WV;      // bytes per row and piece, lanes per row, pieces per issuing wave
  const int nk = (2 * K) / RB;
  unsigned long long c0 = 0, r0 = 0;
  if (tid == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  const bool issuer = wave < WV;
  // raw buffer descriptor over the whole allocation (the host passes a 32-bit-addressable window)
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  u32x4 rsrc;
  rsrc[0] = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)A); rsrc[1] = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)A >> 32));
  rsrc[2] = 0xffffffffu; rsrc[3] = 0x00020000u;
  unsigned off[PPW];
  for (int it = 0; it < tiles_per_cu; ++it) {
    const int t = (int)(((long)blockIdx.x * cu_stride + it) % wrap_tiles);
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int piece = wave * PPW + i, row = (t * 32 + piece) * R + lane / LPR;
      off[i] = (unsigned)row * (unsigned)(2 * K) + (unsigned)(lane % LPR) * 16u;
    }
    auto issue = [&](int kt) {
      if (!issuer) return;
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        const unsigned la = lds0 + (kt % 3) * STAGE_BYTES + (wave * PPW + i) * 1024;
        if constexpr (BUF) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(off[i] + kt * RB), "s"(rsrc), "s"(la) : "memory", "m0");
        else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off[i] + kt * RB), "s"(A), "s"(la) : "memory", "m0");
      }
    };
    issue(0); if (nk > 1) issue(1);
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 2 < nk) { issue(kt + 2); vmwait<2 * PPW>(); } else if (kt + 1 < nk) vmwait<PPW>(); else vmwait<0>();
      __builtin_amdgcn_s_barrier();                    // stage kt landed for everyone
      __builtin_amdgcn_s_barrier();                    // (a consumer would read here) slot free again
    }
  }
  out[(size_t)blockIdx.x * 512 + tid] = *(const float*)(smem + tid * 4);
  __syncthreads();
  if (tid == 0) { stamps[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - c0; stamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
}
typedef void (*ShapeFn)(const bf16*, int, int, int, int, float*, unsigned long long*);
template <int R, bool BUF, int WV> static ShapeFn sinst() {
  ShapeFn f = shape_kernel<R, BUF, WV>;
  HIPCK(hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * STAGE_BYTES));
  return f;
}

struct Cfg { int mode, flags; const char* name; };
typedef void (*KernelFn)(const bf16*, const bf16*, int, int, int, int, float*, unsigned long long*);

template <int MODE, int FLAGS> static KernelFn inst() {
  KernelFn f = ingest_kernel<MODE, FLAGS>;
  HIPCK(hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * STAGE_BYTES));
  return f;
}


// ---------------------------------------------------------------- mode 6: the GEMM's tile walk with WHOLE-LINE pieces
// LDS rows of 128 B (64 k): a block = 256 A rows + 256 B rows x 128 B = 64 KiB feeds two 32-deep K-steps; a piece = 8 rows x 128 B; 16-B chunk c
// of row r at position c ^ ((r >> 1) & 7).  NBLK blocks in the ring (2 = 128 KiB); block j + NBLK - 1 is requested at the start of block j.
// SPLIT: a wave issues half of its 8 pieces in each of the block's two K-steps instead of all in the first.  Schedule of mode 0 (two wave
// groups half a K-step apart).  FLAGS as above.
#define BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
template <int FLAGS, int SPLIT>
__global__ __launch_bounds__(512) void wl_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, int M, int N, int K, int ntiles,
                                                 float* __restrict__ out, unsigned long long* __restrict__ stamps) {
  constexpr bool MF = FLAGS & 1, FR = FLAGS & 2;
  constexpr int NBLK = 2, BLK = 512 * 128, PPW = 8, MI = 8, NI = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(void, smem));
  const int nk = K / 32, nb = K / 64, lda = K, ldb = K;
  Walk w(M, N, ntiles);
  unsigned long long c0 = 0, r0 = 0;
  if (tid == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 rz;
#pragma unroll
  for (int j = 0; j < 8; ++j) rz[j] = (bf16)(0.01f * (float)((lane * 7 + j * 13) % 97) - 0.5f);
  const int wm = wave >> 2, wn = wave & 3;
  const bool grpB = wm >= 1;
  const int fsw = ((lane >> 4) ^ ((lane & 15) >> 1)) << 4;
  const int a_off = (wm * 128 + (lane & 15)) * 128 + fsw, b_off = (256 + wn * 64 + (lane & 15)) * 128 + fsw;
  const bf16* src[PPW];
  auto set_tile = [&](int t) {
    const int m0 = ((t / w.tiles_n) % g_wrap_rb) * BM, n0 = (t % w.tiles_n) * BN;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int piece = wave * PPW + i, row = piece * 8 + (lane >> 3);        // 0..511: A rows then B rows
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      src[i] = row < BM ? A + (size_t)min(m0 + row, M - 1) * lda + c * 8 : B + (size_t)min(n0 + row - BM, N - 1) * ldb + c * 8;
    }
  };
  auto issue = [&](int blk, int i0, int i1) {
#pragma unroll
    for (int i = 0; i < PPW; ++i)
      if (i >= i0 && i < i1) glds16_p<false>(src[i] + blk * 64, lds0 + (blk % NBLK) * BLK + (wave * PPW + i) * 1024);
  };
  if (w.t < w.tend) { set_tile(w.t); issue(0, 0, PPW); }
  while (w.t < w.tend) {
    vmwait<0>();
    BAR();
    if (grpB) BAR();
    for (int kt = 0; kt < nk; ++kt) {
      const int j = kt >> 1, h = kt & 1;
      const bool more = j + 1 < nb;
      const char* st = smem + (j % NBLK) * BLK;
      const int ao = a_off ^ (h << 6), bo = b_off ^ (h << 6);
      bf16x8 af[MI], bfr[NI];
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int jj = 0; jj < NI; ++jj) bfr[jj] = FR ? *(const bf16x8*)(st + bo + jj * 2048) : rz;
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = FR ? *(const bf16x8*)(st + ao + i * 2048) : rz;
      if (more) {
        if (SPLIT) { if (h == 0) issue(j + 1, 0, PPW / 2); else issue(j + 1, PPW / 2, PPW); }
        else if (h == 0) issue(j + 1, 0, PPW);
      }
      __builtin_amdgcn_s_setprio(0);
      if (grpB && h == 1) vmwait<0>();
      BAR();
      if constexpr (MF) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int jj = 0; jj < NI; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[jj], acc[i][jj], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[i][0][0] += (float)af[i][0] + (float)bfr[i & 3][1];
      }
      if (!grpB && h == 1) vmwait<0>();
      BAR();
    }
    if (!grpB) BAR();
    w.t += w.tstride;
    if (w.t < w.tend) { set_tile(w.t); issue(0, 0, PPW); }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[(size_t)blockIdx.x * 512 + tid] = s;
  __syncthreads();
  if (tid == 0) { stamps[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - c0; stamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
}
template <int FLAGS, int SPLIT> static KernelFn winst() {
  KernelFn f = wl_kernel<FLAGS, SPLIT>;
  HIPCK(hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * STAGE_BYTES));
  return f;
}


// ---------------------------------------------------------------- mode 7: whole-line pieces, SEAMLESS ring of five half-block slots
// A half-block = the 256 A rows (or the 256 B rows) x 128 B = 32 KiB = 32 pieces of 8 rows x 128 B; five slots = 160 KiB.  The issue stream
// is A0 B0 A1 | B1 A2 | B2 A3 | ... (slot = position in the stream mod 5) and runs on ACROSS tiles: during block j (K-steps 2j, 2j+1) every
// wave issues its 4 pieces of B(j+1) in the first K-step and of A(j+2) in the second -- 4 pieces per wave and K-step, the L2-resident B rows
// one block ahead, the streamed A rows two.  One counted wait per block (vmcnt(4): the A pieces just issued stay in flight).
// EL / ES: what a DEFERRED fused epilogue would add to the K loop -- per wave and K-step EL streamed 1-KiB loads (8 rows x 128 B of an f32 / bf16
// activation tensor) and ES streamed 1-KiB stores, issued behind the ring pieces in the load phase; the counted waits skip over them
template <int FLAGS, int EL = 0, int ES = 0>
__global__ __launch_bounds__(512) void wl5_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, int M, int N, int K, int ntiles,
                                                  float* __restrict__ out, unsigned long long* __restrict__ stamps) {
  constexpr bool MF = FLAGS & 1, FR = FLAGS & 2;
  constexpr int SLOT = 256 * 128, MI = 8, NI = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(void, smem));
  const int nb = K / 64, lda = K, ldb = K;
  Walk w(M, N, ntiles);
  unsigned long long c0 = 0, r0 = 0;
  if (tid == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 rz;
#pragma unroll
  for (int j = 0; j < 8; ++j) rz[j] = (bf16)(0.01f * (float)((lane * 7 + j * 13) % 97) - 0.5f);
  const int wm = wave >> 2, wn = wave & 3;
  const bool grpB = wm >= 1;
  const int fsw = ((lane >> 4) ^ ((lane & 15) >> 1)) << 4;
  const int a_off = (wm * 128 + (lane & 15)) * 128 + fsw, b_off = (wn * 64 + (lane & 15)) * 128 + fsw;
  // piece i of a wave = rows 8 * wave + 64 * i of the half-block: every piece of a wave has the same swizzle phase, one lane offset
  const unsigned voff = (unsigned)(lane >> 3) * (unsigned)(2 * K) + (unsigned)(((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7)) << 4);
  // issue streams (A and B advance separately, both run on into the next tiles)
  const int wrap_rb = g_wrap_rb;
  int tA = w.t, jA = 0, tB = w.t, jB = 0;
  unsigned sA = 0, sB = 1;                              // slot of the next A / B half-block (stream position mod 5)
  auto rowA = [&](int t) { return (const char*)(A + (size_t)(((t / w.tiles_n) % wrap_rb) * BM + 8 * wave) * lda); };
  const int bstream = g_bstream;
  auto rowB = [&](int t) {
    if (bstream == 1) return (const char*)(A + (size_t)M * 2048 + (size_t)(((t / w.tiles_n) % wrap_rb) * BM + 8 * wave) * ldb);       // second half of the A allocation
    if (bstream == 2) return (const char*)(A + (size_t)M * 2048 + (size_t)((t % (M / BM)) * BM + 8 * wave) * ldb);
    return (const char*)(B + (size_t)((t % w.tiles_n) * BN + 8 * wave) * ldb);
  };
  const char* pA = rowA(tA);                            // wave-uniform: first row of this wave's pieces, advanced 128 B per block
  const char* pB = rowB(tB);
  const size_t step64 = (size_t)64 * lda * 2;           // 64 rows further: the wave's next piece
  auto issueA = [&]() -> bool {
    if (tA >= w.tend) return false;
    const unsigned la = lds0 + sA * SLOT + wave * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16_s(pA + i * step64, voff, la + i * 8192);
    sA = sA >= 3 ? sA - 3 : sA + 2;
    pA += 128;
    if (++jA == nb) { jA = 0; tA += w.tstride; if (tA < w.tend) pA = rowA(tA); }
    return true;
  };
  auto issueB = [&]() -> bool {
    if (tB >= w.tend) return false;
    const unsigned la = lds0 + sB * SLOT + wave * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16_s(pB + i * step64, voff, la + i * 8192);
    sB = sB >= 3 ? sB - 3 : sB + 2;
    pB += 128;
    if (++jB == nb) { jB = 0; tB += w.tstride; if (tB < w.tend) pB = rowB(tB); }
    return true;
  };
  // epilogue-like traffic: every wave streams through its own part of the X (read) and Y (write) buffers, 1 KiB per instruction
  auto uni = [](const char* p) {      // wave-uniform 64-bit pointer in scalar registers
    const unsigned long long v = (unsigned long long)(uintptr_t)p;
    return (const char*)(uintptr_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32)) << 32) | (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)v));
  };
  const char* xp = uni(g_xbuf + ((size_t)blockIdx.x * 8 + wave) * (size_t)(3 << 20));
  const char* yp = uni(g_ybuf + ((size_t)blockIdx.x * 8 + wave) * (size_t)(3 << 20));
  const unsigned xlane = (unsigned)lane * 16u;
  f32x4 xr[EL > 0 ? EL : 1];
  auto extras = [&]() {
#pragma unroll
    for (int e = 0; e < EL; ++e) { asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(xr[e]) : "v"(xlane), "s"(xp) : "memory"); xp += 1024; }
#pragma unroll
    for (int e = 0; e < ES; ++e) { asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(xlane), "v"(xr[EL > 0 ? e % EL : 0]), "s"(yp) : "memory"); yp += 1024; }
  };
  if constexpr (EL == 0 && ES > 0) xr[0] = (f32x4){1.f, 2.f, 3.f, 4.f};
  issueA(); issueB(); issueA();                         // A0 B0 A1
  unsigned ga = 0, gb = 1;                              // slots of the block being multiplied
  vmwait<4>();                                          // A0, B0 landed (A1 may stay in flight)
  BAR();
  if (grpB) BAR();
  while (w.t < w.tend) {
    for (int j = 0; j < nb; ++j, ga = ga >= 3 ? ga - 3 : ga + 2, gb = gb >= 3 ? gb - 3 : gb + 2) {
      const char* sa = smem + ga * SLOT;
      const char* sb = smem + gb * SLOT;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int ao = a_off ^ (h << 6), bo = b_off ^ (h << 6);
        bf16x8 af[MI], bfr[NI];
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int jj = 0; jj < NI; ++jj) bfr[jj] = FR ? *(const bf16x8*)(sb + bo + jj * 2048) : rz;
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = FR ? *(const bf16x8*)(sa + ao + i * 2048) : rz;
        bool issued = false;
        if (h == 0) issueB(); else issued = issueA();
        extras();
        __builtin_amdgcn_s_setprio(0);
        if (grpB && h == 1) { if (issued) vmwait<4 + EL + ES>(); else vmwait<0>(); }
        BAR();
        if constexpr (MF) {
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int jj = 0; jj < NI; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[jj], acc[i][jj], 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < MI; ++i) acc[i][0][0] += (float)af[i][0] + (float)bfr[i & 3][1];
        }
        if (!grpB && h == 1) { if (issued) vmwait<4 + EL + ES>(); else vmwait<0>(); }
        BAR();
      }
    }
    w.t += w.tstride;
  }
  if (!grpB) BAR();
  vmwait<0>();
  float s = EL > 0 ? xr[0][0] * 1e-30f : 0.f;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[(size_t)blockIdx.x * 512 + tid] = s;
  __syncthreads();
  if (tid == 0) { stamps[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - c0; stamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
}
template <int FLAGS, int EL = 0, int ES = 0> static KernelFn w5inst() {
  KernelFn f = wl5_kernel<FLAGS, EL, ES>;
  HIPCK(hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 256 * 128));
  return f;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  const int grid_override = argc > 2 ? atoi(argv[2]) : 0;          // E1: fewer workgroups (64 = 8 CUs per XCD)
  const int wrap_rb = argc > 3 ? atoi(argv[3]) : (1 << 30);          // E2: A row-blocks wrap (32 = 4 per XCD: L2-resident A)
  const int skip_shape = argc > 4 ? atoi(argv[4]) : 0;
  const int bstream = argc > 5 ? atoi(argv[5]) : 0;                  // mode 7 only: B streamed from the second half of the A allocation (1 shared like A, 2 unshared): shapes with K > 2048 are SKIPPED then
  int ncu = 0; HIPCK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
  const int ncu_real = ncu;
  if (argc > 2 && atoi(argv[2]) > 0) ncu = atoi(argv[2]);
  const int M = 262144;
  struct Shape { int N, K; } shapes[] = {{768, 768}, {768, 4096}, {2304, 768}};
  const size_t amax = (size_t)M * 4096, bmax = (size_t)4096 * 4096;
  bf16 *A, *B; float* out; unsigned long long* stamps;
  HIPCK(hipMalloc(&A, amax * 2)); HIPCK(hipMalloc(&B, bmax * 2)); HIPCK(hipMalloc(&out, (size_t)ncu * 512 * 4));
  HIPCK(hipMalloc(&stamps, ncu * 16));
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, A, amax, 1u);
  hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, B, bmax, 77u);
  {   // X / Y buffers of the epilogue-like traffic rows: 3 MiB per wave x 2048 waves = 6 GiB each would be too much: a wave's K-steps move at
      // most (K-steps per CU) x 2 KiB = 1536 x 2 KiB = 3 MiB
    char *xb, *yb;
    HIPCK(hipMalloc(&xb, (size_t)2048 * (3 << 20) + (1 << 20))); HIPCK(hipMalloc(&yb, (size_t)2048 * (3 << 20) + (1 << 20)));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (bf16*)xb, (size_t)2048 * (3 << 20) / 2, 5u);
    HIPCK(hipMemcpyToSymbol(HIP_SYMBOL(g_xbuf), &xb, sizeof(char*))); HIPCK(hipMemcpyToSymbol(HIP_SYMBOL(g_ybuf), &yb, sizeof(char*)));
  }
  HIPCK(hipMemcpyToSymbol(HIP_SYMBOL(g_wrap_rb), &wrap_rb, sizeof(int)));
  HIPCK(hipMemcpyToSymbol(HIP_SYMBOL(g_bstream), &bstream, sizeof(int)));
  HIPCK(hipDeviceSynchronize());
  struct Row { const char* name; KernelFn fn; int dma_frac16; };
  std::vector<Row> rows = {
      {"0 shipped: 8 waves DMA+MFMA (phased)         ", inst<0, 3>(), 0},
      {"0   .. MFMAs off (ingest + fragment reads)   ", inst<0, 2>(), 0},
      {"0   .. MFMAs and fragment reads off (ingest) ", inst<0, 0>(), 0},
      {"0   .. nt on the A stream                    ", inst<0, 7>(), 0},
      {"1 4 loader waves (DMA) + 4 consumers 128x128 ", inst<1, 3>(), 0},
      {"1   .. MFMAs off                             ", inst<1, 2>(), 0},
      {"1   .. MFMAs and fragment reads off (ingest) ", inst<1, 0>(), 0},
      {"1   .. nt on the A stream                    ", inst<1, 7>(), 0},
      {"2 4 loader waves (registers) + 4 consumers   ", inst<2, 3>(), 0},
      {"2   .. MFMAs and fragment reads off (ingest) ", inst<2, 0>(), 0},
      {"3 8 waves through registers (phased)         ", inst<3, 3>(), 0},
      {"3   .. MFMAs and fragment reads off (ingest) ", inst<3, 0>(), 0},
      {"4 A direct to operand registers, B by DMA    ", inst<4, 3>(), 0},
      {"4   .. MFMAs off                             ", inst<4, 2>(), 0},
      {"4   .. MFMAs and fragment reads off (ingest) ", inst<4, 0>(), 0},
      {"4   .. nt on the A stream                    ", inst<4, 7>(), 0},
      {"6 whole-line pieces (8 x 128 B), 2 blocks    ", winst<3, 0>(), 0},
      {"6   .. MFMAs off                             ", winst<2, 0>(), 0},
      {"6   .. MFMAs and fragment reads off (ingest) ", winst<0, 0>(), 0},
      {"6   .. issue split over the block's 2 K-steps", winst<3, 1>(), 0},
      {"6   .. split, MFMAs + fragment reads off     ", winst<0, 1>(), 0},
      {"7 whole lines, seamless 5-slot half-block ring", w5inst<3>(), 1},
      {"7   .. MFMAs off                             ", w5inst<2>(), 1},
      {"7   .. MFMAs and fragment reads off (ingest) ", w5inst<0>(), 1},
      {"7 + 2 KiB loads, 1 KiB stores / wave / K-step ", w5inst<3, 2, 1>(), 1},
      {"7 + 2 KiB loads, 2 KiB stores / wave / K-step ", w5inst<3, 2, 2>(), 1},
      {"7 + 1 KiB loads, 1 KiB stores / wave / K-step ", w5inst<3, 1, 1>(), 1},
      {"7 + 0 KiB loads, 2 KiB stores / wave / K-step ", w5inst<3, 0, 2>(), 1},
      {"7 + 2 KiB loads, 1 KiB stores, MFMAs off      ", w5inst<2, 2, 1>(), 1},
  };
  hipEvent_t e0, e1; HIPCK(hipEventCreate(&e0)); HIPCK(hipEventCreate(&e1));
  std::vector<unsigned long long> hs(ncu * 2);
  printf("ingest probe: %d workgroups on %d CUs, A row-blocks wrap at %d, M = %d, one 512-thread workgroup per CU, 32 KiB of operands per K-step (256 x 256 x 32)\n", ncu, ncu_real, wrap_rb, M);
  for (const Shape& sh : shapes) {
    if (bstream && sh.K > 2048) continue;                           // the streamed B rows would leave the allocation
    const int ntiles = (M / BM) * (sh.N / BN), nk = sh.K / 32;
    const double ksteps_per_cu = (double)ntiles * nk / ncu;
    printf("\nN = %d, K = %d: %d tiles, %d K-steps per tile, %.0f K-steps per CU\n", sh.N, sh.K, ntiles, nk, ksteps_per_cu);
    printf("%-48s %9s %9s %9s %9s %9s %9s\n", "mode", "ms", "GHz", "cyc/Kstep", "B/clk/CU", "GB/s/CU", "GEMM TF/s");
    for (const Row& r : rows) {
      auto go = [&]() { hipLaunchKernelGGL(r.fn, dim3(ncu), dim3(512), r.dma_frac16 ? 5 * 256 * 128 : 4 * STAGE_BYTES, 0, A, B, M, sh.N, sh.K, ntiles, out, stamps); };
      go(); go(); HIPCK(hipDeviceSynchronize());
      HIPCK(hipEventRecord(e0));
      for (int i = 0; i < reps; ++i) go();
      HIPCK(hipEventRecord(e1)); HIPCK(hipEventSynchronize(e1));
      HIPCK(hipGetLastError());
      float ms; HIPCK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
      HIPCK(hipMemcpy(hs.data(), stamps, ncu * 16, hipMemcpyDeviceToHost));
      std::vector<double> cyc, ghz;
      for (int i = 0; i < ncu; ++i) { cyc.push_back((double)hs[2 * i]); ghz.push_back(hs[2 * i + 1] ? (double)hs[2 * i] / ((double)hs[2 * i + 1] * 10.0) : 0.0); }
      std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
      const double mc = cyc[ncu / 2], mg = ghz[ncu / 2];
      const double cyc_k = mc / ksteps_per_cu, bpc = 32768.0 / cyc_k;
      const double gbs = 32768.0 * ksteps_per_cu / (ms * 1e-3) / 1e9;
      const double tf = 2.0 * M * sh.N * sh.K / (ms * 1e-3) / 1e12;
      printf("%-48s %9.3f %9.2f %9.0f %9.1f %9.1f %9.0f\n", r.name, ms, mg, cyc_k, bpc, gbs, tf);
    }
  }
  // ---- mode 5: request shape x source level x CUs
  if (!skip_shape) {
    ncu = ncu_real;
    struct SRow { const char* name; ShapeFn fn; int R; };
    std::vector<SRow> srows = {
        {"16 rows x   64 B (shipped piece)", sinst<16, false, 8>(), 16}, {" 8 rows x  128 B (whole lines)  ", sinst<8, false, 8>(), 8},
        {" 4 rows x  256 B                ", sinst<4, false, 8>(), 4},   {" 1 row  x 1024 B (contiguous)   ", sinst<1, false, 8>(), 1},
        {"16 rows x   64 B, buffer_load   ", sinst<16, true, 8>(), 16},  {" 1 row  x 1024 B, buffer_load   ", sinst<1, true, 8>(), 1},
        {"16 rows x   64 B, 4 waves issue ", sinst<16, false, 4>(), 16}, {"16 rows x   64 B, 2 waves issue ", sinst<16, false, 2>(), 16},
        {" 1 row  x 1024 B, 4 waves issue ", sinst<1, false, 4>(), 1},
    };
    const int K = 768;                                  // 1536-B rows
    struct Src { const char* name; long bytes; } srcs[] = {{"2 MiB set (L2)", 2l << 20}, {"96 MiB set (Infinity Cache)", 96l << 20}, {"1.5 GiB streamed (HBM)", 1536l << 20}};
    for (int grid : {256, 64}) {
      for (const Src& sc : srcs) {
        printf("\nshape probe, %d workgroups, source = %s, rows of %d B\n", grid, sc.name, 2 * K);
        printf("%-36s %9s %9s %9s %9s %9s\n", "piece", "ms", "GHz", "B/clk/CU", "GB/s/CU", "chip TB/s");
        for (const SRow& r : srows) {
          const long tile_bytes = 32l * r.R * 2 * K;     // 32 pieces x R rows x full row
          const int wrap = (int)(sc.bytes / tile_bytes);
          const long per_cu = 48l << 20;                  // bytes each CU takes in per launch
          const int tpc = (int)(per_cu / tile_bytes), stride = sc.bytes > (512l << 20) ? tpc : 7;
          auto go = [&]() { hipLaunchKernelGGL(r.fn, dim3(grid), dim3(512), 3 * STAGE_BYTES, 0, A, K, tpc, wrap, stride, out, stamps); };
          go(); go(); HIPCK(hipDeviceSynchronize());
          HIPCK(hipEventRecord(e0));
          for (int i = 0; i < 5; ++i) go();
          HIPCK(hipEventRecord(e1)); HIPCK(hipEventSynchronize(e1)); HIPCK(hipGetLastError());
          float ms; HIPCK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
          HIPCK(hipMemcpy(hs.data(), stamps, grid * 16, hipMemcpyDeviceToHost));
          std::vector<double> cyc, ghz;
          for (int i = 0; i < grid; ++i) { cyc.push_back((double)hs[2 * i]); ghz.push_back((double)hs[2 * i] / ((double)hs[2 * i + 1] * 10.0)); }
          std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
          const double bytes = (double)tpc * tile_bytes;
          printf("%-36s %9.3f %9.2f %9.1f %9.1f %9.2f\n", r.name, ms, ghz[grid / 2], bytes / cyc[grid / 2], bytes / (ms * 1e-3) / 1e9, bytes * grid / (ms * 1e-3) / 1e12);
        }
      }
    }
  }
  return 0;
}
