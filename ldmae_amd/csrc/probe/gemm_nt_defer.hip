// bf16 NT GEMM with the fused epilogue DEFERRED into the next tile's main loop (gfx950, wave64).
//
// gemm_nt_persist_kernel (gemm.hip) pays every byte of a fused epilogue with the matrix pipe idle: a CU drains its tile at
// 27-38 GB/s (a latency-bound stream of one strip per wave in flight) and nothing else runs on that CU meanwhile -- 13.6 ms per
// train step for the gated-residual, SwiGLU and SwiGLU-backward GEMMs (profiles/r03_epilogue_ablation.txt).  Neither a second
// accumulator set (256 more registers) nor the finished tile in LDS (128 KiB beside a 96-KiB ring) fits the CU, so here the tile
// goes THROUGH L2:
//   * the tile's own ("immediate") epilogue is the plain one -- y = bf16(acc + bias) stored as whole 128-B lines into the tensor
//     that has to be written anyway (y of the gated residual, h12 of SwiGLU, the da half of dh12 for SwiGLU-backward);
//   * the elementwise part (x + gate * y; silu(x1) * x2; the SwiGLU derivative) of tile i runs inside the K loop of tile i + 1 of
//     the same workgroup, as NU "units" per wave -- one unit = 8 rows x 128 B per tensor = one 16-B access per lane -- one unit per
//     K-step: the unit's inputs (y back from L2, xin / h12 from HBM) are requested in the load phase of K-step kt and used in the
//     load phase of K-step kt + 2, the results stored there; the wave's partner on the SIMD runs its 32 MFMAs meanwhile.
// The deferred form is a function of the bf16-ROUNDED product -- what the reference's autocast Linear returns (lightningdit.py:
// 248-249 add `gate * branch(x)` with the branch output in bf16; swiglu_ffn.py:33-36) -- and the fused epilogues of
// gemm_nt_persist_kernel compute the same function, so the two kernels agree bit for bit (tests/test_gpu_kernels.py).
//
// vmcnt bookkeeping.  EVERY vector-memory instruction of the K loop is inline asm -- the ring DMA (glds16_n below), the deferred
// buffer loads and the deferred buffer stores -- so the compiler inserts no vmcnt wait of its own there and all counts are written by
// hand.  (First form, commit "first form": compiler-visible buffer loads.  The compiler counts only what it sees, so its wait for a
// unit was short by the four DMA pieces per step and forced the ring DMA of the PREVIOUS step at the start of the load phase -- half
// a step before the ring needs it: 3-8 % slower than the fused kernels.)  Program order inside a load phase is
//     fragment reads | DMA of stage kt+2 | vmcnt(everything younger than unit kt-2) | arithmetic + stores of unit kt-2 | loads of
//     unit kt | counted ring wait
// so that every deferred access is YOUNGER than the DMA of its step: the ring wait of the next step (vmcnt = everything issued after
// that DMA) does not wait for it, and a deferred access has two K-steps to complete.  All counts are compile-time constants per
// unrolled step (kstep<> below); a count that is too SMALL only over-waits, one that is too LARGE would read a stage (or a unit)
// before it has landed.  The destination registers of an asm load are unprotected until the asm wait that names them ("+v"): the
// steps that carry units are fully unrolled (no loop-carried copy can be made of a register whose data has not landed) and the
// kernel must compile WITHOUT vector-register spills; tools/audit_defer.py checks both in the .s, plus the VMEM instruction count
// between every DMA and its ring wait.
#include <type_traits>

#include "../common.h"
#include "../gemm_nt_common.h"

namespace {

template <int EPI> struct Defer;
template <> struct Defer<LDMAE_EPI_GATE_RES> { static constexpr int NU = 16, NL = 3, NS = 2; };
template <> struct Defer<LDMAE_EPI_SWIGLU> { static constexpr int NU = 8, NL = 2, NS = 1; };
template <> struct Defer<LDMAE_EPI_SWIGLU_BWD> { static constexpr int NU = 16, NL = 3, NS = 2; };

// registers of one unit in flight
template <int EPI> struct Unit;
template <> struct Unit<LDMAE_EPI_GATE_RES> { bf16x8 y; f32x4 xa, xb; };
template <> struct Unit<LDMAE_EPI_SWIGLU> { bf16x8 x1, x2; };
template <> struct Unit<LDMAE_EPI_SWIGLU_BWD> { bf16x8 g, a, b; };
// results of one unit, between its arithmetic and its stores
template <int EPI> struct Res;
template <> struct Res<LDMAE_EPI_GATE_RES> { f32x4 a, b; };
template <> struct Res<LDMAE_EPI_SWIGLU> { bf16x8 h; };
template <> struct Res<LDMAE_EPI_SWIGLU_BWD> { bf16x8 da, db; };

template <int N> __device__ __forceinline__ void vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

typedef __attribute__((ext_vector_type(4))) int i32x4;
// buffer descriptor in four scalar registers: base, stride 0, 2 GiB of records (offsets here stay inside one tile's rows), raw dword format
__device__ __forceinline__ i32x4 make_srd(const void* p) {
  const unsigned long long a = (unsigned long long)(uintptr_t)p;
  return (i32x4){(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), 0x7FFFFFFF, 0x00020000};
}
// asm VMEM: invisible to the compiler's waitcnt pass.  `s_nop 4`: a scalar operand may have been written by a VALU instruction right in
// front of the statement (v_readlane of a spilled SGPR, v_readfirstlane) -- 5 wait states before a VMEM instruction reads it, and the
// compiler pads nothing inside asm.  Stores end with `s_nop 1` (the data registers must stay put until the store has read them).
template <typename V> __device__ __forceinline__ void bload(V& d, unsigned voff, i32x4 srd, unsigned soff) {
  asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(d) : "v"(voff), "s"(srd), "s"(soff) : "memory");
}
template <typename V> __device__ __forceinline__ void bstore(const V& d, unsigned voff, i32x4 srd, unsigned soff) {
  asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(d), "v"(voff), "s"(srd), "s"(soff) : "memory");
}
template <typename V> __device__ __forceinline__ void bstore_nt(const V& d, unsigned voff, i32x4 srd, unsigned soff) {
  asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen nt\n\ts_nop 1" ::"v"(d), "v"(voff), "s"(srd), "s"(soff) : "memory");
}
__device__ __forceinline__ void glds16_n(const void* sbase, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}

using T_ = std::true_type;
using F_ = std::false_type;

}  // namespace

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// DBG (diagnostic build, tune key 13): timing-only ablations with wrong results -- 1 = no deferred work at all, 2 = the unit loads only,
// 3 = loads + arithmetic without the stores; the wait counts follow.
template <int EPI, int DBG = 0>
__global__ __launch_bounds__(512) void gemm_nt_defer_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, int M, int N, int K, int lda,
                                                            int ldb, EpiArgs e, int ntiles) {
  constexpr int STAGES = 3, BM = 256, BN = 256, WN = 4, NW = 8, TM = 128, TNn = 64, MI = 8, NI = 4;
  constexpr int STAGE_BYTES = (BM + BN) * 64, PPW = (BM + BN) / 16 / NW;
  constexpr int NU = Defer<EPI>::NU, NL = DBG == 1 ? 0 : Defer<EPI>::NL, NS = DBG == 0 ? Defer<EPI>::NS : 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const unsigned tiles_n = N / BN;
  float* ew = (float*)(smem + STAGES * STAGE_BYTES) + wave * (16 * 68);
  // same tile walk as gemm_nt_persist_kernel: XCD x owns a contiguous range of A row-blocks
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  const int rbx = ((M / BM) + 7) / 8;
  const int first = xcd * rbx * (int)tiles_n + slot;
  const int tend = min(ntiles, (xcd + 1) * rbx * (int)tiles_n);
  const int tstride = per_xcd;
  const int Hs = EPI == LDMAE_EPI_SWIGLU ? (N >> 1) : N;

  // Ring DMA with wave-uniform bases: waves 0-3 bring the tile's A rows (64 each), waves 4-7 its B rows; a piece = 16 rows x 64 B, so the
  // four pieces of a wave are 16 rows apart and share ONE 32-bit lane offset (row-in-piece * ld + swizzled chunk): no 64-bit address
  // lives in a vector register (eight of them, spilled around the K-steps, drained the ring at every reload)
  const bool isA = wave < 4;
  const unsigned ld_op = isA ? (unsigned)lda : (unsigned)ldb;
  const unsigned dma_voff = ((unsigned)(lane >> 2) * ld_op + (unsigned)(((lane & 3) ^ ring_f((lane >> 4) & 3)) * 8)) * 2u;
  const size_t piece_step = (size_t)16 * ld_op * 2;
  const char* dma_base = nullptr;
  int m0 = 0, n0 = 0;
  auto set_tile = [&](int t) {
    m0 = (t / tiles_n) * BM; n0 = (t % tiles_n) * BN;
    if (isA) dma_base = (const char*)(A + (size_t)(m0 + wave * 64) * lda);
    else {
      int brow = n0 + (wave - 4) * 64;
      // SwiGLU: the tile's 256 columns = x1 columns n0/2 .. n0/2+127 | the x2 columns of the same hidden units: a wave's 64 columns stay
      // contiguous in h12 (whole-line stores) and both factors of a hidden unit are produced by one workgroup
      if constexpr (EPI == LDMAE_EPI_SWIGLU) brow = (wave < 6 ? 0 : Hs - 128) + (n0 >> 1) + (wave - 4) * 64;
      dma_base = (const char*)(B + (size_t)brow * ldb);
    }
  };
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(void, smem)) + wave * PPW * 1024;
  auto issue = [&](int kt) {
    const unsigned la = lds0 + (kt % STAGES) * STAGE_BYTES;
    const char* gb = dma_base + (size_t)kt * 64;
#pragma unroll
    for (int i = 0; i < PPW; ++i) glds16_n(gb + i * piece_step, dma_voff, la + i * 1024);
  };
  const int fpos = ((lane >> 4) ^ ring_f((lane >> 2) & 3)) << 4;
  const int a_off = (wm * TM + (lane & 15)) * 64 + fpos, b_off = (BM + wn * TNn + (lane & 15)) * 64 + fpos;
  const int nk = K / 32;
  // (scalar compares at every site: held as a lane mask the flag and its negation cost four scalar registers and, under pressure, a vector one)
#define grpB (wm != 0)
#define grpA (wm == 0)

  // ---------------------------------------------------------------- deferred units of the PREVIOUS tile (pm0, pn0)
  // buffer addressing: descriptor (scalar registers, base = the wave's first row of the tile) + one 32-bit lane offset per tensor
  // + a scalar unit offset
  // (the descriptors are put together where they are used: two live scalar registers per tensor instead of four)
  const char* rl0 = nullptr; const char* rl1 = nullptr; const char* rs0 = nullptr;   // two load tensors, one store tensor
  // bytes per unit (scalars) and the per-lane byte offsets -- the latter RECOMPUTED from the lane id where they are used (a handful of
  // vector instructions in a load phase that has issue slots to spare) instead of living in registers across the MFMA phases
  unsigned lstep = 0, sstep = 0, lrow = 0, srow = 0, lsh = 0, ssh = 0;     // row strides in bytes / log2 bytes per 8-column group
  if constexpr (EPI == LDMAE_EPI_GATE_RES) {
    lrow = (unsigned)e.ldc * 2u; srow = (unsigned)N * 4u; lsh = 4; ssh = 5;                         // y (bf16, row stride ldc); xin / xout (f32)
  } else if constexpr (EPI == LDMAE_EPI_SWIGLU) {
    lrow = (unsigned)N * 2u; srow = (unsigned)Hs * 2u; lsh = 4; ssh = 4;                            // h12 rows (2 Hs columns); hid rows
  } else {
    lrow = srow = (unsigned)(2 * Hs) * 2u; lsh = ssh = 4;                                           // h12 / dh12 rows
  }
  lstep = 8u * lrow; sstep = 8u * srow;
  // (asm volatile: the builtin form is loop-invariant to the compiler, which hoists it -- and everything derived from it -- back out)
  auto cur_lane = [] {
    unsigned l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
  };
  auto mk_loff = [&] { const unsigned l = cur_lane(); return (l >> 3) * lrow + ((l & 7u) << lsh); };
  auto mk_soff = [&] { const unsigned l = cur_lane(); return (l >> 3) * srow + ((l & 7u) << ssh); };
  auto gslot = [&] { return ew + cur_lane() * 8; };           // gated residual: this lane's 8 gate values of the previous tile (strip area: idle in the K loop)
  auto set_prev = [&](int pm0, int pn0, float4 ga, float4 gb) {
    if constexpr (EPI == LDMAE_EPI_GATE_RES) {
      const size_t pm = pm0 + wm * TM, pc = pn0 + wn * TNn;
      rl0 = (const char*)e.C + (pm * e.ldc + pc) * 2;        // y
      rl1 = (const char*)e.xin + (pm * N + pc) * 4;         // xin
      rs0 = (const char*)e.xout + (pm * N + pc) * 4;        // xout
      float* gsl = gslot();
      *(float4*)gsl = ga; *(float4*)(gsl + 4) = gb;
    } else if constexpr (EPI == LDMAE_EPI_SWIGLU) {
      const size_t pm = pm0 + (wave >> 1) * 64, hc = (pn0 >> 1) + (wave & 1) * 64;
      rl0 = (const char*)e.C + (pm * N + hc) * 2;           // x1 (x2 at + Hs)
      rl1 = rl0;
      rs0 = (const char*)e.xout + (pm * Hs + hc) * 2;       // hid
    } else {
      const size_t pm = pm0 + wm * TM, pc = pn0 + wn * TNn;
      rl0 = (const char*)e.xin + (pm * 2 * Hs + pc) * 2;    // a (b at + Hs)
      rl1 = (const char*)e.C + (pm * 2 * Hs + pc) * 2;      // g = dhid as stored by the immediate epilogue
      rs0 = rl1;                                                                                                       // da goes back there (db at + Hs)
    }
  };
  auto uload = [&](Unit<EPI>& u, int i) {
    const unsigned loff = mk_loff();
    if constexpr (EPI == LDMAE_EPI_GATE_RES) {
      const unsigned soff = mk_soff();
      bload(u.y, loff, make_srd(rl0), i * lstep);
      bload(u.xa, soff, make_srd(rl1), i * sstep); bload(u.xb, soff, make_srd(rl1), i * sstep + 16);
    } else if constexpr (EPI == LDMAE_EPI_SWIGLU) {
      bload(u.x1, loff, make_srd(rl0), i * lstep); bload(u.x2, loff, make_srd(rl0), i * lstep + Hs * 2);
    } else {
      bload(u.g, loff, make_srd(rl1), i * lstep);
      bload(u.a, loff, make_srd(rl0), i * lstep); bload(u.b, loff, make_srd(rl0), i * lstep + Hs * 2);
    }
  };
  // the unit's registers become readable here: everything but the `nyoung` youngest vector-memory instructions has completed
  auto uwait = [&](auto nyoung, Unit<EPI>& u) {
    constexpr int NY = decltype(nyoung)::value;
    if constexpr (EPI == LDMAE_EPI_GATE_RES) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(u.y), "+v"(u.xa), "+v"(u.xb) : "n"(NY) : "memory");
    else if constexpr (EPI == LDMAE_EPI_SWIGLU) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(u.x1), "+v"(u.x2) : "n"(NY) : "memory");
    else asm volatile("s_waitcnt vmcnt(%3)" : "+v"(u.g), "+v"(u.a), "+v"(u.b) : "n"(NY) : "memory");
  };
  auto ucalc = [&](const Unit<EPI>& u) -> Res<EPI> {
    Res<EPI> r;
    if constexpr (EPI == LDMAE_EPI_GATE_RES) {
      const float* gsl = gslot();
      const float4 g0 = *(const float4*)gsl, g1 = *(const float4*)(gsl + 4);
      const float4 ra = gate_res4(make_float4(u.xa[0], u.xa[1], u.xa[2], u.xa[3]), g0, make_float4((float)u.y[0], (float)u.y[1], (float)u.y[2], (float)u.y[3]));
      const float4 rb = gate_res4(make_float4(u.xb[0], u.xb[1], u.xb[2], u.xb[3]), g1, make_float4((float)u.y[4], (float)u.y[5], (float)u.y[6], (float)u.y[7]));
      r.a = (f32x4){ra.x, ra.y, ra.z, ra.w}; r.b = (f32x4){rb.x, rb.y, rb.z, rb.w};
    } else if constexpr (EPI == LDMAE_EPI_SWIGLU) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float a = (float)u.x1[j]; r.h[j] = (bf16)(a * fast_sigmoid(a) * (float)u.x2[j]); }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float g = (float)u.g[j], a = (float)u.a[j], b = (float)u.b[j], sg = fast_sigmoid(a);
        r.da[j] = (bf16)(g * b * sg * (1.f + a * (1.f - sg)));
        r.db[j] = (bf16)(g * a * sg);
      }
    }
    return r;
  };
  auto ustore = [&](const Res<EPI>& r, int i) {
    const unsigned soff = mk_soff();
    if constexpr (EPI == LDMAE_EPI_GATE_RES) {
      bstore(r.a, soff, make_srd(rs0), i * sstep); bstore(r.b, soff, make_srd(rs0), i * sstep + 16);
    } else if constexpr (EPI == LDMAE_EPI_SWIGLU) {
      bstore_nt(r.h, soff, make_srd(rs0), i * sstep);
    } else {
      bstore_nt(r.da, soff, make_srd(rs0), i * sstep); bstore_nt(r.db, soff, make_srd(rs0), i * sstep + Hs * 2);
    }
  };

  // arithmetic + stores of one unit.  Gated residual: in two halves, each ending in its (asm, "memory") store, so that only four gate values
  // and four products are live at a time (all eight kept the unit's registers from fitting beside the MFMA phase's fragments)
  auto ufinish = [&](Unit<EPI>& u, int i) {
    if constexpr (EPI == LDMAE_EPI_GATE_RES) {
      const unsigned soff = mk_soff();
      const float* gsl = gslot();
      {
        const float4 g = *(const float4*)gsl;
        const float4 r = gate_res4(make_float4(u.xa[0], u.xa[1], u.xa[2], u.xa[3]), g, make_float4((float)u.y[0], (float)u.y[1], (float)u.y[2], (float)u.y[3]));
        bstore((f32x4){r.x, r.y, r.z, r.w}, soff, make_srd(rs0), i * sstep);
      }
      {
        const float4 g = *(const float4*)(gsl + 4);
        const float4 r = gate_res4(make_float4(u.xb[0], u.xb[1], u.xb[2], u.xb[3]), g, make_float4((float)u.y[4], (float)u.y[5], (float)u.y[6], (float)u.y[7]));
        bstore((f32x4){r.x, r.y, r.z, r.w}, soff, make_srd(rs0), i * sstep + 16);
      }
    } else {
      ustore(ucalc(u), i);
    }
  };

  // ---------------------------------------------------------------- main loop pieces
  f32x4 acc[MI][NI];
  Unit<EPI> bufA, bufB;
  // one K-step.  C: the unit requested two steps ago is used and stored; L: a unit is requested; PC / PL: the PREVIOUS step did so
  // (its stores / loads were issued after its DMA and are still allowed in flight at this step's ring wait).  Used for the steps
  // kt < NU + 3, all of which have two more stages to request (nk >= NU + 8, host check).
  auto kstep = [&](auto tC, auto tL, auto tPC, auto tPL, Unit<EPI>& buf, int kt) {
    constexpr bool C = decltype(tC)::value, L = decltype(tL)::value, PC = decltype(tPC)::value, PL = decltype(tPL)::value;
    // ring-wait counts: everything issued after the DMA of stage kt+1 (previous step).  Group B waits at the end of its load phase, BEFORE
    // this step's deferred accesses; group A after its MFMA phase and after them.
    constexpr int NWAIT_B = (PC ? NS : 0) + (PL ? NL : 0) + PPW, NWAIT_A = NWAIT_B + (C ? NS : 0) + (L ? NL : 0);
    const char* st = smem + (kt % STAGES) * STAGE_BYTES;
    bf16x8 af[MI], bfr[NI];
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int j = 0; j < NI; ++j) bfr[j] = *(const bf16x8*)(st + b_off + j * 1024);
#pragma unroll
    for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8*)(st + a_off + i * 1024);
    issue(kt + STAGES - 1);
    __builtin_amdgcn_s_setprio(0);
    if (grpB) vmwait<NWAIT_B>();
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    // The deferred accesses sit IN the wave's MFMA phase (the compiler interleaves them with the MFMAs).  This loop is bound by its load
    // phases (DMA issue and landing: ~770 cycles per phase against 512 of MFMA issue), so a wave that has issued its 32 MFMAs idles at the
    // barrier while its partner loads -- that idle time, not the load phase, is where the unit's five memory instructions and ~35 vector
    // instructions are free.  (In the load phase they cost their full issue time: 3-8 % slower than the fused kernels, measured.)
    if constexpr (C && DBG != 1) {
      // younger than the unit's loads (step kt-2): the DMA of step kt-1, that step's stores and loads, this step's DMA
      uwait(std::integral_constant<int, 2 * PPW + (PC ? NS : 0) + (PL ? NL : 0)>{}, buf);
      if constexpr (DBG == 0) ufinish(buf, kt - 2);
      if constexpr (DBG == 3) {
        if constexpr (EPI == LDMAE_EPI_SWIGLU) { const Res<EPI> r = ucalc(buf); asm volatile("" ::"v"(r.h)); }
        else if constexpr (EPI == LDMAE_EPI_GATE_RES) { const Res<EPI> r = ucalc(buf); asm volatile("" ::"v"(r.a), "v"(r.b)); }
      }
    }
    if constexpr (L && DBG != 1) uload(buf, kt);
    if (grpA) vmwait<NWAIT_A>();
    __builtin_amdgcn_s_barrier();
  };
  // plain K-steps kt0 .. nk-1 (no deferred access in them or in the step before kt0)
  auto plain_steps = [&](int kt0) {
    // opaque copies: whatever address registers the compiler derives for this loop are made here, per tile, not hoisted out of the tile
    // loop and kept (spilled) across the peeled steps
    const int pl = (int)cur_lane();
    const int pf = ((pl >> 4) ^ ring_f((pl >> 2) & 3)) << 4;
    const int ao = (wm * TM + (pl & 15)) * 64 + pf, bo = (BM + wn * TNn + (pl & 15)) * 64 + pf;
    for (int kt = kt0; kt < nk; ++kt) {
      const char* st = smem + (kt % STAGES) * STAGE_BYTES;
      bf16x8 af[MI], bfr[NI];
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int j = 0; j < NI; ++j) bfr[j] = *(const bf16x8*)(st + bo + j * 1024);
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8*)(st + ao + i * 1024);
      if (kt + STAGES - 1 < nk) issue(kt + STAGES - 1);
      __builtin_amdgcn_s_setprio(0);
      if (grpB) { if (kt + 2 < nk) vmwait<PPW>(); else if (kt + 1 < nk) vmwait<0>(); }
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      if (grpA) { if (kt + 2 < nk) vmwait<PPW>(); else if (kt + 1 < nk) vmwait<0>(); }
      __builtin_amdgcn_s_barrier();
    }
  };

  int t = first;
  if (t < tend) {
    set_tile(t);
    issue(0); issue(1);
  }
  bool have_prev = false;
  while (t < tend) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // tile start (gemm_nt_persist_kernel): everything this wave has issued is complete -- its share of the previous tile's immediate
    // epilogue included, and behind the barrier every other wave's share too: the deferred units may read that tile
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_barrier();
    if (grpB) __builtin_amdgcn_s_barrier();
    if (have_prev) {
      kstep(F_{}, T_{}, F_{}, F_{}, bufA, 0);
      kstep(F_{}, T_{}, F_{}, T_{}, bufB, 1);
      kstep(T_{}, T_{}, F_{}, T_{}, bufA, 2);
      kstep(T_{}, T_{}, T_{}, T_{}, bufB, 3);
#pragma unroll 1
      for (int kt = 4; kt < NU; kt += 2) {
        kstep(T_{}, T_{}, T_{}, T_{}, bufA, kt);
        kstep(T_{}, T_{}, T_{}, T_{}, bufB, kt + 1);
      }
      kstep(T_{}, F_{}, T_{}, T_{}, bufA, NU);
      kstep(T_{}, F_{}, T_{}, F_{}, bufB, NU + 1);
      kstep(F_{}, F_{}, T_{}, F_{}, bufA, NU + 2);
      plain_steps(NU + 3);
    } else {
      plain_steps(0);
    }
    if (grpA) __builtin_amdgcn_s_barrier();             // every wave is past its last fragment read: the ring is free
    const int em0 = m0, en0 = n0;
    t += tstride;
    if (t < tend) {
      set_tile(t);
      issue(0); issue(1);
    }
    // gate values of this tile for its deferred units: requested ahead of the store tail (vmcnt retires in order), parked in LDS after it
    float4 ga = make_float4(1.f, 1.f, 1.f, 1.f), gb = ga;
    if constexpr (EPI == LDMAE_EPI_GATE_RES) {
      if (e.gate) {
        const float* gp = e.gate + (size_t)(em0 / e.rows_per_batch) * e.gate_ld + en0 + wn * TNn + (cur_lane() & 7) * 8;
        ga = *(const float4*)gp; gb = *(const float4*)(gp + 4);
      }
    }
    // immediate epilogue: the plain one, into the tensor the deferred units read back
    {
      EpiArgs e2{};
      e2.C = e.C; e2.ldc = e.ldc; e2.bias = e.bias;
      int en = en0;
      if constexpr (EPI == LDMAE_EPI_SWIGLU) en = (wn < 2 ? 0 : Hs - 128) + (en0 >> 1);      // + wn * 64 = the wave's h12 column
      // the epilogue's lane-derived addresses are recomputed per tile (opaque copy of the lane id): hoisted out of the tile loop they would
      // be live -- i.e. spilled -- across every K-step
      const int lane_e = (int)cur_lane();
      nt_epilogue<LDMAE_EPI_BIAS, bf16, TM, TNn, MI, NI>(acc, ew, ew, e2, em0, en, wm, wn, lane_e, M, EPI == LDMAE_EPI_SWIGLU_BWD ? 2 * Hs : N);
    }
    set_prev(em0, en0, ga, gb);
    have_prev = true;
  }
  // the last tile's units: nothing left to hide them under
  if (have_prev && DBG == 0) {
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_barrier();
    for (int i = 0; i < NU; ++i) {
      Unit<EPI> u;
      uload(u, i);
      uwait(std::integral_constant<int, 0>{}, u);
      ufinish(u, i);
    }
    vmwait<0>();
  }
}

#undef grpB
#undef grpA

// 0 = shape / arguments outside what the deferred kernel covers (the caller then launches gemm_nt_persist_kernel)
int ldmae_launch_nt_defer(int epi, const void* A, const void* B, int M, int N, int K, int lda, int ldb, const EpiArgs& e, int grid, int ntiles,
                          hipStream_t st) {
  if (M % 256 || N % 256 || K % 32 || grid == ntiles || (grid & 7)) return 0;
  if (((uintptr_t)e.C | (uintptr_t)e.xin | (uintptr_t)e.xout | (uintptr_t)e.bias) & 15) return 0;
  const int nk = K / 32;
  constexpr int lds = 3 * 512 * 64 + 8 * 16 * 68 * 4 + 2048;
  if (epi == LDMAE_EPI_GATE_RES) {
    if (!e.C || e.ldc % 8 || nk < 24) return 0;
    if (e.gate && (e.rows_per_batch % 256 || ((uintptr_t)e.gate & 15) || e.gate_ld % 4)) return 0;
#define DEFER_GO(E, D)                                                                                                          \
  {                                                                                                                            \
    hipFuncSetAttribute((const void*)gemm_nt_defer_kernel<E, D>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);              \
    hipLaunchKernelGGL((gemm_nt_defer_kernel<E, D>), dim3(grid), dim3(512), lds, st, (const bf16*)A, (const bf16*)B, M, N, K, lda, ldb, e, ntiles); \
  }
#ifdef LDMAE_DIAG
#define DEFER_LAUNCH(E)                                                                     \
  switch (ldmae_tune_get(13)) {                                                             \
    case 1: DEFER_GO(E, 1); break;                                                          \
    case 2: DEFER_GO(E, 2); break;                                                          \
    case 3: DEFER_GO(E, 3); break;                                                          \
    default: DEFER_GO(E, 0); break;                                                         \
  }
#else
#define DEFER_LAUNCH(E) DEFER_GO(E, 0)
#endif
    DEFER_LAUNCH(LDMAE_EPI_GATE_RES);
    return 1;
  }
  if (epi == LDMAE_EPI_SWIGLU) {
    if (!e.C || e.ldc != N || nk < 16) return 0;
    DEFER_LAUNCH(LDMAE_EPI_SWIGLU);
    return 1;
  }
  return 0;
}
