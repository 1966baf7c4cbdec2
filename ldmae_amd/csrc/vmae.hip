// VMAE masked-token encoder kernels (tokenizer/models_mae.py): random_masking as an in-LDS bitonic
// sort of (noise, index) keys -- integer / index work, bit-exact against the oracle --, token
// gather / scatter, affine LayerNorm and exact-erf GELU.
#include "common.h"

// order-preserving map f32 -> u32 (handles negatives; noise is in [0,1) but be general)
__device__ __forceinline__ unsigned f32_ordered(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// One workgroup per sample.  Sorts 64-bit keys (ordered(noise) << 32 | index) ascending: equal noise
// values are ordered by index == a stable argsort (models_mae.py:484).  L padded to a power of two LP.
__global__ __launch_bounds__(256) void random_masking_kernel(const float* __restrict__ noise, long long* __restrict__ ids_restore,
                                                             float* __restrict__ mask, long long* __restrict__ ids_keep, int L, int LP, int keep) {
  extern __shared__ unsigned long long keys[];
  const int n = blockIdx.x;
  for (int i = threadIdx.x; i < LP; i += 256)
    keys[i] = i < L ? (((unsigned long long)f32_ordered(noise[(size_t)n * L + i]) << 32) | (unsigned)i) : ~0ull;
  __syncthreads();
  for (int k = 2; k <= LP; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < LP; i += 256) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = keys[i], b = keys[ixj];
          const bool up = (i & k) == 0;
          if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
        }
      }
      __syncthreads();
    }
  // keys[j] low word = ids_shuffle[j];  ids_restore[ids_shuffle[j]] = j;  mask = (rank >= keep)
  for (int j = threadIdx.x; j < L; j += 256) {
    const unsigned idx = (unsigned)keys[j];
    ids_restore[(size_t)n * L + idx] = j;
    mask[(size_t)n * L + idx] = j < keep ? 0.f : 1.f;
    if (j < keep) ids_keep[(size_t)n * keep + j] = idx;
  }
}

extern "C" int ldmae_random_masking(const float* noise, long long* ids_restore, float* mask, long long* ids_keep, int N, int L, int keep,
                                    void* stream) {
  LDMAE_REQUIRE(noise && ids_restore && mask && ids_keep, "random_masking: null pointer");
  LDMAE_REQUIRE(N > 0 && L > 0 && L <= 4096 && keep >= 0 && keep <= L, "random_masking: bad shape N=%d L=%d keep=%d (L <= 4096)", N, L, keep);
  int LP = 1;
  while (LP < L) LP <<= 1;
  hipLaunchKernelGGL(random_masking_kernel, dim3(N), dim3(256), (size_t)LP * 8, as_stream(stream), noise, ids_restore, mask, ids_keep, L, LP, keep);
  LDMAE_CHECK_LAUNCH("random_masking");
  return LDMAE_OK;
}

// out[n, j, :] = x[n, ids[n, j], :]
__global__ void gather_rows_kernel(const float* __restrict__ x, const long long* __restrict__ ids, float* __restrict__ out, int L, int keep, int D) {
  const int n = blockIdx.y, j = blockIdx.x;
  const long long src = ids[(size_t)n * keep + j];
  const float4* s = (const float4*)(x + ((size_t)n * L + src) * D);
  float4* d = (float4*)(out + ((size_t)n * keep + j) * D);
  for (int i = threadIdx.x; i < D / 4; i += blockDim.x) d[i] = s[i];
}
// dx[n, ids[n, j], :] = dout[n, j, :]   (dx must be zeroed by the caller; ids unique per sample)
__global__ void scatter_rows_kernel(const float* __restrict__ dout, const long long* __restrict__ ids, float* __restrict__ dx, int L, int keep, int D) {
  const int n = blockIdx.y, j = blockIdx.x;
  const long long dst = ids[(size_t)n * keep + j];
  const float4* s = (const float4*)(dout + ((size_t)n * keep + j) * D);
  float4* d = (float4*)(dx + ((size_t)n * L + dst) * D);
  for (int i = threadIdx.x; i < D / 4; i += blockDim.x) d[i] = s[i];
}
// Patch-embed input of the KEPT tokens only (inference: the mask depends on the noise alone, so masking can run before the embedding and the
// conv GEMM sees a quarter of the patches at mask_ratio 0.75): tok[n*keep + j, (c, i, jj)] = img[n, c, ph*p + i, pw*p + jj] for patch
// ids[n, j] = ph * grid + pw, in the conv weight's (c, i, jj) order (models_mae.py:342 PatchEmbed), and posg[n*keep + j, :] = pos[ids[n, j], :].
template <typename T>
__global__ void patch_gather_kernel(const float* __restrict__ img, const long long* __restrict__ ids, const float* __restrict__ pos,
                                    T* __restrict__ tok, float* __restrict__ posg, int keep, int C, int S, int p, int D) {
  const int n = blockIdx.y, j = blockIdx.x, grid = S / p, K = C * p * p;
  const int idx = (int)ids[(size_t)n * keep + j], ph = idx / grid, pw = idx % grid;
  const size_t row = (size_t)n * keep + j;
  for (int e = threadIdx.x; e < K; e += blockDim.x) {
    const int c = e / (p * p), i = (e / p) % p, jj = e % p;
    tok[row * K + e] = from_f<T>(img[(((size_t)n * C + c) * S + ph * p + i) * S + pw * p + jj]);
  }
  for (int d = threadIdx.x * 4; d < D; d += blockDim.x * 4) *(float4*)(posg + row * D + d) = *(const float4*)(pos + (size_t)idx * D + d);
}
extern "C" int ldmae_patch_gather(int tok_dtype, const float* img, const long long* ids, const float* pos, void* tok, float* posg, int N, int keep,
                                  int C, int S, int p, int D, void* stream) {
  LDMAE_REQUIRE(img && ids && pos && tok && posg && N > 0 && keep > 0, "patch_gather: null pointer or empty input");
  LDMAE_REQUIRE(p > 0 && S % p == 0 && D % 4 == 0, "patch_gather: image size %d must be a multiple of the patch size %d, D=%d of 4", S, p, D);
  if (tok_dtype == LDMAE_BF16) hipLaunchKernelGGL(patch_gather_kernel<bf16>, dim3(keep, N), dim3(64), 0, as_stream(stream), img, ids, pos, (bf16*)tok, posg, keep, C, S, p, D);
  else hipLaunchKernelGGL(patch_gather_kernel<float>, dim3(keep, N), dim3(64), 0, as_stream(stream), img, ids, pos, (float*)tok, posg, keep, C, S, p, D);
  LDMAE_CHECK_LAUNCH("patch_gather");
  return LDMAE_OK;
}

// Latent-dataset prologue (datasets/img_latent_dataset.py:79-93 of the reference, per batch on the device instead of per item on the host):
// moments [B, 2C, HW] (mean | logvar halves, what extract_features.py stores under data.sample) or plain latents [B, C, HW] ->
// x[b, c, :] = ((mean + exp(0.5 * clamp(logvar, -30, 20)) * noise) - lat_mean[c]) / lat_std[c] * multiplier, the model input.
// One pass: 12 B read + 4 B written per element (HBM-bound, 16 B per lane).
__global__ void latent_prologue_kernel(const float* __restrict__ mom, const float* __restrict__ noise, const float* __restrict__ lmean,
                                       const float* __restrict__ lstd, float mult, float* __restrict__ out, int C, int HW, int sample) {
  const int b = blockIdx.y, c = blockIdx.z;
  const float mu = lmean ? lmean[c] : 0.f;
  const size_t ob = ((size_t)b * C + c) * HW;
  const float* m = mom + ((size_t)b * (sample ? 2 * C : C) + c) * HW;
  const float* lv = m + (size_t)C * HW;
  for (int i = (blockIdx.x * blockDim.x + threadIdx.x) * 4; i < HW; i += gridDim.x * blockDim.x * 4) {
    float4 v = *(const float4*)(m + i);
    if (sample) {
      const float4 l = *(const float4*)(lv + i), e = *(const float4*)(noise + ob + i);
      v.x += expf(0.5f * fminf(fmaxf(l.x, -30.f), 20.f)) * e.x; v.y += expf(0.5f * fminf(fmaxf(l.y, -30.f), 20.f)) * e.y;
      v.z += expf(0.5f * fminf(fmaxf(l.z, -30.f), 20.f)) * e.z; v.w += expf(0.5f * fminf(fmaxf(l.w, -30.f), 20.f)) * e.w;
    }
    // (x - mean) / std, then * multiplier: the reference's two roundings (a true division, not a multiply by the reciprocal)
    if (lstd) { const float sd = lstd[c]; v.x = (v.x - mu) / sd; v.y = (v.y - mu) / sd; v.z = (v.z - mu) / sd; v.w = (v.w - mu) / sd; }
    else { v.x -= mu; v.y -= mu; v.z -= mu; v.w -= mu; }
    *(float4*)(out + ob + i) = make_float4(v.x * mult, v.y * mult, v.z * mult, v.w * mult);
  }
}
extern "C" int ldmae_latent_prologue(const float* moments, const float* noise, const float* lat_mean, const float* lat_std, float multiplier,
                                     float* out, int B, int C, int HW, int sample, void* stream) {
  LDMAE_REQUIRE(moments && out && B > 0 && C > 0 && HW > 0, "latent_prologue: null pointer or empty input");
  LDMAE_REQUIRE(HW % 4 == 0, "latent_prologue: H*W=%d must be a multiple of 4", HW);
  LDMAE_REQUIRE(!sample || noise, "latent_prologue: sample=1 needs the noise tensor");
  LDMAE_REQUIRE((lat_mean == nullptr) == (lat_std == nullptr), "latent_prologue: give both latent mean and std, or neither");
  LDMAE_REQUIRE(B <= 65535 && C <= 65535, "latent_prologue: B=%d C=%d exceed the grid limits", B, C);
  hipLaunchKernelGGL(latent_prologue_kernel, dim3(cdiv(HW, 1024), B, C), dim3(256), 0, as_stream(stream), moments, noise, lat_mean, lat_std, multiplier,
                     out, C, HW, sample);
  LDMAE_CHECK_LAUNCH("latent_prologue");
  return LDMAE_OK;
}

extern "C" int ldmae_gather_rows(const float* x, const long long* ids, float* out, int N, int L, int keep, int D, void* stream) {
  LDMAE_REQUIRE(x && ids && out && N > 0 && L > 0 && keep > 0 && D % 4 == 0, "gather_rows: bad arguments (D=%d multiple of 4)", D);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(keep, N), dim3(64), 0, as_stream(stream), x, ids, out, L, keep, D);
  LDMAE_CHECK_LAUNCH("gather_rows");
  return LDMAE_OK;
}
extern "C" int ldmae_scatter_rows(const float* dout, const long long* ids, float* dx, int N, int L, int keep, int D, void* stream) {
  LDMAE_REQUIRE(dout && ids && dx && N > 0 && L > 0 && keep > 0 && D % 4 == 0, "scatter_rows: bad arguments (D=%d multiple of 4)", D);
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(keep, N), dim3(64), 0, as_stream(stream), dout, ids, dx, L, keep, D);
  LDMAE_CHECK_LAUNCH("scatter_rows");
  return LDMAE_OK;
}

// ------------------------------------------------------------------ decoder input: kept tokens back in place + mask tokens + position embedding
// (models_mae.py:536-541: cat([x, mask_token.repeat]) -> gather(ids_restore) -> + decoder_pos_embed, as ONE pass: the reference's form moves the
// [B, L, D] tensor four times forward -- and an int64 index tensor twice its size -- and three more times backward.)
// out[b, l, :] = (ids[b, l] < keep ? x[b, ids[b, l], :] : mtok[:]) + pos[l, :].   16 lanes per row, 16 B per lane and access.
__global__ __launch_bounds__(256) void restore_tokens_kernel(const float* __restrict__ x, const float* __restrict__ mtok, const float* __restrict__ pos,
                                                             const long long* __restrict__ ids, float* __restrict__ out, long rows, int L, int keep, int D) {
  const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
  for (long r = (long)blockIdx.x * 16 + grp; r < rows; r += (long)gridDim.x * 16) {
    const long b = r / L;
    const int l = (int)(r - b * L);
    const long long j = ids[r];
    const float* src = j < keep ? x + ((size_t)b * keep + j) * D : mtok;
    for (int c = 4 * sub; c < D; c += 64) {
      const float4 v = *(const float4*)(src + c), p = *(const float4*)(pos + (size_t)l * D + c);
      *(float4*)(out + (size_t)r * D + c) = make_float4(v.x + p.x, v.y + p.y, v.z + p.z, v.w + p.w);
    }
  }
}
// backward: dx[b, ids[b, l], :] = dout[b, l, :] where ids[b, l] < keep (ids[b, :] is a permutation: every kept row is written exactly once);
// the other rows are the mask token's: P[workgroup][D] = their column sums (lane-owned columns, the 16 row groups added in order through LDS),
// summed over workgroups by ldmae_colsum.  NV = ceil(D / 64).
template <int NV>
__global__ __launch_bounds__(256) void restore_tokens_bwd_kernel(const float* __restrict__ dout, const long long* __restrict__ ids, float* __restrict__ dx,
                                                                 float* __restrict__ P, long rows, int L, int keep, int D) {
  __shared__ float red[16][NV * 64];
  const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
  float acc[NV][4];
#pragma unroll
  for (int i = 0; i < NV; ++i) { acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 0.f; }
  for (long r = (long)blockIdx.x * 16 + grp; r < rows; r += (long)gridDim.x * 16) {
    const long b = r / L;
    const long long j = ids[r];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = 4 * (sub + 16 * i);
      if (c >= D) continue;
      const float4 g = *(const float4*)(dout + (size_t)r * D + c);
      if (j < keep) *(float4*)(dx + ((size_t)b * keep + j) * D + c) = g;
      else { acc[i][0] += g.x; acc[i][1] += g.y; acc[i][2] += g.z; acc[i][3] += g.w; }
    }
  }
  if (!P) return;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) red[grp][4 * (sub + 16 * i) + k] = acc[i][k];
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    float s2 = 0.f;
    for (int g = 0; g < 16; ++g) s2 += red[g][c];
    P[(size_t)blockIdx.x * D + c] = s2;
  }
}
extern "C" long ldmae_colsum_workspace_bytes(int M, int N);
extern "C" int ldmae_colsum(int dtype, const void* X, int ldx, int M, int N, float* out, float beta, float* workspace, void* stream);
static unsigned restore_grid(long rows) { const long g = (rows + 15) / 16; return (unsigned)(g < 2048 ? (g > 0 ? g : 1) : 2048); }
extern "C" int ldmae_restore_tokens(const float* x, const float* mask_token, const float* pos, const long long* ids_restore, float* out, int B, int L,
                                    int keep, int D, void* stream) {
  LDMAE_REQUIRE(x && mask_token && pos && ids_restore && out && B > 0 && L > 0 && keep > 0 && keep <= L, "restore_tokens: null pointer or empty input");
  LDMAE_REQUIRE(D % 4 == 0, "restore_tokens: D=%d must be a multiple of 4", D);
  const long rows = (long)B * L;
  hipLaunchKernelGGL(restore_tokens_kernel, dim3(restore_grid(rows)), dim3(256), 0, as_stream(stream), x, mask_token, pos, ids_restore, out, rows, L, keep, D);
  LDMAE_CHECK_LAUNCH("restore_tokens");
  return LDMAE_OK;
}
extern "C" long ldmae_restore_tokens_bwd_workspace_bytes(int B, int L, int D) {
  const long g = restore_grid((long)B * L);
  return g * D * 4 + ldmae_colsum_workspace_bytes((int)g, D);
}
// dx [B, keep, D] (every row written), dmask_token [D] (NULL: not formed; workspace may then be NULL)
extern "C" int ldmae_restore_tokens_bwd(const float* dout, const long long* ids_restore, float* dx, float* dmask_token, int B, int L, int keep, int D,
                                        float* workspace, void* stream) {
  LDMAE_REQUIRE(dout && ids_restore && dx && B > 0 && L > 0 && keep > 0 && keep <= L, "restore_tokens_bwd: null pointer or empty input");
  LDMAE_REQUIRE(D % 4 == 0 && D <= 512, "restore_tokens_bwd: D=%d must be a multiple of 4, at most 512", D);
  LDMAE_REQUIRE(!dmask_token || workspace, "restore_tokens_bwd: the mask-token gradient needs the workspace");
  const long rows = (long)B * L;
  const unsigned grid = restore_grid(rows);
  float* P = dmask_token ? workspace : nullptr;
#define RT(NV) hipLaunchKernelGGL(restore_tokens_bwd_kernel<NV>, dim3(grid), dim3(256), 0, as_stream(stream), dout, ids_restore, dx, P, rows, L, keep, D)
  const int nv = (D + 63) / 64;
  if (nv <= 3) RT(3); else if (nv <= 6) RT(6); else RT(8);
#undef RT
  LDMAE_CHECK_LAUNCH("restore_tokens_bwd");
  if (dmask_token) return ldmae_colsum(LDMAE_F32, P, D, (int)grid, D, dmask_token, 0.f, P + (size_t)grid * D, stream);
  return LDMAE_OK;
}

// ------------------------------------------------------------------ LayerNorm (affine)
// 16 lanes per row, 16 B per lane and access (float4 / 4 x bf16): a 256-thread workgroup streams 16 rows at a time.  A lane owns
// the same columns {4 * (sub + 16 * i)} in every row it visits, so w / b live in registers and the dw / db partial sums of the
// backward are per-lane register accumulators (reduced across the 16 row groups through LDS once per workgroup, fixed order).
template <typename T> __device__ __forceinline__ void ld4(const T* p, float (&v)[4]);
template <> __device__ __forceinline__ void ld4<float>(const float* p, float (&v)[4]) { const float4 a = *(const float4*)p; v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; }
template <> __device__ __forceinline__ void ld4<bf16>(const bf16* p, float (&v)[4]) { const bf16x4 a = *(const bf16x4*)p; for (int j = 0; j < 4; ++j) v[j] = (float)a[j]; }
template <> __device__ __forceinline__ void ld4<f16>(const f16* p, float (&v)[4]) { const f16x4 a = *(const f16x4*)p; for (int j = 0; j < 4; ++j) v[j] = (float)a[j]; }
template <typename T> __device__ __forceinline__ void st4(T* p, const float (&v)[4]);
template <> __device__ __forceinline__ void st4<float>(float* p, const float (&v)[4]) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
template <> __device__ __forceinline__ void st4<bf16>(bf16* p, const float (&v)[4]) { bf16x4 a; for (int j = 0; j < 4; ++j) a[j] = (bf16)v[j]; *(bf16x4*)p = a; }
template <> __device__ __forceinline__ void st4<f16>(f16* p, const float (&v)[4]) { f16x4 a; for (int j = 0; j < 4; ++j) a[j] = from_f<f16>(v[j]); *(f16x4*)p = a; }

template <typename OutT, int NV>     // NV = ceil(D / 64): 16-B chunks per lane
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                            OutT* __restrict__ out, float* __restrict__ mean, float* __restrict__ rstd, int M, int D, float eps) {
  const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
  float wv[NV][4], bv[NV][4];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * (sub + 16 * i);
    if (c < D) { ld4<float>(w + c, wv[i]); ld4<float>(b + c, bv[i]); }
  }
  const float invD = 1.f / (float)D;
  for (int m = blockIdx.x * 16 + grp; m < M; m += gridDim.x * 16) {
    const float* xr = x + (size_t)m * D;
    float xv[NV][4], s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = 4 * (sub + 16 * i);
      if (c < D) { ld4<float>(xr + c, xv[i]); s += (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]); }
    }
    const float mu = group_sum<16>(s) * invD;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (4 * (sub + 16 * i) < D) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float cdev = xv[i][j] - mu; v += cdev * cdev; }
      }
    const float rs = rsqrtf(group_sum<16>(v) * invD + eps);
    if (sub == 0) { if (mean) mean[m] = mu; if (rstd) rstd[m] = rs; }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = 4 * (sub + 16 * i);
      if (c < D) {
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (xv[i][j] - mu) * rs * wv[i][j] + bv[i][j];
        st4<OutT>(out + (size_t)m * D + c, o);
      }
    }
  }
}

// dx_accum += dLN/dx ; partial dw/db per workgroup -> reduced in fixed order by ln_reduce_kernel
template <typename T, int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dout, const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dx,
                                                            T* __restrict__ dxc, float* __restrict__ P, int M, int D, int rows_per_wg) {
  extern __shared__ float red[];   // [16 row groups][2][D]
  const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
  float wv[NV][4], aw[NV][4], ab[NV][4];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * (sub + 16 * i);
    if (c < D) ld4<float>(w + c, wv[i]);
#pragma unroll
    for (int j = 0; j < 4; ++j) { aw[i][j] = 0.f; ab[i][j] = 0.f; }
  }
  const float invD = 1.f / (float)D;
  const int mend = min(M, (int)(blockIdx.x + 1) * rows_per_wg);
  for (int m = blockIdx.x * rows_per_wg + grp; m < mend; m += 16) {
    const float mu = mean[m], rs = rstd[m];
    const float* xr = x + (size_t)m * D;
    const T* gr = dout + (size_t)m * D;
    float* dxr = dx + (size_t)m * D;
    float g[NV][4], xh[NV][4], dxo[NV][4], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = 4 * (sub + 16 * i);
      if (c < D) {
        float xv[4];
        ld4<T>(gr + c, g[i]); ld4<float>(xr + c, xv); ld4<float>(dxr + c, dxo[i]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xh[i][j] = (xv[j] - mu) * rs;
          const float gy = g[i][j] * wv[i][j];
          s1 += gy; s2 += gy * xh[i][j];
          aw[i][j] += g[i][j] * xh[i][j]; ab[i][j] += g[i][j];
        }
      }
    }
    s1 = group_sum<16>(s1) * invD; s2 = group_sum<16>(s2) * invD;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = 4 * (sub + 16 * i);
      if (c < D) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dxo[i][j] += rs * (g[i][j] * wv[i][j] - s1 - xh[i][j] * s2);
        st4<float>(dxr + c, dxo[i]);
        if (dxc) st4<T>(dxc + (size_t)m * D + c, dxo[i]);      // the sum, rounded to the activation type: the operand of the next Linear's backward
      }
    }
  }
  float* rw = red + grp * 2 * D;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * (sub + 16 * i);
    if (c < D) { st4<float>(rw + c, aw[i]); st4<float>(rw + D + c, ab[i]); }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * D; i += 256) {
    float t = 0.f;
#pragma unroll
    for (int gq = 0; gq < 16; ++gq) t += red[gq * 2 * D + i];
    P[(size_t)blockIdx.x * 2 * D + i] = t;
  }
}

// column sums of the G partial rows [G][2D] in a fixed order: one WAVE per 4 adjacent columns, lane l sums rows l, l + 64, ... and a butterfly
// folds the 64 lane sums (384 columns x 2048 partial rows used to sit on 6 workgroups walking 512 rows each: 97 us, 50 times per VMAE step)
__global__ __launch_bounds__(256) void ln_reduce_kernel(const float* __restrict__ P, int G, int D, float* __restrict__ dw, float* __restrict__ db, float beta) {
  const int lane = threadIdx.x & 63, col = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
  if (col >= 2 * D) return;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int g = lane; g < G; g += 64) {
    const float4 t = *(const float4*)(P + (size_t)g * 2 * D + col);
    a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a.x += __shfl_xor(a.x, o, 64); a.y += __shfl_xor(a.y, o, 64); a.z += __shfl_xor(a.z, o, 64); a.w += __shfl_xor(a.w, o, 64);
  }
  if (lane == 0) {
    float* dst = col < D ? dw + col : db + (col - D);         // D % 4 == 0: a float4 never straddles the dw / db halves
    float4 o = a;
    if (beta != 0.f) { const float4 q = *(const float4*)dst; o.x += beta * q.x; o.y += beta * q.y; o.z += beta * q.z; o.w += beta * q.w; }
    *(float4*)dst = o;
  }
}

constexpr int LN_ROWS = 128;
#define LN_NV_DISPATCH(D, MACRO) \
  { const int nv_ = (D + 63) / 64; \
    if (nv_ <= 3) MACRO(3) else if (nv_ <= 6) MACRO(6) else if (nv_ <= 12) MACRO(12) else if (nv_ <= 16) MACRO(16) else MACRO(20) }      /* 20: D = 1280, mae_vit_huge_patch14 */
extern "C" int ldmae_layernorm_fwd(int out_dtype, const float* x, const float* w, const float* b, void* out, float* mean, float* rstd,
                                   int M, int D, float eps, void* stream) {
  LDMAE_REQUIRE(x && w && b && out && M > 0 && D > 0 && D % 4 == 0 && D <= 1280, "layernorm_fwd: bad arguments (D=%d: multiple of 4, <= 1280)", D);
  const unsigned grid = cdiv(M, 16) < 4096 ? cdiv(M, 16) : 4096;
#define LN_F(NV) { if (out_dtype == LDMAE_F16) hipLaunchKernelGGL((layernorm_fwd_kernel<f16, NV>), dim3(grid), dim3(256), 0, as_stream(stream), x, w, b, (f16*)out, mean, rstd, M, D, eps); \
                   else if (out_dtype == LDMAE_BF16) hipLaunchKernelGGL((layernorm_fwd_kernel<bf16, NV>), dim3(grid), dim3(256), 0, as_stream(stream), x, w, b, (bf16*)out, mean, rstd, M, D, eps); \
                   else hipLaunchKernelGGL((layernorm_fwd_kernel<float, NV>), dim3(grid), dim3(256), 0, as_stream(stream), x, w, b, (float*)out, mean, rstd, M, D, eps); }
  LN_NV_DISPATCH(D, LN_F);
#undef LN_F
  LDMAE_CHECK_LAUNCH("layernorm_fwd");
  return LDMAE_OK;
}
extern "C" long ldmae_layernorm_bwd_workspace_bytes(int M, int D) { return (long)cdiv(M, LN_ROWS) * 2 * D * 4; }
extern "C" int ldmae_layernorm_bwd_cast(int dtype, const void* dout, const float* x, const float* w, const float* mean, const float* rstd,
                                        float* dx_accum, void* dx_cast, float* dw, float* db, float beta_w, int M, int D, float* workspace, void* stream) {
  LDMAE_REQUIRE(dout && x && w && mean && rstd && dx_accum && dw && db && workspace && M > 0 && D > 0 && D % 4 == 0 && D <= 1280,
                "layernorm_bwd: bad arguments (D=%d: multiple of 4, <= 1280)", D);
  LDMAE_REQUIRE(!dx_cast || dtype != LDMAE_F32, "layernorm_bwd: dx_cast is a copy in the (16-bit) type of dout");
  hipStream_t st = as_stream(stream);
  const int G = cdiv(M, LN_ROWS);
  const size_t lds = (size_t)32 * D * 4;
#define LN_B(NV) { if (dtype == LDMAE_F16) { hipFuncSetAttribute((const void*)layernorm_bwd_kernel<f16, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                     hipLaunchKernelGGL((layernorm_bwd_kernel<f16, NV>), dim3(G), dim3(256), lds, st, (const f16*)dout, x, w, mean, rstd, dx_accum, (f16*)dx_cast, workspace, M, D, LN_ROWS); } \
                   else if (dtype == LDMAE_BF16) { hipFuncSetAttribute((const void*)layernorm_bwd_kernel<bf16, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                     hipLaunchKernelGGL((layernorm_bwd_kernel<bf16, NV>), dim3(G), dim3(256), lds, st, (const bf16*)dout, x, w, mean, rstd, dx_accum, (bf16*)dx_cast, workspace, M, D, LN_ROWS); } \
                   else { hipFuncSetAttribute((const void*)layernorm_bwd_kernel<float, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                     hipLaunchKernelGGL((layernorm_bwd_kernel<float, NV>), dim3(G), dim3(256), lds, st, (const float*)dout, x, w, mean, rstd, dx_accum, (float*)nullptr, workspace, M, D, LN_ROWS); } }
  LN_NV_DISPATCH(D, LN_B);
#undef LN_B
  hipLaunchKernelGGL(ln_reduce_kernel, dim3(cdiv(2 * D / 4, 4)), dim3(256), 0, st, workspace, G, D, dw, db, beta_w);
  LDMAE_CHECK_LAUNCH("layernorm_bwd");
  return LDMAE_OK;
}
extern "C" int ldmae_layernorm_bwd(int dtype, const void* dout, const float* x, const float* w, const float* mean, const float* rstd,
                                   float* dx_accum, float* dw, float* db, float beta_w, int M, int D, float* workspace, void* stream) {
  return ldmae_layernorm_bwd_cast(dtype, dout, x, w, mean, rstd, dx_accum, nullptr, dw, db, beta_w, M, D, workspace, stream);
}

// ------------------------------------------------------------------ exact GELU
template <typename T>
__global__ void gelu_fwd_kernel(const T* __restrict__ x, T* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = to_f<T>(x[i]);
    out[i] = from_f<T>(gelu_act<T>(v));
  }
}
template <typename T>
__global__ void gelu_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ x, T* __restrict__ dx, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = to_f<T>(x[i]);
    const float cdf = 0.5f * (1.f + erf_act<T>(v * 0.70710678118654752f)), pdf = 0.3989422804014327f * __expf(-0.5f * v * v);
    dx[i] = from_f<T>(to_f<T>(dout[i]) * (cdf + v * pdf));
  }
}
static unsigned gelu_grid(long n) { long g = (n + 255) / 256; return (unsigned)(g > 8192 ? 8192 : (g < 1 ? 1 : g)); }
extern "C" int ldmae_gelu_fwd(int dtype, const void* x, void* out, long n, void* stream) {
  LDMAE_REQUIRE(x && out && n > 0, "gelu_fwd: bad arguments");
  if (dtype == LDMAE_F16) hipLaunchKernelGGL(gelu_fwd_kernel<f16>, dim3(gelu_grid(n)), dim3(256), 0, as_stream(stream), (const f16*)x, (f16*)out, n);
  else if (dtype == LDMAE_BF16) hipLaunchKernelGGL(gelu_fwd_kernel<bf16>, dim3(gelu_grid(n)), dim3(256), 0, as_stream(stream), (const bf16*)x, (bf16*)out, n);
  else hipLaunchKernelGGL(gelu_fwd_kernel<float>, dim3(gelu_grid(n)), dim3(256), 0, as_stream(stream), (const float*)x, (float*)out, n);
  LDMAE_CHECK_LAUNCH("gelu_fwd");
  return LDMAE_OK;
}
extern "C" int ldmae_gelu_bwd(int dtype, const void* dout, const void* x, void* dx, long n, void* stream) {
  LDMAE_REQUIRE(dout && x && dx && n > 0, "gelu_bwd: bad arguments");
  if (dtype == LDMAE_F16) hipLaunchKernelGGL(gelu_bwd_kernel<f16>, dim3(gelu_grid(n)), dim3(256), 0, as_stream(stream), (const f16*)dout, (const f16*)x, (f16*)dx, n);
  else if (dtype == LDMAE_BF16) hipLaunchKernelGGL(gelu_bwd_kernel<bf16>, dim3(gelu_grid(n)), dim3(256), 0, as_stream(stream), (const bf16*)dout, (const bf16*)x, (bf16*)dx, n);
  else hipLaunchKernelGGL(gelu_bwd_kernel<float>, dim3(gelu_grid(n)), dim3(256), 0, as_stream(stream), (const float*)dout, (const float*)x, (float*)dx, n);
  LDMAE_CHECK_LAUNCH("gelu_bwd");
  return LDMAE_OK;
}

// ------------------------------------------------------------------ tanh-approximated GELU
// nn.GELU(approximate="tanh"): the activation of the timm Mlp a LightningDiT block gets with use_swiglu=False (lightningdit.py:208,219-224):
// y = 0.5 x (1 + tanh(k (x + 0.044715 x^3))), k = sqrt(2 / pi).  tanh(u) = 1 - 2 / (exp(2u) + 1) (exact at both infinities).
// Backward: dy/dx = 0.5 (1 + t) + 0.5 x (1 - t^2) k (1 + 3 * 0.044715 x^2).  No shipped configuration runs this block: plain elementwise passes.
__device__ __forceinline__ float tanh_exp(float u) { return 1.f - 2.f / (__expf(2.f * u) + 1.f); }
template <typename T>
__global__ void gelu_tanh_fwd_kernel(const T* __restrict__ x, T* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = to_f<T>(x[i]);
    out[i] = from_f<T>(0.5f * v * (1.f + tanh_exp(0.7978845608028654f * (v + 0.044715f * v * v * v))));
  }
}
template <typename T>
__global__ void gelu_tanh_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ x, T* __restrict__ dx, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = to_f<T>(x[i]), t = tanh_exp(0.7978845608028654f * (v + 0.044715f * v * v * v));
    const float d = 0.5f * (1.f + t) + 0.5f * v * (1.f - t * t) * 0.7978845608028654f * (1.f + 0.134145f * v * v);
    dx[i] = from_f<T>(to_f<T>(dout[i]) * d);
  }
}
extern "C" int ldmae_gelu_tanh_fwd(int dtype, const void* x, void* out, long n, void* stream) {
  LDMAE_REQUIRE(x && out && n > 0, "gelu_tanh_fwd: bad arguments");
  LDMAE_REQUIRE(dtype == LDMAE_F32 || dtype == LDMAE_BF16, "gelu_tanh_fwd: f32 or bf16");
  if (dtype == LDMAE_BF16) hipLaunchKernelGGL(gelu_tanh_fwd_kernel<bf16>, dim3(gelu_grid(n)), dim3(256), 0, as_stream(stream), (const bf16*)x, (bf16*)out, n);
  else hipLaunchKernelGGL(gelu_tanh_fwd_kernel<float>, dim3(gelu_grid(n)), dim3(256), 0, as_stream(stream), (const float*)x, (float*)out, n);
  LDMAE_CHECK_LAUNCH("gelu_tanh_fwd");
  return LDMAE_OK;
}
extern "C" int ldmae_gelu_tanh_bwd(int dtype, const void* dout, const void* x, void* dx, long n, void* stream) {
  LDMAE_REQUIRE(dout && x && dx && n > 0, "gelu_tanh_bwd: bad arguments");
  LDMAE_REQUIRE(dtype == LDMAE_F32 || dtype == LDMAE_BF16, "gelu_tanh_bwd: f32 or bf16");
  if (dtype == LDMAE_BF16) hipLaunchKernelGGL(gelu_tanh_bwd_kernel<bf16>, dim3(gelu_grid(n)), dim3(256), 0, as_stream(stream), (const bf16*)dout, (const bf16*)x, (bf16*)dx, n);
  else hipLaunchKernelGGL(gelu_tanh_bwd_kernel<float>, dim3(gelu_grid(n)), dim3(256), 0, as_stream(stream), (const float*)dout, (const float*)x, (float*)dx, n);
  LDMAE_CHECK_LAUNCH("gelu_tanh_bwd");
  return LDMAE_OK;
}

// ------------------------------------------------------------------ 3x3 conv on RGB (conv_decoder_pred.conv_smoother, models_mae.py:254,275)
// direct convolution, stride 1, zero padding 1, C channels in/out (C = 3); one thread per output pixel.
__global__ void conv3x3_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ out,
                               int B, int C, int Hh, int Ww) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * C * Hh * Ww) return;
  const int xx = i % Ww, yy = (i / Ww) % Hh, co = (i / ((long)Ww * Hh)) % C, n = i / ((long)Ww * Hh * C);
  float s = b ? b[co] : 0.f;
  for (int ci = 0; ci < C; ++ci)
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const int y2 = yy + dy, x2 = xx + dx;
        if (y2 >= 0 && y2 < Hh && x2 >= 0 && x2 < Ww)
          s += x[(((size_t)n * C + ci) * Hh + y2) * Ww + x2] * w[((co * C + ci) * 3 + dy + 1) * 3 + dx + 1];
      }
  out[i] = s;
}
// The shipped shape (3 channels, row length a multiple of 4): a thread owns four consecutive pixels of a row for ALL three output channels:
// per input plane and tap row one float4 + the two neighbours (the one-thread-per-output kernel above re-read every input 27 times with
// 64-bit index arithmetic per tap: 0.81 ms per 256 images of 256^2 against ~0.1 ms of HBM time).  The 81 taps are compile-time indices of
// a uniform pointer: scalar loads.  TR: the input-gradient form, dx[ci] = sum_co corr(dout[co], w[co][ci] flipped) -- the same loop with
// the tap index transposed and mirrored.
template <bool TR>
__global__ __launch_bounds__(256) void conv3x3_rgb_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                          float* __restrict__ out, int B, int Hh, int Ww) {
  const int W4 = Ww >> 2;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)B * Hh * W4) return;
  const int x0 = (int)(t % W4) * 4, yy = (int)((t / W4) % Hh);
  const long n = t / ((long)W4 * Hh);
  float acc[3][4];
#pragma unroll
  for (int co = 0; co < 3; ++co)
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[co][p] = (!TR && b) ? b[co] : 0.f;
#pragma unroll
  for (int ci = 0; ci < 3; ++ci) {
    const float* plane = x + ((size_t)n * 3 + ci) * Hh * Ww;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const int y2 = yy + dy;
      if (y2 < 0 || y2 >= Hh) continue;
      const float* row = plane + (size_t)y2 * Ww + x0;
      const float4 m = *(const float4*)row;
      const float v[6] = {x0 > 0 ? row[-1] : 0.f, m.x, m.y, m.z, m.w, x0 + 4 < Ww ? row[4] : 0.f};
#pragma unroll
      for (int co = 0; co < 3; ++co)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const float wt = TR ? w[((ci * 3 + co) * 3 + (1 - dy)) * 3 + (1 - dx)] : w[((co * 3 + ci) * 3 + (dy + 1)) * 3 + (dx + 1)];
#pragma unroll
          for (int p = 0; p < 4; ++p) acc[co][p] += v[p + 1 + dx] * wt;
        }
    }
  }
#pragma unroll
  for (int co = 0; co < 3; ++co)
    *(float4*)(out + (((size_t)n * 3 + co) * Hh + yy) * Ww + x0) = make_float4(acc[co][0], acc[co][1], acc[co][2], acc[co][3]);
}
static bool conv_rgb_shape(const float* a, const float* b_, int C, int W) { return C == 3 && W % 4 == 0 && (((uintptr_t)a | (uintptr_t)b_) & 15) == 0; }

extern "C" int ldmae_conv3x3(const float* x, const float* w, const float* b, float* out, int B, int C, int H, int W, void* stream) {
  LDMAE_REQUIRE(x && w && out && B > 0 && C > 0 && H > 0 && W > 0, "conv3x3: bad arguments");
  const long n = (long)B * C * H * W;
  if (conv_rgb_shape(x, out, C, W)) {
    const long nt = (long)B * H * (W / 4);
    hipLaunchKernelGGL(conv3x3_rgb_kernel<false>, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, as_stream(stream), x, w, b, out, B, H, W);
  } else
  hipLaunchKernelGGL(conv3x3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), x, w, b, out, B, C, H, W);
  LDMAE_CHECK_LAUNCH("conv3x3");
  return LDMAE_OK;
}

// ------------------------------------------------------------------ masked / visible reconstruction loss in IMAGE space (models_mae.py:733-754)
// forward_loss patchifies the target, takes the per-patch mean of (pred - target)^2 and averages it over the masked and over the visible
// patches.  Every patch has the same number of elements, so both numbers are weighted sums over PIXELS: sum_m = sum mask[patch(pixel)] d^2,
// sum_v = sum (1 - mask) d^2, with d taken between the smoothing conv's output image and the input image -- no patchify of either, no
// [B, L, p*p*3] temporaries (the torch formulation was ~20 elementwise / reduce / permute launches per step with its backward).
// fwd: per-workgroup partial sums [G][2] (fixed order; the caller adds the G rows); bwd: dpred_img = 2 d (cm mask + cv (1 - mask)) with the two
// coefficients read from the device (upstream gradient / (elements per patch * patch count): no host round trip).
__global__ __launch_bounds__(256) void mae_loss_fwd_kernel(const float* __restrict__ x, const float* __restrict__ t, const float* __restrict__ mask,
                                                           float* __restrict__ P, long n4, int C, int Hh, int Ww, int p) {
  __shared__ float red[4][2];
  const int W4 = Ww >> 2, gw = Ww / p;
  float am = 0.f, av = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const int x0 = (int)(i % W4) * 4, yy = (int)((i / W4) % Hh);
    const long b = i / ((long)W4 * Hh * C);
    const float m = mask[b * (long)(Hh / p) * gw + (yy / p) * gw + x0 / p];           // (p % 4 == 0: the four pixels share a patch)
    const float4 a = *(const float4*)(x + i * 4), c = *(const float4*)(t + i * 4);
    const float d0 = a.x - c.x, d1 = a.y - c.y, d2 = a.z - c.z, d3 = a.w - c.w;
    const float ss = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    am += m * ss; av += (1.f - m) * ss;
  }
  am = wave_sum(am); av = wave_sum(av);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = am; red[threadIdx.x >> 6][1] = av; }
  __syncthreads();
  if (threadIdx.x < 2) P[(size_t)blockIdx.x * 2 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void mae_loss_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t, const float* __restrict__ mask,
                                                           const float* __restrict__ coef, float* __restrict__ dx, long n4, int C, int Hh, int Ww, int p) {
  const int W4 = Ww >> 2, gw = Ww / p;
  const float cm = coef[0], cv = coef[1];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const int x0 = (int)(i % W4) * 4, yy = (int)((i / W4) % Hh);
    const long b = i / ((long)W4 * Hh * C);
    const float m = mask[b * (long)(Hh / p) * gw + (yy / p) * gw + x0 / p];
    const float k = 2.f * (cm * m + cv * (1.f - m));
    const float4 a = *(const float4*)(x + i * 4), c = *(const float4*)(t + i * 4);
    *(float4*)(dx + i * 4) = make_float4(k * (a.x - c.x), k * (a.y - c.y), k * (a.z - c.z), k * (a.w - c.w));
  }
}
static unsigned mae_loss_grid(long n4) { const long g = (n4 + 255) / 256; return (unsigned)(g < 2048 ? (g > 0 ? g : 1) : 2048); }
extern "C" long ldmae_mae_loss_groups(long elements) { return (long)mae_loss_grid(elements / 4); }
extern "C" int ldmae_mae_loss_fwd(const float* pred_img, const float* imgs, const float* mask, float* partials, int B, int C, int H, int W, int p,
                                  void* stream) {
  LDMAE_REQUIRE(pred_img && imgs && mask && partials && B > 0 && C > 0 && H > 0 && W > 0, "mae_loss_fwd: bad arguments");
  LDMAE_REQUIRE(p > 0 && p % 4 == 0 && H % p == 0 && W % p == 0 && (((uintptr_t)pred_img | (uintptr_t)imgs) & 15) == 0,
                "mae_loss_fwd: patch size %d must be a multiple of 4 that divides the image (%d x %d), images 16-B aligned", p, H, W);
  const long n4 = (long)B * C * H * W / 4;
  hipLaunchKernelGGL(mae_loss_fwd_kernel, dim3(mae_loss_grid(n4)), dim3(256), 0, as_stream(stream), pred_img, imgs, mask, partials, n4, C, H, W, p);
  LDMAE_CHECK_LAUNCH("mae_loss_fwd");
  return LDMAE_OK;
}
extern "C" int ldmae_mae_loss_bwd(const float* pred_img, const float* imgs, const float* mask, const float* coef, float* dpred_img, int B, int C, int H,
                                  int W, int p, void* stream) {
  LDMAE_REQUIRE(pred_img && imgs && mask && coef && dpred_img && B > 0 && C > 0 && H > 0 && W > 0, "mae_loss_bwd: bad arguments");
  LDMAE_REQUIRE(p > 0 && p % 4 == 0 && H % p == 0 && W % p == 0 && (((uintptr_t)pred_img | (uintptr_t)imgs | (uintptr_t)dpred_img) & 15) == 0,
                "mae_loss_bwd: patch size %d must be a multiple of 4 that divides the image (%d x %d), images 16-B aligned", p, H, W);
  const long n4 = (long)B * C * H * W / 4;
  hipLaunchKernelGGL(mae_loss_bwd_kernel, dim3(mae_loss_grid(n4)), dim3(256), 0, as_stream(stream), pred_img, imgs, mask, coef, dpred_img, n4, C, H, W, p);
  LDMAE_CHECK_LAUNCH("mae_loss_bwd");
  return LDMAE_OK;
}

// ------------------------------------------------------------------ backward of the 3x3 RGB smoothing conv (VMAE pre-training, engine_pretrain.py:51-76:
// the decoder's conv_smoother is trained with everything else).  C = 3: dx is the correlation of dout with the transposed taps,
// dw / db are 84 whole-tensor sums -> per-workgroup partials [G][C*C*9 + C] (registers -> wave shuffles -> LDS), summed in fixed order.
__global__ void conv3x3_bwd_dx_kernel(const float* __restrict__ dout, const float* __restrict__ w, float* __restrict__ dx, int B, int C, int Hh, int Ww) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * C * Hh * Ww) return;
  const int xx = i % Ww, yy = (i / Ww) % Hh, ci = (i / ((long)Ww * Hh)) % C, n = i / ((long)Ww * Hh * C);
  float s = 0.f;
  for (int co = 0; co < C; ++co)
    for (int ky = 0; ky < 3; ++ky)
      for (int kx = 0; kx < 3; ++kx) {
        const int y2 = yy - ky + 1, x2 = xx - kx + 1;
        if (y2 >= 0 && y2 < Hh && x2 >= 0 && x2 < Ww)
          s += dout[(((size_t)n * C + co) * Hh + y2) * Ww + x2] * w[((co * C + ci) * 3 + ky) * 3 + kx];
      }
  dx[i] = s;
}
template <int C>
__global__ __launch_bounds__(256) void conv3x3_bwd_dw_kernel(const float* __restrict__ dout, const float* __restrict__ x, float* __restrict__ P,
                                                             int B, int Hh, int Ww) {
  constexpr int NA = C * C * 9 + C;
  __shared__ float red[4][NA];
  float acc[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) acc[a] = 0.f;
  const long npix = (long)B * Hh * Ww;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long)gridDim.x * 256) {
    const int xx = i % Ww, yy = (i / Ww) % Hh, n = i / ((long)Ww * Hh);
    float g[C], xv[C][9];
#pragma unroll
    for (int co = 0; co < C; ++co) g[co] = dout[(((size_t)n * C + co) * Hh + yy) * Ww + xx];
#pragma unroll
    for (int ci = 0; ci < C; ++ci)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int y2 = yy + ky - 1, x2 = xx + kx - 1;
          xv[ci][ky * 3 + kx] = (y2 >= 0 && y2 < Hh && x2 >= 0 && x2 < Ww) ? x[(((size_t)n * C + ci) * Hh + y2) * Ww + x2] : 0.f;
        }
#pragma unroll
    for (int co = 0; co < C; ++co) {
#pragma unroll
      for (int ci = 0; ci < C; ++ci)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[(co * C + ci) * 9 + t] += g[co] * xv[ci][t];
      acc[C * C * 9 + co] += g[co];
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    const float v = wave_sum(acc[a]);
    if (lane == 0) red[wave][a] = v;
  }
  __syncthreads();
  if (threadIdx.x < NA) P[(size_t)blockIdx.x * NA + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// one workgroup per tap / bias gradient: 256 threads stride over the G partials, fixed-order tree (a single 128-thread block walking all
// 1024 partials serially took as long as the pass that produced them)
__global__ __launch_bounds__(256) void conv3x3_bwd_reduce_kernel(const float* __restrict__ P, int G, int NA, int NW, float* __restrict__ dw, float* __restrict__ db) {
  __shared__ float red[4];
  const int a = blockIdx.x;
  float s = 0.f;
  for (int g = threadIdx.x; g < G; g += 256) s += P[(size_t)g * NA + a];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) { const float v = (red[0] + red[1]) + (red[2] + red[3]); if (a < NW) dw[a] = v; else db[a - NW] = v; }
}
constexpr int CONV_BWD_G = 1024;
extern "C" long ldmae_conv3x3_bwd_workspace_bytes(int C) { return (long)CONV_BWD_G * (C * C * 9 + C) * 4; }
extern "C" int ldmae_conv3x3_bwd(const float* dout, const float* x, const float* w, float* dx, float* dw, float* db, int B, int C, int H, int W,
                                 float* workspace, void* stream) {
  LDMAE_REQUIRE(dout && x && w && dw && db && workspace && B > 0 && H > 0 && W > 0, "conv3x3_bwd: bad arguments");
  LDMAE_REQUIRE(C == 3, "conv3x3_bwd: C=%d unsupported (the RGB smoothing conv has 3 channels)", C);
  hipStream_t st = as_stream(stream);
  const long n = (long)B * C * H * W, npix = (long)B * H * W;
  if (dx && conv_rgb_shape(dout, dx, C, W)) {
    const long nt = (long)B * H * (W / 4);
    hipLaunchKernelGGL(conv3x3_rgb_kernel<true>, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, dout, w, (const float*)nullptr, dx, B, H, W);
  } else if (dx) hipLaunchKernelGGL(conv3x3_bwd_dx_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dout, w, dx, B, C, H, W);
  const int G = (int)((npix + 255) / 256 < CONV_BWD_G ? (npix + 255) / 256 : CONV_BWD_G);
  hipLaunchKernelGGL(conv3x3_bwd_dw_kernel<3>, dim3(G), dim3(256), 0, st, dout, x, workspace, B, H, W);
  hipLaunchKernelGGL(conv3x3_bwd_reduce_kernel, dim3(C * C * 9 + C), dim3(256), 0, st, workspace, G, C * C * 9 + C, C * C * 9, dw, db);
  LDMAE_CHECK_LAUNCH("conv3x3_bwd");
  return LDMAE_OK;
}
