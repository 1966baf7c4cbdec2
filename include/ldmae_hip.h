/* libldmae_hip -- C ABI of the MI355X (gfx950) kernels behind the LDMAE hot path.
 *
 * The reference (isno0907/ldmae) has NO native code and NO FFI: its seam for this path is the
 * Python module API (models/lightningdit.py, tokenizer/models_mae.py).  This header
 * is what a binding for that seam calls: one entry per fused region of the LightningDiT block /
 * VMAE encoder, each citing the reference lines whose arithmetic it replaces (paths relative to
 * /root/reference/LDMAE).  The ctypes binding lives in ldmae_amd/_lib.py; INTEGRATION.md shows the
 * stub a maintainer adds on the reference side.
 *
 * Conventions
 *  - plain device pointers + explicit sizes, no torch types; `stream` is a hipStream_t (NULL = default).
 *  - every call only ENQUEUES on `stream`: no allocation, no synchronisation; compute entry points keep no state
 *    between calls and are callable from the Python main thread and the autograd thread concurrently.
 *    Workspaces are caller-owned and used only inside the launches of the call that received them: two calls may share one
 *    workspace exactly when they are ordered on one stream (ldmae_amd/ops.py keys its scratch buffers by device, purpose AND stream).
 *    The only process-wide mutable state is the event list of the opt-in ldmae_prof_*
 *    timing hook (mutex-protected, off by default); per-device launch attributes (CU count, dynamic-LDS opt-in) are looked
 *    up per call.  A/B knobs and timeline stamps live in the separate diagnostic build only (csrc/probe/ldmae_diag.h).
 *  - return 0 on success, negative on error; ldmae_last_error() gives the thread-local message.
 *  - dtype codes select the activation type: LDMAE_F32 (parity path, exact-f32 MFMA) or LDMAE_BF16
 *    (throughput path, bf16 MFMA with f32 accumulation).  Residual stream, norms' statistics,
 *    modulation vectors, gradients of parameters and optimizer state are always f32.
 *  - "rows" M = batch * tokens, row-major, token-major activations [M, D].
 */
#ifndef LDMAE_HIP_H
#define LDMAE_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define LDMAE_OK 0
#define LDMAE_ERR_INVALID (-1)   /* bad argument / unsupported shape */
#define LDMAE_ERR_HIP (-2)       /* HIP runtime error at launch */

#define LDMAE_F32 0
#define LDMAE_BF16 1
/* fp16 activations: the TF32-CLASS forward path (round 5).  What the reference's drivers ask for with torch.backends.cuda.matmul.allow_tf32 = True
 * (inference.py:79, extract_features.py:2-3) is 10-bit-mantissa products with f32 accumulation; gfx950 has no TF32 MFMA, but fp16 has exactly
 * that mantissa at the bf16 rate, and the operands it is used for are O(1) (conversion saturates at +-65504).  Forward-only entry points:
 * ldmae_cast / ldmae_cast_weight (dst), ldmae_layernorm_fwd (out), ldmae_gemm_nt (dtype; epilogues BIAS / GATE_RES / BIAS_POS / BIAS_GELU; out
 * fp16 or f32; shapes of the whole-line kernel: K % 64 == 0, rows on 128-B lines), ldmae_attention_fwd_qkv (head_dim 16).  For VMAE pre-training
 * under fp16 autocast (engine_pretrain.py:51-57) also the backward: ldmae_gemm_nt(EPI_BIAS / EPI_GELU_BWD | LDMAE_EPI_F16_INF), ldmae_gemm_tn,
 * ldmae_attention_bwd_qkv (head_dim 16), ldmae_layernorm_bwd, ldmae_colsum, ldmae_gelu_fwd/bwd. */
#define LDMAE_F16 2

/* GEMM epilogues */
#define LDMAE_EPI_BIAS 0       /* C = acc + bias (+ beta*C)                                   */
#define LDMAE_EPI_GATE_RES 1   /* y = acc + bias ; xout = xin + gate[b]*y  (gate NULL -> 1)    */
#define LDMAE_EPI_BIAS_POS 2   /* C = acc + bias + pos[m % rows_per_batch]  (patch-embed)      */
#define LDMAE_EPI_BIAS_GELU 3  /* C = gelu_erf(acc + bias)  (VMAE Mlp fc1), pre-activation to C2 */
#define LDMAE_EPI_SWIGLU 4     /* bf16 only: B = w12 [2Hs,K]; C = h12 [M,2Hs] = acc+bias (NULL: not stored -- forward-only), xout(as bf16*) = hid [M,Hs] = silu(x1)*x2 */
#define LDMAE_EPI_SWIGLU_BWD 5 /* bf16 only: acc = dhid [M,Hs]; xin(as bf16*) = h12 [M,2Hs]; C = dh12 [M,2Hs]; xout (optional) =
                                  [ceil(M/128)][2Hs] f32 partial column sums of dh12 as stored (bias gradient; caller sums the rows) */
#define LDMAE_EPI_GELU_BWD 6   /* C = dpre [M,N] = g * gelu_erf'(pre), g = acc (+ bias) rounded to out_dtype first; xin (read as out_dtype*) = pre [M, ldc]:
                                  the input gradient of fc2 with the GELU backward of the VMAE Mlp fused (models_mae.py:172: timm Mlp) */
#define LDMAE_EPI_QKV_ROPE 7   /* bf16 only, through ldmae_gemm_nt_qkv_rope: the qkv Linear with the QK-RMSNorm + RoPE front end of the attention fused
                                  (a wave's 64 output columns are one head of q, k or v) */
/* launch mode, or'ed into `epi` per call (bf16 GEMMs): one 256x256 tile per workgroup instead of one persistent workgroup per CU.  A
   data-parallel caller sets it while RCCL's collective kernels share the chip with backward (ldmae_amd/distributed.py); results are
   bitwise equal to the persistent launch */
#define LDMAE_EPI_TILE_LAUNCH 0x100
/* kernel form, or'ed into `epi` per call (bf16 GEMMs): keep the HALF-LINE kernel (64-B LDS rows, 16 x 64-B ring pieces: gemm_nt_persist_kernel)
   for a shape that the whole-line kernel (128-B LDS rows, gemm_nt_lines.hip -- the default wherever operand rows start on 128-B lines and
   K % 64 == 0) would take.  Same products in the same order: bitwise-equal results; tests and tools/bench_nt.py use it for A/B runs */
#define LDMAE_EPI_HALF_LINES 0x200
/* fp16 outputs (dtype LDMAE_F16), or'ed into `epi` per call: values beyond +-65504 become INFINITIES, as under torch's fp16 autocast -- what a
   gradient GEMM under a loss scaler wants (the scaler sees the non-finite gradient and skips the step).  Without the flag fp16 outputs SATURATE
   at +-65504 (the forward / TF32-class calls) */
#define LDMAE_EPI_F16_INF 0x400

const char* ldmae_last_error(void);
const char* ldmae_version(void);
const char* ldmae_arch(void);     /* "gfx950" */

/* ---- Linear layers -------------------------------------------------------------------------- */
/* C[M,N] = A[M,K] . B[N,K]^T with fused epilogue.  Replaces nn.Linear forward (lightningdit.py:59,
 * 64,68,88; swiglu_ffn.py:33,36; models_mae.py:125,127,133,143) and, with the [in,out] weight copy
 * as B, the input-gradient GEMM of the same layers.
 *   EPI_GATE_RES fuses `x = x + gate.unsqueeze(1) * branch(...)` (lightningdit.py:248-249):
 *   C (optional, dtype out_dtype) receives y for the backward pass, xin/xout are the f32 residual
 *   stream (may alias), gate is a [batch, gate_ld] f32 view, rows_per_batch = tokens per sample. */
int ldmae_gemm_nt(int dtype, int out_dtype, int epi, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                  int M, int N, int K, const float* bias, float beta, const float* xin, float* xout,
                  const float* gate, int gate_ld, int rows_per_batch, void* stream);
/* The qkv Linear of the DiT block (lightningdit.py:68) with q_norm / k_norm / RoPE (:70-74) applied in the GEMM's epilogue, head dim 64, bf16:
 *   qkv [B*N, 3*H*64] = A [B*N, K] . W [3*H*64, K]^T + bias as ldmae_gemm_nt(EPI_BIAS) writes it (the backward pass reads the pre-norm q / k rows; with
 *   store_raw_qk = 0 -- forward-only calls -- only the v third is written), and q2, k2 [B, H, N, 64] = what ldmae_qknorm_rope_fwd makes of the stored
 *   (bf16-rounded) q and k: bitwise the same values, without the pass that re-reads them.  wq = wk = NULL: RoPE only (use_qknorm = False).
 *   Covers B*N % 256 == 0, N % 128 == 0, K % 64 == 0, 128-B aligned operands (ldmae_gemm_nt_qkv_rope_ok says whether a shape is covered; the
 *   caller otherwise runs ldmae_gemm_nt + ldmae_qknorm_rope_fwd).  tile_launch: as LDMAE_EPI_TILE_LAUNCH. */
int ldmae_gemm_nt_qkv_rope_ok(int B, int N, int H, int hd, int K, int lda, int ldb);
int ldmae_gemm_nt_qkv_rope(const void* A, int lda, const void* W, int ldb, const float* bias, void* qkv, void* q2, void* k2, const float* wq,
                           const float* wk, const float* cos, const float* sin, int B, int N, int H, int hd, int K, float eps, int store_raw_qk,
                           int tile_launch, void* stream);
/* C[N,K] (f32) = beta*C + A[M,N]^T . B[M,K]: weight gradient of the same layers (contraction over
 * token rows, split over workgroups; partial slabs are summed in fixed order -> deterministic). */
int ldmae_gemm_tn_splits(int dtype, int M, int N, int K);
long ldmae_gemm_tn_workspace_bytes(int dtype, int M, int N, int K);
/* dbias (optional, [N] f32) = beta*dbias + column sums of A: the bias gradient of the same layer, fused. */
int ldmae_gemm_tn(int dtype, const void* A, int lda, const void* B, int ldb, float* C, float* dbias, int M, int N, int K, float beta,
                  float* workspace, long workspace_bytes, void* stream);
/* out[N] (f32) = beta*out + column sums of X[M,N]: bias gradients. workspace >= ldmae_colsum_workspace_bytes */
long ldmae_colsum_workspace_bytes(int M, int N);
int ldmae_colsum(int dtype, const void* X, int ldx, int M, int N, float* out, float beta, float* workspace, void* stream);

/* f32 master weight [R,C] -> dst[R,C] (dtype) and, if dstT != NULL, dstT[C,R] (the [in,out] copy) */
int ldmae_cast_weight(int dst_dtype, const float* src, void* dst, void* dstT, int R, int C, void* stream);
int ldmae_cast(int src_dtype, int dst_dtype, const void* src, void* dst, long n, void* stream);
/* Thin f32 products with a contraction of K = 16 or 32 over M = batch * tokens rows -- the DiT's PatchEmbed at patch size 1
 * (lightningdit.py:309, 402: K = C * p * p) and dX of the FinalLayer's Linear (:270): HBM streaming, one pass over the big tensor.
 *   ldmae_thin_nt: out[M,N] (out_dtype) = T[M,K] . W[N,K]^T + bias[N] (+ pos[m % rows_per_batch, :] when pos != NULL); N % 4 == 0.
 *   ldmae_thin_tn: dW[N,K] = beta * dW + G[M,N]^T . T[M,K]; dbias[N] (may be NULL) = beta * dbias + column sums of G; partial sums per
 *                  512 rows, reduced in fixed order (deterministic); workspace >= ldmae_thin_tn_workspace_bytes(M, N, K). */
int ldmae_thin_nt(int out_dtype, const float* T, const float* W, const float* bias, const float* pos, void* out, int M, int N, int K,
                  int rows_per_batch, void* stream);
long ldmae_thin_tn_workspace_bytes(int M, int N, int K);
int ldmae_thin_tn(const float* G, const float* T, float* dW, float* dbias, int M, int N, int K, float beta, float* workspace,
                  long workspace_bytes, void* stream);
/* dst[i][0..n[i]) += src[i][0..n[i]) (f32) for `count` (<= 32) triples in one launch; dst / src / n are HOST arrays.  The host side uses it
 * to add a block's small parameter gradients into their .grad views in place of one AccumulateGrad add each (train_accum.py:236 backward). */
int ldmae_multi_add(int count, void* const* dst, const void* const* src, const long* n, void* stream);
/* `count` (<= 64) f32 device tensors of n_each elements each -> dst[count * n_each] in dst_dtype, one launch; srcs is a HOST array of device
 * pointers (16-B aligned; n_each % 8 == 0).  Used to stack the adaLN_modulation weights of all blocks (lightningdit.py:233-236) into the
 * [depth * 6D, D] operand of one GEMM. */
int ldmae_cast_stack(int dst_dtype, const void* const* srcs, int count, long n_each, void* dst, void* stream);

/* ---- adaLN-modulated RMSNorm (rmsnorm.py:51-77 + lightningdit.py:26-30) ----------------------- */
/* out = rmsnorm(x; w, eps) * (1 + scale[b]) + shift[b];  rstd[M] saved for backward.
 * shift/scale are [batch, mod_ld] f32 views (column slices of the adaLN output). */
int ldmae_rmsnorm_modulate_fwd(int out_dtype, const float* x, const float* w, const float* shift, const float* scale,
                               int mod_ld, void* out, float* rstd, int M, int D, int rows_per_batch, float eps, void* stream);
/* dx_accum = beta_x * dx_accum + d(x) (beta_x 0 or 1; 0 writes dx without reading it); dshift/dscale [batch, dmod_ld] = per-sample sums; dw[D] += sum (beta_w).  workspace from
 * ldmae_rmsnorm_modulate_bwd_workspace_bytes. */
long ldmae_rmsnorm_modulate_bwd_workspace_bytes(int M, int D, int rows_per_batch);
int ldmae_rmsnorm_modulate_bwd(int dtype, const void* dout, const float* x, const float* w, const float* scale, int mod_ld,
                               const float* rstd, float* dx_accum, float beta_x, float* dshift, float* dscale, int dmod_ld, float* dw,
                               float beta_w, int M, int D, int rows_per_batch, float* workspace, void* stream);
/* The same followed by the gated-residual backward (ldmae_gate_bwd) of the updated dx_accum, in one pass over the rows (the block's
 * norm2 backward feeds the attention branch's gate: lightningdit.py:247-248 read backwards): dy = dx_accum * gate[b] (in `dtype`),
 * dgate [B, dgate_ld] = sum_n dx_accum * y, dbias [D] = column sums of dy. */
long ldmae_rmsnorm_modulate_bwd_gate_workspace_bytes(int M, int D, int rows_per_batch);
int ldmae_rmsnorm_modulate_bwd_gate(int dtype, const void* dout, const float* x, const float* w, const float* scale, int mod_ld,
                                    const float* rstd, float* dx_accum, float beta_x, float* dshift, float* dscale, int dmod_ld, float* dw,
                                    float beta_w, const void* y, const float* gate, int gate_ld, void* dy, float* dgate, int dgate_ld,
                                    float* dbias, int M, int D, int rows_per_batch, float* workspace, void* stream);
/* The three entry points above for blocks built with use_rmsnorm=False: nn.LayerNorm(hidden, elementwise_affine=False, eps=1e-6) + modulate
 * (lightningdit.py:200-201,257; modulate :26-30).  y = (x - mean) * rstd * (1 + scale[b]) + shift[b]; rstd [M] = rsqrt(var + eps) is saved, the
 * backward recomputes the row mean from x.  No weight, no weight gradient; workspaces: ldmae_rmsnorm_modulate_bwd(_gate)_workspace_bytes. */
int ldmae_layernorm_modulate_fwd(int out_dtype, const float* x, const float* shift, const float* scale, int mod_ld, void* out, float* rstd,
                                 int M, int D, int rows_per_batch, float eps, void* stream);
int ldmae_layernorm_modulate_bwd(int dtype, const void* dout, const float* x, const float* scale, int mod_ld, const float* rstd, float* dx_accum,
                                 float beta_x, float* dshift, float* dscale, int dmod_ld, int M, int D, int rows_per_batch, float* workspace,
                                 void* stream);
int ldmae_layernorm_modulate_bwd_gate(int dtype, const void* dout, const float* x, const float* scale, int mod_ld, const float* rstd,
                                      float* dx_accum, float beta_x, float* dshift, float* dscale, int dmod_ld, const void* y, const float* gate,
                                      int gate_ld, void* dy, float* dgate, int dgate_ld, float* dbias, int M, int D, int rows_per_batch,
                                      float* workspace, void* stream);

/* ---- attention front end (lightningdit.py:68-74; rmsnorm.py on head_dim; pos_embed.py:38-42,135) */
/* qkv [B,N,3,H,hd] -> q,k = rope(rmsnorm(.)*w) and v, each [B,H,N,hd]. cos/sin [N,hd] f32.
 * wq = wk = cos = sin = NULL: plain head-major relayout (VMAE attention has no QK-norm / RoPE, models_mae.py:133-134).
 * wq = wk = NULL with cos / sin given: RoPE only -- the block built with use_qknorm=False, q_norm = k_norm = nn.Identity
 * (lightningdit.py:60-61,69; the reference's configs/celeba_hq/lightningdit_b_vmae_f8d16_cfg.yaml:30); backward: the rotation's adjoint, qkv /
 * dwq / dwk may be NULL.
 * v = NULL (fwd) / dv = NULL (bwd), forms with RoPE only: v stays in the packed buffer (see ldmae_attention_fwd_pv / _bwd_pv);
 * the backward then reads dv from the v slot of dqkv for the bias-gradient sums and leaves it in place. */
int ldmae_qknorm_rope_fwd(int dtype, const void* qkv, const float* wq, const float* wk, const float* cos, const float* sin,
                          void* q, void* k, void* v, int B, int N, int H, int hd, float eps, void* stream);
long ldmae_qknorm_rope_bwd_workspace_bytes(int B, int N, int H, int hd);
/* dbias_hqd (optional, [H][3][hd] f32): column sums of dqkv as stored = bias gradient of the qkv Linear, in (head, q|k|v, d) order */
int ldmae_qknorm_rope_bwd(int dtype, const void* dq, const void* dk, const void* dv, const void* qkv, const float* wq,
                          const float* wk, const float* cos, const float* sin, void* dqkv, float* dwq, float* dwk, float beta_w,
                          float* dbias_hqd, int B, int N, int H, int hd, float eps, float* workspace, void* stream);

/* Standalone 2-D RoPE, the callable form of VisionRotaryEmbeddingFast.forward (pos_embed.py:135; rotate_half :38-42):
 * out[r,:] = t[r,:]*cos[r % N,:] + rotate_half(t[r,:])*sin[r % N,:] on [rows, hd] (rows = anything x N).  transposed = 1: the adjoint
 * (backward).  The LightningDiT block does not use it: there the rotation is fused into ldmae_qknorm_rope_fwd / _bwd. */
int ldmae_rope(int dtype, const void* t, const float* cos, const float* sin, void* out, long rows, int N, int hd, int transposed,
               void* stream);

/* ---- attention core (F.scaled_dot_product_attention, lightningdit.py:76-80; manual softmax attention
 *      models_mae.py:135-141).  q,k,v [B,H,N,hd]; o [B,N,H*hd]; lse [B,H,N] f32 (natural log). */
int ldmae_attention_fwd(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int N, int hd,
                        float scale, void* stream);
/* N is arbitrary (the last 64-row tile of a sweep may be ragged: its missing rows are masked to -inf before the exponential).
 * delta: [2][B,H,NP] f32 workspace, NP = N rounded up to a multiple of 64 (bf16: -rowsum(dO*O) | -lse*log2(e), the initial accumulators
 * of the dK/dV pass; f32: slot 0 = rowsum(dO*O), first B*H*N floats); dq,dk,dv [B,H,N,hd]; do_ [B,N,H*hd] */
int ldmae_attention_bwd(int dtype, const void* q, const void* k, const void* v, const void* o, const void* do_, const float* lse,
                        void* dq, void* dk, void* dv, float* delta, int B, int H, int N, int hd, float scale, void* stream);

/* The same on the PACKED token-major qkv [B,N,3,H,hd] that the qkv Linear writes (bf16; the forward also f32 at head_dim 16): the VMAE blocks have no QK-norm / RoPE
 * between the Linear and the attention (models_mae.py:133-141), so q / k / v are read, and dq / dk / dv written, in place (no head-major
 * relayout passes).  dqkv [B,N,3,H,hd]. */
int ldmae_attention_fwd_qkv(int dtype, const void* qkv, void* o, float* lse, int B, int H, int N, int hd, float scale, void* stream);
/* ldmae_attention_fwd_qkv with a static softmax shift per (batch, head) (see ldmae_attention_fwd_pv_bounded): qk_max2 [B*H][2] f32 on the
 * device = (max_i |q_i|^2, max_j |k_j|^2) over the head's rows as stored (first value 0: each query's own norm is used); heads / waves whose
 * bound exceeds 50 keep the running maximum.  The
 * tiled VMAE encoder's q | k | v kernel produces the maxima as it writes the rows. */
int ldmae_attention_fwd_qkv_bounded(int dtype, const void* qkv, void* o, float* lse, const float* qk_max2, int B, int H, int N, int hd,
                                    float scale, void* stream);
/* qk_max2 [B*H][2] = (0, max_j |k_j|^2) from one pass over the k slots of a packed bf16 qkv: a first value of 0 makes the bounded kernel use
 * each query's own norm.  One extra pass over a third of qkv: pays for long sequences (the 1024-token VMAE decoder), not for short ones. */
int ldmae_k_norm_max(const void* qkv, float* qk_max2, int B, int N, int H, int hd, void* stream);
int ldmae_attention_bwd_qkv(int dtype, const void* qkv, const void* o, const void* do_, const float* lse, void* dqkv, float* delta,
                            int B, int H, int N, int hd, float scale, void* stream);

/* Mixed form for the LightningDiT block (bf16 only): q / k (dq / dk) head-major as QK-norm + RoPE produce them, v read from -- and dv
 * written into -- the v slot of the packed token-major qkv / dqkv [B,N,3,H,hd].  Pair with ldmae_qknorm_rope_fwd(v = NULL) and
 * ldmae_qknorm_rope_bwd(dv = NULL): v never gets a head-major copy. */
int ldmae_attention_fwd_pv(int dtype, const void* q, const void* k, const void* qkv, void* o, float* lse, int B, int H, int N, int hd,
                           float scale, void* stream);
/* The same with a STATIC softmax shift: score_bound = one float on the device, a proven upper bound of |q . k| * scale * log2(e) over all
 * queries and keys.  Bounds up to 50 make the kernel skip the running maximum (exact: every exponent lies in [-2 bound, 0]; -8 % of the
 * kernel at head_dim 64); larger ones fall back to the tracked form.  ldmae_qk_score_bound gives the bound for heads that went through
 * QK-RMSNorm + RoPE (lightningdit.py:66-80) from the two norm weights alone: hd * max|wq| * max|wk| * scale * log2(e) * 1.02 (the RoPE
 * tables must be rotations, cos^2 + sin^2 = 1, as models/pos_embed.py builds them). */
int ldmae_attention_fwd_pv_bounded(int dtype, const void* q, const void* k, const void* qkv, void* o, float* lse, const float* score_bound,
                                   int B, int H, int N, int hd, float scale, void* stream);
int ldmae_qk_score_bound(const float* wq, const float* wk, int hd, float scale, float* out, void* stream);
int ldmae_attention_bwd_pv(int dtype, const void* q, const void* k, const void* qkv, const void* o, const void* do_, const float* lse,
                           void* dq, void* dk, void* dqkv, float* delta, int B, int H, int N, int hd, float scale, void* stream);
/* ldmae_attention_bwd_pv + ldmae_qknorm_rope_bwd in one (bf16, head_dim 64 / 128): the QK-RMSNorm / RoPE backward (lightningdit.py:70-75
 * read backwards) runs in the epilogues of the dQ and dK/dV kernels, dq / dk are never written head-major.  dqkv [B,N,3,H,hd] complete;
 * dwq, dwk [hd] norm-weight gradients; dbias [3*H*hd] = column sums of dqkv (the qkv Linear's bias gradient).
 * wq = wk = dwq = dwk = NULL: RoPE adjoint only in the epilogues (use_qknorm=False, see ldmae_qknorm_rope_fwd). */
long ldmae_attention_bwd_pv_qknorm_workspace_bytes(int B, int H, int N, int hd);
int ldmae_attention_bwd_pv_qknorm(int dtype, const void* q, const void* k, const void* qkv, const void* o, const void* do_, const float* lse,
                                  const float* wq, const float* wk, const float* cos, const float* sin, float eps, void* dqkv, float* dwq,
                                  float* dwk, float* dbias, float* workspace, int B, int H, int N, int hd, float scale, void* stream);

/* ---- SwiGLU (swiglu_ffn.py:34-35) ------------------------------------------------------------ */
int ldmae_swiglu_fwd(int dtype, const void* h12, void* hid, int M, int Hs, void* stream);
int ldmae_swiglu_bwd(int dtype, const void* dhid, const void* h12, void* dh12, int M, int Hs, void* stream);

/* ---- gated residual backward (lightningdit.py:248-249): dy = dxout * gate[b]; dgate[b] = sum_n dxout*y;
   dbias (optional, [D]) = column sums of dy as stored = bias gradient of the branch's output Linear (proj / w3) */
long ldmae_gate_bwd_workspace_bytes(int M, int D, int rows_per_batch);
int ldmae_gate_bwd(int dtype, const float* dxout, const void* y, const float* gate, int gate_ld, void* dy, float* dgate,
                   int dgate_ld, float* dbias, int M, int D, int rows_per_batch, float* workspace, void* stream);

/* ---- embedders (lightningdit.py:109-137, 152-169) -------------------------------------------- */
int ldmae_timestep_embedding(const float* t, float* out, int B, int dim, float max_period, void* stream);
int ldmae_silu_fwd(int out_dtype, const float* x, void* out, long n, void* stream);
int ldmae_silu_bwd(const float* dy, const float* x, float* dx, long n, void* stream);   /* dx = dy * silu'(x) */
/* out[b] = table[drop[b] ? num_classes : y[b]]; drop may be NULL */
int ldmae_label_embed_fwd(const float* table, const long long* y, const unsigned char* drop, float* out, int B, int D,
                          int num_classes, void* stream);
/* dtable[rows,D] += scatter of dout (deterministic: one workgroup per table row, fixed b order) */
int ldmae_label_embed_bwd(const float* dout, const long long* y, const unsigned char* drop, float* dtable, int B, int D,
                          int num_classes, int rows, void* stream);

/* ---- optimizer (train_accum.py:121,240 AdamW; :336-347 EMA) ---------------------------------- */
/* one pass over flat f32 buffers: AdamW(lr,b1,b2,eps,wd) on p with grad g (scaled by grad_scale), then
 * ema = decay*ema + (1-decay)*p.  step >= 1. */
int ldmae_adamw_ema(float* p, const float* g, float* m, float* v, float* ema, long n, int step, double lr, double beta1,
                    double beta2, double eps, double weight_decay, double ema_decay, double grad_scale, void* stream);
int ldmae_ema_only(float* ema, const float* p, long n, double ema_decay, void* stream);

/* ---- VMAE masked-token encoder (tokenizer/models_mae.py) ------------------------------------- */
/* random_masking (:472-497) on caller-supplied noise[N,L] f32: stable ascending argsort (ties -> lower
 * index first).  ids_restore i64 [N,L], mask f32 [N,L], ids_keep i64 [N,keep]. L <= 4096. */
int ldmae_random_masking(const float* noise, long long* ids_restore, float* mask, long long* ids_keep, int N, int L, int keep,
                         void* stream);
/* Patch-embed operand of the kept tokens only (inference; the mask depends on the noise alone, :472-497, so it can be applied BEFORE the
 * embedding conv of :502): tok [N*keep, C*p*p] (dtype tok_dtype) = the pixels of patch ids[n,j] of img [N,C,S,S] in the conv weight's
 * (c, i, j) order; posg [N*keep, D] f32 = pos[ids[n,j], :].  ldmae_gemm_nt(EPI_GATE_RES, xin = posg, gate = NULL) then gives the same bits
 * as embedding every patch and gathering. */
int ldmae_patch_gather(int tok_dtype, const float* img, const long long* ids, const float* pos, void* tok, float* posg, int N, int keep,
                       int C, int S, int p, int D, void* stream);
/* Latent-dataset prologue on the device, per batch (reference: datasets/img_latent_dataset.py:79-93 does it per item on the host; the shards
 * are written by extract_features.py:163-212).  sample = 1: moments [B, 2C, HW] f32 (mean | logvar) and noise [B, C, HW] ->
 * out[b,c,:] = ((mean + exp(0.5 * clamp(logvar, -30, 20)) * noise) - lat_mean[c]) / lat_std[c] * multiplier; sample = 0: moments is the
 * plain latent [B, C, HW] and noise is ignored.  lat_mean / lat_std [C] (latents_stats.pt) or both NULL (latent_norm off).  HW % 4 == 0. */
int ldmae_latent_prologue(const float* moments, const float* noise, const float* lat_mean, const float* lat_std, float multiplier,
                          float* out, int B, int C, int HW, int sample, void* stream);
/* out[n,j,:] = x[n, ids[n,j], :]  (torch.gather on dim 1, :486); bwd scatters (ids unique per n) */
int ldmae_gather_rows(const float* x, const long long* ids, float* out, int N, int L, int keep, int D, void* stream);
int ldmae_scatter_rows(const float* dout, const long long* ids, float* dx, int N, int L, int keep, int D, void* stream);
/* Decoder input of the pre-training step (models_mae.py:536-541: cat([x, mask_token.repeat]) -> gather(ids_restore) -> + decoder_pos_embed) in
 * one pass: out[b,l,:] = (ids_restore[b,l] < keep ? x[b, ids_restore[b,l], :] : mask_token[:]) + pos[l,:].  x [B,keep,D], out [B,L,D] f32.
 * Backward: dx [B,keep,D] = the kept rows of dout (ids_restore[b,:] is a permutation), dmask_token [D] = column sums of the other rows. */
int ldmae_restore_tokens(const float* x, const float* mask_token, const float* pos, const long long* ids_restore, float* out, int B, int L,
                         int keep, int D, void* stream);
long ldmae_restore_tokens_bwd_workspace_bytes(int B, int L, int D);
int ldmae_restore_tokens_bwd(const float* dout, const long long* ids_restore, float* dx, float* dmask_token, int B, int L, int keep, int D,
                             float* workspace, void* stream);
/* The whole encoder stack after the gather -- nblocks x Block (models_mae.py:149-187) + the closing LayerNorm (:369, 521) -- as ONE kernel
 * for the shipped geometry at mask_ratio 0.75 (tokens = 256 kept tokens per image, dim 192, 12 heads, hidden 768; anything else returns
 * LDMAE_ERR_INVALID: use the per-layer entry points).  Inference only, bf16 MFMA, f32 residual stream held in registers: one workgroup
 * per image, activations never leave the CU.  x, out [B, tokens, dim] f32.  blob: ldmae_vmae_encoder_blob_bytes(nblocks) bytes of
 * weights as pre-arranged MFMA operand fragments in order of use (layout: csrc/vmae_fused.hip; packer: tokenizer/fused_encoder.py). */
long ldmae_vmae_encoder_blob_bytes(int nblocks);
int ldmae_vmae_encoder_fwd(const float* x, float* out, const void* blob, int B, int tokens, int dim, int heads, int hidden, int nblocks,
                           float eps, void* stream);
/* The same stack on sequences of SEVERAL whole 256-token tiles per image (tokens % 256 == 0; the docking encoder `_encode` runs all 1024
 * patches: models_mae.py:819-833): the same blob, three launches per block -- LayerNorm + q|k|v of a tile (tokens on the lanes, weights
 * through the LDS ring) -> ldmae_attention_fwd_qkv on the packed qkv -> proj + residual + LayerNorm + MLP + residual of a tile with the
 * residual stream in registers.  workspace: ldmae_vmae_encoder_fwd_tiled_workspace_bytes(B, tokens) bytes, 16-B aligned (qkv, attention
 * output, lse).  x may equal out. */
long ldmae_vmae_encoder_fwd_tiled_workspace_bytes(int B, int tokens);
int ldmae_vmae_encoder_fwd_tiled(const float* x, float* out, const void* blob, void* workspace, int B, int tokens, int dim, int heads,
                                 int hidden, int nblocks, float eps, void* stream);
/* the TF32-class form of the same call (fp16 operands = TF32's mantissa; the blob packed in fp16, same layout and size): what f32 docking calls
 * run while torch.backends.cuda.matmul.allow_tf32 is set (inference.py:79, extract_features.py:2-3) */
int ldmae_vmae_encoder_fwd_tiled_f16(const float* x, float* out, const void* blob, void* workspace, int B, int tokens, int dim, int heads,
                                     int hidden, int nblocks, float eps, void* stream);
/* LayerNorm with affine (models_mae.py:163,171,369; eps 1e-6).  mean/rstd [M] saved. */
int ldmae_layernorm_fwd(int out_dtype, const float* x, const float* w, const float* b, void* out, float* mean, float* rstd,
                        int M, int D, float eps, void* stream);
long ldmae_layernorm_bwd_workspace_bytes(int M, int D);
int ldmae_layernorm_bwd(int dtype, const void* dout, const float* x, const float* w, const float* mean, const float* rstd,
                        float* dx_accum, float* dw, float* db, float beta_w, int M, int D, float* workspace, void* stream);
/* the same with a second output: dx_cast [M,D] (bf16 / fp16 = dtype; NULL: none) = dx_accum AFTER the update, rounded -- the operand of the
 * next Linear's backward in a ViT block (saves the separate cast pass over the f32 residual gradient: 0.2 GB read per block at 256 images) */
int ldmae_layernorm_bwd_cast(int dtype, const void* dout, const float* x, const float* w, const float* mean, const float* rstd,
                             float* dx_accum, void* dx_cast, float* dw, float* db, float beta_w, int M, int D, float* workspace, void* stream);
/* exact-erf GELU (timm Mlp act, models_mae.py:172) */
int ldmae_gelu_fwd(int dtype, const void* x, void* out, long n, void* stream);
int ldmae_gelu_bwd(int dtype, const void* dout, const void* x, void* dx, long n, void* stream);
/* nn.GELU(approximate="tanh"): the timm Mlp of a LightningDiT block built with use_swiglu=False (lightningdit.py:208,219-224); f32 / bf16 */
int ldmae_gelu_tanh_fwd(int dtype, const void* x, void* out, long n, void* stream);
int ldmae_gelu_tanh_bwd(int dtype, const void* dout, const void* x, void* dx, long n, void* stream);
/* 3x3 / stride 1 / pad 1 convolution on [B,C,H,W] f32 (conv_decoder_pred.conv_smoother, models_mae.py:254,275) */
int ldmae_conv3x3(const float* x, const float* w, const float* b, float* out, int B, int C, int H, int W, void* stream);
/* its backward (VMAE pre-training trains the smoother, engine_pretrain.py:51-76): dx (optional) [B,C,H,W], dw [C,C,3,3], db [C]; C = 3 */
long ldmae_conv3x3_bwd_workspace_bytes(int C);
int ldmae_conv3x3_bwd(const float* dout, const float* x, const float* w, float* dx, float* dw, float* db, int B, int C, int H, int W,
                      float* workspace, void* stream);

/* The masked / visible reconstruction loss of forward_loss (models_mae.py:733-754) in IMAGE space: pred_img = the smoothing conv's output
 * [B,C,H,W] f32, imgs the input images, mask [B, (H/p)*(W/p)] f32 (1 = masked).  fwd: partials [ldmae_mae_loss_groups(B*C*H*W)][2] =
 * per-workgroup (sum over masked patches' pixels of d^2, the same over visible ones); the caller adds the rows and divides by p*p*C * patch
 * count.  bwd: dpred_img = 2 d (coef[0] mask + coef[1] (1 - mask)), coef on the device.  norm_pix_loss targets stay in torch. */
long ldmae_mae_loss_groups(long elements);
int ldmae_mae_loss_fwd(const float* pred_img, const float* imgs, const float* mask, float* partials, int B, int C, int H, int W, int p, void* stream);
int ldmae_mae_loss_bwd(const float* pred_img, const float* imgs, const float* mask, const float* coef, float* dpred_img, int B, int C, int H, int W,
                       int p, void* stream);

/* ---- optional per-kernel timing hook used by bench.py for the roofline line ------------------- */
/* When enabled, ldmae_gemm_nt brackets each launch with HIP events on the launch stream. */
int ldmae_prof_enable(int on);
int ldmae_prof_collect(double* total_ms, double* total_flops, long* launches);   /* syncs the events; resets */
/* Launch counts by kernel family since the last reset, always on (one relaxed atomic add per entry-point call): which ARITHMETIC TYPE a
 * model's calls were dispatched to -- a bf16 forward that silently runs the f32 kernels (round 3: the VMAE decoder under autocast, 250 of
 * 304 ms) shows up as f32 counts.  counts[0..5] = NT GEMM bf16 / f32, TN GEMM bf16 / f32, attention (fwd or bwd entry) bf16 / f32;
 * counts[6..8] = NT GEMM / attention / TN GEMM in fp16 (the TF32-class forward path; VMAE pre-training under fp16 autocast).  n = how many to copy (<= 9).  reset != 0 zeroes them after the copy. */
#define LDMAE_COUNT_NT_BF16 0
#define LDMAE_COUNT_NT_F32 1
#define LDMAE_COUNT_TN_BF16 2
#define LDMAE_COUNT_TN_F32 3
#define LDMAE_COUNT_ATTN_BF16 4
#define LDMAE_COUNT_ATTN_F32 5
#define LDMAE_COUNT_NT_F16 6      /* round 5: the TF32-class (fp16) forward family */
#define LDMAE_COUNT_ATTN_F16 7
#define LDMAE_COUNT_TN_F16 8
int ldmae_launch_counts(long* counts, int n, int reset);

#ifdef __cplusplus
}
#endif
#endif
