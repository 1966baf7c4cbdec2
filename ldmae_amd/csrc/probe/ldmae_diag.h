/* Diagnostic-build additions to the C ABI (libldmae_hip_diag.so, `make -C ldmae_amd/csrc diag`).  NOT part of the product
 * library or of include/ldmae_hip.h: process-wide A/B knobs and the tile-timeline stamp buffer used by tools/ (bench_nt.py,
 * gemm_timeline.py, tn_test.py, one_gemm.py).  Select the library with LDMAE_HIP_LIB=<repo>/ldmae_amd/libldmae_hip_diag.so. */
#ifndef LDMAE_DIAG_H
#define LDMAE_DIAG_H
#ifdef __cplusplus
extern "C" {
#endif
/* kernel-variant selection for A/B measurements.  key 0: bf16 NT GEMM variant (1 / 2 = probe/gemm_w4.hip kernels), 1: TN fallback kernel,
 * 4: TN wave layout, 5: NT start delay, 7: per-K-step stamp build, 8: NT launch mode (2 = one tile per workgroup; the product library
 * takes this per call through LDMAE_EPI_TILE_LAUNCH), 9: TN split target, 10: row-kernel grid cap, 12: 1 = deferred-epilogue NT kernel (probe/gemm_nt_defer.hip), 13: its timing-only
 * ablations (1 no deferred work, 2 loads only, 3 loads + arithmetic), 14: 1 = 4-deep NT ring, 15: 1 = whole-line NT ring (probe/gemm_nt_wl.hip),
 * 16: NT column-group tile walk (column tiles per group), 17: 1 = one-pass attention backward (attention.hip, hd 64, N % 128 == 0) instead of the
 * dQ + dK/dV kernels, 18: its timing-only ablations (1 no dQ stores / atomics, 2 no dQ product, 3 no dS image).
 * 0 = shipped behaviour. */
int ldmae_tune(int key, int value);
int ldmae_tune_query(int key);
/* device buffer (>= 64 B x 256 workgroups x tiles-per-workgroup) that receives s_memrealtime stamps of the persistent NT GEMM
 * (tile start / main loop end / epilogue issued / stores drained); NULL (default) = off */
void ldmae_debug_nt_stamps(void* buf);
#ifdef __cplusplus
}
#endif
#endif
