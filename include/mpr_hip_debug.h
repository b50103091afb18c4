/* mpr_hip_debug.h -- tuning knobs and timing-experiment hooks of libmpr_hip.so.
 *
 * NOT part of the drop-in boundary (include/mpr_hip.h): nothing here is needed to run the hot path, and the
 * mpr_conv_debug_* hooks make results WRONG by design (they exist to price operand streams and kernel phases, DESIGN.md
 * section 3).  The knobs select between tile / ring variants and kernels that compute the same thing (the autotuner in
 * ops.py and the test-suite use them); all of them are process-global, return the previous setting, and are not
 * thread-safe.
 */
#ifndef MPR_HIP_DEBUG_H
#define MPR_HIP_DEBUG_H

#ifdef __cplusplus
extern "C" {
#endif

/* 3x3 / stride 1 / pad 1 convolutions with source channels % 64 == 0 (forward and data gradient) run on the
 * shifted-window kernel (conv_win.hip: the haloed activation window is loaded once per 64-channel block and all nine
 * taps read it at shifted LDS rows); 0 switches it off (tests / comparisons); returns the previous setting */
int mpr_conv_set_window(int on);

/* forward convolutions on maps narrower than `w` pixels take the LDS-DMA implicit GEMM although the shifted-window kernel is
 * eligible (its padded raster costs (W+1)(H+1) / WH of the work: +31 % at 7 x 7); 0 = never, default 8; returns the previous value */
int mpr_conv_set_window_fwd_min_width(int w);

/* tile / weight-ring variant of the shifted-window kernel (tuning knob, see conv_win.hip; default 5 | 512; bit 8: the 64 -> 64
 * filter-in-registers kernel off, bit 9: its fused data-gradient epilogues off) */
int mpr_conv_set_window_variant(int v);
/* 1: the LDS-DMA gather weight gradient on v_mfma_f32_16x16x32_bf16 (default 0: 32x32x16 -- measured, no gain); returns the previous value */
int mpr_conv_debug_wgrad_mfma16(int on);

/* workgroups of the persistent 64 -> 64 filter-in-registers kernel (conv_win_l1_kernel): 512 fill the chip once; more,
 * shorter-lived ones let a launch that starts beside another stream's kernel rebalance -- measured slower inside the step (default 512); returns the previous value */
int mpr_conv_set_l1_grid(int workgroups);

/* rows (B*P*Q) from which the LDS-DMA ring kernel replaces the register-staged one (default 16384);
 * returns the previous threshold */
int mpr_conv_set_dma_min_rows(int rows);

/* tile / ring-depth variant of the LDS-DMA kernel (tuning knob, see conv_igemm.hip; default 0, 7) */
int mpr_conv_set_variant(int narrow, int wide);

/* stride-2 data gradient: regroup rows into the 4 (h mod 2, w mod 2) classes so a tile walks only the taps that
 * reach it (default 1 = on; 0 = issue every tap with zero-filled holes); returns the previous setting */
int mpr_conv_set_dgrad_parity(int on);

/* timing experiments only (results become wrong): bit 0 drops every load of the activation operand of the LDS-DMA
 * conv kernel, bit 1 of the weight operand (zero-record buffer descriptors); returns the previous mask */
int mpr_conv_debug_drop_operand(int mask);

/* timing experiments only: the LDS-DMA conv kernel writes 4 time stamps (s_memrealtime, 100 MHz: start, prologue
 * done, main loop done, end) per workgroup into buf[4 * workgroups] (uint64, device memory); NULL switches it off */
int mpr_conv_debug_stamps(void* buf);

/* timing experiments only: the shifted-window kernel writes, per workgroup, wave 0's shader-clock sums {total,
 * waiting for DMA, waiting at the barrier, computing, epilogue, end time (100 MHz), -, -} into buf[8 * workgroups] */
int mpr_conv_debug_probe(void* buf);

/* timing experiments only: as mpr_conv_debug_stamps, for the LDS-DMA weight-gradient kernel */
int mpr_conv_debug_wgrad_stamps(void* buf);

/* timing experiments only: the sliding-window weight-gradient kernel writes wave 0's shader-clock sums {main loop,
 * waiting for DMA, barrier, issuing DMA, computing, chunks, -, -} per workgroup into buf[8 * workgroups] */
int mpr_conv_debug_wgrad_probe(void* buf);

/* output pixels (B*P*Q) from which the LDS-DMA weight-gradient kernel is used (default 16384) */
int mpr_conv_set_wgrad_dma_min_pixels(int pixels);

/* workgroups the split over pixels of the LDS-DMA weight-gradient kernel aims at (default 512 = one full round of
 * 2 per CU); returns the previous value */
int mpr_conv_set_wgrad_target_wgs(int n);

/* output tile of the LDS-DMA weight-gradient kernel on big one-tap GEMMs (transformer linears): 0 = 128 x 128 (4 waves),
 * 1 = 256 x 256 (16 waves; default), 2 = 256 x 128, 3 = 128 x 256 (8 waves); returns the previous value */
int mpr_conv_set_wgrad_tile(int v);

/* weight gradients of 3x3 / stride 1 / pad 1 convolutions (C, K multiples of 64) run on the sliding-window kernel
 * (conv_wgrad_win.hip).  Low byte: 0 = off (tests / comparisons), 1 = on (default), 2 = on with the round-1 wave layouts
 * (six waves at K = 64), 3 = on with two pixel blocks per chunk at K % 128 == 0 (experiment).  Bits 8.. are timing-
 * experiment flags: bit 8 = skip the epilogue (results wrong by design), bit 9 = per-wave instead of per-workgroup probe
 * records (mpr_conv_debug_wgrad_probe).  Returns the previous low byte */
int mpr_conv_set_wgrad_window(int on);

#ifdef __cplusplus
}
#endif
#endif /* MPR_HIP_DEBUG_H */
