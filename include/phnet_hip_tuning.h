/* Benchmark and A/B switches of libphnet_hip.so - NOT part of the drop-in boundary (include/phnet_hip.h).
 *
 * Process-global, not thread-safe, never called by the product path: tests/ and tests/tools/ use them to run one kernel
 * against another on the same operands (e.g. the generic weight-gradient kernel against the three-taps one) and to sweep
 * tile / split-K plans.  Kept in a header of their own so that the public header documents only what a caller of the
 * reference's libs.models / libs.ops path needs. */
#ifndef PHNET_HIP_TUNING_H
#define PHNET_HIP_TUNING_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* force tile/split-K of the next phnet_conv2d_fwd / _dgrad calls; bm = 0 -> heuristic */
int phnet_tune_force_conv_tile(int32_t bm, int32_t bn, int32_t splits);
int phnet_tune_force_k_tile(int32_t k_tile);   /* 0 = heuristic, else 16 | 32 | 64; -1 / -2: uniform-tap kernel variant off / on;
                                                  -5 / -6: three-taps 3x3 stride-1 forward / dgrad kernel off / on */
int phnet_tune_wgrad(int32_t allow_bm128, int32_t target_blocks);   /* wgrad tile / split-K policy; bits of the first argument:
                                                                        1 = no few-rows Linear kernel, 3 = no three-taps 3x3 kernel,
                                                                        4 = its 32-pixel steps, 5 = no producer / consumer variant
                                                                        (csrc/wgrad3s.hip), 6 = no 128 x 128 producer / consumer Linear
                                                                        kernel (csrc/wgrad1s.hip); a NEGATIVE second argument sets the
                                                                        three-taps kernel's workgroup target (default 256) */
/* arithmetic of the GEMM kernels: 3 = exact three-term bf16 split staged in LDS (library default), 0 = f32-input MFMA,
 * 1 = two-term bf16 split in registers (3 MFMAs per product, ~2^-16 per product), 2 = three-term split in registers */
int phnet_tune_mma(int32_t mode);
/* packed-weight 3x3 kernel: workgroups a launch is topped up to by split-K (default 512); -1 / -2: 128-column tiles off / on */
int phnet_conv3p_tune(int32_t target_workgroups);
/* routing gate's depth-wise stack: 1 = one wavefront per plane (csrc/gate_wave.hip, default where C = 64, P = 36), 0 = the generic
 * one-workgroup-per-plane kernels (csrc/gate.hip) */
int phnet_tune_gate_wave(int32_t on);
/* per-anchor products of the dynamic head: 1 = matrix-pipe kernels, one wavefront per anchor (csrc/dyn_mfma.hip, default where they
 * apply; forward: one wavefront per anchor AND 16-row fragment, backward: four per anchor), 3 = the one-wavefront-per-anchor forms of both,
 * 0 = the LDS / FMA kernels of csrc/dynhead.hip */
int phnet_tune_dyn_mfma(int32_t on);

#ifdef __cplusplus
}
#endif
#endif
