/* fqss_experiments.h -- entry points of measured, parity-green, SLOWER experiments.  They are NOT part of libfqss_hip.so (the product):
 * `make -C fqss_amd/csrc experiments` builds variants/libfqss_experiments.so with -DFQSS_EXPERIMENTS, which tools/ and the opt-in
 * tests load through FQSS_LIB.  docs/history/DESIGN_rounds_1-5.md 9 has the measurements.
 *   fqss_gndwq_fwd_v1       GroupNormQ + 3-tap depthwise Conv1dNlQ as one launch, per-element arithmetic (round 4): bit-identical, 39.5 us
 *                           against 34.5 us for the two launches
 *   fqss_gndwq_fwd          the same on per-row code tables (round 5, k_gndwq_fwd_t: T[code] for the GroupNorm's output, V[code] for its
 *                           de-quantised value in the FIR): bit-identical, 39.6 us against 33.5 us -- the fused form is not bound by its
 *                           vector instructions (docs/history/DESIGN_rounds_1-5.md 9)
 *   FQSS_DGRAD_RING=1       the student's data-gradient q-GEMM on an LDS-DMA weight ring (csrc/experiments/qgemm_ring.hip): 33.6 / 48.9 us
 *                           against 25.8 / 41.9 us of k_qgemm<1> */
#pragma once
#include "fqss.h"
#ifdef __cplusplus
extern "C" {
#endif
/* GroupNormQ followed by a depthwise Conv1dNlQ (3 taps), both quantizing, as ONE launch: codes -> y1 (the GroupNorm's output codes, kept
 * for both backward passes) -> y2.  stats / nslots: the producer's statistics of xc; stats2 (nullable): [B][C][2] statistics of y2.
 * M <= 4096.  Bit-identical to fqss_gnq_fwd followed by fqss_dwq_fwd (convtasnetq.py:28-30 + qat_layers.py:445-448). */
int fqss_gndwq_fwd_v1(const uint8_t* xc, const float* qmin_x, const float* qmax_x, const float* gamma, const float* beta,
                   float eps, const int64_t* stats, int nslots, float* mean_rstd, uint8_t* y1, const float* qmin1,
                   const float* qmax1, const float* w, const float* bias, int dil, int pad, int act, const float* slope,
                   uint8_t* y2, const float* qmin2, const float* qmax2, int64_t* stats2, int B, int C, int M,
                   int64_t ld_xc, int64_t ld_y1, int64_t ld_y2, fqss_stream_t stream);
/* GroupNormQ followed by the 3-tap depthwise Conv1dNlQ of a TCN block (convtasnetq.py:28-30, qat_layers.py:438-452), both quantizing,
 * codes -> codes -> codes in ONE launch on per-row code tables (csrc/fused_q.hip k_gndwq_fwd_t, round 5): y1 = the GroupNorm's output
 * codes (range 1; both layers' backward reads them), y2 = the depthwise layer's (range 2); stats = the producer's integer statistics of
 * xc ([B][nslots][2]), stats2 (nullable) = [B][C][2] statistics of y2 for a GroupNormQ behind it; mean_rstd [B][2] is written.
 * M <= 4096, K = 3, pad = dil.  Bit-identical to fqss_gnq_fwd + fqss_dwq_fwd. */
int fqss_gndwq_fwd(const uint8_t* xc, const float* qmin_x, const float* qmax_x, const float* gamma, const float* beta, float eps,
                   const int64_t* stats, int nslots, float* mean_rstd, uint8_t* y1, const float* qmin1, const float* qmax1,
                   const float* w, const float* bias, int dil, int pad, int act, const float* slope, uint8_t* y2,
                   const float* qmin2, const float* qmax2, int64_t* stats2, int B, int C, int M, int64_t ld_xc, int64_t ld_y1,
                   int64_t ld_y2, fqss_stream_t stream);

#ifdef __cplusplus
}
#endif
