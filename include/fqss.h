/*
 * fqss.h -- C ABI of libfqss_hip.so: the MI355X (gfx950) kernels of the FQSS QAT hot path.
 *
 * The reference (ssi-research/FQSS @ 2024_10_08) is pure Python/PyTorch and has no FFI of its
 * own; its drop-in surface is the `quantization.qat` module API.  This header
 * is the boundary inserted BENEATH that surface: each entry point replaces the ATen op sequence
 * of the cited reference lines.  The Python host (fqss_amd/) binds it with ctypes; the binding a
 * maintainer of the reference would add is shown in INTEGRATION.md.
 *
 * Conventions
 *  - every pointer is a caller-owned DEVICE pointer (PyTorch caching allocator); the library
 *    allocates nothing and keeps no mutable global state;
 *  - kernels are enqueued on `stream` (a hipStream_t passed as void*) and return without syncing;
 *    safe under hipGraph capture;
 *  - return value: 0 = ok, negative = error (FQSS_E*); fqss_last_error() gives the text;
 *    nothing throws across the boundary;
 *  - activations are fp32 row matrices [rows][cols] with a row stride `ld` (elements, ld >= cols);
 *    a [B][C][M] tensor is rows = B*C; rows whose `ld` is a multiple of 4 floats and whose base
 *    is 16-byte aligned take the 16-B/lane vector path;
 *  - "+=" outputs are accumulated (caller zeroes them once per step), "=" outputs are overwritten.
 */
#ifndef FQSS_H
#define FQSS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FQSS_VERSION 100 /* 0.1.0 */

#define FQSS_OK 0
#define FQSS_EINVAL (-22)
#define FQSS_ELAUNCH (-5)

/* activation-quantizer mode (GradientActivationFakeQuantize.forward, qat_quant.py:227-242) */
#define FQSS_Q_BYPASS 0  /* float teacher / act_quant=False: identity                      */
#define FQSS_Q_OBSERVE 1 /* first 50 calls: pass-through + running min/max (:228-233)      */
#define FQSS_Q_QUANT 2   /* linear_quantize asym 8 bit (:136-147)                          */

/* non-linearity fused in front of the quantizer (Conv1dNlQ / NlQ, qat_layers.py:188-219,511-518) */
#define FQSS_ACT_NONE 0
#define FQSS_ACT_PRELU 1 /* one shared slope, nn.PReLU() */
#define FQSS_ACT_RELU 2
#define FQSS_ACT_GELU 3 /* nn.GELU(), erf form: fqss_actq_fwd / fqss_actq_bwd only (the HTDemucs layers) */
#define FQSS_ACT_POST_RELU 4
#define FQSS_ACT_RELU_Q 5     /* fq2(relu(fq(z))): a LinearQ followed by NlQ(ReLU), both quantizers live (fqss_qrow_fwdq2 / fqss_actq2_bwd_colbias only) */ /* a ReLU BEHIND the quantizer, relu(fq(z)): fqss_actq_fwd / fqss_actq_bwd only (dptnetq.py:92) */

typedef void* fqss_stream_t;

int fqss_version(void);
const char* fqss_last_error(void);
/* self test: *mismatches += #{i : a[i]/b computed by the kernels' 3-instruction division != IEEE a[i]/b} */
int fqss_selftest_div(const float* a, int64_t n, float b, uint64_t* mismatches, fqss_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K1/K1b/K3  per-tensor activation fake-quant (+ fused PReLU/ReLU), observer, STE backward
 * replaces: qat_quant.py:136-147 (8 elementwise ATen passes), :228-233 (x.min()/x.max() + EMA),
 *           autograd of both; nn.PReLU / F.relu in front (qat_layers.py:211, 517)
 * ------------------------------------------------------------------------------------------- */

/* out = fq(act(z)); act: FQSS_ACT_NONE / PRELU / RELU, FQSS_ACT_GELU (erf form, nn.GELU of the HTDemucs layers) or
 * FQSS_ACT_POST_RELU (out = relu(fq(z)): the ReLU sits BEHIND the quantizer, dptnetq.py:92).  idx (optional, u8 [rows][ld_idx]) receives the integer bin index (the codes
 * the q-GEMMs / coded layers consume); with idx given, out may be NULL (codes-only fast path).  OBSERVE: out = act(z) and obs_ws[0..1] (ordered-uint min / max) are updated. */
int fqss_actq_fwd(const float* z, float* out, uint8_t* idx, int64_t rows, int64_t cols,
                  int64_t ld_z, int64_t ld_out, int64_t ld_idx, int act, const float* slope, int qmode,
                  const float* qmin, const float* qmax, uint32_t* obs_ws, fqss_stream_t stream);

/* obs_ws = {0xFFFFFFFF, 0}: must hold before the first OBSERVE launch of a call */
int fqss_obs_reset(uint32_t* obs_ws, int64_t n_pairs, fqss_stream_t stream);

/* min = alpha*min + (1-alpha)*obs_min (same for max), fp32 op order of qat_quant.py:231-232;
 * then resets obs_ws.                                                                           */
int fqss_observer_ema(float* qmin, float* qmax, uint32_t* obs_ws, double alpha, fqss_stream_t stream);

/* backward of out = fq(act(z)) given g = dL/dout:
 *   gz = dL/dz ;
 *   gacc: fp64 scratch of FQSS_GACC_SLOTS x 3 doubles, all-zero on entry: each workgroup adds its
 *         partial (dL/dmin, dL/dmax, dL/dslope) to its own slot (no same-address atomics);
 *         fqss_gacc_flush reduces the slots in a fixed order and re-zeroes them;
 *   gbias[row % C] += sum_cols gz   (optional, fp32 atomics; bias of the producing conv)         */
#define FQSS_GACC_SLOTS 2048
int fqss_actq_bwd(const float* z, const float* g, float* gz, int64_t rows, int64_t cols,
                  int64_t ld_z, int64_t ld_g, int64_t ld_gz, int act, const float* slope, int qmode,
                  const float* qmin, const float* qmax, double* gacc, float* gbias, int64_t C,
                  fqss_stream_t stream);
/* fq(GLU(z)): nn.GLU(dim = 1) of a channel-first tensor z [B][2C][M] in front of an activation quantizer (Conv1dNlQ / Conv2dNlQ /
 * ConvTranspose*NlQ with nl = GLU in the HTDemucs layers, qat_layers.py:198-293, hdemucsq.py:126-127), one pass each way:
 * out [B][C][M] = fq(z[b][c] * sigmoid(z[b][C + c])); backward gz [B][2C][M] from g [B][C][M], range partials to gacc (QUANT).
 * Rows 16-B aligned and padded to a multiple of 4 floats.  qmode as fqss_actq_fwd. */
int fqss_gluq_fwd(const float* z, float* out, int64_t B, int64_t C, int64_t M, int64_t ld_z, int64_t ld_out, int qmode,
                  const float* qmin, const float* qmax, uint32_t* obs_ws, fqss_stream_t stream);
int fqss_gluq_bwd(const float* z, const float* g, float* gz, int64_t B, int64_t C, int64_t M, int64_t ld_z, int64_t ld_g,
                  int64_t ld_gz, int qmode, const float* qmin, const float* qmax, double* gacc, fqss_stream_t stream);

/* out_k += (float) sum_slots gacc[slot][k] for the non-null outputs (k = 0 min, 1 max, 2 slope), then
 * gacc = 0: hands the fp64 range/slope partials of fqss_actq_bwd over to fp32 parameter gradients */
int fqss_gacc_flush(double* gacc, float* gmin, float* gmax, float* gslope, fqss_stream_t stream);

/* fqss_actq_bwd for the output of a ROW linear, z [R][F] with the features contiguous (F % 4 == 0, 16-B aligned rows): also accumulates
 * the linear's bias gradient gbias[f] += sum_r gz[r][f] -- replaces fqss_actq_bwd + fqss_colsum of LinearQ / LinearNlQ
 * (qat_layers.py:521-561) and of the attention output projection */
int fqss_actq_bwd_colbias(const float* z, const float* g, float* gz, int64_t R, int F, int64_t ld_z, int64_t ld_g,
                          int64_t ld_gz, int act, const float* slope, int qmode, const float* qmin,
                          const float* qmax, double* gacc, float* gbias, fqss_stream_t stream);
/* backward of fqss_qrow_fwdq2's epilogue in one pass over (z, g): NlQ's STE + ReLU (range partials -> gacc2), the linear's STE (range
 * partials -> gacc1), gz and the bias column sums -- fqss_actq_bwd(act = ReLU) followed by fqss_actq_bwd_colbias, fused */
int fqss_actq2_bwd_colbias(const float* z, const float* g, float* gz, int64_t R, int F, int64_t ld_z, int64_t ld_g, int64_t ld_gz,
                           const float* qmin1, const float* qmax1, const float* qmin2, const float* qmax2, double* gacc1, double* gacc2,
                           float* gbias, fqss_stream_t stream);

/* running min/max of a plain tensor into obs_ws (used by the splitter's global max, process.py:24) */
int fqss_minmax(const float* x, int64_t rows, int64_t cols, int64_t ld, uint32_t* obs_ws,
                fqss_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K2  per-channel symmetric weight fake-quant.  Weight layout [outer][C][inner]:
 *     ch_out_idx=0 -> outer=1 (Conv1d), ch_out_idx=1 -> outer=shape[0] (ConvTranspose1d).
 * replaces: qat_quant.py:126-135, :372-381 and their autograd
 * ------------------------------------------------------------------------------------------- */
int fqss_wq_observe(const float* w, int64_t outer, int64_t C, int64_t inner, float* qmin,
                    float* qmax, fqss_stream_t stream);
int fqss_wq_fwd(const float* w, float* wq, int8_t* idx, int64_t outer, int64_t C, int64_t inner,
                const float* qmin, const float* qmax, fqss_stream_t stream);
/* accumulate=0: gw = ; gmin[C] = ; gmax[C] =      accumulate=1: all three are "+=" */
int fqss_wq_bwd(const float* w, const float* g, float* gw, float* gmin, float* gmax,
                int64_t outer, int64_t C, int64_t inner, const float* qmin, const float* qmax,
                int accumulate, fqss_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * multi-tensor forms of the per-quantizer small work (csrc/multi.hip): ONE launch each, driven by a
 * device table of int64 words the host builds once.
 *   flush table  : n x 4  = {gacc, gmin, gmax, gslope} addresses (0 = absent)
 *   weight table : n x FQSS_WQ_DESC_WORDS = {w, wq, idx, idxT, dw, rw, qmin, qmax, gwq, gw, gmin, gmax, outer, C,
 *                  inner, first_block, ldT}; entries sorted by first_block, one workgroup per output channel;
 *                  idx/idxT/dw/rw only for pointwise-conv weights (0 otherwise); ldT = row stride of idxT
 *                  (C, or the summed C of layers whose codes are concatenated for the paired q-GEMMs);
 *                  bwd: gw += STE(gwq), gmin/gmax += range gradients  (gwq = accumulated dL/dW_q)
 * ------------------------------------------------------------------------------------------- */
#define FQSS_WQ_DESC_WORDS 17
int fqss_gacc_flush_multi(const int64_t* table, int n, fqss_stream_t stream);
int fqss_wq_multi_fwd(const int64_t* table, int n, int total_channels, fqss_stream_t stream);
int fqss_wq_multi_bwd(const int64_t* table, int n, int total_channels, fqss_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K4/K5  pointwise (k=1) Conv1d as an fp32-MFMA GEMM:  z[b] = W[Co x Ci] * x[b][Ci x M] + bias
 * replaces: F.conv1d(k=1) in Conv1dQ / Conv1dNlQ (qat_layers.py:137-146, 202-212) + autograd
 * ------------------------------------------------------------------------------------------- */
int fqss_pwconv_fwd(const float* x, const float* w, const float* bias, float* z, int B, int Ci,
                    int Co, int M, int64_t ld_x, int64_t ld_z, fqss_stream_t stream);
/* same contract as fqss_pwconv_fwd, computed on the bf16 matrix cores with both fp32 operands split
 * exactly into 3 bf16 pieces (9 exact partial products per k, fp32 accumulation): fp32-grade result
 * at 9/16 of the fp32-MFMA cost.  Needs Ci % 4 == 0 and 16-B aligned rows (csrc/qgemm.hip).        */
int fqss_pwconv_fwd_x3(const float* x, const float* w, const float* bias, float* z, int B, int Ci,
                       int Co, int M, int64_t ld_x, int64_t ld_z, fqss_stream_t stream);
/* the same with the SIX partial products above 2^-24 of each product (what fqss_rowlin_* and the
 * fp32 gradient GEMMs compute): one third fewer MFMAs, error of the order of one fp32 rounding per
 * product.  The general conv layers of HTDemucs (qat_layers.conv_frames / convtr_frames;
 * reference hdemucsq.py:72-162, 261-347) run on it, student and teacher.                          */
int fqss_pwconv_fwd_x3s(const float* x, const float* w, const float* bias, float* z, int B, int Ci,
                        int Co, int M, int64_t ld_x, int64_t ld_z, fqss_stream_t stream);
/* z[b] = (dw[co] * Wi[co][:]) x[b] + bias: the same pointwise conv when the weight is fake-quantized and given as its int8 codes
 * (fqss_wq_codes / fqss_wq_multi_fwd: wi [Co][Ci] dense, dw [Co]) while x is a plain float tensor -- the frame-path convolutions of
 * HTDemucs' student (Conv1dNlQ / Conv2dNlQ on de-quantized inputs, qat_layers.py:188-293): three bf16 products per k instead of six */
int fqss_pwconv_fwd_wq(const float* x, const int8_t* wi, const float* dw, const float* bias, float* z, int B, int Ci, int Co, int M,
                       int64_t ld_x, int64_t ld_z, fqss_stream_t stream);
/* Stride-1 1-D convolution, any kernel width / dilation / zero padding (groups = 1), as an IMPLICIT GEMM on the bf16 matrix cores
 * (six-product split, csrc/gemm_x3.hip): the B operand is read straight from the signal, one shifted row per (channel, tap) -- no
 * frame image.  replaces: nn.Conv1d of the HTDemucs DConv / rewrite layers (hdemucsq.py:72-162, demucsq.py:110-182) and its autograd:
 *   fwd   z[b][co][m] = bias[co] + sum_{ci,t} w[co][ci*taps + t] x[b][ci][m + t*dil - pad],  Mo = M + 2 pad - dil (taps - 1)
 *   dgrad the same entry on gz with the caller's flipped / transposed weight [Ci][Co*taps] and pad' = dil (taps - 1) - pad
 *   wgrad gw[co][ci*taps + t] += sum_{b,m} gz[b][co][m] x[b][ci][m + t*dil - pad]      (gw caller-zeroed)
 * weight rows (stride ld_w) and gz rows 16-B aligned.                                                                                */
int fqss_conv1d_s1_fwd(const float* x, const float* w, const float* bias, float* z, int B, int Ci, int Co, int M, int Mo, int taps,
                       int dil, int pad, int64_t ld_x, int64_t ld_w, int64_t ld_z, fqss_stream_t stream);
int fqss_conv1d_s1_bwd_w(const float* gz, const float* x, float* gw, int B, int Ci, int Co, int M, int Mo, int taps, int dil,
                         int pad, int64_t ld_gz, int64_t ld_x, fqss_stream_t stream);
/* Stride-1 convolutions of ANY kernel shape (3 x 3, 1 x 3, k with dilation; groups = 1) with WIDE outputs, on a HALO-PACKED signal, as
 * implicit GEMMs (round 6): fqss_halo_pack lays the signal out as xp [B][C][plane] -- per channel (H + 2 ph) rows of Wp floats, element
 * (h, w) at row h + ph, column w + pw, zeros everywhere else (the halo is the zero padding; Wp % 4 == 0, Wp >= W + 2 pw; plane % 4 == 0
 * with room for the last row's taps) -- so that every tap is a constant shift of the flat plane and the GEMM kernels read their B
 * operand straight from it (csrc/qgemm.hip k_qgemm<.., IMP>, csrc/gemm_x3.hip): the frame image of fqss_frames_gather (kh kw times
 * the signal, written and read back) and the overlap-add of its data gradient are never made.
 * replaces: nn.Conv2d(C, 2C, 3, 1, 1) / nn.Conv1d(C, 2C, 3, 1, 1) of the HTDemucs decoder layers (`rewrite`, hdemucsq.py:303-347) and
 * their autograd, student (weight on its int8 grid: _wq) and float teacher (_x3s).
 *   fwd    z[b][co][n] = bias[co] + sum_{ci,t} W[co][ci*taps + t] xp[b][ci][n + base + (t / kw) row_step + (t % kw) col_step]
 *          n = h Wp + w over rows_out rows of pitch Wp (N = rows_out * Wp; columns w >= the real width hold junk the caller never reads);
 *          a plain convolution: base 0, row_step dh Wp, col_step dw
 *   dgrad  the same sum over (co, t) on the packed gradient gzp (halo (kh-1) dh - ph, (kw-1) dw - pw) with the caller's regrouped
 *          weight codes [Ci][Co*taps]: base (kh-1) dh Wp + (kw-1) dw, row_step -dh Wp, col_step -dw
 *   wgrad  gw[co][ci*taps + t] += sum_{b,m} gzp[b][co][m] xp[b][ci][m + (t / kw) row_step + (t % kw) col_step - off]   (gw caller-zeroed;
 *          off = the gradient's own halo offset, (ph_g Wp + pw_g); zero outside the plane) */
int fqss_halo_pack(const float* x, float* xp, int64_t B, int64_t C, int64_t H, int64_t W, int64_t sb, int64_t sc, int64_t sh,
                   int ph, int pw, int64_t Wp, int64_t plane, fqss_stream_t stream);
/* Strided convolutions along one axis (kernel k = T s taps, stride s, padding p: the k8 s4 p2 encoder / decoder layers) on the same
 * kernels: the signal phase-packed -- xp[b][c s + r][m] = x[b][c][s (m + q0(r)) + r], rows of Wp floats, zeros outside -- is the input of
 * a stride-1 convolution with T taps over s C channels whose weight is the conv's own, regrouped (ci, r, q') <- (ci, t = t0(r) + s q');
 * fqss_phase_unpack is the inverse move (data gradient back to the signal's layout; a transposed convolution's output, + bias, onto a
 * window that starts `off` positions in).  axis 0: along H of [B][C][H][W] with (k, 1) kernels; axis 1: along W (H = 1). */
int fqss_phase_pack(const float* x, float* xp, int64_t B, int64_t C, int64_t H, int64_t W, int64_t sb, int64_t sc, int64_t sh, int axis,
                    int s, int p, int64_t Wp, int64_t plane, fqss_stream_t stream);
int fqss_phase_unpack(const float* gy, float* gx, int64_t B, int64_t C, int64_t H, int64_t W, int64_t sb, int64_t sc, int64_t sh, int axis,
                      int s, int p, int64_t Hy, int64_t Wp, int64_t plane, int off, const float* bias, fqss_stream_t stream);
int fqss_conv2_fwd_wq(const float* xp, const int8_t* wi, const float* dw, const float* bias, float* z, int B, int Ci, int Co, int taps,
                      int kw, int base, int row_step, int col_step, int64_t N, int64_t plane_in, int64_t plane_out, fqss_stream_t stream);
int fqss_conv2_fwd_x3s(const float* xp, const float* w, const float* bias, float* z, int B, int Ci, int Co, int taps, int kw, int base,
                       int row_step, int col_step, int64_t N, int64_t plane_in, int64_t plane_out, fqss_stream_t stream);
int fqss_conv2_bwd_x_wq(const float* gzp, const int8_t* wiT, const float* dw, float* gx, int B, int Ci, int Co, int taps, int kw,
                        int base, int row_step, int col_step, int64_t N, int64_t plane_in, int64_t plane_out, fqss_stream_t stream);
int fqss_conv2_bwd_w(const float* gzp, const float* xp, float* gw, int B, int Ci, int Co, int taps, int kw, int row_step, int col_step,
                     int off, int64_t plane_g, int64_t plane_x, fqss_stream_t stream);
/* gx[b] = W^T * gz[b] */
int fqss_pwconv_bwd_x(const float* gz, const float* w, float* gx, int B, int Ci, int Co, int M,
                      int64_t ld_gz, int64_t ld_gx, fqss_stream_t stream);
/* gw[Co][Ci] += sum_b gz[b] * x[b]^T */
int fqss_pwconv_bwd_w(const float* gz, const float* x, float* gw, int B, int Ci, int Co, int M,
                      int64_t ld_gz, int64_t ld_x, fqss_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K4q/K5q  the same pointwise conv for operands on 8-bit grids (student, quantizing phase), on the
 * bf16 matrix cores (csrc/qgemm.hip):  z = dw[co]*(dx*sum_ci Wi*c + min_x*sum_ci Wi) + bias  -- the
 * integer sum is exact in the fp32 accumulators.  Gradients are split exactly into 3 bf16 pieces.
 *   xc : u8 activation codes [B][Ci][ld_xc] (fqss_actq_fwd idx), qmin_x/qmax_x its quantizer ranges
 *   wi : int8 weight codes [Co][Ci], wiT [Ci][Co], dw[Co] = delta_w, rw[Co] = sum_ci Wi  (fqss_wq_codes)
 * ------------------------------------------------------------------------------------------- */
int fqss_wq_codes(const float* w, int8_t* idx, int8_t* idxT, float* dw, float* rw, int Co, int Ci,
                  const float* qmin, const float* qmax, fqss_stream_t stream);
int fqss_qpw_fwd(const uint8_t* xc, const int8_t* wi, const float* dw, const float* rw,
                 const float* bias, const float* qmin_x, const float* qmax_x, float* z, int B, int Ci,
                 int Co, int M, int64_t ld_xc, int64_t ld_z, fqss_stream_t stream);
/* gx[b] = W_q^T gz[b] */
int fqss_qpw_bwd_x(const float* gz, const int8_t* wiT, const float* dw, float* gx, int B, int Ci,
                   int Co, int M, int64_t ld_gz, int64_t ld_gx, fqss_stream_t stream);
/* gw[Co][Ci] += sum_b gz[b] x[b]^T   (gradient w.r.t. the fake-quantized weight) */
int fqss_qpw_bwd_w(const float* gz, const uint8_t* xc, const float* qmin_x, const float* qmax_x,
                   float* gw, int B, int Ci, int Co, int M, int64_t ld_gz, int64_t ld_xc,
                   fqss_stream_t stream);
/* Two pointwise convs on the SAME coded input (the res | skip convs of a TCN block, convtasnetq.py:41-42 /
 * reference conv_tasnet ConvBlock) as ONE GEMM over the concatenated output channels: wi [Co1+Co2][Ci],
 * wiT [Ci][Co1+Co2], dw/rw [Co1+Co2], gw [Co1+Co2][Ci]; the per-layer activations / gradients stay separate
 * tensors.  bwd_x2 returns the SUM of both layers' input gradients (autograd's accumulation of the fork). */
/* fqss_qpw_bwd_x + an addend of the output's shape in the epilogue (gradient of the other branch of a residual fork) */
int fqss_qpw_bwd_x_add(const float* gz, const int8_t* wiT, const float* dw, const float* addend, float* gx, int B, int Ci,
                       int Co, int M, int64_t ld_gz, int64_t ld_add, int64_t ld_gx, fqss_stream_t stream);
int fqss_qpw_fwd2(const uint8_t* xc, const int8_t* wi, const float* dw, const float* rw, const float* bias1,
                  const float* bias2, const float* qmin_x, const float* qmax_x, float* z1, float* z2, int B,
                  int Ci, int Co1, int Co2, int M, int64_t ld_xc, int64_t ld_z1, int64_t ld_z2,
                  fqss_stream_t stream);
int fqss_qpw_bwd_x2(const float* gz1, const float* gz2, const int8_t* wiT, const float* dw, float* gx, int B,
                    int Ci, int Co1, int Co2, int M, int64_t ld_gz1, int64_t ld_gz2, int64_t ld_gx,
                    fqss_stream_t stream);
int fqss_qpw_bwd_w2(const float* gz1, const float* gz2, const uint8_t* xc, const float* qmin_x,
                    const float* qmax_x, float* gw, int B, int Ci, int Co1, int Co2, int M, int64_t ld_gz1,
                    int64_t ld_gz2, int64_t ld_xc, fqss_stream_t stream);

/* Weight gradients of SEVERAL quantized 1x1 convolutions in ONE launch (csrc/qgemm.hip k_qwgrad_group; round 5): the gradients of
 * a backward segment's Conv1dQ / Conv1dNlQ weights feed nothing but the optimizer (autograd of F.conv1d in qat_layers.py:137-146,
 * 202-212), so the host may queue them and run them together.  Per job gw [Co1+Co2][Ci] += sum_b [gz1; gz2][b] x[b]^T exactly as
 * fqss_qpw_bwd_w / _w2 (Co2 = 0, gz2 = NULL: a single layer).  No float atomics: tiles cut by the work split are reduced through
 * slab slots in a fixed order, so two runs give the same bits.  `jobs` is a HOST array (copied into the launch, <= 25 jobs per
 * launch, more are split); two jobs of one call must not share gw.  ws: device memory of fqss_qpw_bwd_w_group_ws(jobs, njobs)
 * bytes, 16-B aligned, ZERO-FILLED before its first use (its first 64 KB are arrival tickets that every launch leaves zero);
 * launches that share a workspace must be ordered (one stream). */
typedef struct FqssWgradJob {
    const float* gz1; const float* gz2;          /* [B][Co1][ld_gz1], [B][Co2][ld_gz2] (NULL when Co2 == 0) */
    const uint8_t* xc;                           /* input codes [B][Ci][ld_xc] */
    const float* qmin_x; const float* qmax_x;    /* the input quantizer's range (device scalars) */
    float* gw;                                   /* [Co1+Co2][Ci] dense, accumulated */
    int32_t B, Ci, Co1, Co2, M;
    int64_t ld_gz1, ld_gz2, ld_xc;
} FqssWgradJob;
int64_t fqss_qpw_bwd_w_group_ws(const FqssWgradJob* jobs, int njobs);      /* -1: bad job list (fqss_last_error) */
int fqss_qpw_bwd_w_group(const FqssWgradJob* jobs, int njobs, void* ws, int64_t ws_bytes, fqss_stream_t stream);
/* fqss_qpw_fwd / fqss_qpw_fwd2 with the layer's own activation + output fake-quant fused into the epilogue
 * (Conv1dQ / Conv1dNlQ: conv -> nl -> activation_fake_quantize, qat_layers.py:137-146, 202-212): writes the
 * pre-quant z (kept for the backward) AND the u8 codes yc of fq(act(z)); Co2 = 0 for a single layer */
int fqss_qpw_fwdq(const uint8_t* xc, const int8_t* wi, const float* dw, const float* rw, const float* bias1,
                  const float* bias2, const float* qmin_x, const float* qmax_x, float* z1, float* z2, int act,
                  const float* slope, const float* qmin1, const float* qmax1, const float* qmin2,
                  const float* qmax2, uint8_t* yc1, uint8_t* yc2, int B, int Ci, int Co1, int Co2, int M,
                  int64_t ld_xc, int64_t ld_z1, int64_t ld_z2, int64_t ld_yc1, int64_t ld_yc2,
                  int64_t* stats1, fqss_stream_t stream);
/* stats1 (nullable, Co2 = 0 only): exact integer statistics (sum c, sum c^2) of the output codes yc1, one slot per workgroup,
 * [B][fqss_qpw_stat_slots(Co1, M)][2] int64 -- handed to fqss_gnq_fwd when the consumer is a GroupNormQ ("stats can
 * be produced by the previous kernel's epilogue").  fqss_*_stat_slots return 0 when that shape cannot emit them. */
int fqss_qpw_stat_slots(int Co, int M);
int fqss_dwq_stat_slots(int C, int M);
/* fqss_qpw_fwdq (no activation) with the AddQ layers that are the ONLY consumers of its outputs evaluated in the same epilogue:
 * replaces `self.add(x_res, residual)` of the TCN block and `output = self.adds[i](output, skip)` of the mask generator
 * (/root/reference/quantization/qat/models/convtasnetq.py:41, 110; AddQ: qat_layers.py:62-72) as launches of their own.
 * add1 sits behind output 1, add2 behind output 2, either may be NULL.  y = fq_q(dec_a(a) + dec(yc)), a [B][Co][ld_a] u8 codes of
 * the other operand under (amin, amax), (qmin, qmax) the AddQ's quantizer, y [B][Co][ld_y]: the codes fqss_ewq_fwd returns for the
 * same operands, bit for bit (tests/test_gpu_kernels.py::test_pair_forward_with_fused_adds). */
typedef struct {
    const uint8_t* a; int64_t ld_a;
    const float *amin, *amax, *qmin, *qmax;
    uint8_t* y; int64_t ld_y;
} FqssAddAfter;
int fqss_qpw_fwdq_add(const uint8_t* xc, const int8_t* wi, const float* dw, const float* rw, const float* bias1,
                      const float* bias2, const float* qmin_x, const float* qmax_x, float* z1, float* z2, const float* qmin1,
                      const float* qmax1, const float* qmin2, const float* qmax2, uint8_t* yc1, uint8_t* yc2, int B, int Ci,
                      int Co1, int Co2, int M, int64_t ld_xc, int64_t ld_z1, int64_t ld_z2, int64_t ld_yc1, int64_t ld_yc2,
                      const FqssAddAfter* add1, const FqssAddAfter* add2, fqss_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K6  depthwise dilated Conv1d (groups = C): z[b][c][m] = bias[c] + sum_k w[c][k] x[b][c][m+k*dil-pad]
 * replaces: F.conv1d(groups=C) of convtasnetq.py:28-30 + autograd
 * ------------------------------------------------------------------------------------------- */
int fqss_dwconv_fwd(const float* x, const float* w, const float* bias, float* z, int B, int C,
                    int M, int K, int dil, int pad, int64_t ld_x, int64_t ld_z, fqss_stream_t stream);
int fqss_dwconv_bwd_x(const float* gz, const float* w, float* gx, int B, int C, int M, int K,
                      int dil, int pad, int64_t ld_gz, int64_t ld_gx, fqss_stream_t stream);
/* gw[C][K] += */
int fqss_dwconv_bwd_w(const float* gz, const float* x, float* gw, int B, int C, int M, int K,
                      int dil, int pad, int64_t ld_gz, int64_t ld_x, fqss_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K7  GroupNorm(num_groups=1, C, eps) : per-sample statistics over C*M
 * replaces: nn.GroupNorm in GroupNormQ (qat_layers.py:445-448) + autograd
 * ws: fp64 scratch, fwd needs 2*B doubles, bwd needs 2*B*C + 2*B doubles (callee zeroes what it needs)
 * ------------------------------------------------------------------------------------------- */
int fqss_gn_fwd(const float* x, const float* gamma, const float* beta, float* z, float* mean_rstd,
                int B, int C, int M, int64_t ld_x, int64_t ld_z, float eps, double* ws,
                fqss_stream_t stream);
/* Forward-only GroupNorm(1, C) whose apply pass carries what follows it (csrc/stream_ops.hip k_gn_tail; round 5) -- for networks run
 * without autograd: the frozen float teacher's DConv layers of HTDemucs (demucsq.py:163-182: GroupNorm -> GELU, and GroupNorm -> GLU ->
 * LayerScale -> + residual).  tail 1: y [B][C][M] = gelu(gn(x)); tail 2: y [B][C/2][M] = (a * sigmoid(g)) * ls[c] + res[b][c] with
 * a / g the two channel halves of gn(x).  Value for value the separate kernels (fqss_gn_fwd, fqss_unary_fwd / fqss_glu_fwd,
 * fqss_chan_op, fqss_axpby).  ws: 2 * B doubles. */
int fqss_gn_fwd_tail(const float* x, const float* gamma, const float* beta, float* y, int B, int C, int M, int64_t ld_x, int64_t ld_y,
                     float eps, double* ws, int tail, const float* ls, const float* res, int64_t ld_res, fqss_stream_t stream);
/* gx = ; ggamma[C] += ; gbeta[C] += */
int fqss_gn_bwd(const float* gz, const float* x, const float* gamma, const float* mean_rstd,
                float* gx, float* ggamma, float* gbeta, int B, int C, int M, int64_t ld_gz,
                int64_t ld_x, int64_t ld_gx, double* ws, fqss_stream_t stream);
/* GroupNormQ (qat_layers.py:427-452: GroupNorm(1, C) + its output quantizer) on a FLOAT input in the quantizing phase as the GroupNorm's own
 * passes: y = fq(GroupNorm(x)) from the statistics pass and ONE apply pass (fqss_gn_fwd + fqss_actq_fwd value for value; the pre-quant z is
 * not stored), and its backward with the STE and the range partials (gacc: FQSS_GACC_SLOTS x 3, as fqss_actq_bwd) inside the two data passes
 * of fqss_gn_bwd.  ws: 2 B doubles (forward) / 2 B C + 2 B doubles (backward) */
int fqss_gnq_fwd_f(const float* x, const float* gamma, const float* beta, float* y, uint8_t* yc /* codes of y, nullable */, float* mean_rstd,
                   int B, int C, int M, int64_t ld_x, int64_t ld_y, int64_t ld_yc, float eps, double* ws, const float* qmin, const float* qmax,
                   fqss_stream_t stream);
int fqss_gnq_bwd_f(const float* g, const float* x, const float* gamma, const float* beta, const float* mean_rstd, float* gx, float* ggamma,
                   float* gbeta, int B, int C, int M, int64_t ld_g, int64_t ld_x, int64_t ld_gx, double* ws, const float* qmin,
                   const float* qmax, double* gacc, fqss_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K6q/K7q  "codes-only" streaming layers (csrc/fused_q.hip): input = u8 codes + its quantizer ranges,
 * the layer's own PReLU + fake-quant fused, output = u8 codes (yout: optional fp32 copy, NULL in the
 * fast path), nothing saved -- the backward recomputes the pre-quant value from the input codes.
 * Code rows are 16-B aligned (ld % 16 == 0).  ws: gnq_fwd 2*64*B int64, gnq_bwd 2*B*C + 2*B doubles.
 * ------------------------------------------------------------------------------------------- */
int fqss_decode(const uint8_t* codes, float* out, int64_t rows, int64_t cols, int64_t ld_c,
                int64_t ld_out, const float* qmin, const float* qmax, fqss_stream_t stream);
int fqss_gnq_fwd(const uint8_t* xc, const float* qmin_x, const float* qmax_x, const float* gamma,
                 const float* beta, uint8_t* yc, float* yout, float* mean_rstd, int B, int C, int M,
                 int64_t ld_xc, int64_t ld_yc, int64_t ld_out, float eps, const float* qmin,
                 const float* qmax, void* ws, const int64_t* stats, int nslots, fqss_stream_t stream);
/* stats / nslots: the integer statistics of xc as [B][nslots][2] partial sums when its producer emitted them (fqss_qpw_fwdq,
 * fqss_dwq_fwd); NULL: one statistics pass over xc into ws.  mean / rstd are finished inside the apply pass either way. */
int fqss_gnq_bwd(const uint8_t* xc, const float* qmin_x, const float* qmax_x, const float* g,
                 const float* gamma, const float* beta, const float* mean_rstd, float* gx,
                 float* ggamma, float* gbeta, int B, int C, int M, int64_t ld_xc, int64_t ld_g,
                 int64_t ld_gx, const float* qmin, const float* qmax, double* gacc, double* ws,
                 fqss_stream_t stream);
/* fqss_gnq_bwd that ALSO runs the epilogue backward of the layer that produced xc (its pre-quant output pz, its
 * non-linearity; its output quantizer is (qmin_x, qmax_x)): writes that producer's gz instead of gx and accumulates
 * its range/slope partials (pgacc) and bias gradient (pgbias[C], nullable) -- replaces a separate fqss_actq_bwd pass */
int fqss_gnq_bwd_p(const uint8_t* xc, const float* qmin_x, const float* qmax_x, const float* g,
                   const float* gamma, const float* beta, const float* mean_rstd, float* gz, float* ggamma,
                   float* gbeta, int B, int C, int M, int64_t ld_xc, int64_t ld_g, int64_t ld_gz,
                   const float* qmin, const float* qmax, double* gacc, double* ws, const float* pz,
                   int64_t ld_pz, int pact, const float* pslope, double* pgacc, float* pgbias,
                   fqss_stream_t stream);

int fqss_dwq_fwd(const uint8_t* xc, const float* qmin_x, const float* qmax_x, const float* w,
                 const float* bias, uint8_t* yc, float* yout, int B, int C, int M, int K, int dil,
                 int pad, int64_t ld_xc, int64_t ld_yc, int64_t ld_out, int act, const float* slope,
                 const float* qmin, const float* qmax, int64_t* stats, fqss_stream_t stream);
/* stats (nullable): integer statistics of yc, [B][fqss_dwq_stat_slots(C, M)][2] int64 (see fqss_qpw_fwdq) */
/* gz = dL/d(conv output) recomputed from the input codes; gacc slots += range/slope partials; gbias[C] += */
int fqss_dwq_bwd_z(const uint8_t* xc, const float* qmin_x, const float* qmax_x, const float* w,
                   const float* bias, const float* g, float* gz, int B, int C, int M, int K, int dil,
                   int pad, int64_t ld_xc, int64_t ld_g, int64_t ld_gz, int act, const float* slope,
                   const float* qmin, const float* qmax, double* gacc, float* gbias,
                   fqss_stream_t stream);
/* element-wise layer on codes: y = fq(act(dec(a) + sb * B)), B = dec(bc) | bf (fp32) | absent  -- AddQ,
 * ResidualErrorBlock's Sub, NlQ.  bwd: gz = dL/dz recomputed from the codes (ga = gz, gb = sb*gz). */
int fqss_ewq_fwd(const uint8_t* ac, const float* amin, const float* amax, const uint8_t* bc,
                 const float* bmin, const float* bmax, const float* bf, float sb, uint8_t* yc, float* yout,
                 int64_t rows, int64_t cols, int64_t ld_a, int64_t ld_b, int64_t ld_bf, int64_t ld_y,
                 int64_t ld_out, int act, const float* slope, const float* qmin, const float* qmax,
                 fqss_stream_t stream);
int fqss_ewq_bwd(const uint8_t* ac, const float* amin, const float* amax, const uint8_t* bc,
                 const float* bmin, const float* bmax, const float* bf, float sb, const float* g, float* gz,
                 int64_t rows, int64_t cols, int64_t ld_a, int64_t ld_b, int64_t ld_bf, int64_t ld_g,
                 int64_t ld_gz, int act, const float* slope, const float* qmin, const float* qmax,
                 double* gacc, fqss_stream_t stream);
/* fqss_ewq_bwd for operands that are the fake-quantized outputs of pointwise convs (the res / skip convs of a TCN block
 * feeding AddQ, convtasnetq.py:42, 111-112): p?_z != NULL additionally runs THAT layer's epilogue backward on
 * dL/d(operand) -- its gz goes to p?_out, its range/slope partials to p?_gacc, its bias gradient to p?_gbias[C]
 * (nullable).  The operand's quantizer is (amin,amax) / (bmin,bmax).  gz may be NULL when both operands are fused. */
int fqss_ewq_bwd_p(const uint8_t* ac, const float* amin, const float* amax, const uint8_t* bc,
                   const float* bmin, const float* bmax, float sb, const float* g, float* gz, int64_t rows,
                   int64_t cols, int64_t ld_a, int64_t ld_b, int64_t ld_g, int64_t ld_gz, int act,
                   const float* slope, const float* qmin, const float* qmax, double* gacc, int C,
                   const float* pa_z, int64_t ld_paz, int pa_act, const float* pa_slope, double* pa_gacc,
                   float* pa_gbias, float* pa_out, int64_t ld_pa_out, const float* pb_z, int64_t ld_pbz,
                   int pb_act, const float* pb_slope, double* pb_gacc, float* pb_gbias, float* pb_out,
                   int64_t ld_pb_out, fqss_stream_t stream);
/* MulQ on codes: masked[b][s][c][:] = fq(mask[b][s][c][:] * feat[b][c][:]) -- the masking product of ConvTasNetQ.forward
 * (quantization/qat/models/convtasnetq.py:277 `self.mul(...)`, qat_layers.py:134-153 `MulQ`).  mc: [B*S*C rows], fc: [B*C rows],
 * 1 <= S <= 4.  Same arithmetic as fqss_decode x 2 -> fqss_mul_bcast_fwd -> fqss_actq_fwd: bit-identical codes.  yout nullable. */
int fqss_mulq_fwd(const uint8_t* mc, const float* mmin, const float* mmax, const uint8_t* fc, const float* fmin,
                  const float* fmax, uint8_t* yc, float* yout, int B, int S, int C, int M, int64_t ld_m, int64_t ld_f,
                  int64_t ld_y, int64_t ld_out, const float* qmin, const float* qmax, fqss_stream_t stream);
/* its backward (= fqss_actq_bwd + fqss_mul_bcast_bwd on recomputed values): gmask [B*S*C rows], gfeat [B*C rows] (nullable),
 * range partials to gacc.  pz != NULL: the mask is the fake-quantized output of a pointwise conv (its quantizer = (mmin, mmax))
 * whose epilogue backward runs here as in fqss_ewq_bwd_p: gmask then receives THAT layer's gz, its partials go to pgacc, its
 * bias gradient to pgbias[S*C] (nullable); pact: FQSS_ACT_NONE / RELU / PRELU. */
int fqss_mulq_bwd(const uint8_t* mc, const float* mmin, const float* mmax, const uint8_t* fc, const float* fmin,
                  const float* fmax, const float* g, float* gmask, float* gfeat, int B, int S, int C, int M, int64_t ld_m,
                  int64_t ld_f, int64_t ld_g, int64_t ld_gm, int64_t ld_gf, const float* qmin, const float* qmax,
                  double* gacc, const float* pz, int64_t ld_pz, int pact, const float* pslope, double* pgacc,
                  float* pgbias, fqss_stream_t stream);
/* gw[C][K] += */
int fqss_dwq_bwd_w(const float* gz, const uint8_t* xc, const float* qmin_x, const float* qmax_x,
                   float* gw, int B, int C, int M, int K, int dil, int pad, int64_t ld_gz,
                   int64_t ld_xc, fqss_stream_t stream);
/* the three of them in one launch (one workgroup per row keeps gz in LDS): gx (nullable), gw += , gbias += , gacc slots += ;
 * rows up to 12288 positions */
int fqss_dwq_bwd(const uint8_t* xc, const float* qmin_x, const float* qmax_x, const float* w,
                 const float* bias, const float* g, float* gx, float* gw, int B, int C, int M, int K,
                 int dil, int pad, int64_t ld_xc, int64_t ld_g, int64_t ld_gx, int act, const float* slope,
                 const float* qmin, const float* qmax, double* gacc, float* gbias, fqss_stream_t stream);

/* ---- GroupNormQ <-> depthwise Conv1dNlQ backward hand-over (round 5; csrc/fused_q.hip k_dwq_bwd<3, GA, GB>).  In a TCN block the
 * depthwise layer sits between two GroupNormQ layers (convtasnetq.py:28-30, 37-42; qat_layers.py:438-452).  Their two-pass backward
 * is bound by its bytes; the depthwise layer's backward owns a whole (b, c) row per workgroup and recomputes its output code anyway,
 * so it takes the APPLY pass of the GroupNormQ behind it (on the incoming gradient, as it is loaded) and the ROWS pass of the
 * GroupNormQ in front of it (on gx, as it is produced).  Bit-identical to the separate passes except for the order of fp64 slot
 * atomics.  fqss_gnq_bwd_rows / fqss_gnq_bwd_apply are the two passes of fqss_gnq_bwd(_p) on their own; ws [B*C][2] doubles. */
typedef struct FqssGnAfter {          /* the GroupNormQ that consumes the depthwise layer's output */
    const float* gamma; const float* beta; const float* mean_rstd;   /* [C], [C], [B][2] */
    const double* ws;                 /* its rows pass' (ds, db) per row */
    const float* qmin; const float* qmax;                            /* its output quantizer */
    float* ggamma; float* gbeta;      /* [C], accumulated */
} FqssGnAfter;
typedef struct FqssGnBefore {         /* the GroupNormQ that produced the depthwise layer's input */
    const uint8_t* xc0; int64_t ld_xc0; const float* qmin0; const float* qmax0;   /* ITS input codes [B*C][ld_xc0] and their range */
    const float* gamma; const float* beta; const float* mean_rstd;
    double* ws;                       /* out: (ds, db) per row for fqss_gnq_bwd_apply */
    double* gacc;                     /* partial slots of its output quantizer */
} FqssGnBefore;
int fqss_gnq_bwd_rows(const uint8_t* xc, const float* qmin_x, const float* qmax_x, const float* g, const float* gamma,
                      const float* beta, const float* mean_rstd, int B, int C, int M, int64_t ld_xc, int64_t ld_g,
                      const float* qmin, const float* qmax, double* gacc, double* ws, fqss_stream_t stream);
/* pz NULL: plain gx; else the producer form of fqss_gnq_bwd_p */
int fqss_gnq_bwd_apply(const uint8_t* xc, const float* qmin_x, const float* qmax_x, const float* g, const float* gamma,
                       const float* beta, const float* mean_rstd, float* gx, float* ggamma, float* gbeta, int B, int C, int M,
                       int64_t ld_xc, int64_t ld_g, int64_t ld_gx, const float* qmin, const float* qmax, const double* ws,
                       const float* pz, int64_t ld_pz, int pact, const float* pslope, double* pgacc, float* pgbias,
                       fqss_stream_t stream);
int fqss_dwq_bwd_gn(const uint8_t* xc, const float* qmin_x, const float* qmax_x, const float* w, const float* bias,
                    const float* g, float* gx, float* gw, int B, int C, int M, int K, int dil, int pad, int64_t ld_xc,
                    int64_t ld_g, int64_t ld_gx, int act, const float* slope, const float* qmin, const float* qmax,
                    double* gacc, float* gbias, const FqssGnAfter* after, const FqssGnBefore* before, fqss_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * fused float-teacher chain (csrc/teacher.hip): inference only, frozen weights pre-split into three
 * exact bf16 planes; per TCN block T1 (fqss_tgemm: conv + PReLU + stats), T2 (fqss_tdw: GN-apply +
 * depthwise conv + PReLU + stats), T3 (fqss_tgemm: res+skip as one GEMM, GN-apply prologue, residual /
 * skip accumulation epilogue).  stats: fp64 [B][FQSS_TSTAT_SLOTS][FQSS_TSTAT_STRIDE], the first two
 * doubles of every slot hold a partial (sum, sum^2) of the sample, "+=" (caller zeroes); consumers add the
 * slots up (one slot per 128-B line keeps the fp64 atomics off a single address).
 * replaces: the plain nn.Module forward of the deep-copied float model (mysystem.py:132-133)
 * ------------------------------------------------------------------------------------------- */
#define FQSS_TSTAT_SLOTS 32
#define FQSS_TSTAT_STRIDE 16
int fqss_split3_planes(const float* w, uint16_t* planes, int64_t n, fqss_stream_t stream);
/* pro: 0 none | 1 GroupNorm(1,Ci) affine from pro_stats/gamma/beta on the input | 2 PReLU(pro_slope) on it.
 * rows [0,M1) -> c1 (+ r1), rows [M1,Co) -> c2 (+ r2); stats_out: statistics of everything written */
int fqss_tgemm(const uint16_t* planes, const float* x, int B, int Ci, int Co, int M, int64_t ld_x, int pro,
               const double* pro_stats, const float* pro_gamma, const float* pro_beta, float pro_eps,
               const float* pro_slope, const float* bias, int act, const float* slope, double* stats_out,
               int M1, float* c1, const float* r1, int64_t ld_c1, float* c2, const float* r2,
               int64_t ld_c2, fqss_stream_t stream);
/* The same GEMM on a TILED weight image (round 4, k_tgemm2): fqss_split3_tiles lays the three bf16 planes out as
 * [Co/256][Ci/16][3][256][16] (one k-tile of one 256-row tile = 24 KB contiguous, the two 16-B chunks of a row swapped where
 * (row >> 3) & 1) so that the kernel moves them global -> LDS by LDS-DMA in whole cache lines.  Needs fqss_tgemm_tiled_ok(Ci, Co,
 * M1) (Co % 256 == 0, Ci % 128 == 0, Ci <= 512, M1 % 32 == 0); every other argument as fqss_tgemm.  The accumulation starts
 * at the bias (fqss_tgemm adds it last): the two forms agree to fp32 rounding, not bit for bit. */
int fqss_tgemm_tiled_ok(int Ci, int Co, int M1);
int fqss_split3_tiles(const float* w, uint16_t* tiles, int Co, int Ci, fqss_stream_t stream);
int fqss_tgemm_tiled(const uint16_t* tiles, const float* x, int B, int Ci, int Co, int M, int64_t ld_x, int pro,
                     const double* pro_stats, const float* pro_gamma, const float* pro_beta, float pro_eps,
                     const float* pro_slope, const float* bias, int act, const float* slope, double* stats_out,
                     int M1, float* c1, const float* r1, int64_t ld_c1, float* c2, const float* r2,
                     int64_t ld_c2, fqss_stream_t stream);
int fqss_tdw(const float* x, const double* stats_in, const float* gamma, const float* beta, float eps,
             const float* w, const float* bias, const float* slope, float* y, double* stats_out, int B,
             int C, int M, int K, int dil, int pad, int64_t ld_x, int64_t ld_y, fqss_stream_t stream);
int fqss_tstats(const float* x, int B, int C, int M, int64_t ld, double* ws, fqss_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * BatchNorm1d / BatchNorm2d under BatchNormQ (csrc/batchnorm.hip) on channel-first [B][C][M] tensors (M = H*W for the 2-D form).
 * replaces: nn.BatchNorm1d/2d inside BatchNormQ (qat_layers.py:472-486; qat_utils.py:163, 381-382) + autograd
 *   fqss_bn_moments:    out[c] += (sum x, sum x^2) over (b, m), fp64 [C][2], caller zeroes       (batch statistics, training mode)
 *   fqss_bn_apply:      y = x * a[c] + b[c]
 *   fqss_bn_bwd_reduce: out[c] += (sum g, sum g x), fp64 [C][2], caller zeroes
 *   fqss_bn_bwd_apply:  gx = g * c1[c] + x * c2[c] + c3[c]
 * ------------------------------------------------------------------------------------------- */
int fqss_bn_moments(const float* x, double* out, int B, int C, int M, int64_t ld_x, fqss_stream_t stream);
int fqss_bn_apply(const float* x, const float* a, const float* b, float* y, int B, int C, int M, int64_t ld_x,
                  int64_t ld_y, fqss_stream_t stream);
int fqss_bn_bwd_reduce(const float* g, const float* x, double* out, int B, int C, int M, int64_t ld_g, int64_t ld_x,
                       fqss_stream_t stream);
int fqss_bn_bwd_apply(const float* g, const float* x, const float* c1, const float* c2, const float* c3, float* gx,
                      int B, int C, int M, int64_t ld_g, int64_t ld_x, int64_t ld_gx, fqss_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K8/K9/K14  element-wise producers
 * replaces: torch.add / torch.sub / torch.mul in AddQ, ResidualErrorBlock, MulQ
 *           (qat_layers.py:69-71, 1193, 93-96), postprocess (process.py:44-47)
 * ------------------------------------------------------------------------------------------- */
/* z = sa*a + sb*b   (sa = 1, sb = +-1: add / sub, exact; sa = 2^-8, sb = 0: the combiner's scale) */
int fqss_axpby(const float* a, const float* b, float sa, float sb, float* z, int64_t rows,
               int64_t cols, int64_t ld_a, int64_t ld_b, int64_t ld_z, fqss_stream_t stream);
/* z[b][s][c][:] = mask[b][s][c][:] * feat[b][c][:] */
int fqss_mul_bcast_fwd(const float* mask, const float* feat, float* z, int B, int S, int C, int M,
                       int64_t ld_mask, int64_t ld_feat, int64_t ld_z, fqss_stream_t stream);
/* gmask = gz*feat ; gfeat = sum_s gz*mask */
int fqss_mul_bcast_bwd(const float* gz, const float* mask, const float* feat, float* gmask,
                       float* gfeat, int B, int S, int C, int M, int64_t ld_gz, int64_t ld_mask,
                       int64_t ld_feat, int64_t ld_gmask, int64_t ld_gfeat, fqss_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K10-K13  8-bit splitter, strided framing conv (encoder), transposed conv + overlap-add (decoder)
 * replaces: process.preprocess (process.py:16-37); F.conv1d(k, stride) of Conv1dEncoderQ and of
 *           ResidualErrorBlock (qat_layers.py:1028-1039, 1189-1192); F.conv_transpose1d of
 *           ConvTr1dDecoderQ / ResidualErrorBlock (:1330-1341, 1194-1202) + autograd
 * ------------------------------------------------------------------------------------------- */
/* x [B][T] -> out [B][2][T]; obs_ws must hold the global min/max of x (fqss_minmax) */
int fqss_splitter2(const float* x, float* out, int B, int64_t T, const uint32_t* obs_ws,
                   fqss_stream_t stream);
/* the same with normalize=False (process.py:26-27: threshold = max|x|, x is not divided by it; HTDemucs time branch) */
int fqss_splitter2_raw(const float* x, float* out, int B, int64_t T, const uint32_t* obs_ws, fqss_stream_t stream);
/* z[n][co][m] = sum_{ci,k} w[co][ci][k] * x[n][ci][m*stride+k]   (x: [N][Ci][T] dense)        */
int fqss_frames_conv_fwd(const float* x, const float* w, float* z, int N, int Ci, int Co,
                         int64_t T, int K, int stride, int M, int64_t ld_z, fqss_stream_t stream);
/* the same + add[n][co][m]: as the decoder's input gradient it sums in the gradient that the decoder input receives from its other
 * consumer (ResidualErrorBlock's Y - Y_q, qat_layers.py:1193), which autograd would add in a pass of its own */
int fqss_frames_conv_add_fwd(const float* x, const float* w, const float* add, int64_t ld_add, float* z, int N, int Ci,
                             int Co, int64_t T, int K, int stride, int M, int64_t ld_z, fqss_stream_t stream);
/* out[n][t] = sum_{c} sum_{m*stride+k=t} x[n][c][m] * w[c][k]   (w: [C][K], out: [N][T] dense) */
int fqss_ola_convtr_fwd(const float* x, const float* w, float* out, int N, int C, int M,
                        int64_t ld_x, int K, int stride, int64_t T, fqss_stream_t stream);
/* the same on an operand that arrives as the u8 codes of a per-tensor quantizer (de-quantised on load: the decoder input of the
 * student -- MulQ's output, the residual block's quantized error -- never exists in fp32; qat_layers.py:1305-1361, 1105-1202);
 * bit-identical to fqss_decode + fqss_ola_convtr_fwd.  Code rows 16-B aligned, (K, stride) in {(16, 8), (32, 16)} */
int fqss_ola_convtr_fwd_q(const uint8_t* xc, const float* qmin, const float* qmax, const float* w, float* out, int N,
                          int C, int M, int64_t ld_x, int K, int stride, int64_t T, fqss_stream_t stream);
/* the float model's masking product formed on load: x[n][c][m] = mask[n][c][m] * feat[n / NS][c][m] (convtasnetq.py:277-279 of the
 * float teacher: `masked = mask * feats` then the decoder); bit-identical to fqss_mul_bcast_fwd + fqss_ola_convtr_fwd */
int fqss_ola_convtr_mul_fwd(const float* mask, const float* feat, const float* w, float* out, int N, int NS, int C, int M,
                            int64_t ld_m, int64_t ld_f, int K, int stride, int64_t T, fqss_stream_t stream);
/* gw[c][ci][k] += sum_{n,m} a[n][c][m] * x[n][ci][m*stride+k]    (a: [N][C][M] ld_a; x dense) */
int fqss_frames_wgrad(const float* a, const float* x, float* gw, int N, int C, int Ci, int M,
                      int64_t ld_a, int64_t T, int K, int stride, fqss_stream_t stream);
/* the Ci = 1 case (residual encoder Conv1d(1, C, K, stride); the mono decoder's weight gradient with a = its input, sig = dL/dout)
 * as a dedicated fp32-MFMA kernel: gw[c][k] += sum_{n,m} a[n][c][m] * sig[n][m*stride + k]; (K, stride) in {(16, 8), (32, 16)}, rows
 * of a 16-B aligned (FQSS_EINVAL otherwise: callers fall back to fqss_frames_wgrad).  _q: a as u8 codes (qmin, qmax). */
int fqss_frames_wgrad1(const float* a, const float* sig, float* gw, int N, int C, int M, int64_t ld_a, int64_t T, int K,
                       int stride, fqss_stream_t stream);
/* one input channel of a multi-channel framing conv: sig = x + ci * T (signals sig_ns = Ci * T apart), gw = gw0 + ci * K (rows ld_gw = Ci * K apart) */
int fqss_frames_wgrad1s(const float* a, const float* sig, int64_t sig_ns, float* gw, int64_t ld_gw, int N, int C, int M,
                        int64_t ld_a, int64_t T, int K, int stride, fqss_stream_t stream);
int fqss_frames_wgrad1_q(const uint8_t* ac, const float* qmin, const float* qmax, const float* sig, float* gw, int N, int C,
                         int M, int64_t ld_a, int64_t T, int K, int stride, fqss_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K15  SDR-weighted KD loss with 2-speaker PIT, forward + backward in one call
 * replaces: System.common_step (mysystem.py:124-151), PairwiseWSDR (wsdr.py:56-95),
 *           asteroid PITLossWrapper(pit_from="pw_mtx") for n_src = 2
 * est/fest/tgt: [B][2][T] dense.  stats: fp64 scratch [B][32] (callee zeroes).
 * out[0]=loss out[1]=kd(dB, logged as kd_loss) out[2]=task out[3]=kd (linear); w_out[B]; gest = dL/dest
 * ------------------------------------------------------------------------------------------- */
int fqss_kd_loss(const float* est, const float* fest, const float* tgt, int B, int64_t T,
                 float kd_lambda, double* stats, float* out, float* w_out, float* sisdr_out,
                 float* gest, fqss_stream_t stream);

/* the speechbrain env's form of the objective (speechbrain_librimix_trainer.py:99-115, 141-149; wsdr.py:60-116 of that env): the log
 * is taken PER SAMPLE, out[0] = mean of loss_b over the samples with loss_b > threshold (all samples when use_threshold = 0 or none is
 * above).  KD weights as the reference broadcasts them ([1, n_src, n_src] * [1, B]): B = 1 the sample's w; B = 2 student source j
 * carries w[j] in both samples; B > 2: FQSS_EINVAL (the reference raises).  Everything else as fqss_kd_loss. */
int fqss_kd_loss_per_sample(const float* est, const float* fest, const float* tgt, int B, int64_t T,
                            float kd_lambda, int use_threshold, float threshold, double* stats, float* out,
                            float* w_out, float* sisdr_out, float* gest, fqss_stream_t stream);
/* the teacher-free loss of kd_lambda = 0 (mysystem.py:153-156: PITLossWrapper(pairwise_neg_sisdr, pit_from="pw_mtx") for n_src = 2):
 * out[0] = mean_b min_perm mean_src -10 log10(si_sdr(est_p(i), tgt_i) + eps), out[2] = the mean task ratio (logging), gest = dL/dest;
 * sisdr_out[b] = the sample's best-permutation SI-SDR in dB; buffers as fqss_kd_loss */
int fqss_pit_sisdr_loss(const float* est, const float* tgt, int B, int64_t T, double* stats, float* out, float* w_out,
                        float* sisdr_out, float* gest, fqss_stream_t stream);
/* the streaming pass of fqss_kd_loss alone: stats[b][0..23] = the 24 fp64 second-order moments of sample b (sums of e0 e1 f0 f1 t0 t1,
 * their self products, e_i.t_j, e_i.f_j, f_i.t_j; row stride 32) -- what the evaluation forms of SDR / PairwiseWSDR
 * (wsdr.py:10-95) are computed from */
int fqss_kd_moments(const float* est, const float* fest, const float* tgt, int B, int64_t T, double* stats,
                    fqss_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K16  global-norm clip + Adam over one flat fp32 parameter buffer
 * replaces: pl.Trainer(gradient_clip_val=5.0) + torch.optim.Adam (asteroid_librimix_trainer.py:94,132)
 * ------------------------------------------------------------------------------------------- */
/* FQSS_DETERMINISTIC=1 (round 5): bit-reproducible gradients.  fqss_set_deterministic(slot, grad_base, n, shadow) switches every fp32
 * gradient atomic whose target lies in the n-float arena at grad_base (bias / depthwise-weight row sums, the frame-path weight
 * gradients' split adds) to 64-bit INTEGER atomics on a two-word fixed-point shadow of that arena (shadow: 16 * n bytes, zeroed by
 * the caller, 16-B aligned) -- exact and commutative, so the sums no longer depend on the order workgroups retire in;
 * fqss_det_finish (after the backward, before clip + Adam) rounds the sums once into the arena and clears the shadow.
 * Three arenas can be covered: slot 0 (the parameter gradients; must be set for the mode to be on), slot 1 (the dL/dW_q arena) and
 * slot 2 (a pool the host cuts temporary accumulators from; fqss_det_finish accepts any sub-range: base + off, n, shadow + 16 * off).
 * shadow = NULL clears a slot.  Synchronises the device; call outside graph capture.
 * Not a reference interface: the mode exists so that the convergence gates can separate arithmetic differences from run-to-run
 * noise (tests/test_gpu_converge.py; mysystem.py:124-151 is the step whose gradients these are). */
int fqss_set_deterministic(int slot, const float* grad_base, int64_t n, void* shadow);
int fqss_det_finish(float* grad_base, int64_t n, void* shadow, fqss_stream_t stream);
/* sumsq[0] += sum g^2  (fp64) */
int fqss_sumsq(const float* g, int64_t n, double* sumsq, fqss_stream_t stream);
/* step_t: device int32 global step counter (incremented by the call -> graph-replay safe).
 * t0 (nullable): per-element int32, number of global steps that passed before the element's
 * parameter first received a gradient (torch.optim.Adam counts steps per parameter); INT32_MAX
 * marks parameters that never had one (skipped, like torch skips grad=None).
 * g is scaled by min(1, max_norm/(sqrt(sumsq)+1e-6)) (torch.nn.utils.clip_grad_norm_), grad_scale
 * pre-multiplies g (1/world for DDP averaging).                                                  */
int fqss_adam_clip(float* p, const float* g, float* m, float* v, int64_t n, const double* sumsq,
                   float max_norm, float grad_scale, float lr, float beta1, float beta2, float eps,
                   int32_t* step_t, const int32_t* t0, float* gnorm_out, fqss_stream_t stream);

/* =============================================================================================
 * Dual-path models (DPTNet, cfg 3 -- SURVEY.md §8 row a13).  Inside the dual-path blocks tensors are
 * sequence-first row matrices [L][B'][C] (feature dim contiguous).
 * ============================================================================================= */

/* Row-major linears (csrc/gemm.hip, fp32 MFMA): x [R][Ci] (row stride ld_x), w [Co][Ci] (row stride ld_w).
 * replaces: F.linear of LinearQ (qat_layers.py:521-536), of MultiheadAttentionQ's in/out projections (:889-901, :941),
 *           the input projection inside _VF.lstm (LSTMQ :590), the 1x1 Conv2dQ of DPT.output (dptnetq.py:187) + autograd
 *   fwd   : z[r][o]   = sum_i x[r][i] w[o][i] + bias[o]   (bias optional)
 *   bwd_x : gx[r][i]  = sum_o gz[r][o] w[o][i]
 *   bwd_w : gw[o][i] += sum_r gz[r][o] x[r][i]                                                     */
int fqss_rowlin_fwd(const float* x, const float* w, const float* bias, float* z, int64_t R, int Ci, int Co,
                    int64_t ld_x, int64_t ld_w, int64_t ld_z, fqss_stream_t stream);
/* fqss_rowlin_fwd with the weight given as its three exact bf16 planes [3][Co][Ci] (fqss_split3_planes): for weights that do not
 * change between launches (the frozen float teacher, mysystem.py:132-133) the weight tile is copied into LDS instead of being split
 * by every workgroup.  Same result bits.  Ci % 32 == 0, 16-B aligned activation rows. */
int fqss_rowlin_fwd_w3(const float* x, const uint16_t* w3, const float* bias, float* z, int64_t R, int Ci, int Co, int64_t ld_x,
                       int64_t ld_z, fqss_stream_t stream);
int fqss_rowlin_bwd_x(const float* gz, const float* w, float* gx, int64_t R, int Ci, int Co, int64_t ld_gz,
                      int64_t ld_w, int64_t ld_gx, fqss_stream_t stream);
int fqss_rowlin_bwd_w(const float* gz, const float* x, float* gw, int64_t R, int Ci, int Co, int64_t ld_gz,
                      int64_t ld_x, int64_t ld_gw, fqss_stream_t stream);
/* `batch` weight gradients of one shape in ONE launch: problem p reads gz + p*sb_gz, x + p*sb_x and adds into gw + p*sb_gw (element
 * strides of either sign) -- the W_hh gradients of the two directions of a bidirectional LSTM (torch's LSTM backward,
 * qat_layers.py:571-600): the same dG / h tensors, shifted by one step and one column block */
int fqss_rowlin_bwd_w_batched(const float* gz, const float* x, float* gw, int64_t R, int Ci, int Co, int64_t ld_gz, int64_t ld_x,
                              int64_t ld_gw, int batch, int64_t sb_gz, int64_t sb_x, int64_t sb_gw, fqss_stream_t stream);
/* out[c] += sum_r g[r][c]  (bias gradients of the row linears / LSTM) */
int fqss_colsum(const float* g, float* out, int64_t R, int C, int64_t ld, fqss_stream_t stream);

/* LayerNorm over the last dim of a row matrix (C <= 256), one wavefront per row.
 * replaces: F.layer_norm in LayerNormQ (qat_layers.py:455-465) + autograd.  mean_rstd: [R][2] saved for the backward.
 * bwd: gx = ; ggamma[C] += ; gbeta[C] +=                                                            */
int fqss_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean_rstd,
                       int64_t R, int C, int64_t ld_x, int64_t ld_y, double eps, fqss_stream_t stream);
int fqss_layernorm_bwd(const float* gy, const float* x, const float* gamma, const float* mean_rstd, float* gx,
                       float* ggamma, float* gbeta, int64_t R, int C, int64_t ld_gy, int64_t ld_x, int64_t ld_gx,
                       fqss_stream_t stream);
/* LayerNormQ = LayerNorm + its output quantizer in ONE pass each way (qat_layers.py:455-465, qat_quant.py:136-147): forward writes
 * y = fq(LN(x)) and (yc nullable) its u8 codes, the pre-quant value is not stored; backward takes g = dL/dy, recomputes the pre-quant
 * value from x / mean_rstd, applies the STE, adds the range-gradient partials to gacc (FQSS_GACC_SLOTS x 3, as fqss_actq_bwd) and
 * runs the LayerNorm backward -- replaces fqss_layernorm_fwd + fqss_actq_fwd and fqss_actq_bwd + fqss_layernorm_bwd */
int fqss_layernormq_fwd(const float* x, const float* gamma, const float* beta, float* y, uint8_t* yc, float* mean_rstd,
                        int64_t R, int C, int64_t ld_x, int64_t ld_y, int64_t ld_yc, double eps, const float* qmin,
                        const float* qmax, fqss_stream_t stream);
int fqss_layernormq_bwd(const float* g, const float* x, const float* gamma, const float* beta, const float* mean_rstd,
                        float* gx, float* ggamma, float* gbeta, int64_t R, int C, int64_t ld_g, int64_t ld_x,
                        int64_t ld_gx, const float* qmin, const float* qmax, double* gacc, fqss_stream_t stream);

/* The residual add in front of a pre-norm transformer sub-layer fused into its LayerNorm / LayerNormQ (sepformerq.py:69-82, the float
 * `+` of `x = x + sublayer(norm(x))` followed by the next norm): s = a + b is written once (the residual stream), y = LN(s) or, with
 * qmin / qmax, fq(LN(s)) (+ codes yc, nullable).  bwd: g = dL/dy, gs = dL/ds arriving over the residual stream (nullable); gx = the
 * LayerNorm(Q) backward at s, + gs: the gradient of a AND of b (the sum autograd takes at the fork); replaces fqss_axpby + fqss_layernorm[q]_fwd
 * and fqss_layernorm[q]_bwd + fqss_axpby */
int fqss_add_layernorm_fwd(const float* a, const float* b, const float* gamma, const float* beta, float* s, float* y, uint8_t* yc,
                           float* mean_rstd, int64_t R, int C, int64_t ld_a, int64_t ld_b, int64_t ld_s, int64_t ld_y, int64_t ld_yc,
                           double eps, const float* qmin, const float* qmax, fqss_stream_t stream);
int fqss_add_layernorm_bwd(const float* g, const float* gs, const float* s, const float* gamma, const float* beta,
                           const float* mean_rstd, float* gx, float* ggamma, float* gbeta, int64_t R, int C, int64_t ld_g,
                           int64_t ld_gs, int64_t ld_s, int64_t ld_gx, const float* qmin, const float* qmax, double* gacc,
                           fqss_stream_t stream);
/* the same pair for a QUANTIZED add in front of the norm: y = LN(Q)(fq_s(a + b)) -- AddQ followed by LayerNormQ in the post-norm
 * layers of DPTNet (dptnetq.py:84-97).  z receives the PRE-quant sum a + b (the backward's input); qs_min / qs_max: the AddQ's range.
 * The backward's gx (the gradient of a AND of b) has passed the AddQ's STE; its range partials go to gacc_s. */
int fqss_addq_layernorm_fwd(const float* a, const float* b, const float* gamma, const float* beta, float* z, float* y, uint8_t* yc,
                            float* mean_rstd, int64_t R, int C, int64_t ld_a, int64_t ld_b, int64_t ld_z, int64_t ld_y,
                            int64_t ld_yc, double eps, const float* qmin, const float* qmax, const float* qs_min,
                            const float* qs_max, fqss_stream_t stream);
int fqss_addq_layernorm_bwd(const float* g, const float* gs, const float* z, const float* gamma, const float* beta,
                            const float* mean_rstd, float* gx, float* ggamma, float* gbeta, int64_t R, int C, int64_t ld_g,
                            int64_t ld_gs, int64_t ld_z, int64_t ld_gx, const float* qmin, const float* qmax, double* gacc,
                            const float* qs_min, const float* qs_max, double* gacc_s, fqss_stream_t stream);
/* the same pair with the dual-path layout change folded in (dptnetq.py:313-327: `permute().contiguous()` between the intra- and the
 * inter-chunk transformer): y / yc row (i0*d1 + i1)*d2 + i2 is written at dense row i0*t0 + i1*t1 + i2*t2, the backward reads dL/dy
 * of row r from there (g dense, C floats per row); the forward also takes qs_min = qs_max = NULL: a plain add (the float teacher) */
int fqss_addq_layernorm_fwd_map(const float* a, const float* b, const float* gamma, const float* beta, float* z, float* y, uint8_t* yc,
                                float* mean_rstd, int64_t R, int C, int64_t ld_a, int64_t ld_b, int64_t ld_z, double eps, const float* qmin,
                                const float* qmax, const float* qs_min, const float* qs_max, int64_t d1, int64_t d2, int64_t t0, int64_t t1,
                                int64_t t2, fqss_stream_t stream);
int fqss_addq_layernorm_bwd_map(const float* g, const float* z, const float* gamma, const float* beta, const float* mean_rstd, float* gx,
                                float* ggamma, float* gbeta, int64_t R, int C, int64_t ld_z, int64_t ld_gx, const float* qmin,
                                const float* qmax, double* gacc, const float* qs_min, const float* qs_max, double* gacc_s, int64_t d1,
                                int64_t d2, int64_t t0, int64_t t1, int64_t t2, fqss_stream_t stream);

/* element-wise maps on dense tensors; kind: 0 tanh, 1 sigmoid, 2 division by the scalar p
 * replaces: nn.Tanh / nn.Sigmoid inside Conv1dNlQ (dptnetq.py:286-287), q / sqrt(head_dim) (qat_layers.py:905).
 * bwd takes the forward OUTPUT y (tanh, sigmoid); y may be NULL for kind 2.                          */
#define FQSS_UNARY_TANH 0
#define FQSS_UNARY_SIGMOID 1
#define FQSS_UNARY_DIVS 2
int fqss_unary_fwd(const float* x, float* y, int64_t n, int kind, double p, fqss_stream_t stream);
/* the same maps on a row-strided input x [rows][cols] (row stride ld_x: a column block of a wider matrix, read in place), y [rows][cols]
 * with row stride ld_y; cols, ld_x, ld_y multiples of 4, 16-B aligned rows.  replaces: `q = q / math.sqrt(head_dim)` on the q third of the
 * in-projection in the float MultiheadAttention (qat_layers.py:889-901 with the quantizers off: the teacher) without the slice copy */
int fqss_unary_rows_fwd(const float* x, float* y, int64_t rows, int cols, int64_t ld_x, int64_t ld_y, int kind, double p,
                        fqss_stream_t stream);
/* the value maps of the public STE helpers round_ste / floor_ste / grad_sign / clip_ste (qat_quant.py:88-107): torch.round (half
 * to even), floor, sign, clip(x, p, p2); their backward is the identity times a scale (no kernel of its own)                     */
#define FQSS_UNARY_ROUND 4
#define FQSS_UNARY_FLOOR 5
#define FQSS_UNARY_SIGN 6
#define FQSS_UNARY_CLIP 7
int fqss_unary2_fwd(const float* x, float* y, int64_t n, int kind, double p, double p2, fqss_stream_t stream);
int fqss_unary_bwd(const float* g, const float* y, float* gx, int64_t n, int kind, double p, fqss_stream_t stream);

/* y[i0][i1][i2][c] = x[i0*s0 + i1*s1 + i2*s2 + c], c < C contiguous: the intra-chunk <-> inter-chunk change of view
 * replaces: the permute().contiguous() pairs of DPT.forward / SingleTransformer.forward (dptnetq.py:156, 197-204) */
int fqss_permute4(const float* x, float* y, int64_t n0, int64_t n1, int64_t n2, int C, int64_t s0, int64_t s1,
                  int64_t s2, fqss_stream_t stream);
/* ... into row-padded output: y[((i0 n1 + i1) n2 + i2) ld_y + c], ld_y >= C a multiple of 4, y 16-B aligned; the padding columns are
 * zero-filled.  (The [B, F, C, T] image HTDemucs' frequency-branch DConv works on, hdemucsq.py:126-143 / demucsq.py:168-182, T = 431: with
 * padded rows every element-wise kernel behind the move and every gradient coming back runs its 16-B form.) */
int fqss_permute4_ld(const float* x, float* y, int64_t n0, int64_t n1, int64_t n2, int C, int64_t s0, int64_t s1,
                     int64_t s2, int64_t ld_y, fqss_stream_t stream);

/* split_feature (dptnetq.py:232-259) straight into the intra-chunk row layout:
 *   f [B][N][T] (row stride ld_f) -> seg [K][B*S][N], S = 2*(T + rest + K/2)/K half-overlapped chunks of length K
 * bwd: gf [B][N][T] = sum of the (<= 2) chunk slots holding each position                            */
int fqss_dp_segment_fwd(const float* f, float* seg, int B, int N, int64_t T, int64_t ld_f, int K, int S,
                        fqss_stream_t stream);
int fqss_dp_segment_bwd(const float* gseg, float* gf, int B, int N, int64_t T, int64_t ld_gf, int K, int S,
                        fqss_stream_t stream);
/* merge_feature (dptnetq.py:261-276) up to its AddQ: o [S][B*K][nspk*N] (inter-chunk row layout) -> the two streams
 * a, b [B*nspk][N][Lm], Lm = (S/2)*K - K/2, whose quantized sum is the merged signal; bwd scatters (ga, gb) back */
int fqss_dp_merge_fwd(const float* o, float* a, float* b, int B, int nspk, int N, int K, int S, int64_t Lm,
                      int64_t ld_ab, fqss_stream_t stream);
int fqss_dp_merge_bwd(const float* ga, const float* gb, float* go, int B, int nspk, int N, int K, int S, int64_t Lm,
                      int64_t ld_ga, int64_t ld_gb, fqss_stream_t stream);
/* overlap_and_add of 2-sample frames with hop 1 (dptnetq.py:17-58 as called at :140 with W = 2):
 *   y [N][2][L] (rows of stride ld_y) -> out [N][L+1];   bwd: gy[n][tap][t] = g[n][t + tap]           */
int fqss_ola2_fwd(const float* y, float* out, int64_t N, int64_t L, int64_t ld_y, fqss_stream_t stream);
int fqss_ola2_bwd(const float* g, float* gy, int64_t N, int64_t L, int64_t ld_gy, fqss_stream_t stream);

/* Attention core of MultiheadAttentionQ (qat_layers.py:903-911): o = softmax(q k^T) v per (sequence b, head h).
 * q, k, v, o: row matrices [l*B + b][nh*hd] (head h = column block h); q is the already scaled + quantized query.
 * stats [B*nh][L][2] = (row max, row sum) saved for the backward, which recomputes the probabilities.
 * obs_attn / obs_soft (both or neither): ordered-uint (min, max) of the logits / probabilities -- the reference's
 * `activation_fake_quantize_attn/_softmax` observe them during the first 50 calls and discard their outputs. */
int fqss_attn_fwd(const float* q, const float* k, const float* v, float* o, float* stats, int L, int B, int nh,
                  int hd, int64_t ld_q, int64_t ld_k, int64_t ld_v, int64_t ld_o, uint32_t* obs_attn,
                  uint32_t* obs_soft, fqss_stream_t stream);
int fqss_attn_bwd(const float* q, const float* k, const float* v, const float* o, const float* go,
                  const float* stats, float* gq, float* gk, float* gv, int L, int B, int nh, int hd, int64_t ld_q,
                  int64_t ld_k, int64_t ld_v, int64_t ld_o, int64_t ld_go, int64_t ld_gq, int64_t ld_gk,
                  int64_t ld_gv, fqss_stream_t stream);

/* The prologue of that core in the quantizing phase as one pass each way (qat_layers.py:890-905): X [R][3E] (the in-projection, row stride
 * ld_x) -> q = fq_div(fq_q(X[:, :E]) / scale), k = fq_k(X[:, E:2E]), v = fq_v(X[:, 2E:]), each dense [R][E]; bit-identical to three
 * fqss_actq_fwd on the thirds + fqss_unary_fwd(DIVS) + fqss_actq_fwd.  ranges: host array of 8 device pointers (min, max of the q, k,
 * v, div quantizers).  bwd: (gq, gk, gv) -> gX [R][3E] through the STEs and the division, the intermediate values recomputed from X;
 * gaccs: host array of 4 device pointers, range-gradient partials of q, k, v, div (FQSS_GACC_SLOTS x 3 each, "+=" as fqss_actq_bwd) */
int fqss_mha_prep_fwd(const float* X, float* q, float* k, float* v, int64_t R, int E, int64_t ld_x, double scale,
                      const float* const* ranges, fqss_stream_t stream);
int fqss_mha_prep_bwd(const float* X, const float* gq, const float* gk, const float* gv, float* gX, int64_t R, int E,
                      int64_t ld_x, int64_t ld_gx, double scale, const float* const* ranges, double* const* gaccs,
                      fqss_stream_t stream);
/* fqss_mha_prep_fwd emitting the 8-bit CODES of q (on the division quantizer's grid), k, v ([R][E] u8 each) instead of the
 * de-quantized values: the operands of the coded attention kernels below */
int fqss_mha_prep_fwd_c(const float* X, uint8_t* qc, uint8_t* kc, uint8_t* vc, int64_t R, int E, int64_t ld_x, double scale,
                        const float* const* ranges, fqss_stream_t stream);
/* softmax(q k^T) v of MultiheadAttentionQ's core (qat_layers.py:878-911) in its quantizing phase, from the CODES of q, k, v
 * (x = delta c + min of each quantizer; ranges = device scalars {q_min, q_max, k_min, k_max, v_min, v_max}): an integer code is one
 * exact bf16 plane and terms constant along the softmax axis drop out, so the products need 3 (or 1) MFMAs where the float form
 * needs 6 (csrc/attn_long.hip).  Same row addressing as fqss_attn_long_fwd / _bwd (strides in elements of each tensor: bytes for
 * the codes); head_dim 16 / 32 / 64, code rows 8-B aligned.  stats / dsum: as fqss_attn_long_*; the statistics refer to the
 * logits without their per-query constant.  gq / gk / gv are gradients with respect to the de-quantized values. */
int fqss_attn_long_fwd_c(const uint8_t* qc, const uint8_t* kc, const uint8_t* vc, const float* const* ranges, float* o, float* stats,
                         int Lq, int Lk, int B, int nh, int hd, const int64_t* strides, fqss_stream_t stream);
int fqss_attn_long_bwd_c(const uint8_t* qc, const uint8_t* kc, const uint8_t* vc, const float* const* ranges, const float* o,
                         const float* go, const float* stats, float* gq, float* gk, float* gv, float* dsum, int Lq, int Lk, int B,
                         int nh, int hd, const int64_t* strides, fqss_stream_t stream);

/* Recurrence of the bidirectional single-layer LSTM inside LSTMQ (qat_layers.py:571-600, _VF.lstm with zero state).
 *   pre  [S][B][2][4H] = x W_ih^T + b_ih of both directions (fqss_rowlin_fwd), gate order i, f, g, o
 *   whh  [2][4H][H], bhh [2][4H];  hout [S][B][2H] (forward | reverse)
 *   gsav [S][B][2][4H] gate activations, csav [S][B][2][2][H] cell states c | tanh(c) (saved for the backward; both NULL: inference, nothing saved)
 * bwd: gout [S][B][2H] -> dG [S][B][2][4H] = dL/d(gate pre-activations); the caller finishes with row GEMMs:
 *   gx = dG W_ih, gW_ih += dG^T x, gb_ih = gb_hh += colsum(dG), gW_hh[d] += dG_d[t]^T h_d[t -/+ 1]           */
int fqss_lstm_fwd(const float* pre, const float* whh, const float* bhh, float* hout, float* gsav, float* csav,
                  int S, int B, int H, fqss_stream_t stream);
int fqss_lstm_bwd(const float* gout, const float* whh, const float* gsav, const float* csav, float* dG, int S,
                  int B, int H, fqss_stream_t stream);
/* Test hook: the H = 128 forward's own gate functions (short dependent chains on v_exp_f32 / v_rcp_f32 in place of libm's
 * expf / tanhf and an IEEE division; <= 4 ulp from the exact value) over n arguments: sigmoid_out[i], tanh_out[i] of x[i] */
int fqss_lstm_gate_fn(const float* x, float* sigmoid_out, float* tanh_out, int64_t n, fqss_stream_t stream);
/* fqss_lstm_bwd that also ADDS the column sums of dG over (step, sequence) into gbias [2][4H] (caller-zeroed): the gradient of
 * b_ih and b_hh of each direction (torch's LSTM backward, reached from qat_layers.py:571-600), kept in registers by the
 * threads that produce dG */
int fqss_lstm_bwd_b(const float* gout, const float* whh, const float* gsav, const float* csav, float* dG, float* gbias, int S,
                    int B, int H, fqss_stream_t stream);
/* ... with the sums added straight into the four bias parameters' own gradient buffers: gb[0..3] = b_ih, b_hh of the forward
 * direction, b_ih, b_hh of the reverse direction ([4H] each, "+=") */
int fqss_lstm_bwd_b4(const float* gout, const float* whh, const float* gsav, const float* csav, float* dG, float* const* gb, int S,
                     int B, int H, fqss_stream_t stream);

/* Sepformer (cfg 4 -- SURVEY.md §8 row a14): gLN and the positional-encoding add on the dual-path row layouts.
 * GroupNorm(1, C) over ALL rows of a sample of a row matrix x [R][C]; the sample of row r is b = (r % RB) / X
 * (intra-chunk rows [K][B*S][C]: RB = B*S, X = S; inter-chunk rows [S][B*K][C]: RB = B*K, X = K), B <= 16.
 * replaces: nn.GroupNorm(1, F) of DualPathBlock on [B, F, K, S] (sepformerq.py:141-142, 159, 175) + autograd, without the
 * permute().contiguous() round trips around it.  ws: 2*B doubles of scratch; mean_rstd [B][2] saved for the backward.
 * bwd: gx = ; ggamma[C] += ; gbeta[C] +=                                                            */
int fqss_gnrows_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean_rstd, double* ws,
                    int64_t R, int C, int64_t ld_x, int64_t ld_y, int RB, int X, int B, double eps, fqss_stream_t stream);
int fqss_gnrows_bwd(const float* gy, const float* x, const float* gamma, const float* mean_rstd, float* gx,
                    float* ggamma, float* gbeta, double* ws, int64_t R, int C, int64_t ld_gy, int64_t ld_x,
                    int64_t ld_gx, int RB, int X, int B, fqss_stream_t stream);
/* z[l][b][c] = x[l][b][c] + p[l][c]  (TransformerBlock.pos_add, sepformerq.py:117-118, sequence-first rows);
 * out[l][c] = sum_b g[l][b][c]       (the gradient reaching the ConstQ-quantized positional encoding)        */
int fqss_bcast_add(const float* x, const float* p, float* z, int64_t L, int64_t Bp, int C, fqss_stream_t stream);
int fqss_bcast_sum(const float* g, float* out, int64_t L, int64_t Bp, int C, fqss_stream_t stream);

/* Row-major linear of the dual-path STUDENT on codes (csrc/qrow.hip, int8 MFMA, exact integer sums):
 *   z[r][o] = dw[o] * (dx * sum_i wk[o][i] xc[r][i] + min_x * rw[o]) + bias[o]
 * xc: u8 activation codes [R][Ci] (rows 16-B aligned) with the ranges (qmin_x, qmax_x) of the quantizer that produced them;
 * wk / dw / rw: the weight's int8 codes, per-row step and code sum from fqss_wq_codes.  Ci a multiple of 16.
 * replaces: F.linear on fake-quantized operands (qat_layers.py:521-536, 889-901, 941) in the quantizing phase.            */
int fqss_qrow_fwd(const uint8_t* xc, const int8_t* wk, const float* dw, const float* rw, const float* bias,
                  const float* qmin_x, const float* qmax_x, float* z, int64_t R, int Ci, int Co, int64_t ld_x,
                  int64_t ld_z, fqss_stream_t stream);
/* fqss_qrow_fwd with the layer's OUTPUT quantizer in the epilogue: z (kept for the backward) and y = fq(act(z)) -- what
 * fqss_actq_fwd(z, QUANT) computes -- from one launch.  LinearQ / LinearNlQ / the attention output projection in the quantizing
 * phase (qat_layers.py:521-561, 941-950). */
int fqss_qrow_fwdq(const uint8_t* xc, const int8_t* wk, const float* dw, const float* rw, const float* bias,
                   const float* qmin_x, const float* qmax_x, float* z, float* y, int64_t R, int Ci, int Co, int64_t ld_x,
                   int64_t ld_z, int64_t ld_y, int act, const float* slope, const float* qmin_y, const float* qmax_y,
                   fqss_stream_t stream);
/* LinearQ -> NlQ(ReLU) of a feed-forward block (sepformerq.py:64 `ffn`, qat_layers.py:521-536 and 409-432) from ONE launch:
 * z (kept for the backward), y = fq2(relu(fq1(z))) and its u8 codes yc -- operation for operation what fqss_qrow_fwdq followed by
 * fqss_actq_fwd(act = ReLU) compute; the fp32 image of fq1(z) is never stored.  q1 = the linear's own output quantizer, q2 = NlQ's */
int fqss_qrow_fwdq2(const uint8_t* xc, const int8_t* wk, const float* dw, const float* rw, const float* bias, const float* qmin_x,
                    const float* qmax_x, float* z, float* y, uint8_t* yc, int64_t R, int Ci, int Co, int64_t ld_x, int64_t ld_z, int64_t ld_y,
                    int64_t ld_yc, const float* qmin1, const float* qmax1, const float* qmin2, const float* qmax2, fqss_stream_t stream);
/* the gradient GEMMs of such a linear on the same codes (csrc/gemm_x3.hip, coded-B forms): the 8-bit operand is one exact bf16 plane,
 * three MFMA products per k instead of the six of the fp32 x fp32 form.
 *   fqss_qrow_bwd_x: gx[r][i]  = sum_o gz[r][o] * (dw[o] * wi[o][i])       wi int8 [Co][Ci] dense
 *   fqss_qrow_bwd_w: gw[o][i] += sum_r gz[r][o] * (dx * c[r][i] + min_x)   c u8 [R][ld_xc], (dx, min_x) from the range scalars
 * Ci, Co and the row strides are multiples of 4, operands 16-B (codes 4-B) aligned */
int fqss_qrow_bwd_x(const float* gz, const int8_t* wi, const float* dw, float* gx, int64_t R, int Ci, int Co,
                    int64_t ld_gz, int64_t ld_gx, fqss_stream_t stream);
int fqss_qrow_bwd_w(const float* gz, const uint8_t* xc, const float* qmin_x, const float* qmax_x, float* gw,
                    int64_t R, int Ci, int Co, int64_t ld_gz, int64_t ld_xc, int64_t ld_gw, fqss_stream_t stream);
/* fqss_qrow_bwd_w that also adds the bias gradient gbias[o] += sum_r gz[r][o] (the kernel keeps these sums for the min_x term of the coded
 * product anyway): replaces the `fqss_colsum` pass behind F.linear's autograd (qat_layers.py:521-536) */
int fqss_qrow_bwd_wb(const float* gz, const uint8_t* xc, const float* qmin_x, const float* qmax_x, float* gw, float* gbias, int64_t R, int Ci,
                     int Co, int64_t ld_gz, int64_t ld_xc, int64_t ld_gw, fqss_stream_t stream);

/* The coded weight gradients of SEVERAL row-major linears in one launch per <= 32 jobs (csrc/gemm_x3.hip k_gemm_x3_wq_multi; round 5):
 * per job exactly fqss_qrow_bwd_wb (gbias nullable: fqss_qrow_bwd_w) -- gw[o][i] += sum_r gz[r][o] (dx c[r][i] + min_x), gbias[o] +=
 * sum_r gz[r][o].  The weight gradients of LinearQ / LinearNlQ / the attention projections (autograd of F.linear, qat_layers.py:
 * 521-568, 865-950) feed nothing but the optimizer, so the host may queue them over a backward segment; one launch of thousands of
 * workgroups overlaps the latency-bound k-tile chains that a single 256-workgroup launch serialises.  `jobs` is a HOST array. */
typedef struct FqssRowWgradJob {
    const float* gz; const uint8_t* xc; const float* qmin_x; const float* qmax_x;
    float* gw; float* gbias;
    int64_t R; int32_t Ci, Co;
    int64_t ld_gz, ld_xc, ld_gw;
} FqssRowWgradJob;
int fqss_qrow_bwd_w_group(const FqssRowWgradJob* jobs, int njobs, fqss_stream_t stream);
/* `batch` coded weight gradients of one shape and one input range in ONE launch: problem p reads gz + p*sb_gz (floats), xc + p*sb_xc
 * (bytes) and adds into gw + p*sb_gw (floats) -- the W_ih gradients of the two directions of LSTMQ (qat_layers.py:571-600): two column
 * blocks of dG against the same input codes */
int fqss_qrow_bwd_w_batched(const float* gz, const uint8_t* xc, const float* qmin_x, const float* qmax_x, float* gw, int64_t R, int Ci,
                            int Co, int64_t ld_gz, int64_t ld_xc, int64_t ld_gw, int batch, int64_t sb_gz, int64_t sb_xc, int64_t sb_gw,
                            fqss_stream_t stream);

/* First layer kernels of cfg 5 (HTDemucs; the model itself is not built yet).
 * GELU is kind FQSS_UNARY_GELU of fqss_unary_fwd/bwd (erf form; the backward takes the INPUT x in place of y).
 * GLU over the channel dim of a channel-first tensor (nn.GLU(dim=1), hdemucsq.py:127,314; demucsq.py:168):
 *   x [B][2C][M] -> y [B][C][M] = x[:, :C] * sigmoid(x[:, C:]);  bwd: gx [B][2C][M] =
 * element-wise torch.div (DivQ) with its two gradients; F.embedding gather / scatter-add (EmbeddingQ, ScaledEmbedding).   */
#define FQSS_UNARY_GELU 3
int fqss_glu_fwd(const float* x, float* y, int64_t B, int64_t C, int64_t M, int64_t ld_x, int64_t ld_y, fqss_stream_t stream);
int fqss_glu_bwd(const float* x, const float* gy, float* gx, int64_t B, int64_t C, int64_t M, int64_t ld_x, int64_t ld_gy,
                 int64_t ld_gx, fqss_stream_t stream);
int fqss_div_fwd(const float* a, const float* b, float* y, int64_t n, fqss_stream_t stream);
int fqss_div_bwd(const float* g, const float* a, const float* b, float* ga, float* gb, int64_t n, fqss_stream_t stream);
int fqss_embedding_fwd(const float* w, const int64_t* idx, float* out, int64_t n, int D, int64_t V, fqss_stream_t stream);
int fqss_embedding_bwd(const float* g, const int64_t* idx, float* gw, int64_t n, int D, int64_t V, fqss_stream_t stream);

/* General convolution geometry of the HTDemucs layers: frames of a [B][C][H][W] signal (strides sb, sc, sh in
 * elements, unit stride along W) for a kernel (kh, kw), stride (st_h, st_w), zero padding (ph, pw), dilation (dh, dw):
 *   frames[b][(c*kh + i)*kw + j][ho*Wo + wo] = x[b][c][ho*st_h - ph + i*dh][wo*st_w - pw + j*dw]   (0 outside), row stride ld.
 * fqss_frames_gather followed by a pointwise GEMM over C*kh*kw channels IS nn.Conv1d / nn.Conv2d with groups = 1
 * (replaces F.conv1d / F.conv2d in Conv1dQ / Conv1dNlQ / Conv2dNlQ / Conv1dGnNlQ / Conv*EncoderQ, qat_layers.py:124-259, 262-300);
 * fqss_frames_ola is its adjoint (deterministic gather-form overlap-add, optional per-channel bias): a pointwise GEMM followed
 * by it IS nn.ConvTranspose1d / 2d (ConvTranspose*NlQ, ConvTr2dDecoderQ, qat_layers.py:303-435), and it is the data gradient of
 * the convolution.  fqss_chan_sum: out[c] += sum_{b,m} g[b][c][m] (bias gradient of the transposed convolutions).           */
int fqss_frames_gather(const float* x, float* frames, int64_t B, int64_t C, int64_t H, int64_t W, int64_t sb, int64_t sc, int64_t sh,
                       int kh, int kw, int st_h, int st_w, int ph, int pw, int dh, int dw, int64_t Ho, int64_t Wo, int64_t ld,
                       fqss_stream_t stream);
int fqss_frames_ola(const float* frames, const float* bias, float* y, int64_t B, int64_t C, int64_t H, int64_t W, int64_t sb, int64_t sc,
                    int64_t sh, int kh, int kw, int st_h, int st_w, int ph, int pw, int dh, int dw, int64_t Ho, int64_t Wo, int64_t ld,
                    fqss_stream_t stream);
int fqss_chan_sum(const float* g, float* out, int64_t B, int64_t C, int64_t M, int64_t ld, fqss_stream_t stream);

/* Streaming attention core for long sequences and cross attention (HTDemucs transformer, htdemucsq.py:138-329; the arithmetic of
 * MultiheadAttentionQ.forward between the quantized q / k / v and the heads, qat_layers.py:903-911, with Lq != Lk allowed).
 * Rows are x[l*sl + b*sb + h*hd + d]; `strides` is a HOST array of (sl, sb) element-stride pairs: q, k, v, o for the forward,
 * q, k, v, o, go, gq, gk, gv for the backward.  stats [B*nh][Lq][2] = (row max, row sum); dsum: workspace of B*nh*Lq floats.
 * obs_attn / obs_soft: optional observer workspaces (min / max of the logits and of the probabilities), as fqss_attn_fwd.  */
int fqss_attn_long_fwd(const float* q, const float* k, const float* v, float* o, float* stats, int Lq, int Lk, int B, int nh, int hd,
                       const int64_t* strides, uint32_t* obs_attn, uint32_t* obs_soft, fqss_stream_t stream);
int fqss_attn_long_bwd(const float* q, const float* k, const float* v, const float* o, const float* go, const float* stats, float* gq,
                       float* gk, float* gv, float* dsum, int Lq, int Lk, int B, int nh, int hd, const int64_t* strides,
                       fqss_stream_t stream);

/* Small streaming kernels of the HTDemucs layers (csrc/hd_ops.hip).
 * fqss_chan_op: y[b][c][m] = x[b][c][m] * s[c] (mode 0: LayerScale on channel-first tensors, demucsq.py:19-39) or + s[c] (mode 1:
 *   the frequency-embedding add, htdemucsq.py:1063-1068, c = channel x frequency); fqss_chan_scale_bwd: gx = g * s[c],
 *   gs[c] += sum g * x (gs caller-zeroed).  fqss_col_scale_fwd/bwd: the same on channel-last rows [R][C] (transformer gamma_1/2).
 * fqss_sample_meanstd: ms[b] = (mean, unbiased std) of x[b][:n] (ws: 2*B doubles, zeroed by the caller);
 * fqss_sample_norm: dir 0: (x - mean_b) / (1e-5 + std_b), dir 1: x * std_b + mean_b  (htdemucsq.py:1003-1014, 1034-1035).   */
int fqss_chan_op(const float* x, const float* s, float* y, int64_t B, int64_t C, int64_t M, int64_t ld_x, int64_t ld_y, int mode,
                 fqss_stream_t stream);
int fqss_chan_scale_bwd(const float* g, const float* x, const float* s, float* gx, float* gs, int64_t B, int64_t C, int64_t M,
                        int64_t ld_g, int64_t ld_x, int64_t ld_gx, fqss_stream_t stream);
int fqss_col_scale_fwd(const float* x, const float* s, float* y, int64_t R, int C, int64_t ld_x, int64_t ld_y, fqss_stream_t stream);
int fqss_col_scale_bwd(const float* g, const float* x, const float* s, float* gx, float* gs, int64_t R, int C, int64_t ld_g, int64_t ld_x,
                       int64_t ld_gx, fqss_stream_t stream);
int fqss_sample_meanstd(const float* x, double* ws, float* ms, int64_t B, int64_t n, fqss_stream_t stream);
int fqss_sample_norm(const float* x, const float* ms, float* y, int64_t B, int64_t n, int dir, fqss_stream_t stream);

/* Spectrogram pair of HTDemucsQ (csrc/stft.hip; replaces demucs.spec.spectro / ispectro = torch.stft / istft as called by
 * HTDemucsQ._spec / _ispec, htdemucsq.py:931-960).  win [N]: periodic Hann window; tw [N/2] complex: exp(-2 pi i k / N);
 * env [hop]: sum over the N/hop overlapping frames of win^2.  Spectra are [rows][2 (re, im)][T][N/2] with the Nyquist bin dropped.
 * fqss_stft: frames f = 0 .. T-1 of the signal reflect-padded by `pad` on the left (and as needed on the right), scaled by
 *   1/sqrt(N).  fqss_istft: y[j], j < length, of the inverse with 2 zero frames either side, window-envelope division and the
 *   crop by N/2 + pad; `frames` is a workspace of rows*T*N floats.  fqss_istft_bwd: its adjoint (gradient wrt the spectrum).
 * fqss_transpose2d: y[b][c][r] = x[b][r][c] (dense), the layout change between [.., T, Fr] and the model's [.., Fr, T].         */
int fqss_stft(const float* x, float* z, const float* win, const float* tw, int64_t rows, int64_t L, int64_t ld_x, int N, int hop, int T,
              int pad, fqss_stream_t stream);
int fqss_istft(const float* z, float* frames, float* y, const float* win, const float* env, const float* tw, int64_t rows, int64_t length,
               int64_t ld_y, int N, int hop, int T, int pad, fqss_stream_t stream);
int fqss_istft_bwd(const float* g, float* gz, const float* win, const float* env, const float* tw, int64_t rows, int64_t length,
                   int64_t ld_g, int N, int hop, int T, int pad, fqss_stream_t stream);
int fqss_transpose2d(const float* x, float* y, int64_t batch, int64_t R, int64_t C, fqss_stream_t stream);

/* Training loss of the htdemucs environment (csrc/hd_loss.hip; solver.py:333-366 with demucs' new_sdr): L1 task loss + SDR-weighted
 * L1 distillation loss per source, source weights wt [S].  est / fest / src [B][S][N] dense.  sums: B*S*5 doubles zeroed by the
 * caller; out: [loss, task_s.., kd_s.., w_bs..] = 1 + 2S + B*S floats; coef: B*S*2 floats of scratch; gest (optional) = dloss/dest. */
int fqss_hd_kd_loss(const float* est, const float* fest, const float* src, const float* wt, double* sums, float* out, float* coef,
                    float* gest, int B, int S, int64_t N, float kd_lambda, fqss_stream_t stream);

/* Evaluation side (csrc/infer.hip): the SI-SNR matrix between S estimates and S targets (torchmetrics'
 * ScaleInvariantSignalNoiseRatio restated: zero-mean SI-SDR, eps 2^-23) with the re-ordering decision of swap_channel_order
 * (process.py:105-125) as map[d] = (source estimate, sign); the triangular weighted overlap-add of process.model_infer
 * (:160-183) and its final normalisation.  mom: S*S*5 doubles zeroed by the caller.                                          */
int fqss_sisnr_matrix(const float* est, const float* ref, double* mom, float* db, int* map, int S, int64_t L, int64_t ld_e, int64_t ld_r,
                      fqss_stream_t stream);
int fqss_infer_ola(const float* chunk, const int* map, float* out, float* sum_weight, int S, int C, int64_t n, int64_t seg, int64_t start,
                   int64_t ld_chunk, int64_t ld_out, fqss_stream_t stream);
int fqss_infer_normalize(float* out, const float* sum_weight, int64_t rows, int64_t L, int64_t ld, fqss_stream_t stream);

/* Affine (scale, zero-point) form of the learned quantizers for the true-integer export (csrc/export_q.hip; SURVEY.md §8(f) rank 3;
 * replaces torch.fake_quantize_per_tensor_affine / per_channel_affine inside TorchWeightFakeQuantize / TorchActivationFakeQuantize /
 * TorchDymActivationFakeQuantize, qat_quant.py:15-72).  x viewed as [outer][C][inner]; scale / zp [C] on the device (C = 1: per
 * tensor); y (optional) the de-quantized values, codes (optional, int32) the integers q in [qmin, qmax].                       */
int fqss_fq_affine(const float* x, float* y, int* codes, int64_t outer, int64_t C, int64_t inner, const float* scale, const int* zp, int qmin,
                   int qmax, fqss_stream_t stream);

/* Data side (csrc/data_ops.hip): the SNR augmentation of the LibriMix dataset, batched on the device --
 * process.generate_2mix_snr (mode 0) / generate_mix_noise (mode 1) followed by max_clip(0.9) (process.py:57-103), one (a, b, snr) triple
 * per row.  ws: 2*B doubles, peak: B uint32, both zeroed by the caller; clip = 0 skips max_clip.                                  */
int fqss_snr_mix(const float* a, const float* b, const float* snr, double* ws, uint32_t* peak, float* out, int64_t B, int64_t T, int64_t ld_a,
                 int64_t ld_b, int64_t ld_o, int mode, int clip, fqss_stream_t stream);
/* Polyphase sinc resampler of the LibriMix dataset (librimix_dataset.py:54: torchaudio.transforms.Resample(sample_rate, resample *
 * sample_rate), applied to every clip read at :111-165; torchaudio is third party and absent: the taps follow its published
 * _get_sinc_resample_kernel -- sinc x Hann window, lowpass_filter_width 6, rolloff 0.99 -- and are computed by the host in fp64).
 * x [rows][L] -> y [rows][Lout], Lout <= ceil(new * L / orig); orig / new = the REDUCED ratio (2 / 1 for 16 -> 8 kHz);
 * h [new][2 * width + orig] fp32 taps.                                                                                           */
int fqss_resample_fir(const float* x, const float* h, float* y, int64_t rows, int64_t L, int64_t Lout, int64_t ld_x, int64_t ld_y,
                      int orig, int newf, int width, fqss_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Descriptor-struct forms of the long entry points ("fqss_<op>_{fwd,bwd}(const FqssTensor* in..,
 * FqssTensor* out.., const FqssQParams*, void* workspace, size_t ws_bytes, void* hip_stream)").  Same kernels, same
 * results as the flat forms above (tests/test_gpu_desc_api.py compares them bit for bit); the library validates dtype,
 * rank, innermost-contiguity and the workspace size from the descriptors before it launches anything.
 * ------------------------------------------------------------------------------------------- */
#define FQSS_DT_F32 0
#define FQSS_DT_U8 1  /* activation codes */
#define FQSS_DT_I8 2  /* weight codes */
#define FQSS_DT_F64 3
#define FQSS_DT_U16 4 /* bf16 planes of the frozen teacher weights */
#define FQSS_DT_I64 5
typedef struct FqssTensor {
    void* data;        /* caller-owned device pointer */
    int dtype;         /* FQSS_DT_* */
    int ndim;          /* 1..4 */
    int64_t shape[4];
    int64_t stride[4]; /* elements; the innermost stride must be 1 */
} FqssTensor;
typedef struct FqssQParams { /* one activation quantizer (GradientActivationFakeQuantize, qat_quant.py:171-249) */
    const float* qmin;       /* device scalars min_range / max_range */
    const float* qmax;
    int act;                 /* FQSS_ACT_* in front of the quantizer */
    const float* slope;      /* PReLU slope (device scalar) or NULL */
    double* gacc;            /* backward only: range / slope gradient partials (+=), or NULL */
} FqssQParams;
typedef struct FqssWCodes { /* per-channel int8 weight codes of a pointwise conv, from fqss_wq_codes / fqss_wq_multi_fwd */
    const int8_t* idx;      /* [Co][Ci] */
    const int8_t* idxT;     /* [Ci][Co] */
    const float* dw;        /* [Co] step */
    const float* rw;        /* [Co] integer row sums */
    int Co, Ci;
} FqssWCodes;
typedef struct FqssProducer { /* a pointwise conv whose output-quantizer backward rides in its consumer's backward kernel */
    const FqssTensor* z;      /* its pre-quant output [B][C][M] fp32 */
    int act;
    const float* slope;
    double* gacc;             /* its range / slope partials (+=) */
    float* gbias;             /* [C] (+=) or NULL */
    FqssTensor* out;          /* receives dL/dz of that layer */
} FqssProducer;
typedef struct FqssTGemmDesc { /* one launch of the teacher chain's GEMM (fqss_tgemm) */
    const FqssTensor* planes;  /* [3][Co][Ci] u16 */
    const FqssTensor* x;       /* [B][Ci][M] fp32 */
    int pro;                   /* 0 none | 1 GroupNorm affine | 2 PReLU on the input */
    const double* pro_stats; const float* pro_gamma; const float* pro_beta; float pro_eps; const float* pro_slope;
    const float* bias; int act; const float* slope;
    double* stats_out;         /* nullable */
    int M1;                    /* rows [0,M1) -> c1 (+r1), the rest -> c2 (+r2) */
    FqssTensor* c1; const FqssTensor* r1; FqssTensor* c2; const FqssTensor* r2;
} FqssTGemmDesc;

/* bytes of caller-provided workspace an op needs for activations of `shape` = {B, C, M}; ops: "gln_fq_fwd", "gln_fq_bwd",
 * "pwconv_fq_fwd" (C = Co: the optional code-statistics slots), "dwconv_fq_fwd" (same), "add_fq_fwd", "add_fq_bwd",
 * "tgemm" (0).  Unknown op / bad shape: -1 (fqss_last_error says which). */
int64_t fqss_workspace_bytes(const char* op, const int64_t* shape, int ndim);
/* The backward of a CHAIN of AddQ layers in one launch: out_l = fq_l(dec(out_{l-1}) + dec(b_l)), l = 0 .. nlev-1, every out_l
 * consumed by the next add alone and every b_l the fresh output of a pointwise conv without activation -- the skip sum of
 * MaskGenerator.forward (/root/reference/quantization/qat/models/convtasnetq.py:107-111: `output = self.adds[idx](output, skip)`).
 * Replaces nlev launches of fqss_ewq_bwd_p (same bits: same thread -> element map, summation orders and slots); the gradient travels
 * from the top level to the bottom in registers.  levels[0] = the first add of the forward; per level: codes of both operands and
 * their ranges, the add's own range + gacc, the producer of b (its pre-quant z, its gz out, its gacc, its bias gradient [C] or NULL).
 * g: dL/d(out of the top level).  The bottom level's first operand: EITHER a conv output too (az / aout / agacc / agbias) OR a tensor
 * whose gradient is written to ga_out.  rows = batch x C.  fqss_add_chain_ok: 1 when the shape is served (2 <= nlev <= 24, batch <= 8,
 * cols <= 65536). */
typedef struct {
    const uint8_t* ac; const uint8_t* bc; const float* bz; float* bout;
    const float *amin, *amax, *bmin, *bmax, *qmin, *qmax;
    double* gacc; double* bgacc; float* bgbias;
} FqssAddChainLevel;
int fqss_add_chain_ok(int64_t rows, int64_t cols, int C, int nlev);
int fqss_add_chain_bwd(const FqssAddChainLevel* levels, int nlev, const float* g, int64_t ld_g, float* ga_out, int64_t ld_ga,
                       const float* az, int64_t ld_az, float* aout, int64_t ld_aout, double* agacc, float* agbias,
                       int64_t rows, int64_t cols, int C, int64_t ld_a, int64_t ld_b, int64_t ld_bz, int64_t ld_bout,
                       fqss_stream_t stream);
/* AddQ / Sub / NlQ on codes (fqss_ewq_fwd): b / qb NULL for the unary form; b of dtype F32 is a real operand; y_out nullable */
int fqss_add_fq_fwd(const FqssTensor* a, const FqssQParams* qa, const FqssTensor* b, const FqssQParams* qb, float sb,
                    FqssTensor* y, FqssTensor* y_out, const FqssQParams* q, void* ws, size_t ws_bytes, fqss_stream_t stream);
/* its backward (fqss_ewq_bwd, or fqss_ewq_bwd_p when pa / pb name producers; then b must be codes and gz may be NULL) */
int fqss_add_fq_bwd(const FqssTensor* a, const FqssQParams* qa, const FqssTensor* b, const FqssQParams* qb, float sb,
                    const FqssTensor* g, FqssTensor* gz, const FqssQParams* q, const FqssProducer* pa, const FqssProducer* pb,
                    void* ws, size_t ws_bytes, fqss_stream_t stream);
/* Conv1dQ / Conv1dNlQ 1x1 on codes with the output quantizer fused (fqss_qpw_fwdq); w holds Co1 + Co2 rows; the second
 * layer's tensors are NULL for a single conv.  ws: when ws_bytes >= fqss_workspace_bytes("pwconv_fq_fwd") > 0 and Co2 = 0 the
 * statistics slots of y1 are written there ([B][slots][2] int64) for fqss_gln_fq_fwd */
int fqss_pwconv_fq_fwd(const FqssTensor* x, const FqssQParams* qx, const FqssWCodes* w, const float* bias1, const float* bias2,
                       FqssTensor* z1, FqssTensor* z2, FqssTensor* y1, FqssTensor* y2, const FqssQParams* q1,
                       const FqssQParams* q2, void* ws, size_t ws_bytes, fqss_stream_t stream);
/* GroupNormQ on codes (fqss_gnq_fwd); stats / nslots as there (then ws may be NULL) */
int fqss_gln_fq_fwd(const FqssTensor* x, const FqssQParams* qx, const float* gamma, const float* beta, float eps, FqssTensor* y,
                    FqssTensor* y_out, float* mean_rstd, const FqssQParams* q, void* ws, size_t ws_bytes, const int64_t* stats,
                    int nslots, fqss_stream_t stream);
int fqss_tgemm_desc(const FqssTGemmDesc* d, fqss_stream_t stream);


#ifdef __cplusplus
}
#endif
#endif /* FQSS_H */
