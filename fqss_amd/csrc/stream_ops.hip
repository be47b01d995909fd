// stream_ops.hip -- K6/K7/K8/K9/K14: the HBM-streaming producers of the TCN block:
// depthwise dilated conv, GroupNorm(1 group), add/sub, mask*feats broadcast multiply.
// All are pure memory-bound: 16 B/lane loads where rows are aligned, fp64 accumulation for the
// statistics (the fp64 VALU is far from being the limiter of a 2-flop/byte kernel).
//
// Reference replaced: convtasnetq.py:28-30 (depthwise F.conv1d), nn.GroupNorm(1,C,eps=1e-8)
// (qat_layers.py:445-448), torch.add/sub/mul (qat_layers.py:69-71, 93-96, 1193), and their autograd.
#define FQSS_USES_GRAD_ADD   // the fp32 gradient atomics of this file go through grad_add (fqss_dev.h: FQSS_DETERMINISTIC=1)
#include "fqss_dev.h"

namespace fqss {

// =============================================================================================
// z = sa*a + sb*b       (sa=1, sb=+1: AddQ; sb=-1: ResidualErrorBlock's Y - Y_q; multiplies by +-1 / 2^k are exact)
// =============================================================================================
template <int VEC>
__global__ __launch_bounds__(256) void k_axpby(const float* __restrict__ a, const float* __restrict__ b, float sa, float sb,
                                                float* __restrict__ z, int64_t rows, int64_t cols, int64_t ld_a,
                                                int64_t ld_b, int64_t ld_z) {
    const int64_t cstep = (int64_t)gridDim.x * 256 * VEC;
    for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
        const float* ar = a + row * ld_a;
        const float* br = b + row * ld_b;
        float* zr = z + row * ld_z;
        for (int64_t c0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC; c0 < cols; c0 += cstep) {
            if constexpr (VEC == 4) {
                const float4 x = *reinterpret_cast<const float4*>(ar + c0);
                const float4 y = *reinterpret_cast<const float4*>(br + c0);
                *reinterpret_cast<float4*>(zr + c0) =
                    make_float4(sa * x.x + sb * y.x, sa * x.y + sb * y.y, sa * x.z + sb * y.z, sa * x.w + sb * y.w);
            } else {
                zr[c0] = sa * ar[c0] + sb * br[c0];
            }
        }
    }
}

// =============================================================================================
// z[b][s][c][:] = mask[b][s][c][:] * feat[b][c][:]        (MulQ, convtasnetq.py:209)
// =============================================================================================
template <int VEC>
__global__ __launch_bounds__(256) void k_mul_bcast_fwd(const float* __restrict__ mask, const float* __restrict__ feat,
                                                        float* __restrict__ z, int B, int S, int C, int M,
                                                        int64_t ld_m, int64_t ld_f, int64_t ld_z) {
    const int64_t rows = (int64_t)B * S * C;
    const int64_t cstep = (int64_t)gridDim.x * 256 * VEC;
    for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
        const int64_t b = row / ((int64_t)S * C), c = row % C;
        const float* mr = mask + row * ld_m;
        const float* fr = feat + (b * C + c) * ld_f;
        float* zr = z + row * ld_z;
        for (int64_t c0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC; c0 < M; c0 += cstep) {
            if constexpr (VEC == 4) {
                const float4 x = *reinterpret_cast<const float4*>(mr + c0);
                const float4 y = *reinterpret_cast<const float4*>(fr + c0);
                *reinterpret_cast<float4*>(zr + c0) = make_float4(x.x * y.x, x.y * y.y, x.z * y.z, x.w * y.w);
            } else {
                zr[c0] = mr[c0] * fr[c0];
            }
        }
    }
}

// gmask = gz*feat ; gfeat = sum_s gz*mask   (one pass: each thread owns (b,c,m) and walks s)
template <int VEC>
__global__ __launch_bounds__(256) void k_mul_bcast_bwd(const float* __restrict__ gz, const float* __restrict__ mask,
                                                        const float* __restrict__ feat, float* __restrict__ gmask,
                                                        float* __restrict__ gfeat, int B, int S, int C, int M,
                                                        int64_t ld_gz, int64_t ld_m, int64_t ld_f, int64_t ld_gm,
                                                        int64_t ld_gf) {
    const int64_t rows = (int64_t)B * C;
    const int64_t cstep = (int64_t)gridDim.x * 256 * VEC;
    for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
        const int64_t b = row / C, c = row % C;
        const float* fr = feat + row * ld_f;
        float* gfr = gfeat + row * ld_gf;
        for (int64_t c0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC; c0 < M; c0 += cstep) {
            float f[VEC], acc[VEC];
            if constexpr (VEC == 4) {
                const float4 t = *reinterpret_cast<const float4*>(fr + c0);
                f[0] = t.x; f[1] = t.y; f[2] = t.z; f[3] = t.w;
            } else {
                f[0] = fr[c0];
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] = 0.0f;
            for (int s = 0; s < S; ++s) {
                const int64_t r2 = (b * S + s) * C + c;
                float gv[VEC], mv[VEC], o[VEC];
                if constexpr (VEC == 4) {
                    const float4 t = *reinterpret_cast<const float4*>(gz + r2 * ld_gz + c0);
                    const float4 u = *reinterpret_cast<const float4*>(mask + r2 * ld_m + c0);
                    gv[0] = t.x; gv[1] = t.y; gv[2] = t.z; gv[3] = t.w;
                    mv[0] = u.x; mv[1] = u.y; mv[2] = u.z; mv[3] = u.w;
                } else {
                    gv[0] = gz[r2 * ld_gz + c0];
                    mv[0] = mask[r2 * ld_m + c0];
                }
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    o[j] = gv[j] * f[j];
                    acc[j] = acc[j] + gv[j] * mv[j];
                }
                if constexpr (VEC == 4)
                    *reinterpret_cast<float4*>(gmask + r2 * ld_gm + c0) = make_float4(o[0], o[1], o[2], o[3]);
                else
                    gmask[r2 * ld_gm + c0] = o[0];
            }
            if constexpr (VEC == 4)
                *reinterpret_cast<float4*>(gfr + c0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            else
                gfr[c0] = acc[0];
        }
    }
}

// =============================================================================================
// depthwise dilated conv (cross-correlation, zero padding), K <= 8 taps
// =============================================================================================
constexpr int kMaxTaps = 8;

// FLIP=false: z[m] = bias + sum_k w[k] * x[m + k*dil - pad]            (forward)
// FLIP=true : gx[m] =        sum_k w[k] * gz[m - k*dil + pad]          (input gradient)
template <bool FLIP>
__global__ __launch_bounds__(256) void k_dwconv(const float* __restrict__ x, const float* __restrict__ w,
                                                 const float* __restrict__ bias, float* __restrict__ z, int64_t rows,
                                                 int C, int M, int K, int dil, int pad, int64_t ld_x, int64_t ld_z) {
    const int64_t cstep = (int64_t)gridDim.x * 256;
    for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
        const int c = (int)(row % C);
        const float* xr = x + row * ld_x;
        float* zr = z + row * ld_z;
        float wk[kMaxTaps];
#pragma unroll
        for (int k = 0; k < kMaxTaps; ++k) wk[k] = (k < K) ? w[c * K + k] : 0.0f;
        const float bv = (!FLIP && bias != nullptr) ? bias[c] : 0.0f;
        for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < M; m += cstep) {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < kMaxTaps; ++k) {
                if (k < K) {
                    const int64_t src = FLIP ? (m - (int64_t)k * dil + pad) : (m + (int64_t)k * dil - pad);
                    if (src >= 0 && src < M) acc = fmaf(wk[k], xr[src], acc);
                }
            }
            zr[m] = acc + bv;
        }
    }
}

// 16-B/lane variant: each thread produces 4 consecutive outputs; a tap whose offset is a multiple of
// 4 floats (dil = 4..128 and the centre tap) is ONE aligned float4 load, the others are L1-resident
// scalar loads.  Rows must be 16-B aligned (ld % 4 == 0).
template <bool FLIP>
__global__ __launch_bounds__(256) void k_dwconv_v4(const float* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ bias, float* __restrict__ z,
                                                    int64_t rows, int C, int M, int K, int dil, int pad, int64_t ld_x,
                                                    int64_t ld_z) {
    const int64_t cstep = (int64_t)gridDim.x * 256 * 4;
    for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
        const int c = (int)(row % C);
        const float* xr = x + row * ld_x;
        float* zr = z + row * ld_z;
        float wk[kMaxTaps];
#pragma unroll
        for (int k = 0; k < kMaxTaps; ++k) wk[k] = (k < K) ? w[c * K + k] : 0.0f;
        const float bv = (!FLIP && bias != nullptr) ? bias[c] : 0.0f;
        for (int64_t m = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; m < M; m += cstep) {
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < kMaxTaps; ++k) {
                if (k < K) {
                    const int off = FLIP ? (pad - k * dil) : (k * dil - pad);
                    const int64_t s0 = m + off;
                    float v[4];
                    if ((off & 3) == 0 && s0 >= 0 && s0 + 3 < ld_x) {
                        const float4 t = *reinterpret_cast<const float4*>(xr + s0);
                        v[0] = t.x;
                        v[1] = (s0 + 1 < M) ? t.y : 0.0f;
                        v[2] = (s0 + 2 < M) ? t.z : 0.0f;
                        v[3] = (s0 + 3 < M) ? t.w : 0.0f;
                        if (s0 >= M) v[0] = 0.0f;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int64_t sj = s0 + j;
                            v[j] = (sj >= 0 && sj < M) ? xr[sj] : 0.0f;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = fmaf(wk[k], v[j], acc[j]);
                }
            }
            *reinterpret_cast<float4*>(zr + m) = make_float4(acc[0] + bv, acc[1] + bv, acc[2] + bv, acc[3] + bv);
        }
    }
}

// gw[c][k] += sum_{b,m} gz[b][c][m] * x[b][c][m + k*dil - pad]   ; grid (C, B)
__global__ __launch_bounds__(256) void k_dwconv_bwd_w(const float* __restrict__ gz, const float* __restrict__ x,
                                                       float* gw, int C, int M, int K, int dil, int pad, int64_t ld_gz,
                                                       int64_t ld_x) {
    __shared__ double red[kMaxTaps * 4];
    const int c = blockIdx.x, b = blockIdx.y;
    const int64_t row = (int64_t)b * C + c;
    const float* gr = gz + row * ld_gz;
    const float* xr = x + row * ld_x;
    float p[kMaxTaps];
#pragma unroll
    for (int k = 0; k < kMaxTaps; ++k) p[k] = 0.0f;
    for (int m = threadIdx.x; m < M; m += 256) {
        const float g = gr[m];
#pragma unroll
        for (int k = 0; k < kMaxTaps; ++k) {
            if (k < K) {
                const int64_t src = m + (int64_t)k * dil - pad;
                if (src >= 0 && src < M) p[k] = fmaf(g, xr[src], p[k]);
            }
        }
    }
    double v[kMaxTaps];
#pragma unroll
    for (int k = 0; k < kMaxTaps; ++k) v[k] = (double)p[k];
    block_sum<double, kMaxTaps>(v, red);
    if (threadIdx.x == 0)
        for (int k = 0; k < K; ++k) grad_add(&gw[c * K + k], (float)v[k]);
}

// =============================================================================================
// GroupNorm(1, C): statistics over the C*M elements of one sample
// =============================================================================================
template <int VEC>
__global__ __launch_bounds__(256) void k_gn_stats(const float* __restrict__ x, int C, int M, int64_t ld, double* ws) {
    __shared__ double red[2 * 4];
    const int b = blockIdx.y;
    double s = 0.0, ss = 0.0;
    for (int c = blockIdx.x; c < C; c += gridDim.x) {
        const float* xr = x + ((int64_t)b * C + c) * ld;
        // (gridDim.z column slices: 6 channels x 4 samples of 880 k positions each were 24 workgroups)
        for (int m = ((int)blockIdx.z * 256 + (int)threadIdx.x) * VEC; m < M; m += (int)gridDim.z * 256 * VEC) {
            if constexpr (VEC == 4) {
                const float4 t = *reinterpret_cast<const float4*>(xr + m);
                const float v[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (m + j < M) {
                        s += (double)v[j];
                        ss += (double)v[j] * (double)v[j];
                    }
            } else {
                const double v = (double)xr[m];
                s += v;
                ss += v * v;
            }
        }
    }
    double v[2] = {s, ss};
    block_sum<double, 2>(v, red);
    if (threadIdx.x == 0) {
        atomicAdd(&ws[2 * b], v[0]);
        atomicAdd(&ws[2 * b + 1], v[1]);
    }
}

__device__ __forceinline__ void gn_mean_rstd(const double* ws, int b, int64_t n, float eps, float& mean, float& rstd) {
    const double mu = ws[2 * b] / (double)n;
    double var = ws[2 * b + 1] / (double)n - mu * mu;
    if (var < 0.0) var = 0.0;
    mean = (float)mu;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
}

// y = x*scale + shift ; scale = rstd*gamma[c] ; shift = beta[c] - scale*mean   (ATen GroupNorm kernel form)
// Q: GroupNormQ on a float input in the quantizing phase -- z receives fq(GroupNorm(x)) (the arithmetic of this kernel followed by
// fqss_actq_fwd, value for value); the pre-quant z is never stored, the backward recomputes it from x
template <int VEC, bool Q = false>
__global__ __launch_bounds__(256) void k_gn_apply(const float* __restrict__ x, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, float* __restrict__ z,
                                                   float* mean_rstd, int B, int C, int M, int64_t ld_x, int64_t ld_z,
                                                   float eps, const double* ws, const float* qmin = nullptr, const float* qmax = nullptr,
                                                   uint8_t* __restrict__ yc = nullptr, int64_t ld_yc = 0) {
    // yc (Q only, nullable): the u8 codes of the output, rows of ld_yc bytes (VEC = 4: ld_yc % 4 == 0, 4-B aligned rows)
    QRange qr{0.0f, 1.0f, 1.0f};
    if (Q) qr = load_qrange(qmin, qmax);
    const int64_t rows = (int64_t)B * C;
    const int64_t cstep = (int64_t)gridDim.x * 256 * VEC;
    for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
        const int b = (int)(row / C), c = (int)(row % C);
        float mean, rstd;
        gn_mean_rstd(ws, b, (int64_t)C * M, eps, mean, rstd);
        if (c == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
            mean_rstd[2 * b] = mean;
            mean_rstd[2 * b + 1] = rstd;
        }
        const float scale = rstd * gamma[c];
        const float shift = fmaf(-scale, mean, beta[c]);
        const float* xr = x + row * ld_x;
        float* zr = z + row * ld_z;
        for (int64_t c0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC; c0 < M; c0 += cstep) {
            if constexpr (VEC == 4) {
                const float4 t = *reinterpret_cast<const float4*>(xr + c0);
                float o[4] = {fmaf(t.x, scale, shift), fmaf(t.y, scale, shift), fmaf(t.z, scale, shift), fmaf(t.w, scale, shift)};
                if constexpr (Q) {
                    unsigned int pk = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float cq, u;
                        bool inr;
                        o[j] = fq_asym(o[j], qr, cq, u, inr);
                        pk = pack_code(cq, j, pk);
                    }
                    if (yc != nullptr) *reinterpret_cast<unsigned int*>(yc + row * ld_yc + c0) = pk;      // (row padding absorbs c0 + 3 >= M)
                }
                *reinterpret_cast<float4*>(zr + c0) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
                float o = fmaf(xr[c0], scale, shift);
                if constexpr (Q) {
                    float cq, u;
                    bool inr;
                    o = fq_asym(o, qr, cq, u, inr);
                    if (yc != nullptr) yc[row * ld_yc + c0] = (uint8_t)cq;
                }
                zr[c0] = o;
            }
        }
    }
}

// Forward-only tails of a GroupNorm for a network that runs without autograd (the frozen float TEACHER of HTDemucs: every DConv layer is
// conv -> GroupNorm -> GELU -> 1x1 conv -> GroupNorm -> GLU -> LayerScale, added to the residual; demucsq.py:163-182).  The un-fused
// chain moves the 2C-channel tensor through apply, GLU, LayerScale and the add (7 passes of it per layer); here the second pass of the
// GroupNorm carries what follows, in the arithmetic of the separate kernels (k_gn_apply, k_unary_fwd / k_glu_fwd, k_chan_op, k_axpby):
//   TAIL = 1:  y = gelu(gn(x))                                         rows = B * C
//   TAIL = 2:  y[b][c] = (a * sigmoid(g)) * ls[c] + res[b][c],  a = gn(x)[b][c], g = gn(x)[b][c + C/2]     rows = B * C/2
template <int TAIL>
__global__ __launch_bounds__(256) void k_gn_tail(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  float* __restrict__ y, int B, int C, int M, int64_t ld_x, int64_t ld_y, float eps,
                                                  const double* ws, const float* __restrict__ ls, const float* __restrict__ res, int64_t ld_res,
                                                  int res_vec) {
    // res_vec = 0: the residual's rows are dense with a length that is no multiple of 4 (a module INPUT, e.g. [B, 48, 110250]): 4-B loads
    const int Co = TAIL == 2 ? C / 2 : C;
    const int64_t rows = (int64_t)B * Co;
    const int64_t cstep = (int64_t)gridDim.x * 256 * 4;
    for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
        const int b = (int)(row / Co), c = (int)(row % Co);
        float mean, rstd;
        gn_mean_rstd(ws, b, (int64_t)C * M, eps, mean, rstd);
        const float sa = rstd * gamma[c], sha = fmaf(-sa, mean, beta[c]);
        const float* xa = x + ((int64_t)b * C + c) * ld_x;
        float* yr = y + row * ld_y;
        if constexpr (TAIL == 1) {
            for (int64_t c0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; c0 < M; c0 += cstep) {
                const float4 t = *reinterpret_cast<const float4*>(xa + c0);
                const float v[4] = {fmaf(t.x, sa, sha), fmaf(t.y, sa, sha), fmaf(t.z, sa, sha), fmaf(t.w, sa, sha)};
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (0.5f * v[j]) * (1.0f + erff(v[j] * 0.70710678118654752440f));
                *reinterpret_cast<float4*>(yr + c0) = make_float4(o[0], o[1], o[2], o[3]);
            }
        } else {
            const float sg = rstd * gamma[c + Co], shg = fmaf(-sg, mean, beta[c + Co]);
            const float* xg = x + ((int64_t)b * C + c + Co) * ld_x;
            const float* rr = res + row * ld_res;
            const float lsv = ls[c];
            for (int64_t c0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; c0 < M; c0 += cstep) {
                const float4 ta = *reinterpret_cast<const float4*>(xa + c0), tg = *reinterpret_cast<const float4*>(xg + c0);
                float4 tr;
                if (res_vec) tr = *reinterpret_cast<const float4*>(rr + c0);
                else tr = make_float4(rr[min(c0, (int64_t)M - 1)], rr[min(c0 + 1, (int64_t)M - 1)], rr[min(c0 + 2, (int64_t)M - 1)], rr[min(c0 + 3, (int64_t)M - 1)]);
                const float a[4] = {fmaf(ta.x, sa, sha), fmaf(ta.y, sa, sha), fmaf(ta.z, sa, sha), fmaf(ta.w, sa, sha)};
                const float g[4] = {fmaf(tg.x, sg, shg), fmaf(tg.y, sg, shg), fmaf(tg.z, sg, shg), fmaf(tg.w, sg, shg)};
                const float r[4] = {tr.x, tr.y, tr.z, tr.w};
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = r[j] + (a[j] * (1.0f / (1.0f + expf(-g[j])))) * lsv;
                *reinterpret_cast<float4*>(yr + c0) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
    }
}

// backward pass 1: per (b,c) row sums  ds = sum gz*x, db = sum gz   -> ws[(b*C+c)*2 + {0,1}]
// Q: gz is dL/d fq(GroupNorm(x)): the quantizer's STE runs on the pre-quant value recomputed from x (scale / shift as in k_gn_apply)
// and its range partials go to gacc (slots shared modulo kGaccSlots, fp64 atomics: order-insensitive) -- no fqss_actq_bwd pass, no z
template <int VEC, bool Q = false>
__global__ __launch_bounds__(256) void k_gn_bwd_rows(const float* __restrict__ gz, const float* __restrict__ x, int C,
                                                      int M, int64_t ld_gz, int64_t ld_x, double* ws, const float* __restrict__ gamma = nullptr,
                                                      const float* __restrict__ beta = nullptr, const float* __restrict__ mean_rstd = nullptr,
                                                      const float* qmin = nullptr, const float* qmax = nullptr, double* gacc = nullptr) {
    __shared__ double red[4 * 4];
    const int64_t row = (int64_t)blockIdx.y * C + blockIdx.x;
    const float* gr = gz + row * ld_gz;
    const float* xr = x + row * ld_x;
    QRange qr{0.0f, 1.0f, 1.0f};
    float scale = 1.0f, shift = 0.0f, p_du = 0.0f, p_out = 0.0f;
    if (Q) {
        qr = load_qrange(qmin, qmax);
        const float mean = mean_rstd[2 * blockIdx.y], rstd = mean_rstd[2 * blockIdx.y + 1];
        scale = rstd * gamma[blockIdx.x];
        shift = fmaf(-scale, mean, beta[blockIdx.x]);
    }
    // the STE of one element: returns dL/dz and adds the element's range partials
    auto ste = [&](float g, float xv) {
        float cq, u;
        bool inr;
        (void)fq_asym(fmaf(xv, scale, shift), qr, cq, u, inr);
        p_du += g * (inr ? (cq - u) : cq);
        p_out += inr ? 0.0f : g;
        return inr ? div_by(g * qr.delta, qr.delta, qr.inv) : 0.0f;
    };
    double ds = 0.0, db = 0.0;
    for (int m = ((int)blockIdx.z * 256 + (int)threadIdx.x) * VEC; m < M; m += (int)gridDim.z * 256 * VEC) {
        if constexpr (VEC == 4) {
            const float4 a = *reinterpret_cast<const float4*>(gr + m);
            const float4 t = *reinterpret_cast<const float4*>(xr + m);
            const float gv[4] = {a.x, a.y, a.z, a.w}, xv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (m + j < M) {
                    const float gj = Q ? ste(gv[j], xv[j]) : gv[j];
                    ds += (double)gj * (double)xv[j];
                    db += (double)gj;
                }
        } else {
            const float gj = Q ? ste(gr[m], xr[m]) : gr[m];
            ds += (double)gj * (double)xr[m];
            db += (double)gj;
        }
    }
    if constexpr (Q) {
        double pv[2] = {(double)p_du, (double)p_out};
        block_sum<double, 2>(pv, red + 8);
        if (threadIdx.x == 0) {
            double* slot = gacc + 3 * (row % FQSS_GACC_SLOTS);
            const double dmax = pv[0] / 255.0;
            atomicAdd(&slot[0], pv[1] - dmax);
            atomicAdd(&slot[1], dmax);
        }
    }
    double v[2] = {ds, db};
    block_sum<double, 2>(v, red);
    if (threadIdx.x == 0) {
        if (gridDim.z == 1) {
            ws[2 * row] = v[0];
            ws[2 * row + 1] = v[1];
        } else {                    // column slices: the host zeroed the row sums
            atomicAdd(&ws[2 * row], v[0]);
            atomicAdd(&ws[2 * row + 1], v[1]);
        }
    }
}

// backward pass 2 (tiny): per-sample coefficients c2,c3 (ATen GroupNormBackward form) and gamma/beta grads
//   ds_sum = sum_c gamma_c ds[b][c], db_sum = sum_c gamma_c db[b][c]
//   c2 = (db_sum*mean - ds_sum) * rstd^3 / N ; c3 = -c2*mean - db_sum*rstd/N
//   ggamma[c] += sum_b (ds[b][c] - db[b][c]*mean_b)*rstd_b ; gbeta[c] += sum_b db[b][c]
__global__ __launch_bounds__(256) void k_gn_bwd_coef(const float* __restrict__ gamma, const float* __restrict__ mean_rstd,
                                                      int B, int C, int M, double* ws, float* ggamma, float* gbeta, int nbs) {
    __shared__ double red[2 * 4];
    const int b = blockIdx.x;
    double* coef = ws + 2 * (int64_t)B * C;
    if (b < B) {
        double s_ds = 0.0, s_db = 0.0;
        for (int c = threadIdx.x; c < C; c += 256) {
            const double gmm = (double)gamma[c];
            s_ds += gmm * ws[2 * ((int64_t)b * C + c)];
            s_db += gmm * ws[2 * ((int64_t)b * C + c) + 1];
        }
        double v[2] = {s_ds, s_db};
        block_sum<double, 2>(v, red);
        if (threadIdx.x == 0) {
            const double mean = (double)mean_rstd[2 * b], rstd = (double)mean_rstd[2 * b + 1];
            const double inv_n = 1.0 / ((double)C * (double)M);
            const double c2 = (v[1] * mean - v[0]) * rstd * rstd * rstd * inv_n;
            const double c3 = -c2 * mean - v[1] * rstd * inv_n;
            coef[2 * b] = c2;
            coef[2 * b + 1] = c3;
        }
    } else {
        // extra blocks (blockIdx.x >= B): parameter gradients, one thread per channel and -- with many samples (the [B F][C][T]
        // GroupNorms of the spectrogram DConv: 2,048 of them were ONE workgroup's serial loop, 460 us) -- one slice of the samples
        // per workgroup (nbs slices, float atomics; nbs = 1: the plain += of the small-batch models, order fixed)
        const int e = b - B, ncb = (C + 255) / 256;
        const int c = (e % ncb) * 256 + threadIdx.x, bs = e / ncb;
        const int per = (B + nbs - 1) / nbs, b0 = bs * per, b1 = min(B, b0 + per);
        if (c < C) {
            double gg = 0.0, gb = 0.0;
            for (int bb = b0; bb < b1; ++bb) {
                const double ds = ws[2 * ((int64_t)bb * C + c)], db = ws[2 * ((int64_t)bb * C + c) + 1];
                gg += (ds - db * (double)mean_rstd[2 * bb]) * (double)mean_rstd[2 * bb + 1];
                gb += db;
            }
            if (nbs == 1) {
                ggamma[c] += (float)gg;
                gbeta[c] += (float)gb;
            } else {
                grad_add(&ggamma[c], (float)gg);
                grad_add(&gbeta[c], (float)gb);
            }
        }
    }
}

// backward pass 3: gx = gz*(gamma_c*rstd) + x*c2 + c3
template <int VEC, bool Q = false>
__global__ __launch_bounds__(256) void k_gn_bwd_apply(const float* __restrict__ gz, const float* __restrict__ x,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ mean_rstd, float* __restrict__ gx,
                                                       int B, int C, int M, int64_t ld_gz, int64_t ld_x, int64_t ld_gx,
                                                       const double* ws, const float* __restrict__ beta = nullptr, const float* qmin = nullptr,
                                                       const float* qmax = nullptr) {
    QRange qr{0.0f, 1.0f, 1.0f};
    if (Q) qr = load_qrange(qmin, qmax);
    const double* coef = ws + 2 * (int64_t)B * C;
    const int64_t rows = (int64_t)B * C;
    const int64_t cstep = (int64_t)gridDim.x * 256 * VEC;
    for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
        const int b = (int)(row / C), c = (int)(row % C);
        const float c1 = mean_rstd[2 * b + 1] * gamma[c];
        const float c2 = (float)coef[2 * b], c3 = (float)coef[2 * b + 1];
        const float shift = Q ? fmaf(-c1, mean_rstd[2 * b], beta[c]) : 0.0f;      // (c1 is k_gn_apply's scale)
        // Q: dL/dz of an element from dL/dy -- the STE on the recomputed pre-quant value (the partial sums were taken in pass 1)
        auto ste = [&](float g, float xv) {
            float cq, u;
            bool inr;
            (void)fq_asym(fmaf(xv, c1, shift), qr, cq, u, inr);
            return inr ? div_by(g * qr.delta, qr.delta, qr.inv) : 0.0f;
        };
        const float* gr = gz + row * ld_gz;
        const float* xr = x + row * ld_x;
        float* orow = gx + row * ld_gx;
        for (int64_t c0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC; c0 < M; c0 += cstep) {
            if constexpr (VEC == 4) {
                float4 a = *reinterpret_cast<const float4*>(gr + c0);
                const float4 t = *reinterpret_cast<const float4*>(xr + c0);
                if constexpr (Q) { a.x = ste(a.x, t.x); a.y = ste(a.y, t.y); a.z = ste(a.z, t.z); a.w = ste(a.w, t.w); }
                *reinterpret_cast<float4*>(orow + c0) =
                    make_float4(fmaf(a.x, c1, fmaf(t.x, c2, c3)), fmaf(a.y, c1, fmaf(t.y, c2, c3)),
                                fmaf(a.z, c1, fmaf(t.z, c2, c3)), fmaf(a.w, c1, fmaf(t.w, c2, c3)));
            } else {
                const float a = Q ? ste(gr[c0], xr[c0]) : gr[c0];
                orow[c0] = fmaf(a, c1, fmaf(xr[c0], c2, c3));
            }
        }
    }
}

}  // namespace fqss

using namespace fqss;

#define FQSS_VEC_OK2(p0, l0, p1, l1) (aligned16(p0) && aligned16(p1) && ((l0) % 4 == 0) && ((l1) % 4 == 0))

extern "C" int fqss_axpby(const float* a, const float* b, float sa, float sb, float* z, int64_t rows, int64_t cols,
                          int64_t ld_a, int64_t ld_b, int64_t ld_z, fqss_stream_t stream) {
    FQSS_REQUIRE(a && b && z, "null tensor");
    FQSS_REQUIRE(rows >= 0 && cols >= 0 && ld_a >= cols && ld_b >= cols && ld_z >= cols, "bad shape");
    if (rows == 0 || cols == 0) return FQSS_OK;
    const bool vec = FQSS_VEC_OK2(a, ld_a, b, ld_b) && aligned16(z) && ld_z % 4 == 0;
    if (vec)
        hipLaunchKernelGGL(k_axpby<4>, grid_rows(rows, cols, 4), dim3(256), 0, (hipStream_t)stream, a, b, sa, sb, z, rows,
                           cols, ld_a, ld_b, ld_z);
    else
        hipLaunchKernelGGL(k_axpby<1>, grid_rows(rows, cols, 1), dim3(256), 0, (hipStream_t)stream, a, b, sa, sb, z, rows,
                           cols, ld_a, ld_b, ld_z);
    return launch_status("fqss_axpby");
}

extern "C" int fqss_mul_bcast_fwd(const float* mask, const float* feat, float* z, int B, int S, int C, int M,
                                  int64_t ld_mask, int64_t ld_feat, int64_t ld_z, fqss_stream_t stream) {
    FQSS_REQUIRE(mask && feat && z, "null tensor");
    FQSS_REQUIRE(B >= 0 && S > 0 && C > 0 && M >= 0 && ld_mask >= M && ld_feat >= M && ld_z >= M, "bad shape");
    if (B == 0 || M == 0) return FQSS_OK;
    const int64_t rows = (int64_t)B * S * C;
    const bool vec = FQSS_VEC_OK2(mask, ld_mask, feat, ld_feat) && aligned16(z) && ld_z % 4 == 0;
    if (vec)
        hipLaunchKernelGGL(k_mul_bcast_fwd<4>, grid_rows(rows, M, 4), dim3(256), 0, (hipStream_t)stream, mask, feat, z,
                           B, S, C, M, ld_mask, ld_feat, ld_z);
    else
        hipLaunchKernelGGL(k_mul_bcast_fwd<1>, grid_rows(rows, M, 1), dim3(256), 0, (hipStream_t)stream, mask, feat, z,
                           B, S, C, M, ld_mask, ld_feat, ld_z);
    return launch_status("fqss_mul_bcast_fwd");
}

extern "C" int fqss_mul_bcast_bwd(const float* gz, const float* mask, const float* feat, float* gmask, float* gfeat,
                                  int B, int S, int C, int M, int64_t ld_gz, int64_t ld_mask, int64_t ld_feat,
                                  int64_t ld_gmask, int64_t ld_gfeat, fqss_stream_t stream) {
    FQSS_REQUIRE(gz && mask && feat && gmask && gfeat, "null tensor");
    FQSS_REQUIRE(B >= 0 && S > 0 && C > 0 && M >= 0 && ld_gz >= M && ld_mask >= M && ld_feat >= M && ld_gmask >= M &&
                     ld_gfeat >= M, "bad shape");
    if (B == 0 || M == 0) return FQSS_OK;
    const int64_t rows = (int64_t)B * C;
    const bool vec = FQSS_VEC_OK2(gz, ld_gz, mask, ld_mask) && FQSS_VEC_OK2(feat, ld_feat, gmask, ld_gmask) &&
                     aligned16(gfeat) && ld_gfeat % 4 == 0;
    if (vec)
        hipLaunchKernelGGL(k_mul_bcast_bwd<4>, grid_rows(rows, M, 4), dim3(256), 0, (hipStream_t)stream, gz, mask, feat,
                           gmask, gfeat, B, S, C, M, ld_gz, ld_mask, ld_feat, ld_gmask, ld_gfeat);
    else
        hipLaunchKernelGGL(k_mul_bcast_bwd<1>, grid_rows(rows, M, 1), dim3(256), 0, (hipStream_t)stream, gz, mask, feat,
                           gmask, gfeat, B, S, C, M, ld_gz, ld_mask, ld_feat, ld_gmask, ld_gfeat);
    return launch_status("fqss_mul_bcast_bwd");
}

extern "C" int fqss_dwconv_fwd(const float* x, const float* w, const float* bias, float* z, int B, int C, int M, int K,
                               int dil, int pad, int64_t ld_x, int64_t ld_z, fqss_stream_t stream) {
    FQSS_REQUIRE(x && w && z, "null tensor");
    FQSS_REQUIRE(B >= 0 && C > 0 && M >= 0 && K > 0 && K <= kMaxTaps && dil > 0 && pad >= 0, "bad shape");
    FQSS_REQUIRE(2 * pad == dil * (K - 1), "only 'same' depthwise convs (2*pad == dil*(K-1)) are supported");
    FQSS_REQUIRE(ld_x >= M && ld_z >= M, "bad ld");
    if (B == 0 || M == 0) return FQSS_OK;
    const int64_t rows = (int64_t)B * C;
    if (aligned16(x) && aligned16(z) && ld_x % 4 == 0 && ld_z % 4 == 0)
        hipLaunchKernelGGL(k_dwconv_v4<false>, grid_rows(rows, M, 4), dim3(256), 0, (hipStream_t)stream, x, w, bias, z,
                           rows, C, M, K, dil, pad, ld_x, ld_z);
    else
        hipLaunchKernelGGL(k_dwconv<false>, grid_rows(rows, M, 1), dim3(256), 0, (hipStream_t)stream, x, w, bias, z, rows,
                           C, M, K, dil, pad, ld_x, ld_z);
    return launch_status("fqss_dwconv_fwd");
}

extern "C" int fqss_dwconv_bwd_x(const float* gz, const float* w, float* gx, int B, int C, int M, int K, int dil,
                                 int pad, int64_t ld_gz, int64_t ld_gx, fqss_stream_t stream) {
    FQSS_REQUIRE(gz && w && gx, "null tensor");
    FQSS_REQUIRE(B >= 0 && C > 0 && M >= 0 && K > 0 && K <= kMaxTaps && dil > 0 && pad >= 0, "bad shape");
    FQSS_REQUIRE(2 * pad == dil * (K - 1), "only 'same' depthwise convs are supported");
    FQSS_REQUIRE(ld_gz >= M && ld_gx >= M, "bad ld");
    if (B == 0 || M == 0) return FQSS_OK;
    const int64_t rows = (int64_t)B * C;
    if (aligned16(gz) && aligned16(gx) && ld_gz % 4 == 0 && ld_gx % 4 == 0)
        hipLaunchKernelGGL(k_dwconv_v4<true>, grid_rows(rows, M, 4), dim3(256), 0, (hipStream_t)stream, gz, w,
                           (const float*)nullptr, gx, rows, C, M, K, dil, pad, ld_gz, ld_gx);
    else
        hipLaunchKernelGGL(k_dwconv<true>, grid_rows(rows, M, 1), dim3(256), 0, (hipStream_t)stream, gz, w,
                           (const float*)nullptr, gx, rows, C, M, K, dil, pad, ld_gz, ld_gx);
    return launch_status("fqss_dwconv_bwd_x");
}

extern "C" int fqss_dwconv_bwd_w(const float* gz, const float* x, float* gw, int B, int C, int M, int K, int dil,
                                 int pad, int64_t ld_gz, int64_t ld_x, fqss_stream_t stream) {
    FQSS_REQUIRE(gz && x && gw, "null tensor");
    FQSS_REQUIRE(B >= 0 && C > 0 && M >= 0 && K > 0 && K <= kMaxTaps && dil > 0 && pad >= 0, "bad shape");
    FQSS_REQUIRE(ld_gz >= M && ld_x >= M && B <= 65535, "bad ld / batch");
    if (B == 0 || M == 0) return FQSS_OK;
    hipLaunchKernelGGL(k_dwconv_bwd_w, dim3((unsigned)C, (unsigned)B), dim3(256), 0, (hipStream_t)stream, gz, x, gw, C, M,
                       K, dil, pad, ld_gz, ld_x);
    return launch_status("fqss_dwconv_bwd_w");
}

// column slices (grid.z) of the per-row GroupNorm passes when rows alone leave the chip empty: ~1,024 workgroups, >= 4 passes each
static int gn_col_slices(int64_t row_wgs, int M, int vec) {
    if (row_wgs >= 512) return 1;
    int64_t zs = 1024 / (row_wgs > 0 ? row_wgs : 1);
    const int64_t zmax = cdiv(M, 256 * vec * 4);
    if (zs > zmax) zs = zmax;
    if (zs > 65535) zs = 65535;
    return zs < 1 ? 1 : (int)zs;
}

static int gn_fwd_impl(const char* who, const float* x, const float* gamma, const float* beta, float* z, float* mean_rstd, int B, int C, int M,
                       int64_t ld_x, int64_t ld_z, float eps, double* ws, const float* qmin, const float* qmax, fqss_stream_t stream,
                       uint8_t* yc = nullptr, int64_t ld_yc = 0);

extern "C" int fqss_gn_fwd(const float* x, const float* gamma, const float* beta, float* z, float* mean_rstd, int B,
                           int C, int M, int64_t ld_x, int64_t ld_z, float eps, double* ws, fqss_stream_t stream) {
    return gn_fwd_impl("fqss_gn_fwd", x, gamma, beta, z, mean_rstd, B, C, M, ld_x, ld_z, eps, ws, nullptr, nullptr, stream);
}

// GroupNormQ on a FLOAT input in the quantizing phase: y = fq(GroupNorm(1, C)(x)) from the statistics pass + ONE apply pass (the
// pre-quant value is not stored) -- fqss_gn_fwd followed by fqss_actq_fwd, value for value
extern "C" int fqss_gnq_fwd_f(const float* x, const float* gamma, const float* beta, float* y, uint8_t* yc, float* mean_rstd, int B, int C, int M,
                              int64_t ld_x, int64_t ld_y, int64_t ld_yc, float eps, double* ws, const float* qmin, const float* qmax,
                              fqss_stream_t stream) {
    FQSS_REQUIRE(qmin && qmax, "null range");
    FQSS_REQUIRE(yc == nullptr || (ld_yc >= M && ld_yc % 4 == 0 && (reinterpret_cast<uintptr_t>(yc) & 3u) == 0 && ld_yc >= ((M + 3) & ~3)),
                 "code rows: 4-B aligned, padded to a multiple of 4");
    return gn_fwd_impl("fqss_gnq_fwd_f", x, gamma, beta, y, mean_rstd, B, C, M, ld_x, ld_y, eps, ws, qmin, qmax, stream, yc, ld_yc);
}

static int gn_fwd_impl(const char* who, const float* x, const float* gamma, const float* beta, float* z, float* mean_rstd, int B, int C, int M,
                       int64_t ld_x, int64_t ld_z, float eps, double* ws, const float* qmin, const float* qmax, fqss_stream_t stream, uint8_t* yc,
                       int64_t ld_yc) {
    FQSS_REQUIRE(x && gamma && beta && z && mean_rstd && ws, "null tensor");
    FQSS_REQUIRE(B >= 0 && B <= 65535 && C > 0 && M > 0 && ld_x >= M && ld_z >= M, "bad shape");
    if (B == 0) return FQSS_OK;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(ws, 0, sizeof(double) * 2 * B, s) != hipSuccess) return launch_status(who);
    const bool vec_x = aligned16(x) && ld_x % 4 == 0;
    int nb = C < 64 ? C : 64;
    const int zs = gn_col_slices((int64_t)nb * B, M, vec_x ? 4 : 1);
    if (vec_x)
        hipLaunchKernelGGL(k_gn_stats<4>, dim3((unsigned)nb, (unsigned)B, (unsigned)zs), dim3(256), 0, s, x, C, M, ld_x, ws);
    else
        hipLaunchKernelGGL(k_gn_stats<1>, dim3((unsigned)nb, (unsigned)B, (unsigned)zs), dim3(256), 0, s, x, C, M, ld_x, ws);
    const int64_t rows = (int64_t)B * C;
    const bool v4 = vec_x && aligned16(z) && ld_z % 4 == 0;
    if (qmin != nullptr) {
        if (v4) hipLaunchKernelGGL((k_gn_apply<4, true>), grid_rows(rows, M, 4), dim3(256), 0, s, x, gamma, beta, z, mean_rstd, B, C, M, ld_x, ld_z, eps, ws, qmin, qmax, yc, ld_yc);
        else hipLaunchKernelGGL((k_gn_apply<1, true>), grid_rows(rows, M, 1), dim3(256), 0, s, x, gamma, beta, z, mean_rstd, B, C, M, ld_x, ld_z, eps, ws, qmin, qmax, yc, ld_yc);
    } else if (v4)
        hipLaunchKernelGGL((k_gn_apply<4, false>), grid_rows(rows, M, 4), dim3(256), 0, s, x, gamma, beta, z, mean_rstd, B, C, M,
                           ld_x, ld_z, eps, ws, nullptr, nullptr, nullptr, 0);
    else
        hipLaunchKernelGGL((k_gn_apply<1, false>), grid_rows(rows, M, 1), dim3(256), 0, s, x, gamma, beta, z, mean_rstd, B, C, M,
                           ld_x, ld_z, eps, ws, nullptr, nullptr, nullptr, 0);
    return launch_status(who);
}

static int gn_bwd_impl(const char* who, const float* gz, const float* x, const float* gamma, const float* beta, const float* mean_rstd, float* gx,
                       float* ggamma, float* gbeta, int B, int C, int M, int64_t ld_gz, int64_t ld_x, int64_t ld_gx, double* ws, const float* qmin,
                       const float* qmax, double* gacc, fqss_stream_t stream);

/* Forward-only GroupNorm(1, C) with what follows it fused into the apply pass (k_gn_tail; inference paths: the frozen teacher):
 * tail 1: y [B][C][M] = gelu(gn(x));  tail 2: y [B][C/2][M] = glu(gn(x)) * ls[c] + res (ls [C/2], res [B][C/2][ld_res]).
 * Rows 16-B aligned; ws: 2 * B doubles. */
extern "C" int fqss_gn_fwd_tail(const float* x, const float* gamma, const float* beta, float* y, int B, int C, int M, int64_t ld_x, int64_t ld_y,
                                float eps, double* ws, int tail, const float* ls, const float* res, int64_t ld_res, fqss_stream_t stream) {
    FQSS_REQUIRE(x && gamma && beta && y && ws, "null tensor");
    FQSS_REQUIRE(B >= 0 && B <= 65535 && C > 0 && M > 0 && ld_x >= M && ld_y >= M, "bad shape");
    FQSS_REQUIRE(tail == 1 || (tail == 2 && C % 2 == 0 && ls && res && ld_res >= M), "tail: 1 (GELU) or 2 (GLU * scale + residual; C even)");
    FQSS_REQUIRE(aligned16(x) && aligned16(y) && ld_x % 4 == 0 && ld_y % 4 == 0 && ld_x >= ((M + 3) & ~3) && ld_y >= ((M + 3) & ~3),
                 "x / y rows must be 16-B aligned and padded to 4");
    const int res_vec = (tail == 2 && aligned16(res) && ld_res % 4 == 0 && ld_res >= ((M + 3) & ~3)) ? 1 : 0;
    if (B == 0) return FQSS_OK;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(ws, 0, sizeof(double) * 2 * B, s) != hipSuccess) return launch_status("fqss_gn_fwd_tail");
    const int nb = C < 64 ? C : 64;
    const int zs = gn_col_slices((int64_t)nb * B, M, 4);
    hipLaunchKernelGGL(k_gn_stats<4>, dim3((unsigned)nb, (unsigned)B, (unsigned)zs), dim3(256), 0, s, x, C, M, ld_x, ws);
    const int64_t rows = (int64_t)B * (tail == 2 ? C / 2 : C);
    if (tail == 1)
        hipLaunchKernelGGL(k_gn_tail<1>, grid_rows(rows, M, 4), dim3(256), 0, s, x, gamma, beta, y, B, C, M, ld_x, ld_y, eps, ws, ls, res, ld_res, 0);
    else
        hipLaunchKernelGGL(k_gn_tail<2>, grid_rows(rows, M, 4), dim3(256), 0, s, x, gamma, beta, y, B, C, M, ld_x, ld_y, eps, ws, ls, res, ld_res, res_vec);
    return launch_status("fqss_gn_fwd_tail");
}


extern "C" int fqss_gn_bwd(const float* gz, const float* x, const float* gamma, const float* mean_rstd, float* gx,
                           float* ggamma, float* gbeta, int B, int C, int M, int64_t ld_gz, int64_t ld_x,
                           int64_t ld_gx, double* ws, fqss_stream_t stream) {
    return gn_bwd_impl("fqss_gn_bwd", gz, x, gamma, nullptr, mean_rstd, gx, ggamma, gbeta, B, C, M, ld_gz, ld_x, ld_gx, ws, nullptr, nullptr, nullptr,
                       stream);
}

// backward of fqss_gnq_fwd_f: g = dL/dy; the quantizer's STE runs on the pre-quant value recomputed from x in the two data passes of
// the GroupNorm backward (range partials to gacc, FQSS_GACC_SLOTS x 3 as fqss_actq_bwd) -- no fqss_actq_bwd pass, no stored z
extern "C" int fqss_gnq_bwd_f(const float* g, const float* x, const float* gamma, const float* beta, const float* mean_rstd, float* gx,
                              float* ggamma, float* gbeta, int B, int C, int M, int64_t ld_g, int64_t ld_x, int64_t ld_gx, double* ws,
                              const float* qmin, const float* qmax, double* gacc, fqss_stream_t stream) {
    FQSS_REQUIRE(beta && qmin && qmax && gacc, "null quantizer argument");
    return gn_bwd_impl("fqss_gnq_bwd_f", g, x, gamma, beta, mean_rstd, gx, ggamma, gbeta, B, C, M, ld_g, ld_x, ld_gx, ws, qmin, qmax, gacc, stream);
}

static int gn_bwd_impl(const char* who, const float* gz, const float* x, const float* gamma, const float* beta, const float* mean_rstd, float* gx,
                       float* ggamma, float* gbeta, int B, int C, int M, int64_t ld_gz, int64_t ld_x, int64_t ld_gx, double* ws, const float* qmin,
                       const float* qmax, double* gacc, fqss_stream_t stream) {
    FQSS_REQUIRE(gz && x && gamma && mean_rstd && gx && ggamma && gbeta && ws, "null tensor");
    FQSS_REQUIRE(B >= 0 && B <= 65535 && C > 0 && M > 0 && ld_gz >= M && ld_x >= M && ld_gx >= M, "bad shape");
    if (B == 0) return FQSS_OK;
    hipStream_t s = (hipStream_t)stream;
    const bool vec = FQSS_VEC_OK2(gz, ld_gz, x, ld_x);
    const int zs = gn_col_slices((int64_t)C * B, M, vec ? 4 : 1);
    if (zs > 1 && hipMemsetAsync(ws, 0, sizeof(double) * 2 * (size_t)B * C, s) != hipSuccess) return launch_status(who);
    const dim3 rgrid((unsigned)C, (unsigned)B, (unsigned)zs);
    if (qmin != nullptr) {
        if (vec) hipLaunchKernelGGL((k_gn_bwd_rows<4, true>), rgrid, dim3(256), 0, s, gz, x, C, M, ld_gz, ld_x, ws, gamma, beta, mean_rstd, qmin, qmax, gacc);
        else hipLaunchKernelGGL((k_gn_bwd_rows<1, true>), rgrid, dim3(256), 0, s, gz, x, C, M, ld_gz, ld_x, ws, gamma, beta, mean_rstd, qmin, qmax, gacc);
    } else if (vec)
        hipLaunchKernelGGL((k_gn_bwd_rows<4, false>), rgrid, dim3(256), 0, s, gz, x, C, M, ld_gz, ld_x, ws, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    else
        hipLaunchKernelGGL((k_gn_bwd_rows<1, false>), rgrid, dim3(256), 0, s, gz, x, C, M, ld_gz, ld_x, ws, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    const int nbs = B <= 16 ? 1 : (int)(cdiv(B, 16) > 64 ? 64 : cdiv(B, 16));      // sample slices of the parameter-gradient sums
    hipLaunchKernelGGL(k_gn_bwd_coef, dim3((unsigned)(B + cdiv(C, 256) * nbs)), dim3(256), 0, s, gamma, mean_rstd, B, C, M, ws,
                       ggamma, gbeta, nbs);
    const int64_t rows = (int64_t)B * C;
    const bool v4 = vec && aligned16(gx) && ld_gx % 4 == 0;
    if (qmin != nullptr) {
        if (v4) hipLaunchKernelGGL((k_gn_bwd_apply<4, true>), grid_rows(rows, M, 4), dim3(256), 0, s, gz, x, gamma, mean_rstd, gx, B, C, M, ld_gz, ld_x, ld_gx, ws, beta, qmin, qmax);
        else hipLaunchKernelGGL((k_gn_bwd_apply<1, true>), grid_rows(rows, M, 1), dim3(256), 0, s, gz, x, gamma, mean_rstd, gx, B, C, M, ld_gz, ld_x, ld_gx, ws, beta, qmin, qmax);
    } else if (v4)
        hipLaunchKernelGGL((k_gn_bwd_apply<4, false>), grid_rows(rows, M, 4), dim3(256), 0, s, gz, x, gamma, mean_rstd, gx, B, C,
                           M, ld_gz, ld_x, ld_gx, ws, nullptr, nullptr, nullptr);
    else
        hipLaunchKernelGGL((k_gn_bwd_apply<1, false>), grid_rows(rows, M, 1), dim3(256), 0, s, gz, x, gamma, mean_rstd, gx, B, C,
                           M, ld_gz, ld_x, ld_gx, ws, nullptr, nullptr, nullptr);
    return launch_status(who);
}
