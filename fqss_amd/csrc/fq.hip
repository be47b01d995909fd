// fq.hip -- K1/K1b/K2/K3: per-tensor activation fake-quant (+PReLU/ReLU), observer, STE backward,
// per-channel symmetric weight fake-quant.  HBM-bound streaming kernels: 16 B/lane vector loads
// when rows are 16-B aligned, wave shuffles + one atomic per block for the range reductions.
//
// Reference semantics restated (see include/fqss.h for the entry-point contract):
//   quantization/qat/qat_quant.py:125-147  linear_quantize
//   quantization/qat/qat_quant.py:227-242  GradientActivationFakeQuantize.forward
//   quantization/qat/qat_quant.py:372-381  GradientWeightFakeQuantize.forward
#include <cstdarg>
#include <cstdio>

#define FQSS_USES_GRAD_ADD   // the fp32 gradient atomics of this file go through grad_add (fqss_dev.h: FQSS_DETERMINISTIC=1)
#include "fqss_dev.h"

namespace fqss {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return FQSS_ELAUNCH;
    }
    return FQSS_OK;
}

// =============================================================================================
// activation epilogue forward
// =============================================================================================
// GELU (template): act = FQSS_ACT_GELU, the erf form of nn.GELU (the HTDemucs layers, hdemucsq.py:126-127) applied in front of the
// quantizer in the same pass -- the operations of k_unary_fwd / k_unary_bwd (csrc/dualpath.hip); the ReLU-family instantiations (every
// launch of the ConvTasNet path) are unchanged
__device__ __forceinline__ float gelu_fwd_f(float v) { return (0.5f * v) * (1.0f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
    return cdf + x * pdf;
}
// a 4-float group of a row with `left` columns remaining: one 16-B store, or -- the row's last, partial group -- only its live elements
// (nothing past the last column is written: the output may be a column block of a wider matrix)
template <int N>
__device__ __forceinline__ void store_group4(float* p, const float (&o)[N], int64_t left) {
    static_assert(N == 4, "16-B groups");
    if (left >= 4) {
        *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
        p[0] = o[0];
        if (left > 1) p[1] = o[1];
        if (left > 2) p[2] = o[2];
    }
}

// POSTRELU (template): act = FQSS_ACT_POST_RELU: no map in FRONT of the quantizer, a ReLU BEHIND it -- relu(fq(z)), the `F.relu` between LSTMQ
// and LinearQ of DPTNet's transformer layer (dptnetq.py:84-97) -- in the quantizer's pass each way
template <int VEC, bool GELU = false, bool POSTRELU = false>
__global__ __launch_bounds__(256) void k_actq_fwd(const float* __restrict__ z, float* __restrict__ out,
                                                   uint8_t* __restrict__ idx, int64_t rows, int64_t cols,
                                                   int64_t ld_z, int64_t ld_o, int64_t ld_i, int act,
                                                   const float* __restrict__ slope_p, int qmode,
                                                   const float* __restrict__ qmin,
                                                   const float* __restrict__ qmax, uint32_t* obs) {
    const float slope = (act == FQSS_ACT_PRELU) ? *slope_p : 0.0f;
    QRange r{0.0f, 1.0f, 1.0f};
    if (qmode == FQSS_Q_QUANT) r = load_qrange(qmin, qmax);
    float vmin = INFINITY, vmax = -INFINITY;
    const int64_t cstep = (int64_t)gridDim.x * 256 * VEC;
    for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
        const float* zr = z + row * ld_z;
        float* orow = out + row * ld_o;
        for (int64_t c0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC; c0 < cols; c0 += cstep) {
            float v[VEC];
            if constexpr (VEC == 4) {
                const float4 t = *reinterpret_cast<const float4*>(zr + c0);
                v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            } else {
                v[0] = zr[c0];
            }
            float o[VEC];
            unsigned int packed = 0;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float t = GELU ? gelu_fwd_f(v[j]) : (POSTRELU ? v[j] : act_apply(v[j], act, slope));
                if (qmode == FQSS_Q_QUANT) {
                    float c, u;
                    bool inr;
                    o[j] = fq_asym(t, r, c, u, inr);
                    if (idx != nullptr) {
                        if (VEC == 4 && (ld_i & 3) == 0) packed |= ((unsigned int)c & 0xFFu) << (8 * j);
                        else if (c0 + j < cols) idx[row * ld_i + c0 + j] = (uint8_t)c;
                    }
                } else {
                    o[j] = t;
                    if (qmode == FQSS_Q_OBSERVE && c0 + j < cols) {
                        vmin = fminf(vmin, t);
                        vmax = fmaxf(vmax, t);
                    }
                }
                if (POSTRELU) o[j] = o[j] > 0.0f ? o[j] : 0.0f;
            }
            if constexpr (VEC == 4) {
                if (out != nullptr) store_group4(orow + c0, o, cols - c0);
                if (qmode == FQSS_Q_QUANT && idx != nullptr && (ld_i & 3) == 0)
                    *reinterpret_cast<unsigned int*>(idx + row * ld_i + c0) = packed;
            } else {
                if (out != nullptr) orow[c0] = o[0];
            }
        }
    }
    if (qmode == FQSS_Q_OBSERVE) {
        vmin = wave_min(vmin);
        vmax = wave_max(vmax);
        if ((threadIdx.x & 63) == 0) {
            const uint32_t kmin = f2ord(vmin), kmax = f2ord(vmax);
            // most waves do not improve the running extremum: test with a relaxed load first
            if (kmin < __hip_atomic_load(&obs[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&obs[0], kmin);
            if (kmax > __hip_atomic_load(&obs[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&obs[1], kmax);
        }
    }
}

// k_actq_fwd (QUANT mode) for NARROW matrices [R][F], F <= 512 features per row: the kernel above gives a row to a whole workgroup, so
// a 64-wide matrix keeps 16 of its 256 lanes busy (the q / k / v / div quantizers of the attention layers work on [rows][64 | 256] column
// blocks: 44 us where 12 suffice).  Here a workgroup takes 64-feature slices x 16 rows per pass like k_actq_bwd_colbias.
__global__ __launch_bounds__(256) void k_actq_fwd_narrow(const float* __restrict__ z, float* __restrict__ out, uint8_t* __restrict__ idx,
                                                          int64_t R, int F, int64_t ld_z, int64_t ld_o, int64_t ld_i, int act,
                                                          const float* __restrict__ slope_p, const float* __restrict__ qmin,
                                                          const float* __restrict__ qmax, int64_t rows_per_part) {
    const float slope = (act == FQSS_ACT_PRELU) ? *slope_p : 0.0f;
    const QRange r = load_qrange(qmin, qmax);
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int f0 = blockIdx.x * 64 + tx * 4;
    const int64_t r_beg = (int64_t)blockIdx.y * rows_per_part, r_end = min(R, r_beg + rows_per_part);
    if (f0 >= F) return;
    for (int64_t row0 = r_beg + ty; row0 < r_end; row0 += 64) {
        float4 za[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) za[i] = *reinterpret_cast<const float4*>(z + min(row0 + 16 * i, r_end - 1) * ld_z + f0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t row = row0 + 16 * i;
            if (row < r_end) {
                const float zv[4] = {za[i].x, za[i].y, za[i].z, za[i].w};
                float o[4];
                unsigned int packed = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float c, u;
                    bool inr;
                    o[j] = fq_asym(act_apply(zv[j], act, slope), r, c, u, inr);
                    packed = pack_code(c, j, packed);
                }
                if (out != nullptr) *reinterpret_cast<float4*>(out + row * ld_o + f0) = make_float4(o[0], o[1], o[2], o[3]);
                if (idx != nullptr) *reinterpret_cast<unsigned int*>(idx + row * ld_i + f0) = packed;
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------
// fq(GLU(z)): nn.GLU(dim = 1) of a channel-first tensor [B][2C][M] in front of a quantizer (Conv1dNlQ / Conv2dNlQ with nl = GLU in the
// HTDemucs layers, hdemucsq.py:126-127, 303-347) in ONE pass each way: t = z[b][c] * sigmoid(z[b][C + c]), out = fq(t) -- the
// operations of k_glu_fwd / k_glu_bwd (csrc/dualpath.hip) followed by k_actq_fwd / k_actq_bwd, without the [B][C][M] round trip in
// between.  16-B aligned rows (float4 per lane; the tail of a row lies in its own padding).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gluq_fwd(const float* __restrict__ z, float* __restrict__ out, int64_t B, int64_t C, int64_t M,
                                                   int64_t ld_z, int64_t ld_o, int qmode, const float* __restrict__ qmin,
                                                   const float* __restrict__ qmax, uint32_t* obs) {
    QRange r{0.0f, 1.0f, 1.0f};
    if (qmode == FQSS_Q_QUANT) r = load_qrange(qmin, qmax);
    float vmin = INFINITY, vmax = -INFINITY;
    for (int64_t row = blockIdx.y; row < B * C; row += gridDim.y) {
        const int64_t b = row / C, c = row - b * C;
        const float* pa = z + (b * 2 * C + c) * ld_z;
        const float* pg = z + (b * 2 * C + C + c) * ld_z;
        float* po = out + row * ld_o;
        for (int64_t m0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; m0 < M; m0 += (int64_t)gridDim.x * 1024) {
            const float4 a4 = *reinterpret_cast<const float4*>(pa + m0), g4 = *reinterpret_cast<const float4*>(pg + m0);
            const float av[4] = {a4.x, a4.y, a4.z, a4.w}, gv[4] = {g4.x, g4.y, g4.z, g4.w};
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t = av[j] * (1.0f / (1.0f + expf(-gv[j])));
                if (qmode == FQSS_Q_QUANT) {
                    float cc, u;
                    bool inr;
                    o[j] = fq_asym(t, r, cc, u, inr);
                } else {
                    o[j] = t;
                    if (qmode == FQSS_Q_OBSERVE && m0 + j < M) {
                        vmin = fminf(vmin, t);
                        vmax = fmaxf(vmax, t);
                    }
                }
            }
            *reinterpret_cast<float4*>(po + m0) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    if (qmode == FQSS_Q_OBSERVE) {
        vmin = wave_min(vmin);
        vmax = wave_max(vmax);
        if ((threadIdx.x & 63) == 0) {
            const uint32_t kmin = f2ord(vmin), kmax = f2ord(vmax);
            if (kmin < __hip_atomic_load(&obs[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&obs[0], kmin);
            if (kmax > __hip_atomic_load(&obs[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&obs[1], kmax);
        }
    }
}

__global__ __launch_bounds__(256) void k_gluq_bwd(const float* __restrict__ z, const float* __restrict__ g, float* __restrict__ gz, int64_t B,
                                                   int64_t C, int64_t M, int64_t ld_z, int64_t ld_g, int64_t ld_gz, int qmode,
                                                   const float* __restrict__ qmin, const float* __restrict__ qmax, double* gacc) {
    __shared__ double red[2 * 4];
    QRange r{0.0f, 1.0f, 1.0f};
    if (qmode == FQSS_Q_QUANT) r = load_qrange(qmin, qmax);
    float p_du = 0.0f, p_out = 0.0f;
    for (int64_t row = blockIdx.y; row < B * C; row += gridDim.y) {
        const int64_t b = row / C, c = row - b * C;
        const int64_t ra = b * 2 * C + c, rg = ra + C;
        for (int64_t m0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; m0 < M; m0 += (int64_t)gridDim.x * 1024) {
            const float4 a4 = *reinterpret_cast<const float4*>(z + ra * ld_z + m0), g4 = *reinterpret_cast<const float4*>(z + rg * ld_z + m0);
            const float4 y4 = *reinterpret_cast<const float4*>(g + row * ld_g + m0);
            const float av[4] = {a4.x, a4.y, a4.z, a4.w}, gv[4] = {g4.x, g4.y, g4.z, g4.w}, yv[4] = {y4.x, y4.y, y4.z, y4.w};
            float oa[4], og[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool valid = m0 + j < M;
                const float sg = 1.0f / (1.0f + expf(-gv[j]));
                const float gj = valid ? yv[j] : 0.0f;
                float gt = gj;
                if (qmode == FQSS_Q_QUANT) {
                    float cc, u;
                    bool inr;
                    (void)fq_asym(av[j] * sg, r, cc, u, inr);
                    gt = inr ? div_by(gj * r.delta, r.delta, r.inv) : 0.0f;
                    p_du += valid ? gj * (inr ? (cc - u) : cc) : 0.0f;
                    p_out += (valid && !inr) ? gj : 0.0f;
                }
                oa[j] = gt * sg;
                og[j] = ((gt * av[j]) * (1.0f - sg)) * sg;
            }
            *reinterpret_cast<float4*>(gz + ra * ld_gz + m0) = make_float4(oa[0], oa[1], oa[2], oa[3]);
            *reinterpret_cast<float4*>(gz + rg * ld_gz + m0) = make_float4(og[0], og[1], og[2], og[3]);
        }
    }
    if (qmode == FQSS_Q_QUANT) {
        double v[2] = {(double)p_du, (double)p_out};
        block_sum<double, 2>(v, red);
        if (threadIdx.x == 0) {
            double* slot = gacc + 3 * ((int64_t)blockIdx.y * gridDim.x + blockIdx.x);
            const double dmax = v[0] / 255.0;
            slot[0] += v[1] - dmax;
            slot[1] += dmax;
        }
    }
}

__global__ void k_obs_reset(uint32_t* obs, int64_t n_pairs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_pairs) {
        obs[2 * i] = 0xFFFFFFFFu;
        obs[2 * i + 1] = 0u;
    }
}

__global__ void k_observer_ema(float* qmin, float* qmax, uint32_t* obs, float alpha, float one_minus_alpha) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float tmin = ord2f(obs[0]), tmax = ord2f(obs[1]);
        // self.alpha*self.min_range + (1-self.alpha)*tilde_range_min   (qat_quant.py:231-232)
        *qmin = alpha * (*qmin) + one_minus_alpha * tmin;
        *qmax = alpha * (*qmax) + one_minus_alpha * tmax;
        obs[0] = 0xFFFFFFFFu;
        obs[1] = 0u;
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void k_minmax(const float* __restrict__ x, int64_t rows, int64_t cols,
                                                 int64_t ld, uint32_t* obs) {
    float vmin = INFINITY, vmax = -INFINITY;
    const int64_t cstep = (int64_t)gridDim.x * 256 * VEC;
    for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
        const float* xr = x + row * ld;
        for (int64_t c0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC; c0 < cols; c0 += cstep) {
            if constexpr (VEC == 4) {
                const float4 t = *reinterpret_cast<const float4*>(xr + c0);
                const float v[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c0 + j < cols) {
                        vmin = fminf(vmin, v[j]);
                        vmax = fmaxf(vmax, v[j]);
                    }
            } else {
                vmin = fminf(vmin, xr[c0]);
                vmax = fmaxf(vmax, xr[c0]);
            }
        }
    }
    vmin = wave_min(vmin);
    vmax = wave_max(vmax);
    // one atomic pair per WORKGROUP (same-address atomics cost ~12 ns each: per wave they were 25 of the 45 us the splitter's 1-MB
    // max scan took at the head of every step)
    __shared__ float wmn[4], wmx[4];
    if ((threadIdx.x & 63) == 0) {
        wmn[threadIdx.x >> 6] = vmin;
        wmx[threadIdx.x >> 6] = vmax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        vmin = fminf(fminf(wmn[0], wmn[1]), fminf(wmn[2], wmn[3]));
        vmax = fmaxf(fmaxf(wmx[0], wmx[1]), fmaxf(wmx[2], wmx[3]));
        const uint32_t kmin = f2ord(vmin), kmax = f2ord(vmax);
        if (kmin < __hip_atomic_load(&obs[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&obs[0], kmin);
        if (kmax > __hip_atomic_load(&obs[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&obs[1], kmax);
    }
}

// =============================================================================================
// activation epilogue backward (STE + range gradients + PReLU/ReLU + optional bias row-sums)
//   autograd of qat_quant.py:139-146 restated op by op:
//     g_C = g*delta ; g_u = g_C*m ; g_t = g_u/delta
//     d/dmax = sum g*(c - m*u)/255 ; d/dmin = sum g*(1-m) - d/dmax
// =============================================================================================
// Reductions: NO same-address atomics (4096 blocks x 3 fp64 atomics on one address cost ~100-160 us,
// 3-5x the streaming time of the kernel).  Every block stores its three fp64 partials to its own slot
// of `gacc` ([kGaccSlots][3], all-zero on entry); fqss_gacc_flush sums the slots in a fixed order
// (=> deterministic range/slope gradients) and re-zeroes them.
// Rows are walked channel-major (c outer, batch inner) so that the bias row-sum of a channel is ONE
// block reduction + one fp32 atomic per (column chunk, channel).
constexpr int kGaccSlots = FQSS_GACC_SLOTS;

template <int VEC, bool BIAS, bool GELU = false, bool POSTRELU = false>
__global__ __launch_bounds__(256) void k_actq_bwd(const float* __restrict__ z, const float* __restrict__ g,
                                                   float* __restrict__ gz, int64_t rows, int64_t cols,
                                                   int64_t ld_z, int64_t ld_g, int64_t ld_gz, int act,
                                                   const float* __restrict__ slope_p, int qmode,
                                                   const float* __restrict__ qmin,
                                                   const float* __restrict__ qmax, double* gacc,
                                                   float* gbias, int64_t C) {
    __shared__ double red[3 * 4];
    __shared__ float redf[4];
    const float slope = (act == FQSS_ACT_PRELU) ? *slope_p : 0.0f;
    QRange r{0.0f, 1.0f, 1.0f};
    if (qmode == FQSS_Q_QUANT) r = load_qrange(qmin, qmax);
    float p_du = 0.0f;    // sum g*(c - m*u)
    float p_out = 0.0f;   // sum g*(1-m)
    float p_slope = 0.0f; // sum [z<=0] z*g_t
    const int64_t cstep = (int64_t)gridDim.x * 256 * VEC;
    const int64_t nb = rows / C;  // batch entries per channel
    for (int64_t ch = blockIdx.y; ch < C; ch += gridDim.y) {
        float p_bias = 0.0f;
        // one VEC-wide group: STE, activation backward, partial sums; returns the output group
        auto group = [&](const float (&zv)[VEC], const float (&gv)[VEC], int64_t c0, float (&o)[VEC]) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const bool valid = (c0 + j < cols);
                const float gj = valid ? gv[j] : 0.0f;
                const float t = GELU ? gelu_fwd_f(zv[j]) : (POSTRELU ? zv[j] : act_apply(zv[j], act, slope));
                float gt = gj;
                if (qmode == FQSS_Q_QUANT) {
                    float c, u;
                    bool inr;
                    const float yq = fq_asym(t, r, c, u, inr);
                    if (POSTRELU && !(yq > 0.0f)) gt = 0.0f;
                    const float gin = POSTRELU ? gt : gj;       // (POSTRELU: the gradient behind the ReLU's mask)
                    gt = inr ? div_by(gin * r.delta, r.delta, r.inv) : 0.0f;
                    // selects, not a branch around the sums (a masked-out position may hold anything, so its term is dropped, not
                    // multiplied by 0): the branchy form of these sums in k_mulq_bwd was right alone and off by one term in a few
                    // lanes per launch next to a second stream (docs/history/DESIGN_rounds_1-5.md 9); every such sum is written branch-free since
                    p_du += valid ? gin * (inr ? (c - u) : c) : 0.0f;
                    p_out += (valid && !inr) ? gin : 0.0f;
                } else if (POSTRELU && !(t > 0.0f)) {
                    gt = 0.0f;
                }
                float gzj = GELU ? gt * gelu_grad_f(zv[j]) : (POSTRELU ? gt : act_bwd(zv[j], gt, act, slope, valid, p_slope));
                o[j] = gzj;
                if (BIAS) p_bias += valid ? gzj : 0.0f;
            }
        };
        const int64_t c_first = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC;
        if constexpr (VEC == 4) if (cstep >= cols) {
            // every thread owns ONE column group of each of the channel's nb rows: the loads of 4 rows are issued
            // before the first is consumed (the serial loop exposed one HBM round trip per batch entry)
            if (c_first < cols) {
                // (gridDim.z > 1: the channel's nb rows are dealt to the z-slices in groups of four -- a [N][C][<= 1024] tensor with
                //  N in the hundreds (DConv on the rows of a spectrogram, hdemucsq.py:72-162) ran C workgroups of N / 4 serial passes)
                for (int64_t bi0 = 4 * (int64_t)blockIdx.z; bi0 < nb; bi0 += 4 * (int64_t)gridDim.z) {
                    float4 za[4], ga[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (bi0 + i < nb) {
                            const int64_t row = (bi0 + i) * C + ch;
                            za[i] = *reinterpret_cast<const float4*>(z + row * ld_z + c_first);
                            ga[i] = *reinterpret_cast<const float4*>(g + row * ld_g + c_first);
                        }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (bi0 + i < nb) {
                            const int64_t row = (bi0 + i) * C + ch;
                            const float zv[VEC] = {za[i].x, za[i].y, za[i].z, za[i].w};
                            const float gv[VEC] = {ga[i].x, ga[i].y, ga[i].z, ga[i].w};
                            float o[VEC];
                            group(zv, gv, c_first, o);
                            store_group4(gz + row * ld_gz + c_first, o, cols - c_first);
                        }
                }
            }
        }
        if (VEC != 4 || cstep < cols) {
            for (int64_t bi = blockIdx.z; bi < nb; bi += gridDim.z) {
                const int64_t row = bi * C + ch;
                const float* zr = z + row * ld_z;
                const float* gr = g + row * ld_g;
                float* or_ = gz + row * ld_gz;
                for (int64_t c0 = c_first; c0 < cols; c0 += cstep) {
                    float zv[VEC], gv[VEC], o[VEC];
                    if constexpr (VEC == 4) {
                        const float4 a = *reinterpret_cast<const float4*>(zr + c0);
                        const float4 b = *reinterpret_cast<const float4*>(gr + c0);
                        zv[0] = a.x; zv[1] = a.y; zv[2] = a.z; zv[3] = a.w;
                        gv[0] = b.x; gv[1] = b.y; gv[2] = b.z; gv[3] = b.w;
                    } else {
                        zv[0] = zr[c0];
                        gv[0] = gr[c0];
                    }
                    group(zv, gv, c0, o);
                    if constexpr (VEC == 4) {
                        store_group4(or_ + c0, o, cols - c0);
                    } else {
                        or_[c0] = o[0];
                    }
                }
            }
        }
        if constexpr (BIAS) {
            float pb[1] = {p_bias};
            block_sum<float, 1>(pb, redf);
            if (threadIdx.x == 0) grad_add(&gbias[ch], pb[0]);
        }
    }
    if (qmode == FQSS_Q_QUANT || act == FQSS_ACT_PRELU) {
        double v[3] = {(double)p_du, (double)p_out, (double)p_slope};
        block_sum<double, 3>(v, red);
        if (threadIdx.x == 0) {
            double* slot = gacc + 3 * (((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
            const double dmax = v[0] / 255.0;
            slot[0] += (qmode == FQSS_Q_QUANT) ? v[1] - dmax : 0.0;
            slot[1] += (qmode == FQSS_Q_QUANT) ? dmax : 0.0;
            slot[2] += v[2];
        }
    }
}

// k_actq_bwd for the output of a ROW linear (z = x W^T + b as [R][F], features contiguous): also accumulates the bias gradient
// gbias[f] += sum_r gz[r][f] -- the column sums a separate fqss_colsum pass over gz produced (dual-path models: 85 / 129 launches per
// step).  A workgroup owns a 64-feature slice (16 threads x 4 features = one 256-B piece of every row) and one part of the rows:
// a thread keeps its 4 features for all its rows, so the sums live in registers; the 16 row lanes meet in LDS, then ONE global atomic
// per feature and workgroup.  The grid bounds how many workgroups share a feature (<= 256: same-address atomics serialise -- a flat
// [rows'][16384] view with 2,048 workgroups per feature cost 25 us per launch in atomics alone).
__global__ __launch_bounds__(256) void k_actq_bwd_colbias(const float* __restrict__ z, const float* __restrict__ g, float* __restrict__ gz,
                                                           int64_t R, int F, int64_t ld_z, int64_t ld_g, int64_t ld_gz, int act,
                                                           const float* __restrict__ slope_p, int qmode, const float* __restrict__ qmin,
                                                           const float* __restrict__ qmax, double* gacc, float* gbias,
                                                           int64_t rows_per_part, const float* __restrict__ qmin2 = nullptr,
                                                           const float* __restrict__ qmax2 = nullptr, double* gacc2 = nullptr) {
    __shared__ double red[3 * 4];
    __shared__ float cb[16][64];
    const float slope = (act == FQSS_ACT_PRELU) ? *slope_p : 0.0f;
    const bool post = act == FQSS_ACT_POST_RELU;
    const bool two = act == FQSS_ACT_RELU_Q;         // y = fq2(relu(fq(z))): NlQ(ReLU) behind the linear's own quantizer (host: qmode QUANT)
    QRange r{0.0f, 1.0f, 1.0f}, r2{0.0f, 1.0f, 1.0f};
    if (qmode == FQSS_Q_QUANT) r = load_qrange(qmin, qmax);
    if (two) r2 = load_qrange(qmin2, qmax2);
    float p_du = 0.0f, p_out = 0.0f, p_slope = 0.0f, p2_du = 0.0f, p2_out = 0.0f;
    float pb[4] = {0.f, 0.f, 0.f, 0.f};
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int f0 = blockIdx.x * 64 + tx * 4;                       // host: F % 4 == 0
    const int64_t r_beg = (int64_t)blockIdx.y * rows_per_part, r_end = min(R, r_beg + rows_per_part);
    if (f0 < F) {
        for (int64_t row0 = r_beg + ty; row0 < r_end; row0 += 64) {
            float4 za[4], ga[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t row = min(row0 + 16 * i, r_end - 1);     // clamped: loads stay unconditional, extra rows are masked below
                za[i] = *reinterpret_cast<const float4*>(z + row * ld_z + f0);
                ga[i] = *reinterpret_cast<const float4*>(g + row * ld_g + f0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t row = row0 + 16 * i;
                if (row < r_end) {
                    const float zv[4] = {za[i].x, za[i].y, za[i].z, za[i].w};
                    const float gv[4] = {ga[i].x, ga[i].y, ga[i].z, ga[i].w};
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float gj = gv[j];
                        if (two) {
                            // forward recomputed: y1 = fq(z), t2 = relu(y1), y2 = fq2(t2); backward in the order of the two modules:
                            // NlQ (fqss_actq_bwd, act = ReLU, input y1), then the linear's quantizer (this kernel's plain form)
                            float c1, u1, c2, u2;
                            bool in1, in2;
                            const float y1 = fq_asym(zv[j], r, c1, u1, in1);
                            (void)fq_asym(act_apply(y1, FQSS_ACT_RELU, 0.0f), r2, c2, u2, in2);
                            const float gt2 = in2 ? div_by(gj * r2.delta, r2.delta, r2.inv) : 0.0f;
                            p2_du += gj * (in2 ? (c2 - u2) : c2);
                            p2_out += in2 ? 0.0f : gj;
                            float unused = 0.0f;
                            const float g1 = act_bwd(y1, gt2, FQSS_ACT_RELU, 0.0f, true, unused);
                            o[j] = in1 ? div_by(g1 * r.delta, r.delta, r.inv) : 0.0f;
                            p_du += g1 * (in1 ? (c1 - u1) : c1);
                            p_out += in1 ? 0.0f : g1;
                            pb[j] += o[j];
                            continue;
                        }
                        const float t = post ? zv[j] : act_apply(zv[j], act, slope);
                        float gt = gj;
                        if (qmode == FQSS_Q_QUANT) {
                            float c, u;
                            bool inr;
                            const float yq = fq_asym(t, r, c, u, inr);
                            if (post && !(yq > 0.0f)) gj = 0.0f;          // FQSS_ACT_POST_RELU: the ReLU sits BEHIND the quantizer
                            gt = inr ? div_by(gj * r.delta, r.delta, r.inv) : 0.0f;
                            p_du += gj * (inr ? (c - u) : c);
                            p_out += inr ? 0.0f : gj;
                        } else if (post && !(t > 0.0f)) {
                            gt = 0.0f;
                        }
                        o[j] = post ? gt : act_bwd(zv[j], gt, act, slope, true, p_slope);
                        pb[j] += o[j];
                    }
                    *reinterpret_cast<float4*>(gz + row * ld_gz + f0) = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
        }
    }
    if (gbias != nullptr) {      // (workgroup-uniform)
#pragma unroll
        for (int j = 0; j < 4; ++j) cb[ty][tx * 4 + j] = pb[j];
        __syncthreads();
        if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < F) {
            float sum = 0.0f;
#pragma unroll
            for (int k = 0; k < 16; ++k) sum += cb[k][threadIdx.x];
            grad_add(&gbias[blockIdx.x * 64 + threadIdx.x], sum);
        }
    }
    if (qmode == FQSS_Q_QUANT || act == FQSS_ACT_PRELU) {
        double v[3] = {(double)p_du, (double)p_out, (double)p_slope};
        block_sum<double, 3>(v, red);
        if (threadIdx.x == 0) {
            double* slot = gacc + 3 * ((int64_t)blockIdx.y * gridDim.x + blockIdx.x);
            const double dmax = v[0] / 255.0;
            slot[0] += (qmode == FQSS_Q_QUANT) ? v[1] - dmax : 0.0;
            slot[1] += (qmode == FQSS_Q_QUANT) ? dmax : 0.0;
            slot[2] += v[2];
        }
    }
    if (two) {      // (workgroup-uniform) NlQ's range partials into ITS slots
        __syncthreads();
        double v[3] = {(double)p2_du, (double)p2_out, 0.0};
        block_sum<double, 3>(v, red);
        if (threadIdx.x == 0) {
            double* slot = gacc2 + 3 * ((int64_t)blockIdx.y * gridDim.x + blockIdx.x);
            const double dmax = v[0] / 255.0;
            slot[0] += v[1] - dmax;
            slot[1] += dmax;
        }
    }
}

// =============================================================================================
// per-channel symmetric weight quantizer; tensor layout [outer][C][inner]
// =============================================================================================
__device__ __forceinline__ float wq_delta(float lo, float hi) {
    const float a = fmaxf(fabsf(lo), fabsf(hi));
    return (2.0f * a) / 255.0f;
}

__global__ __launch_bounds__(256) void k_wq_observe(const float* __restrict__ w, int64_t outer, int64_t C,
                                                     int64_t inner, float* qmin, float* qmax) {
    __shared__ float smin[4], smax[4];
    const int64_t c = blockIdx.x;
    float vmin = INFINITY, vmax = -INFINITY;
    const int64_t n = outer * inner;
    for (int64_t e = threadIdx.x; e < n; e += blockDim.x) {
        const int64_t o = e / inner, i = e - o * inner;
        const float v = w[(o * C + c) * inner + i];
        vmin = fminf(vmin, v);
        vmax = fmaxf(vmax, v);
    }
    vmin = wave_min(vmin);
    vmax = wave_max(vmax);
    if ((threadIdx.x & 63) == 0) {
        smin[threadIdx.x >> 6] = vmin;
        smax[threadIdx.x >> 6] = vmax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < (int)(blockDim.x >> 6); ++k) {
            vmin = fminf(vmin, smin[k]);
            vmax = fmaxf(vmax, smax[k]);
        }
        qmin[c] = vmin;
        qmax[c] = vmax;
    }
}

__global__ __launch_bounds__(256) void k_wq_fwd(const float* __restrict__ w, float* __restrict__ wq,
                                                 int8_t* __restrict__ idx, int64_t n, int64_t C, int64_t inner,
                                                 const float* __restrict__ qmin, const float* __restrict__ qmax) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = (e / inner) % C;
        const float delta = wq_delta(qmin[c], qmax[c]);
        const float X = rintf(w[e] / delta);
        const float q = fminf(fmaxf(X, -128.0f), 127.0f);
        wq[e] = delta * q;
        if (idx != nullptr) idx[e] = (int8_t)q;
    }
}

__global__ __launch_bounds__(256) void k_wq_bwd(const float* __restrict__ w, const float* __restrict__ g,
                                                 float* __restrict__ gw, float* gmin, float* gmax, int64_t outer,
                                                 int64_t C, int64_t inner, const float* __restrict__ qmin,
                                                 const float* __restrict__ qmax, int accumulate) {
    __shared__ double red[4];
    const int64_t c = blockIdx.x;
    const float lo = qmin[c], hi = qmax[c];
    const float delta = wq_delta(lo, hi);
    const int64_t n = outer * inner;
    float p = 0.0f;
    for (int64_t e = threadIdx.x; e < n; e += blockDim.x) {
        const int64_t o = e / inner, i = e - o * inner;
        const int64_t k = (o * C + c) * inner + i;
        const float u = w[k] / delta;
        const float X = rintf(u);
        const bool inr = (X >= -128.0f) && (X <= 127.0f);
        const float q = fminf(fmaxf(X, -128.0f), 127.0f);
        const float gk = g[k];
        const float gwk = inr ? (gk * delta) / delta : 0.0f;
        gw[k] = accumulate ? gw[k] + gwk : gwk;
        p += gk * (inr ? (q - u) : q);
    }
    double v[1] = {(double)p};
    block_sum<double, 1>(v, red);
    if (threadIdx.x == 0) {
        const double D = v[0] * (2.0 / 255.0);
        const float al = fabsf(lo), ah = fabsf(hi);
        // torch.maximum routes the gradient to the larger operand, 1/2-1/2 on ties; |x|' = sign(x)
        const double wl = al > ah ? 1.0 : (al == ah ? 0.5 : 0.0);
        const double wh = ah > al ? 1.0 : (al == ah ? 0.5 : 0.0);
        const double sl = lo > 0.0f ? 1.0 : (lo < 0.0f ? -1.0 : 0.0);
        const double sh = hi > 0.0f ? 1.0 : (hi < 0.0f ? -1.0 : 0.0);
        const float dmin = (float)(D * wl * sl), dmax = (float)(D * wh * sh);
        gmin[c] = accumulate ? gmin[c] + dmin : dmin;
        gmax[c] = accumulate ? gmax[c] + dmax : dmax;
    }
}

__global__ __launch_bounds__(256) void k_gacc_flush(double* gacc, float* gmin, float* gmax, float* gslope) {
    __shared__ double red[3 * 4];
    double v[3] = {0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < kGaccSlots; i += 256) {   // fixed order => deterministic sums
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            v[k] += gacc[3 * i + k];
            gacc[3 * i + k] = 0.0;
        }
    }
    block_sum<double, 3>(v, red);
    if (threadIdx.x == 0) {
        if (gmin) *gmin += (float)v[0];
        if (gmax) *gmax += (float)v[1];
        if (gslope) *gslope += (float)v[2];
    }
}

// brute-force self test: count elements where div_by differs from the IEEE division (bitwise)
__global__ void k_selftest_div(const float* __restrict__ a, int64_t n, float b, unsigned long long* mism) {
    const float y = 1.0f / b;
    unsigned long long m = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float q0 = a[i] / b, q1 = div_by(a[i], b, y);
        m += (__float_as_uint(q0) != __float_as_uint(q1)) ? 1ull : 0ull;
    }
    if (m) atomicAdd(mism, m);
}

}  // namespace fqss

// =============================================================================================
// C ABI
// =============================================================================================
using namespace fqss;

extern "C" int fqss_version(void) { return FQSS_VERSION; }

extern "C" int fqss_selftest_div(const float* a, int64_t n, float b, uint64_t* mismatches, fqss_stream_t stream) {
    if (n == 0) return FQSS_OK;   // empty input: nothing to do (a 0-element tensor has a null data pointer)
    FQSS_REQUIRE(a && mismatches && n >= 0, "bad args");
    if (n == 0) return FQSS_OK;
    hipLaunchKernelGGL(k_selftest_div, dim3(2048), dim3(256), 0, (hipStream_t)stream, a, n, b, (unsigned long long*)mismatches);
    return launch_status("fqss_selftest_div");
}
extern "C" const char* fqss_last_error(void) { return fqss::g_err; }

extern "C" int fqss_actq_fwd(const float* z, float* out, uint8_t* idx, int64_t rows, int64_t cols, int64_t ld_z,
                             int64_t ld_out, int64_t ld_idx, int act, const float* slope, int qmode, const float* qmin,
                             const float* qmax, uint32_t* obs_ws, fqss_stream_t stream) {
    if (rows == 0 || cols == 0) return FQSS_OK;   // empty input: nothing to do (a 0-element tensor has a null data pointer)
    FQSS_REQUIRE(z && (out || (idx && qmode == FQSS_Q_QUANT)), "null tensor");
    FQSS_REQUIRE(!idx || ld_idx >= cols, "bad ld_idx");
    if (!out) ld_out = ld_z;
    FQSS_REQUIRE(!idx || (ld_idx & 3) != 0 || (reinterpret_cast<uintptr_t>(idx) & 3u) == 0, "idx rows must be 4-B aligned");
    FQSS_REQUIRE(rows >= 0 && cols >= 0 && ld_z >= cols && ld_out >= cols, "bad shape");
    FQSS_REQUIRE(act >= 0 && act <= FQSS_ACT_POST_RELU && qmode >= 0 && qmode <= 2, "bad act/qmode");
    FQSS_REQUIRE(act != FQSS_ACT_PRELU || slope, "PReLU needs a slope");
    FQSS_REQUIRE(qmode != FQSS_Q_QUANT || (qmin && qmax), "QUANT needs ranges");
    FQSS_REQUIRE(qmode != FQSS_Q_OBSERVE || obs_ws, "OBSERVE needs obs_ws");
    if (rows == 0 || cols == 0) return FQSS_OK;
    // The 16-B path reads whole float4 groups (the last group of a row stays inside its stride) and stores a row's last, partial group
    // element by element (store_group4: the output may be a column block of a wider matrix).  Round 6; before, "fewer than 4 floats of
    // row padding" was required and every activation padded to 64 B by more -- [B, C, 110250] -> 110256 -- took the one-element path.
    // (The packed codes of a partial group go out as one word: code rows are padded by their owner, kernels.empty_codes.)
    bool vec = aligned16(z) && (!out || aligned16(out)) && (ld_z % 4 == 0) && (ld_out % 4 == 0);
#ifdef FQSS_OLD_TAIL_RULE      // (A/B builds only: `make variant SRC=fq NAME=oldtail DEFS=-DFQSS_OLD_TAIL_RULE`)
    vec = vec && ((cols % 4 == 0) || (out == nullptr) || (ld_out - cols < 4));
#endif
    hipStream_t s = (hipStream_t)stream;
    if (vec && act < FQSS_ACT_GELU && qmode == FQSS_Q_QUANT && cols % 4 == 0 && cols <= 512 && rows >= 1024 &&
        (!idx || ((ld_idx & 3) == 0 && (reinterpret_cast<uintptr_t>(idx) & 3u) == 0))) {
        // narrow matrix: row-tiled kernel (64-feature slices x row parts), every lane busy
        const int64_t slices = cdiv(cols, 64);
        int64_t parts = 2048 / slices;
        if (parts > cdiv(rows, 64)) parts = cdiv(rows, 64);
        const int64_t rows_per_part = cdiv(rows, parts);
        parts = cdiv(rows, rows_per_part);
        hipLaunchKernelGGL(k_actq_fwd_narrow, dim3((unsigned)slices, (unsigned)parts), dim3(256), 0, s, z, out, idx, rows, (int)cols, ld_z,
                           ld_out, ld_idx, act, slope, qmin, qmax, rows_per_part);
        return launch_status("fqss_actq_fwd");
    }
    if (act == FQSS_ACT_POST_RELU) {
        if (vec)
            hipLaunchKernelGGL((k_actq_fwd<4, false, true>), grid_rows(rows, cols, 4), dim3(256), 0, s, z, out, idx, rows, cols, ld_z, ld_out, ld_idx,
                               act, slope, qmode, qmin, qmax, obs_ws);
        else
            hipLaunchKernelGGL((k_actq_fwd<1, false, true>), grid_rows(rows, cols, 1), dim3(256), 0, s, z, out, idx, rows, cols, ld_z, ld_out, ld_idx,
                               act, slope, qmode, qmin, qmax, obs_ws);
    } else if (act == FQSS_ACT_GELU) {
        if (vec)
            hipLaunchKernelGGL((k_actq_fwd<4, true>), grid_rows(rows, cols, 4), dim3(256), 0, s, z, out, idx, rows, cols, ld_z, ld_out, ld_idx, act,
                               slope, qmode, qmin, qmax, obs_ws);
        else
            hipLaunchKernelGGL((k_actq_fwd<1, true>), grid_rows(rows, cols, 1), dim3(256), 0, s, z, out, idx, rows, cols, ld_z, ld_out, ld_idx, act,
                               slope, qmode, qmin, qmax, obs_ws);
    } else if (vec) {
        hipLaunchKernelGGL(k_actq_fwd<4>, grid_rows(rows, cols, 4), dim3(256), 0, s, z, out, idx, rows, cols, ld_z,
                           ld_out, ld_idx, act, slope, qmode, qmin, qmax, obs_ws);
    } else {
        hipLaunchKernelGGL(k_actq_fwd<1>, grid_rows(rows, cols, 1), dim3(256), 0, s, z, out, idx, rows, cols, ld_z,
                           ld_out, ld_idx, act, slope, qmode, qmin, qmax, obs_ws);
    }
    return launch_status("fqss_actq_fwd");
}

extern "C" int fqss_obs_reset(uint32_t* obs_ws, int64_t n_pairs, fqss_stream_t stream) {
    if (n_pairs == 0) return FQSS_OK;   // empty input: nothing to do (a 0-element tensor has a null data pointer)
    FQSS_REQUIRE(obs_ws && n_pairs >= 0, "bad args");
    if (n_pairs == 0) return FQSS_OK;
    hipLaunchKernelGGL(k_obs_reset, dim3((unsigned)cdiv(n_pairs, 256)), dim3(256), 0, (hipStream_t)stream, obs_ws,
                       n_pairs);
    return launch_status("fqss_obs_reset");
}

extern "C" int fqss_observer_ema(float* qmin, float* qmax, uint32_t* obs_ws, double alpha, fqss_stream_t stream) {
    FQSS_REQUIRE(qmin && qmax && obs_ws, "null pointer");
    // python evaluates alpha and (1 - alpha) in double; ATen casts each scalar to fp32 and multiplies
    // in fp32 (qat_quant.py:231): 1-0.9 = 0.09999999999999998 -> 0.1f, NOT 1-(float)0.9
    const float a = (float)alpha, oma = (float)(1.0 - alpha);
    hipLaunchKernelGGL(k_observer_ema, dim3(1), dim3(64), 0, (hipStream_t)stream, qmin, qmax, obs_ws, a, oma);
    return launch_status("fqss_observer_ema");
}

extern "C" int fqss_minmax(const float* x, int64_t rows, int64_t cols, int64_t ld, uint32_t* obs_ws,
                           fqss_stream_t stream) {
    if (rows == 0 || cols == 0) return FQSS_OK;   // empty input: nothing to do (a 0-element tensor has a null data pointer)
    FQSS_REQUIRE(x && obs_ws && rows >= 0 && cols >= 0 && ld >= cols, "bad args");
    if (rows == 0 || cols == 0) return FQSS_OK;
    const bool vec = aligned16(x) && (ld % 4 == 0);
    if (vec)
        hipLaunchKernelGGL(k_minmax<4>, grid_rows(rows, cols, 4, 1024), dim3(256), 0, (hipStream_t)stream, x, rows,
                           cols, ld, obs_ws);
    else
        hipLaunchKernelGGL(k_minmax<1>, grid_rows(rows, cols, 1, 1024), dim3(256), 0, (hipStream_t)stream, x, rows,
                           cols, ld, obs_ws);
    return launch_status("fqss_minmax");
}

extern "C" int fqss_actq_bwd(const float* z, const float* g, float* gz, int64_t rows, int64_t cols, int64_t ld_z,
                             int64_t ld_g, int64_t ld_gz, int act, const float* slope, int qmode, const float* qmin,
                             const float* qmax, double* gacc, float* gbias, int64_t C, fqss_stream_t stream) {
    if (rows == 0 || cols == 0) return FQSS_OK;   // empty input: nothing to do (a 0-element tensor has a null data pointer)
    FQSS_REQUIRE(z && g && gz, "null tensor");
    FQSS_REQUIRE(rows >= 0 && cols >= 0 && ld_z >= cols && ld_g >= cols && ld_gz >= cols, "bad shape");
    FQSS_REQUIRE(act >= 0 && act <= FQSS_ACT_POST_RELU && qmode >= 0 && qmode <= 2, "bad act/qmode");
    FQSS_REQUIRE(act < FQSS_ACT_GELU || !gbias, "GELU / POST_RELU: no bias-gradient form");
    FQSS_REQUIRE(act != FQSS_ACT_PRELU || slope, "PReLU needs a slope");
    FQSS_REQUIRE(qmode != FQSS_Q_QUANT || (qmin && qmax), "QUANT needs ranges");
    FQSS_REQUIRE((qmode != FQSS_Q_QUANT && act != FQSS_ACT_PRELU) || gacc, "range/slope grads need gacc");
    FQSS_REQUIRE(!gbias || (C > 0 && rows % C == 0), "gbias needs C dividing rows");
    if (rows == 0 || cols == 0) return FQSS_OK;
    if (!gbias && act < FQSS_ACT_GELU && cols % 4 == 0 && cols <= 512 && rows >= 1024 && aligned16(z) && aligned16(g) && aligned16(gz) && ld_z % 4 == 0 &&
        ld_g % 4 == 0 && ld_gz % 4 == 0)
        // narrow matrix (see k_actq_fwd_narrow): the row-tiled kernel of fqss_actq_bwd_colbias, without the bias sums
        return fqss_actq_bwd_colbias(z, g, gz, rows, (int)cols, ld_z, ld_g, ld_gz, act, slope, qmode, qmin, qmax, gacc, nullptr, stream);
    if (!gbias || C <= 0) C = rows;   // no channel semantics: every row is its own "channel"
    // (16-B groups whenever the rows are 16-B aligned: the loads of a row's last group stay inside its stride, the store of a partial
    // group writes its live elements only -- round 6; the rule before, "fewer than 4 floats of padding", sent every activation whose
    // rows are padded to 64 B by more than that -- [B, C, 110250] -> 110256 -- down the one-element path)
    bool vec = aligned16(z) && aligned16(g) && aligned16(gz) && (ld_z % 4 == 0) && (ld_g % 4 == 0) && (ld_gz % 4 == 0);
#ifdef FQSS_OLD_TAIL_RULE
    vec = vec && ((cols % 4 == 0) || (ld_gz - cols < 4));
#endif
    hipStream_t s = (hipStream_t)stream;
    const int v = vec ? 4 : 1;
    int64_t gx = cdiv(cols, 256 * (int64_t)v);
    if (gx > 64) gx = 64;
    int64_t gy = kGaccSlots / gx;
    if (gy > C) gy = C;
    if (gy < 1) gy = 1;
    // few columns and few channels but many rows per channel: slices of the rows in grid.z (<= kGaccSlots workgroups: one partial slot each)
    int64_t gzs = 1;
    if (gbias && gx * gy < 1024 && rows / C >= 8) {
        gzs = 1024 / (gx * gy);
        if (gzs > cdiv(rows / C, 4)) gzs = cdiv(rows / C, 4);
        if (gzs > kGaccSlots / (gx * gy)) gzs = kGaccSlots / (gx * gy);
        if (gzs > 65535) gzs = 65535;
        if (gzs < 1) gzs = 1;
    }
    dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)gzs);
#define FQSS_LAUNCH_BWD(V, Bi)                                                                                       \
    hipLaunchKernelGGL((k_actq_bwd<V, Bi>), grid, dim3(256), 0, s, z, g, gz, rows, cols, ld_z, ld_g, ld_gz, act,     \
                       slope, qmode, qmin, qmax, gacc, gbias, C)
    if (act == FQSS_ACT_POST_RELU) {
        if (vec) hipLaunchKernelGGL((k_actq_bwd<4, false, false, true>), grid, dim3(256), 0, s, z, g, gz, rows, cols, ld_z, ld_g, ld_gz, act, slope, qmode,
                                    qmin, qmax, gacc, gbias, C);
        else hipLaunchKernelGGL((k_actq_bwd<1, false, false, true>), grid, dim3(256), 0, s, z, g, gz, rows, cols, ld_z, ld_g, ld_gz, act, slope, qmode,
                                qmin, qmax, gacc, gbias, C);
    } else if (act == FQSS_ACT_GELU) {
        if (vec) hipLaunchKernelGGL((k_actq_bwd<4, false, true>), grid, dim3(256), 0, s, z, g, gz, rows, cols, ld_z, ld_g, ld_gz, act, slope, qmode, qmin,
                                    qmax, gacc, gbias, C);
        else hipLaunchKernelGGL((k_actq_bwd<1, false, true>), grid, dim3(256), 0, s, z, g, gz, rows, cols, ld_z, ld_g, ld_gz, act, slope, qmode, qmin,
                                qmax, gacc, gbias, C);
    } else if (vec) {
        if (gbias) FQSS_LAUNCH_BWD(4, true); else FQSS_LAUNCH_BWD(4, false);
    } else {
        if (gbias) FQSS_LAUNCH_BWD(1, true); else FQSS_LAUNCH_BWD(1, false);
    }
#undef FQSS_LAUNCH_BWD
    return launch_status("fqss_actq_bwd");
}

static int actq_bwd_colbias_impl(const float* z, const float* g, float* gz, int64_t R, int F, int64_t ld_z, int64_t ld_g, int64_t ld_gz, int act,
                                 const float* slope, int qmode, const float* qmin, const float* qmax, double* gacc, float* gbias,
                                 const float* qmin2, const float* qmax2, double* gacc2, fqss_stream_t stream);

extern "C" int fqss_actq_bwd_colbias(const float* z, const float* g, float* gz, int64_t R, int F, int64_t ld_z, int64_t ld_g,
                                     int64_t ld_gz, int act, const float* slope, int qmode, const float* qmin, const float* qmax,
                                     double* gacc, float* gbias, fqss_stream_t stream) {
    FQSS_REQUIRE(act != FQSS_ACT_RELU_Q, "two quantizers: fqss_actq2_bwd_colbias");
    return actq_bwd_colbias_impl(z, g, gz, R, F, ld_z, ld_g, ld_gz, act, slope, qmode, qmin, qmax, gacc, gbias, nullptr, nullptr, nullptr, stream);
}

extern "C" int fqss_actq2_bwd_colbias(const float* z, const float* g, float* gz, int64_t R, int F, int64_t ld_z, int64_t ld_g, int64_t ld_gz,
                                      const float* qmin1, const float* qmax1, const float* qmin2, const float* qmax2, double* gacc1,
                                      double* gacc2, float* gbias, fqss_stream_t stream) {
    FQSS_REQUIRE(qmin1 && qmax1 && qmin2 && qmax2 && gacc1 && gacc2, "two quantizers: ranges and partial slots of both");
    return actq_bwd_colbias_impl(z, g, gz, R, F, ld_z, ld_g, ld_gz, FQSS_ACT_RELU_Q, nullptr, FQSS_Q_QUANT, qmin1, qmax1, gacc1, gbias, qmin2, qmax2,
                                 gacc2, stream);
}

static int actq_bwd_colbias_impl(const float* z, const float* g, float* gz, int64_t R, int F, int64_t ld_z, int64_t ld_g, int64_t ld_gz, int act,
                                 const float* slope, int qmode, const float* qmin, const float* qmax, double* gacc, float* gbias,
                                 const float* qmin2, const float* qmax2, double* gacc2, fqss_stream_t stream) {
    if (R == 0 || F == 0) return FQSS_OK;
    FQSS_REQUIRE(z && g && gz, "null tensor");     // gbias may be NULL: the row-tiled pass alone (narrow matrices, see fqss_actq_bwd)
    FQSS_REQUIRE(R > 0 && F > 0 && ld_z >= F && ld_g >= F && ld_gz >= F, "bad shape");
    FQSS_REQUIRE(((act >= 0 && act <= 2) || act == FQSS_ACT_POST_RELU || act == FQSS_ACT_RELU_Q) && qmode >= 0 && qmode <= 2, "bad act/qmode");
    FQSS_REQUIRE(act != FQSS_ACT_PRELU || slope, "PReLU needs a slope");
    FQSS_REQUIRE(qmode != FQSS_Q_QUANT || (qmin && qmax), "QUANT needs ranges");
    FQSS_REQUIRE((qmode != FQSS_Q_QUANT && act != FQSS_ACT_PRELU) || gacc, "range/slope grads need gacc");
    FQSS_REQUIRE(F % 4 == 0 && aligned16(z) && aligned16(g) && aligned16(gz) && ld_z % 4 == 0 && ld_g % 4 == 0 && ld_gz % 4 == 0,
                 "feature count and row strides must be multiples of 4 floats, rows 16-B aligned");
    const int64_t slices = cdiv(F, 64);
    FQSS_REQUIRE(slices <= kGaccSlots, "too many features");
    int64_t parts = 1024 / slices;                   // ~1024 workgroups over the chip ...
    if (parts > 256) parts = 256;                    // ... at most 256 of them adding to the same feature
    if (parts > kGaccSlots / slices) parts = kGaccSlots / slices;
    if (parts > cdiv(R, 64)) parts = cdiv(R, 64);    // at least one full round of rows per workgroup
    if (parts < 1) parts = 1;
    const int64_t rows_per_part = cdiv(R, parts);
    parts = cdiv(R, rows_per_part);
    hipLaunchKernelGGL(k_actq_bwd_colbias, dim3((unsigned)slices, (unsigned)parts), dim3(256), 0, (hipStream_t)stream, z, g, gz, R, F, ld_z,
                       ld_g, ld_gz, act, slope, qmode, qmin, qmax, gacc, gbias, rows_per_part, qmin2, qmax2, gacc2);
    return launch_status("fqss_actq_bwd_colbias");
}

extern "C" int fqss_wq_observe(const float* w, int64_t outer, int64_t C, int64_t inner, float* qmin, float* qmax,
                               fqss_stream_t stream) {
    FQSS_REQUIRE(w && qmin && qmax && outer > 0 && C > 0 && inner > 0, "bad args");
    hipLaunchKernelGGL(k_wq_observe, dim3((unsigned)C), dim3(256), 0, (hipStream_t)stream, w, outer, C, inner, qmin, qmax);
    return launch_status("fqss_wq_observe");
}

extern "C" int fqss_wq_fwd(const float* w, float* wq, int8_t* idx, int64_t outer, int64_t C, int64_t inner,
                           const float* qmin, const float* qmax, fqss_stream_t stream) {
    FQSS_REQUIRE(w && wq && qmin && qmax && outer > 0 && C > 0 && inner > 0, "bad args");
    const int64_t n = outer * C * inner;
    int64_t nb = cdiv(n, 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_wq_fwd, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, w, wq, idx, n, C, inner, qmin, qmax);
    return launch_status("fqss_wq_fwd");
}

extern "C" int fqss_gacc_flush(double* gacc, float* gmin, float* gmax, float* gslope, fqss_stream_t stream) {
    FQSS_REQUIRE(gacc, "null accumulator");
    hipLaunchKernelGGL(k_gacc_flush, dim3(1), dim3(256), 0, (hipStream_t)stream, gacc, gmin, gmax, gslope);
    return launch_status("fqss_gacc_flush");
}

extern "C" int fqss_wq_bwd(const float* w, const float* g, float* gw, float* gmin, float* gmax, int64_t outer,
                           int64_t C, int64_t inner, const float* qmin, const float* qmax, int accumulate,
                           fqss_stream_t stream) {
    FQSS_REQUIRE(w && g && gw && gmin && gmax && qmin && qmax && outer > 0 && C > 0 && inner > 0, "bad args");
    hipLaunchKernelGGL(k_wq_bwd, dim3((unsigned)C), dim3(256), 0, (hipStream_t)stream, w, g, gw, gmin, gmax, outer, C,
                       inner, qmin, qmax, accumulate);
    return launch_status("fqss_wq_bwd");
}

static bool gluq_rows_ok(const void* a, const void* b, int64_t lda, int64_t ldb, int64_t M) {
    return aligned16(a) && aligned16(b) && lda % 4 == 0 && ldb % 4 == 0 && lda >= ((M + 3) & ~(int64_t)3) && ldb >= ((M + 3) & ~(int64_t)3);
}

extern "C" int fqss_gluq_fwd(const float* z, float* out, int64_t B, int64_t C, int64_t M, int64_t ld_z, int64_t ld_out, int qmode,
                             const float* qmin, const float* qmax, uint32_t* obs_ws, fqss_stream_t stream) {
    if (B * C * M == 0) return FQSS_OK;
    FQSS_REQUIRE(z && out && B > 0 && C > 0 && M > 0 && qmode >= 0 && qmode <= 2, "bad args");
    FQSS_REQUIRE(qmode != FQSS_Q_QUANT || (qmin && qmax), "QUANT needs ranges");
    FQSS_REQUIRE(qmode != FQSS_Q_OBSERVE || obs_ws, "OBSERVE needs obs_ws");
    FQSS_REQUIRE(gluq_rows_ok(z, out, ld_z, ld_out, M), "rows must be 16-B aligned and padded to a multiple of 4");
    int64_t gx = cdiv(M, 1024);
    if (gx > 64) gx = 64;
    int64_t gy = 4096 / gx;
    if (gy > B * C) gy = B * C;
    hipLaunchKernelGGL(k_gluq_fwd, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, z, out, B, C, M, ld_z, ld_out, qmode, qmin,
                       qmax, obs_ws);
    return launch_status("fqss_gluq_fwd");
}

extern "C" int fqss_gluq_bwd(const float* z, const float* g, float* gz, int64_t B, int64_t C, int64_t M, int64_t ld_z, int64_t ld_g,
                             int64_t ld_gz, int qmode, const float* qmin, const float* qmax, double* gacc, fqss_stream_t stream) {
    if (B * C * M == 0) return FQSS_OK;
    FQSS_REQUIRE(z && g && gz && B > 0 && C > 0 && M > 0 && qmode >= 0 && qmode <= 2, "bad args");
    FQSS_REQUIRE(qmode != FQSS_Q_QUANT || (qmin && qmax && gacc), "QUANT needs ranges and gacc");
    FQSS_REQUIRE(gluq_rows_ok(z, g, ld_z, ld_g, M) && gluq_rows_ok(z, gz, ld_z, ld_gz, M), "rows must be 16-B aligned and padded to a multiple of 4");
    int64_t gx = cdiv(M, 1024);
    if (gx > 64) gx = 64;
    int64_t gy = kGaccSlots / gx;
    if (gy > B * C) gy = B * C;
    if (gy < 1) gy = 1;
    hipLaunchKernelGGL(k_gluq_bwd, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, z, g, gz, B, C, M, ld_z, ld_g, ld_gz, qmode,
                       qmin, qmax, gacc);
    return launch_status("fqss_gluq_bwd");
}
