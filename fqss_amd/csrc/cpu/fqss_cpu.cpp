// fqss_cpu.cpp -- the CPU backend behind include/fqss.h for cfg 1 of BASELINE.json ("convtasnet_2spks_8k.yaml on asteroid env, CPU,
// batch 2, 1 s ... plumbing, no GPU"; reference train.py:31: device = "cpu" if use_cpu).  Plain C++ (g++, OpenMP), host pointers, the
// `stream` argument is ignored, every call is synchronous.  It serves the entry points the UN-FUSED ConvTasNet QAT step uses (the
// per-layer kernels that the G1 layer fixtures pin: KDTrainStep(coded=False, batched_quantizers=False) with the module-path teacher);
// anything else is absent from this library and fqss_amd/_lib.py raises.  Arithmetic: the same op sequences as the HIP kernels
// (fp32, one IEEE operation per operator: built with -ffp-contract=off; IEEE division; round-half-even), reductions in fp64.
// NOT the oracle: oracle/ is test infrastructure and is never loaded by the product; this file is product code selected by --use_cpu.
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "fqss.h"

namespace {
thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
#define REQUIRE(cond, msg)                          \
    do {                                            \
        if (!(cond)) {                              \
            set_error("%s: %s", __func__, msg);     \
            return FQSS_EINVAL;                     \
        }                                           \
    } while (0)

inline uint32_t f2ord(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
inline float ord2f(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
inline float act_neg_scale(int act, float slope) { return act == FQSS_ACT_PRELU ? slope : (act == FQSS_ACT_RELU ? 0.0f : 1.0f); }
inline float act_apply(float z, int act, float slope) { return z > 0.0f ? z : act_neg_scale(act, slope) * z; }
struct QRange {
    float lo, delta;
};
inline QRange load_qrange(const float* qmin, const float* qmax) { return QRange{*qmin, (*qmax - *qmin) / 255.0f}; }
// qat_quant.py:139-146 op for op; c = clamped index, u = pre-round coordinate
inline float fq_asym(float t, const QRange& r, float& c, float& u, bool& inr) {
    u = (t - r.lo) / r.delta;
    const float X = nearbyintf(u);   // round-half-to-even (default rounding mode) == torch.round
    inr = (X >= 0.0f) && (X <= 255.0f);
    c = fminf(fmaxf(X, 0.0f), 255.0f);
    return r.delta * c + r.lo;
}
inline float wq_delta(float lo, float hi) { return (2.0f * fmaxf(fabsf(lo), fabsf(hi))) / 255.0f; }
}  // namespace

extern "C" {

int fqss_version(void) { return FQSS_VERSION; }
const char* fqss_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------ activation quantizer
int fqss_actq_fwd(const float* z, float* out, uint8_t* idx, int64_t rows, int64_t cols, int64_t ld_z, int64_t ld_out, int64_t ld_idx,
                  int act, const float* slope_p, int qmode, const float* qmin, const float* qmax, uint32_t* obs_ws, fqss_stream_t) {
    if (rows == 0 || cols == 0) return FQSS_OK;
    REQUIRE(z && (out || idx) && rows >= 0 && cols >= 0 && ld_z >= cols, "bad args");
    REQUIRE(act >= 0 && act <= FQSS_ACT_RELU, "act not served by the CPU backend (NONE / PReLU / ReLU only)");
    REQUIRE(act != FQSS_ACT_PRELU || slope_p, "PReLU needs a slope");
    REQUIRE(qmode != FQSS_Q_QUANT || (qmin && qmax), "QUANT needs ranges");
    REQUIRE(qmode != FQSS_Q_OBSERVE || obs_ws, "OBSERVE needs obs_ws");
    const float slope = act == FQSS_ACT_PRELU ? *slope_p : 0.0f;
    QRange r{0.0f, 1.0f};
    if (qmode == FQSS_Q_QUANT) r = load_qrange(qmin, qmax);
    float vmin = INFINITY, vmax = -INFINITY;
#pragma omp parallel for reduction(min : vmin) reduction(max : vmax) schedule(static)
    for (int64_t row = 0; row < rows; ++row) {
        const float* zr = z + row * ld_z;
        for (int64_t c0 = 0; c0 < cols; ++c0) {
            const float t = act_apply(zr[c0], act, slope);
            float o = t;
            if (qmode == FQSS_Q_QUANT) {
                float c, u;
                bool inr;
                o = fq_asym(t, r, c, u, inr);
                if (idx) idx[row * ld_idx + c0] = (uint8_t)c;
            } else if (qmode == FQSS_Q_OBSERVE) {
                vmin = fminf(vmin, t);
                vmax = fmaxf(vmax, t);
            }
            if (out) out[row * ld_out + c0] = o;
        }
    }
    if (qmode == FQSS_Q_OBSERVE) {
        const uint32_t kmin = f2ord(vmin), kmax = f2ord(vmax);
        if (kmin < obs_ws[0]) obs_ws[0] = kmin;
        if (kmax > obs_ws[1]) obs_ws[1] = kmax;
    }
    return FQSS_OK;
}

int fqss_obs_reset(uint32_t* obs_ws, int64_t n_pairs, fqss_stream_t) {
    REQUIRE(obs_ws && n_pairs >= 0, "bad args");
    for (int64_t i = 0; i < n_pairs; ++i) {
        obs_ws[2 * i] = 0xFFFFFFFFu;
        obs_ws[2 * i + 1] = 0u;
    }
    return FQSS_OK;
}

int fqss_observer_ema(float* qmin, float* qmax, uint32_t* obs_ws, double alpha, fqss_stream_t) {
    REQUIRE(qmin && qmax && obs_ws, "null pointer");
    const float a = (float)alpha, oma = (float)(1.0 - alpha);
    const float tmin = ord2f(obs_ws[0]), tmax = ord2f(obs_ws[1]);
    *qmin = a * (*qmin) + oma * tmin;   // qat_quant.py:231-232
    *qmax = a * (*qmax) + oma * tmax;
    obs_ws[0] = 0xFFFFFFFFu;
    obs_ws[1] = 0u;
    return FQSS_OK;
}

int fqss_minmax(const float* x, int64_t rows, int64_t cols, int64_t ld, uint32_t* obs_ws, fqss_stream_t) {
    if (rows == 0 || cols == 0) return FQSS_OK;
    REQUIRE(x && obs_ws && ld >= cols, "bad args");
    float vmin = INFINITY, vmax = -INFINITY;
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t c = 0; c < cols; ++c) {
            vmin = fminf(vmin, x[r * ld + c]);
            vmax = fmaxf(vmax, x[r * ld + c]);
        }
    const uint32_t kmin = f2ord(vmin), kmax = f2ord(vmax);
    if (kmin < obs_ws[0]) obs_ws[0] = kmin;
    if (kmax > obs_ws[1]) obs_ws[1] = kmax;
    return FQSS_OK;
}

// backward of out = fq(act(z)): g_C = g*delta ; g_u = g_C*m ; g_t = g_u/delta ; d/dmax = sum g*(c - m*u)/255 ; d/dmin = sum g*(1-m) - d/dmax
int fqss_actq_bwd(const float* z, const float* g, float* gz, int64_t rows, int64_t cols, int64_t ld_z, int64_t ld_g, int64_t ld_gz, int act,
                  const float* slope_p, int qmode, const float* qmin, const float* qmax, double* gacc, float* gbias, int64_t C,
                  fqss_stream_t) {
    if (rows == 0 || cols == 0) return FQSS_OK;
    REQUIRE(z && g && gz && ld_z >= cols && ld_g >= cols && ld_gz >= cols, "bad args");
    REQUIRE(act >= 0 && act <= FQSS_ACT_RELU, "act not served by the CPU backend (NONE / PReLU / ReLU only)");
    REQUIRE(act != FQSS_ACT_PRELU || slope_p, "PReLU needs a slope");
    REQUIRE(qmode != FQSS_Q_QUANT || (qmin && qmax), "QUANT needs ranges");
    REQUIRE((qmode != FQSS_Q_QUANT && act != FQSS_ACT_PRELU) || gacc, "range/slope grads need gacc");
    REQUIRE(!gbias || (C > 0 && rows % C == 0), "gbias needs C dividing rows");
    if (!gbias || C <= 0) C = rows;
    const float slope = act == FQSS_ACT_PRELU ? *slope_p : 0.0f;
    QRange r{0.0f, 1.0f};
    if (qmode == FQSS_Q_QUANT) r = load_qrange(qmin, qmax);
    double p_du = 0.0, p_out = 0.0, p_slope = 0.0;
    std::vector<double> bias(gbias ? (size_t)C : 0, 0.0);
    for (int64_t row = 0; row < rows; ++row) {
        double pb = 0.0;
        for (int64_t c0 = 0; c0 < cols; ++c0) {
            const float zv = z[row * ld_z + c0], gj = g[row * ld_g + c0];
            const float t = act_apply(zv, act, slope);
            float gt = gj;
            if (qmode == FQSS_Q_QUANT) {
                float c, u;
                bool inr;
                (void)fq_asym(t, r, c, u, inr);
                gt = inr ? (gj * r.delta) / r.delta : 0.0f;
                p_du += (double)(gj * (inr ? (c - u) : c));
                p_out += inr ? 0.0 : (double)gj;
            }
            const bool neg = !(zv > 0.0f);
            if (act == FQSS_ACT_PRELU && neg) p_slope += (double)(zv * gt);
            const float gzj = neg ? act_neg_scale(act, slope) * gt : gt;
            gz[row * ld_gz + c0] = gzj;
            pb += (double)gzj;
        }
        if (gbias) bias[(size_t)(row % C)] += pb;
    }
    if (gbias)
        for (int64_t c = 0; c < C; ++c) gbias[c] += (float)bias[(size_t)c];
    if (qmode == FQSS_Q_QUANT || act == FQSS_ACT_PRELU) {
        const double dmax = p_du / 255.0;
        gacc[0] += qmode == FQSS_Q_QUANT ? p_out - dmax : 0.0;
        gacc[1] += qmode == FQSS_Q_QUANT ? dmax : 0.0;
        gacc[2] += p_slope;
    }
    return FQSS_OK;
}

int fqss_gacc_flush(double* gacc, float* gmin, float* gmax, float* gslope, fqss_stream_t) {
    REQUIRE(gacc, "null gacc");
    double v[3] = {0.0, 0.0, 0.0};
    for (int i = 0; i < FQSS_GACC_SLOTS; ++i)
        for (int k = 0; k < 3; ++k) {
            v[k] += gacc[3 * i + k];
            gacc[3 * i + k] = 0.0;
        }
    if (gmin) *gmin += (float)v[0];
    if (gmax) *gmax += (float)v[1];
    if (gslope) *gslope += (float)v[2];
    return FQSS_OK;
}

// ------------------------------------------------------------------------------------------------ weight quantizer (qat_quant.py:126-135)
int fqss_wq_observe(const float* w, int64_t outer, int64_t C, int64_t inner, float* qmin, float* qmax, fqss_stream_t) {
    REQUIRE(w && qmin && qmax && outer > 0 && C > 0 && inner > 0, "bad args");
    for (int64_t c = 0; c < C; ++c) {
        float vmin = INFINITY, vmax = -INFINITY;
        for (int64_t o = 0; o < outer; ++o)
            for (int64_t i = 0; i < inner; ++i) {
                const float v = w[(o * C + c) * inner + i];
                vmin = fminf(vmin, v);
                vmax = fmaxf(vmax, v);
            }
        qmin[c] = vmin;
        qmax[c] = vmax;
    }
    return FQSS_OK;
}

int fqss_wq_fwd(const float* w, float* wq, int8_t* idx, int64_t outer, int64_t C, int64_t inner, const float* qmin, const float* qmax,
                fqss_stream_t) {
    REQUIRE(w && wq && qmin && qmax && outer > 0 && C > 0 && inner > 0, "bad args");
    const int64_t n = outer * C * inner;
    for (int64_t e = 0; e < n; ++e) {
        const int64_t c = (e / inner) % C;
        const float delta = wq_delta(qmin[c], qmax[c]);
        const float X = nearbyintf(w[e] / delta);
        const float q = fminf(fmaxf(X, -128.0f), 127.0f);
        wq[e] = delta * q;
        if (idx) idx[e] = (int8_t)q;
    }
    return FQSS_OK;
}

int fqss_wq_bwd(const float* w, const float* g, float* gw, float* gmin, float* gmax, int64_t outer, int64_t C, int64_t inner,
                const float* qmin, const float* qmax, int accumulate, fqss_stream_t) {
    REQUIRE(w && g && gw && gmin && gmax && qmin && qmax && outer > 0 && C > 0 && inner > 0, "bad args");
    for (int64_t c = 0; c < C; ++c) {
        const float lo = qmin[c], hi = qmax[c];
        const float delta = wq_delta(lo, hi);
        double p = 0.0;
        for (int64_t o = 0; o < outer; ++o)
            for (int64_t i = 0; i < inner; ++i) {
                const int64_t k = (o * C + c) * inner + i;
                const float u = w[k] / delta;
                const float X = nearbyintf(u);
                const bool inr = (X >= -128.0f) && (X <= 127.0f);
                const float q = fminf(fmaxf(X, -128.0f), 127.0f);
                const float gk = g[k];
                const float gwk = inr ? (gk * delta) / delta : 0.0f;
                gw[k] = accumulate ? gw[k] + gwk : gwk;
                p += (double)(gk * (inr ? (q - u) : q));
            }
        const double D = p * (2.0 / 255.0);
        const float al = fabsf(lo), ah = fabsf(hi);
        const double wl = al > ah ? 1.0 : (al == ah ? 0.5 : 0.0), wh = ah > al ? 1.0 : (al == ah ? 0.5 : 0.0);
        const double sl = lo > 0.0f ? 1.0 : (lo < 0.0f ? -1.0 : 0.0), sh = hi > 0.0f ? 1.0 : (hi < 0.0f ? -1.0 : 0.0);
        const float dmin = (float)(D * wl * sl), dmax = (float)(D * wh * sh);
        gmin[c] = accumulate ? gmin[c] + dmin : dmin;
        gmax[c] = accumulate ? gmax[c] + dmax : dmax;
    }
    return FQSS_OK;
}

// ------------------------------------------------------------------------------------------------ pointwise conv (qat_layers.py:137-146)
int fqss_pwconv_fwd(const float* x, const float* w, const float* bias, float* z, int B, int Ci, int Co, int M, int64_t ld_x, int64_t ld_z,
                    fqss_stream_t) {
    if (B == 0 || M == 0) return FQSS_OK;
    REQUIRE(x && w && z && Ci > 0 && Co > 0 && ld_x >= M && ld_z >= M, "bad args");
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Co; ++co) {
            float* zr = z + ((int64_t)b * Co + co) * ld_z;
            const float bv = bias ? bias[co] : 0.0f;
            for (int m = 0; m < M; ++m) zr[m] = bv;
            for (int ci = 0; ci < Ci; ++ci) {
                const float wv = w[(int64_t)co * Ci + ci];
                const float* xr = x + ((int64_t)b * Ci + ci) * ld_x;
                for (int m = 0; m < M; ++m) zr[m] += wv * xr[m];
            }
        }
    return FQSS_OK;
}
int fqss_pwconv_fwd_x3(const float* x, const float* w, const float* bias, float* z, int B, int Ci, int Co, int M, int64_t ld_x,
                       int64_t ld_z, fqss_stream_t s) {
    return fqss_pwconv_fwd(x, w, bias, z, B, Ci, Co, M, ld_x, ld_z, s);
}
int fqss_pwconv_bwd_x(const float* gz, const float* w, float* gx, int B, int Ci, int Co, int M, int64_t ld_gz, int64_t ld_gx, fqss_stream_t) {
    if (B == 0 || M == 0) return FQSS_OK;
    REQUIRE(gz && w && gx && Ci > 0 && Co > 0 && ld_gz >= M && ld_gx >= M, "bad args");
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int ci = 0; ci < Ci; ++ci) {
            float* gr = gx + ((int64_t)b * Ci + ci) * ld_gx;
            for (int m = 0; m < M; ++m) gr[m] = 0.0f;
            for (int co = 0; co < Co; ++co) {
                const float wv = w[(int64_t)co * Ci + ci];
                const float* zr = gz + ((int64_t)b * Co + co) * ld_gz;
                for (int m = 0; m < M; ++m) gr[m] += wv * zr[m];
            }
        }
    return FQSS_OK;
}
int fqss_pwconv_bwd_w(const float* gz, const float* x, float* gw, int B, int Ci, int Co, int M, int64_t ld_gz, int64_t ld_x, fqss_stream_t) {
    if (B == 0 || M == 0) return FQSS_OK;
    REQUIRE(gz && x && gw && Ci > 0 && Co > 0 && ld_gz >= M && ld_x >= M, "bad args");
#pragma omp parallel for collapse(2) schedule(static)
    for (int co = 0; co < Co; ++co)
        for (int ci = 0; ci < Ci; ++ci) {
            double s = 0.0;
            for (int b = 0; b < B; ++b) {
                const float* zr = gz + ((int64_t)b * Co + co) * ld_gz;
                const float* xr = x + ((int64_t)b * Ci + ci) * ld_x;
                float p = 0.0f;
                for (int m = 0; m < M; ++m) p += zr[m] * xr[m];
                s += (double)p;
            }
            gw[(int64_t)co * Ci + ci] += (float)s;
        }
    return FQSS_OK;
}

// ------------------------------------------------------------------------------------------------ depthwise conv (convtasnetq.py:28-30)
int fqss_dwconv_fwd(const float* x, const float* w, const float* bias, float* z, int B, int C, int M, int K, int dil, int pad, int64_t ld_x,
                    int64_t ld_z, fqss_stream_t) {
    if (B == 0 || M == 0) return FQSS_OK;
    REQUIRE(x && w && z && C > 0 && K > 0 && dil > 0 && 2 * pad == dil * (K - 1), "bad args");
#pragma omp parallel for schedule(static)
    for (int64_t row = 0; row < (int64_t)B * C; ++row) {
        const int c = (int)(row % C);
        const float* xr = x + row * ld_x;
        float* zr = z + row * ld_z;
        for (int m = 0; m < M; ++m) {
            float acc = bias ? bias[c] : 0.0f;
            for (int k = 0; k < K; ++k) {
                const int j = m + k * dil - pad;
                if (j >= 0 && j < M) acc += w[c * K + k] * xr[j];
            }
            zr[m] = acc;
        }
    }
    return FQSS_OK;
}
int fqss_dwconv_bwd_x(const float* gz, const float* w, float* gx, int B, int C, int M, int K, int dil, int pad, int64_t ld_gz, int64_t ld_gx,
                      fqss_stream_t) {
    if (B == 0 || M == 0) return FQSS_OK;
    REQUIRE(gz && w && gx && C > 0 && K > 0 && dil > 0, "bad args");
#pragma omp parallel for schedule(static)
    for (int64_t row = 0; row < (int64_t)B * C; ++row) {
        const int c = (int)(row % C);
        const float* gr = gz + row * ld_gz;
        float* xr = gx + row * ld_gx;
        for (int j = 0; j < M; ++j) {
            float acc = 0.0f;
            for (int k = 0; k < K; ++k) {
                const int m = j - k * dil + pad;
                if (m >= 0 && m < M) acc += w[c * K + k] * gr[m];
            }
            xr[j] = acc;
        }
    }
    return FQSS_OK;
}
int fqss_dwconv_bwd_w(const float* gz, const float* x, float* gw, int B, int C, int M, int K, int dil, int pad, int64_t ld_gz, int64_t ld_x,
                      fqss_stream_t) {
    if (B == 0 || M == 0) return FQSS_OK;
    REQUIRE(gz && x && gw && C > 0 && K > 0 && dil > 0, "bad args");
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c)
        for (int k = 0; k < K; ++k) {
            double s = 0.0;
            for (int b = 0; b < B; ++b) {
                const float* gr = gz + ((int64_t)b * C + c) * ld_gz;
                const float* xr = x + ((int64_t)b * C + c) * ld_x;
                for (int m = 0; m < M; ++m) {
                    const int j = m + k * dil - pad;
                    if (j >= 0 && j < M) s += (double)(gr[m] * xr[j]);
                }
            }
            gw[c * K + k] += (float)s;
        }
    return FQSS_OK;
}

// ------------------------------------------------------------------------------------------------ GroupNorm(1, C) (qat_layers.py:445-448)
int fqss_gn_fwd(const float* x, const float* gamma, const float* beta, float* z, float* mean_rstd, int B, int C, int M, int64_t ld_x,
                int64_t ld_z, float eps, double* ws, fqss_stream_t) {
    if (B == 0) return FQSS_OK;
    REQUIRE(x && gamma && beta && z && mean_rstd && C > 0 && M > 0 && ld_x >= M && ld_z >= M, "bad args");
    (void)ws;
    for (int b = 0; b < B; ++b) {
        double s = 0.0, ss = 0.0;
        for (int c = 0; c < C; ++c) {
            const float* xr = x + ((int64_t)b * C + c) * ld_x;
            for (int m = 0; m < M; ++m) {
                s += (double)xr[m];
                ss += (double)xr[m] * (double)xr[m];
            }
        }
        const double n = (double)C * M, mu = s / n;
        double var = ss / n - mu * mu;
        if (var < 0.0) var = 0.0;
        const float mean = (float)mu, rstd = (float)(1.0 / sqrt(var + (double)eps));
        mean_rstd[2 * b] = mean;
        mean_rstd[2 * b + 1] = rstd;
        for (int c = 0; c < C; ++c) {
            const float scale = rstd * gamma[c], shift = fmaf(-scale, mean, beta[c]);
            const float* xr = x + ((int64_t)b * C + c) * ld_x;
            float* zr = z + ((int64_t)b * C + c) * ld_z;
            for (int m = 0; m < M; ++m) zr[m] = fmaf(xr[m], scale, shift);
        }
    }
    return FQSS_OK;
}
int fqss_gn_bwd(const float* gz, const float* x, const float* gamma, const float* mean_rstd, float* gx, float* ggamma, float* gbeta, int B,
                int C, int M, int64_t ld_gz, int64_t ld_x, int64_t ld_gx, double* ws, fqss_stream_t) {
    if (B == 0) return FQSS_OK;
    REQUIRE(gz && x && gamma && mean_rstd && gx && ggamma && gbeta && C > 0 && M > 0, "bad args");
    (void)ws;
    std::vector<double> gg((size_t)C, 0.0), gb((size_t)C, 0.0);
    for (int b = 0; b < B; ++b) {
        const float mean = mean_rstd[2 * b], rstd = mean_rstd[2 * b + 1];
        double s1 = 0.0, s2 = 0.0;      // sum gamma*g, sum gamma*g*xhat over the sample
        for (int c = 0; c < C; ++c) {
            const float* gr = gz + ((int64_t)b * C + c) * ld_gz;
            const float* xr = x + ((int64_t)b * C + c) * ld_x;
            double a = 0.0, d = 0.0;
            for (int m = 0; m < M; ++m) {
                const float xh = (xr[m] - mean) * rstd;
                a += (double)gr[m];
                d += (double)(gr[m] * xh);
            }
            gb[(size_t)c] += a;
            gg[(size_t)c] += d;
            s1 += (double)gamma[c] * a;
            s2 += (double)gamma[c] * d;
        }
        const double n = (double)C * M;
        const float m1 = (float)(s1 / n), m2 = (float)(s2 / n);
        for (int c = 0; c < C; ++c) {
            const float* gr = gz + ((int64_t)b * C + c) * ld_gz;
            const float* xr = x + ((int64_t)b * C + c) * ld_x;
            float* or_ = gx + ((int64_t)b * C + c) * ld_gx;
            for (int m = 0; m < M; ++m) {
                const float xh = (xr[m] - mean) * rstd;
                or_[m] = rstd * ((gamma[c] * gr[m] - m1) - xh * m2);
            }
        }
    }
    for (int c = 0; c < C; ++c) {
        ggamma[c] += (float)gg[(size_t)c];
        gbeta[c] += (float)gb[(size_t)c];
    }
    return FQSS_OK;
}

// ------------------------------------------------------------------------------------------------ element-wise producers
int fqss_axpby(const float* a, const float* b, float sa, float sb, float* z, int64_t rows, int64_t cols, int64_t ld_a, int64_t ld_b,
               int64_t ld_z, fqss_stream_t) {
    if (rows == 0 || cols == 0) return FQSS_OK;
    REQUIRE(a && b && z, "null tensor");
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t c = 0; c < cols; ++c) z[r * ld_z + c] = sa * a[r * ld_a + c] + sb * b[r * ld_b + c];
    return FQSS_OK;
}
int fqss_mul_bcast_fwd(const float* mask, const float* feat, float* z, int B, int S, int C, int M, int64_t ld_mask, int64_t ld_feat,
                       int64_t ld_z, fqss_stream_t) {
    if (B == 0 || M == 0) return FQSS_OK;
    REQUIRE(mask && feat && z, "null tensor");
    for (int b = 0; b < B; ++b)
        for (int s = 0; s < S; ++s)
            for (int c = 0; c < C; ++c) {
                const int64_t r = ((int64_t)b * S + s) * C + c;
                const float* fr = feat + ((int64_t)b * C + c) * ld_feat;
                for (int m = 0; m < M; ++m) z[r * ld_z + m] = mask[r * ld_mask + m] * fr[m];
            }
    return FQSS_OK;
}
int fqss_mul_bcast_bwd(const float* gz, const float* mask, const float* feat, float* gmask, float* gfeat, int B, int S, int C, int M,
                       int64_t ld_gz, int64_t ld_mask, int64_t ld_feat, int64_t ld_gmask, int64_t ld_gfeat, fqss_stream_t) {
    if (B == 0 || M == 0) return FQSS_OK;
    REQUIRE(gz && mask && feat && gmask && gfeat, "null tensor");
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            const float* fr = feat + ((int64_t)b * C + c) * ld_feat;
            float* gf = gfeat + ((int64_t)b * C + c) * ld_gfeat;
            for (int m = 0; m < M; ++m) gf[m] = 0.0f;
            for (int s = 0; s < S; ++s) {      // ascending s, like the kernel
                const int64_t r = ((int64_t)b * S + s) * C + c;
                for (int m = 0; m < M; ++m) {
                    const float gv = gz[r * ld_gz + m];
                    gmask[r * ld_gmask + m] = gv * fr[m];
                    gf[m] += gv * mask[r * ld_mask + m];
                }
            }
        }
    return FQSS_OK;
}

// ------------------------------------------------------------------------------------------------ splitter, framing conv, overlap-add
static inline float split_q(float x) {      // process.py:10-14 with threshold = 1, n_bits = 8, sign = True
    const float delta = 0.0078125f;
    return fminf(fmaxf(floorf(x / delta), -128.0f), 127.0f) * delta;
}
int fqss_splitter2(const float* x, float* out, int B, int64_t T, const uint32_t* obs_ws, fqss_stream_t) {
    if (B == 0 || T == 0) return FQSS_OK;
    REQUIRE(x && out && obs_ws, "null pointer");
    const float thr = fmaxf(fabsf(ord2f(obs_ws[0])), fabsf(ord2f(obs_ws[1])));   // process.py:24
    const float delta = 0.0078125f;
    for (int64_t b = 0; b < B; ++b)
        for (int64_t t = 0; t < T; ++t) {
            const float v = x[b * T + t] / thr;
            const float q0 = split_q(v);
            const float r = ((2.0f * (v - q0)) * 1.0f) / delta - 1.0f;          // process.py:35 op order
            out[(b * 2 + 0) * T + t] = q0;
            out[(b * 2 + 1) * T + t] = split_q(r);
        }
    return FQSS_OK;
}
int fqss_frames_conv_fwd(const float* x, const float* w, float* z, int N, int Ci, int Co, int64_t T, int K, int stride, int M, int64_t ld_z,
                         fqss_stream_t) {
    if (N == 0 || M == 0) return FQSS_OK;
    REQUIRE(x && w && z && Ci > 0 && Co > 0 && K > 0 && stride > 0 && (int64_t)(M - 1) * stride + K <= T && ld_z >= M, "bad args");
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Co; ++co)
            for (int m = 0; m < M; ++m) {
                float acc = 0.0f;
                for (int ci = 0; ci < Ci; ++ci) {
                    const float* xp = x + ((int64_t)n * Ci + ci) * T + (int64_t)m * stride;
                    const float* wr = w + ((int64_t)co * Ci + ci) * K;
                    for (int k = 0; k < K; ++k) acc = fmaf(xp[k], wr[k], acc);      // the kernel's j-ordered fmaf chain
                }
                z[((int64_t)n * Co + co) * ld_z + m] = acc;
            }
    return FQSS_OK;
}
int fqss_ola_convtr_fwd(const float* x, const float* w, float* out, int N, int C, int M, int64_t ld_x, int K, int stride, int64_t T,
                        fqss_stream_t) {
    if (N == 0 || M == 0) return FQSS_OK;
    REQUIRE(x && w && out && C > 0 && K > 0 && stride > 0 && T == (int64_t)(M - 1) * stride + K && ld_x >= M, "T must equal (M-1)*stride + K");
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
        std::vector<double> acc((size_t)T, 0.0);
        for (int c = 0; c < C; ++c) {
            const float* xr = x + ((int64_t)n * C + c) * ld_x;
            for (int m = 0; m < M; ++m)
                for (int k = 0; k < K; ++k) acc[(size_t)((int64_t)m * stride + k)] += (double)(xr[m] * w[c * K + k]);
        }
        for (int64_t t = 0; t < T; ++t) out[(int64_t)n * T + t] = (float)acc[(size_t)t];
    }
    return FQSS_OK;
}
int fqss_frames_wgrad1s(const float* a, const float* sig, int64_t sig_ns, float* gw, int64_t ld_gw, int N, int C, int M, int64_t ld_a,
                        int64_t T, int K, int stride, fqss_stream_t) {
    if (N == 0 || M == 0) return FQSS_OK;
    REQUIRE(a && sig && gw && C > 0 && K > 0 && stride > 0 && ld_a >= M && (int64_t)(M - 1) * stride + K <= T && sig_ns >= T && ld_gw >= K,
            "bad args");
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c)
        for (int k = 0; k < K; ++k) {
            double s = 0.0;
            for (int n = 0; n < N; ++n) {
                const float* ar = a + ((int64_t)n * C + c) * ld_a;
                const float* sr = sig + (int64_t)n * sig_ns + k;
                for (int m = 0; m < M; ++m) s += (double)(ar[m] * sr[(int64_t)m * stride]);
            }
            gw[(int64_t)c * ld_gw + k] += (float)s;
        }
    return FQSS_OK;
}
int fqss_frames_wgrad1(const float* a, const float* sig, float* gw, int N, int C, int M, int64_t ld_a, int64_t T, int K, int stride,
                       fqss_stream_t s) {
    return fqss_frames_wgrad1s(a, sig, T, gw, K, N, C, M, ld_a, T, K, stride, s);
}
int fqss_frames_wgrad(const float* a, const float* x, float* gw, int N, int C, int Ci, int M, int64_t ld_a, int64_t T, int K, int stride,
                      fqss_stream_t s) {
    for (int ci = 0; ci < Ci; ++ci) {
        const int rc = fqss_frames_wgrad1s(a, x + (int64_t)ci * T, (int64_t)Ci * T, gw + (int64_t)ci * K, (int64_t)Ci * K, N, C, M, ld_a, T, K,
                                           stride, s);
        if (rc) return rc;
    }
    return FQSS_OK;
}

// ------------------------------------------------------------------------------------------------ KD loss (mysystem.py:124-151, wsdr.py:56-95)
namespace {
constexpr double kEps = 1e-8;
struct PairSdr {
    double sdr, cx, cy;
};
PairSdr pair_sdr(double Sx, double Sy, double Sxx, double Syy, double Sxy, double T) {
    const double X2 = Sxx - Sx * Sx / T, Y2 = Syy - Sy * Sy / T, D = Sxy - Sx * Sy / T;
    const double E = Y2 + kEps, alpha = D / E, P = alpha * alpha * Y2;
    double Nn = X2 - 2.0 * alpha * D + alpha * alpha * Y2;
    if (Nn < 0.0) Nn = 0.0;
    const double den = Nn + kEps;
    PairSdr r;
    r.sdr = P / den;
    r.cx = -2.0 * P / (den * den);
    r.cy = 2.0 * alpha * Y2 / (E * den) + P * (2.0 * alpha + 2.0 * D * kEps / (E * E)) / (den * den);
    return r;
}
}  // namespace

int fqss_kd_loss(const float* est, const float* fest, const float* tgt, int B, int64_t T, float kd_lambda, double* stats, float* out,
                 float* w_out, float* sisdr_out, float* gest, fqss_stream_t) {
    REQUIRE(est && fest && tgt && out && w_out && sisdr_out && B > 0 && T > 0, "bad args");
    (void)stats;
    struct Smp {
        double m[24];
        PairSdr st[2][2], sf[2][2];
        int pt, pf;
        double task, kd, w;
    };
    std::vector<Smp> S((size_t)B);
    double task = 0.0, kd = 0.0;
    const double Td = (double)T;
    for (int b = 0; b < B; ++b) {
        Smp& q = S[(size_t)b];
        for (int i = 0; i < 24; ++i) q.m[i] = 0.0;
        const float* sig[6] = {est + (int64_t)b * 2 * T,  est + (int64_t)b * 2 * T + T, fest + (int64_t)b * 2 * T, fest + (int64_t)b * 2 * T + T,
                               tgt + (int64_t)b * 2 * T,  tgt + (int64_t)b * 2 * T + T};
        for (int64_t t = 0; t < T; ++t) {
            double s[6];
            for (int i = 0; i < 6; ++i) s[i] = (double)sig[i][t];
            for (int i = 0; i < 6; ++i) {
                q.m[i] += s[i];
                q.m[6 + i] += s[i] * s[i];
            }
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j) {
                    q.m[12 + i * 2 + j] += s[i] * s[4 + j];
                    q.m[16 + i * 2 + j] += s[i] * s[2 + j];
                    q.m[20 + i * 2 + j] += s[2 + i] * s[4 + j];
                }
        }
        const double* m = q.m;
        double nl_ft[2][2], nl_et[2][2];
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j) {
                q.st[i][j] = pair_sdr(m[i], m[4 + j], m[6 + i], m[10 + j], m[12 + i * 2 + j], Td);
                q.sf[i][j] = pair_sdr(m[i], m[2 + j], m[6 + i], m[8 + j], m[16 + i * 2 + j], Td);
                const PairSdr ft = pair_sdr(m[2 + i], m[4 + j], m[8 + i], m[10 + j], m[20 + i * 2 + j], Td);
                nl_ft[i][j] = -10.0 * log10(ft.sdr + kEps);
                nl_et[i][j] = -10.0 * log10(q.st[i][j].sdr + kEps);
            }
        // PIT (pw_mtx): p0 = (0,1), p1 = (1,0); torch.min keeps the first on ties
        const double lf0 = 0.5 * (nl_ft[0][0] + nl_ft[1][1]), lf1 = 0.5 * (nl_ft[1][0] + nl_ft[0][1]);
        const double le0 = 0.5 * (nl_et[0][0] + nl_et[1][1]), le1 = 0.5 * (nl_et[1][0] + nl_et[0][1]);
        const double sdrs = lf1 < lf0 ? lf1 : lf0, sdrqs = le1 < le0 ? le1 : le0;
        q.w = pow(10.0, (sdrs - sdrqs) / 10.0);     // mysystem.py:141
        const double t0 = -0.5 * (q.st[0][0].sdr + q.st[1][1].sdr), t1 = -0.5 * (q.st[1][0].sdr + q.st[0][1].sdr);
        q.pt = t1 < t0 ? 1 : 0;
        q.task = -(q.pt ? t1 : t0);
        const double k0 = -0.5 * q.w * (q.sf[0][0].sdr + q.sf[1][1].sdr), k1 = -0.5 * q.w * (q.sf[1][0].sdr + q.sf[0][1].sdr);
        q.pf = k1 < k0 ? 1 : 0;
        q.kd = -(q.pf ? k1 : k0);
        w_out[b] = (float)q.w;
        sisdr_out[b] = (float)(-sdrqs);
        task += q.task;
        kd += q.kd;
    }
    task /= (double)B;
    kd /= (double)B;
    const double lam = (double)kd_lambda, arg = (1.0 - lam) * task + lam * kd + kEps;
    out[0] = (float)(-10.0 * log10(arg));
    out[1] = (float)(-10.0 * log10(kd + kEps));
    out[2] = (float)task;
    out[3] = (float)kd;
    if (gest) {
        const double dL = -10.0 / (log(10.0) * arg);
        for (int b = 0; b < B; ++b) {
            const Smp& q = S[(size_t)b];
            const double gt = dL * (1.0 - lam) / (2.0 * (double)B), gk = dL * lam * q.w / (2.0 * (double)B);
            for (int i = 0; i < 2; ++i) {
                const int jt = q.pt ? 1 - i : i, jf = q.pf ? 1 - i : i;
                const float A = (float)(gt * q.st[i][jt].cx + gk * q.sf[i][jf].cx), Bt = (float)(gt * q.st[i][jt].cy),
                            Bf = (float)(gk * q.sf[i][jf].cy);
                const float me = (float)(q.m[i] / Td), mt = (float)(q.m[4 + jt] / Td), mf = (float)(q.m[2 + jf] / Td);
                const float* e = est + ((int64_t)b * 2 + i) * T;
                const float* tt = tgt + ((int64_t)b * 2 + jt) * T;
                const float* ff = fest + ((int64_t)b * 2 + jf) * T;
                float* g = gest + ((int64_t)b * 2 + i) * T;
                for (int64_t t = 0; t < T; ++t) g[t] = A * (e[t] - me) + Bt * (tt[t] - mt) + Bf * (ff[t] - mf);
            }
        }
    }
    return FQSS_OK;
}

// ------------------------------------------------------------------------------------------------ clip + Adam (torch.optim.Adam, clip_grad_norm_)
int fqss_sumsq(const float* g, int64_t n, double* sumsq, fqss_stream_t) {
    REQUIRE(g && sumsq && n >= 0, "bad args");
    double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (int64_t i = 0; i < n; ++i) s += (double)g[i] * (double)g[i];
    *sumsq += s;
    return FQSS_OK;
}
int fqss_adam_clip(float* p, const float* g, float* m, float* v, int64_t n, const double* sumsq, float max_norm, float grad_scale, float lr,
                   float beta1, float beta2, float eps, int32_t* step_t, const int32_t* t0, float* gnorm_out, fqss_stream_t) {
    REQUIRE(p && g && m && v && sumsq && step_t && n >= 0, "bad args");
    const int tg = *step_t + 1;
    const double norm = sqrt(*sumsq) * (double)grad_scale;
    double coef = (double)max_norm / (norm + 1e-6);
    if (coef > 1.0) coef = 1.0;
    if (max_norm <= 0.0f) coef = 1.0;
    const float gs = (float)((double)grad_scale * coef);
    const float omb1 = (float)(1.0 - (double)beta1), omb2 = (float)(1.0 - (double)beta2);
    int cached = -1;
    float step_size = 0.0f, bc2_sqrt = 1.0f;
    for (int64_t i = 0; i < n; ++i) {
        const int t = tg - (t0 ? t0[i] : 0);
        if (t <= 0) continue;
        if (t != cached) {
            const double bc1 = 1.0 - pow((double)beta1, (double)t), bc2 = 1.0 - pow((double)beta2, (double)t);
            step_size = (float)((double)lr / bc1);
            bc2_sqrt = (float)sqrt(bc2);
            cached = t;
        }
        const float gi = g[i] * gs;
        const float mi = m[i] + omb1 * (gi - m[i]);
        const float vi = v[i] * beta2 + (omb2 * gi) * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] + ((-step_size) * mi) / denom;
    }
    *step_t = tg;
    if (gnorm_out) *gnorm_out = (float)norm;
    return FQSS_OK;
}

// ------------------------------------------------------------------------------------------------ data side (LibriMix dataset, librimix_dataset.py:54, 139-153)
// the op sequences of csrc/data_ops.hip: energies in fp64, gains in fp32 (process.py:77-103), max_clip(0.9) (process.py:57-62)
int fqss_snr_mix(const float* a, const float* b, const float* snr, double* ws, uint32_t* peak, float* out, int64_t B, int64_t T, int64_t ld_a,
                 int64_t ld_b, int64_t ld_o, int mode, int clip, fqss_stream_t) {
    REQUIRE(a && b && snr && ws && peak && out, "null pointer");
    REQUIRE(B > 0 && T > 0 && ld_a >= T && ld_b >= T && ld_o >= T && (mode == 0 || mode == 1), "bad shape");
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < B; ++r) {
        double ea = 0.0, eb = 0.0;
        for (int64_t i = 0; i < T; ++i) {
            const double x = a[r * ld_a + i], y = b[r * ld_b + i];
            ea += x * x;
            eb += y * y;
        }
        ws[2 * r] += ea;
        ws[2 * r + 1] += eb;
        const float Ea = (float)(ws[2 * r] / (double)T), Eb = (float)(ws[2 * r + 1] / (double)T);
        float ga = 1.0f, gb = 1.0f;
        if (mode == 0) {
            if (Ea > 0.0f && Eb > 0.0f) {
                const float now = 10.0f * log10f(Ea / Eb);
                if (now < snr[r]) gb = sqrtf((Ea / Eb) * powf(10.0f, -snr[r] / 10.0f));
                else ga = sqrtf((Eb / Ea) * powf(10.0f, snr[r] / 10.0f));
            }
        } else if (Ea > 0.0f) {
            gb = sqrtf((Ea / Eb) / powf(10.0f, snr[r] / 10.0f));
        }
        float mx = 0.0f;
        for (int64_t i = 0; i < T; ++i) {
            const float m = a[r * ld_a + i] * ga + b[r * ld_b + i] * gb;
            out[r * ld_o + i] = m;
            mx = fmaxf(mx, fabsf(m));
        }
        memcpy(&peak[r], &mx, 4);
        if (clip && mx >= 0.9f) {
            const float gain = 0.9f / mx;
            for (int64_t i = 0; i < T; ++i) out[r * ld_o + i] *= gain;
        }
    }
    return FQSS_OK;
}

// y[r][n * newf + p] = sum_k h[p][k] * xpad[r][n * orig + k]: one fma per tap, in tap order (the HIP kernel's chain)
int fqss_resample_fir(const float* x, const float* h, float* y, int64_t rows, int64_t L, int64_t Lout, int64_t ld_x, int64_t ld_y, int orig,
                      int newf, int width, fqss_stream_t) {
    if (rows == 0 || Lout == 0) return FQSS_OK;
    REQUIRE(x && h && y, "null pointer");
    REQUIRE(rows > 0 && L > 0 && Lout > 0 && ld_x >= L && ld_y >= Lout && orig > 0 && newf > 0 && width >= 0, "bad shape");
    REQUIRE(Lout <= ((int64_t)newf * L + orig - 1) / orig, "output longer than ceil(new * L / orig)");
    const int K = 2 * width + orig;
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t o = 0; o < Lout; ++o) {
            const int64_t n = o / newf, base = n * orig - width;
            const int p = (int)(o - n * newf);
            float acc = 0.0f;
            for (int k = 0; k < K; ++k) {
                const int64_t i = base + k;
                acc = fmaf(h[p * K + k], (i >= 0 && i < L) ? x[r * ld_x + i] : 0.0f, acc);
            }
            y[r * ld_y + o] = acc;
        }
    return FQSS_OK;
}

}  // extern "C"
