/*
 * oracle/fq_core.c -- TEST INFRASTRUCTURE ONLY (never linked into the product path).
 *
 * Plain-C, fp32, scalar restatement of the reference's uniform quantizers.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this file.
 *
 * Reference followed (ssi-research/FQSS @ 2024_10_08):
 *   quantization/qat/qat_quant.py:125-147   linear_quantize (sym branch :126-135, asym :136-147)
 *   quantization/qat/qat_quant.py:88-103    round_ste / grad_scale (identity when scale_grad=False)
 *   process.py:10-14                        quantize (floor quantizer of the splitter)
 *   process.py:16-37 / 39-52                preprocess / postprocess (n_splitter = n_combiner = 2)
 * Backward formulas are the autograd of those op sequences, accumulated in double.
 *
 * Pinned against golden vectors generated from the reference itself: tests/golden/fq_act.npz,
 * fq_w.npz, process.npz (tools/make_goldens.py), see tests/test_oracle_goldens.py.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile): every float op below is one
 * IEEE-754 binary32 operation, like the ATen ops it restates.
 */
#include <math.h>
#include <stdint.h>

/* y = delta*clip(rint((x-min)/delta),0,255)+min ; delta = (max-min)/255      qat_quant.py:139-146 */
void fqo_act_fwd(const float* x, int64_t n, float lo, float hi, float* y, uint8_t* idx) {
    const float delta = (hi - lo) / 255.0f;
    for (int64_t i = 0; i < n; ++i) {
        float u = (x[i] - lo) / delta;
        float X = rintf(u); /* torch.round = round-half-to-even */
        float c = fminf(fmaxf(X, 0.0f), 255.0f);
        if (y) y[i] = delta * c + lo;
        if (idx) idx[i] = (uint8_t)c;
    }
}

/* autograd of the op sequence above, element-wise part restated op by op:
 *   g_C = g*delta ; g_u = g_C*m ; gx = g_u/delta        (so gx == g only up to two fp32 roundings)
 *   gmax = sum g*(c-m*u)/255 ; gmin = sum g*((1-m) - (c-m*u)/255)          (sums in double) */
void fqo_act_bwd(const float* x, const float* g, int64_t n, float lo, float hi, float* gx,
                 double* gmin, double* gmax) {
    const float delta = (hi - lo) / 255.0f;
    double smin = 0.0, smax = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        float u = (x[i] - lo) / delta;
        float X = rintf(u);
        int m = (X >= 0.0f) && (X <= 255.0f);
        float c = fminf(fmaxf(X, 0.0f), 255.0f);
        float t = m ? (c - u) : c; /* c - m*u */
        if (gx) gx[i] = m ? (g[i] * delta) / delta : 0.0f;
        smax += (double)g[i] * ((double)t / 255.0);
        smin += (double)g[i] * ((m ? 0.0 : 1.0) - (double)t / 255.0);
    }
    *gmin = smin;
    *gmax = smax;
}

/* per-channel symmetric: a=max(|min_c|,|max_c|), delta=2a/255, y=delta*clip(rint(w/delta),-128,127)
 * tensor layout [outer][C][inner]; ch_out_idx=0 -> outer=1; ch_out_idx=1 -> outer=shape[0]
 * qat_quant.py:127-135 */
void fqo_w_fwd(const float* w, int64_t outer, int64_t C, int64_t inner, const float* lo,
               const float* hi, float* y, int8_t* idx) {
    for (int64_t o = 0; o < outer; ++o)
        for (int64_t c = 0; c < C; ++c) {
            float a = fmaxf(fabsf(lo[c]), fabsf(hi[c]));
            float delta = (2.0f * a) / 255.0f;
            for (int64_t i = 0; i < inner; ++i) {
                int64_t k = (o * C + c) * inner + i;
                float X = rintf(w[k] / delta);
                float q = fminf(fmaxf(X, -128.0f), 127.0f);
                if (y) y[k] = delta * q;
                if (idx) idx[k] = (int8_t)q;
            }
        }
}

void fqo_w_bwd(const float* w, const float* g, int64_t outer, int64_t C, int64_t inner,
               const float* lo, const float* hi, float* gw, float* gmin, float* gmax) {
    for (int64_t c = 0; c < C; ++c) {
        float a = fmaxf(fabsf(lo[c]), fabsf(hi[c]));
        float delta = (2.0f * a) / 255.0f;
        double D = 0.0;
        for (int64_t o = 0; o < outer; ++o)
            for (int64_t i = 0; i < inner; ++i) {
                int64_t k = (o * C + c) * inner + i;
                float u = w[k] / delta;
                float X = rintf(u);
                int m = (X >= -128.0f) && (X <= 127.0f);
                float q = fminf(fmaxf(X, -128.0f), 127.0f);
                float t = m ? (q - u) : q;
                if (gw) gw[k] = m ? (g[k] * delta) / delta : 0.0f;
                D += (double)g[k] * (double)t;
            }
        D *= 2.0 / 255.0;
        /* torch.maximum splits the gradient 1/2-1/2 on ties; abs' = sign (sign(0)=0) */
        float al = fabsf(lo[c]), ah = fabsf(hi[c]);
        double wl = al > ah ? 1.0 : (al == ah ? 0.5 : 0.0);
        double wh = ah > al ? 1.0 : (al == ah ? 0.5 : 0.0);
        double sl = lo[c] > 0 ? 1.0 : (lo[c] < 0 ? -1.0 : 0.0);
        double sh = hi[c] > 0 ? 1.0 : (hi[c] < 0 ? -1.0 : 0.0);
        gmin[c] = (float)(D * wl * sl);
        gmax[c] = (float)(D * wh * sh);
    }
}

/* process.py:10-14, threshold=1, n_bits=8, sign=True: clip(floor(x/delta),-128,127)*delta, delta=1/128 */
static inline float split_q(float x) {
    const float delta = 0.0078125f;
    return fminf(fmaxf(floorf(x / delta), -128.0f), 127.0f) * delta;
}

/* process.py:16-37 with n_splitter=2, normalize=True.  x: [B][T] -> out: [B][2][T]; thr = global absmax */
void fqo_splitter2(const float* x, int64_t B, int64_t T, float* out) {
    float mn = INFINITY, mx = -INFINITY;
    for (int64_t i = 0; i < B * T; ++i) { mn = fminf(mn, x[i]); mx = fmaxf(mx, x[i]); }
    float thr = fmaxf(fabsf(mn), fabsf(mx));
    const float delta = 0.0078125f;
    for (int64_t b = 0; b < B; ++b)
        for (int64_t t = 0; t < T; ++t) {
            float v = x[b * T + t] / thr;
            float q0 = split_q(v);
            float r = ((2.0f * (v - q0)) * 1.0f) / delta - 1.0f;
            out[(b * 2 + 0) * T + t] = q0;
            out[(b * 2 + 1) * T + t] = split_q(r);
        }
}

/* process.py:39-52 with n_combiner=2: y = x[0] + x[1]*(0.5/128) */
void fqo_combine2(const float* x0, const float* x1, int64_t n, float* y) {
    for (int64_t i = 0; i < n; ++i) y[i] = x0[i] + x1[i] * 0.00390625f;
}
