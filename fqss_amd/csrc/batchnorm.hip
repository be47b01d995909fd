// batchnorm.hip -- BatchNorm1d / BatchNorm2d under BatchNormQ (quantization/qat/qat_layers.py:472-486 of the reference: y = fq_act(
// batchnorm(x)); qat_utils.py:163, 381-382 map nn.BatchNorm1d/2d to it).  None of the five shipped configurations instantiates the
// layer (ConvTasNet / DPTNet / Sepformer / HTDemucs normalise per sample); it completes the quantization.qat module surface.
// Channel-first tensors [B][C][M] (M = H*W for the 2-D form), rows padded like every activation.  Four HBM streams:
//   fqss_bn_moments      per-channel (sum, sum of squares) in fp64                      -> batch statistics (training mode)
//   fqss_bn_apply        y = x * a[c] + b[c]    (a = invstd * gamma, b = beta - mean * a: ATen's alpha / beta form)
//   fqss_bn_bwd_reduce   per-channel (sum g, sum g x) in fp64                           -> gamma / beta gradients, the backward's coefficients
//   fqss_bn_bwd_apply    gx = g * c1[c] + x * c2[c] + c3[c]
// The C-sized arithmetic between them (mean, variance, running statistics, coefficients) is host-side torch on C-element tensors.
#include "fqss_dev.h"

namespace fqss {

// one workgroup per (channel, batch slice): fp32 loads, fp64 sums, one pair of atomics per workgroup
template <bool WITH_G>
__global__ __launch_bounds__(256) void k_bn_reduce(const float* __restrict__ x, const float* __restrict__ g, double* __restrict__ out, int B,
                                                    int C, int M, int64_t ld_x, int64_t ld_g) {
    __shared__ double red[2 * 4];
    const int c = blockIdx.x;
    double s0 = 0.0, s1 = 0.0;
    for (int b = blockIdx.y; b < B; b += gridDim.y) {
        const float* xr = x + ((int64_t)b * C + c) * ld_x;
        const float* gr = WITH_G ? g + ((int64_t)b * C + c) * ld_g : nullptr;
        for (int m = threadIdx.x; m < M; m += 256) {
            const double xv = (double)xr[m];
            if (WITH_G) {
                const double gv = (double)gr[m];
                s0 += gv;
                s1 += gv * xv;
            } else {
                s0 += xv;
                s1 += xv * xv;
            }
        }
    }
    double v[2] = {s0, s1};
    block_sum<double, 2>(v, red);
    if (threadIdx.x == 0) {
        atomicAdd(&out[2 * c], v[0]);
        atomicAdd(&out[2 * c + 1], v[1]);
    }
}

// y = p * a[c] + q * b[c] + d[c]  (q == nullptr: the two-operand form y = p * a[c] + d[c])
__global__ __launch_bounds__(256) void k_bn_affine(const float* __restrict__ p, const float* __restrict__ q, const float* __restrict__ a,
                                                    const float* __restrict__ b, const float* __restrict__ d, float* __restrict__ y, int C, int M,
                                                    int64_t ld_p, int64_t ld_q, int64_t ld_y) {
    const int row = blockIdx.y, c = row % C;
    const float ca = a[c], cd = d[c], cb = q ? b[c] : 0.0f;
    const float* pr = p + (int64_t)row * ld_p;
    const float* qr = q ? q + (int64_t)row * ld_q : nullptr;
    float* yr = y + (int64_t)row * ld_y;
    for (int m = blockIdx.x * 256 + threadIdx.x; m < M; m += gridDim.x * 256) {
        float t = pr[m] * ca;
        if (qr) t = t + qr[m] * cb;
        yr[m] = t + cd;
    }
}

}  // namespace fqss

using namespace fqss;

static dim3 bn_reduce_grid(int B, int C) {
    int gy = 1;
    while (gy < B && (int64_t)C * gy < 1024) gy *= 2;     // enough workgroups for small channel counts
    if (gy > B) gy = B;
    return dim3((unsigned)C, (unsigned)gy);
}

extern "C" int fqss_bn_moments(const float* x, double* out, int B, int C, int M, int64_t ld_x, fqss_stream_t stream) {
    FQSS_REQUIRE(x && out && B >= 0 && C > 0 && C <= 65535 && M >= 0 && ld_x >= M, "bad args");
    if (B == 0 || M == 0) return FQSS_OK;
    hipLaunchKernelGGL(k_bn_reduce<false>, bn_reduce_grid(B, C), dim3(256), 0, (hipStream_t)stream, x, (const float*)nullptr, out, B, C, M, ld_x, (int64_t)0);
    return launch_status("fqss_bn_moments");
}

extern "C" int fqss_bn_bwd_reduce(const float* g, const float* x, double* out, int B, int C, int M, int64_t ld_g, int64_t ld_x,
                                  fqss_stream_t stream) {
    FQSS_REQUIRE(g && x && out && B >= 0 && C > 0 && C <= 65535 && M >= 0 && ld_x >= M && ld_g >= M, "bad args");
    if (B == 0 || M == 0) return FQSS_OK;
    hipLaunchKernelGGL(k_bn_reduce<true>, bn_reduce_grid(B, C), dim3(256), 0, (hipStream_t)stream, x, g, out, B, C, M, ld_x, ld_g);
    return launch_status("fqss_bn_bwd_reduce");
}

static int bn_affine(const char* fn, const float* p, const float* q, const float* a, const float* b, const float* d, float* y, int B, int C, int M,
                     int64_t ld_p, int64_t ld_q, int64_t ld_y, fqss_stream_t stream) {
    if (!(p && a && d && y && B >= 0 && C > 0 && M >= 0 && ld_p >= M && ld_y >= M && (!q || (b && ld_q >= M)) && (int64_t)B * C <= 65535)) {
        set_error("%s: bad args", fn);
        return FQSS_EINVAL;
    }
    if (B == 0 || M == 0) return FQSS_OK;
    int gx = (int)cdiv(M, 256 * 4);
    if (gx < 1) gx = 1;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(k_bn_affine, dim3((unsigned)gx, (unsigned)(B * C)), dim3(256), 0, (hipStream_t)stream, p, q, a, b, d, y, C, M, ld_p, ld_q, ld_y);
    return launch_status(fn);
}

extern "C" int fqss_bn_apply(const float* x, const float* a, const float* b, float* y, int B, int C, int M, int64_t ld_x, int64_t ld_y,
                             fqss_stream_t stream) {
    return bn_affine("fqss_bn_apply", x, nullptr, a, nullptr, b, y, B, C, M, ld_x, 0, ld_y, stream);
}

extern "C" int fqss_bn_bwd_apply(const float* g, const float* x, const float* c1, const float* c2, const float* c3, float* gx, int B, int C,
                                 int M, int64_t ld_g, int64_t ld_x, int64_t ld_gx, fqss_stream_t stream) {
    return bn_affine("fqss_bn_bwd_apply", g, x, c1, c2, c3, gx, B, C, M, ld_g, ld_x, ld_gx, stream);
}
