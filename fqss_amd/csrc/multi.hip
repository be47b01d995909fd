// multi.hip -- multi-tensor launches for the per-quantizer small work of a step.
// A ConvTasNetQ step has 200 activation quantizers and 101 weight quantizers; running their flush /
// fake-quant / STE kernels one launch each costs ~780 launches (~3 ms of GPU timeline).  Here each kind
// of work is ONE launch driven by a device-side descriptor table built once by the host
// (fqss_amd/runtime.py QuantTables).
//
//   fqss_gacc_flush_multi   all range/slope partial slots -> fp32 parameter gradients
//   fqss_wq_multi_fwd       every weight: per-channel symmetric fake-quant (+ int8 codes for pointwise convs)
//   fqss_wq_multi_bwd       every weight: STE + range gradients from the accumulated dL/dW_q
// Reference replaced: the per-module calls of GradientWeightFakeQuantize.forward (qat_quant.py:372-381)
// and the autograd of linear_quantize's range parameters (qat_quant.py:126-147).
#include "fqss_dev.h"

namespace fqss {

constexpr int kSlotsM = FQSS_GACC_SLOTS;
constexpr int kWqFields = FQSS_WQ_DESC_WORDS;   // int64 words per weight descriptor

// table[q] = {gacc, gmin, gmax, gslope} (addresses; 0 = absent)
__global__ __launch_bounds__(256) void k_gacc_flush_multi(const long long* __restrict__ table) {
    __shared__ double red[3 * 4];
    const long long* t = table + 4 * (int64_t)blockIdx.x;
    double* gacc = reinterpret_cast<double*>(t[0]);
    float* gmin = reinterpret_cast<float*>(t[1]);
    float* gmax = reinterpret_cast<float*>(t[2]);
    float* gslope = reinterpret_cast<float*>(t[3]);
    double v[3] = {0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < kSlotsM; i += 256) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double x = gacc[3 * i + k];
            if (x != 0.0) {
                v[k] += x;
                gacc[3 * i + k] = 0.0;
            }
        }
    }
    block_sum<double, 3>(v, red);
    if (threadIdx.x == 0) {
        if (gmin) *gmin += (float)v[0];
        if (gmax) *gmax += (float)v[1];
        if (gslope) *gslope += (float)v[2];
    }
}

// weight descriptor (int64 words): 0 w, 1 wq, 2 idx, 3 idxT, 4 dw, 5 rw, 6 qmin, 7 qmax, 8 gwq, 9 gw, 10 gmin, 11 gmax,
//                                  12 outer, 13 C, 14 inner, 15 first block, 16 row stride of idxT (>= C: the codes of
//                                  paired layers are concatenated along the output channels)
__device__ __forceinline__ const long long* find_desc(const long long* table, int n, int blk) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {   // last descriptor whose first block <= blk
        const int mid = (lo + hi + 1) >> 1;
        if ((int)table[(int64_t)mid * kWqFields + 15] <= blk) lo = mid; else hi = mid - 1;
    }
    return table + (int64_t)lo * kWqFields;
}

__global__ __launch_bounds__(256) void k_wq_multi_fwd(const long long* __restrict__ table, int n) {
    __shared__ float red[4];
    const long long* d = find_desc(table, n, blockIdx.x);
    const float* w = reinterpret_cast<const float*>(d[0]);
    float* wq = reinterpret_cast<float*>(d[1]);
    signed char* idx = reinterpret_cast<signed char*>(d[2]);
    signed char* idxT = reinterpret_cast<signed char*>(d[3]);
    float* dw = reinterpret_cast<float*>(d[4]);
    float* rw = reinterpret_cast<float*>(d[5]);
    const float* qmin = reinterpret_cast<const float*>(d[6]);
    const float* qmax = reinterpret_cast<const float*>(d[7]);
    const int outer = (int)d[12], C = (int)d[13], inner = (int)d[14];
    const int c = blockIdx.x - (int)d[15];
    const int64_t ldT = d[16];
    const float a = fmaxf(fabsf(qmin[c]), fabsf(qmax[c]));
    const float delta = (2.0f * a) / 255.0f;
    const float inv = 1.0f / delta;
    float s = 0.0f;
    const int nel = outer * inner;
    for (int e = threadIdx.x; e < nel; e += 256) {
        const int o = e / inner, i = e - o * inner;
        const int64_t k = ((int64_t)o * C + c) * inner + i;
        const float X = rintf(div_by(w[k], delta, inv));
        const float q = __builtin_amdgcn_fmed3f(X, -128.0f, 127.0f);
        wq[k] = delta * q;
        if (idx) {   // pointwise conv weight [C][outer*inner == Ci]: codes for the q-GEMMs
            idx[(int64_t)c * nel + e] = (signed char)q;
            idxT[(int64_t)e * ldT + c] = (signed char)q;
            s += q;
        }
    }
    if (idx) {
        float v[1] = {s};
        block_sum<float, 1>(v, red);
        if (threadIdx.x == 0) {
            dw[c] = delta;
            rw[c] = v[0];
        }
    }
}

__global__ __launch_bounds__(256) void k_wq_multi_bwd(const long long* __restrict__ table, int n) {
    __shared__ double red[4];
    const long long* d = find_desc(table, n, blockIdx.x);
    const float* w = reinterpret_cast<const float*>(d[0]);
    const float* qmin = reinterpret_cast<const float*>(d[6]);
    const float* qmax = reinterpret_cast<const float*>(d[7]);
    const float* gwq = reinterpret_cast<const float*>(d[8]);
    float* gw = reinterpret_cast<float*>(d[9]);
    float* gmin = reinterpret_cast<float*>(d[10]);
    float* gmax = reinterpret_cast<float*>(d[11]);
    const int outer = (int)d[12], C = (int)d[13], inner = (int)d[14];
    const int c = blockIdx.x - (int)d[15];
    const float lo = qmin[c], hi = qmax[c];
    const float a = fmaxf(fabsf(lo), fabsf(hi));
    const float delta = (2.0f * a) / 255.0f;
    const float inv = 1.0f / delta;
    float p = 0.0f;
    const int nel = outer * inner;
    for (int e = threadIdx.x; e < nel; e += 256) {
        const int o = e / inner, i = e - o * inner;
        const int64_t k = ((int64_t)o * C + c) * inner + i;
        const float u = div_by(w[k], delta, inv);
        const float X = rintf(u);
        const bool inr = (X >= -128.0f) && (X <= 127.0f);
        const float q = __builtin_amdgcn_fmed3f(X, -128.0f, 127.0f);
        const float gk = gwq[k];
        gw[k] += inr ? div_by(gk * delta, delta, inv) : 0.0f;
        p += gk * (inr ? (q - u) : q);
    }
    double v[1] = {(double)p};
    block_sum<double, 1>(v, red);
    if (threadIdx.x == 0) {
        const double D = v[0] * (2.0 / 255.0);
        const float al = fabsf(lo), ah = fabsf(hi);
        const double wl = al > ah ? 1.0 : (al == ah ? 0.5 : 0.0);
        const double wh = ah > al ? 1.0 : (al == ah ? 0.5 : 0.0);
        const double sl = lo > 0.0f ? 1.0 : (lo < 0.0f ? -1.0 : 0.0);
        const double sh = hi > 0.0f ? 1.0 : (hi < 0.0f ? -1.0 : 0.0);
        gmin[c] += (float)(D * wl * sl);
        gmax[c] += (float)(D * wh * sh);
    }
}

}  // namespace fqss

using namespace fqss;

extern "C" int fqss_gacc_flush_multi(const int64_t* table, int n, fqss_stream_t stream) {
    FQSS_REQUIRE(table && n >= 0, "bad args");
    if (n == 0) return FQSS_OK;
    hipLaunchKernelGGL(k_gacc_flush_multi, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, (const long long*)table);
    return launch_status("fqss_gacc_flush_multi");
}

extern "C" int fqss_wq_multi_fwd(const int64_t* table, int n, int total_channels, fqss_stream_t stream) {
    FQSS_REQUIRE(table && n >= 0 && total_channels >= 0, "bad args");
    if (n == 0 || total_channels == 0) return FQSS_OK;
    hipLaunchKernelGGL(k_wq_multi_fwd, dim3((unsigned)total_channels), dim3(256), 0, (hipStream_t)stream,
                       (const long long*)table, n);
    return launch_status("fqss_wq_multi_fwd");
}

extern "C" int fqss_wq_multi_bwd(const int64_t* table, int n, int total_channels, fqss_stream_t stream) {
    FQSS_REQUIRE(table && n >= 0 && total_channels >= 0, "bad args");
    if (n == 0 || total_channels == 0) return FQSS_OK;
    hipLaunchKernelGGL(k_wq_multi_bwd, dim3((unsigned)total_channels), dim3(256), 0, (hipStream_t)stream,
                       (const long long*)table, n);
    return launch_status("fqss_wq_multi_bwd");
}
