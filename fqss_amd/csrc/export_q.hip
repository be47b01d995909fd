// export_q.hip -- the affine (scale, zero-point) form of the learned quantizers: the arithmetic of
// torch.fake_quantize_per_tensor_affine / per_channel_affine that the reference's export wrappers TorchWeightFakeQuantize /
// TorchActivationFakeQuantize / TorchDymActivationFakeQuantize apply (qat_quant.py:15-72):
//     q = clamp(zero_point + nearbyint(x * (1 / scale)), qmin, qmax);   y = (q - zero_point) * scale
// with the integer codes q written next to y (int8 weights / uint8 activations of a true-integer deployment).  One HBM stream.
#include "fqss_dev.h"

namespace fqss {

// x viewed as [outer][C][inner]; scale / zp per channel c (C = 1: per tensor)
__global__ __launch_bounds__(256) void k_fq_affine(const float* __restrict__ x, float* __restrict__ y, int* __restrict__ codes,
                                                    int64_t outer, int64_t C, int64_t inner, const float* __restrict__ scale,
                                                    const int* __restrict__ zp, int qmin, int qmax) {
    const int64_t n = outer * C * inner;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t c = (i / inner) % C;
        const float sc = scale[c];
        const float inv = 1.0f / sc;
        const int z = zp[c];
        int64_t q = (int64_t)z + (int64_t)nearbyintf(x[i] * inv);
        q = q < qmin ? qmin : (q > qmax ? qmax : q);
        if (y != nullptr) y[i] = (float)(q - z) * sc;
        if (codes != nullptr) codes[i] = (int)q;
    }
}

}  // namespace fqss

using namespace fqss;

extern "C" int fqss_fq_affine(const float* x, float* y, int* codes, int64_t outer, int64_t C, int64_t inner, const float* scale, const int* zp,
                              int qmin, int qmax, fqss_stream_t stream) {
    FQSS_REQUIRE(x && scale && zp && (y || codes), "null pointer");
    FQSS_REQUIRE(outer > 0 && C > 0 && inner > 0 && qmin < qmax, "bad shape");
    int64_t nb = cdiv(outer * C * inner, 1024);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(k_fq_affine, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, y, codes, outer, C, inner, scale, zp, qmin, qmax);
    return launch_status("fqss_fq_affine");
}
