// codec.hip -- K10-K13: the waveform side of the network.
//   splitter2        16-bit waveform -> 2 x 8-bit channels (global max-normalise + floor quantizer)
//   frames_conv_fwd  strided framing conv (encoder fwd; also the decoder's input gradient)
//   ola_convtr_fwd   transposed conv = per-frame GEMV + overlap-add staged in LDS
//                    (decoder fwd; also the residual encoder's input gradient)
// Waveform frames are read with coalesced loads (lane <-> frame), filter taps are wave-uniform
// scalar loads, and the overlap-add is resolved inside an LDS tile so no global atomics are needed.
//
// Reference replaced: process.preprocess (process.py:16-37); F.conv1d(k=16,s=8) of Conv1dEncoderQ /
// ResidualErrorBlock (qat_layers.py:1028-1039, 1189-1192); F.conv_transpose1d of ConvTr1dDecoderQ /
// ResidualErrorBlock (qat_layers.py:1330-1341, 1194-1202) and their autograd.
#include "fqss_dev.h"

namespace fqss {

// process.py:10-14 with threshold=1, n_bits=8, sign=True
__device__ __forceinline__ float split_q(float x) {
    const float delta = 0.0078125f;
    return fminf(fmaxf(floorf(x / delta), -128.0f), 127.0f) * delta;
}

__global__ __launch_bounds__(256) void k_splitter2(const float* __restrict__ x, float* __restrict__ out, int B,
                                                    int64_t T, const uint32_t* obs) {
    const float mn = ord2f(obs[0]), mx = ord2f(obs[1]);
    const float thr = fmaxf(fabsf(mn), fabsf(mx));  // max(abs(x.min()), abs(x.max()))  process.py:24
    const float delta = 0.0078125f;
    const int64_t n = (int64_t)B * T;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / T, t = i - b * T;
        const float v = x[i] / thr;
        const float q0 = split_q(v);
        const float r = ((2.0f * (v - q0)) * 1.0f) / delta - 1.0f;  // process.py:35 op order
        out[(b * 2 + 0) * T + t] = q0;
        out[(b * 2 + 1) * T + t] = split_q(r);
    }
}

// preprocess(x, n_splitter=2, normalize=False) (process.py:26-36; the time branch of HTDemucsQ.pre_process): the threshold is
// max|x| and x is NOT divided by it; delta = thr / 128
__global__ __launch_bounds__(256) void k_splitter2_raw(const float* __restrict__ x, float* __restrict__ out, int B, int64_t T,
                                                        const uint32_t* obs) {
    const float mn = ord2f(obs[0]), mx = ord2f(obs[1]);
    const float thr = fmaxf(fabsf(mn), fabsf(mx));
    const float delta = thr / 128.0f;
    const int64_t n = (int64_t)B * T;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / T, t = i - b * T;
        const float v = x[i];
        const float q0 = fminf(fmaxf(floorf(v / delta), -128.0f), 127.0f) * delta;
        const float r = ((2.0f * (v - q0)) * thr) / delta - thr;      // process.py:35 op order
        out[(b * 2 + 0) * T + t] = q0;
        out[(b * 2 + 1) * T + t] = fminf(fmaxf(floorf(r / delta), -128.0f), 127.0f) * delta;
    }
}

// z[n][co][m] = sum_{ci,k} w[co][ci][k] * x[n][ci][m*STRIDE + k] ; block = 64 frames x all Co (4 waves split Co)
template <int CI, int K>
__global__ __launch_bounds__(256) void k_frames_conv_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                          float* __restrict__ z, int Co, int64_t T, int stride, int M,
                                                          int64_t ld_z) {
    constexpr int CK = CI * K;
    const int n = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = blockIdx.x * 64 + lane;
    const int mc = m < M ? m : M - 1;  // clamp: out-of-range lanes read a valid frame and do not store
    float xr[CK];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) {
        const float* xp = x + ((int64_t)n * CI + ci) * T + (int64_t)mc * stride;
#pragma unroll
        for (int k = 0; k < K; ++k) xr[ci * K + k] = xp[k];
    }
    for (int co = wave; co < Co; co += 4) {
        const float* wr = w + (int64_t)co * CK;  // wave-uniform -> scalar loads
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < CK; ++j) acc = fmaf(xr[j], wr[j], acc);
        if (m < M) z[((int64_t)n * Co + co) * ld_z + m] = acc;
    }
}

// out[n][t] = sum_c sum_{m*S+k=t} x[n][c][m] * w[c][k] ; R = K/S frames overlap on every output slot.
// A block owns FB = 64-(R-1) output slots (S samples each); its 64 lanes hold the 64 frames that
// touch them; the 4 waves split the channel reduction; partial P[k][frame] go through LDS.
template <int K, int S>
__global__ __launch_bounds__(256) void k_ola_convtr_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                         float* __restrict__ out, int C, int M, int64_t ld_x, int64_t T) {
    constexpr int R = K / S;
    constexpr int FB = 64 - (R - 1);
    constexpr int PL = 65;  // padded frame dimension
    __shared__ float P[4][K][PL];
    const int n = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q0 = blockIdx.x * FB;      // first output slot of this block
    const int m = q0 - (R - 1) + lane;   // frame held by this lane
    const bool mv = (m >= 0) && (m < M);
    float acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0.0f;
    const float* xn = x + (int64_t)n * C * ld_x;
    for (int c = wave; c < C; c += 4) {
        const float xv = mv ? xn[(int64_t)c * ld_x + m] : 0.0f;
        const float* wr = w + (int64_t)c * K;  // wave-uniform
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = fmaf(xv, wr[k], acc[k]);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) P[wave][k][lane] = acc[k];
    __syncthreads();
    const int nslots = M + R - 1;
    for (int e = threadIdx.x; e < FB * S; e += 256) {
        const int ql = e / S, j = e - ql * S;
        const int q = q0 + ql;
        if (q < nslots) {
            float v = 0.0f;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int fl = ql + (R - 1) - r;  // lane index of frame q - r
#pragma unroll
                for (int wv = 0; wv < 4; ++wv) v += P[wv][r * S + j][fl];
            }
            const int64_t t = (int64_t)q * S + j;
            if (t < T) out[(int64_t)n * T + t] = v;
        }
    }
}

}  // namespace fqss

using namespace fqss;

extern "C" int fqss_splitter2(const float* x, float* out, int B, int64_t T, const uint32_t* obs_ws,
                              fqss_stream_t stream) {
    FQSS_REQUIRE(x && out && obs_ws && B >= 0 && T >= 0, "bad args");
    if (B == 0 || T == 0) return FQSS_OK;
    int64_t nb = cdiv((int64_t)B * T, 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_splitter2, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, out, B, T, obs_ws);
    return launch_status("fqss_splitter2");
}

extern "C" int fqss_splitter2_raw(const float* x, float* out, int B, int64_t T, const uint32_t* obs_ws, fqss_stream_t stream) {
    FQSS_REQUIRE(x && out && obs_ws, "null tensor");
    FQSS_REQUIRE(B >= 0 && T >= 0, "bad shape");
    if (B == 0 || T == 0) return FQSS_OK;
    int64_t nb = cdiv((int64_t)B * T, 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_splitter2_raw, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, out, B, T, obs_ws);
    return launch_status("fqss_splitter2_raw");
}

extern "C" int fqss_frames_conv_fwd(const float* x, const float* w, float* z, int N, int Ci, int Co, int64_t T, int K,
                                    int stride, int M, int64_t ld_z, fqss_stream_t stream) {
    FQSS_REQUIRE(x && w && z, "null tensor");
    FQSS_REQUIRE(N >= 0 && N <= 65535 && Ci > 0 && Co > 0 && K > 0 && stride > 0 && M >= 0 && ld_z >= M, "bad shape");
    FQSS_REQUIRE(M == 0 || (int64_t)(M - 1) * stride + K <= T, "frames exceed the signal");
    if (N == 0 || M == 0) return FQSS_OK;
    dim3 grid((unsigned)cdiv(M, 64), (unsigned)N), block(256);
    hipStream_t s = (hipStream_t)stream;
#define FQSS_FC(CI_, K_) hipLaunchKernelGGL((k_frames_conv_fwd<CI_, K_>), grid, block, 0, s, x, w, z, Co, T, stride, M, ld_z)
    if (Ci == 1 && K == 16) FQSS_FC(1, 16);
    else if (Ci == 2 && K == 16) FQSS_FC(2, 16);
    else if (Ci == 1 && K == 32) FQSS_FC(1, 32);
    else if (Ci == 2 && K == 32) FQSS_FC(2, 32);
    else if (Ci == 1 && K == 2) FQSS_FC(1, 2);      // DPTNet encoder: 2-sample window, hop 1 (dptnetq.py:116)
    else if (Ci == 2 && K == 2) FQSS_FC(2, 2);
    else {
        set_error("fqss_frames_conv_fwd: unsupported (Ci=%d, K=%d); built for Ci in {1,2}, K in {2,16,32}", Ci, K);
        return FQSS_EINVAL;
    }
#undef FQSS_FC
    return launch_status("fqss_frames_conv_fwd");
}

extern "C" int fqss_ola_convtr_fwd(const float* x, const float* w, float* out, int N, int C, int M, int64_t ld_x, int K,
                                   int stride, int64_t T, fqss_stream_t stream) {
    FQSS_REQUIRE(x && w && out, "null tensor");
    FQSS_REQUIRE(N >= 0 && N <= 65535 && C > 0 && M >= 0 && ld_x >= M && K > 0 && stride > 0, "bad shape");
    FQSS_REQUIRE(M == 0 || T == (int64_t)(M - 1) * stride + K, "T must equal (M-1)*stride + K");
    if (N == 0 || M == 0) return FQSS_OK;
    hipStream_t s = (hipStream_t)stream;
    if (K == 16 && stride == 8) {
        dim3 grid((unsigned)cdiv(M + 1, 63), (unsigned)N);
        hipLaunchKernelGGL((k_ola_convtr_fwd<16, 8>), grid, dim3(256), 0, s, x, w, out, C, M, ld_x, T);
    } else if (K == 32 && stride == 16) {
        dim3 grid((unsigned)cdiv(M + 1, 63), (unsigned)N);
        hipLaunchKernelGGL((k_ola_convtr_fwd<32, 16>), grid, dim3(256), 0, s, x, w, out, C, M, ld_x, T);
    } else if (K == 2 && stride == 1) {   // DPTNet's 2-sample window, hop 1 (input gradient of its encoder)
        dim3 grid((unsigned)cdiv(M + 1, 63), (unsigned)N);
        hipLaunchKernelGGL((k_ola_convtr_fwd<2, 1>), grid, dim3(256), 0, s, x, w, out, C, M, ld_x, T);
    } else {
        set_error("fqss_ola_convtr_fwd: unsupported (K=%d, stride=%d); built for (2,1), (16,8) and (32,16)", K, stride);
        return FQSS_EINVAL;
    }
    return launch_status("fqss_ola_convtr_fwd");
}
