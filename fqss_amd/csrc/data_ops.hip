// data_ops.hip -- the arithmetic of the data side that lives in the reference's own files: the SNR
// augmentation of the LibriMix dataset (process.py:57-103 generate_2mix_snr / generate_mix_noise / max_clip, train_utils.py:30-52),
// batched on the device: one (a, b, snr) triple per row.  File reading and resampling (soundfile, torchaudio) are third party.
//   mode 0 (generate_2mix_snr): the louder-than-requested signal is attenuated:  snr_now = 10 log10(Ea / Eb);
//           snr_now < snr ? b *= sqrt((Ea / Eb) 10^(-snr/10)) : a *= sqrt((Eb / Ea) 10^(snr/10));   untouched when an energy is 0
//   mode 1 (generate_mix_noise):  b *= sqrt((Ea / Eb) / 10^(snr/10))   (gain 1 when Ea = 0)
//   then mix = a + b and max_clip: max|mix| >= 0.9 -> mix *= 0.9 / max|mix|.
// Three streams (energies, mix + peak, clip); the per-row scalars never leave the device.
#include "fqss_dev.h"

namespace fqss {

__global__ __launch_bounds__(256) void k_row_energies(const float* __restrict__ a, const float* __restrict__ b, double* __restrict__ ws, int64_t T,
                                                       int64_t ld_a, int64_t ld_b) {
    __shared__ double smem[2 * 4];
    const int64_t r = blockIdx.y;
    double v[2] = {0.0, 0.0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < T; i += (int64_t)gridDim.x * 256) {
        const double x = a[r * ld_a + i], y = b[r * ld_b + i];
        v[0] += x * x;
        v[1] += y * y;
    }
    block_sum<double, 2>(v, smem);
    if (threadIdx.x == 0) { atomicAdd(ws + 2 * r, v[0]); atomicAdd(ws + 2 * r + 1, v[1]); }
}

__device__ __forceinline__ void snr_gains(const double* ws, int64_t r, int64_t T, float snr, int mode, float& ga, float& gb) {
    const float Ea = (float)(ws[2 * r] / (double)T), Eb = (float)(ws[2 * r + 1] / (double)T);
    ga = gb = 1.0f;
    if (mode == 0) {
        if (Ea > 0.0f && Eb > 0.0f) {
            const float now = 10.0f * log10f(Ea / Eb);
            if (now < snr) gb = sqrtf((Ea / Eb) * powf(10.0f, -snr / 10.0f));
            else ga = sqrtf((Eb / Ea) * powf(10.0f, snr / 10.0f));
        }
    } else if (Ea > 0.0f) {
        gb = sqrtf((Ea / Eb) / powf(10.0f, snr / 10.0f));
    }
}

__global__ __launch_bounds__(256) void k_snr_mix(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ snr,
                                                  const double* __restrict__ ws, float* __restrict__ out, uint32_t* __restrict__ peak, int64_t T,
                                                  int64_t ld_a, int64_t ld_b, int64_t ld_o, int mode) {
    const int64_t r = blockIdx.y;
    float ga, gb;
    snr_gains(ws, r, T, snr[r], mode, ga, gb);
    float mx = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < T; i += (int64_t)gridDim.x * 256) {
        const float m = a[r * ld_a + i] * ga + b[r * ld_b + i] * gb;
        out[r * ld_o + i] = m;
        mx = fmaxf(mx, fabsf(m));
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) atomicMax(peak + r, __float_as_uint(mx));      // non-negative floats order like their bit patterns
}

__global__ __launch_bounds__(256) void k_max_clip(float* __restrict__ x, const uint32_t* __restrict__ peak, int64_t T, int64_t ld, float max_check,
                                                   float max_clip) {
    const int64_t r = blockIdx.y;
    const float mx = __uint_as_float(peak[r]);
    if (!(mx >= max_check)) return;
    const float gain = max_clip / mx;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < T; i += (int64_t)gridDim.x * 256) x[r * ld + i] *= gain;
}

// Polyphase sinc resampler (librimix_dataset.py:54, 111-165: torchaudio.transforms.Resample(16000, 8000) on every clip read):
//   y[r][n * nw + p] = sum_k h[p][k] * xpad[r][n * og + k],  xpad = x with `width` zeros in front (and zeros behind), K = 2 width + og
// with og / nw the reduced rate ratio (2 / 1 for 16 -> 8 kHz).  The taps h (sinc * Hann window, torchaudio's published
// _get_sinc_resample_kernel) are computed on the host in fp64; the stream is read once (the K-fold tap overlap is served by L1 / L2).
__global__ __launch_bounds__(256) void k_resample_fir(const float* __restrict__ x, const float* __restrict__ h, float* __restrict__ y,
                                                       int64_t L, int64_t Lout, int64_t ld_x, int64_t ld_y, int og, int nw, int width, int K) {
    const int64_t r = blockIdx.y;
    const float* xr = x + r * ld_x;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < Lout; o += (int64_t)gridDim.x * 256) {
        const int64_t n = o / nw;
        const int p = (int)(o - n * nw);
        const int64_t base = n * og - width;
        float acc = 0.0f;
        for (int k = 0; k < K; ++k) {
            const int64_t i = base + k;
            const float v = (i >= 0 && i < L) ? xr[i] : 0.0f;
            acc = fmaf(h[p * K + k], v, acc);
        }
        y[r * ld_y + o] = acc;
    }
}

}  // namespace fqss

using namespace fqss;

extern "C" int fqss_resample_fir(const float* x, const float* h, float* y, int64_t rows, int64_t L, int64_t Lout, int64_t ld_x,
                                 int64_t ld_y, int orig, int newf, int width, fqss_stream_t stream) {
    if (rows == 0 || Lout == 0) return FQSS_OK;
    FQSS_REQUIRE(x && h && y, "null pointer");
    FQSS_REQUIRE(rows > 0 && rows <= 65535 && L > 0 && Lout > 0 && ld_x >= L && ld_y >= Lout && orig > 0 && newf > 0 && width >= 0, "bad shape");
    FQSS_REQUIRE(Lout <= (newf * L + orig - 1) / orig, "output longer than ceil(new * L / orig)");
    int64_t gx = cdiv(Lout, 1024);
    if (gx > 512) gx = 512;
    hipLaunchKernelGGL(k_resample_fir, dim3((unsigned)gx, (unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, h, y, L, Lout, ld_x, ld_y,
                       orig, newf, width, 2 * width + orig);
    return launch_status("fqss_resample_fir");
}

// a, b, out: B rows of T samples; snr [B] (dB) on the device; ws: 2*B doubles and peak: B uint32, both zeroed by the caller
extern "C" int fqss_snr_mix(const float* a, const float* b, const float* snr, double* ws, uint32_t* peak, float* out, int64_t B, int64_t T,
                            int64_t ld_a, int64_t ld_b, int64_t ld_o, int mode, int clip, fqss_stream_t stream) {
    FQSS_REQUIRE(a && b && snr && ws && peak && out, "null pointer");
    FQSS_REQUIRE(B > 0 && B <= 65535 && T > 0 && ld_a >= T && ld_b >= T && ld_o >= T && (mode == 0 || mode == 1), "bad shape");
    int64_t gx = cdiv(T, 2048);
    if (gx > 256) gx = 256;
    dim3 grid((unsigned)gx, (unsigned)B);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_row_energies, grid, dim3(256), 0, s, a, b, ws, T, ld_a, ld_b);
    hipLaunchKernelGGL(k_snr_mix, grid, dim3(256), 0, s, a, b, snr, ws, out, peak, T, ld_a, ld_b, ld_o, mode);
    if (clip) hipLaunchKernelGGL(k_max_clip, grid, dim3(256), 0, s, out, peak, T, ld_o, 0.9f, 0.9f);
    return launch_status("fqss_snr_mix");
}
