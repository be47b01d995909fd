// stft.hip -- the spectrogram pair of HTDemucsQ: `_spec` and `_ispec` (htdemucsq.py:931-960) around
// demucs.spec.spectro / ispectro (third party, demucs~=4.0.0, absent from the reference tree: published algorithm =
// torch.stft / torch.istft with a periodic Hann window of n_fft points, hop n_fft/4, normalized=True, center=True, reflect padding).
//
// Both directions are ONE radix-2 FFT of n_fft (<= 4096) complex points per frame, held in LDS (32 KB at 4096): a workgroup owns a
// frame; the twiddles exp(-2 pi i k / N) come from a table the host computed in fp64.  The re-padding of `_spec` (reflect by 3/2 hop,
// frames 2 .. 2+le of the centred STFT, Nyquist bin dropped) is folded into the load indices, so the padded signal is never written;
// `_ispec` (zero Nyquist bin, 2 zero frames either side, overlap-add, division by the window envelope, crop) is a frame pass (inverse
// FFT x window) plus a deterministic gather-form overlap-add.  The backward of `_ispec` -- the loss sits on the waveform -- is the
// adjoint: crop / envelope / window, one forward FFT per frame, (2 - [k = 0]) / sqrt(N) scaling.
// Spectra are exchanged as separate real / imaginary planes [rows][2][T][N/2] (bins contiguous: coalesced); fqss_transpose2d turns
// them into the model's [.., 2, Fr, T] "complex as channels" layout and back.
#include "fqss_dev.h"

namespace fqss {

// in-place radix-2 DIT FFT of N = 2^logn points in LDS; data must have been stored in bit-reversed order.
// tw[k] = exp(-2 pi i k / N), k < N/2; inverse: conjugated twiddles (no 1/N).
__device__ __forceinline__ void fft_lds(float2* s, const float2* __restrict__ tw, int N, int logn, bool inverse) {
    for (int st = 0; st < logn; ++st) {
        const int half = 1 << st;
        const int tstep = N >> (st + 1);
        __syncthreads();
        for (int b = threadIdx.x; b < N / 2; b += blockDim.x) {
            const int pos = b & (half - 1);
            const int i = ((b >> st) << (st + 1)) + pos, j = i + half;
            float2 w = tw[pos * tstep];
            if (inverse) w.y = -w.y;
            const float2 a = s[i], c = s[j];
            const float2 t = make_float2(c.x * w.x - c.y * w.y, c.x * w.y + c.y * w.x);
            s[i] = make_float2(a.x + t.x, a.y + t.y);
            s[j] = make_float2(a.x - t.x, a.y - t.y);
        }
    }
    __syncthreads();
}
__device__ __forceinline__ int bitrev(int v, int logn) { return (int)(__brev((unsigned)v) >> (32 - logn)); }

// z[row][0/1][f][k] = (1/sqrt(N)) * FFT_k( w[n] * x2[f*hop + n] ),  k < N/2,  x2 = reflect-pad(x, pad) (`_spec`)
__global__ __launch_bounds__(256) void k_stft(const float* __restrict__ x, float* __restrict__ z, const float* __restrict__ win,
                                               const float2* __restrict__ tw, int N, int logn, int hop, int T, int pad, int64_t L, int64_t ld_x) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    const int f = blockIdx.x;
    const int64_t row = blockIdx.y;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        int64_t i = (int64_t)f * hop + n - pad;
        if (i < 0) i = -i;
        if (i >= L) i = 2 * (L - 1) - i;
        sm[bitrev(n, logn)] = make_float2(win[n] * x[row * ld_x + i], 0.f);
    }
    fft_lds(sm, tw, N, logn, false);
    const float sc = 1.0f / sqrtf((float)N);
    float* zr = z + ((row * 2 + 0) * T + f) * (int64_t)(N / 2);
    float* zi = z + ((row * 2 + 1) * T + f) * (int64_t)(N / 2);
    for (int k = threadIdx.x; k < N / 2; k += blockDim.x) {
        zr[k] = sm[k].x * sc;
        zi[k] = sm[k].y * sc;
    }
}

// fr[row][t][n] = w[n] * sqrt(N) * irfft(Z_t)[n],  Z_t = bins 0 .. N/2-1 of z[row][0/1][t][:], Nyquist bin = 0
__global__ __launch_bounds__(256) void k_istft_frames(const float* __restrict__ z, float* __restrict__ fr, const float* __restrict__ win,
                                                       const float2* __restrict__ tw, int N, int logn, int T) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    const int t = blockIdx.x;
    const int64_t row = blockIdx.y;
    const float* zr = z + ((row * 2 + 0) * T + t) * (int64_t)(N / 2);
    const float* zi = z + ((row * 2 + 1) * T + t) * (int64_t)(N / 2);
    for (int k = threadIdx.x; k < N / 2; k += blockDim.x) {
        const float re = zr[k], im = k == 0 ? 0.f : zi[k];      // irfft ignores the imaginary part of bin 0
        sm[bitrev(k, logn)] = make_float2(re, im);
        if (k > 0) sm[bitrev(N - k, logn)] = make_float2(re, -im);
        else sm[bitrev(N / 2, logn)] = make_float2(0.f, 0.f);
    }
    fft_lds(sm, tw, N, logn, true);
    const float sc = sqrtf((float)N) / (float)N;
    float* o = fr + (row * T + t) * (int64_t)N;
    for (int n = threadIdx.x; n < N; n += blockDim.x) o[n] = win[n] * (sm[n].x * sc);
}

// y[row][j] = sum_t fr[row][t][n - (t+2)*hop] / env[n % hop],  n = j + N/2 + pad  (2 zero frames precede frame 0; `_ispec`)
__global__ __launch_bounds__(256) void k_istft_ola(const float* __restrict__ fr, float* __restrict__ y, const float* __restrict__ env, int N, int hop,
                                                    int T, int pad, int64_t length, int64_t ld_y) {
    const int64_t row = blockIdx.y;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < length; j += (int64_t)gridDim.x * 256) {
        const int64_t n = j + N / 2 + pad;
        const int64_t tp = n / hop;                      // last padded frame that covers n
        float acc = 0.f;
        for (int k = N / hop - 1; k >= 0; --k) {         // ascending frame order
            const int64_t t = tp - k - 2;
            if (t < 0 || t >= T) continue;
            acc += fr[(row * T + t) * (int64_t)N + (n - (t + 2) * hop)];
        }
        y[row * ld_y + j] = acc / env[n % hop];
    }
}

// adjoint of k_istft_ola + k_istft_frames: gz[row][0/1][t][k] = c_k * FFT_k( w[m] * g[row][(t+2)*hop + m - N/2 - pad] / env ),
// c_0 = 1/sqrt(N) (imaginary part 0), c_k = 2/sqrt(N)
__global__ __launch_bounds__(256) void k_istft_bwd(const float* __restrict__ g, float* __restrict__ gz, const float* __restrict__ win,
                                                    const float* __restrict__ env, const float2* __restrict__ tw, int N, int logn, int hop, int T,
                                                    int pad, int64_t length, int64_t ld_g) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    const int t = blockIdx.x;
    const int64_t row = blockIdx.y;
    for (int m = threadIdx.x; m < N; m += blockDim.x) {
        const int64_t n = (int64_t)(t + 2) * hop + m, j = n - N / 2 - pad;
        float v = 0.f;
        if (j >= 0 && j < length) v = win[m] * (g[row * ld_g + j] / env[n % hop]);
        sm[bitrev(m, logn)] = make_float2(v, 0.f);
    }
    fft_lds(sm, tw, N, logn, false);
    const float sc = 1.0f / sqrtf((float)N);
    float* zr = gz + ((row * 2 + 0) * T + t) * (int64_t)(N / 2);
    float* zi = gz + ((row * 2 + 1) * T + t) * (int64_t)(N / 2);
    for (int k = threadIdx.x; k < N / 2; k += blockDim.x) {
        const float c = k == 0 ? sc : 2.0f * sc;
        zr[k] = sm[k].x * c;
        zi[k] = k == 0 ? 0.f : sm[k].y * c;
    }
}

// y[b][c][r] = x[b][r][c]: batched transpose through 32 x 33 LDS tiles (both sides coalesced)
__global__ __launch_bounds__(256) void k_transpose2d(const float* __restrict__ x, float* __restrict__ y, int64_t R, int64_t C) {
    __shared__ float tile[32][33];
    const int64_t b = blockIdx.z;
    const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < R && c0 + tx < C) tile[i][tx] = x[(b * R + r0 + i) * C + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < C && r0 + tx < R) y[(b * C + c0 + i) * R + r0 + tx] = tile[tx][i];
}

static int check_fft(int N, int hop, int* logn) {
    int l = 0;
    while ((1 << l) < N) ++l;
    FQSS_REQUIRE(N >= 8 && N <= 4096 && (1 << l) == N, "n_fft must be a power of two in [8, 4096]");
    FQSS_REQUIRE(hop > 0 && N % hop == 0, "hop must divide n_fft");
    *logn = l;
    return FQSS_OK;
}

}  // namespace fqss

using namespace fqss;

// x [rows][L] (row stride ld_x) -> z [rows][2][T][N/2];  T frames, reflect padding `pad` on the left (must be < L, as must the right one)
extern "C" int fqss_stft(const float* x, float* z, const float* win, const float* tw, int64_t rows, int64_t L, int64_t ld_x, int N, int hop,
                         int T, int pad, fqss_stream_t stream) {
    FQSS_REQUIRE(x && z && win && tw, "null pointer");
    int logn;
    if (int rc = check_fft(N, hop, &logn)) return rc;
    FQSS_REQUIRE(rows > 0 && rows <= 65535 && T > 0 && L > 1 && ld_x >= L, "bad shape");
    const int64_t last = (int64_t)(T - 1) * hop + N - 1 - pad;       // right-most sample index touched
    FQSS_REQUIRE(pad >= 0 && pad < L && last - (L - 1) < L, "reflect padding longer than the signal");
    hipLaunchKernelGGL(k_stft, dim3((unsigned)T, (unsigned)rows), dim3(256), (size_t)N * sizeof(float2), (hipStream_t)stream, x, z, win,
                       reinterpret_cast<const float2*>(tw), N, logn, hop, T, pad, L, ld_x);
    return launch_status("fqss_stft");
}

// z [rows][2][T][N/2] -> y [rows][length] (row stride ld_y); frames: workspace of rows*T*N floats; env [hop] = window envelope
extern "C" int fqss_istft(const float* z, float* frames, float* y, const float* win, const float* env, const float* tw, int64_t rows,
                          int64_t length, int64_t ld_y, int N, int hop, int T, int pad, fqss_stream_t stream) {
    FQSS_REQUIRE(z && frames && y && win && env && tw, "null pointer");
    int logn;
    if (int rc = check_fft(N, hop, &logn)) return rc;
    FQSS_REQUIRE(rows > 0 && rows <= 65535 && T > 0 && length > 0 && ld_y >= length && pad >= 0, "bad shape");
    hipLaunchKernelGGL(k_istft_frames, dim3((unsigned)T, (unsigned)rows), dim3(256), (size_t)N * sizeof(float2), (hipStream_t)stream, z, frames, win,
                       reinterpret_cast<const float2*>(tw), N, logn, T);
    int64_t gx = cdiv(length, 1024);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(k_istft_ola, dim3((unsigned)gx, (unsigned)rows), dim3(256), 0, (hipStream_t)stream, frames, y, env, N, hop, T, pad, length, ld_y);
    return launch_status("fqss_istft");
}

// g [rows][length] -> gz [rows][2][T][N/2]
extern "C" int fqss_istft_bwd(const float* g, float* gz, const float* win, const float* env, const float* tw, int64_t rows, int64_t length,
                              int64_t ld_g, int N, int hop, int T, int pad, fqss_stream_t stream) {
    FQSS_REQUIRE(g && gz && win && env && tw, "null pointer");
    int logn;
    if (int rc = check_fft(N, hop, &logn)) return rc;
    FQSS_REQUIRE(rows > 0 && rows <= 65535 && T > 0 && length > 0 && ld_g >= length && pad >= 0, "bad shape");
    hipLaunchKernelGGL(k_istft_bwd, dim3((unsigned)T, (unsigned)rows), dim3(256), (size_t)N * sizeof(float2), (hipStream_t)stream, g, gz, win, env,
                       reinterpret_cast<const float2*>(tw), N, logn, hop, T, pad, length, ld_g);
    return launch_status("fqss_istft_bwd");
}

// y [batch][C][R] = x [batch][R][C]  (dense)
extern "C" int fqss_transpose2d(const float* x, float* y, int64_t batch, int64_t R, int64_t C, fqss_stream_t stream) {
    FQSS_REQUIRE(x && y, "null pointer");
    FQSS_REQUIRE(batch > 0 && batch <= 65535 && R > 0 && C > 0 && cdiv(R, 32) <= 65535, "bad shape");
    hipLaunchKernelGGL(k_transpose2d, dim3((unsigned)cdiv(C, 32), (unsigned)cdiv(R, 32), (unsigned)batch), dim3(256), 0, (hipStream_t)stream, x, y, R, C);
    return launch_status("fqss_transpose2d");
}
