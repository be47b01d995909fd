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
#include <stdlib.h>

#include "fqss_dev.h"

namespace fqss {

// in-place DIT FFT of N = 2^logn points in LDS (radix-2 butterflies, fused three stages at a time); data must have been stored in
// bit-reversed order.
// tw[k] = exp(-2 pi i k / N), k < N/2; inverse: conjugated twiddles (no 1/N).
// tws (nullable): the twiddle table staged in LDS behind the data (N/2 entries) -- every butterfly then reads its twiddle from LDS
// instead of a (cached, but ~a microsecond per dependent round trip) global load
__device__ __forceinline__ void fft_lds(float2* s, const float2* __restrict__ tw, int N, int logn, bool inverse, float2* tws = nullptr) {
    if (tws != nullptr) {
        for (int k = threadIdx.x; k < N / 2; k += blockDim.x) tws[k] = tw[k];
        tw = tws;           // (generic address space: LDS reads from here on; the first stage's barrier orders the stores)
    }
    // one radix-2 butterfly, written once: the fused passes below apply it in the SAME order as the stage-by-stage form (the results
    // are bit-identical to it); w = tw[pos * tstep] of the butterfly's stage
    auto bfly = [&](float2& a, float2& c, float2 w) {
        if (inverse) w.y = -w.y;
        const float2 t = make_float2(c.x * w.x - c.y * w.y, c.x * w.y + c.y * w.x);
        const float2 a0 = a;
        a = make_float2(a0.x + t.x, a0.y + t.y);
        c = make_float2(a0.x - t.x, a0.y - t.y);
    };
    // Stages are fused in registers, three at a time (radix 8: a thread owns the eight points base + k h, h = 2^st, k < 8; stage st
    // pairs (0,1) (2,3) (4,5) (6,7) with one twiddle, stage st + 1 pairs (0,2) (1,3) (4,6) (5,7) with two, stage st + 2 pairs (0,4) ..
    // (3,7) with four), then two (radix 4) or one if that is what remains.  A third of the LDS traffic and of the barriers of the
    // stage-by-stage form, which was bound by exactly those (4096 points: 12 passes x 64 KB through LDS per frame: 674 -> 4xx us for
    // the 13.8 k inverse transforms of a cfg 5 step).
    int st = 0;
    for (; logn - st >= 3; st += 3) {
        const int h = 1 << st;
        const int ts0 = N >> (st + 1), ts1 = N >> (st + 2), ts2 = N >> (st + 3);
        __syncthreads();
        for (int gq = threadIdx.x; gq < N / 8; gq += blockDim.x) {
            const int pos = gq & (h - 1);
            const int base = ((gq >> st) << (st + 3)) + pos;
            float2 x[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = s[base + k * h];
            const float2 wa = tw[pos * ts0];
            bfly(x[0], x[1], wa);
            bfly(x[2], x[3], wa);
            bfly(x[4], x[5], wa);
            bfly(x[6], x[7], wa);
            const float2 wb0 = tw[pos * ts1], wb1 = tw[(pos + h) * ts1];
            bfly(x[0], x[2], wb0);
            bfly(x[1], x[3], wb1);
            bfly(x[4], x[6], wb0);
            bfly(x[5], x[7], wb1);
#pragma unroll
            for (int k = 0; k < 4; ++k) bfly(x[k], x[k + 4], tw[(pos + k * h) * ts2]);
#pragma unroll
            for (int k = 0; k < 8; ++k) s[base + k * h] = x[k];
        }
    }
    if (logn - st == 2) {
        const int h = 1 << st;
        const int ts0 = N >> (st + 1), ts1 = N >> (st + 2);
        __syncthreads();
        for (int gq = threadIdx.x; gq < N / 4; gq += blockDim.x) {
            const int pos = gq & (h - 1);
            const int base = ((gq >> st) << (st + 2)) + pos;
            float2 x0 = s[base], x1 = s[base + h], x2 = s[base + 2 * h], x3 = s[base + 3 * h];
            const float2 wa = tw[pos * ts0];
            bfly(x0, x1, wa);
            bfly(x2, x3, wa);
            bfly(x0, x2, tw[pos * ts1]);
            bfly(x1, x3, tw[(pos + h) * ts1]);
            s[base] = x0;
            s[base + h] = x1;
            s[base + 2 * h] = x2;
            s[base + 3 * h] = x3;
        }
    } else if (logn - st == 1) {
        const int h = 1 << st, ts0 = N >> (st + 1);
        __syncthreads();
        for (int b = threadIdx.x; b < N / 2; b += blockDim.x) {
            const int pos = b & (h - 1);
            const int i0 = ((b >> st) << (st + 1)) + pos;
            float2 a = s[i0], c = s[i0 + h];
            bfly(a, c, tw[pos * ts0]);
            s[i0] = a;
            s[i0 + h] = c;
        }
    }
    __syncthreads();
}
__device__ __forceinline__ int bitrev(int v, int logn) { return (int)(__brev((unsigned)v) >> (32 - logn)); }

// z[row][0/1][f][k] = (1/sqrt(N)) * FFT_k( w[n] * x2[f*hop + n] ),  k < N/2,  x2 = reflect-pad(x, pad) (`_spec`)
__global__ __launch_bounds__(1024) void k_stft(const float* __restrict__ x, float* __restrict__ z, const float* __restrict__ win,
                                               const float2* __restrict__ tw, int N, int logn, int hop, int T, int pad, int64_t L, int64_t ld_x, int twl) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    const int f = blockIdx.x;
    const int64_t row = blockIdx.y;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        int64_t i = (int64_t)f * hop + n - pad;
        if (i < 0) i = -i;
        if (i >= L) i = 2 * (L - 1) - i;
        sm[bitrev(n, logn)] = make_float2(win[n] * x[row * ld_x + i], 0.f);
    }
    fft_lds(sm, tw, N, logn, false, twl ? sm + N : nullptr);
    const float sc = 1.0f / sqrtf((float)N);
    float* zr = z + ((row * 2 + 0) * T + f) * (int64_t)(N / 2);
    float* zi = z + ((row * 2 + 1) * T + f) * (int64_t)(N / 2);
    for (int k = threadIdx.x; k < N / 2; k += blockDim.x) {
        zr[k] = sm[k].x * sc;
        zi[k] = sm[k].y * sc;
    }
}

// fr[row][t][n] = w[n] * sqrt(N) * irfft(Z_t)[n],  Z_t = bins 0 .. N/2-1 of z[row][0/1][t][:], Nyquist bin = 0
__global__ __launch_bounds__(1024) void k_istft_frames(const float* __restrict__ z, float* __restrict__ fr, const float* __restrict__ win,
                                                       const float2* __restrict__ tw, int N, int logn, int T, int twl) {
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
    fft_lds(sm, tw, N, logn, true, twl ? sm + N : nullptr);
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
__global__ __launch_bounds__(1024) void k_istft_bwd(const float* __restrict__ g, float* __restrict__ gz, const float* __restrict__ win,
                                                    const float* __restrict__ env, const float2* __restrict__ tw, int N, int logn, int hop, int T,
                                                    int pad, int64_t length, int64_t ld_g, int twl) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    const int t = blockIdx.x;
    const int64_t row = blockIdx.y;
    for (int m = threadIdx.x; m < N; m += blockDim.x) {
        const int64_t n = (int64_t)(t + 2) * hop + m, j = n - N / 2 - pad;
        float v = 0.f;
        if (j >= 0 && j < length) v = win[m] * (g[row * ld_g + j] / env[n % hop]);
        sm[bitrev(m, logn)] = make_float2(v, 0.f);
    }
    fft_lds(sm, tw, N, logn, false, twl ? sm + N : nullptr);
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

// threads per frame and whether the twiddles are staged in LDS (A/B knobs FQSS_FFT_THREADS / FQSS_FFT_TW_LDS; see k-comments)
static int fft_threads() {
    static const int v = [] { const char* e = getenv("FQSS_FFT_THREADS"); const int t = e ? atoi(e) : 512; return (t == 256 || t == 1024) ? t : 512; }();
    return v;
}
static int fft_tw_lds() {
    static const int v = [] { const char* e = getenv("FQSS_FFT_TW_LDS"); return e ? atoi(e) : 0; }();
    return v;
}
static size_t fft_lds_bytes(int N) { return (size_t)N * sizeof(float2) + (fft_tw_lds() ? (size_t)(N / 2) * sizeof(float2) : 0); }

// x [rows][L] (row stride ld_x) -> z [rows][2][T][N/2];  T frames, reflect padding `pad` on the left (must be < L, as must the right one)
extern "C" int fqss_stft(const float* x, float* z, const float* win, const float* tw, int64_t rows, int64_t L, int64_t ld_x, int N, int hop,
                         int T, int pad, fqss_stream_t stream) {
    FQSS_REQUIRE(x && z && win && tw, "null pointer");
    int logn;
    if (int rc = check_fft(N, hop, &logn)) return rc;
    FQSS_REQUIRE(rows > 0 && rows <= 65535 && T > 0 && L > 1 && ld_x >= L, "bad shape");
    const int64_t last = (int64_t)(T - 1) * hop + N - 1 - pad;       // right-most sample index touched
    FQSS_REQUIRE(pad >= 0 && pad < L && last - (L - 1) < L, "reflect padding longer than the signal");
    hipLaunchKernelGGL(k_stft, dim3((unsigned)T, (unsigned)rows), dim3(fft_threads()), fft_lds_bytes(N), (hipStream_t)stream, x, z, win,
                       reinterpret_cast<const float2*>(tw), N, logn, hop, T, pad, L, ld_x, fft_tw_lds());
    return launch_status("fqss_stft");
}

// z [rows][2][T][N/2] -> y [rows][length] (row stride ld_y); frames: workspace of rows*T*N floats; env [hop] = window envelope
extern "C" int fqss_istft(const float* z, float* frames, float* y, const float* win, const float* env, const float* tw, int64_t rows,
                          int64_t length, int64_t ld_y, int N, int hop, int T, int pad, fqss_stream_t stream) {
    FQSS_REQUIRE(z && frames && y && win && env && tw, "null pointer");
    int logn;
    if (int rc = check_fft(N, hop, &logn)) return rc;
    FQSS_REQUIRE(rows > 0 && rows <= 65535 && T > 0 && length > 0 && ld_y >= length && pad >= 0, "bad shape");
    hipLaunchKernelGGL(k_istft_frames, dim3((unsigned)T, (unsigned)rows), dim3(fft_threads()), fft_lds_bytes(N), (hipStream_t)stream, z, frames, win,
                       reinterpret_cast<const float2*>(tw), N, logn, T, fft_tw_lds());
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
    hipLaunchKernelGGL(k_istft_bwd, dim3((unsigned)T, (unsigned)rows), dim3(fft_threads()), fft_lds_bytes(N), (hipStream_t)stream, g, gz, win, env,
                       reinterpret_cast<const float2*>(tw), N, logn, hop, T, pad, length, ld_g, fft_tw_lds());
    return launch_status("fqss_istft_bwd");
}

// y [batch][C][R] = x [batch][R][C]  (dense)
extern "C" int fqss_transpose2d(const float* x, float* y, int64_t batch, int64_t R, int64_t C, fqss_stream_t stream) {
    FQSS_REQUIRE(x && y, "null pointer");
    FQSS_REQUIRE(batch > 0 && batch <= 65535 && R > 0 && C > 0 && cdiv(R, 32) <= 65535, "bad shape");
    hipLaunchKernelGGL(k_transpose2d, dim3((unsigned)cdiv(C, 32), (unsigned)cdiv(R, 32), (unsigned)batch), dim3(256), 0, (hipStream_t)stream, x, y, R, C);
    return launch_status("fqss_transpose2d");
}
