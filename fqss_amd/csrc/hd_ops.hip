// hd_ops.hip -- small streaming kernels of the HTDemucs layers:
//   * LayerScale (demucsq.py:19-39): x * scale[c] on channel-first [B][C][M] tensors (DConv) and on channel-last rows [R][C]
//     (transformer gamma_1 / gamma_2), with the scale gradient (sum of g * x per channel);
//   * the frequency-embedding add (htdemucsq.py:1063-1068): x[b][c][m] + e[c]  (c runs over channel x frequency);
//   * the mean / std normalisation of HTDemucsQ.pre_process / post_process (htdemucsq.py:1003-1014, 1034-1035): per-sample
//     mean and unbiased std over all other dims, (x - mean) / (1e-5 + std) and its inverse x * std + mean.
// All HBM streams: one pass, 16-B accesses where rows are aligned, wave shuffles + one atomic per workgroup for the sums.
#define FQSS_USES_GRAD_ADD   // the fp32 gradient atomics of this file go through grad_add (fqss_dev.h: FQSS_DETERMINISTIC=1)
#include <cstdlib>
#include "fqss_dev.h"

namespace fqss {

// y[r][m] = x[r][m] * s[r % C]   (mode 0)   |   x[r][m] + s[r % C]   (mode 1);  rows r = b*C + c
__global__ __launch_bounds__(256) void k_chan_op(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ y, int64_t R,
                                                  int64_t C, int64_t M, int64_t ld_x, int64_t ld_y, int mode) {
    for (int64_t r = blockIdx.y; r < R; r += gridDim.y) {
        const float sv = s[r % C];
        for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < M; m += (int64_t)gridDim.x * 256) {
            const float xv = x[r * ld_x + m];
            y[r * ld_y + m] = mode == 0 ? xv * sv : xv + sv;
        }
    }
}
// gx[r][m] = g[r][m] * s[r % C];  gs[r % C] += sum_m g[r][m] * x[r][m]
// A workgroup owns ONE channel (blockIdx.y), a column range (x) and a slice of the channel's R / C rows (z): the sum stays in registers
// over all of them -- one reduction and one atomic per workgroup.  (The first form, one row per workgroup with a reduction and an atomic
// per row, ran LayerScale's backward on the [B * F][C][T] tensors of the spectrogram DConv at 0.85 TB/s: 98 k same-address atomics.)
__global__ __launch_bounds__(256) void k_chan_scale_bwd(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ s,
                                                         float* __restrict__ gx, float* __restrict__ gs, int64_t R, int64_t C, int64_t M,
                                                         int64_t ld_g, int64_t ld_x, int64_t ld_gx) {
    __shared__ float smem[4];
    const int64_t c = blockIdx.y, nb = R / C;
    const float sv = s[c];
    float acc[1] = {0.f};
    const int64_t mstep = (int64_t)gridDim.x * 1024;
    for (int64_t bi = blockIdx.z; bi < nb; bi += gridDim.z) {
        const int64_t r = bi * C + c;
        const float* gr = g + r * ld_g;
        const float* xr = x + r * ld_x;
        float* orow = gx + r * ld_gx;
        for (int64_t m0 = (int64_t)blockIdx.x * 1024 + threadIdx.x; m0 < M; m0 += mstep) {
            float gv[4], xv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {       // four independent (clamped) loads before the first use
                const int64_t m = min(m0 + 256 * u, M - 1);
                gv[u] = gr[m];
                xv[u] = xr[m];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t m = m0 + 256 * u;
                if (m < M) {
                    orow[m] = gv[u] * sv;
                    acc[0] += gv[u] * xv[u];
                }
            }
        }
    }
    block_sum<float, 1>(acc, smem);
    if (threadIdx.x == 0) grad_add(gs + c, acc[0]);
}
// channel-last rows: y[r][c] = x[r][c] * s[c]
__global__ __launch_bounds__(256) void k_col_scale_fwd(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ y, int64_t R,
                                                        int C, int64_t ld_x, int64_t ld_y) {
    const int64_t total = R * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / C;
        const int c = (int)(i % C);
        y[r * ld_y + c] = x[r * ld_x + c] * s[c];
    }
}
// gx[r][c] = g[r][c] * s[c];  gs[c] += sum_r g[r][c] * x[r][c]   (a workgroup owns a band of rows, a thread a column stripe)
__global__ __launch_bounds__(256) void k_col_scale_bwd(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ s,
                                                        float* __restrict__ gx, float* __restrict__ gs, int64_t R, int C, int64_t ld_g,
                                                        int64_t ld_x, int64_t ld_gx, int64_t rows_per_block) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(R, r0 + rows_per_block);
    for (int c = threadIdx.x; c < C; c += 256) {
        const float sv = s[c];
        float acc = 0.f;
        for (int64_t r = r0; r < r1; ++r) {
            const float gv = g[r * ld_g + c];
            gx[r * ld_gx + c] = gv * sv;
            acc += gv * x[r * ld_x + c];
        }
        grad_add(gs + c, acc);
    }
}

// per-sample sum and sum of squares in fp64 partials: ws[b][2] += (sum, sum^2) of x[b][:n]
__global__ __launch_bounds__(256) void k_sample_moments(const float* __restrict__ x, double* __restrict__ ws, int64_t n) {
    __shared__ double smem[8];
    const int64_t b = blockIdx.y;
    const float* xb = x + b * n;
    double v[2] = {0.0, 0.0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double t = xb[i];
        v[0] += t;
        v[1] += t * t;
    }
    block_sum<double, 2>(v, smem);
    if (threadIdx.x == 0) {
        atomicAdd(ws + 2 * b, v[0]);
        atomicAdd(ws + 2 * b + 1, v[1]);
    }
}
// ms[b] = (mean, unbiased std) from the fp64 moments (torch.std default: correction = 1)
__global__ void k_moments_finish(const double* __restrict__ ws, float* __restrict__ ms, int64_t B, int64_t n) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double mean = ws[2 * b] / (double)n;
    double var = (ws[2 * b + 1] - (double)n * mean * mean) / (double)(n - 1);
    if (var < 0.0) var = 0.0;
    ms[2 * b] = (float)mean;
    ms[2 * b + 1] = (float)sqrt(var);
}
// dir 0: y = (x - mean_b) / (1e-5 + std_b)      dir 1: y = x * std_b + mean_b        (x [B][n] dense)
__global__ __launch_bounds__(256) void k_sample_norm(const float* __restrict__ x, const float* __restrict__ ms, float* __restrict__ y, int64_t n,
                                                      int dir) {
    const int64_t b = blockIdx.y;
    const float mean = ms[2 * b], sd = ms[2 * b + 1];
    const float den = 1e-5f + sd;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float t = x[b * n + i];
        y[b * n + i] = dir == 0 ? (t - mean) / den : t * sd + mean;
    }
}

static inline unsigned blocks_for(int64_t n, int64_t cap = 4096) {
    int64_t b = cdiv(n, 1024);
    return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace fqss

using namespace fqss;

extern "C" int fqss_chan_op(const float* x, const float* s, float* y, int64_t B, int64_t C, int64_t M, int64_t ld_x, int64_t ld_y, int mode,
                            fqss_stream_t stream) {
    FQSS_REQUIRE(x && s && y, "null pointer");
    FQSS_REQUIRE(B > 0 && C > 0 && M > 0 && ld_x >= M && ld_y >= M && (mode == 0 || mode == 1), "bad shape");
    const int64_t R = B * C;
    dim3 grid(blocks_for(M, 64), (unsigned)(R > 16384 ? 16384 : R));
    hipLaunchKernelGGL(k_chan_op, grid, dim3(256), 0, (hipStream_t)stream, x, s, y, R, C, M, ld_x, ld_y, mode);
    return launch_status("fqss_chan_op");
}

extern "C" int fqss_chan_scale_bwd(const float* g, const float* x, const float* s, float* gx, float* gs, int64_t B, int64_t C, int64_t M,
                                   int64_t ld_g, int64_t ld_x, int64_t ld_gx, fqss_stream_t stream) {
    FQSS_REQUIRE(g && x && s && gx && gs, "null pointer");
    FQSS_REQUIRE(B > 0 && C > 0 && M > 0 && ld_g >= M && ld_x >= M && ld_gx >= M, "bad shape");
    const int64_t R = B * C;
    FQSS_REQUIRE(C <= 65535, "too many channels");
    int64_t gxb = cdiv(M, 1024);
    if (gxb > 256) gxb = 256;
    int64_t gzb = 2048 / (gxb * C);                 // ~2,048 workgroups over the chip
    if (gzb > B) gzb = B;
    if (gzb > 65535) gzb = 65535;
    if (gzb < 1) gzb = 1;
    hipLaunchKernelGGL(k_chan_scale_bwd, dim3((unsigned)gxb, (unsigned)C, (unsigned)gzb), dim3(256), 0, (hipStream_t)stream, g, x, s, gx, gs, R, C, M,
                       ld_g, ld_x, ld_gx);
    return launch_status("fqss_chan_scale_bwd");
}

extern "C" int fqss_col_scale_fwd(const float* x, const float* s, float* y, int64_t R, int C, int64_t ld_x, int64_t ld_y, fqss_stream_t stream) {
    FQSS_REQUIRE(x && s && y, "null pointer");
    FQSS_REQUIRE(R > 0 && C > 0 && ld_x >= C && ld_y >= C, "bad shape");
    hipLaunchKernelGGL(k_col_scale_fwd, dim3(blocks_for(R * C)), dim3(256), 0, (hipStream_t)stream, x, s, y, R, C, ld_x, ld_y);
    return launch_status("fqss_col_scale_fwd");
}

extern "C" int fqss_col_scale_bwd(const float* g, const float* x, const float* s, float* gx, float* gs, int64_t R, int C, int64_t ld_g,
                                  int64_t ld_x, int64_t ld_gx, fqss_stream_t stream) {
    FQSS_REQUIRE(g && x && s && gx && gs, "null pointer");
    FQSS_REQUIRE(R > 0 && C > 0 && ld_g >= C && ld_x >= C && ld_gx >= C, "bad shape");
    // rows per workgroup: every workgroup ends in C float adds on the same C addresses, so taller bands win until the serial walk of a
    // band shows (13792 x 512, 20 launches of a cfg-5 step: 14 rows 1.02 ms, 32 rows 0.87, 64 rows 1.24, 128 rows 2.20; FQSS_COLSCALE_RPB for A/B)
    static const int64_t rpb_env = [] { const char* e = getenv("FQSS_COLSCALE_RPB"); return e ? (int64_t)atoi(e) : (int64_t)0; }();
    const int64_t rpb = rpb_env > 0 ? rpb_env : (cdiv(R, 1024) < 32 ? (R >= 8192 ? 32 : 8) : cdiv(R, 1024));
    hipLaunchKernelGGL(k_col_scale_bwd, dim3((unsigned)cdiv(R, rpb)), dim3(256), 0, (hipStream_t)stream, g, x, s, gx, gs, R, C, ld_g, ld_x, ld_gx, rpb);
    return launch_status("fqss_col_scale_bwd");
}

// ms [B][2] = (mean, std) per sample of x [B][n]; ws: B*2 doubles, zeroed by the caller
extern "C" int fqss_sample_meanstd(const float* x, double* ws, float* ms, int64_t B, int64_t n, fqss_stream_t stream) {
    FQSS_REQUIRE(x && ws && ms, "null pointer");
    FQSS_REQUIRE(B > 0 && B <= 65535 && n > 1, "bad shape");
    hipLaunchKernelGGL(k_sample_moments, dim3(blocks_for(n, 256), (unsigned)B), dim3(256), 0, (hipStream_t)stream, x, ws, n);
    hipLaunchKernelGGL(k_moments_finish, dim3((unsigned)cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, ws, ms, B, n);
    return launch_status("fqss_sample_meanstd");
}

extern "C" int fqss_sample_norm(const float* x, const float* ms, float* y, int64_t B, int64_t n, int dir, fqss_stream_t stream) {
    FQSS_REQUIRE(x && ms && y, "null pointer");
    FQSS_REQUIRE(B > 0 && B <= 65535 && n > 0 && (dir == 0 || dir == 1), "bad shape");
    hipLaunchKernelGGL(k_sample_norm, dim3(blocks_for(n, 1024), (unsigned)B), dim3(256), 0, (hipStream_t)stream, x, ms, y, n, dir);
    return launch_status("fqss_sample_norm");
}
