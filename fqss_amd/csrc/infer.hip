// infer.hip -- the evaluation side of the path: `process.model_infer` chunked inference with triangular
// overlap-add and per-chunk source re-ordering (process.py:105-194) and the SI-SNR the re-ordering and `val.py` are built on
// (torchmetrics' ScaleInvariantSignalNoiseRatio, third party: restated from its published form, zero-mean SI-SDR with eps = 2^-23).
// All of it stays on the device: five fp64 moments per (estimate, target) pair, a one-wave finish that also takes the re-ordering
// decision of `swap_channel_order` (so no host round trip per chunk), and one stream for the weighted overlap-add.
#include "fqss_dev.h"

namespace fqss {

// mom[p][q][5] += (sum e, sum r, sum e r, sum e e, sum r r) of est[p][:], ref[q][:]
__global__ __launch_bounds__(256) void k_sisnr_moments(const float* __restrict__ est, const float* __restrict__ ref, double* __restrict__ mom,
                                                        int S, int64_t L, int64_t ld_e, int64_t ld_r) {
    __shared__ double smem[5 * 4];
    const int p = blockIdx.y / S, q = blockIdx.y % S;
    const float *e = est + p * ld_e, *r = ref + q * ld_r;
    double v[5] = {0, 0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < L; i += (int64_t)gridDim.x * 256) {
        const double a = e[i], b = r[i];
        v[0] += a; v[1] += b; v[2] += a * b; v[3] += a * a; v[4] += b * b;
    }
    block_sum<double, 5>(v, smem);
    if (threadIdx.x == 0)
        for (int k = 0; k < 5; ++k) atomicAdd(mom + ((int64_t)p * S + q) * 5 + k, v[k]);
}

// db[p][q] = SI-SNR(est_p, ref_q) in dB; map[d] = (source index, sign) of swap_channel_order (process.py:105-125):
// for every estimate p in order: d = argmax_q db[p][q] (first maximum), out[d] = est_p if p == d else -est_p; untouched d keep est_d
__global__ void k_sisnr_finish(const double* __restrict__ mom, float* __restrict__ db, int* __restrict__ map, int S, int64_t L, int do_map) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double eps = 1.1920928955078125e-07;      // torch.finfo(torch.float32).eps
    for (int p = 0; p < S; ++p)
        for (int q = 0; q < S; ++q) {
            const double* m = mom + ((int64_t)p * S + q) * 5;
            const double n = (double)L;
            const double spt = m[2] - m[0] * m[1] / n, spp = m[3] - m[0] * m[0] / n, stt = m[4] - m[1] * m[1] / n;
            const double alpha = (spt + eps) / (stt + eps);
            const double num = alpha * alpha * stt, den = alpha * alpha * stt - 2.0 * alpha * spt + spp;
            db[p * S + q] = (float)(10.0 * log10((num + eps) / ((den < 0.0 ? 0.0 : den) + eps)));
        }
    if (!do_map) return;
    for (int d = 0; d < S; ++d) { map[2 * d] = d; map[2 * d + 1] = 1; }
    for (int p = 0; p < S; ++p) {
        int best = 0;
        float bv = -INFINITY;
        for (int q = 0; q < S; ++q)
            if (db[p * S + q] > bv) { bv = db[p * S + q]; best = q; }
        map[2 * best] = p;
        map[2 * best + 1] = p == best ? 1 : -1;
    }
}

// triangular chunk weight of model_infer (process.py:166-168): 1..h, (seg-h)..1 over max, h = seg / 2
__device__ __forceinline__ float tri_weight(int64_t t, int64_t seg) {
    const int64_t h = seg / 2;
    const float mx = (float)(seg - h);
    return (t < h ? (float)(t + 1) : (float)(seg - t)) / mx;
}

// out[d][c][start + t] += w[t] * sign_d * chunk[src_d][c][t];  sum_weight[start + t] += w[t]   (t < n)
__global__ __launch_bounds__(256) void k_infer_ola(const float* __restrict__ chunk, const int* __restrict__ map, float* __restrict__ out,
                                                    float* __restrict__ sum_weight, int S, int C, int64_t n, int64_t seg, int64_t start,
                                                    int64_t ld_chunk, int64_t ld_out) {
    const int dc = blockIdx.y, d = dc / C, c = dc % C;
    const int src = map != nullptr ? map[2 * d] : d;
    const float sign = map != nullptr ? (float)map[2 * d + 1] : 1.0f;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
        const float w = tri_weight(t, seg);
        out[((int64_t)d * C + c) * ld_out + start + t] += w * (sign * chunk[((int64_t)src * C + c) * ld_chunk + t]);
        if (dc == 0) sum_weight[start + t] += w;
    }
}

__global__ __launch_bounds__(256) void k_infer_normalize(float* __restrict__ out, const float* __restrict__ sum_weight, int64_t rows, int64_t L,
                                                          int64_t ld) {
    for (int64_t r = blockIdx.y; r < rows; r += gridDim.y)
        for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < L; t += (int64_t)gridDim.x * 256) out[r * ld + t] /= sum_weight[t];
}

}  // namespace fqss

using namespace fqss;

// est / ref: S rows of L samples; mom: S*S*5 doubles zeroed by the caller; db [S][S]; map (optional): 2*S ints
extern "C" int fqss_sisnr_matrix(const float* est, const float* ref, double* mom, float* db, int* map, int S, int64_t L, int64_t ld_e,
                                 int64_t ld_r, fqss_stream_t stream) {
    FQSS_REQUIRE(est && ref && mom && db, "null pointer");
    FQSS_REQUIRE(S > 0 && S <= 16 && L > 0 && ld_e >= L && ld_r >= L, "bad shape (S <= 16)");
    int64_t gx = cdiv(L, 4096);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(k_sisnr_moments, dim3((unsigned)gx, (unsigned)(S * S)), dim3(256), 0, (hipStream_t)stream, est, ref, mom, S, L, ld_e, ld_r);
    hipLaunchKernelGGL(k_sisnr_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, mom, db, map, S, L, map != nullptr);
    return launch_status("fqss_sisnr_matrix");
}

extern "C" int fqss_infer_ola(const float* chunk, const int* map, float* out, float* sum_weight, int S, int C, int64_t n, int64_t seg,
                              int64_t start, int64_t ld_chunk, int64_t ld_out, fqss_stream_t stream) {
    FQSS_REQUIRE(chunk && out && sum_weight, "null pointer");
    FQSS_REQUIRE(S > 0 && C > 0 && S * C <= 65535 && n > 0 && n <= seg && start >= 0 && ld_chunk >= n && ld_out >= start + n, "bad shape");
    int64_t gx = cdiv(n, 1024);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(k_infer_ola, dim3((unsigned)gx, (unsigned)(S * C)), dim3(256), 0, (hipStream_t)stream, chunk, map, out, sum_weight, S, C, n,
                       seg, start, ld_chunk, ld_out);
    return launch_status("fqss_infer_ola");
}

extern "C" int fqss_infer_normalize(float* out, const float* sum_weight, int64_t rows, int64_t L, int64_t ld, fqss_stream_t stream) {
    FQSS_REQUIRE(out && sum_weight, "null pointer");
    FQSS_REQUIRE(rows > 0 && L > 0 && ld >= L, "bad shape");
    int64_t gx = cdiv(L, 1024);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(k_infer_normalize, dim3((unsigned)gx, (unsigned)(rows > 1024 ? 1024 : rows)), dim3(256), 0, (hipStream_t)stream, out,
                       sum_weight, rows, L, ld);
    return launch_status("fqss_infer_normalize");
}
