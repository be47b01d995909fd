// hd_loss.hip -- the training loss of the htdemucs environment (train_env/htdemucs_musdbhq/solver.py:333-366):
//   task_s = mean_b mean_{c,t} |est - src| ;  kd_s = mean_b ( w_bs * mean_{c,t} |est - fest| ) ,
//   w_bs = exp((sdr(src, fest) - sdr(src, est)) / 10) with demucs' new_sdr (third party, demucs~=4.0.0:
//   10 log10((sum src^2 + 1e-7) / (sum (src - x)^2 + 1e-7)) over channels and time), detached;
//   loss = sum_s wt_s ((1 - lambda) task_s + lambda kd_s) / sum_s wt_s.
// Two streams over the three [B][S][N] tensors: five fp64 sums per (b, s), a one-wave finish, then the gradient
// d loss / d est = ct_s sign(est - src) + ck_bs sign(est - fest).  No host synchronisation.
#include "fqss_dev.h"

namespace fqss {

__global__ __launch_bounds__(256) void k_hd_loss_sums(const float* __restrict__ est, const float* __restrict__ fest, const float* __restrict__ src,
                                                       double* __restrict__ sums, int64_t N) {
    __shared__ double smem[5 * 4];
    const int64_t bs = blockIdx.y;
    const float *e = est + bs * N, *f = fest + bs * N, *s = src + bs * N;
    double v[5] = {0, 0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
        const float ev = e[i], fv = f[i], sv = s[i];
        v[0] += fabsf(ev - sv);
        v[1] += fabsf(ev - fv);
        v[2] += (double)sv * sv;
        const float d1 = sv - fv, d2 = sv - ev;
        v[3] += (double)d1 * d1;
        v[4] += (double)d2 * d2;
    }
    block_sum<double, 5>(v, smem);
    if (threadIdx.x == 0)
        for (int k = 0; k < 5; ++k) atomicAdd(sums + bs * 5 + k, v[k]);
}

// out: [0] loss, [1 .. 1+S) task_s, [1+S .. 1+2S) kd_s, then w [B][S];  coef [B][S][2] = (ct, ck)
__global__ void k_hd_loss_finish(const double* __restrict__ sums, const float* __restrict__ wt, float* __restrict__ out, float* __restrict__ coef,
                                 int B, int S, int64_t N, float lambda) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double wsum = 0.0, loss = 0.0;
    for (int s = 0; s < S; ++s) wsum += wt[s];
    for (int s = 0; s < S; ++s) {
        double task = 0.0, kd = 0.0;
        for (int b = 0; b < B; ++b) {
            const double* q = sums + ((int64_t)b * S + s) * 5;
            const float sdr_t = 10.0f * log10f((float)((q[2] + 1e-7) / (q[3] + 1e-7)));
            const float sdr_q = 10.0f * log10f((float)((q[2] + 1e-7) / (q[4] + 1e-7)));
            const float w = expf((sdr_t - sdr_q) / 10.0f);
            out[1 + 2 * S + b * S + s] = w;
            task += q[0] / (double)N;
            kd += (double)w * (q[1] / (double)N);
            coef[((int64_t)b * S + s) * 2 + 0] = (float)((double)wt[s] / wsum * (1.0 - (double)lambda) / ((double)B * (double)N));
            coef[((int64_t)b * S + s) * 2 + 1] = (float)((double)wt[s] / wsum * (double)lambda * (double)w / ((double)B * (double)N));
        }
        task /= B;
        kd /= B;
        out[1 + s] = (float)task;
        out[1 + S + s] = (float)kd;
        loss += (double)wt[s] * ((1.0 - (double)lambda) * task + (double)lambda * kd);
    }
    out[0] = (float)(loss / wsum);
}

__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

__global__ __launch_bounds__(256) void k_hd_loss_grad(const float* __restrict__ est, const float* __restrict__ fest, const float* __restrict__ src,
                                                       const float* __restrict__ coef, float* __restrict__ g, int64_t N) {
    const int64_t bs = blockIdx.y;
    const float ct = coef[bs * 2], ck = coef[bs * 2 + 1];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
        const float ev = est[bs * N + i];
        g[bs * N + i] = ct * sgn(ev - src[bs * N + i]) + ck * sgn(ev - fest[bs * N + i]);
    }
}

}  // namespace fqss

using namespace fqss;

// est / fest / src [B][S][N] dense; wt [S]; sums: B*S*5 doubles (zeroed by the caller); out: 1 + 2S + B*S floats; coef: B*S*2 floats;
// gest may be null (evaluation only)
extern "C" int fqss_hd_kd_loss(const float* est, const float* fest, const float* src, const float* wt, double* sums, float* out, float* coef,
                               float* gest, int B, int S, int64_t N, float kd_lambda, fqss_stream_t stream) {
    FQSS_REQUIRE(est && fest && src && wt && sums && out && coef, "null pointer");
    FQSS_REQUIRE(B > 0 && S > 0 && N > 0 && (int64_t)B * S <= 65535, "bad shape");
    int64_t gx = cdiv(N, 2048);
    if (gx > 256) gx = 256;
    dim3 grid((unsigned)gx, (unsigned)(B * S));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_hd_loss_sums, grid, dim3(256), 0, s, est, fest, src, sums, N);
    hipLaunchKernelGGL(k_hd_loss_finish, dim3(1), dim3(64), 0, s, sums, wt, out, coef, B, S, N, kd_lambda);
    if (gest != nullptr) hipLaunchKernelGGL(k_hd_loss_grad, grid, dim3(256), 0, s, est, fest, src, coef, gest, N);
    return launch_status("fqss_hd_kd_loss");
}
