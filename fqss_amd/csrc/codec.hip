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
#include <cstdlib>

#define FQSS_USES_GRAD_ADD   // the fp32 gradient atomics of this file go through grad_add (fqss_dev.h: FQSS_DETERMINISTIC=1)
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
                                                          int64_t ld_z, const float* __restrict__ add, int64_t ld_add) {
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
        if (m < M) {
            // + the other gradient of a two-consumer tensor (the decoder input also feeds the residual block): no separate sum pass
            if (add != nullptr) acc += add[((int64_t)n * Co + co) * ld_add + m];
            z[((int64_t)n * Co + co) * ld_z + m] = acc;
        }
    }
}

// The same with FOUR frames per lane (a workgroup covers 256 frames x all Co): the output row -- and the optional addend row -- move
// 16 B per lane (1-KB runs per row instead of 256 B), the taps come from LDS (staged once per workgroup) instead of one scalar-memory
// round trip per output channel.  Same j-ordered fmaf chain per output: bit-identical to k_frames_conv_fwd.
template <int CI, int K, int S>
__global__ __launch_bounds__(256) void k_frames_conv4(const float* __restrict__ x, const float* __restrict__ w,
                                                       float* __restrict__ z, int Co, int64_t T, int M, int64_t ld_z,
                                                       const float* __restrict__ add, int64_t ld_add, int co_tile, int co_per) {
    // co_per: output channels per grid.z slice (256 frames x ALL 512 channels per workgroup were 128 workgroups for 8 x 3,999 frames)
    constexpr int CK = CI * K, SW = 3 * S + K;
    static_assert(SW % 4 == 0 && CK % 4 == 0, "whole float4s");
    extern __shared__ __attribute__((aligned(16))) float Wl[];     // [co_tile][CK]: the taps of co_tile output channels at a time (<= 48 KB)
    const int n = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m0 = (blockIdx.x * 64 + lane) * 4;
    const bool live = m0 < M;
    const int mld = live ? m0 : 0;
    float xr[CI][SW];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) {
        const float* xp = x + ((int64_t)n * CI + ci) * T;
#pragma unroll
        for (int j = 0; j < SW / 4; ++j) {
            // groups past the end of the signal belong to frames >= M, whose results are not stored: clamped, not branched
            const int64_t idx = min((int64_t)mld * S + 4 * j, T - 4);
            const float4 t = *reinterpret_cast<const float4*>(xp + idx);
            xr[ci][4 * j] = t.x; xr[ci][4 * j + 1] = t.y; xr[ci][4 * j + 2] = t.z; xr[ci][4 * j + 3] = t.w;
        }
    }
    constexpr int PF = 8;      // output channels per group: the addend rows of a group are requested before its FMAs start
    const int ct = co_tile < 0 ? Co : co_tile;
    const int co_lo = (int)blockIdx.z * co_per, co_hi = min(Co, co_lo + co_per);
    for (int cb = co_lo; cb < co_hi; cb += ct) {
        const int ce = min(co_hi, cb + ct);
        if (co_tile > 0) {
        if (cb > co_lo) __syncthreads();   // every wave is done with the previous tile's taps
        for (int i = threadIdx.x; i < (ce - cb) * CK; i += 256) Wl[i] = w[(int64_t)cb * CK + i];
        __syncthreads();
        }
        for (int cg = cb + wave; cg < ce; cg += 4 * PF) {
            float4 a4[PF];
            if (add != nullptr) {
#pragma unroll
                for (int i = 0; i < PF; ++i) {
                    const int64_t row = (int64_t)n * Co + min(cg + 4 * i, ce - 1);
                    a4[i] = *reinterpret_cast<const float4*>(add + row * ld_add + mld);
                }
            }
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const int co = cg + 4 * i;
                if (co >= ce) break;       // wave-uniform
                float wk[CK];
                if (co_tile < 0) {           // taps straight from memory (wave-uniform): the A/B form
                    const float* wr = w + (int64_t)co * CK;
#pragma unroll
                    for (int j = 0; j < CK; ++j) wk[j] = wr[j];
                } else {
#pragma unroll
                for (int j4 = 0; j4 < CK / 4; ++j4) {
                    const float4 t = *reinterpret_cast<const float4*>(&Wl[(co - cb) * CK + 4 * j4]);
                    wk[4 * j4] = t.x; wk[4 * j4 + 1] = t.y; wk[4 * j4 + 2] = t.z; wk[4 * j4 + 3] = t.w;
                }
                }
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ci = 0; ci < CI; ++ci)
#pragma unroll
                    for (int k = 0; k < K; ++k)
#pragma unroll
                        for (int f = 0; f < 4; ++f) acc[f] = fmaf(xr[ci][f * S + k], wk[ci * K + k], acc[f]);
                if (live) {
                    const int64_t row = (int64_t)n * Co + co;
                    if (add != nullptr) {
                        acc[0] += a4[i].x; acc[1] += a4[i].y; acc[2] += a4[i].z; acc[3] += a4[i].w;
                    }
                    *reinterpret_cast<float4*>(z + row * ld_z + m0) = make_float4(acc[0], acc[1], acc[2], acc[3]);   // row padding absorbs m0 + 3 >= M
                }
            }
        }
    }
}

// out[n][t] = sum_c sum_{m*S+k=t} x[n][c][m] * w[c][k] ; R = K/S frames overlap on every output slot.
// A block owns FB = 64-(R-1) output slots (S samples each); its 64 lanes hold the 64 frames that
// touch them; the 4 waves split the channel reduction; partial P[k][frame] go through LDS.
// MODE 0: fp32 x; 1: x as u8 codes of a per-tensor quantizer, de-quantised on load (delta * c + min in two roundings, like
// fqss_decode: the student's MulQ / residual outputs never exist in fp32); 2: fp32 mask[n][c][m] * feat[n / NS][c][m] formed on load
// (the float teacher's masking product).  Same sums as decode / product followed by the MODE 0 kernel, bit for bit.
template <int K, int S, int MODE>
__global__ __launch_bounds__(256) void k_ola_convtr_fwd(const void* __restrict__ x_, const float* __restrict__ feat,
                                                         const float* __restrict__ w, float* __restrict__ out, int C, int M,
                                                         int64_t ld_x, int64_t ld_f, int NS, int64_t T, const float* qmin,
                                                         const float* qmax) {
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
    const int mc = mv ? m : 0;           // a readable position of the row (its value is dropped)
    QRange rx{0.f, 1.f, 1.f};
    if (MODE == 1) rx = load_qrange(qmin, qmax);
    float acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0.0f;
    const float* xf = (const float*)x_ + (int64_t)n * C * ld_x + mc;
    const uint8_t* xc = (const uint8_t*)x_ + (int64_t)n * C * ld_x + mc;
    const float* ff = (MODE == 2) ? feat + (int64_t)(n / NS) * C * ld_f + mc : nullptr;
    for (int c = wave; c < C; c += 4) {
        float xv;
        if (MODE == 1) xv = rx.delta * (float)xc[(int64_t)c * ld_x] + rx.lo;
        else if (MODE == 2) xv = xf[(int64_t)c * ld_x] * ff[(int64_t)c * ld_f];
        else xv = xf[(int64_t)c * ld_x];
        xv = mv ? xv : 0.0f;
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


// the four codes of a word de-quantised like csrc/fused_q.hip's dec4 (delta * c + min in two roundings)
__device__ __forceinline__ void ola_dec4(unsigned int w, const QRange& r, float (&v)[4]) {
    v[0] = r.delta * (float)(w & 255u) + r.lo;
    v[1] = r.delta * (float)((w >> 8) & 255u) + r.lo;
    v[2] = r.delta * (float)((w >> 16) & 255u) + r.lo;
    v[3] = r.delta * (float)(w >> 24) + r.lo;
}

typedef float f32x4m __attribute__((ext_vector_type(4)));

// The (16, 8) decoder on the fp32 matrix cores: frames x taps = sum over channels of x[c][frame] * w[c][tap] as v_mfma_f32_16x16x4_f32
// steps of four channels (bit for bit a k-ordered fmaf chain).  A lane owns FOUR consecutive frames of one channel -- one float4, or four
// codes in one dword -- and they feed four accumulator tiles (tile j = frames mb + 4 i + j, i = 0 .. 15), so the operand row is read
// in 256-B runs and each element once; the taps of the step's four channels are one 4-B load per lane.  A workgroup covers 64 frames (60
// output slots); its four waves split the channels and meet in LDS for the overlap-add, as in k_ola_convtr_fwd.  Operand modes as there.
template <int MODE>
__global__ __launch_bounds__(256) void k_ola_mfma16(const void* __restrict__ x_, const float* __restrict__ feat,
                                                     const float* __restrict__ w, float* __restrict__ out, int C, int M,
                                                     int64_t ld_x, int64_t ld_f, int NS, int64_t T, const float* qmin,
                                                     const float* qmax) {
    constexpr int K = 16, S = 8, R = 2, FB = 60, PL = 65, U = 8;
    __shared__ float P[4][K][PL];
    const int n = blockIdx.y;
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q0 = blockIdx.x * FB, mb = q0 - 4;      // first output slot / first frame of this workgroup (a multiple of 4)
    const int m0 = mb + 4 * li;                       // this lane's frames m0 .. m0 + 3
    const bool in_row = m0 >= 0 && m0 < M;            // rows are padded to 4 elements: the group is readable or wholly outside
    const int mld = in_row ? m0 : 0;
    bool fv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) fv[j] = in_row && m0 + j < M;
    QRange rx{0.f, 1.f, 1.f};
    if (MODE == 1) rx = load_qrange(qmin, qmax);
    const int cw = ((C + 15) / 16) * 4;               // channels per wave, a multiple of the MFMA's 4
    const int c_beg = wave * cw, c_end = min(C, c_beg + cw);
    f32x4m acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4m{0.f, 0.f, 0.f, 0.f};
    const int64_t xb = (int64_t)n * C * ld_x + mld, fb = (MODE == 2) ? (int64_t)(n / NS) * C * ld_f + mld : 0;
    for (int c0 = c_beg; c0 < c_end; c0 += 4 * U) {
        float4 xa[U], xf2[U];
        unsigned int xc[U];
        float wv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {                 // the loads of U steps are issued before the first is used
            const int c = min(c0 + 4 * u + lk, C - 1);
            if (MODE == 1) xc[u] = *reinterpret_cast<const unsigned int*>((const uint8_t*)x_ + xb + (int64_t)c * ld_x);
            else xa[u] = *reinterpret_cast<const float4*>((const float*)x_ + xb + (int64_t)c * ld_x);
            if (MODE == 2) xf2[u] = *reinterpret_cast<const float4*>(feat + fb + (int64_t)c * ld_f);
            wv[u] = w[(int64_t)c * K + li];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool cok = c0 + 4 * u + lk < c_end;  // channels past this wave's range contribute zeros
            float v[4];
            if (MODE == 1) {
                ola_dec4(xc[u], rx, v);
            } else {
                v[0] = xa[u].x; v[1] = xa[u].y; v[2] = xa[u].z; v[3] = xa[u].w;
                if (MODE == 2) {
                    v[0] *= xf2[u].x; v[1] *= xf2[u].y; v[2] *= xf2[u].z; v[3] *= xf2[u].w;
                }
            }
            const float wk = cok ? wv[u] : 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32((fv[j] && cok) ? v[j] : 0.0f, wk, acc[j], 0, 0, 0);
        }
    }
    // register r of lane (li, lk) in tile j is (frame mb + 4 (4 lk + r) + j, tap li)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) P[wave][li][4 * (4 * lk + r) + j] = acc[j][r];
    __syncthreads();
    const int nslots = M + R - 1;
    for (int e = threadIdx.x; e < FB * S; e += 256) {
        const int ql = e / S, jj = e - ql * S;
        const int q = q0 + ql;
        if (q < nslots) {
            float v = 0.0f;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int fl = ql + 4 - r;       // local index of frame q - r
#pragma unroll
                for (int wv2 = 0; wv2 < 4; ++wv2) v += P[wv2][r * S + jj][fl];
            }
            const int64_t t = (int64_t)q * S + jj;
            if (t < T) out[(int64_t)n * T + t] = v;
        }
    }
}

// Weight gradient of the single-channel framing conv / the mono decoder: gw[c][k] += sum_{n,m} a[n][c][m] * sig[n][m*S + k]
// (qat_layers.py:1028-1039, 1330-1341 of the reference through autograd) = a [C x M] . frames(sig) [M x K] product per signal, on the
// fp32 matrix cores (v_mfma_f32_16x16x4_f32: bit for bit a k-ordered fmaf chain, so the contract of the VALU form holds).  A wave owns
// 64 channels (4 tiles of 16) of one signal; the operand row is read VL values per lane ALONG m -- 2 x float4, or 16 codes as one
// dwordx4 (CODED: the student decoder's input) -- and, the sum over m being order-free, element j of lane group g is the k-slice
// m = m0 + VL g + j of MFMA step j; the matching frame taps sig[(m0 + VL g + j) S + tap] are one 4-B load per step (64 B contiguous per
// lane group).  The four waves of a workgroup interleave their steps (adjacent 128-B lines of every row) and add their tiles in
// LDS before the atomics.  The generic fp32 GEMM spent 67-78 us on this shape (16 columns, strided 4-B signal loads), a VALU form
// with four frames per lane 70-100 (its per-lane signal windows are 64 cache lines per load instruction).
template <int K, int S, bool CODED>
__global__ __launch_bounds__(256) void k_frames_wgrad_mfma(const void* __restrict__ a_, const float* __restrict__ sig,
                                                            float* __restrict__ gw, int C, int M, int64_t ld_a, int64_t T, int mchunk,
                                                            const float* qmin, const float* qmax, int64_t sig_ns, int64_t ld_gw) {
    constexpr int TC = 4, NT = K / 16;
    constexpr int VL = CODED ? 16 : 8;       // operand values per lane and iteration
    constexpr int FI = 4 * VL;               // frames per iteration
    static_assert(K % 16 == 0, "whole 16-tap tiles");
    __shared__ float red[4][TC * NT * 4][64];
    const int n = blockIdx.y;
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c0 = blockIdx.x * 16 * TC;
    const int mbeg = blockIdx.z * mchunk, mend = min(M, mbeg + mchunk);
    QRange ra{0.f, 1.f, 1.f};
    if (CODED) ra = load_qrange(qmin, qmax);
    f32x4m acc[TC][NT];
#pragma unroll
    for (int t = 0; t < TC; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[t][u] = f32x4m{0.f, 0.f, 0.f, 0.f};
    const float* sn = sig + (int64_t)n * sig_ns;     // (one input channel of a [N][Ci][T] signal: sig_ns = Ci T, gw rows Ci K apart)
    int64_t arow[TC];
#pragma unroll
    for (int t = 0; t < TC; ++t) arow[t] = ((int64_t)n * C + min(c0 + 16 * t + li, C - 1)) * ld_a;
    for (int mb = mbeg + FI * wave; mb < mend; mb += 4 * FI) {
        const int m0 = mb + VL * lk;                       // this lane's VL frames m0 .. m0 + VL - 1
        float av[TC][VL];
#pragma unroll
        for (int t = 0; t < TC; ++t) {
            if (CODED) {
                const int cl = (int)min((int64_t)m0, ld_a - 16);      // rows are padded to 16 codes: a group is inside or wholly past M
                const uint4 cw = *reinterpret_cast<const uint4*>((const uint8_t*)a_ + arow[t] + cl);
                const unsigned int wq[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float d[4];
                    ola_dec4(wq[q], ra, d);
#pragma unroll
                    for (int e = 0; e < 4; ++e) av[t][4 * q + e] = d[e];
                }
            } else {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int cl = (int)min((int64_t)m0 + 4 * q, ld_a - 4);
                    const float4 v = *reinterpret_cast<const float4*>((const float*)a_ + arow[t] + cl);
                    av[t][4 * q] = v.x; av[t][4 * q + 1] = v.y; av[t][4 * q + 2] = v.z; av[t][4 * q + 3] = v.w;
                }
            }
        }
        float bv[VL][NT];
#pragma unroll
        for (int j = 0; j < VL; ++j)
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                // frames past the end only ever meet a zeroed operand: clamped, not branched
                const int64_t idx = min((int64_t)(m0 + j) * S + 16 * u + li, T - 1);
                bv[j][u] = sn[idx];
            }
#pragma unroll
        for (int j = 0; j < VL; ++j) {
            const bool ok = m0 + j < mend;
#pragma unroll
            for (int t = 0; t < TC; ++t) {
                const float x = ok ? av[t][j] : 0.0f;
#pragma unroll
                for (int u = 0; u < NT; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, bv[j][u], acc[t][u], 0, 0, 0);
            }
        }
    }
    // the four waves' tiles meet in LDS; register r of lane (li, lk) is (channel 4 lk + r, tap li) of its 16 x 16 tile
#pragma unroll
    for (int t = 0; t < TC; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][(t * NT + u) * 4 + r][lane] = acc[t][u][r];
    __syncthreads();
    for (int e = threadIdx.x; e < TC * NT * 4 * 64; e += 256) {
        const int q = e >> 6, l = e & 63;
        const float v = (red[0][q][l] + red[1][q][l]) + (red[2][q][l] + red[3][q][l]);
        const int t = q / (NT * 4), u = (q / 4) % NT, r = q & 3;
        const int c = c0 + 16 * t + 4 * (l >> 4) + r;
        if (c < C) grad_add(&gw[(int64_t)c * ld_gw + 16 * u + (l & 15)], v);
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

static int frames_conv_impl(const char* who, const float* x, const float* w, float* z, int N, int Ci, int Co, int64_t T, int K,
                            int stride, int M, int64_t ld_z, const float* add, int64_t ld_add, fqss_stream_t stream) {
    FQSS_REQUIRE(x && w && z, "null tensor");
    FQSS_REQUIRE(N >= 0 && N <= 65535 && Ci > 0 && Co > 0 && K > 0 && stride > 0 && M >= 0 && ld_z >= M, "bad shape");
    FQSS_REQUIRE(M == 0 || (int64_t)(M - 1) * stride + K <= T, "frames exceed the signal");
    FQSS_REQUIRE(!add || ld_add >= M, "bad addend rows");
    if (N == 0 || M == 0) return FQSS_OK;
    hipStream_t s = (hipStream_t)stream;
    {   // four frames per lane: vector-aligned rows with readable / writable padding, dense 16-B aligned signals
        // A/B switch, bit (Ci - 1)
        static const int wide_ci = [] { const char* e = getenv("FQSS_FRAMES_WIDE"); return e ? atoi(e) : 3; }();
        const bool wide_on = Ci <= 2 && ((wide_ci >> (Ci - 1)) & 1);
        const int64_t m4 = (M + 3) & ~(int64_t)3;
        const bool k16 = K == 16 && stride == 8, k32 = K == 32 && stride == 16;
        int co_tile = (int)((48 * 1024) / ((size_t)Ci * K * sizeof(float))) & ~3;      // output channels whose taps fit 48 KB of LDS
        if (co_tile > Co) co_tile = (Co + 3) & ~3;
        size_t lds = (size_t)co_tile * Ci * K * sizeof(float);
        static const int taps_lds = [] { const char* e = getenv("FQSS_TAPS_LDS"); return e ? atoi(e) : 1; }();   // A/B switch
        if (!taps_lds) { co_tile = -1; lds = 0; }       // taps by wave-uniform loads from memory
        if (wide_on && (k16 || k32) && (Ci == 1 || Ci == 2) && aligned16(x) && T % 4 == 0 && T >= 4 && aligned16(z) && ld_z % 4 == 0 &&
            ld_z >= m4 && (!add || (aligned16(add) && ld_add % 4 == 0 && ld_add >= m4))) {
            // slices of the output channels in grid.z until ~512 workgroups are in flight (>= 32 channels per slice)
            int64_t zs = cdiv(512, cdiv(M, 256) * (int64_t)N);
            if (zs > Co / 32) zs = Co / 32;
            if (zs < 1) zs = 1;
            const int co_per = (int)(cdiv(cdiv(Co, zs), 4) * 4);
            zs = cdiv(Co, co_per);
            if (co_tile > co_per) { co_tile = co_per; lds = (size_t)co_tile * Ci * K * sizeof(float); }
            dim3 grid4((unsigned)cdiv(M, 256), (unsigned)N, (unsigned)zs);
#define FQSS_FC4(CI_, K_, S_) \
    hipLaunchKernelGGL((k_frames_conv4<CI_, K_, S_>), grid4, dim3(256), lds, s, x, w, z, Co, T, M, ld_z, add, ld_add, co_tile, co_per)
            if (k16) { if (Ci == 1) FQSS_FC4(1, 16, 8); else FQSS_FC4(2, 16, 8); }
            else     { if (Ci == 1) FQSS_FC4(1, 32, 16); else FQSS_FC4(2, 32, 16); }
#undef FQSS_FC4
            return launch_status(who);
        }
    }
    dim3 grid((unsigned)cdiv(M, 64), (unsigned)N), block(256);
#define FQSS_FC(CI_, K_) hipLaunchKernelGGL((k_frames_conv_fwd<CI_, K_>), grid, block, 0, s, x, w, z, Co, T, stride, M, ld_z, add, ld_add)
    if (Ci == 1 && K == 16) FQSS_FC(1, 16);
    else if (Ci == 2 && K == 16) FQSS_FC(2, 16);
    else if (Ci == 1 && K == 32) FQSS_FC(1, 32);
    else if (Ci == 2 && K == 32) FQSS_FC(2, 32);
    else if (Ci == 1 && K == 2) FQSS_FC(1, 2);      // DPTNet encoder: 2-sample window, hop 1 (dptnetq.py:116)
    else if (Ci == 2 && K == 2) FQSS_FC(2, 2);
    else {
        set_error("%s: unsupported (Ci=%d, K=%d); built for Ci in {1,2}, K in {2,16,32}", who, Ci, K);
        return FQSS_EINVAL;
    }
#undef FQSS_FC
    return launch_status(who);
}

extern "C" int fqss_frames_conv_fwd(const float* x, const float* w, float* z, int N, int Ci, int Co, int64_t T, int K,
                                    int stride, int M, int64_t ld_z, fqss_stream_t stream) {
    return frames_conv_impl("fqss_frames_conv_fwd", x, w, z, N, Ci, Co, T, K, stride, M, ld_z, nullptr, 0, stream);
}

extern "C" int fqss_frames_conv_add_fwd(const float* x, const float* w, const float* add, int64_t ld_add, float* z, int N, int Ci, int Co,
                                        int64_t T, int K, int stride, int M, int64_t ld_z, fqss_stream_t stream) {
    FQSS_REQUIRE(add, "null addend");
    return frames_conv_impl("fqss_frames_conv_add_fwd", x, w, z, N, Ci, Co, T, K, stride, M, ld_z, add, ld_add, stream);
}

// mode 0: fp32 x; 1: u8 codes x + (qmin, qmax); 2: fp32 x * feat[n / NS]
static int ola_convtr_impl(const char* who, int mode, const void* x, const float* feat, const float* w, float* out, int N, int C, int M,
                           int64_t ld_x, int64_t ld_f, int NS, int K, int stride, int64_t T, const float* qmin, const float* qmax,
                           fqss_stream_t stream) {
    FQSS_REQUIRE(x && w && out, "null tensor");
    FQSS_REQUIRE(N >= 0 && N <= 65535 && C > 0 && M >= 0 && ld_x >= M && K > 0 && stride > 0, "bad shape");
    FQSS_REQUIRE(M == 0 || T == (int64_t)(M - 1) * stride + K, "T must equal (M-1)*stride + K");
    FQSS_REQUIRE(mode != 1 || (qmin && qmax), "coded operand needs its ranges");
    FQSS_REQUIRE(mode != 2 || (feat && NS >= 1 && N % NS == 0 && ld_f >= M), "masking form: feat [N / NS][C][M]");
    if (N == 0 || M == 0) return FQSS_OK;
    hipStream_t s = (hipStream_t)stream;
    if (NS < 1) NS = 1;
    {   // (16, 8) window on vector-aligned rows: the matrix-core form
        static const bool mfma_on = [] { const char* e = getenv("FQSS_OLA_MFMA"); return !(e && e[0] == '0'); }();   // A/B switch
        const int64_t m4 = (M + 3) & ~(int64_t)3;
        bool ok = mfma_on && K == 16 && stride == 8 && ld_x >= m4 && aligned16(x) && (mode == 1 ? ld_x % 16 == 0 : ld_x % 4 == 0);
        if (mode == 2) ok = ok && aligned16(feat) && ld_f % 4 == 0 && ld_f >= m4;
        if (ok) {
            dim3 gridm((unsigned)cdiv((int64_t)M + 1, 60), (unsigned)N);
            if (mode == 0) hipLaunchKernelGGL(k_ola_mfma16<0>, gridm, dim3(256), 0, s, x, feat, w, out, C, M, ld_x, ld_f, NS, T, qmin, qmax);
            else if (mode == 1) hipLaunchKernelGGL(k_ola_mfma16<1>, gridm, dim3(256), 0, s, x, feat, w, out, C, M, ld_x, ld_f, NS, T, qmin, qmax);
            else hipLaunchKernelGGL(k_ola_mfma16<2>, gridm, dim3(256), 0, s, x, feat, w, out, C, M, ld_x, ld_f, NS, T, qmin, qmax);
            return launch_status(who);
        }
    }
    dim3 grid((unsigned)cdiv(M + 1, 63), (unsigned)N);
#define FQSS_OLA(K_, S_)                                                                                                                  \
    do {                                                                                                                                  \
        if (mode == 0) hipLaunchKernelGGL((k_ola_convtr_fwd<K_, S_, 0>), grid, dim3(256), 0, s, x, feat, w, out, C, M, ld_x, ld_f, NS, T, qmin, qmax); \
        else if (mode == 1) hipLaunchKernelGGL((k_ola_convtr_fwd<K_, S_, 1>), grid, dim3(256), 0, s, x, feat, w, out, C, M, ld_x, ld_f, NS, T, qmin, qmax); \
        else hipLaunchKernelGGL((k_ola_convtr_fwd<K_, S_, 2>), grid, dim3(256), 0, s, x, feat, w, out, C, M, ld_x, ld_f, NS, T, qmin, qmax); \
    } while (0)
    if (K == 16 && stride == 8) FQSS_OLA(16, 8);
    else if (K == 32 && stride == 16) FQSS_OLA(32, 16);
    else if (K == 2 && stride == 1) FQSS_OLA(2, 1);   // DPTNet's 2-sample window, hop 1 (input gradient of its encoder)
    else {
        set_error("%s: unsupported (K=%d, stride=%d); built for (2,1), (16,8) and (32,16)", who, K, stride);
        return FQSS_EINVAL;
    }
#undef FQSS_OLA
    return launch_status(who);
}

extern "C" int fqss_ola_convtr_fwd(const float* x, const float* w, float* out, int N, int C, int M, int64_t ld_x, int K,
                                   int stride, int64_t T, fqss_stream_t stream) {
    return ola_convtr_impl("fqss_ola_convtr_fwd", 0, x, nullptr, w, out, N, C, M, ld_x, 0, 1, K, stride, T, nullptr, nullptr, stream);
}

extern "C" int fqss_ola_convtr_fwd_q(const uint8_t* xc, const float* qmin, const float* qmax, const float* w, float* out, int N, int C,
                                     int M, int64_t ld_x, int K, int stride, int64_t T, fqss_stream_t stream) {
    return ola_convtr_impl("fqss_ola_convtr_fwd_q", 1, xc, nullptr, w, out, N, C, M, ld_x, 0, 1, K, stride, T, qmin, qmax, stream);
}

extern "C" int fqss_ola_convtr_mul_fwd(const float* mask, const float* feat, const float* w, float* out, int N, int NS, int C, int M,
                                       int64_t ld_m, int64_t ld_f, int K, int stride, int64_t T, fqss_stream_t stream) {
    return ola_convtr_impl("fqss_ola_convtr_mul_fwd", 2, mask, feat, w, out, N, C, M, ld_m, ld_f, NS, K, stride, T, nullptr, nullptr, stream);
}

// Ci = 1 forms of fqss_frames_wgrad: a fp32 (coded = 0) or u8 codes; returns false when the shape is not served
static bool frames_wgrad1_launch(int coded, const void* a, const float* sig, float* gw, int N, int C, int M, int64_t ld_a, int64_t T, int K,
                                 int stride, const float* qmin, const float* qmax, hipStream_t s, int64_t sig_ns = 0, int64_t ld_gw = 0) {
    if (sig_ns <= 0) sig_ns = T;
    if (ld_gw <= 0) ld_gw = K;
    const bool k16 = K == 16 && stride == 8, k32 = K == 32 && stride == 16;
    if (!(k16 || k32) || T < 1 || N > 65535) return false;
    if (coded ? !(aligned16(a) && ld_a % 16 == 0 && ld_a >= 16) : !(aligned16(a) && ld_a % 4 == 0 && ld_a >= 4)) return false;
    const int fi4 = 4 * 4 * (coded ? 16 : 8);      // frames per workgroup step
    const int64_t gx = cdiv(C, 64);
    // >= 512 workgroups (two waves per SIMD): split the frame range when N x channel groups is small; the partial sums meet in the atomics
    int msplit = (int)cdiv(512, (int64_t)N * gx);
    if (msplit < 1) msplit = 1;
    if (msplit > 16) msplit = 16;
    int mchunk = (int)(cdiv(cdiv(M, msplit), fi4) * fi4);
    msplit = (int)cdiv(M, mchunk);
    dim3 grid((unsigned)gx, (unsigned)N, (unsigned)msplit);
#define FQSS_FW1(K_, S_, CD_) \
    hipLaunchKernelGGL((k_frames_wgrad_mfma<K_, S_, CD_>), grid, dim3(256), 0, s, a, sig, gw, C, M, ld_a, T, mchunk, qmin, qmax, sig_ns, ld_gw)
    if (k16) { if (coded) FQSS_FW1(16, 8, true); else FQSS_FW1(16, 8, false); }
    else     { if (coded) FQSS_FW1(32, 16, true); else FQSS_FW1(32, 16, false); }
#undef FQSS_FW1
    return true;
}

extern "C" int fqss_frames_wgrad1(const float* a, const float* sig, float* gw, int N, int C, int M, int64_t ld_a, int64_t T, int K,
                                  int stride, fqss_stream_t stream) {
    FQSS_REQUIRE(a && sig && gw, "null tensor");
    FQSS_REQUIRE(N >= 0 && C > 0 && M >= 0 && ld_a >= M && K > 0 && stride > 0, "bad shape");
    FQSS_REQUIRE(M == 0 || (int64_t)(M - 1) * stride + K <= T, "frames exceed the signal");
    if (M == 0 || N == 0) return FQSS_OK;
    if (!frames_wgrad1_launch(0, a, sig, gw, N, C, M, ld_a, T, K, stride, nullptr, nullptr, (hipStream_t)stream)) {
        set_error("fqss_frames_wgrad1: needs (K, stride) in {(16, 8), (32, 16)} and 16-B aligned rows");
        return FQSS_EINVAL;
    }
    return launch_status("fqss_frames_wgrad1");
}

/* one input channel ci of a multi-channel framing conv: sig = x + ci T with sig_ns = Ci T between signals, gw = gw0 + ci K with rows
 * ld_gw = Ci K apart (the student encoder's two splitter channels: Conv1dEncoderQ, qat_layers.py:993-1046) */
extern "C" int fqss_frames_wgrad1s(const float* a, const float* sig, int64_t sig_ns, float* gw, int64_t ld_gw, int N, int C, int M,
                                   int64_t ld_a, int64_t T, int K, int stride, fqss_stream_t stream) {
    FQSS_REQUIRE(a && sig && gw, "null tensor");
    FQSS_REQUIRE(N >= 0 && C > 0 && M >= 0 && ld_a >= M && K > 0 && stride > 0 && sig_ns >= T && ld_gw >= K, "bad shape");
    FQSS_REQUIRE(M == 0 || (int64_t)(M - 1) * stride + K <= T, "frames exceed the signal");
    if (M == 0 || N == 0) return FQSS_OK;
    if (!frames_wgrad1_launch(0, a, sig, gw, N, C, M, ld_a, T, K, stride, nullptr, nullptr, (hipStream_t)stream, sig_ns, ld_gw)) {
        set_error("fqss_frames_wgrad1s: needs (K, stride) in {(16, 8), (32, 16)} and 16-B aligned rows");
        return FQSS_EINVAL;
    }
    return launch_status("fqss_frames_wgrad1s");
}

extern "C" int fqss_frames_wgrad1_q(const uint8_t* ac, const float* qmin, const float* qmax, const float* sig, float* gw, int N, int C,
                                    int M, int64_t ld_a, int64_t T, int K, int stride, fqss_stream_t stream) {
    FQSS_REQUIRE(ac && qmin && qmax && sig && gw, "null tensor");
    FQSS_REQUIRE(N >= 0 && C > 0 && M >= 0 && ld_a >= M && K > 0 && stride > 0, "bad shape");
    FQSS_REQUIRE(M == 0 || (int64_t)(M - 1) * stride + K <= T, "frames exceed the signal");
    if (M == 0 || N == 0) return FQSS_OK;
    if (!frames_wgrad1_launch(1, ac, sig, gw, N, C, M, ld_a, T, K, stride, qmin, qmax, (hipStream_t)stream)) {
        set_error("fqss_frames_wgrad1_q: needs (K, stride) in {(16, 8), (32, 16)} and 16-B aligned code rows");
        return FQSS_EINVAL;
    }
    return launch_status("fqss_frames_wgrad1_q");
}
