// lstm.hip -- the recurrence of LSTMQ (qat_layers.py:571-600: torch's fused bidirectional LSTM on fake-quantized weights,
// zero initial state).  The input projection x W_ih^T + b_ih of BOTH directions is one row GEMM (fqss_rowlin_fwd, Co = 8H)
// in front of this kernel and the weight / input gradients are row GEMMs behind the backward kernel; what is left here is the
// strictly sequential part: per step one [NB x H] x [H x 4H] product and the cell update.
//
// MI355X mapping.  Sequences are independent, so a workgroup owns NB = 2 sequences of one direction for ALL S steps: no grid
// synchronisation, the whole recurrence is one launch per layer (388 sequence-directions -> 194 workgroups on 256 CUs).
// With H = 128 the recurrent matrix is 512 x 128 fp32 = 256 KB: too big for LDS (160 KB) but it fits the register file of
// ONE workgroup -- 512 threads x 128 VGPRs of weights for the whole kernel, so a step reads nothing from memory but the
// pre-computed input projection (prefetched one step ahead).  In the forward a quad of lanes shares four gate rows and each lane
// keeps one k-quarter of them: a lane reads a quarter of h_{t-1} per step from LDS (all of it per lane = 512 KB of LDS reads per
// step was the whole step time: 430 -> 359 us), partial sums meet by quad-permute DPP adds.  The backward kernel holds the same matrix column-wise
// (thread (k, gate block) keeps W_hh[block*H .. +H][k]) for dh_{t-1} = dgates_t W_hh.
// A generic variant (template H = 0) streams W_hh from L2 instead; it serves odd sizes (tests) only.
// Measured dead ends (round 1 and 2): v_pk_fma_f32, pairing either the two sequences (weight splat: 433 -> 1141 us, the splat pairs
// double the live registers) or two consecutive k (weights and h naturally pairwise in registers, half the FMA instructions, no extra
// moves: 383 -> 376 us) -- a wave64 v_pk_fma_f32 occupies the SIMD for two plain FMAs' time here, the step stays at ~3,650 cycles:
// 2 waves x (256 FMA + ~210 other) VALU issues x 4 cycles per SIMD.  An MFMA form does not pay either: v_mfma_f32_4x4x1 (the only fp32
// shape whose N = 4 is not mostly padding at 2 sequences) has the plain-FMA rate at N = 2, and more sequences per workgroup do not
// shorten a step (there are <= 250 sequences for 256 CUs: one workgroup per CU already; the recurrence is latency-, not throughput-bound).
// What did help in round 2: each gate row applies its own sigmoid / tanh (all 8 waves, instead of the cell threads doing five
// transcendentals per element on 4 waves) and the input projection is requested four steps ahead (409 -> 383 us per layer at S = 250);
// in the backward the quad / row-quarter split of the dgate image (the forward's trick: 64 -> 16 LDS reads per wave and step) and the
// same four-step register ring for the saved activations: cfg 3 38.6 -> 37.2 ms per step.
//
// Saved for the backward: the four gate activations and the cell state per step (5H floats per step and sequence-direction).
#include "fqss_dev.h"

namespace fqss {

constexpr int kNB = 2;   // sequences per workgroup

struct LstmBiasGrads {      // gradient buffers of b_ih / b_hh, forward and reverse direction ([4H] each; null: not wanted)
    float *ih_f, *hh_f, *ih_r, *hh_r;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// pre   [S][B][2][4H]  input projection incl. b_ih (gate order i, f, g, o)
// whh   [2][4H][H], bhh [2][4H]
// hout  [S][B][2H]     (forward | reverse halves)
// gsav  [S][B][2][4H]  gate activations, csav [S][B][2][2][H] cell states c | tanh(c) (the backward needs both: no tanh in its loop)
template <int HT>
__global__ __launch_bounds__(HT > 0 ? 4 * HT : 1024) void k_lstm_fwd(const float* __restrict__ pre, const float* __restrict__ whh,
                                                                      const float* __restrict__ bhh, float* __restrict__ hout,
                                                                      float* __restrict__ gsav, float* __restrict__ csav, int S,
                                                                      int B, int Hrt) {
    const int H = HT > 0 ? HT : Hrt;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* hs = smem;                 // [kNB][H]
    float* gs = smem + kNB * (H + 16);       // [kNB][4H]   (hs: [kNB][4 quarters][H/4 + 4] when H is compile-time)
    const int dir = blockIdx.y;
    const int b0 = blockIdx.x * kNB;
    const int j = threadIdx.x;        // gate row (j < 4H)
    const bool jv = j < 4 * H;
    const float* W = whh + ((int64_t)dir * 4 * H + (jv ? j : 0)) * H;
    // HT > 0: a QUAD of lanes shares four gate rows (4q .. 4q+3), lane kq of the quad keeps the k-quarter [32 kq, 32 kq + 32) of
    // each of them: the same 128 weight registers as "one row per thread", but a lane now reads a quarter of h per step instead
    // of all of it (every lane fetching all 2 x 128 values was 512 KB of LDS reads per step = the whole step time at 128 B/clk);
    // the four partial sums of a row meet by two quad-permute DPP adds
    constexpr int KQ = HT > 0 ? HT / 4 : 1;                 // k values per lane and row
    constexpr int HQS = KQ + 4;                             // quarter stride in LDS (floats): the 4 quarters hit different banks
    const int q4 = j >> 2, kq = j & 3;
    float wreg[HT > 0 ? HT : 1];
    if constexpr (HT > 0) {
        const float* Wq = whh + ((int64_t)dir * 4 * H + q4 * 4) * H + kq * KQ;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < KQ; ++k) wreg[r * KQ + k] = Wq[(int64_t)r * H + k];
    }
    const float bj = jv ? bhh[dir * 4 * H + j] : 0.f;
    for (int e = threadIdx.x; e < kNB * (H + 16); e += blockDim.x) hs[e] = 0.f;
    float c = 0.f;                    // cell state of (nb, k) = (threadIdx / H, threadIdx % H) for threadIdx < kNB*H
    const int cn = threadIdx.x / H, ck = threadIdx.x - cn * H;
    const bool cell = threadIdx.x < kNB * H && (b0 + cn) < B;
    __syncthreads();
    // The input projection of a step is requested kPF steps before it is used (a ring of registers): a step is ~1 us of work, a
    // global-memory round trip is about as long, and with a one-step lookahead every step ended waiting for it.  Unconditional
    // loads from clamped indices (a branch around a global load makes the compiler wait for it on the spot; out-of-range rows /
    // gate slots read valid memory and are unused).
    constexpr int kPF = 4;
    float pf[kPF][kNB];
    const int jc = jv ? j : 4 * H - 1;
    auto fetch = [&](float (&dst)[kNB], int step) {
        const int sn = min(step, S - 1);
        const int tn = dir == 0 ? sn : S - 1 - sn;
#pragma unroll
        for (int nb = 0; nb < kNB; ++nb) dst[nb] = pre[(((int64_t)tn * B + min(b0 + nb, B - 1)) * 2 + dir) * 4 * H + jc];
    };
#pragma unroll
    for (int u = 0; u < kPF; ++u) fetch(pf[u], u);
    auto one_step = [&](int step, const float (&pcur)[kNB]) {
        const int t = dir == 0 ? step : S - 1 - step;
        if (jv) {
            float acc[kNB];
#pragma unroll
            for (int nb = 0; nb < kNB; ++nb) acc[nb] = 0.f;
            if constexpr (HT > 0) {
                float part[4][kNB];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int nb = 0; nb < kNB; ++nb) part[r][nb] = 0.f;
#pragma unroll
                for (int k = 0; k < KQ; k += 4) {
#pragma unroll
                    for (int nb = 0; nb < kNB; ++nb) {
                        const float4 hv = *reinterpret_cast<const float4*>(hs + (nb * 4 + kq) * HQS + k);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            part[r][nb] = fmaf(wreg[r * KQ + k], hv.x, part[r][nb]);
                            part[r][nb] = fmaf(wreg[r * KQ + k + 1], hv.y, part[r][nb]);
                            part[r][nb] = fmaf(wreg[r * KQ + k + 2], hv.z, part[r][nb]);
                            part[r][nb] = fmaf(wreg[r * KQ + k + 3], hv.w, part[r][nb]);
                        }
                    }
                }
                // sum over the quad (lanes kq = 0..3), then lane kq keeps row 4q + kq = its own row j
#pragma unroll
                for (int nb = 0; nb < kNB; ++nb) {
                    float mine = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float t = part[r][nb];
                        t += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(t), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
                        t += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(t), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
                        mine = (kq == r) ? t : mine;
                    }
                    acc[nb] = mine;
                }
            } else {
                for (int k = 0; k < H; ++k) {
                    const float wv = W[k];
#pragma unroll
                    for (int nb = 0; nb < kNB; ++nb) acc[nb] = fmaf(wv, hs[nb * H + k], acc[nb]);
                }
            }
            // every gate row applies its OWN non-linearity here, all waves busy and the kNB sequences of a lane independent (the cell
            // threads used to do all four per element: 3 sigmoids + 2 tanh as one dependent chain on half of the waves, the longest
            // part of a step), and writes the saved activation itself: one contiguous 4H row per sequence
            const bool is_g = (j >= 2 * H) && (j < 3 * H);      // gate order i, f, g, o; wave-uniform for H % 64 == 0
#pragma unroll
            for (int nb = 0; nb < kNB; ++nb) {
                const float pre_act = pcur[nb] + (acc[nb] + bj);
                const float a = is_g ? tanhf(pre_act) : sigmoidf_(pre_act);
                gs[nb * 4 * H + j] = a;
                if (gsav != nullptr && b0 + nb < B) gsav[((((int64_t)t * B + b0 + nb) * 2) + dir) * 4 * H + j] = a;   // (NULL: inference)
            }
        }
        __syncthreads();
        if (cell) {
            const float* g = gs + cn * 4 * H;
            const float gi = g[ck], gf = g[H + ck], gg = g[2 * H + ck], go = g[3 * H + ck];
            c = gf * c + gi * gg;
            const float tc = tanhf(c);
            const float h = go * tc;
            if constexpr (HT > 0) hs[(cn * 4 + ck / KQ) * HQS + (ck % KQ)] = h;
            else hs[cn * H + ck] = h;
            const int64_t sb = (int64_t)t * B + b0 + cn;
            hout[sb * 2 * H + dir * H + ck] = h;
            if (csav != nullptr) {
                csav[(sb * 2 + dir) * 2 * H + ck] = c;
                csav[(sb * 2 + dir) * 2 * H + H + ck] = tc;
            }
        }
        __syncthreads();
    };
    int step = 0;
    for (; step + kPF <= S; step += kPF) {
#pragma unroll
        for (int u = 0; u < kPF; ++u) {
            float pc[kNB];
#pragma unroll
            for (int nb = 0; nb < kNB; ++nb) pc[nb] = pf[u][nb];
            fetch(pf[u], step + u + kPF);
            one_step(step + u, pc);
        }
    }
#pragma unroll
    for (int u = 0; u < kPF - 1; ++u)       // the last S % kPF steps: already in the ring
        if (step + u < S) one_step(step + u, pf[u]);
}

// gout [S][B][2H] -> dG [S][B][2][4H] (gradient w.r.t. the gate pre-activations); everything else follows by GEMMs
template <int HT>
__global__ __launch_bounds__(HT > 0 ? 4 * HT : 1024) void k_lstm_bwd(const float* __restrict__ gout, const float* __restrict__ whh,
                                                                      const float* __restrict__ gsav, const float* __restrict__ csav,
                                                                      float* __restrict__ dG, float* __restrict__ gbias, const LstmBiasGrads gb4, int S, int B, int Hrt) {
    const int H = HT > 0 ? HT : Hrt;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dgs = smem;                   // [kNB][4H]   (HT > 0: [kNB][4 blocks][4 quarters][HT/4 + 4])
    float* ps = smem + (HT > 0 ? kNB * 16 * (HT / 4 + 4) : kNB * 4 * H);      // [4][kNB][H] partial dh_{t-1}
    const int dir = blockIdx.y;
    const int b0 = blockIdx.x * kNB;
    const int tid = threadIdx.x;
    const bool tv = tid < 4 * H;
    const int part = tv ? tid / H : 0, k = tv ? tid - part * H : 0;
    const float* Wc = whh + ((int64_t)dir * 4 * H + part * H) * H + k;   // column k of gate block `part`, row stride H
    // HT > 0: a QUAD of lanes shares the four columns 4q .. 4q+3 of its gate block, lane kq of the quad keeps the row quarter
    // [32 kq, 32 kq + 32) of each of them -- the same 128 weight registers as "one column per thread", but a lane reads a quarter of the
    // block's dgates per step instead of all of them (64 -> 16 ds_read_b128 per wave and step: the LDS pipe of the CU was the longest
    // part of a step, as in the forward); the four partial sums of a column meet by two quad-permute DPP adds
    constexpr int RQ = HT > 0 ? HT / 4 : 1;                 // rows per lane and column
    constexpr int DQS = RQ + 4;                             // quarter stride of the dgate image in LDS (floats): quarters on different banks
    const int q4 = k >> 2, kq = k & 3;
    float wreg[HT > 0 ? HT : 1];
    if constexpr (HT > 0) {
        const float* Wq = whh + ((int64_t)dir * 4 * H + part * H + kq * RQ) * H + q4 * 4;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int r = 0; r < RQ; ++r) wreg[cc * RQ + r] = Wq[(int64_t)r * HT + cc];
    }
    for (int e = tid; e < 4 * kNB * H; e += blockDim.x) ps[e] = 0.f;
    const int cn = tid / H, ck = tid - cn * H;
    const bool cell = tid < kNB * H && (b0 + cn) < B;
    float dc_rec = 0.f;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};       // column sums of this thread's dgates over the steps: the bias gradient (gbias)
    // saved gate activations, cell state and incoming gradient of a step are fetched one step ahead (unconditional loads from
    // clamped indices, see k_lstm_fwd): without the prefetch every step began with an exposed global-memory round trip
    const int cnc = min(b0 + (cell ? cn : 0), B - 1), ckc = cell ? ck : 0;
    // ... four steps ahead since round 2 (a ring of registers, as in k_lstm_fwd): a step is about as long as a global-memory round trip
    struct BSt { float dh, i, f, g, o, c, cp; };
    constexpr int kPF = 4;
    BSt pf[kPF];
    auto fetch = [&](BSt& d, int step) {
        const int sc = max(step, 0);
        const int t = dir == 0 ? sc : S - 1 - sc;
        const int64_t sb = (int64_t)t * B + cnc;
        d.dh = gout[sb * 2 * H + dir * H + ckc];
        const float* gsv = gsav + (sb * 2 + dir) * 4 * H;
        d.i = gsv[ckc]; d.f = gsv[H + ckc]; d.g = gsv[2 * H + ckc]; d.o = gsv[3 * H + ckc];
        d.c = csav[(sb * 2 + dir) * 2 * H + H + ckc];      // tanh(c_t), saved by the forward
        const int sp = max(sc - 1, 0);
        const int tp = dir == 0 ? sp : S - 1 - sp;
        d.cp = csav[((((int64_t)tp * B + cnc) * 2) + dir) * 2 * H + ckc];
    };
#pragma unroll
    for (int u = 0; u < kPF; ++u) fetch(pf[u], S - 1 - u);
    __syncthreads();
    auto one_step = [&](int step, const BSt& v) {
        const int t = dir == 0 ? step : S - 1 - step;
        const float c_dh = v.dh, gi = v.i, gf = v.f, gg = v.g, go = v.o, cc = v.c, cprev = step > 0 ? v.cp : 0.f;
        if (cell) {
            const int64_t sb = (int64_t)t * B + b0 + cn;
            const float dh = c_dh +
                             ((ps[(0 * kNB + cn) * H + ck] + ps[(1 * kNB + cn) * H + ck]) + (ps[(2 * kNB + cn) * H + ck] + ps[(3 * kNB + cn) * H + ck]));
            const float tc = cc;                   // tanh(c_t)
            const float dc = dc_rec + (dh * go) * (1.0f - tc * tc);
            const float d_o = ((dh * tc) * (1.0f - go)) * go;
            const float d_i = ((dc * gg) * (1.0f - gi)) * gi;
            const float d_f = ((dc * cprev) * (1.0f - gf)) * gf;
            const float d_g = (dc * gi) * (1.0f - gg * gg);
            dc_rec = dc * gf;
            bsum[0] += d_i; bsum[1] += d_f; bsum[2] += d_g; bsum[3] += d_o;
            if constexpr (HT > 0) {
                float* d = dgs + cn * 16 * DQS + (ck / RQ) * DQS + (ck % RQ);        // [nb][gate block][row quarter][RQ + pad]
                d[0] = d_i; d[4 * DQS] = d_f; d[8 * DQS] = d_g; d[12 * DQS] = d_o;
            } else {
                float* d = dgs + cn * 4 * H;
                d[ck] = d_i; d[H + ck] = d_f; d[2 * H + ck] = d_g; d[3 * H + ck] = d_o;
            }
            float* o = dG + (sb * 2 + dir) * 4 * H;
            o[ck] = d_i; o[H + ck] = d_f; o[2 * H + ck] = d_g; o[3 * H + ck] = d_o;
        }
        __syncthreads();
        if (tv) {
            float acc[kNB];
#pragma unroll
            for (int nb = 0; nb < kNB; ++nb) acc[nb] = 0.f;
            if constexpr (HT > 0) {
                float pc[4][kNB];
#pragma unroll
                for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                    for (int nb = 0; nb < kNB; ++nb) pc[cc][nb] = 0.f;
#pragma unroll
                for (int r = 0; r < RQ; r += 4) {
#pragma unroll
                    for (int nb = 0; nb < kNB; ++nb) {
                        const float4 dv = *reinterpret_cast<const float4*>(dgs + nb * 16 * DQS + (part * 4 + kq) * DQS + r);
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) {
                            pc[cc][nb] = fmaf(wreg[cc * RQ + r], dv.x, pc[cc][nb]);
                            pc[cc][nb] = fmaf(wreg[cc * RQ + r + 1], dv.y, pc[cc][nb]);
                            pc[cc][nb] = fmaf(wreg[cc * RQ + r + 2], dv.z, pc[cc][nb]);
                            pc[cc][nb] = fmaf(wreg[cc * RQ + r + 3], dv.w, pc[cc][nb]);
                        }
                    }
                }
                // sum over the quad (lanes kq = 0..3), then lane kq keeps column 4q + kq = its own column k
#pragma unroll
                for (int nb = 0; nb < kNB; ++nb) {
                    float mine = 0.f;
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        float t = pc[cc][nb];
                        t += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(t), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
                        t += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(t), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
                        mine = (kq == cc) ? t : mine;
                    }
                    acc[nb] = mine;
                }
            } else {
                for (int r = 0; r < H; ++r) {
                    const float wv = Wc[(int64_t)r * H];
#pragma unroll
                    for (int nb = 0; nb < kNB; ++nb) acc[nb] = fmaf(wv, dgs[nb * 4 * H + part * H + r], acc[nb]);
                }
            }
#pragma unroll
            for (int nb = 0; nb < kNB; ++nb) ps[(part * kNB + nb) * H + k] = acc[nb];
        }
        __syncthreads();
    };
    int done = 0;                               // steps S-1, S-2, ... are processed in rounds of the ring
    for (; done + kPF <= S; done += kPF) {
#pragma unroll
        for (int u = 0; u < kPF; ++u) {
            const BSt cur = pf[u];
            fetch(pf[u], S - 1 - (done + u + kPF));
            one_step(S - 1 - (done + u), cur);
        }
    }
#pragma unroll
    for (int u = 0; u < kPF - 1; ++u)           // the last S % kPF steps: already in the ring
        if (done + u < S) one_step(S - 1 - (done + u), pf[u]);
    // gbias: ONE [2][4H] buffer (fqss_lstm_bwd_b);  gb4 (fqss_lstm_bwd_b4): the four bias parameters' own gradient buffers, [4H] each
    if (cell) {
        float* p0 = dir == 0 ? gb4.ih_f : gb4.ih_r;
        float* p1 = dir == 0 ? gb4.hh_f : gb4.hh_r;
#pragma unroll
        for (int gt = 0; gt < 4; ++gt) {
            if (gbias != nullptr) atomicAdd(gbias + (dir * 4 + gt) * H + ck, bsum[gt]);
            if (p0 != nullptr) atomicAdd(p0 + gt * H + ck, bsum[gt]);
            if (p1 != nullptr) atomicAdd(p1 + gt * H + ck, bsum[gt]);
        }
    }
}

}  // namespace fqss

using namespace fqss;

extern "C" int fqss_lstm_fwd(const float* pre, const float* whh, const float* bhh, float* hout, float* gsav, float* csav,
                             int S, int B, int H, fqss_stream_t stream) {
    FQSS_REQUIRE(pre && whh && bhh && hout && ((gsav == nullptr) == (csav == nullptr)), "null tensor (gsav / csav: both or neither)");
    FQSS_REQUIRE(S > 0 && B > 0 && H > 0 && H <= 256, "bad shape (H <= 256)");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)cdiv(B, kNB), 2);
    const size_t lds = (size_t)(kNB * (H + 16) + kNB * 4 * H) * sizeof(float);
    if (H == 128) {
        hipLaunchKernelGGL((k_lstm_fwd<128>), grid, dim3(512), lds, s, pre, whh, bhh, hout, gsav, csav, S, B, H);
    } else {
        const int threads = (int)cdiv(4 * H, 64) * 64;
        hipLaunchKernelGGL((k_lstm_fwd<0>), grid, dim3(threads), lds, s, pre, whh, bhh, hout, gsav, csav, S, B, H);
    }
    return launch_status("fqss_lstm_fwd");
}

static int lstm_bwd_impl(const char* who, const float* gout, const float* whh, const float* gsav, const float* csav, float* dG, float* gbias,
                         int S, int B, int H, fqss_stream_t stream, LstmBiasGrads gb4 = LstmBiasGrads{nullptr, nullptr, nullptr, nullptr}) {
    FQSS_REQUIRE(gout && whh && gsav && csav && dG, "null tensor");
    FQSS_REQUIRE(S > 0 && B > 0 && H > 0 && H <= 256, "bad shape (H <= 256)");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)cdiv(B, kNB), 2);
    const size_t lds = (size_t)(kNB * 4 * H + 4 * kNB * H) * sizeof(float);
    if (H == 128) {
        const size_t lds128 = (size_t)(kNB * 16 * (128 / 4 + 4) + 4 * kNB * H) * sizeof(float);   // padded dgate image + partial dh
        hipLaunchKernelGGL((k_lstm_bwd<128>), grid, dim3(512), lds128, s, gout, whh, gsav, csav, dG, gbias, gb4, S, B, H);
    } else {
        const int threads = (int)cdiv(4 * H, 64) * 64;
        hipLaunchKernelGGL((k_lstm_bwd<0>), grid, dim3(threads), lds, s, gout, whh, gsav, csav, dG, gbias, gb4, S, B, H);
    }
    return launch_status(who);
}

extern "C" int fqss_lstm_bwd(const float* gout, const float* whh, const float* gsav, const float* csav, float* dG, int S,
                             int B, int H, fqss_stream_t stream) {
    return lstm_bwd_impl("fqss_lstm_bwd", gout, whh, gsav, csav, dG, nullptr, S, B, H, stream);
}

// the same, and the column sums of dG (= the gradient of b_ih and of b_hh, [2][4H]) ADDED into gbias: the cell threads keep them in
// registers over the steps (a separate pass over dG, the largest tensor of the layer, was 34 us per LSTM at cfg 3)
extern "C" int fqss_lstm_bwd_b(const float* gout, const float* whh, const float* gsav, const float* csav, float* dG, float* gbias, int S,
                               int B, int H, fqss_stream_t stream) {
    FQSS_REQUIRE(gbias, "null tensor");
    return lstm_bwd_impl("fqss_lstm_bwd_b", gout, whh, gsav, csav, dG, gbias, S, B, H, stream);
}

// ... and with the sums ADDED straight into the four bias parameters' own gradient buffers (gb[0..3] = b_ih, b_hh forward, b_ih, b_hh
// reverse; [4H] each, e.g. their slots of a flat gradient arena): no [8H] intermediate, no four adds behind the kernel
extern "C" int fqss_lstm_bwd_b4(const float* gout, const float* whh, const float* gsav, const float* csav, float* dG, float* const* gb,
                                int S, int B, int H, fqss_stream_t stream) {
    FQSS_REQUIRE(gb && gb[0] && gb[1] && gb[2] && gb[3], "null tensor");
    return lstm_bwd_impl("fqss_lstm_bwd_b4", gout, whh, gsav, csav, dG, nullptr, S, B, H, stream, LstmBiasGrads{gb[0], gb[1], gb[2], gb[3]});
}
