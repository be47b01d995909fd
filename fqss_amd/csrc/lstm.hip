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
#include <cstdlib>

#define FQSS_USES_GRAD_ADD   // the fp32 gradient atomics of this file go through grad_add (fqss_dev.h: FQSS_DETERMINISTIC=1)
#include "fqss_dev.h"

namespace fqss {

constexpr int kNB = 2;   // sequences per workgroup

// timing experiments only (make variant SRC=lstm DEFS=-DFQSS_LSTM_ABL=n; never in the product library): 1 no FMAs, 2 no LDS reads of h,
// 4 no gate transcendentals, 8 no tanh(c), 16 no global stores, 32 no input-projection loads
#ifndef FQSS_LSTM_ABL
#define FQSS_LSTM_ABL 0
#endif

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
        for (int nb = 0; nb < kNB; ++nb)
            dst[nb] = (FQSS_LSTM_ABL & 32) != 0 ? 0.25f : pre[(((int64_t)tn * B + min(b0 + nb, B - 1)) * 2 + dir) * 4 * H + jc];
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
                        float4 hv;
                        if constexpr ((FQSS_LSTM_ABL & 2) != 0) hv = make_float4(wreg[k], wreg[k + 1], wreg[k + 2], wreg[k + 3]);
                        else hv = *reinterpret_cast<const float4*>(hs + (nb * 4 + kq) * HQS + k);
                        if constexpr ((FQSS_LSTM_ABL & 1) != 0) {
                            part[k & 3][nb] += (hv.x + hv.y) + (hv.z + hv.w);
                            continue;
                        }
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
                const float a = (FQSS_LSTM_ABL & 4) != 0 ? pre_act * 0.01f : (is_g ? tanhf(pre_act) : sigmoidf_(pre_act));
                gs[nb * 4 * H + j] = a;
                if ((FQSS_LSTM_ABL & 16) == 0 && gsav != nullptr && b0 + nb < B) gsav[((((int64_t)t * B + b0 + nb) * 2) + dir) * 4 * H + j] = a;   // (NULL: inference)
            }
        }
        __syncthreads();
        if (cell) {
            const float* g = gs + cn * 4 * H;
            const float gi = g[ck], gf = g[H + ck], gg = g[2 * H + ck], go = g[3 * H + ck];
            c = gf * c + gi * gg;
            const float tc = (FQSS_LSTM_ABL & 8) != 0 ? c * 0.5f : tanhf(c);
            const float h = go * tc;
            if constexpr (HT > 0) hs[(cn * 4 + ck / KQ) * HQS + (ck % KQ)] = h;
            else hs[cn * H + ck] = h;
            const int64_t sb = (int64_t)t * B + b0 + cn;
            if ((FQSS_LSTM_ABL & 16) == 0 || step == S - 1) hout[sb * 2 * H + dir * H + ck] = h;
            if ((FQSS_LSTM_ABL & 16) == 0 && csav != nullptr) {
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

// Gate non-linearities of the H = 128 forward: ONE short dependent chain for both kinds (a step's critical path runs through it
// twice: gate, then tanh(c)), fp32-grade results.  e = 2^t on v_exp_f32 with t = -x log2(e) (sigmoid) or -2 |x| log2(e) (tanh), the
// rounding of that product and of the constant carried as a first-order factor (1 + t_lo ln 2) folded into the FMAs that form
// 1 +- e; the quotient n / (1 + e) (n = 1: sigmoid, n = 1 - e: tanh) as v_rcp_f32 and one residual correction; tanh below |x| = 0.36,
// where 1 - e cancels, is x + x^3 P(x^2) (P: the degree-3 interpolant of (tanh(sqrt u) / sqrt u - 1) / u on the Chebyshev nodes of
// [0, 0.36^2], 2.3e-9 relative in exact arithmetic).  Arguments are clamped to +-87 (the results there are already 0 / 1 / +-1 to
// fp32).  Measured against float64 (tools/lstm_gate_err.py, tests/test_gpu_dptnet.py::test_lstm_gate_functions): sigmoid <= 3.5 ulp
// (<= 1.5 for x > 0), tanh <= 2.6 ulp, either within 1.0e-7 of the exact value -- v_exp_f32 itself is good to ~1.5 ulp.
__device__ __forceinline__ float gate_fn(float x, bool th) {
    const float a = fabsf(x);
    const float z = __builtin_amdgcn_fmed3f(th ? a : x, -87.0f, 87.0f);
    const float kh = th ? -2.8853900432586669921875f : -1.44269502162933349609375f;      // fl(-2 log2 e), fl(-log2 e)
    const float kl = th ? -3.851925849e-08f : -1.925962925e-08f;                             // the constants' own rounding
    const float t = kh * z;
    const float tl = fmaf(kl, z, fmaf(kh, z, -t));
    const float e = __builtin_amdgcn_exp2f(t);
    const float corr = fmaf(tl, 0.693147182464599609375f, 1.0f);
    const float d = fmaf(e, corr, 1.0f);
    const float n = th ? fmaf(-e, corr, 1.0f) : 1.0f;
    const float r = __builtin_amdgcn_rcpf(d);
    float q = n * r;
    q = fmaf(fmaf(-d, q, n), r, q);
    // v_med3_f32 returns min3 when an input is NaN: the clamp above would launder a NaN pre-activation into sigmoid ~ 1.7e-38.  A NaN
    // must stay a NaN (libm's did): one v_cmp_u + v_cndmask off the critical chain
    if (!th) return (x != x) ? x : q;
    const float u = a * a;
    float p = fmaf(u, 0.019728317856788635f, -0.0537986196577549f);
    p = fmaf(u, p, 0.13332897424697876f);
    p = fmaf(u, p, -0.3333333134651184f);
    const float small = fmaf(a * u, p, a);
    return copysignf(a < 0.36f ? small : q, x);
}

__global__ void k_lstm_gate_fn(const float* __restrict__ x, float* __restrict__ sg, float* __restrict__ th, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        sg[i] = gate_fn(x[i], false);
        th[i] = gate_fn(x[i], true);
    }
}

// The H = 128 forward, round 4.  What a step of the round-2 form costs (`profiles/r04_lstm_ablation.txt`, ~3,300 cycles at S = 250 x 194
// sequences; `profiles/r04_lstm_pmc_v1_vs_quad.txt`): 359 vector instructions per wave and step, 256 of them the FMAs (~1,300 cycles: two
// waves per SIMD issue FMAs at the SIMD's rate, tools/ubench/dpp_fma_rate.hip), the gate non-linearities ~670 (libm's expf / tanhf and an
// IEEE division: ~40 instructions each, two per lane), the LDS reads of h ~650 in front of the FMAs, tanh(c) + stores + loads ~430, the
// two barriers and the LDS round trips of the cell update ~480.  It is the NUMBER of non-FMA instructions that the time follows (~5 cycles
// of the SIMD each, the two waves of a SIMD hardly overlapping there), not the depth of the dependent chains.  Measured on the way:
//   * sixteen waves, an OCT of lanes per four rows (64 weight registers, 4 waves per SIMD): 425 us against 390 -- the FMAs are no faster
//     (two waves already saturate the SIMD), the reduction and the barrier grow;
//   * the four gates of a hidden unit in ONE quad, the cell update inside the quad by quad broadcasts, one barrier per step: 361-378 us --
//     every lane then evaluates tanh(c) (4 x the evaluations), 422 instructions per wave and step instead of 359; the same with the chain
//     dealt out between the FMAs by hand (a scheduling barrier per k): 464 us, all eight LDS reads of a phase are in flight at once;
//   * `v_fmac_f32_dpp row_newbcast` to share h across 16 lanes instead of LDS reads: 18 x slower than a plain FMA on this part.
// This form (339 us at the same shape on the probe's random operands, -13 %; inside the cfg-3 step the launches average 298 us against
// 303 -- profiles/r04_cfg3_step_table.txt -- : the model's narrow pre-activations keep libm on its short paths): the round-2 layout (a quad shares four gate rows, the gate kind is wave-uniform), gate
// functions of 10-20 instructions (above), and the two sequences HALF A STEP APART: a phase is [the cell update of the OTHER sequence, on
// two waves] and [this sequence's 128 FMAs per lane on all waves], then this sequence's gate function; one barrier per phase, two per
// step as before, and a phase reads half of the h bytes in one burst.  Same sums in the same order as the round-2 form; the gate
// functions differ from libm's by <= 4 ulp (tests/test_gpu_dptnet.py::test_lstm_gate_functions).
template <int H>
__global__ __launch_bounds__(4 * H) void k_lstm_fwd_st(const float* __restrict__ pre, const float* __restrict__ whh, const float* __restrict__ bhh,
                                                       float* __restrict__ hout, float* __restrict__ gsav, float* __restrict__ csav, int S, int B) {
    static_assert(kNB == 2 && H % 64 == 0, "two sequences half a step apart; wave-uniform gate kinds and cell roles");
    constexpr int KQ = H / 4, HQS = KQ + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* hs = smem;                              // [kNB][4 quarters][HQS]
    float* gs = smem + kNB * 4 * HQS;              // [kNB][4H]
    const int dir = blockIdx.y;
    const int b0 = blockIdx.x * kNB;
    const int j = threadIdx.x;                     // gate row
    const int q4 = j >> 2, kq = j & 3;
    float wreg[H];
    {
        const float* Wq = whh + ((int64_t)dir * 4 * H + q4 * 4) * H + kq * KQ;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < KQ; ++k) wreg[r * KQ + k] = Wq[(int64_t)r * H + k];
    }
    const float bj = bhh[dir * 4 * H + j];
    for (int e = threadIdx.x; e < kNB * 4 * HQS; e += blockDim.x) hs[e] = 0.f;
    // cell roles: threads [0, H) keep sequence 1's cell state, threads [H, 2H) sequence 0's (whole waves; each role's update sits
    // in the phase of the OTHER sequence's FMAs)
    const int crole = j < H ? 1 : (j < 2 * H ? 0 : -1);
    const int ck = j & (H - 1);
    const bool cell_ok = crole >= 0 && b0 + crole < B;
    float c = 0.f;
    __syncthreads();
    constexpr int kPF = 4;
    float pf[kPF][kNB];
    auto fetch = [&](float (&dst)[kNB], int step) {
        const int sn = min(step, S - 1);
        const int tn = dir == 0 ? sn : S - 1 - sn;
#pragma unroll
        for (int nb = 0; nb < kNB; ++nb) dst[nb] = pre[(((int64_t)tn * B + min(b0 + nb, B - 1)) * 2 + dir) * 4 * H + j];
    };
#pragma unroll
    for (int u = 0; u < kPF; ++u) fetch(pf[u], u);
    const bool is_g = (j >= 2 * H) && (j < 3 * H);           // gate order i, f, g, o; wave-uniform
    auto time_of = [&](int step) { return dir == 0 ? step : S - 1 - step; };
    auto cell_update = [&](int nb, int step) {               // by the threads whose role is nb (wave-uniform branch at the call)
        const int t = time_of(step);
        const float* g = gs + nb * 4 * H;
        const float gi = g[ck], gf = g[H + ck], gg = g[2 * H + ck], go = g[3 * H + ck];
        c = gf * c + gi * gg;
        const float tc = gate_fn(c, true);
        const float h = go * tc;
        hs[(nb * 4 + ck / KQ) * HQS + (ck % KQ)] = h;
        const int64_t sb = (int64_t)t * B + b0 + nb;
        hout[sb * 2 * H + dir * H + ck] = h;
        if (csav != nullptr) {
            csav[(sb * 2 + dir) * 2 * H + ck] = c;
            csav[(sb * 2 + dir) * 2 * H + H + ck] = tc;
        }
    };
    auto gates = [&](int nb, int step, float pcur) {         // W_hh h_{t-1} of sequence nb, its gate function, the saved activation
        const int t = time_of(step);
        float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KQ; k += 4) {
            const float4 hv = *reinterpret_cast<const float4*>(hs + (nb * 4 + kq) * HQS + k);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                part[r] = fmaf(wreg[r * KQ + k], hv.x, part[r]);
                part[r] = fmaf(wreg[r * KQ + k + 1], hv.y, part[r]);
                part[r] = fmaf(wreg[r * KQ + k + 2], hv.z, part[r]);
                part[r] = fmaf(wreg[r * KQ + k + 3], hv.w, part[r]);
            }
        }
        float mine = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = part[r];
            v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
            v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
            mine = (kq == r) ? v : mine;
        }
        const float pre_act = pcur + (mine + bj);
        const float a = is_g ? gate_fn(pre_act, true) : gate_fn(pre_act, false);
        gs[nb * 4 * H + j] = a;
        if (gsav != nullptr && b0 + nb < B) gsav[((((int64_t)t * B + b0 + nb) * 2) + dir) * 4 * H + j] = a;
    };
    auto one_step = [&](int step, const float (&pcur)[kNB]) {
        if (crole == 1 && cell_ok && step > 0) cell_update(1, step - 1);
        gates(0, step, pcur[0]);
        __syncthreads();
        if (crole == 0 && cell_ok) cell_update(0, step);
        gates(1, step, pcur[1]);
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
    for (int u = 0; u < kPF - 1; ++u)
        if (step + u < S) one_step(step + u, pf[u]);
    if (crole == 1 && cell_ok) cell_update(1, S - 1);
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
            if (gbias != nullptr) grad_add(gbias + (dir * 4 + gt) * H + ck, bsum[gt]);
            if (p0 != nullptr) grad_add(p0 + gt * H + ck, bsum[gt]);
            if (p1 != nullptr) grad_add(p1 + gt * H + ck, bsum[gt]);
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
    // FQSS_LSTM_V1=1 (read per call: tests and A/B runs toggle it): the round-2 form of the H = 128 forward
    const char* v1 = getenv("FQSS_LSTM_V1");
    if (H == 128 && !(v1 && atoi(v1) != 0)) {
        const size_t lds2 = (size_t)(kNB * 4 * (H / 4 + 4) + kNB * 4 * H) * sizeof(float);
        hipLaunchKernelGGL(k_lstm_fwd_st<128>, grid, dim3(512), lds2, s, pre, whh, bhh, hout, gsav, csav, S, B);
    } else if (H == 128) {
        hipLaunchKernelGGL((k_lstm_fwd<128>), grid, dim3(512), lds, s, pre, whh, bhh, hout, gsav, csav, S, B, H);
    } else {
        const int threads = (int)cdiv(4 * H, 64) * 64;
        hipLaunchKernelGGL((k_lstm_fwd<0>), grid, dim3(threads), lds, s, pre, whh, bhh, hout, gsav, csav, S, B, H);
    }
    return launch_status("fqss_lstm_fwd");
}

// test hook: the gate functions of the H = 128 forward on a vector of arguments (tests/test_gpu_dptnet.py measures them against float64)
extern "C" int fqss_lstm_gate_fn(const float* x, float* sigmoid_out, float* tanh_out, int64_t n, fqss_stream_t stream) {
    FQSS_REQUIRE(x && sigmoid_out && tanh_out, "null tensor");
    FQSS_REQUIRE(n >= 0, "bad size");
    if (n == 0) return FQSS_OK;
    hipLaunchKernelGGL(k_lstm_gate_fn, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, sigmoid_out, tanh_out, n);
    return launch_status("fqss_lstm_gate_fn");
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
