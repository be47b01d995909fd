// attn.hip -- the attention core of MultiheadAttentionQ (qat_layers.py:903-911): softmax(q k^T) v per (sequence, head) for
// the short sequences of the dual-path models (L = chunk length 250 or number of chunks ~200, head_dim 16).
//
// One workgroup per (sequence b, head h).  K and V of the head (L x HD floats each, 16 KB at L = 250) sit in LDS; a thread
// owns whole query rows (q row in registers), so every LDS operand fetch is a wave-wide broadcast (conflict-free) and no
// cross-lane reduction is needed anywhere: s_ij, the row max, exp, the row sum and o_i = sum_j p_ij v_j all stay in the
// owning lane.  fp32 FMA on the vector ALU: the problem is 2*L*L*HD = 2 MFLOP per head -- far too small and too skinny
// (K = 16) to be worth MFMA tiles; the full layer is ~3 GFLOP.  Rows are [l*B + b][E] matrices (sequence-first), head h
// is the column block [h*HD, (h+1)*HD): exactly the reference's reshape(L, B*nh, hd).permute(1, 0, 2) view, without the copy.
//
// The backward recomputes p_ij from the saved row statistics (max, sum) in two owner-computes phases -- thread <-> query row
// for dq, thread <-> key row for dk / dv -- with q, k, v, dO of the head in LDS: no atomics, deterministic.
//
// The reference also runs `activation_fake_quantize_attn(attn)` / `_softmax(attn)` and DISCARDS their results (`attn - fq(attn)`,
// :907, :909): only their observers see data during the first 50 calls.  obs_attn / obs_soft (optional) receive the running
// min / max of the logits and of the probabilities for exactly that purpose.
#include <stdlib.h>

#include "fqss_dev.h"

namespace fqss {

typedef float f32x16a __attribute__((ext_vector_type(16)));

template <int HD>
__device__ __forceinline__ float dot_row(const float (&q)[HD], const float* __restrict__ k) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) s = fmaf(q[d], k[d], s);
    return s;
}

template <int HD>
__global__ __launch_bounds__(256) void k_attn_fwd(const float* __restrict__ q, const float* __restrict__ k,
                                                   const float* __restrict__ v, float* __restrict__ o, float* __restrict__ stats,
                                                   int L, int B, int nh, int64_t ld_q, int64_t ld_k, int64_t ld_v, int64_t ld_o,
                                                   uint32_t* obs_attn, uint32_t* obs_soft) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = smem + (size_t)L * HD;
    const int b = blockIdx.x / nh, h = blockIdx.x % nh;
    for (int e = threadIdx.x; e < L * HD; e += 256) {
        const int j = e / HD, d = e % HD;
        const int64_t row = (int64_t)j * B + b;
        Ks[e] = k[row * ld_k + h * HD + d];
        Vs[e] = v[row * ld_v + h * HD + d];
    }
    __syncthreads();
    float smin_all = INFINITY, smax_all = -INFINITY, pmin_all = INFINITY, pmax_all = -INFINITY;
    for (int i = threadIdx.x; i < L; i += 256) {
        const int64_t row = (int64_t)i * B + b;
        float qr[HD], acc[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) { qr[d] = q[row * ld_q + h * HD + d]; acc[d] = 0.f; }
        float m = -INFINITY, mn = INFINITY;
        for (int j = 0; j < L; ++j) {
            const float s = dot_row<HD>(qr, Ks + j * HD);
            m = fmaxf(m, s);
            mn = fminf(mn, s);
        }
        float l = 0.f;
        for (int j = 0; j < L; ++j) {
            const float p = expf(dot_row<HD>(qr, Ks + j * HD) - m);
            l += p;
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[d] = fmaf(p, Vs[j * HD + d], acc[d]);
        }
#pragma unroll
        for (int d = 0; d < HD; ++d) o[row * ld_o + h * HD + d] = acc[d] / l;
        stats[((int64_t)blockIdx.x * L + i) * 2] = m;
        stats[((int64_t)blockIdx.x * L + i) * 2 + 1] = l;
        smin_all = fminf(smin_all, mn);
        smax_all = fmaxf(smax_all, m);
        pmax_all = fmaxf(pmax_all, 1.0f / l);
        pmin_all = fminf(pmin_all, expf(mn - m) / l);
    }
    if (obs_attn != nullptr) {     // observer phase only (wave-uniform branch)
        smin_all = wave_min(smin_all); smax_all = wave_max(smax_all);
        pmin_all = wave_min(pmin_all); pmax_all = wave_max(pmax_all);
        if ((threadIdx.x & 63) == 0 && smin_all <= smax_all) {
            atomicMin(obs_attn, f2ord(smin_all)); atomicMax(obs_attn + 1, f2ord(smax_all));
            atomicMin(obs_soft, f2ord(pmin_all)); atomicMax(obs_soft + 1, f2ord(pmax_all));
        }
    }
}

template <int HD>
__global__ __launch_bounds__(256) void k_attn_bwd(const float* __restrict__ q, const float* __restrict__ k,
                                                   const float* __restrict__ v, const float* __restrict__ o,
                                                   const float* __restrict__ go, const float* __restrict__ stats,
                                                   float* __restrict__ gq, float* __restrict__ gk, float* __restrict__ gv, int L,
                                                   int B, int nh, int64_t ld_q, int64_t ld_k, int64_t ld_v, int64_t ld_o,
                                                   int64_t ld_go, int64_t ld_gq, int64_t ld_gk, int64_t ld_gv) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qs = smem;
    float* Ks = Qs + (size_t)L * HD;
    float* Vs = Ks + (size_t)L * HD;
    float* Gs = Vs + (size_t)L * HD;
    float* Ms = Gs + (size_t)L * HD;     // row max
    float* Ls = Ms + L;                  // 1 / row sum
    float* Ds = Ls + L;                  // D_i = sum_d dO_i[d] O_i[d]  (= sum_j p_ij dP_ij)
    const int b = blockIdx.x / nh, h = blockIdx.x % nh;
    for (int e = threadIdx.x; e < L * HD; e += 256) {
        const int j = e / HD, d = e % HD;
        const int64_t row = (int64_t)j * B + b;
        Qs[e] = q[row * ld_q + h * HD + d];
        Ks[e] = k[row * ld_k + h * HD + d];
        Vs[e] = v[row * ld_v + h * HD + d];
        Gs[e] = go[row * ld_go + h * HD + d];
    }
    for (int i = threadIdx.x; i < L; i += 256) {
        const int64_t row = (int64_t)i * B + b;
        float dsum = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) dsum = fmaf(go[row * ld_go + h * HD + d], o[row * ld_o + h * HD + d], dsum);
        Ms[i] = stats[((int64_t)blockIdx.x * L + i) * 2];
        Ls[i] = 1.0f / stats[((int64_t)blockIdx.x * L + i) * 2 + 1];
        Ds[i] = dsum;
    }
    __syncthreads();
    // phase A: thread <-> query row i: dq_i = sum_j dS_ij k_j
    for (int i = threadIdx.x; i < L; i += 256) {
        float qr[HD], gr[HD], acc[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) { qr[d] = Qs[i * HD + d]; gr[d] = Gs[i * HD + d]; acc[d] = 0.f; }
        const float m = Ms[i], il = Ls[i], D = Ds[i];
        for (int j = 0; j < L; ++j) {
            const float p = expf(dot_row<HD>(qr, Ks + j * HD) - m) * il;
            const float dS = p * (dot_row<HD>(gr, Vs + j * HD) - D);
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[d] = fmaf(dS, Ks[j * HD + d], acc[d]);
        }
        const int64_t row = (int64_t)i * B + b;
#pragma unroll
        for (int d = 0; d < HD; ++d) gq[row * ld_gq + h * HD + d] = acc[d];
    }
    // phase B: thread <-> key row j: dk_j = sum_i dS_ij q_i, dv_j = sum_i p_ij dO_i
    for (int j = threadIdx.x; j < L; j += 256) {
        float kr[HD], vr[HD], ak[HD], av[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) { kr[d] = Ks[j * HD + d]; vr[d] = Vs[j * HD + d]; ak[d] = av[d] = 0.f; }
        for (int i = 0; i < L; ++i) {
            const float p = expf(dot_row<HD>(kr, Qs + i * HD) - Ms[i]) * Ls[i];
            const float dS = p * (dot_row<HD>(vr, Gs + i * HD) - Ds[i]);
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                ak[d] = fmaf(dS, Qs[i * HD + d], ak[d]);
                av[d] = fmaf(p, Gs[i * HD + d], av[d]);
            }
        }
        const int64_t row = (int64_t)j * B + b;
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            gk[row * ld_gk + h * HD + d] = ak[d];
            gv[row * ld_gv + h * HD + d] = av[d];
        }
    }
}

// head h of a row matrix -> LDS image [Lp][HD + 1], rows >= L zero: 16-B loads when the rows allow it, several in flight per
// thread (unconditional, clamped row; a loop that loads, waits and stores one scalar per iteration spent more time filling LDS
// than the matrix cores spent on the tile products)
template <int HD>
__device__ __forceinline__ void fill_head(const float* __restrict__ src, int64_t ld, float* __restrict__ dst, int L, int Lp, int B, int b,
                                          int h) {
    constexpr int RS = HD + 1, V4 = HD / 4;
    const bool vec = (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0);
    if (vec) {
#pragma unroll 4
        for (int f = threadIdx.x; f < Lp * V4; f += 256) {
            const int j = f / V4, d = (f % V4) * 4;
            const float4 t = *reinterpret_cast<const float4*>(src + ((int64_t)min(j, L - 1) * B + b) * ld + h * HD + d);
            const bool ok = j < L;
            dst[j * RS + d] = ok ? t.x : 0.f;
            dst[j * RS + d + 1] = ok ? t.y : 0.f;
            dst[j * RS + d + 2] = ok ? t.z : 0.f;
            dst[j * RS + d + 3] = ok ? t.w : 0.f;
        }
    } else {
#pragma unroll 4
        for (int e = threadIdx.x; e < Lp * HD; e += 256) {
            const int j = e / HD, d = e % HD;
            const float t = src[((int64_t)min(j, L - 1) * B + b) * ld + h * HD + d];
            dst[j * RS + d] = j < L ? t : 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Matrix-core backward for the production shapes (head_dim 16 / 32, L <= 256): fp32 MFMA 32x32x2 (exact fp32 products and sums).
// The VALU kernel above spends its time on LDS broadcast reads (every (i, j) pair fetches a K / V row for all 64 lanes); here a
// 32 x 32 tile of logits costs head_dim/2 MFMAs.  Two owner-computes loops like above, and NO transposes through LDS: the MFMA
// result layout (lane = column, 16 rows in registers) of
//    S  = Q_it K_jt^T  is exactly the A-operand layout of  dS^T, P^T  for  dK_jt += dS^T Q_it,  dV_jt += P^T dO_it   (loop B),
//    S^T = K_jt Q_it^T is exactly the A-operand layout of  dS        for  dQ_it += dS K_jt                          (loop A),
// because the k index of an MFMA step may be any bijection as long as both operands use the same one (here: the row map of the
// result registers).  Rows of the LDS images are padded to head_dim + 1 floats (A-operand reads walk down a column).
template <int HD>
__global__ __launch_bounds__(256) void k_attn_bwd_mfma(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, const float* __restrict__ o,
                                                        const float* __restrict__ go, const float* __restrict__ stats,
                                                        float* __restrict__ gq, float* __restrict__ gk, float* __restrict__ gv, int L,
                                                        int B, int nh, int64_t ld_q, int64_t ld_k, int64_t ld_v, int64_t ld_o,
                                                        int64_t ld_go, int64_t ld_gq, int64_t ld_gk, int64_t ld_gv) {
    constexpr int RS = HD + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int NT = (L + 31) / 32, Lp = NT * 32;
    float* Qs = smem;
    float* Ks = Qs + (size_t)Lp * RS;
    float* Vs = Ks + (size_t)Lp * RS;
    float* Gs = Vs + (size_t)Lp * RS;
    float* Ms = Gs + (size_t)Lp * RS;    // row max
    float* Is = Ms + Lp;                 // 1 / row sum (0 for padded rows)
    float* Ds = Is + Lp;                 // D_i = sum_d dO_i[d] O_i[d]
    const int b = blockIdx.x / nh, h = blockIdx.x % nh;
    fill_head<HD>(q, ld_q, Qs, L, Lp, B, b, h);
    fill_head<HD>(k, ld_k, Ks, L, Lp, B, b, h);
    fill_head<HD>(v, ld_v, Vs, L, Lp, B, b, h);
    fill_head<HD>(go, ld_go, Gs, L, Lp, B, b, h);
    for (int i = threadIdx.x; i < Lp; i += 256) {
        const int ic = min(i, L - 1);
        const int64_t row = (int64_t)ic * B + b;
        float dsum = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) dsum = fmaf(go[row * ld_go + h * HD + d], o[row * ld_o + h * HD + d], dsum);
        const float m = stats[((int64_t)blockIdx.x * L + ic) * 2], l = stats[((int64_t)blockIdx.x * L + ic) * 2 + 1];
        Ms[i] = i < L ? m : 0.f;
        Is[i] = i < L ? 1.0f / l : 0.f;
        Ds[i] = i < L ? dsum : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, lk = lane >> 5;
    const int cd = c < HD ? c : HD - 1;          // column of a B operand over head_dim (lanes >= HD compute unused columns)

    // ---- loop A: the wave owns query tile `it`: dQ_it = sum_jt dS K_jt, via the transposed logits tile T = K_jt Q_it^T
    for (int it = blockIdx.y * 4 + wave; it < NT; it += 4 * gridDim.y) {
        const float m_c = Ms[it * 32 + c], il_c = Is[it * 32 + c], D_c = Ds[it * 32 + c];
        f32x16a dq;
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[r] = 0.f;
        for (int jt = 0; jt < NT; ++jt) {
            f32x16a T, dPt;
#pragma unroll
            for (int r = 0; r < 16; ++r) { T[r] = 0.f; dPt[r] = 0.f; }
#pragma unroll
            for (int s = 0; s < HD / 2; ++s) {
                const float qa = Qs[(it * 32 + c) * RS + 2 * s + lk], ka = Ks[(jt * 32 + c) * RS + 2 * s + lk];
                const float ga = Gs[(it * 32 + c) * RS + 2 * s + lk], va = Vs[(jt * 32 + c) * RS + 2 * s + lk];
                T = __builtin_amdgcn_mfma_f32_32x32x2f32(ka, qa, T, 0, 0, 0);        // rows j, cols i
                dPt = __builtin_amdgcn_mfma_f32_32x32x2f32(va, ga, dPt, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const float p = j < L ? expf(T[r] - m_c) * il_c : 0.f;
                const float dS = p * (dPt[r] - D_c);
                dq = __builtin_amdgcn_mfma_f32_32x32x2f32(dS, Ks[j * RS + cd], dq, 0, 0, 0);   // k index = j (this lane's row map)
            }
        }
        if (c < HD) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = it * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (i < L) gq[((int64_t)i * B + b) * ld_gq + h * HD + c] = dq[r];
            }
        }
    }
    // ---- loop B: the wave owns key tile `jt`: dK_jt = sum_it dS^T Q_it, dV_jt = sum_it P^T dO_it, via S = Q_it K_jt^T
    for (int jt = blockIdx.y * 4 + wave; jt < NT; jt += 4 * gridDim.y) {
        const bool jv = jt * 32 + c < L;
        f32x16a dk, dv;
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[r] = 0.f; dv[r] = 0.f; }
        for (int it = 0; it < NT; ++it) {
            f32x16a S, dP;
#pragma unroll
            for (int r = 0; r < 16; ++r) { S[r] = 0.f; dP[r] = 0.f; }
#pragma unroll
            for (int s = 0; s < HD / 2; ++s) {
                const float qa = Qs[(it * 32 + c) * RS + 2 * s + lk], ka = Ks[(jt * 32 + c) * RS + 2 * s + lk];
                const float ga = Gs[(it * 32 + c) * RS + 2 * s + lk], va = Vs[(jt * 32 + c) * RS + 2 * s + lk];
                S = __builtin_amdgcn_mfma_f32_32x32x2f32(qa, ka, S, 0, 0, 0);         // rows i, cols j
                dP = __builtin_amdgcn_mfma_f32_32x32x2f32(ga, va, dP, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = it * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const float p = jv ? expf(S[r] - Ms[i]) * Is[i] : 0.f;                 // Is = 0 for padded query rows
                const float dS = p * (dP[r] - Ds[i]);
                dk = __builtin_amdgcn_mfma_f32_32x32x2f32(dS, Qs[i * RS + cd], dk, 0, 0, 0);   // k index = i
                dv = __builtin_amdgcn_mfma_f32_32x32x2f32(p, Gs[i * RS + cd], dv, 0, 0, 0);
            }
        }
        if (c < HD) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (j < L) {
                    gk[((int64_t)j * B + b) * ld_gk + h * HD + c] = dk[r];
                    gv[((int64_t)j * B + b) * ld_gv + h * HD + c] = dv[r];
                }
            }
        }
    }
}

// Matrix-core forward (head_dim 16 / 32, L <= 256).  A wave owns a 32-row query tile and works on TRANSPOSED logits tiles
// T = K_jt Q_it^T (lane = query row i, 16 keys in registers), so the row max / row sum are in-lane reductions plus one exchange
// between the two lane halves, and exp(T - m) is already the A operand of O_it += P V_jt.  Two passes over the keys (max, then
// exp / sum / PV), like the VALU kernel: same arithmetic, same saved statistics.
template <int HD>
__global__ __launch_bounds__(256) void k_attn_fwd_mfma(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, float* __restrict__ o, float* __restrict__ stats,
                                                        int L, int B, int nh, int64_t ld_q, int64_t ld_k, int64_t ld_v, int64_t ld_o,
                                                        uint32_t* obs_attn, uint32_t* obs_soft) {
    constexpr int RS = HD + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int NT = (L + 31) / 32, Lp = NT * 32;
    float* Qs = smem;
    float* Ks = Qs + (size_t)Lp * RS;
    float* Vs = Ks + (size_t)Lp * RS;
    const int b = blockIdx.x / nh, h = blockIdx.x % nh;
    fill_head<HD>(q, ld_q, Qs, L, Lp, B, b, h);
    fill_head<HD>(k, ld_k, Ks, L, Lp, B, b, h);
    fill_head<HD>(v, ld_v, Vs, L, Lp, B, b, h);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, lk = lane >> 5;
    const int cd = c < HD ? c : HD - 1;
    float smin_all = INFINITY, smax_all = -INFINITY, pmin_all = INFINITY, pmax_all = -INFINITY;
    for (int it = blockIdx.y * 4 + wave; it < NT; it += 4 * gridDim.y) {
        float m = -INFINITY, mn = INFINITY;
        for (int jt = 0; jt < NT; ++jt) {
            f32x16a T;
#pragma unroll
            for (int r = 0; r < 16; ++r) T[r] = 0.f;
#pragma unroll
            for (int s = 0; s < HD / 2; ++s)
                T = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[(jt * 32 + c) * RS + 2 * s + lk], Qs[(it * 32 + c) * RS + 2 * s + lk], T, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (j < L) { m = fmaxf(m, T[r]); mn = fminf(mn, T[r]); }
            }
        }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        mn = fminf(mn, __shfl_xor(mn, 32, 64));
        float l = 0.f;
        f32x16a acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int jt = 0; jt < NT; ++jt) {
            f32x16a T;
#pragma unroll
            for (int r = 0; r < 16; ++r) T[r] = 0.f;
#pragma unroll
            for (int s = 0; s < HD / 2; ++s)
                T = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[(jt * 32 + c) * RS + 2 * s + lk], Qs[(it * 32 + c) * RS + 2 * s + lk], T, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const float p = j < L ? expf(T[r] - m) : 0.f;
                l += p;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(p, Vs[j * RS + cd], acc, 0, 0, 0);   // rows i, cols d; k index = j
            }
        }
        l += __shfl_xor(l, 32, 64);
        const int i_own = it * 32 + c;
        if (lk == 0 && i_own < L) {
            stats[((int64_t)blockIdx.x * L + i_own) * 2] = m;
            stats[((int64_t)blockIdx.x * L + i_own) * 2 + 1] = l;
        }
        if (i_own < L) {
            smin_all = fminf(smin_all, mn); smax_all = fmaxf(smax_all, m);
            pmax_all = fmaxf(pmax_all, 1.0f / l); pmin_all = fminf(pmin_all, expf(mn - m) / l);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int il = (r & 3) + 8 * (r >> 2) + 4 * lk;             // row of this register inside the tile
            const float lr_ = __shfl(l, il, 64);                         // its row sum lives in the lane that owns that row
            const int i = it * 32 + il;
            if (c < HD && i < L) o[((int64_t)i * B + b) * ld_o + h * HD + c] = acc[r] / lr_;
        }
    }
    if (obs_attn != nullptr) {
        smin_all = wave_min(smin_all); smax_all = wave_max(smax_all);
        pmin_all = wave_min(pmin_all); pmax_all = wave_max(pmax_all);
        if (lane == 0 && smin_all <= smax_all) {
            atomicMin(obs_attn, f2ord(smin_all)); atomicMax(obs_attn + 1, f2ord(smax_all));
            atomicMin(obs_soft, f2ord(pmin_all)); atomicMax(obs_soft + 1, f2ord(pmax_all));
        }
    }
}

template <typename KernelT>
static int ensure_lds(KernelT kern, size_t bytes, const char* what) {
    if (bytes > 160 * 1024) {
        set_error("%s: sequence too long for the LDS-resident kernel (%zu bytes of LDS)", what, bytes);
        return FQSS_EINVAL;
    }
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) {
            set_error("%s: hipFuncSetAttribute(%zu): %s", what, bytes, hipGetErrorString(e));
            return FQSS_ELAUNCH;
        }
    }
    return FQSS_OK;
}

}  // namespace fqss

using namespace fqss;

extern "C" int fqss_attn_fwd(const float* q, const float* k, const float* v, float* o, float* stats, int L, int B, int nh,
                             int hd, int64_t ld_q, int64_t ld_k, int64_t ld_v, int64_t ld_o, uint32_t* obs_attn,
                             uint32_t* obs_soft, fqss_stream_t stream) {
    FQSS_REQUIRE(q && k && v && o && stats, "null tensor");
    FQSS_REQUIRE(L > 0 && B > 0 && nh > 0 && (int64_t)B * nh < (1ll << 31), "bad shape");
    FQSS_REQUIRE(ld_q >= nh * hd && ld_k >= nh * hd && ld_v >= nh * hd && ld_o >= nh * hd, "row stride below embed dim");
    FQSS_REQUIRE((obs_attn == nullptr) == (obs_soft == nullptr), "observer workspaces come in pairs");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)(B * nh)), block(256);
    static const bool use_mfma = [] { const char* e = getenv("FQSS_ATTN_MFMA"); return !(e && e[0] == '0'); }();
    // head_dim 16 stays on the VALU kernel in the forward: half of every 32-wide P V tile would be padding (measured 133 vs 113 us
    // at the DPTNet shapes); the backward wins on the matrix cores for both widths
    if (use_mfma && hd == 32 && L <= 256) {
        const int Lp = (L + 31) / 32 * 32;
        const size_t ldsm = (size_t)3 * Lp * (hd + 1) * sizeof(float);
        grid.y = (unsigned)cdiv(Lp / 32, 4);      // one 32-row tile per wave: (b, h) pairs alone leave CUs idle or wrap around (272 on 256)
        if (hd == 16) {
            int rc = ensure_lds(k_attn_fwd_mfma<16>, ldsm, "fqss_attn_fwd");
            if (rc != FQSS_OK) return rc;
            hipLaunchKernelGGL((k_attn_fwd_mfma<16>), grid, block, ldsm, s, q, k, v, o, stats, L, B, nh, ld_q, ld_k, ld_v, ld_o, obs_attn, obs_soft);
        } else {
            int rc = ensure_lds(k_attn_fwd_mfma<32>, ldsm, "fqss_attn_fwd");
            if (rc != FQSS_OK) return rc;
            hipLaunchKernelGGL((k_attn_fwd_mfma<32>), grid, block, ldsm, s, q, k, v, o, stats, L, B, nh, ld_q, ld_k, ld_v, ld_o, obs_attn, obs_soft);
        }
        return launch_status("fqss_attn_fwd");
    }
    const size_t lds = (size_t)2 * L * hd * sizeof(float);
#define FQSS_AF(HD_)                                                                                                        \
    {                                                                                                                       \
        int rc = ensure_lds(k_attn_fwd<HD_>, lds, "fqss_attn_fwd");                                                          \
        if (rc != FQSS_OK) return rc;                                                                                       \
        hipLaunchKernelGGL((k_attn_fwd<HD_>), grid, block, lds, s, q, k, v, o, stats, L, B, nh, ld_q, ld_k, ld_v, ld_o,      \
                           obs_attn, obs_soft);                                                                             \
    }
    switch (hd) {
        case 2: FQSS_AF(2) break;
        case 4: FQSS_AF(4) break;
        case 8: FQSS_AF(8) break;
        case 16: FQSS_AF(16) break;
        case 32: FQSS_AF(32) break;
        default: set_error("fqss_attn_fwd: head_dim %d not built (2, 4, 8, 16, 32)", hd); return FQSS_EINVAL;
    }
#undef FQSS_AF
    return launch_status("fqss_attn_fwd");
}

extern "C" int fqss_attn_bwd(const float* q, const float* k, const float* v, const float* o, const float* go,
                             const float* stats, float* gq, float* gk, float* gv, int L, int B, int nh, int hd, int64_t ld_q,
                             int64_t ld_k, int64_t ld_v, int64_t ld_o, int64_t ld_go, int64_t ld_gq, int64_t ld_gk,
                             int64_t ld_gv, fqss_stream_t stream) {
    FQSS_REQUIRE(q && k && v && o && go && stats && gq && gk && gv, "null tensor");
    FQSS_REQUIRE(L > 0 && B > 0 && nh > 0 && (int64_t)B * nh < (1ll << 31), "bad shape");
    const int E = nh * hd;
    FQSS_REQUIRE(ld_q >= E && ld_k >= E && ld_v >= E && ld_o >= E && ld_go >= E && ld_gq >= E && ld_gk >= E && ld_gv >= E,
                 "row stride below embed dim");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)(B * nh)), block(256);
    static const bool use_mfma = [] { const char* e = getenv("FQSS_ATTN_MFMA"); return !(e && e[0] == '0'); }();
    if (use_mfma && (hd == 16 || hd == 32) && L <= 256) {
        const int Lp = (L + 31) / 32 * 32;
        const size_t ldsm = ((size_t)4 * Lp * (hd + 1) + 3 * (size_t)Lp) * sizeof(float);
        grid.y = (unsigned)cdiv(Lp / 32, 4);
        if (hd == 16) {
            int rc = ensure_lds(k_attn_bwd_mfma<16>, ldsm, "fqss_attn_bwd");
            if (rc != FQSS_OK) return rc;
            hipLaunchKernelGGL((k_attn_bwd_mfma<16>), grid, block, ldsm, s, q, k, v, o, go, stats, gq, gk, gv, L, B, nh, ld_q, ld_k, ld_v, ld_o,
                               ld_go, ld_gq, ld_gk, ld_gv);
        } else {
            int rc = ensure_lds(k_attn_bwd_mfma<32>, ldsm, "fqss_attn_bwd");
            if (rc != FQSS_OK) return rc;
            hipLaunchKernelGGL((k_attn_bwd_mfma<32>), grid, block, ldsm, s, q, k, v, o, go, stats, gq, gk, gv, L, B, nh, ld_q, ld_k, ld_v, ld_o,
                               ld_go, ld_gq, ld_gk, ld_gv);
        }
        return launch_status("fqss_attn_bwd");
    }
    const size_t lds = ((size_t)4 * L * hd + 3 * (size_t)L) * sizeof(float);
#define FQSS_AB(HD_)                                                                                                        \
    {                                                                                                                       \
        int rc = ensure_lds(k_attn_bwd<HD_>, lds, "fqss_attn_bwd");                                                          \
        if (rc != FQSS_OK) return rc;                                                                                       \
        hipLaunchKernelGGL((k_attn_bwd<HD_>), grid, block, lds, s, q, k, v, o, go, stats, gq, gk, gv, L, B, nh, ld_q, ld_k,  \
                           ld_v, ld_o, ld_go, ld_gq, ld_gk, ld_gv);                                                         \
    }
    switch (hd) {
        case 2: FQSS_AB(2) break;
        case 4: FQSS_AB(4) break;
        case 8: FQSS_AB(8) break;
        case 16: FQSS_AB(16) break;
        case 32: FQSS_AB(32) break;
        default: set_error("fqss_attn_bwd: head_dim %d not built (2, 4, 8, 16, 32)", hd); return FQSS_EINVAL;
    }
#undef FQSS_AB
    return launch_status("fqss_attn_bwd");
}
