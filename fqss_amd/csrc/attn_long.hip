// attn_long.hip -- softmax(q k^T) v for the LONG sequences and the CROSS attention of the HTDemucs transformer (SURVEY.md §8 row
// a15; htdemucsq.py:138-329: ~3.4 k spectrogram tokens x ~1.7 k waveform tokens, head_dim 48 or 64), where K / V of a head no longer
// fit in LDS (attn.hip keeps them resident for the 250-step sequences of the dual-path models) and Lq != Lk.
//
// Streaming ("flash") form.  Forward: a workgroup owns 256 query rows of one (batch, head), a thread owns ONE query row -- q, the
// running output, the running max m and sum l stay in its registers -- while K / V stream through LDS in tiles of TK keys; all lanes
// read the same LDS word (a broadcast, conflict-free) and nothing is ever reduced across lanes.  The softmax is the online one
// (rescale on a new maximum); the saved statistics (m, l) let the backward recompute p_ij = exp(s_ij - m_i) / l_i.
// Backward: two owner-computes passes, no atomics, deterministic: thread <-> query row for dq (K / V stream), thread <-> key row for
// dk / dv (q, dO and the row statistics stream); D_i = dO_i . o_i is produced by the first pass for the second.
// fp32 FMA on the vector ALU: a first correct path (the matrix-core form of attn.hip's short-sequence kernels is the next step).
//
// Rows are addressed as x[l * sl + b * sb + h * HD + d] (element strides), which serves the sequence-first [L, B, E] tensors of
// nn.MultiheadAttention as well as the batch-first [B, T, E] tensors of the HTDemucs transformer without a transposing copy.
// As in attn.hip, obs_attn / obs_soft (optional, observer phase) receive the min / max of the logits and of the probabilities: the
// reference runs two quantizers on them and discards the results (qat_layers.py:907-909).
#include <stdlib.h>

#include "fqss_dev.h"

namespace fqss {

struct RowView {
    int64_t sl, sb;   // element strides of the sequence and the batch index
};
struct AttnGeom {
    int Lq, Lk, B, nh;
    RowView q, k, v, o, go, gq, gk, gv;
};

constexpr int kTK = 32;   // keys (or queries, in the dk/dv pass) per LDS tile
constexpr int kAcc = 4;   // partial sums per dot product in the backward passes

template <int N>
__device__ __forceinline__ float tree_sum(const float (&v)[N]) {
    float t[N];
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = v[i];
#pragma unroll
    for (int w = N / 2; w > 0; w /= 2)
#pragma unroll
        for (int i = 0; i < w; ++i) t[i] += t[i + w];
    return t[0];
}

template <int HD>
__global__ __launch_bounds__(256) void k_attn_long_fwd(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                        float* __restrict__ o, float* __restrict__ stats, const AttnGeom g,
                                                        uint32_t* obs_attn, uint32_t* obs_soft) {
    __shared__ __attribute__((aligned(16))) float Ks[kTK * HD];
    __shared__ __attribute__((aligned(16))) float Vs[kTK * HD];
    const int bh = blockIdx.y, b = bh / g.nh, h = bh % g.nh;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < g.Lq;
    float qr[HD], acc[HD];
    const float* qp = q + (int64_t)(live ? i : 0) * g.q.sl + (int64_t)b * g.q.sb + h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) { qr[d] = qp[d]; acc[d] = 0.f; }
    float m = -INFINITY, l = 0.f, smin = INFINITY;
    for (int j0 = 0; j0 < g.Lk; j0 += kTK) {
        const int nj = min(kTK, g.Lk - j0);
        __syncthreads();
        for (int e = threadIdx.x; e < nj * HD; e += 256) {
            const int j = e / HD, d = e % HD;
            Ks[e] = k[(int64_t)(j0 + j) * g.k.sl + (int64_t)b * g.k.sb + h * HD + d];
            Vs[e] = v[(int64_t)(j0 + j) * g.v.sl + (int64_t)b * g.v.sb + h * HD + d];
        }
        // padding keys of the last tile: their probability is exp(-inf) = 0, but 0 * (stale LDS contents) must stay 0
        for (int e = nj * HD + threadIdx.x; e < kTK * HD; e += 256) { Ks[e] = 0.f; Vs[e] = 0.f; }
        __syncthreads();
        float s[kTK];
        float tmax = -INFINITY;
#pragma unroll
        for (int j = 0; j < kTK; ++j) {
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) a = fmaf(qr[d], Ks[j * HD + d], a);
            s[j] = j < nj ? a : -INFINITY;
            tmax = fmaxf(tmax, s[j]);
            smin = fminf(smin, j < nj ? a : INFINITY);
        }
        if (tmax > m) {
            const float r = expf(m - tmax);      // 0 on the first tile (m = -inf)
            l *= r;
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[d] *= r;
            m = tmax;
        }
#pragma unroll
        for (int j = 0; j < kTK; ++j) {
            const float p = expf(s[j] - m);      // exp(-inf) = 0 for the padding keys of the last tile
            l += p;
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[d] = fmaf(p, Vs[j * HD + d], acc[d]);
        }
    }
    float smin_all = INFINITY, smax_all = -INFINITY, pmin_all = INFINITY, pmax_all = -INFINITY;
    if (live) {
        float* op = o + (int64_t)i * g.o.sl + (int64_t)b * g.o.sb + h * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) op[d] = acc[d] / l;
        stats[((int64_t)bh * g.Lq + i) * 2] = m;
        stats[((int64_t)bh * g.Lq + i) * 2 + 1] = l;
        smin_all = smin; smax_all = m; pmax_all = 1.0f / l; pmin_all = expf(smin - m) / l;
    }
    if (obs_attn != nullptr) {     // observer phase only (uniform branch)
        smin_all = wave_min(smin_all); smax_all = wave_max(smax_all);
        pmin_all = wave_min(pmin_all); pmax_all = wave_max(pmax_all);
        if ((threadIdx.x & 63) == 0 && smin_all <= smax_all) {
            atomicMin(obs_attn, f2ord(smin_all)); atomicMax(obs_attn + 1, f2ord(smax_all));
            atomicMin(obs_soft, f2ord(pmin_all)); atomicMax(obs_soft + 1, f2ord(pmax_all));
        }
    }
}

// dq_i = sum_j p_ij (dO_i . v_j - D_i) k_j ;  also writes D_i = dO_i . o_i for the dk / dv pass
template <int HD>
__global__ __launch_bounds__(256) void k_attn_long_bwd_q(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                          const float* __restrict__ o, const float* __restrict__ go,
                                                          const float* __restrict__ stats, float* __restrict__ gq, float* __restrict__ dsum,
                                                          const AttnGeom g) {
    __shared__ __attribute__((aligned(16))) float Ks[kTK * HD];
    __shared__ __attribute__((aligned(16))) float Vs[kTK * HD];
    const int bh = blockIdx.y, b = bh / g.nh, h = bh % g.nh;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < g.Lq;
    const int ii = live ? i : 0;
    float qr[HD], gor[HD], acc[HD];
    const float* qp = q + (int64_t)ii * g.q.sl + (int64_t)b * g.q.sb + h * HD;
    const float* gop = go + (int64_t)ii * g.go.sl + (int64_t)b * g.go.sb + h * HD;
    const float* op = o + (int64_t)ii * g.o.sl + (int64_t)b * g.o.sb + h * HD;
    float D = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) { qr[d] = qp[d]; gor[d] = gop[d]; acc[d] = 0.f; D = fmaf(gor[d], op[d], D); }
    const float m = stats[((int64_t)bh * g.Lq + ii) * 2], rl = 1.0f / stats[((int64_t)bh * g.Lq + ii) * 2 + 1];
    for (int j0 = 0; j0 < g.Lk; j0 += kTK) {
        const int nj = min(kTK, g.Lk - j0);
        __syncthreads();
        for (int e = threadIdx.x; e < nj * HD; e += 256) {
            const int j = e / HD, d = e % HD;
            Ks[e] = k[(int64_t)(j0 + j) * g.k.sl + (int64_t)b * g.k.sb + h * HD + d];
            Vs[e] = v[(int64_t)(j0 + j) * g.v.sl + (int64_t)b * g.v.sb + h * HD + d];
        }
        __syncthreads();
        // each dot product is split over 2 * kAcc partial sums: 16 independent FMA chains per lane (one wave per SIMD at this register
        // budget: the dependent 64-long chains of the plain form left the FMA pipe idle 3 cycles in 4)
        for (int j = 0; j < nj; ++j) {
            const float* k0 = Ks + j * HD;
            const float* v0 = Vs + j * HD;
            float sa[2 * kAcc], pa[2 * kAcc];
#pragma unroll
            for (int a = 0; a < 2 * kAcc; ++a) sa[a] = pa[a] = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                sa[d % (2 * kAcc)] = fmaf(qr[d], k0[d], sa[d % (2 * kAcc)]);
                pa[d % (2 * kAcc)] = fmaf(gor[d], v0[d], pa[d % (2 * kAcc)]);
            }
            const float ds0 = (expf(tree_sum<2 * kAcc>(sa) - m) * rl) * (tree_sum<2 * kAcc>(pa) - D);
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[d] = fmaf(ds0, k0[d], acc[d]);
        }
    }
    if (live) {
        float* gp = gq + (int64_t)i * g.gq.sl + (int64_t)b * g.gq.sb + h * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) gp[d] = acc[d];
        dsum[(int64_t)bh * g.Lq + i] = D;
    }
}

// dv_j = sum_i p_ij dO_i ;  dk_j = sum_i p_ij (dO_i . v_j - D_i) q_i
template <int HD>
__global__ __launch_bounds__(256) void k_attn_long_bwd_kv(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                           const float* __restrict__ go, const float* __restrict__ stats,
                                                           const float* __restrict__ dsum, float* __restrict__ gk, float* __restrict__ gv,
                                                           const AttnGeom g) {
    __shared__ __attribute__((aligned(16))) float Qs[kTK * HD];
    __shared__ __attribute__((aligned(16))) float Gs[kTK * HD];
    __shared__ float Ms[kTK], Rs[kTK], Ds[kTK];
    const int bh = blockIdx.y, b = bh / g.nh, h = bh % g.nh;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const bool live = j < g.Lk;
    const int jj = live ? j : 0;
    float kr[HD], vr[HD], ak[HD], av[HD];
    const float* kp = k + (int64_t)jj * g.k.sl + (int64_t)b * g.k.sb + h * HD;
    const float* vp = v + (int64_t)jj * g.v.sl + (int64_t)b * g.v.sb + h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) { kr[d] = kp[d]; vr[d] = vp[d]; ak[d] = 0.f; av[d] = 0.f; }
    for (int i0 = 0; i0 < g.Lq; i0 += kTK) {
        const int ni = min(kTK, g.Lq - i0);
        __syncthreads();
        for (int e = threadIdx.x; e < ni * HD; e += 256) {
            const int i = e / HD, d = e % HD;
            Qs[e] = q[(int64_t)(i0 + i) * g.q.sl + (int64_t)b * g.q.sb + h * HD + d];
            Gs[e] = go[(int64_t)(i0 + i) * g.go.sl + (int64_t)b * g.go.sb + h * HD + d];
        }
        if (threadIdx.x < ni) {
            Ms[threadIdx.x] = stats[((int64_t)bh * g.Lq + i0 + threadIdx.x) * 2];
            Rs[threadIdx.x] = 1.0f / stats[((int64_t)bh * g.Lq + i0 + threadIdx.x) * 2 + 1];
            Ds[threadIdx.x] = dsum[(int64_t)bh * g.Lq + i0 + threadIdx.x];
        }
        __syncthreads();
        for (int i = 0; i < ni; ++i) {            // one query row per iteration (the four 64-wide register arrays leave no room for two)
            const float* q0 = Qs + i * HD;
            const float* g0 = Gs + i * HD;
            float sa[2 * kAcc], pa[2 * kAcc];
#pragma unroll
            for (int a = 0; a < 2 * kAcc; ++a) sa[a] = pa[a] = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                sa[d % (2 * kAcc)] = fmaf(q0[d], kr[d], sa[d % (2 * kAcc)]);
                pa[d % (2 * kAcc)] = fmaf(g0[d], vr[d], pa[d % (2 * kAcc)]);
            }
            const float p0 = expf(tree_sum<2 * kAcc>(sa) - Ms[i]) * Rs[i];
            const float ds0 = p0 * (tree_sum<2 * kAcc>(pa) - Ds[i]);
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                av[d] = fmaf(p0, g0[d], av[d]);
                ak[d] = fmaf(ds0, q0[d], ak[d]);
            }
        }
    }
    if (live) {
        float* gkp = gk + (int64_t)j * g.gk.sl + (int64_t)b * g.gk.sb + h * HD;
        float* gvp = gv + (int64_t)j * g.gv.sl + (int64_t)b * g.gv.sb + h * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) { gkp[d] = ak[d]; gvp[d] = av[d]; }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Matrix-core forms (head_dim a multiple of 32: the HTDemucs transformer with bottom_channels 512 has 64).  fp32 MFMA 32x32x2: the
// same fmaf arithmetic as the vector kernels at the matrix pipe's rate.  A wave owns a 32-row query tile and keeps its Q fragment in
// registers; a workgroup (4 waves = 128 queries) streams 32-key K / V tiles through LDS.  Scores are computed TRANSPOSED,
// T = K_tile Q^T: lane (c, half) holds query c and 16 of the 32 keys, so the row max / sum of the online softmax are in-lane
// reductions plus one exchange between the lane halves, and exp(T - m) is already the A operand of O += P V_tile.
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int tile_row(int r, int lk) { return (r & 3) + 8 * (r >> 2) + 4 * lk; }     // row of accumulator register r

// a 32-row tile of one head moves global -> registers -> LDS in two steps, so the loads of tile t+1 are in flight while tile t is
// being multiplied (one wave per SIMD or two: the global latency of ~1 us would otherwise sit between every pair of barriers)
template <int HD>
struct TileRegs {
    float v[32 * HD / 256];
};
template <int HD>
__device__ __forceinline__ void tile_fetch(const float* __restrict__ x, const RowView rv, int b, int h, int r0, int nrows, TileRegs<HD>& t) {
#pragma unroll
    for (int u = 0; u < 32 * HD / 256; ++u) {
        const int e = threadIdx.x + 256 * u, j = e / HD, d = e % HD;
        t.v[u] = j < nrows ? x[(int64_t)(r0 + j) * rv.sl + (int64_t)b * rv.sb + h * HD + d] : 0.f;
    }
}
template <int HD>
__device__ __forceinline__ void tile_store(const TileRegs<HD>& t, float* dst) {
    constexpr int RS = HD + 1;
#pragma unroll
    for (int u = 0; u < 32 * HD / 256; ++u) {
        const int e = threadIdx.x + 256 * u;
        dst[(e / HD) * RS + e % HD] = t.v[u];
    }
}

template <int HD>
__global__ __launch_bounds__(256) void k_attn_long_fwd_mfma(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                             float* __restrict__ o, float* __restrict__ stats, const AttnGeom g,
                                                             uint32_t* obs_attn, uint32_t* obs_soft) {
    constexpr int RS = HD + 1, NS = HD / 2, ND = HD / 32;
    __shared__ float Ks[32 * RS];
    __shared__ float Vs[32 * RS];
    const int bh = blockIdx.y, b = bh / g.nh, h = bh % g.nh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, lk = lane >> 5;
    const int it = blockIdx.x * 4 + wave;
    const int i_own = it * 32 + c;
    const bool live = i_own < g.Lq;
    const float* qp = q + (int64_t)(live ? i_own : g.Lq - 1) * g.q.sl + (int64_t)b * g.q.sb + h * HD;
    float qf[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) qf[s] = qp[2 * s + lk];
    f32x16 acc[ND];
#pragma unroll
    for (int nd = 0; nd < ND; ++nd)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nd][r] = 0.f;
    float m = -INFINITY, l = 0.f, smin = INFINITY;
    TileRegs<HD> kt, vt;
    tile_fetch<HD>(k, g.k, b, h, 0, min(32, g.Lk), kt);
    tile_fetch<HD>(v, g.v, b, h, 0, min(32, g.Lk), vt);
    for (int j0 = 0; j0 < g.Lk; j0 += 32) {
        const int nj = min(32, g.Lk - j0);
        __syncthreads();
        tile_store<HD>(kt, Ks);
        tile_store<HD>(vt, Vs);
        __syncthreads();
        if (j0 + 32 < g.Lk) {
            tile_fetch<HD>(k, g.k, b, h, j0 + 32, min(32, g.Lk - j0 - 32), kt);
            tile_fetch<HD>(v, g.v, b, h, j0 + 32, min(32, g.Lk - j0 - 32), vt);
        }
        f32x16 T0, T1;
#pragma unroll
        for (int r = 0; r < 16; ++r) T0[r] = T1[r] = 0.f;
        // the LDS operands of a whole phase are fetched into distinct registers first: left to itself the compiler recycles one
        // register pair and puts a full LDS round trip (s_waitcnt lgkmcnt(0)) in front of every pair of MFMAs
        float ka[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) ka[s] = Ks[c * RS + 2 * s + lk];
        __builtin_amdgcn_sched_barrier(0);       // (the scheduler otherwise sinks the reads back next to their MFMAs)
#pragma unroll
        for (int s = 0; s < NS; s += 2) {        // two accumulators: consecutive MFMAs do not wait on each other
            T0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[s], qf[s], T0, 0, 0, 0);
            T1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[s + 1], qf[s + 1], T1, 0, 0, 0);
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float t = T0[r] + T1[r];
            const bool ok = tile_row(r, lk) < nj;
            T0[r] = ok ? t : -INFINITY;
            tmax = fmaxf(tmax, T0[r]);
            smin = fminf(smin, ok ? t : INFINITY);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m, tmax);
        const float sc = expf(m - m_new);            // 0 on the first tile
        if (__any(sc != 1.0f)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float f = __shfl(sc, tile_row(r, lk), 64);     // the factor of the query that owns accumulator row r
#pragma unroll
                for (int nd = 0; nd < ND; ++nd) acc[nd][r] *= f;
            }
        }
        l *= sc;
        m = m_new;
        float vb[16][ND], pr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int nd = 0; nd < ND; ++nd) vb[r][nd] = Vs[tile_row(r, lk) * RS + c + 32 * nd];
            pr[r] = expf(T0[r] - m);
            l += pr[r];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int nd = 0; nd < ND; ++nd) acc[nd] = __builtin_amdgcn_mfma_f32_32x32x2f32(pr[r], vb[r][nd], acc[nd], 0, 0, 0);
    }
    l += __shfl_xor(l, 32, 64);
    smin = fminf(smin, __shfl_xor(smin, 32, 64));
    if (lk == 0 && live) {
        stats[((int64_t)bh * g.Lq + i_own) * 2] = m;
        stats[((int64_t)bh * g.Lq + i_own) * 2 + 1] = l;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int il = tile_row(r, lk);
        const float lr = __shfl(l, il, 64);
        const int i = it * 32 + il;
        if (i < g.Lq) {
            float* op = o + (int64_t)i * g.o.sl + (int64_t)b * g.o.sb + h * HD + c;
#pragma unroll
            for (int nd = 0; nd < ND; ++nd) op[32 * nd] = acc[nd][r] / lr;
        }
    }
    if (obs_attn != nullptr) {
        float smin_all = live ? smin : INFINITY, smax_all = live ? m : -INFINITY;
        float pmax_all = live ? 1.0f / l : -INFINITY, pmin_all = live ? expf(smin - m) / l : INFINITY;
        smin_all = wave_min(smin_all); smax_all = wave_max(smax_all);
        pmin_all = wave_min(pmin_all); pmax_all = wave_max(pmax_all);
        if (lane == 0 && smin_all <= smax_all) {
            atomicMin(obs_attn, f2ord(smin_all)); atomicMax(obs_attn + 1, f2ord(smax_all));
            atomicMin(obs_soft, f2ord(pmin_all)); atomicMax(obs_soft + 1, f2ord(pmax_all));
        }
    }
}

// dq on the matrix cores: per 32-key tile  T = K Q^T,  U = V dO^T  (both transposed: lane = query, registers = keys, so the softmax
// statistics m, 1/l and D = dO . o of the lane's query apply in-lane),  dS = P (U - D),  dq += dS K  (dS is already the A operand).
template <int HD>
__global__ __launch_bounds__(256) void k_attn_long_bwd_q_mfma(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                               const float* __restrict__ o, const float* __restrict__ go,
                                                               const float* __restrict__ stats, float* __restrict__ gq, float* __restrict__ dsum,
                                                               const AttnGeom g) {
    constexpr int RS = HD + 1, NS = HD / 2, ND = HD / 32;
    __shared__ float Ks[32 * RS];
    __shared__ float Vs[32 * RS];
    const int bh = blockIdx.y, b = bh / g.nh, h = bh % g.nh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, lk = lane >> 5;
    const int it = blockIdx.x * 4 + wave;
    const int i_own = it * 32 + c;
    const bool live = i_own < g.Lq;
    const int ic = live ? i_own : g.Lq - 1;
    const float* qp = q + (int64_t)ic * g.q.sl + (int64_t)b * g.q.sb + h * HD;
    const float* gp = go + (int64_t)ic * g.go.sl + (int64_t)b * g.go.sb + h * HD;
    const float* op = o + (int64_t)ic * g.o.sl + (int64_t)b * g.o.sb + h * HD;
    float qf[NS], gf[NS], D = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        qf[s] = qp[2 * s + lk];
        gf[s] = gp[2 * s + lk];
        D = fmaf(gf[s], op[2 * s + lk], D);
    }
    D += __shfl_xor(D, 32, 64);
    const float m = stats[((int64_t)bh * g.Lq + ic) * 2], rl = 1.0f / stats[((int64_t)bh * g.Lq + ic) * 2 + 1];
    f32x16 acc[ND];
#pragma unroll
    for (int nd = 0; nd < ND; ++nd)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nd][r] = 0.f;
    TileRegs<HD> kt, vt;
    tile_fetch<HD>(k, g.k, b, h, 0, min(32, g.Lk), kt);
    tile_fetch<HD>(v, g.v, b, h, 0, min(32, g.Lk), vt);
    for (int j0 = 0; j0 < g.Lk; j0 += 32) {
        const int nj = min(32, g.Lk - j0);
        __syncthreads();
        tile_store<HD>(kt, Ks);
        tile_store<HD>(vt, Vs);
        __syncthreads();
        if (j0 + 32 < g.Lk) {
            tile_fetch<HD>(k, g.k, b, h, j0 + 32, min(32, g.Lk - j0 - 32), kt);
            tile_fetch<HD>(v, g.v, b, h, j0 + 32, min(32, g.Lk - j0 - 32), vt);
        }
        f32x16 T, U;
#pragma unroll
        for (int r = 0; r < 16; ++r) T[r] = U[r] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            T = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[c * RS + 2 * s + lk], qf[s], T, 0, 0, 0);
            U = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[c * RS + 2 * s + lk], gf[s], U, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = tile_row(r, lk) < nj ? expf(T[r] - m) * rl : 0.f;
            const float ds = p * (U[r] - D);
            const float* kr = Ks + tile_row(r, lk) * RS + c;
#pragma unroll
            for (int nd = 0; nd < ND; ++nd) acc[nd] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds, kr[32 * nd], acc[nd], 0, 0, 0);
        }
    }
    if (lk == 0 && live) dsum[(int64_t)bh * g.Lq + i_own] = D;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = it * 32 + tile_row(r, lk);
        if (i < g.Lq) {
            float* gp2 = gq + (int64_t)i * g.gq.sl + (int64_t)b * g.gq.sb + h * HD + c;
#pragma unroll
            for (int nd = 0; nd < ND; ++nd) gp2[32 * nd] = acc[nd][r];
        }
    }
}

// dk / dv on the matrix cores: a wave owns 32 keys (K, V fragments in registers), 32-query tiles of q, dO and the row statistics
// stream through LDS:  T' = Q K^T,  U' = dO V^T  (lane = key, registers = queries),  P = exp(T' - m_i) / l_i,  dS = P (U' - D_i),
// dv += P^T dO,  dk += dS^T Q  (P / dS in this layout ARE the A operands of the two transposed products).
template <int HD>
__global__ __launch_bounds__(256) void k_attn_long_bwd_kv_mfma(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                                const float* __restrict__ go, const float* __restrict__ stats,
                                                                const float* __restrict__ dsum, float* __restrict__ gk, float* __restrict__ gv,
                                                                const AttnGeom g) {
    constexpr int RS = HD + 1, NS = HD / 2, ND = HD / 32;
    __shared__ float Qs[32 * RS];
    __shared__ float Gs[32 * RS];
    __shared__ float Ms[32], Rs[32], Ds[32];
    const int bh = blockIdx.y, b = bh / g.nh, h = bh % g.nh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, lk = lane >> 5;
    const int jt = blockIdx.x * 4 + wave;
    const int j_own = jt * 32 + c;
    const int jc = j_own < g.Lk ? j_own : g.Lk - 1;
    const float* kp = k + (int64_t)jc * g.k.sl + (int64_t)b * g.k.sb + h * HD;
    const float* vp = v + (int64_t)jc * g.v.sl + (int64_t)b * g.v.sb + h * HD;
    float kf[NS], vf[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) { kf[s] = kp[2 * s + lk]; vf[s] = vp[2 * s + lk]; }
    f32x16 ak[ND], av[ND];
#pragma unroll
    for (int nd = 0; nd < ND; ++nd)
#pragma unroll
        for (int r = 0; r < 16; ++r) ak[nd][r] = av[nd][r] = 0.f;
    TileRegs<HD> qt, gt;
    tile_fetch<HD>(q, g.q, b, h, 0, min(32, g.Lq), qt);
    tile_fetch<HD>(go, g.go, b, h, 0, min(32, g.Lq), gt);
    for (int i0 = 0; i0 < g.Lq; i0 += 32) {
        const int ni = min(32, g.Lq - i0);
        __syncthreads();
        tile_store<HD>(qt, Qs);
        tile_store<HD>(gt, Gs);
        if (threadIdx.x < 32) {
            const bool ok = threadIdx.x < ni;
            const int64_t si = (int64_t)bh * g.Lq + i0 + (ok ? threadIdx.x : 0);
            Ms[threadIdx.x] = ok ? stats[si * 2] : 0.f;
            Rs[threadIdx.x] = ok ? 1.0f / stats[si * 2 + 1] : 0.f;      // padding queries: probability 0
            Ds[threadIdx.x] = ok ? dsum[si] : 0.f;
        }
        __syncthreads();
        if (i0 + 32 < g.Lq) {
            tile_fetch<HD>(q, g.q, b, h, i0 + 32, min(32, g.Lq - i0 - 32), qt);
            tile_fetch<HD>(go, g.go, b, h, i0 + 32, min(32, g.Lq - i0 - 32), gt);
        }
        f32x16 T, U;
#pragma unroll
        for (int r = 0; r < 16; ++r) T[r] = U[r] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            T = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[c * RS + 2 * s + lk], kf[s], T, 0, 0, 0);
            U = __builtin_amdgcn_mfma_f32_32x32x2f32(Gs[c * RS + 2 * s + lk], vf[s], U, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = tile_row(r, lk);
            const float p = expf(T[r] - Ms[row]) * Rs[row];
            const float ds = p * (U[r] - Ds[row]);
            const float* gr = Gs + row * RS + c;
            const float* qr = Qs + row * RS + c;
#pragma unroll
            for (int nd = 0; nd < ND; ++nd) {
                av[nd] = __builtin_amdgcn_mfma_f32_32x32x2f32(p, gr[32 * nd], av[nd], 0, 0, 0);
                ak[nd] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds, qr[32 * nd], ak[nd], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int j = jt * 32 + tile_row(r, lk);
        if (j < g.Lk) {
            float* gkp = gk + (int64_t)j * g.gk.sl + (int64_t)b * g.gk.sb + h * HD + c;
            float* gvp = gv + (int64_t)j * g.gv.sl + (int64_t)b * g.gv.sb + h * HD + c;
#pragma unroll
            for (int nd = 0; nd < ND; ++nd) { gkp[32 * nd] = ak[nd][r]; gvp[32 * nd] = av[nd][r]; }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------
// The same three passes on the bf16 matrix cores (16x the fp32 MFMA rate): every fp32 operand -- q, k, v, dO, and the probabilities
// and dS computed in between -- is split EXACTLY into three bf16 pieces (8 mantissa bits each) and the six partial products above
// 2^-24 of each product are accumulated in fp32, smallest first: the arithmetic of csrc/gemm_x3.hip, fp32-grade results at 6/16 of
// the fp32-MFMA instruction time.  v_mfma_f32_32x32x16_bf16 takes 8 consecutive k per lane:
//   * tiles live in LDS as three bf16 planes [32 rows][HD + 8]; the A operand of T = K Q^T is one 16-B read per plane and k-step;
//     lane c supplies row kappa(c) of the tile, a fixed permutation chosen so that accumulator register r of lane half lk holds the
//     logit of tile row krow(r, lk) = 16 (r >> 3) + 8 lk + (r & 7): registers 8s .. 8s+7 of a lane are then the 8 consecutive k of
//     k-step s, i.e. exp(T - m) (and dS) in registers ARE the A operands of the second product, as in the fp32 form;
//   * the lane's own row (q; k / v in the dk/dv pass; dO) is the B operand, split once into registers;
//   * the B operand of the second product (V, K, dO, Q as [row = k][d = n]) comes out of the same planes through the transposing
//     LDS read ds_read_tr16_b64 (the idiom of csrc/qgemm.hip).
// Needs 16-B aligned rows (float4 global loads); the fp32-MFMA kernels above remain for everything else (FQSS_ATTN_MFMA=f32 forces them).
// ------------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// Workgroups are handed to the 8 XCDs round-robin by linear id; every XCD has its own 4 MB L2.  With blockIdx.x = query block the 27
// blocks that stream the SAME K / V of a head land on 8 different L2s, and 32 heads x 1.8 MB do not fit any of them: PMC showed 4.7x the
// algorithmic bytes fetched (profiles/r03_pmc_traffic_cfg345.txt).  Remapped: linear id l -> XCD l % 8, slot l / 8; the heads h with
// h % 8 == XCD are walked block by block inside that XCD.  (Grids whose head count is not a multiple of 8 keep the plain map.)
__device__ __forceinline__ void al_block(int& bx, int& by) {
    const int nx = gridDim.x, ny = gridDim.y;
    if ((ny & 7) == 0) {
        const int lin = blockIdx.y * nx + blockIdx.x, xcd = lin & 7, slot = lin >> 3;
        by = (slot / nx) * 8 + xcd;
        bx = slot - (slot / nx) * nx;
    } else {
        bx = blockIdx.x;
        by = blockIdx.y;
    }
}

__device__ __forceinline__ int krow(int r, int lk) { return 16 * (r >> 3) + 8 * lk + (r & 7); }
__device__ __forceinline__ int kappa(int c) { return (c & 16) + 8 * ((c >> 2) & 1) + (c & 3) + 4 * ((c >> 3) & 1); }
__device__ __forceinline__ float al_trunc(float f) { return __uint_as_float(__float_as_uint(f) & 0xFFFF0000u); }

// exp(x) for x <= 0 (a logit minus the row maximum) as v_exp_f32(x log2 e): the rounding of the product is |x| 2^-24 relative in the
// result, far below the fp32 noise of the dot products; -inf -> 0
__device__ __forceinline__ float al_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

struct Frag3 {
    bf16x8 p[3];      // head, middle, tail
};
// 8 fp32 values -> three bf16x8 (v_perm picks the high halves of two fp32 words = truncation; the remainders are exact)
__device__ __forceinline__ Frag3 split8(const float (&x)[8]) {
    union { bf16x8 v; unsigned int w[4]; } h, m, l;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a = x[2 * q], b = x[2 * q + 1];
        const float ar = a - al_trunc(a), br = b - al_trunc(b);
        const float al = ar - al_trunc(ar), bl = br - al_trunc(br);
        h.w[q] = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
        m.w[q] = __builtin_amdgcn_perm(__float_as_uint(br), __float_as_uint(ar), 0x07060302u);
        l.w[q] = __builtin_amdgcn_perm(__float_as_uint(bl), __float_as_uint(al), 0x07060302u);
    }
    Frag3 f;
    f.p[0] = h.v; f.p[1] = m.v; f.p[2] = l.v;
    return f;
}

// the six products, smallest first: (A piece, B piece) = l.h, h.l, m.m, m.h, h.m, h.h
#define FQSS_X3_PRODUCTS(ACC, A, B)                                                                   \
    do {                                                                                                \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A).p[2], (B).p[0], ACC, 0, 0, 0);               \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A).p[0], (B).p[2], ACC, 0, 0, 0);               \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A).p[1], (B).p[1], ACC, 0, 0, 0);               \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A).p[1], (B).p[0], ACC, 0, 0, 0);               \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A).p[0], (B).p[1], ACC, 0, 0, 0);               \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A).p[0], (B).p[0], ACC, 0, 0, 0);               \
    } while (0)

// a 32-row tile of one head: global (float4, rows clamped into the operand) -> registers -> three bf16 planes in LDS
template <int HD>
struct Tile3 {
    // head_dim 16: the planes are 32 columns wide, columns 16 .. 31 zero (x3_zero_planes), so that the second products still run on the
    // 32 x 32 MFMA; half of the threads move the tile
    static constexpr int LD = (HD < 32 ? 32 : HD) + 8, NV = HD < 32 ? 1 : HD / 32;
    float4 v[NV];
    int nrows_;
    __device__ __forceinline__ void fetch(const float* __restrict__ x, const RowView rv, int b, int h, int r0, int nrows) {
        nrows_ = nrows;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int e = min((int)threadIdx.x + 256 * u, 8 * HD - 1), j = e / (HD / 4), d = (e % (HD / 4)) * 4;
            v[u] = *reinterpret_cast<const float4*>(x + (int64_t)(r0 + min(j, nrows - 1)) * rv.sl + (int64_t)b * rv.sb + h * HD + d);
        }
    }
    __device__ __forceinline__ void store(unsigned short (*pl)[32][LD]) const {
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int e = threadIdx.x + 256 * u, j = e / (HD / 4), d = (e % (HD / 4)) * 4;
            if (HD < 32 && e >= 8 * HD) break;
            const bool ok = j < nrows_;
            const float x[4] = {ok ? v[u].x : 0.f, ok ? v[u].y : 0.f, ok ? v[u].z : 0.f, ok ? v[u].w : 0.f};
            unsigned int w[3][2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float a = x[2 * q], c = x[2 * q + 1];
                const float ar = a - al_trunc(a), cr = c - al_trunc(c);
                const float al = ar - al_trunc(ar), cl = cr - al_trunc(cr);
                w[0][q] = __builtin_amdgcn_perm(__float_as_uint(c), __float_as_uint(a), 0x07060302u);
                w[1][q] = __builtin_amdgcn_perm(__float_as_uint(cr), __float_as_uint(ar), 0x07060302u);
                w[2][q] = __builtin_amdgcn_perm(__float_as_uint(cl), __float_as_uint(al), 0x07060302u);
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(&pl[p][j][d]) = make_uint2(w[p][0], w[p][1]);
        }
    }
};

template <int HD, int LD>
__device__ __forceinline__ void x3_zero_planes(unsigned short (*pl)[32][LD]) {
    if constexpr (HD < 32) {
        for (int e = threadIdx.x; e < 3 * 32 * (LD - HD); e += 256) {
            const int p = e / (32 * (LD - HD)), r = (e / (LD - HD)) % 32, c = HD + e % (LD - HD);
            pl[p][r][c] = 0;
        }
    }
}

// A operand of a first product: row kappa(c) of the tile, k-step ks
template <int LD>
__device__ __forceinline__ Frag3 frag_rows(const unsigned short (*pl)[32][LD], int kap, int ks, int lk) {
    Frag3 f;
#pragma unroll
    for (int p = 0; p < 3; ++p) f.p[p] = *reinterpret_cast<const bf16x8*>(&pl[p][kap][16 * ks + 8 * lk]);
    return f;
}
// B operand of a second product: B[k = tile row 16 ks + 8 lh + 0..7][n = d0 + (lane & 31)] through the transposing read
template <int LD>
__device__ __forceinline__ Frag3 frag_cols(const unsigned short (*pl)[32][LD], int ks, int d0, int lane) {
    const int gq = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    const int kr = ks * 16 + 8 * (gq >> 1) + tq, nc = d0 + 16 * (gq & 1) + 4 * tp;
    Frag3 f;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        union { bf16x8 v; s16x4 h[2]; } u;
        u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(&pl[p][kr][nc]));
        u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(&pl[p][kr + 4][nc]));
        f.p[p] = u.v;
    }
    return f;
}
// the lane's own row as a B operand: x[16 ks + 8 lk + 0..7]
__device__ __forceinline__ Frag3 frag_own(const float* __restrict__ row, int ks, int lk) {
    const float4 a = *reinterpret_cast<const float4*>(row + 16 * ks + 8 * lk), b = *reinterpret_cast<const float4*>(row + 16 * ks + 8 * lk + 4);
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return split8(x);
}

template <int HD, bool OBS>
__global__ __launch_bounds__(256) void k_attn_long_fwd_x3(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                           float* __restrict__ o, float* __restrict__ stats, const AttnGeom g,
                                                           uint32_t* obs_attn, uint32_t* obs_soft) {
    constexpr int LD = Tile3<HD>::LD, KS = HD / 16, ND = HD < 32 ? 1 : HD / 32;
    __shared__ __attribute__((aligned(16))) unsigned short Kp[3][32][LD];
    __shared__ __attribute__((aligned(16))) unsigned short Vp[3][32][LD];
    __shared__ __attribute__((aligned(16))) float scs[4][32];
    constexpr float kLazy = 8.0f;
    x3_zero_planes<HD, LD>(Vp);
    int bx, bh;
    al_block(bx, bh);          // XCD-aware: the query (key) blocks of one head share an XCD's L2
    const int b = bh / g.nh, h = bh % g.nh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, lk = lane >> 5;
    const int it = bx * 4 + wave;
    const int i_own = it * 32 + c;
    const bool live = i_own < g.Lq;
    const float* qp = q + (int64_t)(live ? i_own : g.Lq - 1) * g.q.sl + (int64_t)b * g.q.sb + h * HD;
    Frag3 qb[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qb[ks] = frag_own(qp, ks, lk);
    f32x16 acc[ND];
#pragma unroll
    for (int nd = 0; nd < ND; ++nd)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nd][r] = 0.f;
    float m = -INFINITY, l = 0.f, smin = INFINITY, mtrue = -INFINITY;
    const int kap = kappa(c);
    Tile3<HD> kt, vt;
    kt.fetch(k, g.k, b, h, 0, min(32, g.Lk));
    vt.fetch(v, g.v, b, h, 0, min(32, g.Lk));
    for (int j0 = 0; j0 < g.Lk; j0 += 32) {
        const int nj = min(32, g.Lk - j0);
        __syncthreads();
        kt.store(Kp);
        vt.store(Vp);
        __syncthreads();
        if (j0 + 32 < g.Lk) {
            kt.fetch(k, g.k, b, h, j0 + 32, min(32, g.Lk - j0 - 32));
            vt.fetch(v, g.v, b, h, j0 + 32, min(32, g.Lk - j0 - 32));
        }
        f32x16 T0;
#pragma unroll
        for (int r = 0; r < 16; ++r) T0[r] = 0.f;
        Frag3 ka[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) ka[ks] = frag_rows<LD>(Kp, kap, ks, lk);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) FQSS_X3_PRODUCTS(T0, ka[ks], qb[ks]);
        if (nj < 32) {                                // the last tile: padding keys leave the softmax
#pragma unroll
            for (int r = 0; r < 16; ++r) T0[r] = krow(r, lk) < nj ? T0[r] : -INFINITY;
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, T0[r]);
        if (OBS) {
#pragma unroll
            for (int r = 0; r < 16; ++r) smin = fminf(smin, T0[r] == -INFINITY ? INFINITY : T0[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        if (OBS) mtrue = fmaxf(mtrue, tmax);
        // lazy reference maximum: m only moves when the tile's maximum exceeds it by more than kLazy (probabilities stay below
        // e^kLazy, far inside fp32; o = acc / l and the saved (m, l) are the same function of the logits for ANY reference value),
        // so that after the first few tiles the rescaling pass below is skipped
        const bool move = tmax > m + kLazy;          // first tile: m = -inf
        if (__any(move)) {
            const float m_new = move ? tmax : m;
            const float sc = al_exp(m - m_new);      // 1 for the queries that keep their reference, 0 on the first tile
            l *= sc;
            m = m_new;
            if (j0 > 0) {
                // the factor of the query that owns accumulator row r: rows tile_row(r, lk) = 4 consecutive ones per r >> 2
                if (lk == 0) scs[wave][c] = sc;
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float4 f = *reinterpret_cast<const float4*>(&scs[wave][8 * r4 + 4 * lk]);
                    const float ff[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int nd = 0; nd < ND; ++nd) acc[nd][4 * r4 + e] *= ff[e];
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        Frag3 pa[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float pr[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                pr[e] = al_exp(T0[8 * s2 + e] - m);
                l += pr[e];
            }
            pa[s2] = split8(pr);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int nd = 0; nd < ND; ++nd) {
                const Frag3 vb = frag_cols<LD>(Vp, s2, 32 * nd, lane);
                FQSS_X3_PRODUCTS(acc[nd], pa[s2], vb);
            }
    }
    l += __shfl_xor(l, 32, 64);
    smin = fminf(smin, __shfl_xor(smin, 32, 64));
    if (lk == 0 && live) {
        stats[((int64_t)bh * g.Lq + i_own) * 2] = m;
        stats[((int64_t)bh * g.Lq + i_own) * 2 + 1] = l;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int il = tile_row(r, lk);
        const float lr = __shfl(l, il, 64);
        const int i = it * 32 + il;
        if (i < g.Lq) {
            float* op = o + (int64_t)i * g.o.sl + (int64_t)b * g.o.sb + h * HD + c;
#pragma unroll
            for (int nd = 0; nd < ND; ++nd)
                if (HD >= 32 || c < HD) op[32 * nd] = acc[nd][r] / lr;
        }
    }
    if (OBS) {
        float smin_all = live ? smin : INFINITY, smax_all = live ? mtrue : -INFINITY;
        float pmax_all = live ? expf(mtrue - m) / l : -INFINITY, pmin_all = live ? expf(smin - m) / l : INFINITY;
        smin_all = wave_min(smin_all); smax_all = wave_max(smax_all);
        pmin_all = wave_min(pmin_all); pmax_all = wave_max(pmax_all);
        if (lane == 0 && smin_all <= smax_all) {
            atomicMin(obs_attn, f2ord(smin_all)); atomicMax(obs_attn + 1, f2ord(smax_all));
            atomicMin(obs_soft, f2ord(pmin_all)); atomicMax(obs_soft + 1, f2ord(pmax_all));
        }
    }
}

template <int HD>
__global__ __launch_bounds__(256) void k_attn_long_bwd_q_x3(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                             const float* __restrict__ o, const float* __restrict__ go,
                                                             const float* __restrict__ stats, float* __restrict__ gq, float* __restrict__ dsum,
                                                             const AttnGeom g) {
    constexpr int LD = Tile3<HD>::LD, KS = HD / 16, ND = HD < 32 ? 1 : HD / 32;
    __shared__ __attribute__((aligned(16))) unsigned short Kp[3][32][LD];
    __shared__ __attribute__((aligned(16))) unsigned short Vp[3][32][LD];
    x3_zero_planes<HD, LD>(Kp);
    int bx, bh;
    al_block(bx, bh);          // XCD-aware: the query (key) blocks of one head share an XCD's L2
    const int b = bh / g.nh, h = bh % g.nh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, lk = lane >> 5;
    const int it = bx * 4 + wave;
    const int i_own = it * 32 + c;
    const bool live = i_own < g.Lq;
    const int ic = live ? i_own : g.Lq - 1;
    const float* qp = q + (int64_t)ic * g.q.sl + (int64_t)b * g.q.sb + h * HD;
    const float* gp = go + (int64_t)ic * g.go.sl + (int64_t)b * g.go.sb + h * HD;
    const float* op = o + (int64_t)ic * g.o.sl + (int64_t)b * g.o.sb + h * HD;
    Frag3 qb[KS], gb[KS];
    float D = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qb[ks] = frag_own(qp, ks, lk);
        gb[ks] = frag_own(gp, ks, lk);
#pragma unroll
        for (int e = 0; e < 8; ++e) D = fmaf(gp[16 * ks + 8 * lk + e], op[16 * ks + 8 * lk + e], D);
    }
    D += __shfl_xor(D, 32, 64);
    const float m = stats[((int64_t)bh * g.Lq + ic) * 2], rl = 1.0f / stats[((int64_t)bh * g.Lq + ic) * 2 + 1];
    f32x16 acc[ND];
#pragma unroll
    for (int nd = 0; nd < ND; ++nd)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nd][r] = 0.f;
    const int kap = kappa(c);
    Tile3<HD> kt, vt;
    kt.fetch(k, g.k, b, h, 0, min(32, g.Lk));
    vt.fetch(v, g.v, b, h, 0, min(32, g.Lk));
    for (int j0 = 0; j0 < g.Lk; j0 += 32) {
        const int nj = min(32, g.Lk - j0);
        __syncthreads();
        kt.store(Kp);
        vt.store(Vp);
        __syncthreads();
        if (j0 + 32 < g.Lk) {
            kt.fetch(k, g.k, b, h, j0 + 32, min(32, g.Lk - j0 - 32));
            vt.fetch(v, g.v, b, h, j0 + 32, min(32, g.Lk - j0 - 32));
        }
        f32x16 T0, U0;
#pragma unroll
        for (int r = 0; r < 16; ++r) T0[r] = U0[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const Frag3 ka = frag_rows<LD>(Kp, kap, ks, lk), va = frag_rows<LD>(Vp, kap, ks, lk);
            FQSS_X3_PRODUCTS(T0, ka, qb[ks]);
            FQSS_X3_PRODUCTS(U0, va, gb[ks]);
        }
        if (nj < 32) {
#pragma unroll
            for (int r = 0; r < 16; ++r) T0[r] = krow(r, lk) < nj ? T0[r] : -INFINITY;      // probability 0
        }
        Frag3 da[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float ds[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int r = 8 * s2 + e;
                ds[e] = al_exp(T0[r] - m) * rl * (U0[r] - D);
            }
            da[s2] = split8(ds);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int nd = 0; nd < ND; ++nd) {
                const Frag3 kb = frag_cols<LD>(Kp, s2, 32 * nd, lane);
                FQSS_X3_PRODUCTS(acc[nd], da[s2], kb);
            }
    }
    if (lk == 0 && live) dsum[(int64_t)bh * g.Lq + i_own] = D;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = it * 32 + tile_row(r, lk);
        if (i < g.Lq) {
            float* gp2 = gq + (int64_t)i * g.gq.sl + (int64_t)b * g.gq.sb + h * HD + c;
#pragma unroll
            for (int nd = 0; nd < ND; ++nd)
                if (HD >= 32 || c < HD) gp2[32 * nd] = acc[nd][r];
        }
    }
}

template <int HD>
__global__ __launch_bounds__(256) void k_attn_long_bwd_kv_x3(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                              const float* __restrict__ go, const float* __restrict__ stats,
                                                              const float* __restrict__ dsum, float* __restrict__ gk, float* __restrict__ gv,
                                                              const AttnGeom g) {
    constexpr int LD = Tile3<HD>::LD, KS = HD / 16, ND = HD < 32 ? 1 : HD / 32;
    __shared__ __attribute__((aligned(16))) unsigned short Qp[3][32][LD];
    __shared__ __attribute__((aligned(16))) unsigned short Gp[3][32][LD];
    __shared__ float Ms[32], Rs[32], Ds[32];
    x3_zero_planes<HD, LD>(Qp);
    x3_zero_planes<HD, LD>(Gp);
    int bx, bh;
    al_block(bx, bh);          // XCD-aware: the query (key) blocks of one head share an XCD's L2
    const int b = bh / g.nh, h = bh % g.nh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, lk = lane >> 5;
    const int jt = bx * 4 + wave;
    const int j_own = jt * 32 + c;
    const int jc = j_own < g.Lk ? j_own : g.Lk - 1;
    const float* kp = k + (int64_t)jc * g.k.sl + (int64_t)b * g.k.sb + h * HD;
    const float* vp = v + (int64_t)jc * g.v.sl + (int64_t)b * g.v.sb + h * HD;
    Frag3 kb[KS], vb[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { kb[ks] = frag_own(kp, ks, lk); vb[ks] = frag_own(vp, ks, lk); }
    f32x16 ak[ND], av[ND];
#pragma unroll
    for (int nd = 0; nd < ND; ++nd)
#pragma unroll
        for (int r = 0; r < 16; ++r) ak[nd][r] = av[nd][r] = 0.f;
    const int kap = kappa(c);
    Tile3<HD> qt, gt;
    qt.fetch(q, g.q, b, h, 0, min(32, g.Lq));
    gt.fetch(go, g.go, b, h, 0, min(32, g.Lq));
    for (int i0 = 0; i0 < g.Lq; i0 += 32) {
        const int ni = min(32, g.Lq - i0);
        __syncthreads();
        qt.store(Qp);
        gt.store(Gp);
        if (threadIdx.x < 32) {
            const bool ok = threadIdx.x < ni;
            const int64_t si = (int64_t)bh * g.Lq + i0 + (ok ? threadIdx.x : 0);
            Ms[threadIdx.x] = ok ? stats[si * 2] : 0.f;
            Rs[threadIdx.x] = ok ? 1.0f / stats[si * 2 + 1] : 0.f;      // padding queries: probability 0
            Ds[threadIdx.x] = ok ? dsum[si] : 0.f;
        }
        __syncthreads();
        if (i0 + 32 < g.Lq) {
            qt.fetch(q, g.q, b, h, i0 + 32, min(32, g.Lq - i0 - 32));
            gt.fetch(go, g.go, b, h, i0 + 32, min(32, g.Lq - i0 - 32));
        }
        f32x16 T0, U0;
#pragma unroll
        for (int r = 0; r < 16; ++r) T0[r] = U0[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const Frag3 qa = frag_rows<LD>(Qp, kap, ks, lk), ga = frag_rows<LD>(Gp, kap, ks, lk);
            FQSS_X3_PRODUCTS(T0, qa, kb[ks]);
            FQSS_X3_PRODUCTS(U0, ga, vb[ks]);
        }
        Frag3 pa[2], da[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float pv[8], ds[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int r = 8 * s2 + e, row = krow(r, lk);
                pv[e] = al_exp(T0[r] - Ms[row]) * Rs[row];
                ds[e] = pv[e] * (U0[r] - Ds[row]);
            }
            pa[s2] = split8(pv);
            da[s2] = split8(ds);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int nd = 0; nd < ND; ++nd) {
                const Frag3 gr = frag_cols<LD>(Gp, s2, 32 * nd, lane), qr = frag_cols<LD>(Qp, s2, 32 * nd, lane);
                FQSS_X3_PRODUCTS(av[nd], pa[s2], gr);
                FQSS_X3_PRODUCTS(ak[nd], da[s2], qr);
            }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int j = jt * 32 + tile_row(r, lk);
        if (j < g.Lk) {
            float* gkp = gk + (int64_t)j * g.gk.sl + (int64_t)b * g.gk.sb + h * HD + c;
            float* gvp = gv + (int64_t)j * g.gv.sl + (int64_t)b * g.gv.sb + h * HD + c;
#pragma unroll
            for (int nd = 0; nd < ND; ++nd)
                if (HD >= 32 || c < HD) { gkp[32 * nd] = ak[nd][r]; gvp[32 * nd] = av[nd][r]; }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------
// CODED operands (the student in its quantizing phase): q, k, v arrive as the 8-bit codes of their quantizers, x = delta c + lo
// (csrc/dualpath.hip fqss_mha_prep_fwd_c).  An integer code is ONE exact bf16 plane, and terms that are constant along the softmax
// axis drop out of it:
//   s_ij = q_i . k_j = dk (q_i . ck_j) + lo_k sum_d q_id            -> s'_ij = (dk q_i) . ck_j       3 products (q: 3 pieces, ck: 1)
//   o_i  = sum_j p_ij v_j = dv (sum_j p_ij cv_j) + lo_v              (sum_j p_ij = 1)                 3 products (p: 3, cv: 1)
//   dp_ij - D_i = dv (dO_i . cv_j) - D'_i,  D'_i = D_i - lo_v sum_d dO_id                            3 products (dO: 3, cv: 1)
//   dq_i = dk sum_j ds_ij ck_j + lo_k sum_j ds_ij                                                     3 products (ds: 3, ck: 1)
//   dk_j = dq' sum_i ds_ij cq_i + lo_q sum_i ds_ij   (dq', lo_q: the q quantizer's grid)              3 products (ds: 3, cq: 1)
//   dv_j = sum_i p_ij dO_i                                                                            6 products (no codes)
//   dk/dv pass: s'_ij = dk (dq' (cq_i . ck_j) + lo_q sum_d ck_jd): ONE product, exact integers
// so the forward issues 24 MFMAs per 32 x 32 tile instead of 48, dq 36 instead of 72, dk/dv 52 instead of 96, the K / V / Q tiles
// are 1 B per element in HBM and one plane in LDS, and their 3-way split disappears.  The saved statistics (m, l) refer to s'.
// Same tiling, lane <-> row maps, lazy reference maximum and LDS idioms as the split-bf16 kernels above.
// ------------------------------------------------------------------------------------------------------------------
#define FQSS_X3_A1_B3(ACC, A1, B)                                                                     \
    do {                                                                                                \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A1), (B).p[2], ACC, 0, 0, 0);                   \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A1), (B).p[1], ACC, 0, 0, 0);                   \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A1), (B).p[0], ACC, 0, 0, 0);                   \
    } while (0)
#define FQSS_X3_A3_B1(ACC, A, B1)                                                                     \
    do {                                                                                                \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A).p[2], (B1), ACC, 0, 0, 0);                   \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A).p[1], (B1), ACC, 0, 0, 0);                   \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A).p[0], (B1), ACC, 0, 0, 0);                   \
    } while (0)

// 8 codes (two 32-bit words) -> 8 floats / one exact bf16x8
__device__ __forceinline__ void codes_to_f(uint2 w, float (&c)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        c[e] = (float)((w.x >> (8 * e)) & 0xFFu);
        c[4 + e] = (float)((w.y >> (8 * e)) & 0xFFu);
    }
}
__device__ __forceinline__ bf16x8 codes_to_bf(uint2 w) {
    float c[8];
    codes_to_f(w, c);
    union { bf16x8 v; unsigned int u[4]; } r;
#pragma unroll
    for (int q = 0; q < 4; ++q) r.u[q] = __builtin_amdgcn_perm(__float_as_uint(c[2 * q + 1]), __float_as_uint(c[2 * q]), 0x07060302u);
    return r.v;
}

// a 32-row tile of codes of one head: global (8 B per thread, rows clamped) -> registers -> ONE bf16 plane in LDS
template <int HD>
struct TileC {
    static constexpr int LD = (HD < 32 ? 32 : HD) + 8;
    uint2 v;
    int nrows_;
    __device__ __forceinline__ void fetch(const unsigned char* __restrict__ x, const RowView rv, int b, int h, int r0, int nrows) {
        nrows_ = nrows;
        const int e = min((int)threadIdx.x, 4 * HD - 1), j = e / (HD / 8), d = (e % (HD / 8)) * 8;
        v = *reinterpret_cast<const uint2*>(x + (int64_t)(r0 + min(j, nrows - 1)) * rv.sl + (int64_t)b * rv.sb + h * HD + d);
    }
    __device__ __forceinline__ void store(unsigned short (*pl)[LD]) const {
        const int e = threadIdx.x, j = e / (HD / 8), d = (e % (HD / 8)) * 8;
        if (e >= 4 * HD) return;
        union { bf16x8 b; uint4 u; } r;
        r.b = codes_to_bf(j < nrows_ ? v : make_uint2(0u, 0u));
        *reinterpret_cast<uint4*>(&pl[j][d]) = r.u;
    }
};
template <int HD, int LD>
__device__ __forceinline__ void x3_zero_plane1(unsigned short (*pl)[LD]) {
    if constexpr (HD < 32) {
        for (int e = threadIdx.x; e < 32 * (LD - HD); e += 256) pl[e / (LD - HD)][HD + e % (LD - HD)] = 0;
    }
}
template <int LD>
__device__ __forceinline__ bf16x8 frag_rows1(const unsigned short (*pl)[LD], int kap, int ks, int lk) {
    return *reinterpret_cast<const bf16x8*>(&pl[kap][16 * ks + 8 * lk]);
}
template <int LD>
__device__ __forceinline__ bf16x8 frag_cols1(const unsigned short (*pl)[LD], int ks, int d0, int lane) {
    const int gq = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    const int kr = ks * 16 + 8 * (gq >> 1) + tq, nc = d0 + 16 * (gq & 1) + 4 * tp;
    union { bf16x8 v; s16x4 h[2]; } u;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(&pl[kr][nc]));
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(&pl[kr + 4][nc]));
    return u.v;
}
// the lane's own query row from its codes: (dq' c + lo_q) * dk, split in three (the same operations in the forward and in the dq pass)
__device__ __forceinline__ Frag3 own_q_scaled(const unsigned char* __restrict__ row, int ks, int lk, const QRange rq, float dk) {
    float c[8];
    codes_to_f(*reinterpret_cast<const uint2*>(row + 16 * ks + 8 * lk), c);
#pragma unroll
    for (int e = 0; e < 8; ++e) c[e] = (rq.delta * c[e] + rq.lo) * dk;
    return split8(c);
}

struct AttnRanges {       // device scalars: the grids of q (behind the division), k, v
    const float *q_lo, *q_hi, *k_lo, *k_hi, *v_lo, *v_hi;
};

template <int HD>
__global__ __launch_bounds__(256) void k_attn_long_fwd_c(const unsigned char* __restrict__ qc, const unsigned char* __restrict__ kc,
                                                          const unsigned char* __restrict__ vc, float* __restrict__ o, float* __restrict__ stats,
                                                          const AttnGeom g, const AttnRanges rg) {
    constexpr int LD = TileC<HD>::LD, KS = HD / 16, ND = HD < 32 ? 1 : HD / 32;
    __shared__ __attribute__((aligned(16))) unsigned short Kp[32][LD];
    __shared__ __attribute__((aligned(16))) unsigned short Vp[32][LD];
    __shared__ __attribute__((aligned(16))) float scs[4][32];
    constexpr float kLazy = 8.0f;
    x3_zero_plane1<HD, LD>(Vp);
    const QRange rq = load_qrange(rg.q_lo, rg.q_hi), rk = load_qrange(rg.k_lo, rg.k_hi), rv = load_qrange(rg.v_lo, rg.v_hi);
    int bx, bh;
    al_block(bx, bh);          // XCD-aware: the query (key) blocks of one head share an XCD's L2
    const int b = bh / g.nh, h = bh % g.nh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, lk = lane >> 5;
    const int it = bx * 4 + wave;
    const int i_own = it * 32 + c;
    const bool live = i_own < g.Lq;
    const unsigned char* qp = qc + (int64_t)(live ? i_own : g.Lq - 1) * g.q.sl + (int64_t)b * g.q.sb + h * HD;
    Frag3 qb[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qb[ks] = own_q_scaled(qp, ks, lk, rq, rk.delta);
    f32x16 acc[ND];
#pragma unroll
    for (int nd = 0; nd < ND; ++nd)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nd][r] = 0.f;
    float m = -INFINITY, l = 0.f;
    const int kap = kappa(c);
    TileC<HD> kt, vt;
    kt.fetch(kc, g.k, b, h, 0, min(32, g.Lk));
    vt.fetch(vc, g.v, b, h, 0, min(32, g.Lk));
    for (int j0 = 0; j0 < g.Lk; j0 += 32) {
        const int nj = min(32, g.Lk - j0);
        __syncthreads();
        kt.store(Kp);
        vt.store(Vp);
        __syncthreads();
        if (j0 + 32 < g.Lk) {
            kt.fetch(kc, g.k, b, h, j0 + 32, min(32, g.Lk - j0 - 32));
            vt.fetch(vc, g.v, b, h, j0 + 32, min(32, g.Lk - j0 - 32));
        }
        f32x16 T0;
#pragma unroll
        for (int r = 0; r < 16; ++r) T0[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 ka = frag_rows1<LD>(Kp, kap, ks, lk);
            FQSS_X3_A1_B3(T0, ka, qb[ks]);
        }
        if (nj < 32) {
#pragma unroll
            for (int r = 0; r < 16; ++r) T0[r] = krow(r, lk) < nj ? T0[r] : -INFINITY;
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, T0[r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const bool move = tmax > m + kLazy;
        if (__any(move)) {
            const float m_new = move ? tmax : m;
            const float sc = al_exp(m - m_new);
            l *= sc;
            m = m_new;
            if (j0 > 0) {
                if (lk == 0) scs[wave][c] = sc;
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float4 f = *reinterpret_cast<const float4*>(&scs[wave][8 * r4 + 4 * lk]);
                    const float ff[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int nd = 0; nd < ND; ++nd) acc[nd][4 * r4 + e] *= ff[e];
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        Frag3 pa[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float pr[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                pr[e] = al_exp(T0[8 * s2 + e] - m);
                l += pr[e];
            }
            pa[s2] = split8(pr);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int nd = 0; nd < ND; ++nd) {
                const bf16x8 vb = frag_cols1<LD>(Vp, s2, 32 * nd, lane);
                FQSS_X3_A3_B1(acc[nd], pa[s2], vb);
            }
    }
    l += __shfl_xor(l, 32, 64);
    if (lk == 0 && live) {
        stats[((int64_t)bh * g.Lq + i_own) * 2] = m;
        stats[((int64_t)bh * g.Lq + i_own) * 2 + 1] = l;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int il = tile_row(r, lk);
        const float lr = __shfl(l, il, 64);
        const int i = it * 32 + il;
        if (i < g.Lq) {
            float* op = o + (int64_t)i * g.o.sl + (int64_t)b * g.o.sb + h * HD + c;
#pragma unroll
            for (int nd = 0; nd < ND; ++nd)
                if (HD >= 32 || c < HD) op[32 * nd] = rv.delta * (acc[nd][r] / lr) + rv.lo;
        }
    }
}

template <int HD>
__global__ __launch_bounds__(256) void k_attn_long_bwd_q_c(const unsigned char* __restrict__ qc, const unsigned char* __restrict__ kc,
                                                            const unsigned char* __restrict__ vc, const float* __restrict__ o,
                                                            const float* __restrict__ go, const float* __restrict__ stats, float* __restrict__ gq,
                                                            float* __restrict__ dsum, const AttnGeom g, const AttnRanges rg) {
    constexpr int LD = TileC<HD>::LD, KS = HD / 16, ND = HD < 32 ? 1 : HD / 32;
    __shared__ __attribute__((aligned(16))) unsigned short Kp[32][LD];
    __shared__ __attribute__((aligned(16))) unsigned short Vp[32][LD];
    __shared__ __attribute__((aligned(16))) float scs[4][32];
    x3_zero_plane1<HD, LD>(Kp);
    const QRange rq = load_qrange(rg.q_lo, rg.q_hi), rk = load_qrange(rg.k_lo, rg.k_hi), rv = load_qrange(rg.v_lo, rg.v_hi);
    int bx, bh;
    al_block(bx, bh);          // XCD-aware: the query (key) blocks of one head share an XCD's L2
    const int b = bh / g.nh, h = bh % g.nh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, lk = lane >> 5;
    const int it = bx * 4 + wave;
    const int i_own = it * 32 + c;
    const bool live = i_own < g.Lq;
    const int ic = live ? i_own : g.Lq - 1;
    const unsigned char* qp = qc + (int64_t)ic * g.q.sl + (int64_t)b * g.q.sb + h * HD;
    const float* gp = go + (int64_t)ic * g.go.sl + (int64_t)b * g.go.sb + h * HD;
    const float* op = o + (int64_t)ic * g.o.sl + (int64_t)b * g.o.sb + h * HD;
    Frag3 qb[KS], gb[KS];
    float D = 0.f, gs = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qb[ks] = own_q_scaled(qp, ks, lk, rq, rk.delta);
        gb[ks] = frag_own(gp, ks, lk);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float gg = gp[16 * ks + 8 * lk + e];
            D = fmaf(gg, op[16 * ks + 8 * lk + e], D);
            gs += gg;
        }
    }
    D += __shfl_xor(D, 32, 64);
    gs += __shfl_xor(gs, 32, 64);
    const float Dp = D - rv.lo * gs;               // D' = D - lo_v sum_d dO
    const float m = stats[((int64_t)bh * g.Lq + ic) * 2], rl = 1.0f / stats[((int64_t)bh * g.Lq + ic) * 2 + 1];
    f32x16 acc[ND];
#pragma unroll
    for (int nd = 0; nd < ND; ++nd)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nd][r] = 0.f;
    float dss = 0.f;                               // sum_j ds_ij of this lane's query (its half of the keys)
    const int kap = kappa(c);
    TileC<HD> kt, vt;
    kt.fetch(kc, g.k, b, h, 0, min(32, g.Lk));
    vt.fetch(vc, g.v, b, h, 0, min(32, g.Lk));
    for (int j0 = 0; j0 < g.Lk; j0 += 32) {
        const int nj = min(32, g.Lk - j0);
        __syncthreads();
        kt.store(Kp);
        vt.store(Vp);
        __syncthreads();
        if (j0 + 32 < g.Lk) {
            kt.fetch(kc, g.k, b, h, j0 + 32, min(32, g.Lk - j0 - 32));
            vt.fetch(vc, g.v, b, h, j0 + 32, min(32, g.Lk - j0 - 32));
        }
        f32x16 T0, U0;
#pragma unroll
        for (int r = 0; r < 16; ++r) T0[r] = U0[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 ka = frag_rows1<LD>(Kp, kap, ks, lk), va = frag_rows1<LD>(Vp, kap, ks, lk);
            FQSS_X3_A1_B3(T0, ka, qb[ks]);
            FQSS_X3_A1_B3(U0, va, gb[ks]);
        }
        if (nj < 32) {
#pragma unroll
            for (int r = 0; r < 16; ++r) T0[r] = krow(r, lk) < nj ? T0[r] : -INFINITY;
        }
        Frag3 da[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float ds[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int r = 8 * s2 + e;
                ds[e] = al_exp(T0[r] - m) * rl * (rv.delta * U0[r] - Dp);
                dss += ds[e];
            }
            da[s2] = split8(ds);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int nd = 0; nd < ND; ++nd) {
                const bf16x8 kb = frag_cols1<LD>(Kp, s2, 32 * nd, lane);
                FQSS_X3_A3_B1(acc[nd], da[s2], kb);
            }
    }
    dss += __shfl_xor(dss, 32, 64);
    if (lk == 0 && live) dsum[(int64_t)bh * g.Lq + i_own] = Dp;
    if (lk == 0) scs[wave][c] = dss;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int il = tile_row(r, lk);
        const int i = it * 32 + il;
        const float sj = scs[wave][il];
        if (i < g.Lq) {
            float* gp2 = gq + (int64_t)i * g.gq.sl + (int64_t)b * g.gq.sb + h * HD + c;
#pragma unroll
            for (int nd = 0; nd < ND; ++nd)
                if (HD >= 32 || c < HD) gp2[32 * nd] = rk.delta * acc[nd][r] + rk.lo * sj;
        }
    }
}

template <int HD>
__global__ __launch_bounds__(256) void k_attn_long_bwd_kv_c(const unsigned char* __restrict__ qc, const unsigned char* __restrict__ kc,
                                                             const unsigned char* __restrict__ vc, const float* __restrict__ go,
                                                             const float* __restrict__ stats, const float* __restrict__ dsum,
                                                             float* __restrict__ gk, float* __restrict__ gv, const AttnGeom g, const AttnRanges rg) {
    constexpr int LD = TileC<HD>::LD, KS = HD / 16, ND = HD < 32 ? 1 : HD / 32;
    static_assert(LD == Tile3<HD>::LD, "one row stride for both tile kinds");
    __shared__ __attribute__((aligned(16))) unsigned short Qp[32][LD];
    __shared__ __attribute__((aligned(16))) unsigned short Gp[3][32][LD];
    __shared__ float Ms[32], Rs[32], Ds[32];
    __shared__ __attribute__((aligned(16))) float scs[4][32];
    x3_zero_plane1<HD, LD>(Qp);
    x3_zero_planes<HD, LD>(Gp);
    const QRange rq = load_qrange(rg.q_lo, rg.q_hi), rk = load_qrange(rg.k_lo, rg.k_hi), rv = load_qrange(rg.v_lo, rg.v_hi);
    int bx, bh;
    al_block(bx, bh);          // XCD-aware: the query (key) blocks of one head share an XCD's L2
    const int b = bh / g.nh, h = bh % g.nh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, lk = lane >> 5;
    const int jt = bx * 4 + wave;
    const int j_own = jt * 32 + c;
    const int jc = j_own < g.Lk ? j_own : g.Lk - 1;
    const unsigned char* kp = kc + (int64_t)jc * g.k.sl + (int64_t)b * g.k.sb + h * HD;
    const unsigned char* vp = vc + (int64_t)jc * g.v.sl + (int64_t)b * g.v.sb + h * HD;
    bf16x8 kb[KS], vb[KS];
    float sk = 0.f;                                 // sum_d ck_jd
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const uint2 wk = *reinterpret_cast<const uint2*>(kp + 16 * ks + 8 * lk), wv = *reinterpret_cast<const uint2*>(vp + 16 * ks + 8 * lk);
        float cf[8];
        codes_to_f(wk, cf);
#pragma unroll
        for (int e = 0; e < 8; ++e) sk += cf[e];
        kb[ks] = codes_to_bf(wk);
        vb[ks] = codes_to_bf(wv);
    }
    sk += __shfl_xor(sk, 32, 64);
    const float t_c0 = rk.delta * rq.delta, t_c1 = rk.delta * (rq.lo * sk);     // s'_ij = t_c0 I_ij + t_c1
    f32x16 ak[ND], av[ND];
#pragma unroll
    for (int nd = 0; nd < ND; ++nd)
#pragma unroll
        for (int r = 0; r < 16; ++r) ak[nd][r] = av[nd][r] = 0.f;
    float dss = 0.f;                                // sum_i ds_ij of this lane's key (its half of the queries)
    const int kap = kappa(c);
    TileC<HD> qt;
    Tile3<HD> gt;
    qt.fetch(qc, g.q, b, h, 0, min(32, g.Lq));
    gt.fetch(go, g.go, b, h, 0, min(32, g.Lq));
    for (int i0 = 0; i0 < g.Lq; i0 += 32) {
        const int ni = min(32, g.Lq - i0);
        __syncthreads();
        qt.store(Qp);
        gt.store(Gp);
        if (threadIdx.x < 32) {
            const bool ok = threadIdx.x < ni;
            const int64_t si = (int64_t)bh * g.Lq + i0 + (ok ? threadIdx.x : 0);
            Ms[threadIdx.x] = ok ? stats[si * 2] : INFINITY;             // padding queries: exp(-inf) = 0
            Rs[threadIdx.x] = ok ? 1.0f / stats[si * 2 + 1] : 0.f;
            Ds[threadIdx.x] = ok ? dsum[si] : 0.f;
        }
        __syncthreads();
        if (i0 + 32 < g.Lq) {
            qt.fetch(qc, g.q, b, h, i0 + 32, min(32, g.Lq - i0 - 32));
            gt.fetch(go, g.go, b, h, i0 + 32, min(32, g.Lq - i0 - 32));
        }
        f32x16 T0, U0;
#pragma unroll
        for (int r = 0; r < 16; ++r) T0[r] = U0[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 qa = frag_rows1<LD>(Qp, kap, ks, lk);
            const Frag3 ga = frag_rows<LD>(Gp, kap, ks, lk);
            T0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kb[ks], T0, 0, 0, 0);
            FQSS_X3_A3_B1(U0, ga, vb[ks]);
        }
        Frag3 pa[2], da[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float pv[8], ds[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int r = 8 * s2 + e, row = krow(r, lk);
                pv[e] = al_exp((t_c0 * T0[r] + t_c1) - Ms[row]) * Rs[row];
                ds[e] = pv[e] * (rv.delta * U0[r] - Ds[row]);
                dss += ds[e];
            }
            pa[s2] = split8(pv);
            da[s2] = split8(ds);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int nd = 0; nd < ND; ++nd) {
                const Frag3 gr = frag_cols<LD>(Gp, s2, 32 * nd, lane);
                const bf16x8 qr = frag_cols1<LD>(Qp, s2, 32 * nd, lane);
                FQSS_X3_PRODUCTS(av[nd], pa[s2], gr);
                FQSS_X3_A3_B1(ak[nd], da[s2], qr);
            }
    }
    dss += __shfl_xor(dss, 32, 64);
    if (lk == 0) scs[wave][c] = dss;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int jl = tile_row(r, lk);
        const int j = jt * 32 + jl;
        const float sj = scs[wave][jl];
        if (j < g.Lk) {
            float* gkp = gk + (int64_t)j * g.gk.sl + (int64_t)b * g.gk.sb + h * HD + c;
            float* gvp = gv + (int64_t)j * g.gv.sl + (int64_t)b * g.gv.sb + h * HD + c;
#pragma unroll
            for (int nd = 0; nd < ND; ++nd)
                if (HD >= 32 || c < HD) { gkp[32 * nd] = rq.delta * ak[nd][r] + rq.lo * sj; gvp[32 * nd] = av[nd][r]; }
        }
    }
}

// 0: vector ALU, 1: fp32 MFMA, 2: split-bf16 MFMA (default where head_dim and alignment allow)
static int attn_mfma_mode() {
    static const int mode = [] {
        const char* e = getenv("FQSS_ATTN_MFMA");
        if (e && e[0] == '0') return 0;
        if (e && (e[0] == 'f' || e[0] == '1')) return 1;
        return 2;
    }();
    return mode;
}
static bool rows_aligned16(const void* const* ptrs, int np, const int64_t* st, int ns) {
    for (int t = 0; t < np; ++t) if (!aligned16(ptrs[t])) return false;
    for (int t = 0; t < ns; ++t) if (st[t] % 4 != 0) return false;
    return true;
}

static int check_attn(int Lq, int Lk, int B, int nh, int hd, const int64_t* st, int n) {
    FQSS_REQUIRE(Lq > 0 && Lk > 0 && B > 0 && nh > 0 && (int64_t)B * nh <= 65535, "bad shape");
    for (int t = 0; t < n; ++t) FQSS_REQUIRE(st[2 * t] >= (int64_t)nh * hd && st[2 * t + 1] >= (int64_t)nh * hd, "row stride below embed dim");
    return FQSS_OK;
}

}  // namespace fqss

using namespace fqss;

#define FQSS_HD_SWITCH(hd, CALL)                                                                                   \
    switch (hd) {                                                                                                  \
        case 2: CALL(2) break;                                                                                     \
        case 4: CALL(4) break;                                                                                     \
        case 8: CALL(8) break;                                                                                     \
        case 16: CALL(16) break;                                                                                   \
        case 32: CALL(32) break;                                                                                   \
        case 48: CALL(48) break;                                                                                   \
        case 64: CALL(64) break;                                                                                   \
        default: set_error("%s: head_dim %d not built (2, 4, 8, 16, 32, 48, 64)", __func__, hd); return FQSS_EINVAL; \
    }

// strides: (sl, sb) pairs in elements for q, k, v, o
extern "C" int fqss_attn_long_fwd(const float* q, const float* k, const float* v, float* o, float* stats, int Lq, int Lk, int B, int nh,
                                  int hd, const int64_t* strides, uint32_t* obs_attn, uint32_t* obs_soft, fqss_stream_t stream) {
    FQSS_REQUIRE(q && k && v && o && stats && strides, "null tensor");
    FQSS_REQUIRE((obs_attn == nullptr) == (obs_soft == nullptr), "observer workspaces come in pairs");
    if (int rc = check_attn(Lq, Lk, B, nh, hd, strides, 4)) return rc;
    AttnGeom g{Lq, Lk, B, nh, {strides[0], strides[1]}, {strides[2], strides[3]}, {strides[4], strides[5]}, {strides[6], strides[7]}, {}, {}, {}, {}};
    const int mode = attn_mfma_mode();
    const bool use_mfma = mode != 0;
    if (mode == 2 && (hd == 16 || hd == 32 || hd == 64)) {
        const void* ptrs[3] = {q, k, v};
        if (rows_aligned16(ptrs, 3, strides, 6)) {
            dim3 gm((unsigned)cdiv(Lq, 128), (unsigned)(B * nh));
            const bool obs = obs_attn != nullptr;
#define FQSS_AL_FWD(HD_, OBS_) hipLaunchKernelGGL((k_attn_long_fwd_x3<HD_, OBS_>), gm, dim3(256), 0, (hipStream_t)stream, q, k, v, o, stats, g, obs_attn, obs_soft)
            if (hd == 16) { if (obs) FQSS_AL_FWD(16, true); else FQSS_AL_FWD(16, false); }
            else if (hd == 32) { if (obs) FQSS_AL_FWD(32, true); else FQSS_AL_FWD(32, false); }
            else { if (obs) FQSS_AL_FWD(64, true); else FQSS_AL_FWD(64, false); }
#undef FQSS_AL_FWD
            return launch_status("fqss_attn_long_fwd");
        }
    }
    if (use_mfma && (hd == 32 || hd == 64)) {
        dim3 gm((unsigned)cdiv(Lq, 128), (unsigned)(B * nh));
        if (hd == 32) hipLaunchKernelGGL((k_attn_long_fwd_mfma<32>), gm, dim3(256), 0, (hipStream_t)stream, q, k, v, o, stats, g, obs_attn, obs_soft);
        else hipLaunchKernelGGL((k_attn_long_fwd_mfma<64>), gm, dim3(256), 0, (hipStream_t)stream, q, k, v, o, stats, g, obs_attn, obs_soft);
        return launch_status("fqss_attn_long_fwd");
    }
    dim3 grid((unsigned)cdiv(Lq, 256), (unsigned)(B * nh));
#define CALL(HD_) hipLaunchKernelGGL((k_attn_long_fwd<HD_>), grid, dim3(256), 0, (hipStream_t)stream, q, k, v, o, stats, g, obs_attn, obs_soft);
    FQSS_HD_SWITCH(hd, CALL)
#undef CALL
    return launch_status("fqss_attn_long_fwd");
}

// strides: (sl, sb) pairs for q, k, v, o, go, gq, gk, gv;  dsum: workspace of B*nh*Lq floats
extern "C" int fqss_attn_long_bwd(const float* q, const float* k, const float* v, const float* o, const float* go, const float* stats,
                                  float* gq, float* gk, float* gv, float* dsum, int Lq, int Lk, int B, int nh, int hd,
                                  const int64_t* strides, fqss_stream_t stream) {
    FQSS_REQUIRE(q && k && v && o && go && stats && gq && gk && gv && dsum && strides, "null tensor");
    if (int rc = check_attn(Lq, Lk, B, nh, hd, strides, 8)) return rc;
    const int64_t* s = strides;
    AttnGeom g{Lq, Lk, B, nh, {s[0], s[1]}, {s[2], s[3]}, {s[4], s[5]}, {s[6], s[7]}, {s[8], s[9]}, {s[10], s[11]}, {s[12], s[13]}, {s[14], s[15]}};
    const int mode = attn_mfma_mode();
    const bool use_mfma = mode != 0;
    if (mode == 2 && (hd == 16 || hd == 32 || hd == 64)) {
        const void* ptrs[5] = {q, k, v, o, go};
        if (rows_aligned16(ptrs, 5, strides, 10)) {
            dim3 gq_((unsigned)cdiv(Lq, 128), (unsigned)(B * nh)), gk_((unsigned)cdiv(Lk, 128), (unsigned)(B * nh));
            hipStream_t st = (hipStream_t)stream;
            if (hd == 16) {
                hipLaunchKernelGGL((k_attn_long_bwd_q_x3<16>), gq_, dim3(256), 0, st, q, k, v, o, go, stats, gq, dsum, g);
                hipLaunchKernelGGL((k_attn_long_bwd_kv_x3<16>), gk_, dim3(256), 0, st, q, k, v, go, stats, dsum, gk, gv, g);
            } else if (hd == 32) {
                hipLaunchKernelGGL((k_attn_long_bwd_q_x3<32>), gq_, dim3(256), 0, st, q, k, v, o, go, stats, gq, dsum, g);
                hipLaunchKernelGGL((k_attn_long_bwd_kv_x3<32>), gk_, dim3(256), 0, st, q, k, v, go, stats, dsum, gk, gv, g);
            } else {
                hipLaunchKernelGGL((k_attn_long_bwd_q_x3<64>), gq_, dim3(256), 0, st, q, k, v, o, go, stats, gq, dsum, g);
                hipLaunchKernelGGL((k_attn_long_bwd_kv_x3<64>), gk_, dim3(256), 0, st, q, k, v, go, stats, dsum, gk, gv, g);
            }
            return launch_status("fqss_attn_long_bwd");
        }
    }
    if (use_mfma && (hd == 32 || hd == 64)) {
        dim3 gq_((unsigned)cdiv(Lq, 128), (unsigned)(B * nh)), gk_((unsigned)cdiv(Lk, 128), (unsigned)(B * nh));
        hipStream_t st = (hipStream_t)stream;
        if (hd == 32) {
            hipLaunchKernelGGL((k_attn_long_bwd_q_mfma<32>), gq_, dim3(256), 0, st, q, k, v, o, go, stats, gq, dsum, g);
            hipLaunchKernelGGL((k_attn_long_bwd_kv_mfma<32>), gk_, dim3(256), 0, st, q, k, v, go, stats, dsum, gk, gv, g);
        } else {
            hipLaunchKernelGGL((k_attn_long_bwd_q_mfma<64>), gq_, dim3(256), 0, st, q, k, v, o, go, stats, gq, dsum, g);
            hipLaunchKernelGGL((k_attn_long_bwd_kv_mfma<64>), gk_, dim3(256), 0, st, q, k, v, go, stats, dsum, gk, gv, g);
        }
        return launch_status("fqss_attn_long_bwd");
    }
    dim3 grid_q((unsigned)cdiv(Lq, 256), (unsigned)(B * nh)), grid_k((unsigned)cdiv(Lk, 256), (unsigned)(B * nh));
#define CALL(HD_)                                                                                                                       \
    hipLaunchKernelGGL((k_attn_long_bwd_q<HD_>), grid_q, dim3(256), 0, (hipStream_t)stream, q, k, v, o, go, stats, gq, dsum, g);         \
    hipLaunchKernelGGL((k_attn_long_bwd_kv<HD_>), grid_k, dim3(256), 0, (hipStream_t)stream, q, k, v, go, stats, dsum, gk, gv, g);
    FQSS_HD_SWITCH(hd, CALL)
#undef CALL
    return launch_status("fqss_attn_long_bwd");
}

// ---- coded operands (see k_attn_long_fwd_c): strides in ELEMENTS of each tensor (bytes for the code tensors)
static int check_coded(const void* const* codes, const int64_t* st, int hd) {
    FQSS_REQUIRE(hd == 16 || hd == 32 || hd == 64, "coded attention: head_dim 16, 32 or 64");
    for (int t = 0; t < 3; ++t) {
        FQSS_REQUIRE(((uintptr_t)codes[t] & 7) == 0 && st[2 * t] % 8 == 0 && st[2 * t + 1] % 8 == 0, "coded attention: code rows must be 8-B aligned");
    }
    return FQSS_OK;
}

// ranges: device scalars {q_min, q_max (the grid q is on: the division's quantizer), k_min, k_max, v_min, v_max}
// strides: (sl, sb) pairs for qc, kc, vc, o
extern "C" int fqss_attn_long_fwd_c(const uint8_t* qc, const uint8_t* kc, const uint8_t* vc, const float* const* ranges, float* o, float* stats,
                                    int Lq, int Lk, int B, int nh, int hd, const int64_t* strides, fqss_stream_t stream) {
    FQSS_REQUIRE(qc && kc && vc && ranges && o && stats && strides, "null tensor");
    for (int t = 0; t < 6; ++t) FQSS_REQUIRE(ranges[t], "null range");
    if (int rc = check_attn(Lq, Lk, B, nh, hd, strides, 4)) return rc;
    const void* codes[3] = {qc, kc, vc};
    if (int rc = check_coded(codes, strides, hd)) return rc;
    AttnGeom g{Lq, Lk, B, nh, {strides[0], strides[1]}, {strides[2], strides[3]}, {strides[4], strides[5]}, {strides[6], strides[7]}, {}, {}, {}, {}};
    const AttnRanges rg{ranges[0], ranges[1], ranges[2], ranges[3], ranges[4], ranges[5]};
    dim3 gm((unsigned)cdiv(Lq, 128), (unsigned)(B * nh));
    hipStream_t st = (hipStream_t)stream;
    if (hd == 16) hipLaunchKernelGGL((k_attn_long_fwd_c<16>), gm, dim3(256), 0, st, qc, kc, vc, o, stats, g, rg);
    else if (hd == 32) hipLaunchKernelGGL((k_attn_long_fwd_c<32>), gm, dim3(256), 0, st, qc, kc, vc, o, stats, g, rg);
    else hipLaunchKernelGGL((k_attn_long_fwd_c<64>), gm, dim3(256), 0, st, qc, kc, vc, o, stats, g, rg);
    return launch_status("fqss_attn_long_fwd_c");
}

// strides: (sl, sb) pairs for qc, kc, vc, o, go, gq, gk, gv;  dsum: workspace of B*nh*Lq floats.  gq / gk / gv: dL/dq, dL/dk, dL/dv
// with respect to the de-quantized VALUES (what fqss_mha_prep_bwd takes)
extern "C" int fqss_attn_long_bwd_c(const uint8_t* qc, const uint8_t* kc, const uint8_t* vc, const float* const* ranges, const float* o,
                                    const float* go, const float* stats, float* gq, float* gk, float* gv, float* dsum, int Lq, int Lk, int B,
                                    int nh, int hd, const int64_t* strides, fqss_stream_t stream) {
    FQSS_REQUIRE(qc && kc && vc && ranges && o && go && stats && gq && gk && gv && dsum && strides, "null tensor");
    for (int t = 0; t < 6; ++t) FQSS_REQUIRE(ranges[t], "null range");
    if (int rc = check_attn(Lq, Lk, B, nh, hd, strides, 8)) return rc;
    const void* codes[3] = {qc, kc, vc};
    if (int rc = check_coded(codes, strides, hd)) return rc;
    const int64_t* s = strides;
    FQSS_REQUIRE(aligned16(o) && aligned16(go) && s[6] % 4 == 0 && s[7] % 4 == 0 && s[8] % 4 == 0 && s[9] % 4 == 0, "o / go rows must be 16-B aligned");
    AttnGeom g{Lq, Lk, B, nh, {s[0], s[1]}, {s[2], s[3]}, {s[4], s[5]}, {s[6], s[7]}, {s[8], s[9]}, {s[10], s[11]}, {s[12], s[13]}, {s[14], s[15]}};
    const AttnRanges rg{ranges[0], ranges[1], ranges[2], ranges[3], ranges[4], ranges[5]};
    dim3 gq_((unsigned)cdiv(Lq, 128), (unsigned)(B * nh)), gk_((unsigned)cdiv(Lk, 128), (unsigned)(B * nh));
    hipStream_t st = (hipStream_t)stream;
#define FQSS_AL_BWD_C(HD_)                                                                                                          \
    hipLaunchKernelGGL((k_attn_long_bwd_q_c<HD_>), gq_, dim3(256), 0, st, qc, kc, vc, o, go, stats, gq, dsum, g, rg);               \
    hipLaunchKernelGGL((k_attn_long_bwd_kv_c<HD_>), gk_, dim3(256), 0, st, qc, kc, vc, go, stats, dsum, gk, gv, g, rg);
    if (hd == 16) { FQSS_AL_BWD_C(16) } else if (hd == 32) { FQSS_AL_BWD_C(32) } else { FQSS_AL_BWD_C(64) }
#undef FQSS_AL_BWD_C
    return launch_status("fqss_attn_long_bwd_c");
}
