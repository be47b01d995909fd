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
    dim3 grid_q((unsigned)cdiv(Lq, 256), (unsigned)(B * nh)), grid_k((unsigned)cdiv(Lk, 256), (unsigned)(B * nh));
#define CALL(HD_)                                                                                                                       \
    hipLaunchKernelGGL((k_attn_long_bwd_q<HD_>), grid_q, dim3(256), 0, (hipStream_t)stream, q, k, v, o, go, stats, gq, dsum, g);         \
    hipLaunchKernelGGL((k_attn_long_bwd_kv<HD_>), grid_k, dim3(256), 0, (hipStream_t)stream, q, k, v, go, stats, dsum, gk, gv, g);
    FQSS_HD_SWITCH(hd, CALL)
#undef CALL
    return launch_status("fqss_attn_long_bwd");
}
