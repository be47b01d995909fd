// dualpath.hip -- row-major ("channels-last") layers and data movement of the dual-path models (DPTNet, SURVEY.md §8
// row a13).  Inside the dual-path blocks every tensor is a row matrix [L][B'][C] (sequence-first, feature dim C
// contiguous: 64 floats = one 256-B line per (position, sequence)), so LayerNorm is one wavefront per row (C = 64 = one
// lane per feature, reductions by DPP/shuffle only), the linears are plain row GEMMs (csrc/gemm.hip fqss_rowlin_*) and the
// intra-/inter-chunk views differ by ONE transposing copy (fqss_permute4) instead of the reference's four
// permute().contiguous() round trips per block (dptnetq.py:156, 197-204).
//
// Reference replaced: F.layer_norm in LayerNormQ (qat_layers.py:455-465); torch.tanh / torch.sigmoid of Conv1dNlQ
// (dptnetq.py:286-287); q / sqrt(head_dim) (qat_layers.py:905); split_feature / merge_feature (dptnetq.py:232-276);
// overlap_and_add with a 2-tap frame (dptnetq.py:17-58, 140); bias gradients (column sums) of the row linears.
#include <stdlib.h>

#define FQSS_USES_GRAD_ADD   // the fp32 gradient atomics of this file go through grad_add (fqss_dev.h: FQSS_DETERMINISTIC=1)
#include "fqss_dev.h"

namespace fqss {

// ------------------------------------------------------------------------------------------------ LayerNorm rows
// One wavefront per row, JC = ceil(C / 64) features per lane.  Two-pass statistics in registers (the row is loaded once).
// Q: LayerNormQ's output quantizer in the same pass (qat_layers.py:455-465 + qat_quant.py:136-147): y receives fq(LN(x)), yc (nullable)
// its u8 codes; the pre-quant value is never stored -- the backward recomputes it from x, mean, rstd with the same operations.
// VEC (JC % 4 == 0, rows 16-B aligned): a lane owns 4 CONSECUTIVE features per 256-feature group and moves them as one float4 -- 1 KB per
// wave instruction instead of 256 B; the kernels are bound by the number of memory instructions in flight per row, not by bytes (an extra
// operand stream cost the 4-B form 17 us of 40).
template <int JC, bool VEC>
__device__ __forceinline__ int ln_col(int lane, int j) {
    return VEC ? ((j >> 2) * 256 + lane * 4 + (j & 3)) : (lane + 64 * j);
}
template <int JC, bool VEC>
__device__ __forceinline__ void ln_load(const float* __restrict__ row, int lane, int C, float (&v)[JC]) {
    if constexpr (VEC) {
#pragma unroll
        for (int jj = 0; jj < JC / 4; ++jj) {
            const int c = jj * 256 + lane * 4;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < C) t = *reinterpret_cast<const float4*>(row + c);
            v[4 * jj] = t.x; v[4 * jj + 1] = t.y; v[4 * jj + 2] = t.z; v[4 * jj + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < JC; ++j) {
            const int c = lane + 64 * j;
            v[j] = c < C ? row[c] : 0.f;
        }
    }
}
template <int JC, bool VEC>
__device__ __forceinline__ void ln_store(float* __restrict__ row, int lane, int C, const float (&v)[JC]) {
    if constexpr (VEC) {
#pragma unroll
        for (int jj = 0; jj < JC / 4; ++jj) {
            const int c = jj * 256 + lane * 4;
            if (c < C) *reinterpret_cast<float4*>(row + c) = make_float4(v[4 * jj], v[4 * jj + 1], v[4 * jj + 2], v[4 * jj + 3]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < JC; ++j) {
            const int c = lane + 64 * j;
            if (c < C) row[c] = v[j];
        }
    }
}

// G = lanes per row: 64, or 16 for rows of at most 64 features (the 64-wide dual-path blocks of DPTNet): a wave then normalises FOUR rows
// at once, 16 lanes x float4 each -- the one-row form left 3/4 of every memory instruction's lanes on 4-B accesses
template <int G>
__device__ __forceinline__ float ln_group_sum(float v) {
    if constexpr (G == 64) {
        return wave_sum(v);
    } else {
        static_assert(G == 16, "rows of 16 lanes (four rows per wave)");
        return row16_sum(v);
    }
}

// Row map of a LayerNorm's OUTPUT (forward: y and its codes; backward: where dL/dy of row r lives): row r = (i0 * d1 + i1) * d2 + i2 of
// the input goes to row i0 * t0 + i1 * t1 + i2 * t2 -- the intra- <-> inter-chunk layout change of the dual-path models
// ([K][B*S] <-> [S][B*K], dptnetq.py:313-327) done by the kernel that writes the rows anyway.  d2 == 0: identity.
struct RowMap {
    int64_t d1, d2, t0, t1, t2;
};
__device__ __forceinline__ int64_t map_row(const RowMap& m, int64_t r) {
    if (m.d2 == 0) return r;
    const int64_t q = r / m.d2, i2 = r - q * m.d2, i0 = q / m.d1, i1 = q - i0 * m.d1;
    return i0 * m.t0 + i1 * m.t1 + i2 * m.t2;
}

template <int JC, bool Q, bool VEC, int G = 64>
__global__ __launch_bounds__(256) void k_layernorm_fwd(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y,
                                                        float* __restrict__ mean_rstd, int64_t R, int C, int64_t ld_x,
                                                        int64_t ld_y, float eps, const float* __restrict__ qmin,
                                                        const float* __restrict__ qmax, unsigned char* __restrict__ yc, int64_t ld_yc,
                                                        const float* __restrict__ xadd, int64_t ld_a, float* __restrict__ xsum,
                                                        int64_t ld_s, const float* __restrict__ qs_min, const float* __restrict__ qs_max,
                                                        RowMap om) {
    // xadd / xsum (both or neither): the row that is normalised is x + xadd -- the residual add in front of a pre-norm transformer
    // sub-layer -- and the sum is also written out (it is the residual stream of the NEXT add and the backward's input)
    // qs_min / qs_max (nullable, with xadd): the add is an AddQ (the post-norm layers of DPTNet, dptnetq.py:84-97): xsum receives the
    // PRE-quant sum z = x + xadd (what the backward's STE needs) and the row that is normalised is fq_s(z)
    QRange qr{0.0f, 1.0f, 1.0f};
    if (Q) qr = load_qrange(qmin, qmax);
    QRange qs{0.0f, 1.0f, 1.0f};
    const bool QS = qs_min != nullptr;
    if (QS) qs = load_qrange(qs_min, qs_max);
    static_assert(G == 64 || (G == 16 && JC == 4 && VEC), "narrow rows: 16 lanes x float4");
    constexpr int RPW = 64 / G;                              // rows per wave
    const int lane = (threadIdx.x & 63) % G, rw = (threadIdx.x & 63) / G;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
    float ga[JC], be[JC];
    ln_load<JC, VEC>(gamma, lane, C, ga);
    ln_load<JC, VEC>(beta, lane, C, be);
    const float invC = 1.0f / (float)C;
    for (int64_t r0 = wave * RPW; r0 < R; r0 += nw * RPW) {
        const bool row_ok = r0 + rw < R;                     // (a wave's last group of rows may be short: those lanes re-read row R-1)
        const int64_t r = row_ok ? r0 + rw : R - 1;
        float v[JC], s = 0.f;
        ln_load<JC, VEC>(x + r * ld_x, lane, C, v);
        if (xadd != nullptr) {
            float a2[JC];
            ln_load<JC, VEC>(xadd + r * ld_a, lane, C, a2);
#pragma unroll
            for (int j = 0; j < JC; ++j) v[j] = v[j] + a2[j];
            if (row_ok) ln_store<JC, VEC>(xsum + r * ld_s, lane, C, v);
            if (QS) {
#pragma unroll
                for (int j = 0; j < JC; ++j) {
                    float c, u;
                    bool inr;
                    const float t = fq_asym(v[j], qs, c, u, inr);
                    v[j] = ln_col<JC, VEC>(lane, j) < C ? t : 0.f;      // (columns past C stay 0: they enter the row sums)
                }
            }
        }
#pragma unroll
        for (int j = 0; j < JC; ++j) s += v[j];
        const float mean = ln_group_sum<G>(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < JC; ++j) {
            const float d = ln_col<JC, VEC>(lane, j) < C ? v[j] - mean : 0.f;
            q += d * d;
        }
        const float var = ln_group_sum<G>(q) * invC;
        const float rstd = 1.0f / sqrtf(var + eps);
        float o[JC];
        unsigned char oc[JC];
#pragma unroll
        for (int j = 0; j < JC; ++j) {
            const float z = ((v[j] - mean) * rstd) * ga[j] + be[j];
            if (Q) {
                const float code = fq_code(z, qr);
                o[j] = qr.delta * code + qr.lo;
                oc[j] = (unsigned char)code;
            } else {
                o[j] = z;
            }
        }
        const int64_t ro = map_row(om, r);                   // (y and its codes may leave in another row order: RowMap)
        if (row_ok) ln_store<JC, VEC>(y + ro * ld_y, lane, C, o);
        if (Q && yc != nullptr && row_ok) {
            if (VEC && (ld_yc & 3) == 0) {      // four codes per lane and group: one 4-B store
#pragma unroll
                for (int jj = 0; jj < JC / 4; ++jj) {
                    const int c = jj * 256 + lane * 4;
                    if (c < C)
                        *reinterpret_cast<unsigned int*>(yc + ro * ld_yc + c) = (unsigned int)oc[4 * jj] | ((unsigned int)oc[4 * jj + 1] << 8) |
                                                                               ((unsigned int)oc[4 * jj + 2] << 16) | ((unsigned int)oc[4 * jj + 3] << 24);
                }
            } else {
#pragma unroll
                for (int j = 0; j < JC; ++j) {
                    const int c = ln_col<JC, VEC>(lane, j);
                    if (c < C) yc[ro * ld_yc + c] = oc[j];
                }
            }
        }
        if (lane == 0 && row_ok) {
            mean_rstd[2 * r] = mean;
            mean_rstd[2 * r + 1] = rstd;
        }
    }
}

// gx = rstd * (dxh - mean(dxh) - xh * mean(dxh * xh)), dxh = gy * gamma; per-workgroup partial column sums for the
// affine gradients (registers across the rows of a wave, LDS across the 4 waves, then one atomic per column and workgroup)
// Q: gy is dL/d fq(LN(x)): the quantizer's STE (and its range-gradient partials, one gacc slot per workgroup like k_actq_bwd) runs on
// the pre-quant value recomputed from x -- no separate fqss_actq_bwd pass, no stored z.
template <int JC, bool Q, bool VEC, int G = 64>
__global__ __launch_bounds__(256) void k_layernorm_bwd(const float* __restrict__ gy, const float* __restrict__ x,
                                                        const float* __restrict__ gamma, const float* __restrict__ mean_rstd,
                                                        float* __restrict__ gx, float* __restrict__ ggamma,
                                                        float* __restrict__ gbeta, int64_t R, int C, int64_t ld_gy,
                                                        int64_t ld_x, int64_t ld_gx, const float* __restrict__ beta,
                                                        const float* __restrict__ qmin, const float* __restrict__ qmax, double* gacc,
                                                        const float* __restrict__ gadd, int64_t ld_ga, const float* __restrict__ qs_min,
                                                        const float* __restrict__ qs_max, double* gacc_s, RowMap gm) {
    // gadd (nullable): the gradient arriving on the residual stream behind the fused add (k_layernorm_fwd's xadd form): gx = LN' + gadd is
    // then the gradient of BOTH addends -- the sum autograd would take at the fork in a pass of its own
    // qs_min / qs_max / gacc_s (nullable): the add was an AddQ: x holds the PRE-quant sum z, the normalised row is fq_s(z) and gx passes
    // through that quantizer's STE (range partials to gacc_s, one slot per workgroup)
    __shared__ float red[2][4][64 * JC];
    __shared__ double redq[2 * 4];
    static_assert(G == 64 || (G == 16 && JC == 4 && VEC), "narrow rows: 16 lanes x float4");
    constexpr int RPW = 64 / G;
    const int lane = (threadIdx.x & 63) % G, rw = (threadIdx.x & 63) / G, w = threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * 4 + w, nw = (int64_t)gridDim.x * 4;
    QRange qr{0.0f, 1.0f, 1.0f};
    if (Q) qr = load_qrange(qmin, qmax);
    float p_du = 0.0f, p_out = 0.0f;      // sum g*(c - m*u), sum g*(1-m)   (k_actq_bwd)
    float s_du = 0.0f, s_out = 0.0f;      // the same for the sum quantizer
    QRange qs{0.0f, 1.0f, 1.0f};
    const bool QS = qs_min != nullptr;
    if (QS) qs = load_qrange(qs_min, qs_max);
    float ga[JC], be[JC], agg[JC], agb[JC];
    ln_load<JC, VEC>(gamma, lane, C, ga);
#pragma unroll
    for (int j = 0; j < JC; ++j) be[j] = agg[j] = agb[j] = 0.f;
    if (Q) ln_load<JC, VEC>(beta, lane, C, be);
    const float invC = 1.0f / (float)C;
    for (int64_t r0 = wave * RPW; r0 < R; r0 += nw * RPW) {
        const bool row_ok = r0 + rw < R;
        const int64_t r = row_ok ? r0 + rw : R - 1;
        const float mean = mean_rstd[2 * r], rstd = mean_rstd[2 * r + 1];
        float gv[JC], xv[JC], av[JC], xh[JC], dxh[JC], a = 0.f, b = 0.f;
        ln_load<JC, VEC>(gy + map_row(gm, r) * ld_gy, lane, C, gv);      // (the forward wrote y through the same RowMap)
        if (RPW > 1 && !row_ok) {
#pragma unroll
            for (int jz = 0; jz < JC; ++jz) gv[jz] = 0.f;          // no contribution to the column sums / range partials
        }
        ln_load<JC, VEC>(x + r * ld_x, lane, C, xv);
        if (gadd != nullptr) ln_load<JC, VEC>(gadd + r * ld_ga, lane, C, av);
        float scu[JC];            // sum quantizer: c - u (in range) or c (clamped); NaN-free marker for "clamped" is s_in[j]
        bool s_in[JC];
#pragma unroll
        for (int j = 0; j < JC; ++j) {
            scu[j] = 0.f;
            s_in[j] = true;
            if (QS) {
                float c, u;
                bool inr;
                xv[j] = fq_asym(xv[j], qs, c, u, inr);      // the row the forward normalised
                s_in[j] = inr;
                scu[j] = inr ? (c - u) : c;
            }
        }
#pragma unroll
        for (int j = 0; j < JC; ++j) {
            const bool live = ln_col<JC, VEC>(lane, j) < C;
            float g = gv[j];
            xh[j] = live ? (xv[j] - mean) * rstd : 0.f;
            if (Q) {
                float code, u;
                bool inr;
                (void)fq_asym(xh[j] * ga[j] + be[j], qr, code, u, inr);      // the forward's z, operation for operation
                p_du += live ? g * (inr ? (code - u) : code) : 0.0f;
                p_out += (live && !inr) ? g : 0.0f;
                g = inr ? div_by(g * qr.delta, qr.delta, qr.inv) : 0.0f;
            }
            dxh[j] = g * ga[j];
            a += dxh[j];
            b += dxh[j] * xh[j];
            agg[j] += g * xh[j];
            agb[j] += g;
        }
        a = ln_group_sum<G>(a) * invC;
        b = ln_group_sum<G>(b) * invC;
        float o[JC];
#pragma unroll
        for (int j = 0; j < JC; ++j) {
            o[j] = rstd * ((dxh[j] - a) - xh[j] * b);
            if (gadd != nullptr) o[j] = o[j] + av[j];
            if (QS) {
                const bool live = row_ok && ln_col<JC, VEC>(lane, j) < C;
                const float gq = o[j];
                s_du += live ? gq * scu[j] : 0.0f;
                s_out += (live && !s_in[j]) ? gq : 0.0f;
                o[j] = s_in[j] ? div_by(gq * qs.delta, qs.delta, qs.inv) : 0.0f;
            }
        }
        if (row_ok) ln_store<JC, VEC>(gx + r * ld_gx, lane, C, o);
    }
    if constexpr (RPW > 1) {      // the row groups of a wave hold partial sums of the same columns
#pragma unroll
        for (int j = 0; j < JC; ++j)
#pragma unroll
            for (int o = G; o < 64; o <<= 1) {
                agg[j] += __shfl_xor(agg[j], o, 64);
                agb[j] += __shfl_xor(agb[j], o, 64);
            }
    }
    if (rw == 0) {
#pragma unroll
        for (int j = 0; j < JC; ++j) {
            const int c = ln_col<JC, VEC>(lane, j);       // < 64 * JC
            red[0][w][c] = agg[j];
            red[1][w][c] = agb[j];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * 64 * JC; e += 256) {
        const int which = e / (64 * JC), c = e % (64 * JC);
        if (c < C && c < G * JC) {
            const float s = (red[which][0][c] + red[which][1][c]) + (red[which][2][c] + red[which][3][c]);
            grad_add((which == 0 ? ggamma : gbeta) + c, s);
        }
    }
    if (Q) {
        double v[2] = {(double)p_du, (double)p_out};
        block_sum<double, 2>(v, redq);
        if (threadIdx.x == 0) {
            double* slot = gacc + 3 * (int64_t)blockIdx.x;
            const double dmax = v[0] / 255.0;
            slot[0] += v[1] - dmax;
            slot[1] += dmax;
        }
    }
    if (QS) {
        __syncthreads();
        double v[2] = {(double)s_du, (double)s_out};
        block_sum<double, 2>(v, redq);
        if (threadIdx.x == 0) {
            double* slot = gacc_s + 3 * (int64_t)blockIdx.x;
            const double dmax = v[0] / 255.0;
            slot[0] += v[1] - dmax;
            slot[1] += dmax;
        }
    }
}

// ------------------------------------------------------------------------------------------------ column sums
// out[c] += sum_r g[r][c].  The 256 threads of a workgroup form RG = 256 / CW row groups of CW columns each (CW = the column
// count rounded up to a power of two, at most 256): every lane is busy also for the 64-wide tensors of the dual-path blocks, a
// row group streams whole rows (coalesced), the RG partial sums of a column meet in LDS, one atomic per column and workgroup.
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ g, float* __restrict__ out, int64_t R, int C,
                                                 int64_t ld, int64_t rows_per_block, int CW) {
    __shared__ float red[256];
    const int RG = 256 / CW;
    const int cl = threadIdx.x % CW, rg = threadIdx.x / CW;
    const int c = blockIdx.x * CW + cl;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = min(R, r0 + rows_per_block);
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
        int64_t r = r0 + rg;
        for (; r + RG < r1; r += 2 * RG) {
            s0 += g[r * ld + c];
            s1 += g[(r + RG) * ld + c];
        }
        if (r < r1) s0 += g[r * ld + c];
    }
    red[threadIdx.x] = s0 + s1;
    __syncthreads();
    if (rg == 0 && c < C) {
        float s = red[cl];
        for (int k = 1; k < RG; ++k) s += red[k * CW + cl];
        grad_add(out + c, s);
    }
}

// ------------------------------------------------------------------------------------------------ unary maps
// kind 0: tanh   1: sigmoid = 1 / (1 + exp(-x))   2: x / p (IEEE division: q / sqrt(head_dim))
// kind 3: GELU (erf form, torch's default): 0.5 x (1 + erf(x / sqrt 2))      [HTDemucs layers, SURVEY §8 row a15]
// kinds 4-7: the value maps of the public STE helpers of qat_quant.py:88-107 (round / floor / sign / clip(p, p2)); their backward is
// the identity (times a scale), so they have no k_unary_bwd case
__global__ __launch_bounds__(256) void k_unary_fwd(const float* __restrict__ x, float* __restrict__ y, int64_t n, int kind,
                                                    float p, float p2) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = x[i];
        float r;
        if (kind == 0) r = tanhf(v);
        else if (kind == 1) r = 1.0f / (1.0f + expf(-v));
        else if (kind == 2) r = v / p;
        else if (kind == 3) r = (0.5f * v) * (1.0f + erff(v * 0.70710678118654752440f));
        else if (kind == 4) r = rintf(v);                                        // torch.round: half to even (round_ste)
        else if (kind == 5) r = floorf(v);                                       // floor_ste
        else if (kind == 6) r = (v > 0.0f) ? 1.0f : ((v < 0.0f) ? -1.0f : 0.0f); // torch.sign (grad_sign)
        else r = fminf(fmaxf(v, p), p2);                                         // torch.clip(x, p, p2) (clip_ste)
        y[i] = r;
    }
}
// the same maps on a ROW-STRIDED input (a column block of a wider matrix: the q third of an attention in-projection [R][3E]), dense output:
// the value read in place instead of through a `.contiguous()` copy (round 6: 32 such copies per Sepformer step).  cols % 4 == 0,
// 16-B aligned rows; grid (column chunks, rows)
__global__ __launch_bounds__(256) void k_unary_rows_fwd(const float* __restrict__ x, float* __restrict__ y, int64_t rows, int cols, int64_t ld_x,
                                                         int64_t ld_y, int kind, float p) {
    // a thread per (row, 4-column group), rows and groups flattened: narrow blocks (64 columns of an attention head block: 16 groups)
    // keep every lane busy (one workgroup per row left 16 of 256 lanes with work)
    const int groups = cols >> 2;
    const int64_t total = rows * groups;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / groups;
        const int c = (int)(i - r * groups) * 4;
        const float4 v4 = *reinterpret_cast<const float4*>(x + r * ld_x + c);
        const float v[4] = {v4.x, v4.y, v4.z, v4.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (kind == 0) o[j] = tanhf(v[j]);
            else if (kind == 1) o[j] = 1.0f / (1.0f + expf(-v[j]));
            else if (kind == 2) o[j] = v[j] / p;
            else o[j] = (0.5f * v[j]) * (1.0f + erff(v[j] * 0.70710678118654752440f));
        }
        *reinterpret_cast<float4*>(y + r * ld_y + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
}
// ATen: tanh_backward = g * (1 - y*y); sigmoid_backward = g * (1 - y) * y; div by a scalar: g / p
__global__ __launch_bounds__(256) void k_unary_bwd(const float* __restrict__ g, const float* __restrict__ y,
                                                    float* __restrict__ gx, int64_t n, int kind, float p) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gv = g[i];
        float r;
        if (kind == 0) { const float t = y[i]; r = gv * (1.0f - t * t); }
        else if (kind == 1) { const float t = y[i]; r = (gv * (1.0f - t)) * t; }
        else if (kind == 2) r = gv / p;
        else {   // GELU: `y` holds the INPUT x; d/dx = Phi(x) + x phi(x)
            const float x = y[i];
            const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
            const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
            r = gv * (cdf + x * pdf);
        }
        gx[i] = r;
    }
}

// ------------------------------------------------------------------------------------------------ transposing copy
// y[i0][i1][i2][c] = x[i0*s0 + i1*s1 + i2*s2 + c], c < C contiguous on both sides
__global__ __launch_bounds__(256) void k_permute4(const float* __restrict__ x, float* __restrict__ y, int64_t n0, int64_t n1,
                                                   int64_t n2, int C, int64_t s0, int64_t s1, int64_t s2) {
    const int64_t total = n0 * n1 * n2 * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int64_t i2 = t % n2; t /= n2;
        const int64_t i1 = t % n1;
        const int64_t i0 = t / n1;
        y[i] = x[i0 * s0 + i1 * s1 + i2 * s2 + c];
    }
}

// ... with the output rows `ld_y` floats apart (ld_y % 4 == 0, y 16-B aligned: a row-padded activation, so that everything behind the
// move -- and every gradient coming back -- has 16-B aligned rows): a thread moves 4 consecutive c with one 16-B store; the loads are
// one (possibly unaligned) 16-B request where the whole group lies inside the row.  Padding columns c >= C receive zeros.
struct __attribute__((packed, aligned(4))) P4U { float x, y, z, w; };
__global__ __launch_bounds__(256) void k_permute4_ld(const float* __restrict__ x, float* __restrict__ y, int64_t rows, int n1, int n2, int C,
                                                      int64_t s0, int64_t s1, int64_t s2, int64_t ld_y) {
    const int groups = (int)(ld_y >> 2);
    const int64_t total = rows * groups;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / groups;
        const int c = (int)(i - r * groups) * 4;
        const int64_t t = r / n2;
        const int i2 = (int)(r - t * n2);
        const int64_t i0 = t / n1;
        const int i1 = (int)(t - i0 * n1);
        const float* xr = x + i0 * s0 + (int64_t)i1 * s1 + (int64_t)i2 * s2;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c + 3 < C) {
            const P4U u = *reinterpret_cast<const P4U*>(xr + c);
            v = make_float4(u.x, u.y, u.z, u.w);
        } else {
            if (c < C) v.x = xr[c];
            if (c + 1 < C) v.y = xr[c + 1];
            if (c + 2 < C) v.z = xr[c + 2];
        }
        *reinterpret_cast<float4*>(y + r * ld_y + c) = v;
    }
}

// ------------------------------------------------------------------------------------------------ chunking
// split_feature (dptnetq.py:247-259) fused with the move to the intra-chunk row layout:
//   seg[k][b*S + s][n] = fpad[b][n][(s>>1)*K + (s&1)*P + k],  fpad = P zeros | f[b][n][0..T) | rest zeros | P zeros
__global__ __launch_bounds__(256) void k_dp_segment_fwd(const float* __restrict__ f, float* __restrict__ seg, int B, int N,
                                                         int64_t T, int64_t ld_f, int K, int S) {
    const int P = K / 2;
    const int64_t total = (int64_t)K * B * S * N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int n = (int)(i % N);
        int64_t t = i / N;
        const int s = (int)(t % S); t /= S;
        const int b = (int)(t % B);
        const int k = (int)(t / B);
        const int64_t j = (int64_t)(s >> 1) * K + (s & 1) * P + k - P;   // position in f
        seg[i] = (j >= 0 && j < T) ? f[((int64_t)b * N + n) * ld_f + j] : 0.f;
    }
}
// gf[b][n][t] = sum of the (at most two) chunk slots that hold position t
__global__ __launch_bounds__(256) void k_dp_segment_bwd(const float* __restrict__ gseg, float* __restrict__ gf, int B, int N,
                                                         int64_t T, int64_t ld_gf, int K, int S) {
    const int P = K / 2;
    const int64_t total = (int64_t)B * N * T;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t t = i % T;
        const int64_t bn = i / T;
        const int n = (int)(bn % N), b = (int)(bn / N);
        const int64_t j = t + P;                    // position in fpad
        float v = 0.f;
        {   // even chunk s = 2a covers fpad[a*K, a*K + K)
            const int64_t a = j / K;
            const int k = (int)(j - a * K);
            if (2 * a < S) v += gseg[(((int64_t)k * B + b) * S + 2 * a) * N + n];
        }
        {   // odd chunk s = 2a+1 covers fpad[P + a*K, P + a*K + K)
            const int64_t a = (j - P) / K;
            const int k = (int)((j - P) - a * K);
            if (2 * a + 1 < S) v += gseg[(((int64_t)k * B + b) * S + 2 * a + 1) * N + n];
        }
        gf[bn * ld_gf + t] = v;
    }
}

// merge_feature (dptnetq.py:261-276) up to its AddQ: gathers the two half-overlapped streams from the inter-chunk row
// layout o[s][b*K + k][spk*N + n] into channel-first a, b [B*nspk][N][Lm], Lm = (S/2)*K - P:
//   a[t] = o[s = 2*((t+P)/K)    ][k = (t+P)%K]      b[t] = o[s = 2*(t/K) + 1][k = t%K]
__global__ __launch_bounds__(256) void k_dp_merge_fwd(const float* __restrict__ o, float* __restrict__ a, float* __restrict__ b,
                                                       int B, int nspk, int N, int K, int S, int64_t Lm, int64_t ld_ab) {
    const int P = K / 2, C2 = nspk * N;
    const int64_t total = (int64_t)B * nspk * N * Lm;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t t = i % Lm;
        int64_t r = i / Lm;
        const int n = (int)(r % N); r /= N;
        const int spk = (int)(r % nspk);
        const int bb = (int)(r / nspk);
        const int64_t row = ((int64_t)bb * nspk + spk) * N + n;
        const int c = spk * N + n;
        {
            const int64_t q = (t + P) / K;
            const int k = (int)((t + P) - q * K);
            a[row * ld_ab + t] = o[((2 * q) * ((int64_t)B * K) + (int64_t)bb * K + k) * C2 + c];
        }
        {
            const int64_t q = t / K;
            const int k = (int)(t - q * K);
            b[row * ld_ab + t] = o[((2 * q + 1) * ((int64_t)B * K) + (int64_t)bb * K + k) * C2 + c];
        }
    }
}
// go[s][b*K+k][c]: even s from ga (zero for the first P positions of chunk 0), odd s from gb (zero for the last P)
__global__ __launch_bounds__(256) void k_dp_merge_bwd(const float* __restrict__ ga, const float* __restrict__ gb,
                                                       float* __restrict__ go, int B, int nspk, int N, int K, int S,
                                                       int64_t Lm, int64_t ld_ga, int64_t ld_gb) {
    const int P = K / 2, C2 = nspk * N;
    const int64_t total = (int64_t)S * B * K * C2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C2);
        int64_t r = i / C2;
        const int k = (int)(r % K); r /= K;
        const int bb = (int)(r % B);
        const int s = (int)(r / B);
        const int spk = c / N, n = c - spk * N;
        const int64_t row = ((int64_t)bb * nspk + spk) * N + n;
        float v = 0.f;
        if ((s & 1) == 0) {
            const int64_t t = (int64_t)(s >> 1) * K + k - P;
            if (t >= 0 && t < Lm) v = ga[row * ld_ga + t];
        } else {
            const int64_t t = (int64_t)(s >> 1) * K + k;
            if (t < Lm) v = gb[row * ld_gb + t];
        }
        go[i] = v;
    }
}

// overlap_and_add of 2-sample frames with hop 1 (dptnetq.py:140 with W = 2): y [N][2][L] -> out [N][L+1]
__global__ __launch_bounds__(256) void k_ola2_fwd(const float* __restrict__ y, float* __restrict__ out, int64_t N, int64_t L,
                                                   int64_t ld_y) {
    const int64_t total = N * (L + 1);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t t = i % (L + 1), n = i / (L + 1);
        const float a = t < L ? y[(2 * n) * ld_y + t] : 0.f;
        const float b = t >= 1 ? y[(2 * n + 1) * ld_y + t - 1] : 0.f;
        out[i] = a + b;
    }
}
__global__ __launch_bounds__(256) void k_ola2_bwd(const float* __restrict__ g, float* __restrict__ gy, int64_t N, int64_t L,
                                                   int64_t ld_gy) {
    const int64_t total = N * 2 * L;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t t = i % L, r = i / L;       // r = 2n + tap
        const int64_t n = r >> 1, tap = r & 1;
        gy[r * ld_gy + t] = g[n * (L + 1) + t + tap];
    }
}


// ------------------------------------------------------------------------------------------------ gLN on row matrices
// GroupNorm(1, C) of the Sepformer dual-path blocks (sepformerq.py:141-142, 159, 175) on the row layouts: the statistics run
// over ALL rows of a sample; the sample of row r is b = (r % RB) / X  (intra-chunk rows [K][B*S][C]: RB = B*S, X = S;
// inter-chunk rows [S][B*K][C]: RB = B*K, X = K).  Pass 1 reduces per-sample (sum, sum^2) in fp64 (one wavefront per row,
// per-workgroup LDS slots, one fp64 atomic per sample and workgroup); pass 2 applies.  kMaxB samples per launch.
constexpr int kMaxB = 16;

__global__ __launch_bounds__(256) void k_gnrows_stats(const float* __restrict__ x, double* __restrict__ ws, int64_t R, int C,
                                                       int64_t ld, int RB, int X, int B) {
    __shared__ double acc[kMaxB][2];
    if (threadIdx.x < 2 * kMaxB) acc[threadIdx.x >> 1][threadIdx.x & 1] = 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
    for (int64_t r = wave; r < R; r += nw) {
        float s = 0.f, q = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float v = x[r * ld + c];
            s += v;
            q += v * v;
        }
        double sd = wave_sum((double)s), qd = wave_sum((double)q);
        if (lane == 0) {
            const int b = (int)((r % RB) / X);
            atomicAdd(&acc[b][0], sd);
            atomicAdd(&acc[b][1], qd);
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * B) atomicAdd(&ws[threadIdx.x], acc[threadIdx.x >> 1][threadIdx.x & 1]);
}

// mean_rstd[b] = (mean, 1/sqrt(var + eps)) from the fp64 sums; var = E[x^2] - mean^2 in fp64
__global__ void k_gnrows_finalize(const double* __restrict__ ws, float* __restrict__ mean_rstd, int B, double n, float eps) {
    const int b = threadIdx.x;
    if (b < B) {
        const double m = ws[2 * b] / n;
        double var = ws[2 * b + 1] / n - m * m;
        if (var < 0.0) var = 0.0;
        mean_rstd[2 * b] = (float)m;
        mean_rstd[2 * b + 1] = 1.0f / sqrtf((float)var + eps);
    }
}

__global__ __launch_bounds__(256) void k_gnrows_apply(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ mean_rstd,
                                                       float* __restrict__ y, int64_t R, int C, int64_t ld_x, int64_t ld_y, int RB,
                                                       int X) {
    const int64_t total = R * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / C;
        const int c = (int)(i - r * C);
        const int b = (int)((r % RB) / X);
        const float mean = mean_rstd[2 * b], rstd = mean_rstd[2 * b + 1];
        y[r * ld_y + c] = ((x[r * ld_x + c] - mean) * rstd) * gamma[c] + beta[c];
    }
}

// backward pass 1: per-sample a = sum gy*gamma, b = sum gy*gamma*xhat (fp64 slots) and the per-channel affine gradients
__global__ __launch_bounds__(256) void k_gnrows_bwd_reduce(const float* __restrict__ gy, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean_rstd,
                                                            double* __restrict__ ws, float* __restrict__ ggamma,
                                                            float* __restrict__ gbeta, int64_t R, int C, int64_t ld_gy, int64_t ld_x,
                                                            int RB, int X, int B, int64_t rows_per_block) {
    __shared__ double acc[kMaxB][2];
    if (threadIdx.x < 2 * kMaxB) acc[threadIdx.x >> 1][threadIdx.x & 1] = 0.0;
    __syncthreads();
    // thread <-> channel(s); the workgroup walks a slab of rows, so the channel sums stay in registers (C <= 1024)
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(R, r0 + rows_per_block);
    float gg[4] = {0.f, 0.f, 0.f, 0.f}, gb[4] = {0.f, 0.f, 0.f, 0.f}, gam[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int c = threadIdx.x + 256 * j; gam[j] = c < C ? gamma[c] : 0.f; }
    for (int64_t r = r0; r < r1; ++r) {
        const int b = (int)((r % RB) / X);
        const float mean = mean_rstd[2 * b], rstd = mean_rstd[2 * b + 1];
        float a = 0.f, bb = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = threadIdx.x + 256 * j;
            if (c < C) {
                const float g = gy[r * ld_gy + c];
                const float xh = (x[r * ld_x + c] - mean) * rstd;
                const float d = g * gam[j];
                a += d;
                bb += d * xh;
                gg[j] += g * xh;
                gb[j] += g;
            }
        }
        double ad = wave_sum((double)a), bd = wave_sum((double)bb);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&acc[b][0], ad);
            atomicAdd(&acc[b][1], bd);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = threadIdx.x + 256 * j;
        if (c < C) {
            grad_add(ggamma + c, gg[j]);
            grad_add(gbeta + c, gb[j]);
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * B) atomicAdd(&ws[threadIdx.x], acc[threadIdx.x >> 1][threadIdx.x & 1]);
}

__global__ __launch_bounds__(256) void k_gnrows_bwd_apply(const float* __restrict__ gy, const float* __restrict__ x,
                                                           const float* __restrict__ gamma, const float* __restrict__ mean_rstd,
                                                           const double* __restrict__ ws, float* __restrict__ gx, int64_t R, int C,
                                                           int64_t ld_gy, int64_t ld_x, int64_t ld_gx, int RB, int X, double n) {
    const int64_t total = R * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / C;
        const int c = (int)(i - r * C);
        const int b = (int)((r % RB) / X);
        const float mean = mean_rstd[2 * b], rstd = mean_rstd[2 * b + 1];
        const float a = (float)(ws[2 * b] / n), bb = (float)(ws[2 * b + 1] / n);
        const float xh = (x[r * ld_x + c] - mean) * rstd;
        gx[r * ld_gx + c] = rstd * ((gy[r * ld_gy + c] * gamma[c] - a) - xh * bb);
    }
}

// ------------------------------------------------------------------------------------------------ positional encoding
// z[l][b][c] = x[l][b][c] + p[l][c]  (TransformerBlock.pos_add on sequence-first rows, sepformerq.py:117-118) and the batch
// reduction its backward needs: out[l][c] = sum_b g[l][b][c]
__global__ __launch_bounds__(256) void k_bcast_add(const float* __restrict__ x, const float* __restrict__ p, float* __restrict__ z,
                                                    int64_t L, int64_t Bp, int C) {
    const int64_t total = L * Bp * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int64_t l = i / ((int64_t)Bp * C);
        z[i] = x[i] + p[l * C + c];
    }
}
__global__ __launch_bounds__(256) void k_bcast_sum(const float* __restrict__ g, float* __restrict__ out, int64_t L, int64_t Bp,
                                                    int C) {
    const int64_t total = L * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int64_t l = i / C;
        const float* gp = g + l * Bp * C + c;
        float s0 = 0.f, s1 = 0.f;
        int64_t b = 0;
        for (; b + 1 < Bp; b += 2) {
            s0 += gp[b * C];
            s1 += gp[(b + 1) * C];
        }
        if (b < Bp) s0 += gp[b * C];
        out[i] = s0 + s1;
    }
}


// ------------------------------------------------------------------------------------------------ GLU / divide / embedding
// nn.GLU(dim=1) on channel-first tensors (HEncLayer / HDecLayer / DConv `rewrite` convs, hdemucsq.py:127, 314, demucsq.py:168):
// x [B][2C][M] -> y [B][C][M] = a * sigmoid(b), a = first C channels, b = last C
__global__ __launch_bounds__(256) void k_glu_fwd(const float* __restrict__ x, float* __restrict__ y, int64_t B, int64_t C, int64_t M,
                                                  int64_t ld_x, int64_t ld_y) {
    const int64_t total = B * C * M;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i % M, bc = i / M, c = bc % C, b = bc / C;
        const float a = x[(b * 2 * C + c) * ld_x + m], g = x[(b * 2 * C + C + c) * ld_x + m];
        y[(b * C + c) * ld_y + m] = a * (1.0f / (1.0f + expf(-g)));
    }
}
__global__ __launch_bounds__(256) void k_glu_bwd(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ gx,
                                                  int64_t B, int64_t C, int64_t M, int64_t ld_x, int64_t ld_gy, int64_t ld_gx) {
    const int64_t total = B * C * M;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i % M, bc = i / M, c = bc % C, b = bc / C;
        const float a = x[(b * 2 * C + c) * ld_x + m], g = x[(b * 2 * C + C + c) * ld_x + m];
        const float s = 1.0f / (1.0f + expf(-g)), gv = gy[(b * C + c) * ld_gy + m];
        gx[(b * 2 * C + c) * ld_gx + m] = gv * s;
        gx[(b * 2 * C + C + c) * ld_gx + m] = ((gv * a) * (1.0f - s)) * s;
    }
}
// torch.div(x1, x2) element-wise (DivQ) and its gradients g / x2, -g x1 / x2^2 (ATen: -grad * self / (other * other))
__global__ __launch_bounds__(256) void k_div_fwd(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = a[i] / b[i];
}
__global__ __launch_bounds__(256) void k_div_bwd(const float* __restrict__ g, const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ ga, float* __restrict__ gb, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float bv = b[i], gv = g[i];
        ga[i] = gv / bv;
        gb[i] = (-gv * a[i]) / (bv * bv);
    }
}
// F.embedding: out[i][:] = w[idx[i]][:]; gw[idx[i]][:] += g[i][:]
__global__ __launch_bounds__(256) void k_embedding_fwd(const float* __restrict__ w, const int64_t* __restrict__ idx, float* __restrict__ out,
                                                        int64_t n, int D, int64_t V) {
    const int64_t total = n * D;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / D;
        const int d = (int)(i - r * D);
        const int64_t v = idx[r];
        out[i] = (v >= 0 && v < V) ? w[v * D + d] : 0.f;
    }
}
__global__ __launch_bounds__(256) void k_embedding_bwd(const float* __restrict__ g, const int64_t* __restrict__ idx, float* __restrict__ gw,
                                                        int64_t n, int D, int64_t V) {
    const int64_t total = n * D;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / D;
        const int d = (int)(i - r * D);
        const int64_t v = idx[r];
        if (v >= 0 && v < V) grad_add(gw + v * D + d, g[i]);
    }
}

static inline unsigned flat_grid(int64_t n) {
    int64_t nb = cdiv(n, 256);
    if (nb < 1) nb = 1;
    if (nb > 16384) nb = 16384;
    return (unsigned)nb;
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Attention-core prologue of MultiheadAttentionQ in the quantizing phase as ONE pass each way (qat_layers.py:890-905): the in-projection
// X [R][3E] is cut in thirds, each third goes through its own quantizer (q / k / v), the query is divided by sqrt(head_dim) and goes
// through the `div` quantizer.  Replaces three narrow fqss_actq_fwd + fqss_unary_fwd + fqss_actq_fwd (and their five backward
// launches): same op sequence per element -- fq_asym, one IEEE division, fq_asym -- so q / k / v are bit-identical to the un-fused chain.
// The backward recomputes the intermediate values from X: nothing but X is saved.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mha_prep_fwd(const float* __restrict__ X, float* __restrict__ q, float* __restrict__ k,
                                                       float* __restrict__ v, int64_t R, int E, int64_t ld_x, float scale,
                                                       const float* qmin_q, const float* qmax_q, const float* qmin_k, const float* qmax_k,
                                                       const float* qmin_v, const float* qmax_v, const float* qmin_d, const float* qmax_d) {
    const QRange rq = load_qrange(qmin_q, qmax_q), rk = load_qrange(qmin_k, qmax_k), rv = load_qrange(qmin_v, qmax_v),
                 rd = load_qrange(qmin_d, qmax_d);
    const int e4 = E >> 2;
    const int64_t n4 = R * e4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / e4;
        const int f = (int)(i - row * e4) * 4;
        const float* xr = X + row * ld_x + f;
        const float4 a = *reinterpret_cast<const float4*>(xr), b = *reinterpret_cast<const float4*>(xr + E),
                     c = *reinterpret_cast<const float4*>(xr + 2 * E);
        const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w}, cv[4] = {c.x, c.y, c.z, c.w};
        float oq[4], ok[4], ov[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float cc, u;
            bool inr;
            const float qp = fq_asym(av[j], rq, cc, u, inr);
            oq[j] = fq_asym(qp / scale, rd, cc, u, inr);
            ok[j] = fq_asym(bv[j], rk, cc, u, inr);
            ov[j] = fq_asym(cv[j], rv, cc, u, inr);
        }
        const int64_t o = row * E + f;
        *reinterpret_cast<float4*>(q + o) = make_float4(oq[0], oq[1], oq[2], oq[3]);
        *reinterpret_cast<float4*>(k + o) = make_float4(ok[0], ok[1], ok[2], ok[3]);
        *reinterpret_cast<float4*>(v + o) = make_float4(ov[0], ov[1], ov[2], ov[3]);
    }
}

// the same quantizer chain, emitting the 8-bit CODES of q (on the division quantizer's grid), k and v: what the coded attention kernels
// (csrc/attn_long.hip, fqss_attn_long_fwd_c) consume -- 3 B per feature instead of 12
__global__ __launch_bounds__(256) void k_mha_prep_fwd_c(const float* __restrict__ X, unsigned char* __restrict__ q, unsigned char* __restrict__ k,
                                                       unsigned char* __restrict__ v, int64_t R, int E, int64_t ld_x, float scale,
                                                       const float* qmin_q, const float* qmax_q, const float* qmin_k, const float* qmax_k,
                                                       const float* qmin_v, const float* qmax_v, const float* qmin_d, const float* qmax_d) {
    const QRange rq = load_qrange(qmin_q, qmax_q), rk = load_qrange(qmin_k, qmax_k), rv = load_qrange(qmin_v, qmax_v),
                 rd = load_qrange(qmin_d, qmax_d);
    const int e4 = E >> 2;
    const int64_t n4 = R * e4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / e4;
        const int f = (int)(i - row * e4) * 4;
        const float* xr = X + row * ld_x + f;
        const float4 a = *reinterpret_cast<const float4*>(xr), b = *reinterpret_cast<const float4*>(xr + E),
                     c = *reinterpret_cast<const float4*>(xr + 2 * E);
        const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w}, cv[4] = {c.x, c.y, c.z, c.w};
        unsigned int wq = 0, wk = 0, wv = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float cc, u;
            bool inr;
            const float qp = fq_asym(av[j], rq, cc, u, inr);
            (void)fq_asym(qp / scale, rd, cc, u, inr);
            wq |= (unsigned int)cc << (8 * j);
            (void)fq_asym(bv[j], rk, cc, u, inr);
            wk |= (unsigned int)cc << (8 * j);
            (void)fq_asym(cv[j], rv, cc, u, inr);
            wv |= (unsigned int)cc << (8 * j);
        }
        const int64_t o = row * E + f;
        *reinterpret_cast<unsigned int*>(q + o) = wq;
        *reinterpret_cast<unsigned int*>(k + o) = wk;
        *reinterpret_cast<unsigned int*>(v + o) = wv;
    }
}

// STE + range partials of one quantizer at one element (the arithmetic of k_actq_bwd, ACT_NONE)
__device__ __forceinline__ float mha_ste(float t, float g, const QRange& r, float& p_du, float& p_out) {
    float c, u;
    bool inr;
    (void)fq_asym(t, r, c, u, inr);
    p_du += g * (inr ? (c - u) : c);
    p_out += inr ? 0.0f : g;
    return inr ? div_by(g * r.delta, r.delta, r.inv) : 0.0f;
}

__global__ __launch_bounds__(256) void k_mha_prep_bwd(const float* __restrict__ X, const float* __restrict__ gq, const float* __restrict__ gk,
                                                       const float* __restrict__ gv, float* __restrict__ gX, int64_t R, int E, int64_t ld_x,
                                                       int64_t ld_gx, float scale, const float* qmin_q, const float* qmax_q,
                                                       const float* qmin_k, const float* qmax_k, const float* qmin_v, const float* qmax_v,
                                                       const float* qmin_d, const float* qmax_d, double* gacc_q, double* gacc_k,
                                                       double* gacc_v, double* gacc_d) {
    __shared__ double red[8 * 4];
    const QRange rq = load_qrange(qmin_q, qmax_q), rk = load_qrange(qmin_k, qmax_k), rv = load_qrange(qmin_v, qmax_v),
                 rd = load_qrange(qmin_d, qmax_d);
    float du[4] = {0.f, 0.f, 0.f, 0.f}, po[4] = {0.f, 0.f, 0.f, 0.f};      // q, k, v, div
    const int e4 = E >> 2;
    const int64_t n4 = R * e4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / e4;
        const int f = (int)(i - row * e4) * 4;
        const float* xr = X + row * ld_x + f;
        const int64_t o = row * E + f;
        const float4 a = *reinterpret_cast<const float4*>(xr), b = *reinterpret_cast<const float4*>(xr + E),
                     c = *reinterpret_cast<const float4*>(xr + 2 * E);
        const float4 g0 = *reinterpret_cast<const float4*>(gq + o), g1 = *reinterpret_cast<const float4*>(gk + o),
                     g2 = *reinterpret_cast<const float4*>(gv + o);
        const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w}, cv[4] = {c.x, c.y, c.z, c.w};
        const float gqv[4] = {g0.x, g0.y, g0.z, g0.w}, gkv[4] = {g1.x, g1.y, g1.z, g1.w}, gvv[4] = {g2.x, g2.y, g2.z, g2.w};
        float oa[4], ob[4], oc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float cc, u;
            bool inr;
            const float qp = fq_asym(av[j], rq, cc, u, inr);              // the value the div quantizer saw: (fq_q(x)) / scale
            const float gd = mha_ste(qp / scale, gqv[j], rd, du[3], po[3]);
            oa[j] = mha_ste(av[j], gd / scale, rq, du[0], po[0]);         // d(q / s) = g / s (fqss_unary_bwd), then the q quantizer's STE
            ob[j] = mha_ste(bv[j], gkv[j], rk, du[1], po[1]);
            oc[j] = mha_ste(cv[j], gvv[j], rv, du[2], po[2]);
        }
        float* gr = gX + row * ld_gx + f;
        *reinterpret_cast<float4*>(gr) = make_float4(oa[0], oa[1], oa[2], oa[3]);
        *reinterpret_cast<float4*>(gr + E) = make_float4(ob[0], ob[1], ob[2], ob[3]);
        *reinterpret_cast<float4*>(gr + 2 * E) = make_float4(oc[0], oc[1], oc[2], oc[3]);
    }
    double v[8] = {(double)du[0], (double)po[0], (double)du[1], (double)po[1], (double)du[2], (double)po[2], (double)du[3], (double)po[3]};
    block_sum<double, 8>(v, red);
    if (threadIdx.x == 0) {
        double* accs[4] = {gacc_q, gacc_k, gacc_v, gacc_d};
        for (int t = 0; t < 4; ++t) {          // one slot per workgroup (grid <= FQSS_GACC_SLOTS), as k_actq_bwd
            double* slot = accs[t] + 3 * (int64_t)blockIdx.x;
            const double dmax = v[2 * t] / 255.0;
            slot[0] += v[2 * t + 1] - dmax;
            slot[1] += dmax;
        }
    }
}

}  // namespace fqss

using namespace fqss;

static int layernorm_fwd_impl(const char* who, const float* x, const float* gamma, const float* beta, float* y, uint8_t* yc,
                              float* mean_rstd, int64_t R, int C, int64_t ld_x, int64_t ld_y, int64_t ld_yc, double eps,
                              const float* qmin, const float* qmax, fqss_stream_t stream, const float* xadd = nullptr, int64_t ld_a = 0,
                              float* xsum = nullptr, int64_t ld_s = 0, const float* qs_min = nullptr, const float* qs_max = nullptr,
                              RowMap om = RowMap{0, 0, 0, 0, 0}) {
    if (R == 0) return FQSS_OK;
    hipStream_t s = (hipStream_t)stream;
    const float e = (float)eps;
    // float4 form: every row (and gamma / beta) 16-B aligned, C a multiple of 4
    const bool vec = C % 4 == 0 && ld_x % 4 == 0 && ld_y % 4 == 0 && aligned16(x) && aligned16(y) && aligned16(gamma) && aligned16(beta) &&
                     (xadd == nullptr || (ld_a % 4 == 0 && ld_s % 4 == 0 && aligned16(xadd) && aligned16(xsum)));
    const bool narrow = vec && C <= 64;                 // four rows per wave
    int64_t nb = cdiv(R, narrow ? 16 : 4);
    if (nb > 4096) nb = 4096;
#define FQSS_LN_FWD(JC, Q, ...) \
    hipLaunchKernelGGL((k_layernorm_fwd<JC, Q, __VA_ARGS__>), dim3((unsigned)nb), dim3(256), 0, s, x, gamma, beta, y, mean_rstd, R, C, ld_x, ld_y, e, qmin, \
                       qmax, yc, ld_yc, xadd, ld_a, xsum, ld_s, qs_min, qs_max, om)
#define FQSS_LN_FWD_Q(Q) \
    if (narrow) FQSS_LN_FWD(4, Q, true, 16); \
    else if (C <= 64) FQSS_LN_FWD(1, Q, false); \
    else if (C <= 256) { if (vec) FQSS_LN_FWD(4, Q, true); else FQSS_LN_FWD(4, Q, false); } \
    else { if (vec) FQSS_LN_FWD(8, Q, true); else FQSS_LN_FWD(8, Q, false); }      /* HTDemucs transformer: 384 / 512 */
    if (qmin != nullptr) { FQSS_LN_FWD_Q(true) } else { FQSS_LN_FWD_Q(false) }
#undef FQSS_LN_FWD_Q
#undef FQSS_LN_FWD
    return launch_status(who);
}

extern "C" int fqss_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean_rstd,
                                  int64_t R, int C, int64_t ld_x, int64_t ld_y, double eps, fqss_stream_t stream) {
    FQSS_REQUIRE(x && gamma && beta && y && mean_rstd, "null tensor");
    FQSS_REQUIRE(R >= 0 && C > 0 && C <= 512 && ld_x >= C && ld_y >= C, "bad shape (C <= 512)");
    return layernorm_fwd_impl("fqss_layernorm_fwd", x, gamma, beta, y, nullptr, mean_rstd, R, C, ld_x, ld_y, 0, eps, nullptr, nullptr, stream);
}

extern "C" int fqss_layernormq_fwd(const float* x, const float* gamma, const float* beta, float* y, uint8_t* yc, float* mean_rstd,
                                   int64_t R, int C, int64_t ld_x, int64_t ld_y, int64_t ld_yc, double eps, const float* qmin,
                                   const float* qmax, fqss_stream_t stream) {
    FQSS_REQUIRE(x && gamma && beta && y && mean_rstd && qmin && qmax, "null tensor");
    FQSS_REQUIRE(R >= 0 && C > 0 && C <= 512 && ld_x >= C && ld_y >= C && (yc == nullptr || ld_yc >= C), "bad shape (C <= 512)");
    return layernorm_fwd_impl("fqss_layernormq_fwd", x, gamma, beta, y, yc, mean_rstd, R, C, ld_x, ld_y, ld_yc, eps, qmin, qmax, stream);
}

static int layernorm_bwd_impl(const char* who, const float* gy, const float* x, const float* gamma, const float* beta,
                              const float* mean_rstd, float* gx, float* ggamma, float* gbeta, int64_t R, int C, int64_t ld_gy,
                              int64_t ld_x, int64_t ld_gx, const float* qmin, const float* qmax, double* gacc, fqss_stream_t stream,
                              const float* gadd = nullptr, int64_t ld_ga = 0, const float* qs_min = nullptr, const float* qs_max = nullptr,
                              double* gacc_s = nullptr, RowMap gm = RowMap{0, 0, 0, 0, 0}) {
    if (R == 0) return FQSS_OK;
    hipStream_t s = (hipStream_t)stream;
    const bool vec = C % 4 == 0 && ld_gy % 4 == 0 && ld_x % 4 == 0 && ld_gx % 4 == 0 && aligned16(gy) && aligned16(x) && aligned16(gx) &&
                     aligned16(gamma) && (beta == nullptr || aligned16(beta)) && (gadd == nullptr || (ld_ga % 4 == 0 && aligned16(gadd)));
    const bool narrow = vec && C <= 64;
    // rows per wave: a trade between waves in flight (16 k rows x 256 features at 16 rows per wave leave ONE wave per SIMD: the pass is
    // latency-bound) and the 2 C atomics per workgroup.  Measured at the step level (FQSS_LN_BWD_ROWS sweep): 8 rows per wave for the wide
    // rows (cfg 4 29.0 -> 28.7 ms; 4: 28.8, 2: 29.6), 8 groups of four for the 64-wide rows of cfg 3 (fewer is slower there)
    static const int rpw_env = [] { const char* e = getenv("FQSS_LN_BWD_ROWS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 0; }();
    const int rpw = rpw_env > 0 ? rpw_env : (narrow ? 16 : 8);
    int64_t nb = cdiv(R, narrow ? 4 * (rpw / 2 > 0 ? rpw / 2 : 1) * 4 : 4 * rpw);
    if (nb < 1) nb = 1;
    if (nb > 2048) nb = 2048;           // (also the number of gacc slots)
#define FQSS_LN_BWD(JC, Q, ...) \
    hipLaunchKernelGGL((k_layernorm_bwd<JC, Q, __VA_ARGS__>), dim3((unsigned)nb), dim3(256), 0, s, gy, x, gamma, mean_rstd, gx, ggamma, gbeta, R, C, \
                       ld_gy, ld_x, ld_gx, beta, qmin, qmax, gacc, gadd, ld_ga, qs_min, qs_max, gacc_s, gm)
#define FQSS_LN_BWD_Q(Q) \
    if (narrow) FQSS_LN_BWD(4, Q, true, 16); \
    else if (C <= 64) FQSS_LN_BWD(1, Q, false); \
    else if (C <= 256) { if (vec) FQSS_LN_BWD(4, Q, true); else FQSS_LN_BWD(4, Q, false); } \
    else { if (vec) FQSS_LN_BWD(8, Q, true); else FQSS_LN_BWD(8, Q, false); }
    if (qmin != nullptr) { FQSS_LN_BWD_Q(true) } else { FQSS_LN_BWD_Q(false) }
#undef FQSS_LN_BWD_Q
#undef FQSS_LN_BWD
    return launch_status(who);
}

extern "C" int fqss_layernorm_bwd(const float* gy, const float* x, const float* gamma, const float* mean_rstd, float* gx,
                                  float* ggamma, float* gbeta, int64_t R, int C, int64_t ld_gy, int64_t ld_x, int64_t ld_gx,
                                  fqss_stream_t stream) {
    FQSS_REQUIRE(gy && x && gamma && mean_rstd && gx && ggamma && gbeta, "null tensor");
    FQSS_REQUIRE(R >= 0 && C > 0 && C <= 512 && ld_gy >= C && ld_x >= C && ld_gx >= C, "bad shape (C <= 512)");
    return layernorm_bwd_impl("fqss_layernorm_bwd", gy, x, gamma, nullptr, mean_rstd, gx, ggamma, gbeta, R, C, ld_gy, ld_x, ld_gx, nullptr,
                              nullptr, nullptr, stream);
}

extern "C" int fqss_layernormq_bwd(const float* g, const float* x, const float* gamma, const float* beta, const float* mean_rstd,
                                   float* gx, float* ggamma, float* gbeta, int64_t R, int C, int64_t ld_g, int64_t ld_x, int64_t ld_gx,
                                   const float* qmin, const float* qmax, double* gacc, fqss_stream_t stream) {
    FQSS_REQUIRE(g && x && gamma && beta && mean_rstd && gx && ggamma && gbeta && qmin && qmax && gacc, "null tensor");
    FQSS_REQUIRE(R >= 0 && C > 0 && C <= 512 && ld_g >= C && ld_x >= C && ld_gx >= C, "bad shape (C <= 512)");
    return layernorm_bwd_impl("fqss_layernormq_bwd", g, x, gamma, beta, mean_rstd, gx, ggamma, gbeta, R, C, ld_g, ld_x, ld_gx, qmin, qmax,
                              gacc, stream);
}


/* the residual add in FRONT of a pre-norm sub-layer fused into its LayerNorm(Q) (sepformerq.py:69-82: `x = x + mha(norm1(x))`, then
 * `norm2(x)` ...): s = a + b is written once (the residual stream), y = LN(s) or fq(LN(s)) (qmin / qmax / yc NULL: plain) */
extern "C" int fqss_add_layernorm_fwd(const float* a, const float* b, const float* gamma, const float* beta, float* s, float* y, uint8_t* yc,
                                      float* mean_rstd, int64_t R, int C, int64_t ld_a, int64_t ld_b, int64_t ld_s, int64_t ld_y,
                                      int64_t ld_yc, double eps, const float* qmin, const float* qmax, fqss_stream_t stream) {
    FQSS_REQUIRE(a && b && gamma && beta && s && y && mean_rstd && ((qmin == nullptr) == (qmax == nullptr)), "null tensor");
    FQSS_REQUIRE(R >= 0 && C > 0 && C <= 512 && ld_a >= C && ld_b >= C && ld_s >= C && ld_y >= C && (yc == nullptr || (qmin && ld_yc >= C)),
                 "bad shape (C <= 512)");
    return layernorm_fwd_impl("fqss_add_layernorm_fwd", a, gamma, beta, y, yc, mean_rstd, R, C, ld_a, ld_y, ld_yc, eps, qmin, qmax, stream, b,
                              ld_b, s, ld_s);
}

/* its backward: g = dL/dy, gs = dL/ds arriving over the residual stream (nullable); gx = LayerNorm(Q) backward at s + gs: the gradient
 * of a AND of b; ggamma / gbeta "+=", the quantizer's range partials to gacc (qmin / qmax / gacc NULL: plain) */
extern "C" int fqss_add_layernorm_bwd(const float* g, const float* gs, const float* s, const float* gamma, const float* beta,
                                      const float* mean_rstd, float* gx, float* ggamma, float* gbeta, int64_t R, int C, int64_t ld_g,
                                      int64_t ld_gs, int64_t ld_s, int64_t ld_gx, const float* qmin, const float* qmax, double* gacc,
                                      fqss_stream_t stream) {
    FQSS_REQUIRE(g && s && gamma && mean_rstd && gx && ggamma && gbeta, "null tensor");
    FQSS_REQUIRE((qmin == nullptr) == (qmax == nullptr) && (qmin == nullptr || (gacc && beta)), "quantizer: ranges, beta and gacc together");
    FQSS_REQUIRE(R >= 0 && C > 0 && C <= 512 && ld_g >= C && ld_s >= C && ld_gx >= C && (gs == nullptr || ld_gs >= C), "bad shape (C <= 512)");
    return layernorm_bwd_impl("fqss_add_layernorm_bwd", g, s, gamma, beta, mean_rstd, gx, ggamma, gbeta, R, C, ld_g, ld_s, ld_gx, qmin, qmax,
                              gacc, stream, gs, ld_gs);
}

/* the same pair for a QUANTIZED add in front of the norm -- y = LN(Q)(fq_s(a + b)), the post-norm layers of DPTNet (dptnetq.py:84-97:
 * AddQ then LayerNormQ): z receives the PRE-quant sum (the backward's input), qs_min / qs_max are the AddQ's range; the backward's gx
 * (the gradient of a AND of b) has passed the AddQ's STE, its range partials go to gacc_s */
extern "C" int fqss_addq_layernorm_fwd(const float* a, const float* b, const float* gamma, const float* beta, float* z, float* y, uint8_t* yc,
                                       float* mean_rstd, int64_t R, int C, int64_t ld_a, int64_t ld_b, int64_t ld_z, int64_t ld_y,
                                       int64_t ld_yc, double eps, const float* qmin, const float* qmax, const float* qs_min,
                                       const float* qs_max, fqss_stream_t stream) {
    FQSS_REQUIRE(a && b && gamma && beta && z && y && mean_rstd && qs_min && qs_max && ((qmin == nullptr) == (qmax == nullptr)), "null tensor");
    FQSS_REQUIRE(R >= 0 && C > 0 && C <= 512 && ld_a >= C && ld_b >= C && ld_z >= C && ld_y >= C && (yc == nullptr || (qmin && ld_yc >= C)),
                 "bad shape (C <= 512)");
    return layernorm_fwd_impl("fqss_addq_layernorm_fwd", a, gamma, beta, y, yc, mean_rstd, R, C, ld_a, ld_y, ld_yc, eps, qmin, qmax, stream, b,
                              ld_b, z, ld_z, qs_min, qs_max);
}

extern "C" int fqss_addq_layernorm_bwd(const float* g, const float* gs, const float* z, const float* gamma, const float* beta,
                                       const float* mean_rstd, float* gx, float* ggamma, float* gbeta, int64_t R, int C, int64_t ld_g,
                                       int64_t ld_gs, int64_t ld_z, int64_t ld_gx, const float* qmin, const float* qmax, double* gacc,
                                       const float* qs_min, const float* qs_max, double* gacc_s, fqss_stream_t stream) {
    FQSS_REQUIRE(g && z && gamma && mean_rstd && gx && ggamma && gbeta && qs_min && qs_max && gacc_s, "null tensor");
    FQSS_REQUIRE((qmin == nullptr) == (qmax == nullptr) && (qmin == nullptr || (gacc && beta)), "quantizer: ranges, beta and gacc together");
    FQSS_REQUIRE(R >= 0 && C > 0 && C <= 512 && ld_g >= C && ld_z >= C && ld_gx >= C && (gs == nullptr || ld_gs >= C), "bad shape (C <= 512)");
    return layernorm_bwd_impl("fqss_addq_layernorm_bwd", g, z, gamma, beta, mean_rstd, gx, ggamma, gbeta, R, C, ld_g, ld_z, ld_gx, qmin, qmax,
                              gacc, stream, gs, ld_gs, qs_min, qs_max, gacc_s);
}

/* fqss_addq_layernorm_fwd / _bwd with the dual-path layout change folded in: y (and yc) row (i0 * d1 + i1) * d2 + i2 is WRITTEN at row
 * i0 * t0 + i1 * t1 + i2 * t2 (dense rows of C), and the backward reads dL/dy of row r from there -- the transposing copy between the
 * intra- and the inter-chunk transformer (fqss_permute4 each way, plus one for the codes) disappears.  The map must be a permutation
 * of the R rows (host: d1 * d2 divides R, t's as of a transposed [i2][i1][i0] / [i0 ...] layout; checked: every t >= 1). */
extern "C" int fqss_addq_layernorm_fwd_map(const float* a, const float* b, const float* gamma, const float* beta, float* z, float* y,
                                           uint8_t* yc, float* mean_rstd, int64_t R, int C, int64_t ld_a, int64_t ld_b, int64_t ld_z, double eps,
                                           const float* qmin, const float* qmax, const float* qs_min, const float* qs_max, int64_t d1, int64_t d2,
                                           int64_t t0, int64_t t1, int64_t t2, fqss_stream_t stream) {
    FQSS_REQUIRE(a && b && gamma && beta && z && y && mean_rstd && ((qs_min == nullptr) == (qs_max == nullptr)) &&
                     ((qmin == nullptr) == (qmax == nullptr)),
                 "null tensor");      // (qs NULL: a plain add in front of the norm -- the float teacher's layers, forward only)
    FQSS_REQUIRE(R >= 0 && C > 0 && C <= 512 && ld_a >= C && ld_b >= C && ld_z >= C && (yc == nullptr || qmin), "bad shape (C <= 512)");
    FQSS_REQUIRE(d1 >= 1 && d2 >= 1 && t0 >= 1 && t1 >= 1 && t2 >= 1 && R % (d1 * d2) == 0, "row map: d1 * d2 must divide the row count");
    return layernorm_fwd_impl("fqss_addq_layernorm_fwd_map", a, gamma, beta, y, yc, mean_rstd, R, C, ld_a, C, C, eps, qmin, qmax, stream, b, ld_b,
                              z, ld_z, qs_min, qs_max, RowMap{d1, d2, t0, t1, t2});
}

extern "C" int fqss_addq_layernorm_bwd_map(const float* g, const float* z, const float* gamma, const float* beta, const float* mean_rstd,
                                           float* gx, float* ggamma, float* gbeta, int64_t R, int C, int64_t ld_z, int64_t ld_gx,
                                           const float* qmin, const float* qmax, double* gacc, const float* qs_min, const float* qs_max,
                                           double* gacc_s, int64_t d1, int64_t d2, int64_t t0, int64_t t1, int64_t t2, fqss_stream_t stream) {
    FQSS_REQUIRE(g && z && gamma && mean_rstd && gx && ggamma && gbeta && qs_min && qs_max && gacc_s, "null tensor");
    FQSS_REQUIRE((qmin == nullptr) == (qmax == nullptr) && (qmin == nullptr || (gacc && beta)), "quantizer: ranges, beta and gacc together");
    FQSS_REQUIRE(R >= 0 && C > 0 && C <= 512 && ld_z >= C && ld_gx >= C, "bad shape (C <= 512)");
    FQSS_REQUIRE(d1 >= 1 && d2 >= 1 && t0 >= 1 && t1 >= 1 && t2 >= 1 && R % (d1 * d2) == 0, "row map: d1 * d2 must divide the row count");
    return layernorm_bwd_impl("fqss_addq_layernorm_bwd_map", g, z, gamma, beta, mean_rstd, gx, ggamma, gbeta, R, C, C, ld_z, ld_gx, qmin, qmax,
                              gacc, stream, nullptr, 0, qs_min, qs_max, gacc_s, RowMap{d1, d2, t0, t1, t2});
}

extern "C" int fqss_colsum(const float* g, float* out, int64_t R, int C, int64_t ld, fqss_stream_t stream) {
    FQSS_REQUIRE(g && out && R >= 0 && C > 0 && ld >= C, "bad args");
    if (R == 0) return FQSS_OK;
    int CW = 256;
    while (CW / 2 >= C && CW > 8) CW /= 2;
    const int64_t gx = cdiv(C, CW);
    int64_t gy = cdiv(1024, gx);
    int64_t rpb = cdiv(R, gy);
    if (rpb < 64) rpb = 64;
    gy = cdiv(R, rpb);
    hipLaunchKernelGGL(k_colsum, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, g, out, R, C, ld, rpb, CW);
    return launch_status("fqss_colsum");
}


extern "C" int fqss_mha_prep_fwd(const float* X, float* q, float* k, float* v, int64_t R, int E, int64_t ld_x, double scale,
                                 const float* const* ranges, fqss_stream_t stream) {
    if (R == 0) return FQSS_OK;
    FQSS_REQUIRE(X && q && k && v && ranges && R > 0 && E > 0 && E % 4 == 0 && ld_x >= 3 * (int64_t)E && ld_x % 4 == 0, "bad shape");
    FQSS_REQUIRE(aligned16(X) && aligned16(q) && aligned16(k) && aligned16(v), "rows must be 16-B aligned");
    FQSS_REQUIRE(scale != 0.0, "division by zero");
    for (int i = 0; i < 8; ++i) FQSS_REQUIRE(ranges[i], "null range (q, k, v, div: min, max each)");
    hipLaunchKernelGGL(k_mha_prep_fwd, dim3(flat_grid(R * (E / 4))), dim3(256), 0, (hipStream_t)stream, X, q, k, v, R, E, ld_x, (float)scale,
                       ranges[0], ranges[1], ranges[2], ranges[3], ranges[4], ranges[5], ranges[6], ranges[7]);
    return launch_status("fqss_mha_prep_fwd");
}

extern "C" int fqss_mha_prep_fwd_c(const float* X, uint8_t* qc, uint8_t* kc, uint8_t* vc, int64_t R, int E, int64_t ld_x, double scale,
                                   const float* const* ranges, fqss_stream_t stream) {
    if (R == 0) return FQSS_OK;
    FQSS_REQUIRE(X && qc && kc && vc && ranges && R > 0 && E > 0 && E % 4 == 0 && ld_x >= 3 * (int64_t)E && ld_x % 4 == 0, "bad shape");
    FQSS_REQUIRE(aligned16(X) && ((uintptr_t)qc & 3) == 0 && ((uintptr_t)kc & 3) == 0 && ((uintptr_t)vc & 3) == 0, "rows must be aligned");
    FQSS_REQUIRE(scale != 0.0, "division by zero");
    for (int i = 0; i < 8; ++i) FQSS_REQUIRE(ranges[i], "null range (q, k, v, div: min, max each)");
    hipLaunchKernelGGL(k_mha_prep_fwd_c, dim3(flat_grid(R * (E / 4))), dim3(256), 0, (hipStream_t)stream, X, qc, kc, vc, R, E, ld_x, (float)scale,
                       ranges[0], ranges[1], ranges[2], ranges[3], ranges[4], ranges[5], ranges[6], ranges[7]);
    return launch_status("fqss_mha_prep_fwd_c");
}

extern "C" int fqss_mha_prep_bwd(const float* X, const float* gq, const float* gk, const float* gv, float* gX, int64_t R, int E,
                                 int64_t ld_x, int64_t ld_gx, double scale, const float* const* ranges, double* const* gaccs,
                                 fqss_stream_t stream) {
    if (R == 0) return FQSS_OK;
    FQSS_REQUIRE(X && gq && gk && gv && gX && ranges && gaccs && R > 0 && E > 0 && E % 4 == 0, "bad shape");
    FQSS_REQUIRE(ld_x >= 3 * (int64_t)E && ld_gx >= 3 * (int64_t)E && ld_x % 4 == 0 && ld_gx % 4 == 0, "bad row strides");
    FQSS_REQUIRE(aligned16(X) && aligned16(gq) && aligned16(gk) && aligned16(gv) && aligned16(gX), "rows must be 16-B aligned");
    FQSS_REQUIRE(scale != 0.0, "division by zero");
    for (int i = 0; i < 8; ++i) FQSS_REQUIRE(ranges[i], "null range (q, k, v, div: min, max each)");
    for (int i = 0; i < 4; ++i) FQSS_REQUIRE(gaccs[i], "null gacc (q, k, v, div)");
    unsigned grid = flat_grid(R * (E / 4));
    if (grid > FQSS_GACC_SLOTS) grid = FQSS_GACC_SLOTS;
    hipLaunchKernelGGL(k_mha_prep_bwd, dim3(grid), dim3(256), 0, (hipStream_t)stream, X, gq, gk, gv, gX, R, E, ld_x, ld_gx, (float)scale,
                       ranges[0], ranges[1], ranges[2], ranges[3], ranges[4], ranges[5], ranges[6], ranges[7], gaccs[0], gaccs[1], gaccs[2],
                       gaccs[3]);
    return launch_status("fqss_mha_prep_bwd");
}

extern "C" int fqss_unary_fwd(const float* x, float* y, int64_t n, int kind, double p, fqss_stream_t stream) {
    FQSS_REQUIRE(x && y && n >= 0 && kind >= 0 && kind <= 3, "bad args");
    FQSS_REQUIRE(kind != 2 || p != 0.0, "division by zero");
    if (n == 0) return FQSS_OK;
    hipLaunchKernelGGL(k_unary_fwd, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, kind, (float)p, 0.0f);
    return launch_status("fqss_unary_fwd");
}

extern "C" int fqss_unary_rows_fwd(const float* x, float* y, int64_t rows, int cols, int64_t ld_x, int64_t ld_y, int kind, double p,
                                   fqss_stream_t stream) {
    FQSS_REQUIRE(x && y && rows >= 0 && cols >= 0 && kind >= 0 && kind <= 3, "bad args");
    FQSS_REQUIRE(kind != 2 || p != 0.0, "division by zero");
    FQSS_REQUIRE(cols % 4 == 0 && ld_x % 4 == 0 && ld_y % 4 == 0 && ld_x >= cols && ld_y >= cols && aligned16(x) && aligned16(y),
                 "rows of 4-float groups, 16-B aligned");
    if (rows == 0 || cols == 0) return FQSS_OK;
    hipLaunchKernelGGL(k_unary_rows_fwd, dim3(flat_grid(rows * (cols / 4))), dim3(256), 0, (hipStream_t)stream, x, y, rows, cols, ld_x, ld_y, kind, (float)p);
    return launch_status("fqss_unary_rows_fwd");
}

extern "C" int fqss_unary2_fwd(const float* x, float* y, int64_t n, int kind, double p, double p2, fqss_stream_t stream) {
    if (n == 0) return FQSS_OK;
    FQSS_REQUIRE(x && y && n >= 0 && kind >= 0 && kind <= FQSS_UNARY_CLIP, "bad args");
    hipLaunchKernelGGL(k_unary_fwd, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, kind, (float)p, (float)p2);
    return launch_status("fqss_unary2_fwd");
}

extern "C" int fqss_unary_bwd(const float* g, const float* y, float* gx, int64_t n, int kind, double p, fqss_stream_t stream) {
    FQSS_REQUIRE(g && gx && n >= 0 && kind >= 0 && kind <= 3 && (kind == 2 || y), "bad args");
    if (n == 0) return FQSS_OK;
    hipLaunchKernelGGL(k_unary_bwd, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream, g, y, gx, n, kind, (float)p);
    return launch_status("fqss_unary_bwd");
}

extern "C" int fqss_permute4(const float* x, float* y, int64_t n0, int64_t n1, int64_t n2, int C, int64_t s0, int64_t s1,
                             int64_t s2, fqss_stream_t stream) {
    FQSS_REQUIRE(x && y && n0 >= 0 && n1 >= 0 && n2 >= 0 && C > 0, "bad args");
    const int64_t n = n0 * n1 * n2 * C;
    if (n == 0) return FQSS_OK;
    hipLaunchKernelGGL(k_permute4, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream, x, y, n0, n1, n2, C, s0, s1, s2);
    return launch_status("fqss_permute4");
}

extern "C" int fqss_permute4_ld(const float* x, float* y, int64_t n0, int64_t n1, int64_t n2, int C, int64_t s0, int64_t s1,
                                int64_t s2, int64_t ld_y, fqss_stream_t stream) {
    FQSS_REQUIRE(x && y && n0 >= 0 && n1 >= 0 && n2 >= 0 && C > 0, "bad args");
    FQSS_REQUIRE(ld_y >= C && ld_y % 4 == 0 && aligned16(y) && n1 < (1ll << 31) && n2 < (1ll << 31), "padded output rows: 16-B aligned, a multiple of 4 floats apart");
    const int64_t rows = n0 * n1 * n2;
    if (rows == 0) return FQSS_OK;
    hipLaunchKernelGGL(k_permute4_ld, dim3(flat_grid(rows * (ld_y / 4))), dim3(256), 0, (hipStream_t)stream, x, y, rows, (int)n1, (int)n2, C, s0, s1, s2, ld_y);
    return launch_status("fqss_permute4_ld");
}

extern "C" int fqss_dp_segment_fwd(const float* f, float* seg, int B, int N, int64_t T, int64_t ld_f, int K, int S,
                                   fqss_stream_t stream) {
    FQSS_REQUIRE(f && seg && B > 0 && N > 0 && T > 0 && ld_f >= T && K >= 2 && K % 2 == 0 && S >= 2 && S % 2 == 0, "bad args");
    FQSS_REQUIRE((int64_t)(S / 2) * K >= T + K / 2, "S chunks do not cover the signal");
    hipLaunchKernelGGL(k_dp_segment_fwd, dim3(flat_grid((int64_t)K * B * S * N)), dim3(256), 0, (hipStream_t)stream, f, seg, B, N, T, ld_f, K, S);
    return launch_status("fqss_dp_segment_fwd");
}

extern "C" int fqss_dp_segment_bwd(const float* gseg, float* gf, int B, int N, int64_t T, int64_t ld_gf, int K, int S,
                                   fqss_stream_t stream) {
    FQSS_REQUIRE(gseg && gf && B > 0 && N > 0 && T > 0 && ld_gf >= T && K >= 2 && K % 2 == 0 && S >= 2 && S % 2 == 0, "bad args");
    hipLaunchKernelGGL(k_dp_segment_bwd, dim3(flat_grid((int64_t)B * N * T)), dim3(256), 0, (hipStream_t)stream, gseg, gf, B, N, T, ld_gf, K, S);
    return launch_status("fqss_dp_segment_bwd");
}

extern "C" int fqss_dp_merge_fwd(const float* o, float* a, float* b, int B, int nspk, int N, int K, int S, int64_t Lm,
                                 int64_t ld_ab, fqss_stream_t stream) {
    FQSS_REQUIRE(o && a && b && B > 0 && nspk > 0 && N > 0 && K >= 2 && K % 2 == 0 && S >= 2 && S % 2 == 0, "bad args");
    FQSS_REQUIRE(Lm == (int64_t)(S / 2) * K - K / 2 && ld_ab >= Lm, "Lm must be (S/2)*K - K/2");
    hipLaunchKernelGGL(k_dp_merge_fwd, dim3(flat_grid((int64_t)B * nspk * N * Lm)), dim3(256), 0, (hipStream_t)stream, o, a, b, B, nspk, N, K, S, Lm, ld_ab);
    return launch_status("fqss_dp_merge_fwd");
}

extern "C" int fqss_dp_merge_bwd(const float* ga, const float* gb, float* go, int B, int nspk, int N, int K, int S, int64_t Lm,
                                 int64_t ld_ga, int64_t ld_gb, fqss_stream_t stream) {
    FQSS_REQUIRE(ga && gb && go && B > 0 && nspk > 0 && N > 0 && K >= 2 && K % 2 == 0 && S >= 2 && S % 2 == 0, "bad args");
    FQSS_REQUIRE(Lm == (int64_t)(S / 2) * K - K / 2 && ld_ga >= Lm && ld_gb >= Lm, "Lm must be (S/2)*K - K/2");
    hipLaunchKernelGGL(k_dp_merge_bwd, dim3(flat_grid((int64_t)S * B * K * nspk * N)), dim3(256), 0, (hipStream_t)stream, ga, gb, go, B, nspk, N, K, S, Lm, ld_ga, ld_gb);
    return launch_status("fqss_dp_merge_bwd");
}

extern "C" int fqss_ola2_fwd(const float* y, float* out, int64_t N, int64_t L, int64_t ld_y, fqss_stream_t stream) {
    FQSS_REQUIRE(y && out && N >= 0 && L > 0 && ld_y >= L, "bad args");
    if (N == 0) return FQSS_OK;
    hipLaunchKernelGGL(k_ola2_fwd, dim3(flat_grid(N * (L + 1))), dim3(256), 0, (hipStream_t)stream, y, out, N, L, ld_y);
    return launch_status("fqss_ola2_fwd");
}

extern "C" int fqss_ola2_bwd(const float* g, float* gy, int64_t N, int64_t L, int64_t ld_gy, fqss_stream_t stream) {
    FQSS_REQUIRE(g && gy && N >= 0 && L > 0 && ld_gy >= L, "bad args");
    if (N == 0) return FQSS_OK;
    hipLaunchKernelGGL(k_ola2_bwd, dim3(flat_grid(N * 2 * L)), dim3(256), 0, (hipStream_t)stream, g, gy, N, L, ld_gy);
    return launch_status("fqss_ola2_bwd");
}

extern "C" int fqss_gnrows_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean_rstd, double* ws,
                               int64_t R, int C, int64_t ld_x, int64_t ld_y, int RB, int X, int B, double eps,
                               fqss_stream_t stream) {
    FQSS_REQUIRE(x && gamma && beta && y && mean_rstd && ws, "null tensor");
    FQSS_REQUIRE(R > 0 && C > 0 && ld_x >= C && ld_y >= C && RB > 0 && X > 0 && B > 0 && B <= kMaxB && RB == B * X && R % RB == 0,
                 "bad shape (rows = n * B * X, B <= 16)");
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(ws, 0, sizeof(double) * 2 * B, s);
    int64_t nb = cdiv(R, 4 * 8);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(k_gnrows_stats, dim3((unsigned)nb), dim3(256), 0, s, x, ws, R, C, ld_x, RB, X, B);
    const double n = (double)(R / B) * C;
    hipLaunchKernelGGL(k_gnrows_finalize, dim3(1), dim3(64), 0, s, ws, mean_rstd, B, n, (float)eps);
    hipLaunchKernelGGL(k_gnrows_apply, dim3(flat_grid(R * C)), dim3(256), 0, s, x, gamma, beta, mean_rstd, y, R, C, ld_x, ld_y, RB, X);
    return launch_status("fqss_gnrows_fwd");
}

extern "C" int fqss_gnrows_bwd(const float* gy, const float* x, const float* gamma, const float* mean_rstd, float* gx,
                               float* ggamma, float* gbeta, double* ws, int64_t R, int C, int64_t ld_gy, int64_t ld_x,
                               int64_t ld_gx, int RB, int X, int B, fqss_stream_t stream) {
    FQSS_REQUIRE(gy && x && gamma && mean_rstd && gx && ggamma && gbeta && ws, "null tensor");
    FQSS_REQUIRE(R > 0 && C > 0 && C <= 1024 && ld_gy >= C && ld_x >= C && ld_gx >= C && RB > 0 && X > 0 && B > 0 && B <= kMaxB &&
                     RB == B * X && R % RB == 0, "bad shape (rows = n * B * X, B <= 16, C <= 1024)");
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(ws, 0, sizeof(double) * 2 * B, s);
    int64_t rpb = cdiv(R, 1024);
    if (rpb < 8) rpb = 8;
    hipLaunchKernelGGL(k_gnrows_bwd_reduce, dim3((unsigned)cdiv(R, rpb)), dim3(256), 0, s, gy, x, gamma, mean_rstd, ws, ggamma, gbeta, R,
                       C, ld_gy, ld_x, RB, X, B, rpb);
    const double n = (double)(R / B) * C;
    hipLaunchKernelGGL(k_gnrows_bwd_apply, dim3(flat_grid(R * C)), dim3(256), 0, s, gy, x, gamma, mean_rstd, ws, gx, R, C, ld_gy, ld_x,
                       ld_gx, RB, X, n);
    return launch_status("fqss_gnrows_bwd");
}

extern "C" int fqss_bcast_add(const float* x, const float* p, float* z, int64_t L, int64_t Bp, int C, fqss_stream_t stream) {
    FQSS_REQUIRE(x && p && z && L >= 0 && Bp >= 0 && C > 0, "bad args");
    if (L * Bp == 0) return FQSS_OK;
    hipLaunchKernelGGL(k_bcast_add, dim3(flat_grid(L * Bp * C)), dim3(256), 0, (hipStream_t)stream, x, p, z, L, Bp, C);
    return launch_status("fqss_bcast_add");
}

extern "C" int fqss_bcast_sum(const float* g, float* out, int64_t L, int64_t Bp, int C, fqss_stream_t stream) {
    FQSS_REQUIRE(g && out && L >= 0 && Bp >= 0 && C > 0, "bad args");
    if (L == 0) return FQSS_OK;
    hipLaunchKernelGGL(k_bcast_sum, dim3(flat_grid(L * C)), dim3(256), 0, (hipStream_t)stream, g, out, L, Bp, C);
    return launch_status("fqss_bcast_sum");
}

extern "C" int fqss_glu_fwd(const float* x, float* y, int64_t B, int64_t C, int64_t M, int64_t ld_x, int64_t ld_y, fqss_stream_t stream) {
    FQSS_REQUIRE(x && y && B >= 0 && C > 0 && M >= 0 && ld_x >= M && ld_y >= M, "bad args");
    if (B * M == 0) return FQSS_OK;
    hipLaunchKernelGGL(k_glu_fwd, dim3(flat_grid(B * C * M)), dim3(256), 0, (hipStream_t)stream, x, y, B, C, M, ld_x, ld_y);
    return launch_status("fqss_glu_fwd");
}

extern "C" int fqss_glu_bwd(const float* x, const float* gy, float* gx, int64_t B, int64_t C, int64_t M, int64_t ld_x, int64_t ld_gy,
                            int64_t ld_gx, fqss_stream_t stream) {
    FQSS_REQUIRE(x && gy && gx && B >= 0 && C > 0 && M >= 0 && ld_x >= M && ld_gy >= M && ld_gx >= M, "bad args");
    if (B * M == 0) return FQSS_OK;
    hipLaunchKernelGGL(k_glu_bwd, dim3(flat_grid(B * C * M)), dim3(256), 0, (hipStream_t)stream, x, gy, gx, B, C, M, ld_x, ld_gy, ld_gx);
    return launch_status("fqss_glu_bwd");
}

extern "C" int fqss_div_fwd(const float* a, const float* b, float* y, int64_t n, fqss_stream_t stream) {
    FQSS_REQUIRE(a && b && y && n >= 0, "bad args");
    if (n == 0) return FQSS_OK;
    hipLaunchKernelGGL(k_div_fwd, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream, a, b, y, n);
    return launch_status("fqss_div_fwd");
}

extern "C" int fqss_div_bwd(const float* g, const float* a, const float* b, float* ga, float* gb, int64_t n, fqss_stream_t stream) {
    FQSS_REQUIRE(g && a && b && ga && gb && n >= 0, "bad args");
    if (n == 0) return FQSS_OK;
    hipLaunchKernelGGL(k_div_bwd, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream, g, a, b, ga, gb, n);
    return launch_status("fqss_div_bwd");
}

extern "C" int fqss_embedding_fwd(const float* w, const int64_t* idx, float* out, int64_t n, int D, int64_t V, fqss_stream_t stream) {
    FQSS_REQUIRE(w && idx && out && n >= 0 && D > 0 && V > 0, "bad args");
    if (n == 0) return FQSS_OK;
    hipLaunchKernelGGL(k_embedding_fwd, dim3(flat_grid(n * D)), dim3(256), 0, (hipStream_t)stream, w, idx, out, n, D, V);
    return launch_status("fqss_embedding_fwd");
}

extern "C" int fqss_embedding_bwd(const float* g, const int64_t* idx, float* gw, int64_t n, int D, int64_t V, fqss_stream_t stream) {
    FQSS_REQUIRE(g && idx && gw && n >= 0 && D > 0 && V > 0, "bad args");
    if (n == 0) return FQSS_OK;
    hipLaunchKernelGGL(k_embedding_bwd, dim3(flat_grid(n * D)), dim3(256), 0, (hipStream_t)stream, g, idx, gw, n, D, V);
    return launch_status("fqss_embedding_bwd");
}
