// fused_q.hip -- "codes-only" streaming layers of the student in the quantizing phase.
//
// Every LayerQ output is a per-tensor 8-bit code c with (delta, min); its fp32 value delta*c + min is
// recomputed on load (bit-identical to what the producer's epilogue would have stored).  So a layer
//   reads  u8 codes (1 B/elem instead of 4),
//   fuses  its own non-linearity + fake-quant epilogue and writes u8 codes (1 B/elem instead of 4+4+1),
//   saves  nothing for the backward: the pre-quant value z is recomputed there from the input codes.
// GroupNormQ drops from 53 to 17 HBM bytes per element (fwd+bwd), the depthwise Conv1dNlQ from 45 to 24.
//
//   fqss_gnq_fwd / fqss_gnq_bwd      GroupNorm(1,C) + fake-quant          (qat_layers.py:445-448, qat_quant.py:136-147)
//   fqss_gnq_bwd_p                   ... whose second pass also runs the PRODUCING conv's epilogue backward
//   fqss_dwq_fwd / fqss_dwq_bwd      depthwise dilated conv + PReLU + fake-quant (convtasnetq.py:28-30); the backward is
//                                    one launch, one workgroup per row with gz in LDS (dwq_bwd_z / bwd_w: long rows)
//   fqss_ewq_fwd / fqss_ewq_bwd(_p)  AddQ / NlQ / residual Sub on codes (qat_layers.py:69-71, 511-518, 1193); _p also
//                                    runs the epilogue backward of the convs that produced the operands
//   fqss_decode                      codes -> fp32 (fallback for consumers without a coded-input kernel)
// Statistics of a coded tensor are exact integer sums (sum c, sum c^2 in int64).
#include <cstdlib>
#include <type_traits>

#define FQSS_USES_GRAD_ADD   // the fp32 gradient atomics of this file go through grad_add (fqss_dev.h: FQSS_DETERMINISTIC=1)
#include "fqss_dev.h"

namespace fqss {

constexpr int kGnSlots = 64;       // partial-sum slots per sample (one per stats workgroup)
constexpr int kSlots = FQSS_GACC_SLOTS;

__device__ __forceinline__ float dec(unsigned int c, const QRange& r) { return r.delta * (float)c + r.lo; }
__device__ __forceinline__ void dec4(unsigned int w, const QRange& r, float (&v)[4]) {
    v[0] = dec(w & 255u, r);          // v_cvt_f32_ubyte0..3 + mul + add (two roundings, like the reference)
    v[1] = dec((w >> 8) & 255u, r);
    v[2] = dec((w >> 16) & 255u, r);
    v[3] = dec(w >> 24, r);
}

// ---------------------------------------------------------------------------------------------
// codes -> fp32
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_decode(const uint8_t* __restrict__ c, float* __restrict__ out, int64_t rows,
                                                 int64_t cols, int64_t ld_c, int64_t ld_o, const float* qmin,
                                                 const float* qmax) {
    const QRange r = load_qrange(qmin, qmax);
    const int64_t cstep = (int64_t)gridDim.x * 256 * 4;
    for (int64_t row = blockIdx.y; row < rows; row += gridDim.y)
        for (int64_t c0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; c0 < cols; c0 += cstep) {
            const unsigned int w = *reinterpret_cast<const unsigned int*>(c + row * ld_c + c0);
            *reinterpret_cast<float4*>(out + row * ld_o + c0) =
                make_float4(dec(w & 255u, r), dec((w >> 8) & 255u, r), dec((w >> 16) & 255u, r), dec(w >> 24, r));
        }
}

// ---------------------------------------------------------------------------------------------
// GroupNorm(1, C) on coded input
// ---------------------------------------------------------------------------------------------
// stats: exact integer partial sums of one sample per workgroup slot: ws[(b*gridDim.x + blk)*2 + {0,1}] (int64)
__global__ __launch_bounds__(256) void k_gnq_stats(const uint8_t* __restrict__ xc, int C, int M, int64_t ld_c,
                                                    long long* ws) {
    __shared__ long long red[2 * 4];
    const int b = blockIdx.y;
    long long s = 0, ss = 0;
    for (int c = blockIdx.x; c < C; c += gridDim.x) {
        const uint8_t* xr = xc + ((int64_t)b * C + c) * ld_c;
        for (int m = threadIdx.x * 16; m < M; m += 256 * 16) {
            const uint4 v = *reinterpret_cast<const uint4*>(xr + m);
            const unsigned int w[4] = {v.x, v.y, v.z, v.w};
            unsigned int ls = 0, lss = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned int cv = (m + 4 * q + e < M) ? ((w[q] >> (8 * e)) & 255u) : 0u;
                    ls += cv;
                    lss += cv * cv;
                }
            s += ls;
            ss += lss;
        }
    }
    long long v[2] = {s, ss};
    block_sum<long long, 2>(v, red);
    if (threadIdx.x == 0) {
        ws[((int64_t)b * gridDim.x + blockIdx.x) * 2] = v[0];
        ws[((int64_t)b * gridDim.x + blockIdx.x) * 2 + 1] = v[1];
    }
}

// mean / rstd of the DECODED tensor of sample b from the exact integer partial sums ws[b][nslots][2] (written by k_gnq_stats or by
// the epilogue of the kernel that produced the codes: q-GEMM, depthwise layer).  Every workgroup of the apply pass does this for
// its own sample (nslots <= 1024: 16 KB from L2): no separate finalize launch.  Block-uniform result in mr[0..1].
__device__ __forceinline__ void gnq_sample_stats(const long long* __restrict__ ws, int nslots, int b, int64_t n, float eps,
                                                 const QRange& rx, long long* red, float* mr) {
    long long v[2] = {0, 0};
    for (int i = threadIdx.x; i < nslots; i += 256) {
        v[0] += ws[((int64_t)b * nslots + i) * 2];
        v[1] += ws[((int64_t)b * nslots + i) * 2 + 1];
    }
    block_sum<long long, 2>(v, red);
    if (threadIdx.x == 0) {
        const double mc = (double)v[0] / (double)n;
        double vc = (double)v[1] / (double)n - mc * mc;
        if (vc < 0.0) vc = 0.0;
        const double d = (double)rx.delta;
        mr[0] = (float)(d * mc + (double)rx.lo);
        mr[1] = (float)(1.0 / sqrt(d * d * vc + (double)eps));
    }
    __syncthreads();
}

// y = fq( decode(x)*scale + shift ) -> codes (and fp32 out when asked); 16 elements per thread
__global__ __launch_bounds__(256) void k_gnq_apply(const uint8_t* __restrict__ xc, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, uint8_t* __restrict__ yc,
                                                    float* __restrict__ yout, float* __restrict__ mean_rstd,
                                                    const long long* __restrict__ ws, int nslots, float eps, int B, int C,
                                                    int M, int64_t ld_xc, int64_t ld_yc, int64_t ld_o, const float* qmin_x,
                                                    const float* qmax_x, const float* qmin, const float* qmax, int rpw) {
    __shared__ long long red[2 * 4];
    __shared__ float mr[2];
    const QRange rx = load_qrange(qmin_x, qmax_x), ry = load_qrange(qmin, qmax);
    const int rows = B * C;
    const int cstep = gridDim.x * 256 * 16;
    int b_have = -1;
    // a workgroup owns `rpw` CONSECUTIVE rows (same sample unless a sample boundary falls inside): the statistics reduction -- as
    // many instructions as the quantizing of a whole 4000-element row -- is paid once per workgroup, not once per row
    for (int r0 = blockIdx.y * rpw; r0 < rows; r0 += gridDim.y * rpw)
    for (int row = r0; row < min(rows, r0 + rpw); ++row) {
        const int b = row / C, c = row - b * C;
        // this thread's codes are requested before the statistics are reduced: the two round trips overlap
        const int c_first = (blockIdx.x * 256 + threadIdx.x) * 16;
        uint4 v0 = make_uint4(0, 0, 0, 0);
        const uint8_t* xr = xc + (int64_t)row * ld_xc;
        if (c_first < M) v0 = *reinterpret_cast<const uint4*>(xr + c_first);
        if (b != b_have) {    // block-uniform
            gnq_sample_stats(ws, nslots, b, (int64_t)C * M, eps, rx, red, mr);
            b_have = b;
            if (c == 0 && blockIdx.x == 0 && threadIdx.x == 0) {   // saved for the backward
                mean_rstd[2 * b] = mr[0];
                mean_rstd[2 * b + 1] = mr[1];
            }
        }
        const float mean = mr[0], rstd = mr[1];
        const float scale = rstd * gamma[c];
        const float shift = fmaf(-scale, mean, beta[c]);
        for (int c0 = c_first; c0 < M; c0 += cstep) {
            const uint4 v = (c0 == c_first) ? v0 : *reinterpret_cast<const uint4*>(xr + c0);
            const unsigned int w[4] = {v.x, v.y, v.z, v.w};
            unsigned int o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float xv[4], cq[4];
                dec4(w[q], rx, xv);
                unsigned int pk = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    cq[e] = fq_code(fmaf(xv[e], scale, shift), ry);
                    pk = pack_code(cq[e], e, pk);
                }
                o[q] = pk;
                if (yout != nullptr && c0 + 4 * q < M)   // (block-uniform pointer test: the fp32 copy costs nothing when not asked for)
                    *reinterpret_cast<float4*>(yout + (int64_t)row * ld_o + c0 + 4 * q) =
                        make_float4(ry.delta * cq[0] + ry.lo, ry.delta * cq[1] + ry.lo, ry.delta * cq[2] + ry.lo, ry.delta * cq[3] + ry.lo);
            }
            *reinterpret_cast<uint4*>(yc + (int64_t)row * ld_yc + c0) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}

// The same layer as a TABLE LOOKUP (round 5).  Per (b, c) row `scale` / `shift` are constants and the input is a u8 code, so the
// output code is a 256-entry function of the input code: thread t evaluates fq_code(fmaf(dec(t), scale, shift), ry) ONCE per row for
// code t = threadIdx.x (the arithmetic of k_gnq_apply, hence bit-identical), the bytes go to a 256-B LDS table (64 dwords = one per
// bank: any two lanes on one bank read the same dword, which the LDS broadcasts -- conflict-free for every input), and the row is
// out = T[in]: bfe + ds_read_u8 + or per element instead of ~25 VALU instructions (these kernels are VALU-issue bound, docs/history/DESIGN_rounds_1-5.md 4).
// A workgroup owns RPW consecutive rows of ONE sample (host: C % RPW == 0): all their code loads are issued first, the tables of
// all RPW rows are computed while they fly, one barrier.  Rows longer than 4096 positions loop.
template <int RPW>
__global__ __launch_bounds__(256) void k_gnq_apply_t(const uint8_t* __restrict__ xc, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, uint8_t* __restrict__ yc,
                                                      float* __restrict__ yout, float* __restrict__ mean_rstd,
                                                      const long long* __restrict__ ws, int nslots, float eps, int B, int C,
                                                      int M, int64_t ld_xc, int64_t ld_yc, int64_t ld_o, const float* qmin_x,
                                                      const float* qmax_x, const float* qmin, const float* qmax) {
    __shared__ long long red[2 * 4];
    __shared__ float mr[2];
    __shared__ __attribute__((aligned(16))) uint8_t tab[RPW][256];
    const QRange rx = load_qrange(qmin_x, qmax_x), ry = load_qrange(qmin, qmax);
    const int row0 = blockIdx.x * RPW;                 // rows row0 .. row0 + RPW - 1, all of sample b
    const int b = row0 / C, c0 = row0 - b * C;
    const int m_first = threadIdx.x * 16;
    uint4 v0[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        v0[r] = make_uint4(0, 0, 0, 0);
        if (m_first < M) v0[r] = *reinterpret_cast<const uint4*>(xc + (int64_t)(row0 + r) * ld_xc + m_first);
    }
    gnq_sample_stats(ws, nslots, b, (int64_t)C * M, eps, rx, red, mr);
    if (c0 == 0 && threadIdx.x == 0) {   // saved for the backward
        mean_rstd[2 * b] = mr[0];
        mean_rstd[2 * b + 1] = mr[1];
    }
    const float mean = mr[0], rstd = mr[1];
    const float xt = dec(threadIdx.x, rx);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const float scale = rstd * gamma[c0 + r];
        const float shift = fmaf(-scale, mean, beta[c0 + r]);
        tab[r][threadIdx.x] = (uint8_t)fq_code(fmaf(xt, scale, shift), ry);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const uint8_t* xr = xc + (int64_t)(row0 + r) * ld_xc;
        uint8_t* yr = yc + (int64_t)(row0 + r) * ld_yc;
        for (int m = m_first; m < M; m += 4096) {
            const uint4 v = (m == m_first) ? v0[r] : *reinterpret_cast<const uint4*>(xr + m);
            const unsigned int w[4] = {v.x, v.y, v.z, v.w};
            unsigned int o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned int t0 = tab[r][w[q] & 255u], t1 = tab[r][(w[q] >> 8) & 255u], t2 = tab[r][(w[q] >> 16) & 255u],
                                   t3 = tab[r][w[q] >> 24];
                o[q] = t0 | (t1 << 8) | (t2 << 16) | (t3 << 24);
                if (yout != nullptr && m + 4 * q < M)   // (block-uniform pointer test)
                    *reinterpret_cast<float4*>(yout + (int64_t)(row0 + r) * ld_o + m + 4 * q) =
                        make_float4(ry.delta * (float)t0 + ry.lo, ry.delta * (float)t1 + ry.lo, ry.delta * (float)t2 + ry.lo,
                                    ry.delta * (float)t3 + ry.lo);
            }
            *reinterpret_cast<uint4*>(yr + m) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}

// the per-sample coefficients of a GroupNormQ's backward apply pass from its two C-term sums (every consumer workgroup reduces the C pairs
// of its sample itself; finishing them once in the PRODUCER of the row sums behind a ticket per row was tried in round 6 and costs the
// producers ten times what the consumers gain: profiles/r06_dwb_stamps.txt)
__device__ __forceinline__ void gn_coef_from_sums(const double (&sv)[2], float mean, float rstd, int C, int M, double& c2d, double& c3d) {
    const double md = (double)mean, rd = (double)rstd;
    const double inv_n = 1.0 / ((double)C * (double)M);
    c2d = (sv[1] * md - sv[0]) * rd * rd * rd * inv_n;
    c3d = -c2d * md - sv[1] * rd * inv_n;
}

// backward pass 1: per (b,c) row: recompute z and the STE, ds = sum gz*x, db = sum gz; range partials to gacc slots
__global__ __launch_bounds__(256) void k_gnq_bwd_rows(const uint8_t* __restrict__ xc, const float* __restrict__ g,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ mean_rstd, int C, int M, int64_t ld_xc,
                                                       int64_t ld_g, double* ws, const float* qmin_x, const float* qmax_x,
                                                       const float* qmin, const float* qmax, double* gacc) {
    __shared__ float redf[4 * 4];
    const QRange rx = load_qrange(qmin_x, qmax_x), ry = load_qrange(qmin, qmax);
    const int b = blockIdx.y, c = blockIdx.x;
    const int64_t row = (int64_t)b * C + c;
    const float mean = mean_rstd[2 * b], rstd = mean_rstd[2 * b + 1];
    const float scale = rstd * gamma[c];
    const float shift = fmaf(-scale, mean, beta[c]);
    const uint8_t* xr = xc + row * ld_xc;
    const float* gr = g + row * ld_g;
    float ds = 0.f, db = 0.f, p_du = 0.f, p_out = 0.f;
    // 4 float4 groups per thread and pass, all loads issued before the first is consumed (one exposed HBM round
    // trip per pass instead of one per group)
    for (int m0 = threadIdx.x * 4; m0 < M; m0 += 4096) {
        unsigned int wq_[4];
        float4 gq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + 1024 * i;
            if (m < M) {
                wq_[i] = *reinterpret_cast<const unsigned int*>(xr + m);
                gq[i] = *reinterpret_cast<const float4*>(gr + m);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + 1024 * i;
            if (m >= M) break;
            const unsigned int w = wq_[i];
            const float gv[4] = {gq[i].x, gq[i].y, gq[i].z, gq[i].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (m + e < M) {
                    const float x = dec((w >> (8 * e)) & 255u, rx);
                    const float z = fmaf(x, scale, shift);
                    float cq, u;
                    bool inr;
                    (void)fq_asym(z, ry, cq, u, inr);
                    const float gz = inr ? div_by(gv[e] * ry.delta, ry.delta, ry.inv) : 0.0f;
                    p_du += gv[e] * (inr ? (cq - u) : cq);
                    p_out += inr ? 0.0f : gv[e];
                    ds = fmaf(gz, x, ds);
                    db += gz;
                }
            }
        }
    }
    double v[4];
    {
        const float pf[4] = {ds, db, p_du, p_out};
        block_sum_f32w<4>(pf, redf, v);
    }
    if (threadIdx.x == 0) {
        ws[2 * row] = v[0];
        ws[2 * row + 1] = v[1];
        double* slot = gacc + 3 * (row % kSlots);
        const double dmax = v[2] / 255.0;
        atomicAdd(&slot[0], v[3] - dmax);   // rows share slots modulo kSlots: few adders per address, order-insensitive in fp64
        atomicAdd(&slot[1], dmax);
    }
}

// The layer that PRODUCED this GroupNorm's input (a 1x1 conv + PReLU + fake-quant) needs, for its own backward, exactly
// the gradient computed here pushed through its output quantizer (STE) and non-linearity.  With `pz` given the apply
// pass does that on the spot -- it reads the producer's saved pre-quant z, writes gz of the PRODUCER instead of gx and
// accumulates the producer's range / slope / bias partials -- which removes a whole 12 B/element pass (k_actq_bwd).
struct GnProducer {
    const float* pz; int64_t ld_pz;     // producer's pre-quant output (NULL: plain gx)
    int act; const float* slope;        // its non-linearity
    const float *qmin, *qmax;           // its output quantizer (== the ranges of xc)
    double* gacc; float* gbias;         // its partial slots / bias gradient [C]
};

// backward pass 3: gx = gz*(gamma*rstd) + x*c2 + c3 with gz recomputed from (x codes, g).
// Grid (C, B): one workgroup per row, so the row constants need no integer division; a pass covers 4096 positions, 4 float4 groups
// per thread with ALL their loads issued before the first is consumed.  (The first form -- 4 elements per thread and iteration,
// row = f(blockIdx.y) with a division -- spent more than half of its VALU instructions on per-iteration overhead: these
// kernels are VALU-issue bound, measured ~5 cycles per wave instruction and SIMD.)
template <bool FUSE>
__global__ __launch_bounds__(256) void k_gnq_bwd_apply(const uint8_t* __restrict__ xc, const float* __restrict__ g,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ mean_rstd, float* __restrict__ gx, int B,
                                                        int C, int M, int64_t ld_xc, int64_t ld_g, int64_t ld_gx,
                                                        const double* ws, const float* qmin_x, const float* qmax_x,
                                                        const float* qmin, const float* qmax, GnProducer P, float* ggamma,
                                                        float* gbeta) {
    __shared__ float predf[4 * 4];
    const bool det_on = det_preload();
    const QRange rp = FUSE ? load_qrange(P.qmin, P.qmax) : QRange{0.f, 1.f, 1.f};
    const float pslope = (FUSE && P.act == FQSS_ACT_PRELU) ? *P.slope : 0.0f;
    float p_du = 0.f, p_out = 0.f, p_slope = 0.f, p_bias = 0.f;
    const QRange rx = load_qrange(qmin_x, qmax_x), ry = load_qrange(qmin, qmax);
    const int c = blockIdx.x, b = blockIdx.y;
    const int64_t row = (int64_t)b * C + c;
    const float mean = mean_rstd[2 * b], rstd = mean_rstd[2 * b + 1];
    const float scale = rstd * gamma[c];
    const float shift = fmaf(-scale, mean, beta[c]);
    // The per-sample coefficients c2, c3 need sum_c gamma_c (ds, db)[b][c] over the row sums of pass 1: every workgroup reduces the
    // C pairs of ITS sample itself (16 B x C from L2, fixed order: deterministic) instead of a separate coefficient launch; the
    // workgroups of sample 0 also finish the gamma / beta gradients of their channel (a B-term sum).
    __shared__ double cred[2 * 4];
    __shared__ float c23[2];
    float c2, c3;
    auto gamma_beta_grads = [&]() {     // the workgroups of sample 0 finish the gamma / beta gradients of their channel (a B-term sum)
        double gg = 0.0, gb = 0.0;
        for (int bb = 0; bb < B; ++bb) {
            const double ds = ws[2 * ((int64_t)bb * C + c)], db = ws[2 * ((int64_t)bb * C + c) + 1];
            gg += (ds - db * (double)mean_rstd[2 * bb]) * (double)mean_rstd[2 * bb + 1];
            gb += db;
        }
        ggamma[c] += (float)gg;
        gbeta[c] += (float)gb;
    };
    {
        double sv[2] = {0.0, 0.0};
        for (int cc = threadIdx.x; cc < C; cc += 256) {
            const double gmm = (double)gamma[cc];
            sv[0] += gmm * ws[2 * ((int64_t)b * C + cc)];
            sv[1] += gmm * ws[2 * ((int64_t)b * C + cc) + 1];
        }
        block_sum<double, 2>(sv, cred);
        if (threadIdx.x == 0) {
            double c2d, c3d;
            gn_coef_from_sums(sv, mean, rstd, C, M, c2d, c3d);
            c23[0] = (float)c2d;
            c23[1] = (float)c3d;
            if (b == 0) gamma_beta_grads();
        }
        __syncthreads();
        c2 = c23[0];
        c3 = c23[1];
    }
    const uint8_t* xr = xc + row * ld_xc;
    const float* gr = g + row * ld_g;
    const float* pzr = FUSE ? P.pz + row * P.ld_pz : nullptr;
    float* orow = gx + row * ld_gx;
    for (int m0 = threadIdx.x * 4; m0 < M; m0 += 4096) {
        unsigned int wq_[4];
        float4 gq[4], pq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + 1024 * i;
            if (m < M) {
                wq_[i] = *reinterpret_cast<const unsigned int*>(xr + m);
                gq[i] = *reinterpret_cast<const float4*>(gr + m);
                if (FUSE) pq[i] = *reinterpret_cast<const float4*>(pzr + m);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + 1024 * i;
            if (m >= M) break;
            const float gv[4] = {gq[i].x, gq[i].y, gq[i].z, gq[i].w};
            const float pzv[4] = {FUSE ? pq[i].x : 0.f, FUSE ? pq[i].y : 0.f, FUSE ? pq[i].z : 0.f, FUSE ? pq[i].w : 0.f};
            float xv[4], o[4];
            dec4(wq_[i], rx, xv);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = xv[e];
                const float z = fmaf(x, scale, shift);
                float cq, u;
                bool inr;
                (void)fq_asym(z, ry, cq, u, inr);
                const float gz = inr ? div_by(gv[e] * ry.delta, ry.delta, ry.inv) : 0.0f;
                o[e] = fmaf(gz, scale, fmaf(x, c2, c3));
                if (FUSE) {   // the producer's epilogue backward (same arithmetic as k_actq_bwd) on gj = gx
                    const bool valid = (m + e < M);
                    const float gj = valid ? o[e] : 0.0f;
                    const float t = act_apply(pzv[e], P.act, pslope);
                    float pc, pu;
                    bool pin;
                    (void)fq_asym(t, rp, pc, pu, pin);
                    const float gt = pin ? div_by(gj * rp.delta, rp.delta, rp.inv) : 0.0f;
                    p_du += gj * (pin ? (pc - pu) : pc);     // gj is 0 outside the row
                    p_out += pin ? 0.0f : gj;
                    const float gzj = act_bwd(pzv[e], gt, P.act, pslope, valid, p_slope);
                    o[e] = gzj;
                    p_bias += valid ? gzj : 0.0f;
                }
            }
            *reinterpret_cast<float4*>(orow + m) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    if (FUSE) {
        double v[4];
        {
            const float pf[4] = {p_du, p_out, p_slope, p_bias};
            block_sum_f32w<4>(pf, predf, v);
        }
        // ONE atomic per row (per-wave atomics on the same 512 addresses cost 1.6 ms/step)
        if (threadIdx.x == 0 && P.gbias != nullptr) grad_add(&P.gbias[c], (float)v[3], det_on);
        if (threadIdx.x == 0) {
            double* slot = P.gacc + 3 * (row % kSlots);
            const double dmax = v[0] / 255.0;
            atomicAdd(&slot[0], v[1] - dmax);   // rows share slots modulo kSlots: order-insensitive in fp64
            atomicAdd(&slot[1], dmax);
            if (P.act == FQSS_ACT_PRELU) atomicAdd(&slot[2], v[2]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// depthwise dilated conv + PReLU/none + fake-quant on coded input
// ---------------------------------------------------------------------------------------------
constexpr int kTaps = 8;

// the 4 codes xr[s0..s0+3] of a row for any (negative / unaligned) s0: two aligned word loads + v_alignbyte
// (bytes outside [0, ld) read as 0; the caller masks elements outside [0, M))
__device__ __forceinline__ unsigned int load_codes4(const uint8_t* __restrict__ xr, int s0, int ld) {
    const int a0 = s0 & ~3;
    const unsigned int sh = (unsigned int)(s0 - a0);
    const unsigned int lo = (a0 >= 0 && a0 < ld) ? *reinterpret_cast<const unsigned int*>(xr + a0) : 0u;
    if (sh == 0) return lo;
    const unsigned int hi = (a0 + 4 >= 0 && a0 + 4 < ld) ? *reinterpret_cast<const unsigned int*>(xr + a0 + 4) : 0u;
    return __builtin_amdgcn_alignbyte(hi, lo, sh);
}

// Row-edge groups: the 4 codes xr[s0..s0+3] for ANY s0 as four byte loads at positions clamped into [0, M) -- unconditional, so
// the compiler issues all of a thread's edge loads back to back (the conditional word loads of load_codes4 cost the first and the
// last wave of every row one exposed round trip per tap and group: 12-24 in a row, which set the run time of the whole kernel);
// the caller masks the positions outside [0, M).
__device__ __forceinline__ unsigned int load_codes4_clamped(const uint8_t* __restrict__ xr, int s0, int M) {
    unsigned int w = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = min(max(s0 + j, 0), M - 1);
        w |= (unsigned int)xr[p] << (8 * j);
    }
    return w;
}

// Interior groups (every tap of all 4 positions inside the row -- all but <= 2*pad positions per row): the 4 codes of
// a tap are one unaligned dword load (gfx950 handles the misalignment in hardware) and need no zero-padding masks;
// this path has ~1/3 of the instructions of the general one, and these kernels are VALU-issue bound.
__device__ __forceinline__ unsigned int load_codes4_unaligned(const uint8_t* __restrict__ p) {
    unsigned int w;
    __builtin_memcpy(&w, p, 4);
    return w;
}

// z[m..m+3] from the coded row (zero padding)
__device__ __forceinline__ void dwq_z4(const uint8_t* __restrict__ xr, int m, int M, int ld_c, const float* wk, int K,
                                        int dil, int pad, float bv, const QRange& rx, float (&z)[4]) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (m - pad >= 0 && m + 3 + pad < M) {
#pragma unroll
        for (int k = 0; k < kTaps; ++k) {
            if (k < K) {
                float v[4];
                dec4(load_codes4_unaligned(xr + (m + k * dil - pad)), rx, v);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = fmaf(wk[k], v[j], acc[j]);
            }
        }
    } else {
        unsigned int wv[kTaps];
#pragma unroll
        for (int k = 0; k < kTaps; ++k)
            if (k < K) wv[k] = load_codes4_clamped(xr, m + k * dil - pad, M);
#pragma unroll
        for (int k = 0; k < kTaps; ++k) {
            if (k < K) {
                const int s0 = m + k * dil - pad;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (s0 + j >= 0 && s0 + j < M) ? dec((wv[k] >> (8 * j)) & 255u, rx) : 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = fmaf(wk[k], v[j], acc[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) z[j] = acc[j] + bv;
}

// 16 outputs per thread.  Interior threads (every tap of all 16 positions inside the row) fetch each tap's 16 codes with ONE
// unaligned 16-B load, all taps requested before the first is decoded: one exposed HBM round trip per thread (the first form,
// four sequential groups of per-tap dword loads, ran at 1.2 TB/s: latency-bound).  `stats` (optional): exact integer (sum c,
// sum c^2) of the output codes, one slot per (row, column chunk): stats[((b*C + c)*gridDim.x + blockIdx.x)*2 + {0,1}] -- the
// GroupNorm that consumes the codes then needs no statistics pass of its own.
template <int KT>   // taps known at compile time (3 on the training path) or 0: runtime K <= kTaps
__global__ __launch_bounds__(256) void k_dwq_fwd(const uint8_t* __restrict__ xc, const float* __restrict__ w,
                                                  const float* __restrict__ bias, uint8_t* __restrict__ yc,
                                                  float* __restrict__ yout, int rows, int C, int M, int K, int dil, int pad,
                                                  int ld_xc, int ld_yc, int ld_o, int act, const float* slope_p,
                                                  const float* qmin_x, const float* qmax_x, const float* qmin,
                                                  const float* qmax, long long* stats) {
    constexpr int NT = KT ? KT : kTaps;
    if (KT) K = KT;
    __shared__ unsigned int sred[2 * 4];
    const QRange rx = load_qrange(qmin_x, qmax_x), ry = load_qrange(qmin, qmax);
    const float slope = (act == FQSS_ACT_PRELU) ? *slope_p : 0.0f;
    for (int row = blockIdx.y; row < rows; row += gridDim.y) {
        const int c = row % C;
        float wk[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k) wk[k] = (k < K) ? w[c * K + k] : 0.0f;
        const float bv = bias ? bias[c] : 0.0f;
        const uint8_t* xr = xc + (int64_t)row * ld_xc;
        uint8_t* yr = yc + (int64_t)row * ld_yc;
        unsigned int st_s = 0, st_ss = 0;
        for (int m0 = (blockIdx.x * 256 + threadIdx.x) * 16; m0 < M; m0 += gridDim.x * 256 * 16) {
            const bool inner = (m0 - pad >= 0) && (m0 + 15 + pad < M);
            uint4 cw[NT];
            if (inner) {
#pragma unroll
                for (int k = 0; k < NT; ++k)
                    if (k < K) __builtin_memcpy(&cw[k], xr + (m0 + k * dil - pad), 16);
            }
            unsigned int o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + 4 * q;
                float z[4], fo[4];
                if (inner) {
                    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < NT; ++k) {
                        if (k < K) {
                            const unsigned int wq4[4] = {cw[k].x, cw[k].y, cw[k].z, cw[k].w};
                            float v[4];
                            dec4(wq4[q], rx, v);
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[j] = fmaf(wk[k], v[j], acc[j]);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) z[j] = acc[j] + bv;
                } else {
                    dwq_z4(xr, m, M, ld_xc, wk, K, dil, pad, bv, rx, z);
                }
                unsigned int pk = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    fo[j] = fq_code(act_apply(z[j], act, slope), ry);
                    pk = pack_code(fo[j], j, pk);
                }
                o[q] = pk;
                // statistics over the positions < M only (the last group of a row may hang over)
                const unsigned int live = (m + 3 < M) ? 0xFFFFFFFFu : ((m < M) ? (0xFFFFFFFFu >> (8 * (4 - (M - m)))) : 0u);
                code_stats4(pk & live, st_s, st_ss);
                if (yout != nullptr && m < M)
                    *reinterpret_cast<float4*>(yout + (int64_t)row * ld_o + m) =
                        make_float4(ry.delta * fo[0] + ry.lo, ry.delta * fo[1] + ry.lo, ry.delta * fo[2] + ry.lo, ry.delta * fo[3] + ry.lo);
            }
            *reinterpret_cast<uint4*>(yr + m0) = make_uint4(o[0], o[1], o[2], o[3]);
        }
        if (stats != nullptr) {   // workgroup-uniform.  < 2^32: a workgroup's share of a row is at most 2^16 positions (host check)
            unsigned int v[2] = {st_s, st_ss};
            block_sum<unsigned int, 2>(v, sred);
            if (threadIdx.x == 0) {
                long long* slot = stats + 2 * ((int64_t)row * gridDim.x + blockIdx.x);
                slot[0] = (long long)v[0];
                slot[1] = (long long)v[1];
            }
        }
    }
}

#ifdef FQSS_EXPERIMENTS   // round-4 experiment, measured slower: built by `make experiments` only (declared in include/fqss_experiments.h)
// gLN + fake-quant FOLLOWED BY depthwise conv + PReLU + fake-quant, both on codes, as ONE launch (round 4; the teacher's k_tdw does the
// same in float): a workgroup takes `rpw` consecutive rows; per row it normalises + re-quantises the 16 input codes of every thread
// (k_gnq_apply's arithmetic, codes written to HBM -- the backward of both layers reads them -- AND to an LDS row), then runs the
// 3-tap FIR + PReLU + quantizer of k_dwq_fwd<3> out of that LDS row.  One launch and one 4-KB re-read per row less; every value is
// computed by the op sequences of the two kernels it replaces (gate: tests/test_gpu_kernels.py::test_gn_dw_fused_bit_identical).
// Rows of at most 4096 positions (256 threads x 16), K = 3, statistics of the INPUT codes supplied by their producer.
__global__ __launch_bounds__(256) void k_gndwq_fwd(const uint8_t* __restrict__ xc, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    uint8_t* __restrict__ y1, float* __restrict__ mean_rstd, const long long* __restrict__ ws,
                                                    int nslots, float eps, int B, int C, int M, int64_t ld_xc, int64_t ld_y1,
                                                    const float* qmin_x, const float* qmax_x, const float* qmin1, const float* qmax1,
                                                    const float* __restrict__ w, const float* __restrict__ bias, int dil, int pad, int act,
                                                    const float* slope_p, uint8_t* __restrict__ y2, int64_t ld_y2, const float* qmin2,
                                                    const float* qmax2, long long* stats2, int rpw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rowbuf[];      // [padl + 4096 + pad + 16]
    __shared__ long long red[2 * 4];
    __shared__ float mr[2];
    __shared__ unsigned int sred[2 * 4];
    const QRange rx = load_qrange(qmin_x, qmax_x), r1 = load_qrange(qmin1, qmax1), r2 = load_qrange(qmin2, qmax2);
    const float slope = (act == FQSS_ACT_PRELU) ? *slope_p : 0.0f;
    const int rows = B * C, padl = (pad + 15) & ~15;
    const int m0 = threadIdx.x * 16;
    int b_have = -1;
    for (int r0 = blockIdx.x * rpw; r0 < rows; r0 += gridDim.x * rpw)
    for (int row = r0; row < min(rows, r0 + rpw); ++row) {
        const int b = row / C, c = row - b * C;
        uint4 v0 = make_uint4(0, 0, 0, 0);
        if (m0 < M) v0 = *reinterpret_cast<const uint4*>(xc + (int64_t)row * ld_xc + m0);
        if (b != b_have) {    // block-uniform
            gnq_sample_stats(ws, nslots, b, (int64_t)C * M, eps, rx, red, mr);
            b_have = b;
            if (c == 0 && threadIdx.x == 0) {   // saved for the GroupNorm's backward
                mean_rstd[2 * b] = mr[0];
                mean_rstd[2 * b + 1] = mr[1];
            }
        }
        const float scale = mr[1] * gamma[c];
        const float shift = fmaf(-scale, mr[0], beta[c]);
        if (m0 < M) {        // ---- GroupNorm + quantizer: k_gnq_apply's arithmetic
            const unsigned int wv[4] = {v0.x, v0.y, v0.z, v0.w};
            unsigned int o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float xv[4];
                dec4(wv[q], rx, xv);
                unsigned int pk = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) pk = pack_code(fq_code(fmaf(xv[e], scale, shift), r1), e, pk);
                o[q] = pk;
            }
            const uint4 ov = make_uint4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<uint4*>(y1 + (int64_t)row * ld_y1 + m0) = ov;
            *reinterpret_cast<uint4*>(rowbuf + padl + m0) = ov;
        }
        __syncthreads();
        // ---- depthwise 3-tap FIR + PReLU + quantizer: k_dwq_fwd<3>'s arithmetic, taps from the LDS row
        float wk[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) wk[k] = w[c * 3 + k];
        const float bv = bias ? bias[c] : 0.0f;
        unsigned int st_s = 0, st_ss = 0;
        if (m0 < M) {
            const bool inner = (m0 - pad >= 0) && (m0 + 15 + pad < M);
            unsigned int o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + 4 * q;
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int s0 = m + k * dil - pad;
                    float v[4];
                    if (inner) {
                        const int a = padl + s0;                               // >= 0: inner
                        const unsigned int lo = *reinterpret_cast<const unsigned int*>(rowbuf + (a & ~3));
                        const unsigned int hi = *reinterpret_cast<const unsigned int*>(rowbuf + (a & ~3) + 4);
                        dec4(__builtin_amdgcn_alignbyte(hi, lo, (unsigned)(a & 3)), r1, v);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = (s0 + j >= 0 && s0 + j < M) ? dec((unsigned int)rowbuf[padl + s0 + j], r1) : 0.0f;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = fmaf(wk[k], v[j], acc[j]);
                }
                unsigned int pk = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) pk = pack_code(fq_code(act_apply(acc[j] + bv, act, slope), r2), j, pk);
                o[q] = pk;
                const unsigned int live = (m + 3 < M) ? 0xFFFFFFFFu : ((m < M) ? (0xFFFFFFFFu >> (8 * (4 - (M - m)))) : 0u);
                code_stats4(pk & live, st_s, st_ss);
            }
            *reinterpret_cast<uint4*>(y2 + (int64_t)row * ld_y2 + m0) = make_uint4(o[0], o[1], o[2], o[3]);
        }
        if (stats2 != nullptr) {   // one slot per row; block_sum's barriers also protect the LDS row against the next row's stores
            unsigned int v[2] = {st_s, st_ss};
            block_sum<unsigned int, 2>(v, sred);
            if (threadIdx.x == 0) {
                stats2[2 * (int64_t)row] = (long long)v[0];
                stats2[2 * (int64_t)row + 1] = (long long)v[1];
            }
        } else {
            __syncthreads();
        }
    }
}
#endif  // FQSS_EXPERIMENTS

#ifdef FQSS_EXPERIMENTS   // round-5 experiment (the table form of the fused forward): bit-identical, STILL slower than the two launches (39.6 vs 33.5 us)
// gLN + fake-quant FOLLOWED BY the 3-tap depthwise conv + PReLU + fake-quant, codes -> codes -> codes, as ONE launch -- round 5, on
// code tables (the round-4 form above computed both layers per element and lost to the two launches: both halves were bound by vector
// issue).  Per (b, c) row the GroupNorm's output code is a function T of the input code (k_gnq_apply_t), and what the FIR needs of it
// -- its de-quantised value -- is another: V[c] = delta1 * T[c] + min1.  A workgroup takes RPW consecutive rows of one sample; per row
// thread t evaluates T[t] and V[t] once (256-B + 1-KB LDS tables), the row's INPUT codes go to an LDS row, and then
//   y1[m] = T[x[m]]                                   (one byte lookup per element, stored for the backward of both layers)
//   z[m]  = fma(w2, V[x[m + d]], fma(w1, V[x[m]], w0 * V[x[m - d]])) + b     (three float lookups instead of three decodes)
//   y2[m] = fq(PReLU(z[m]))                           (k_dwq_fwd<3>'s arithmetic), + the integer statistics of y2 for the next gLN.
// Every value is what fqss_gnq_fwd + fqss_dwq_fwd compute (tests/test_gpu_kernels.py::test_gn_dw_fused_bit_identical); ~20 vector
// instructions per element instead of 3 + 27, one launch and one 16-MB re-read less.  Rows of at most 4096 positions, K = 3.
__global__ __launch_bounds__(256) void k_gndwq_fwd_t(const uint8_t* __restrict__ xc, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      uint8_t* __restrict__ y1, float* __restrict__ mean_rstd, const long long* __restrict__ ws,
                                                      int nslots, float eps, int B, int C, int M, int64_t ld_xc, int64_t ld_y1,
                                                      const float* qmin_x, const float* qmax_x, const float* qmin1, const float* qmax1,
                                                      const float* __restrict__ w, const float* __restrict__ bias, int dil, int pad, int act,
                                                      const float* slope_p, uint8_t* __restrict__ y2, int64_t ld_y2, const float* qmin2,
                                                      const float* qmax2, long long* stats2, int rpw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rowbuf[];      // [padl + 4096 + pad + 16]: the row's INPUT codes
    __shared__ long long red[2 * 4];
    __shared__ float mr[2];
    __shared__ unsigned int sred[2 * 4];
    __shared__ __attribute__((aligned(16))) uint8_t tabC[256];
    __shared__ __attribute__((aligned(16))) float tabV[256];
    const QRange rx = load_qrange(qmin_x, qmax_x), r1 = load_qrange(qmin1, qmax1), r2 = load_qrange(qmin2, qmax2);
    const float slope = (act == FQSS_ACT_PRELU) ? *slope_p : 0.0f;
    const int rows = B * C, padl = (pad + 15) & ~15;
    const int m0 = threadIdx.x * 16;
    const float xt = dec(threadIdx.x, rx);
    int b_have = -1;
    for (int r0 = blockIdx.x * rpw; r0 < rows; r0 += gridDim.x * rpw)
    for (int row = r0; row < min(rows, r0 + rpw); ++row) {
        const int b = row / C, c = row - b * C;
        uint4 v0 = make_uint4(0, 0, 0, 0);
        if (m0 < M) v0 = *reinterpret_cast<const uint4*>(xc + (int64_t)row * ld_xc + m0);
        if (b != b_have) {    // block-uniform
            gnq_sample_stats(ws, nslots, b, (int64_t)C * M, eps, rx, red, mr);
            b_have = b;
            if (c == 0 && threadIdx.x == 0) {   // saved for the GroupNorm's backward
                mean_rstd[2 * b] = mr[0];
                mean_rstd[2 * b + 1] = mr[1];
            }
        }
        const float scale = mr[1] * gamma[c];
        const float shift = fmaf(-scale, mr[0], beta[c]);
        {
            const float c1 = fq_code(fmaf(xt, scale, shift), r1);      // k_gnq_apply's arithmetic, once per code
            tabC[threadIdx.x] = (uint8_t)c1;
            tabV[threadIdx.x] = r1.delta * c1 + r1.lo;                   // dec() of that code, as k_dwq_fwd reads it
        }
        if (m0 < M) *reinterpret_cast<uint4*>(rowbuf + padl + m0) = v0;
        __syncthreads();
        float wk[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) wk[k] = w[c * 3 + k];
        const float bv = bias ? bias[c] : 0.0f;
        unsigned int st_s = 0, st_ss = 0;
        if (m0 < M) {
            {   // ---- the GroupNorm's output codes (for the backward of both layers)
                const unsigned int wv[4] = {v0.x, v0.y, v0.z, v0.w};
                unsigned int o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    o[q] = (unsigned int)tabC[wv[q] & 255u] | ((unsigned int)tabC[(wv[q] >> 8) & 255u] << 8) |
                           ((unsigned int)tabC[(wv[q] >> 16) & 255u] << 16) | ((unsigned int)tabC[wv[q] >> 24] << 24);
                *reinterpret_cast<uint4*>(y1 + (int64_t)row * ld_y1 + m0) = make_uint4(o[0], o[1], o[2], o[3]);
            }
            const bool inner = (m0 - pad >= 0) && (m0 + 15 + pad < M);
            unsigned int o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + 4 * q;
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int s0 = m + k * dil - pad;
                    float v[4];
                    if (inner) {
                        const int a = padl + s0;                               // >= 0: inner
                        const unsigned int lo = *reinterpret_cast<const unsigned int*>(rowbuf + (a & ~3));
                        const unsigned int hi = *reinterpret_cast<const unsigned int*>(rowbuf + (a & ~3) + 4);
                        const unsigned int cw = __builtin_amdgcn_alignbyte(hi, lo, (unsigned)(a & 3));
                        v[0] = tabV[cw & 255u]; v[1] = tabV[(cw >> 8) & 255u]; v[2] = tabV[(cw >> 16) & 255u]; v[3] = tabV[cw >> 24];
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = (s0 + j >= 0 && s0 + j < M) ? tabV[rowbuf[padl + s0 + j]] : 0.0f;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = fmaf(wk[k], v[j], acc[j]);
                }
                unsigned int pk = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) pk = pack_code(fq_code(act_apply(acc[j] + bv, act, slope), r2), j, pk);
                o[q] = pk;
                const unsigned int live = (m + 3 < M) ? 0xFFFFFFFFu : ((m < M) ? (0xFFFFFFFFu >> (8 * (4 - (M - m)))) : 0u);
                code_stats4(pk & live, st_s, st_ss);
            }
            *reinterpret_cast<uint4*>(y2 + (int64_t)row * ld_y2 + m0) = make_uint4(o[0], o[1], o[2], o[3]);
        }
        if (stats2 != nullptr) {   // one slot per row; block_sum's barriers also protect the LDS row and tables against the next row's stores
            unsigned int v[2] = {st_s, st_ss};
            block_sum<unsigned int, 2>(v, sred);
            if (threadIdx.x == 0) {
                stats2[2 * (int64_t)row] = (long long)v[0];
                stats2[2 * (int64_t)row + 1] = (long long)v[1];
            }
        } else {
            __syncthreads();
        }
    }
}

#endif  // FQSS_EXPERIMENTS

// backward: recompute z, STE + PReLU -> gz (fp32), bias row-sums, range/slope partials (gacc slots)
template <int NQ>   // float4 groups per thread and pass: a workgroup covers NQ * 1024 consecutive positions
__global__ __launch_bounds__(256) void k_dwq_bwd_z(const uint8_t* __restrict__ xc, const float* __restrict__ w,
                                                    const float* __restrict__ bias, const float* __restrict__ g,
                                                    float* __restrict__ gz, int B, int C, int M, int K, int dil, int pad,
                                                    int64_t ld_xc, int64_t ld_g, int64_t ld_gz, int act, const float* slope_p,
                                                    const float* qmin_x, const float* qmax_x, const float* qmin,
                                                    const float* qmax, double* gacc, float* gbias) {
    __shared__ double red[3 * 4];
    __shared__ float redf[4];
    const QRange rx = load_qrange(qmin_x, qmax_x), ry = load_qrange(qmin, qmax);
    const float slope = (act == FQSS_ACT_PRELU) ? *slope_p : 0.0f;
    const int cstep = gridDim.x * 1024 * NQ;
    float p_du = 0.f, p_out = 0.f, p_slope = 0.f;
    for (int c = blockIdx.y; c < C; c += gridDim.y) {
        float wk[kTaps];
#pragma unroll
        for (int k = 0; k < kTaps; ++k) wk[k] = (k < K) ? w[c * K + k] : 0.0f;
        const float bv = bias ? bias[c] : 0.0f;
        float p_bias = 0.f;
        for (int b = 0; b < B; ++b) {
            const int64_t row = (int64_t)b * C + c;
            const uint8_t* xr = xc + row * ld_xc;
            // consecutive lanes own consecutive float4 of g / gz (1-KiB wave accesses) -- 16 consecutive positions
            // per lane made every fp32 access 64-B strided (measured: 2.1x write amplification)
            for (int m0 = blockIdx.x * 1024 * NQ; m0 < M; m0 += cstep)
#pragma unroll
            for (int qq = 0; qq < NQ; ++qq) {
                const int m = m0 + 1024 * qq + 4 * threadIdx.x;
                if (m >= M) break;
                float z[4], o[4];
                dwq_z4(xr, m, M, (int)ld_xc, wk, K, dil, pad, bv, rx, z);
                const float4 gv4 = *reinterpret_cast<const float4*>(g + row * ld_g + m);
                const float gv[4] = {gv4.x, gv4.y, gv4.z, gv4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool valid = (m + j < M);
                    const float gj = valid ? gv[j] : 0.0f;
                    const float t = act_apply(z[j], act, slope);
                    float cq, u;
                    bool inr;
                    (void)fq_asym(t, ry, cq, u, inr);
                    const float gt = inr ? div_by(gj * ry.delta, ry.delta, ry.inv) : 0.0f;
                    p_du += valid ? gj * (inr ? (cq - u) : cq) : 0.0f;      // (selects, not branches: fq.hip k_actq_bwd)
                    p_out += (valid && !inr) ? gj : 0.0f;
                    float gzj = act_bwd(z[j], gt, act, slope, valid, p_slope);
                    o[j] = gzj;
                    p_bias += valid ? gzj : 0.0f;
                }
                *reinterpret_cast<float4*>(gz + row * ld_gz + m) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
        if (gbias != nullptr) {
            float pb[1] = {p_bias};
            block_sum<float, 1>(pb, redf);
            if (threadIdx.x == 0) grad_add(&gbias[c], pb[0]);
        }
    }
    double v[3] = {(double)p_du, (double)p_out, (double)p_slope};
    block_sum<double, 3>(v, red);
    if (threadIdx.x == 0) {
        double* slot = gacc + 3 * ((int64_t)blockIdx.y * gridDim.x + blockIdx.x);
        const double dmax = v[0] / 255.0;
        slot[0] += v[1] - dmax;
        slot[1] += dmax;
        slot[2] += v[2];
    }
}

// ---- the whole depthwise backward of one (sample, channel) row in ONE workgroup ------------------------------
// gz (the gradient at the conv output, after the fake-quant STE and the PReLU) is needed at +-dil neighbours
// for gx and against the shifted input for gw.  A row is at most kDwRowMax positions: the workgroup keeps
// its gz in LDS instead of HBM, so the layer's backward moves  codes (1 B) + g (4 B) in, gx (4 B) out  per
// position -- instead of 9 + 8 + 5 B over three kernels.
//   phase 1 (per float4 group, consecutive lanes = consecutive groups): z from the 3 taps of the input codes,
//           STE/PReLU -> gz to LDS; partials for the bias / weight / range / slope gradients
//   phase 2: gx[m] = sum_k w[k] * gz[m + pad - k*dil] from LDS
constexpr int kDwRowMax = 12 * 1024;   // 48 KiB of LDS

// Round 5: the two GroupNormQ layers around a TCN block's depthwise layer hand half of their backward to this kernel (VERDICT r04
// next #1b -- their two-pass backward runs at the practical HBM rate, so only fewer bytes make it faster):
//   GA ("gLN after"): the GroupNormQ that CONSUMES this layer's output.  Its backward APPLY pass -- gx = gz * (gamma rstd) + x c2 + c3
//       with gz = the incoming gradient pushed through ITS output quantizer -- is a function of the incoming gradient and of the CODE
//       of this layer's output, which phase 1 recomputes anyway: the apply runs on g as it is loaded (per row a 256-entry table
//       {fma(x, c2, c3), in-range flag} indexed by that code), and k_gnq_bwd_apply<false> with its 65 MB read + 65 MB write is gone.
//       The per-sample coefficients c2 / c3 and the gamma / beta gradients are finished in this kernel's prologue exactly as there.
//   GB ("gLN before"): the GroupNormQ that PRODUCED this layer's input.  Its backward ROWS pass -- per row sum gz x, sum gz and the
//       range partials of its output quantizer, gz = gx pushed through that quantizer -- needs gx (produced here, a row per
//       workgroup) and the GroupNorm's INPUT code (one more byte per element, a 256-entry table {x, c - u | c, in-range} per row):
//       phase 2 takes the sums while gx is in registers, same thread -> element order as k_gnq_bwd_rows, and that launch (65 + 16 MB
//       read) is gone.  Both are bit-identical to the separate kernels in everything but the order of the fp64 slot atomics.
struct DwGnAfter {
    const float *gamma, *beta, *mean_rstd;   // [C], [C], [B][2] of that GroupNormQ
    const double* ws;                        // [B*C][2]: (ds, db) of its rows pass (k_gnq_bwd_rows)
    const float *qmin, *qmax;                // its OUTPUT quantizer
    float *ggamma, *gbeta;                   // [C], "+=" (workgroups of sample 0)
    int B;
};
struct DwGnBefore {
    const uint8_t* xc0; int64_t ld_xc0;      // ITS input codes [B*C][ld_xc0] ...
    const float *qmin0, *qmax0;              // ... and their range
    const float *gamma, *beta, *mean_rstd;
    double* ws;                              // out [B*C][2]: (ds, db) for its apply pass
    double* gacc;                            // partial slots of its output quantizer (= this layer's input range)
};

#ifndef FQSS_DWB_ABL
#define FQSS_DWB_ABL 0      // timing ablations (tools only): 1 no coefficient sums in the GA prologue, 2 no phase 2, 4 no final reductions, 8 no phase-1 arithmetic
#endif
// FQSS_DWB_STAMPS (tools/dwb_stamps.py only): wave 0 of every workgroup records s_memtime at its phase boundaries; the eight deltas
// (16 shader cycles per unit) replace the row's two doubles in GBd.ws
#ifdef FQSS_DWB_STAMPS
#define DWB_STAMP(k) do { if (threadIdx.x == 0) stamp_[k] = __builtin_readcyclecounter(); } while (0)
#else
#define DWB_STAMP(k) do { } while (0)
#endif
#ifndef FQSS_DWB_GPP
#define FQSS_DWB_GPP 2      // float4 groups per thread and pass (a pass = 1024 x GPP positions; the thread -> element order does not depend on it)
#endif
template <int KT, bool GA = false, bool GB = false, int ACTC = -1>   // KT: taps known at compile time (3 on the training path) or 0: runtime K <= kTaps;
                                                                       // ACTC >= 0: the activation known at compile time (PReLU in the TCN blocks)
__global__ __launch_bounds__(256) void k_dwq_bwd(const uint8_t* __restrict__ xc, const float* __restrict__ w,
                                                  const float* __restrict__ bias, const float* __restrict__ g,
                                                  float* __restrict__ gx, float* gw, int C, int M, int K, int dil, int pad,
                                                  int64_t ld_xc, int64_t ld_g, int64_t ld_gx, int act, const float* slope_p,
                                                  const float* qmin_x, const float* qmax_x, const float* qmin,
                                                  const float* qmax, double* gacc, float* gbias, int want_gx, DwGnAfter GAd,
                                                  DwGnBefore GBd) {
    constexpr int NT = KT ? KT : kTaps;
    if (KT) K = KT;
    if (ACTC >= 0) act = ACTC;
    extern __shared__ __attribute__((aligned(16))) float sgz[];   // [ceil4(M)]
    __shared__ __attribute__((aligned(16))) float redf[(8 + kTaps) * 4];
    // one 4-byte plane per table column: a lookup by a data-dependent code then touches ONE bank per lane and column (the 8-B / 16-B
    // entries made every lookup a 2- / 4-bank access: LDS conflict rate 1.59, profiles/r05_sq_counters.txt)
    __shared__ float tabA0[GA ? 256 : 1], tabA1[GA ? 256 : 1];            // GA: fma(x, c2, c3) | in-range, per code of THIS layer's output
    __shared__ float tabB0[GB ? 256 : 1], tabB1[GB ? 256 : 1], tabB2[GB ? 256 : 1];   // GB: x | c - u or c | in-range, per code of the GroupNorm's input
#ifdef FQSS_DWB_STAMPS
    unsigned long long stamp_[9];
#endif
    DWB_STAMP(0);
    const bool det_on = det_preload();
    const QRange rx = load_qrange(qmin_x, qmax_x), ry = load_qrange(qmin, qmax);
    const float slope = (act == FQSS_ACT_PRELU) ? *slope_p : 0.0f;
    const int row = blockIdx.x, c = row % C;
    // a pass covers 4096 positions: 4 float4 groups per thread, ALL their loads issued before the first is consumed
    // (the straight loop exposed one HBM round trip per group: PMC showed the waves parked 46 % of the time)
    const uint8_t* xr = xc + (int64_t)row * ld_xc;
    const float* gr = g + (int64_t)row * ld_g;
    constexpr int GPP = FQSS_DWB_GPP;
    float4 gq[GPP];
    unsigned int cw[GPP][NT];
    bool inner[GPP];
    auto issue_loads = [&](int m0) {
#pragma unroll
        for (int i = 0; i < GPP; ++i) {
            const int m = m0 + 1024 * i;
            inner[i] = (m - pad >= 0) && (m + 3 + pad < M);
            if (m < M) gq[i] = *reinterpret_cast<const float4*>(gr + m);
            if (inner[i]) {
#pragma unroll
                for (int k = 0; k < NT; ++k)
                    if (k < K) cw[i][k] = load_codes4_unaligned(xr + (m + k * dil - pad));
            } else if (m < M) {   // row edge: clamped byte loads, requested with the rest
#pragma unroll
                for (int k = 0; k < NT; ++k)
                    if (k < K) cw[i][k] = load_codes4_clamped(xr, m + k * dil - pad, M);
            }
        }
    };
    QRange r2 = QRange{0.f, 1.f, 1.f};
    float scale2 = 0.f;
    if constexpr (GA) {
        // ---- prologue of k_gnq_bwd_apply: c2 / c3 of this sample from the rows pass' sums, gamma / beta gradients by sample 0
        __shared__ double cred[2 * 4];
        __shared__ float c23[2];
        const int b = row / C;
        r2 = load_qrange(GAd.qmin, GAd.qmax);
        const float mean = GAd.mean_rstd[2 * b], rstd = GAd.mean_rstd[2 * b + 1];
        scale2 = rstd * GAd.gamma[c];
        const float shift2 = fmaf(-scale2, mean, GAd.beta[c]);
        auto gamma_beta_grads = [&]() {
            double gg = 0.0, gb = 0.0;
            for (int bb = 0; bb < GAd.B; ++bb) {
                const double ds = GAd.ws[2 * ((int64_t)bb * C + c)], db = GAd.ws[2 * ((int64_t)bb * C + c) + 1];
                gg += (ds - db * (double)GAd.mean_rstd[2 * bb]) * (double)GAd.mean_rstd[2 * bb + 1];
                gb += db;
            }
            GAd.ggamma[c] += (float)gg;
            GAd.gbeta[c] += (float)gb;
        };
        float c2v, c3v;
        {
            double sv[2] = {0.0, 0.0};
            for (int cc = threadIdx.x; cc < ((FQSS_DWB_ABL & 1) ? 0 : C); cc += 256) {
                const double gmm = (double)GAd.gamma[cc];
                sv[0] += gmm * GAd.ws[2 * ((int64_t)b * C + cc)];
                sv[1] += gmm * GAd.ws[2 * ((int64_t)b * C + cc) + 1];
            }
            block_sum<double, 2>(sv, cred);
            if (threadIdx.x == 0) {
                double c2d, c3d;
                gn_coef_from_sums(sv, mean, rstd, C, M, c2d, c3d);
                c23[0] = (float)c2d;
                c23[1] = (float)c3d;
                if (b == 0) gamma_beta_grads();
            }
            __syncthreads();
            c2v = c23[0];
            c3v = c23[1];
        }
        {   // the table over the codes of this layer's output (= that GroupNorm's input, range ry)
            const float x2 = dec(threadIdx.x, ry);
            float cq2, u2;
            bool in2;
            (void)fq_asym(fmaf(x2, scale2, shift2), r2, cq2, u2, in2);
            tabA0[threadIdx.x] = fmaf(x2, c2v, c3v);
            tabA1[threadIdx.x] = in2 ? 1.0f : 0.0f;
        }
    }
    if constexpr (GB) {
        const int b = row / C;
        const QRange r0 = load_qrange(GBd.qmin0, GBd.qmax0);
        const float mean = GBd.mean_rstd[2 * b], rstd = GBd.mean_rstd[2 * b + 1];
        const float scale1 = rstd * GBd.gamma[c];
        const float shift1 = fmaf(-scale1, mean, GBd.beta[c]);
        const float x0 = dec(threadIdx.x, r0);
        float cq1, u1;
        bool in1;
        (void)fq_asym(fmaf(x0, scale1, shift1), rx, cq1, u1, in1);     // that GroupNorm's output quantizer = this layer's input range
        tabB0[threadIdx.x] = x0;
        tabB1[threadIdx.x] = in1 ? (cq1 - u1) : cq1;
        tabB2[threadIdx.x] = in1 ? 1.0f : 0.0f;
    }
    if constexpr (GA || GB) __syncthreads();
    DWB_STAMP(1);      // prologue done
    float wk[NT], pw[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        wk[k] = (k < K) ? w[c * K + k] : 0.0f;
        pw[k] = 0.0f;
    }
    const float bv = bias ? bias[c] : 0.0f;
    float p_du = 0.f, p_out = 0.f, p_slope = 0.f, p_bias = 0.f;
    // GB: the GroupNorm-input codes of this thread's first four phase-2 groups are requested NOW (4 registers): loaded inside that
    // loop they cost one exposed memory round trip per iteration of a workgroup that lives ~20 us
    unsigned int w0pre[4] = {0u, 0u, 0u, 0u};
    if constexpr (GB) {
        const uint8_t* x0p = GBd.xc0 + (int64_t)row * GBd.ld_xc0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (4 * (int)threadIdx.x + 1024 * i < M) w0pre[i] = *reinterpret_cast<const unsigned int*>(x0p + 4 * threadIdx.x + 1024 * i);
    }

    for (int m0 = 4 * threadIdx.x; m0 < M; m0 += 1024 * GPP) {
        issue_loads(m0);
#pragma unroll
        for (int i = 0; i < GPP; ++i) {
            const int m = m0 + 1024 * i;
            if (m >= M) break;
            if constexpr (FQSS_DWB_ABL & 8) {
                float4 t = gq[i];
#pragma unroll
                for (int k = 0; k < NT; ++k) t.x += __uint_as_float(cw[i][k]);
                *reinterpret_cast<float4*>(&sgz[m]) = t;
                continue;
            }
            float v[NT][4], acc[4] = {0.f, 0.f, 0.f, 0.f};
            if (inner[i]) {   // no padding masks, one unaligned dword per tap
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    if (k < K) {
                        dec4(cw[i][k], rx, v[k]);
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[j] = fmaf(wk[k], v[k][j], acc[j]);
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    if (k < K) {
                        const int s0 = m + k * dil - pad;
                        const unsigned int cwk = cw[i][k];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            v[k][j] = (s0 + j >= 0 && s0 + j < M) ? dec((cwk >> (8 * j)) & 255u, rx) : 0.0f;
                            acc[j] = fmaf(wk[k], v[k][j], acc[j]);
                        }
                    }
                }
            }
            const float gv[4] = {gq[i].x, gq[i].y, gq[i].z, gq[i].w};
            float o[4];
            // ALL = every position of the group lies inside the row (all but the row's last group): the per-element validity selects
            // (5 of them) drop out of the instruction stream; the hand-over forms also take the branch-free activation helpers -- the
            // branching ones compiled to ~18 scalar branches per element with their exec-mask bookkeeping and register copies
            auto elems = [&](auto ALL_) {
                constexpr bool ALL = decltype(ALL_)::value;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool valid = ALL ? true : (m + j < M);
                    const float z = acc[j] + bv;
                    const float t = (GA || GB) ? act_apply(z, act, slope) : act_apply_br(z, act, slope);
                    float cq, u;
                    bool inr;
                    (void)fq_asym(t, ry, cq, u, inr);
                    float gj = valid ? gv[j] : 0.0f;
                    if constexpr (GA) {     // the consuming GroupNormQ's backward apply, on the code this layer's forward wrote
                        const unsigned int ca = (unsigned int)cq & 255u;
                        const float2 e2 = make_float2(tabA0[ca], tabA1[ca]);
                        const float gz2 = (e2.y != 0.0f) ? div_by(gv[j] * r2.delta, r2.delta, r2.inv) : 0.0f;
                        gj = valid ? fmaf(gz2, scale2, e2.x) : 0.0f;
                    }
                    const float gt = inr ? div_by(gj * ry.delta, ry.delta, ry.inv) : 0.0f;
                    p_du += valid ? gj * (inr ? (cq - u) : cq) : 0.0f;
                    p_out += (valid && !inr) ? gj : 0.0f;
                    float gzj = (GA || GB) ? act_bwd(z, gt, act, slope, valid, p_slope) : act_bwd_br(z, gt, act, slope, valid, p_slope);
                    gzj = valid ? gzj : 0.0f;
                    o[j] = gzj;
                    p_bias += gzj;
#pragma unroll
                    for (int k = 0; k < NT; ++k)
                        if (k < K) pw[k] = fmaf(gzj, v[k][j], pw[k]);   // v is 0 outside the row (zero padding)
                }
            };
            if (m + 3 < M) elems(std::true_type{});
            else elems(std::false_type{});
            *reinterpret_cast<float4*>(&sgz[m]) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    DWB_STAMP(2);      // phase 1 done (loads + arithmetic)
    __syncthreads();
    DWB_STAMP(3);      // barrier passed

    float q_ds = 0.f, q_db = 0.f, q_du = 0.f, q_out = 0.f;      // GB: the producing GroupNormQ's row sums / range partials
    if (want_gx && !(FQSS_DWB_ABL & 2)) {
        float* xo = gx + (int64_t)row * ld_gx;
        const bool aligned = ((dil & 3) == 0) && ((pad & 3) == 0);
        const uint8_t* x0r = GB ? GBd.xc0 + (int64_t)row * GBd.ld_xc0 : nullptr;
        int it = 0;
        for (int m = 4 * threadIdx.x; m < M; m += 1024, ++it) {
            unsigned int w0 = 0;
            if constexpr (GB) w0 = (it < 4) ? w0pre[it] : *reinterpret_cast<const unsigned int*>(x0r + m);
            float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                if (k < K) {
                    const int s0 = m + pad - k * dil;
                    if (aligned) {
                        if (s0 >= 0 && s0 < M) {   // whole float4 inside [0, ceil4(M)): sgz beyond M holds zeros
                            const float4 t = *reinterpret_cast<const float4*>(&sgz[s0]);
                            a[0] = fmaf(wk[k], t.x, a[0]);
                            a[1] = fmaf(wk[k], t.y, a[1]);
                            a[2] = fmaf(wk[k], t.z, a[2]);
                            a[3] = fmaf(wk[k], t.w, a[3]);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int sj = s0 + j;
                            const float t = (sj >= 0 && sj < M) ? sgz[sj] : 0.0f;
                            a[j] = fmaf(wk[k], t, a[j]);
                        }
                    }
                }
            }
            *reinterpret_cast<float4*>(xo + m) = make_float4(a[0], a[1], a[2], a[3]);
            if constexpr (GB) {     // k_gnq_bwd_rows' arithmetic on gx while it is in registers (same thread -> element order)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (m + j < M) {
                        const unsigned int cb = (w0 >> (8 * j)) & 255u;
                        const float4 e1 = make_float4(tabB0[cb], tabB1[cb], tabB2[cb], 0.0f);
                        const bool in1 = e1.z != 0.0f;
                        const float gz1 = in1 ? div_by(a[j] * rx.delta, rx.delta, rx.inv) : 0.0f;
                        q_du += a[j] * e1.y;
                        q_out += in1 ? 0.0f : a[j];
                        q_ds = fmaf(gz1, e1.x, q_ds);
                        q_db += gz1;
                    }
                }
            }
        }
    }
    DWB_STAMP(4);      // phase 2 done
    if constexpr (FQSS_DWB_ABL & 4) {
        if (p_du + p_out + p_slope + p_bias + pw[0] + q_ds + q_db + q_du + q_out == 1.2345f) gacc[0] = 1.0;
        return;
    }
    // ONE block reduction for the layer's own partials and (GB) the producing GroupNormQ's: one barrier pair and one serial section of
    // thread 0 instead of two (each value keeps its own summation tree: same bits as two reductions)
    constexpr int NQ = GB ? 4 : 0;
    double v[4 + NT + (GB ? 4 : 1)];
    {
        float pf[4 + NT + (GB ? 4 : 1)];
        pf[0] = p_du; pf[1] = p_out; pf[2] = p_slope; pf[3] = p_bias;
#pragma unroll
        for (int k = 0; k < NT; ++k) pf[4 + k] = pw[k];
        if constexpr (GB) { pf[4 + NT] = q_ds; pf[5 + NT] = q_db; pf[6 + NT] = q_du; pf[7 + NT] = q_out; }
        else pf[4 + NT] = 0.f;
        block_sum_f32w<4 + NT + (GB ? 4 : 1)>(pf, redf, v);
    }
    DWB_STAMP(5);
    if (threadIdx.x == 0) {
        if constexpr (GB) {
            const double* qv = v + 4 + NT;
            GBd.ws[2 * (int64_t)row] = qv[0];
            GBd.ws[2 * (int64_t)row + 1] = qv[1];
            double* slot = GBd.gacc + 3 * (row % kSlots);
            const double dmax = qv[2] / 255.0;
            atomicAdd(&slot[0], qv[3] - dmax);
            atomicAdd(&slot[1], dmax);
        }
        double* slot = gacc + 3 * (row % kSlots);
        const double dmax = v[0] / 255.0;
        atomicAdd(&slot[0], v[1] - dmax);   // rows share slots modulo kSlots: few adders per address, order-insensitive in fp64
        atomicAdd(&slot[1], dmax);
        if (act == FQSS_ACT_PRELU) atomicAdd(&slot[2], v[2]);
        if (gbias != nullptr) grad_add(&gbias[c], (float)v[3], det_on);
        if (gw != nullptr)
            for (int k = 0; k < K; ++k) grad_add(&gw[c * K + k], (float)v[4 + k], det_on);
    }
    (void)NQ;
#ifdef FQSS_DWB_STAMPS
    DWB_STAMP(6);
    if constexpr (GB) {
        if (threadIdx.x == 0) {
            unsigned long long w0_ = 0, w1_ = 0;
            for (int k = 0; k < 6; ++k) {
                unsigned long long d = (stamp_[k + 1] - stamp_[k]) >> 4;
                d = d > 65535 ? 65535 : d;
                if (k < 4) w0_ |= d << (16 * k); else w1_ |= d << (16 * (k - 4));
            }
            w1_ |= ((stamp_[0] >> 8) & 0xffffffffull) << 32;        // start time in units of 256 cycles (relative order of the workgroups)
            reinterpret_cast<unsigned long long*>(GBd.ws)[2 * (int64_t)row] = w0_;
            reinterpret_cast<unsigned long long*>(GBd.ws)[2 * (int64_t)row + 1] = w1_;
        }
    }
#endif
}

// gw[c][k] += sum_{b,m} gz[b][c][m] * decode(x[b][c][m + k*dil - pad])   ; grid (C, B), 4 m per thread
__global__ __launch_bounds__(256) void k_dwq_bwd_w(const float* __restrict__ gz, const uint8_t* __restrict__ xc, float* gw,
                                                    int C, int M, int K, int dil, int pad, int64_t ld_gz, int64_t ld_xc,
                                                    const float* qmin_x, const float* qmax_x) {
    __shared__ double red[kTaps * 4];
    const QRange rx = load_qrange(qmin_x, qmax_x);
    const int c = blockIdx.x, b = blockIdx.y;
    const int64_t row = (int64_t)b * C + c;
    const float* gr = gz + row * ld_gz;
    const uint8_t* xr = xc + row * ld_xc;
    float p[kTaps];
#pragma unroll
    for (int k = 0; k < kTaps; ++k) p[k] = 0.0f;
    for (int m = threadIdx.x * 4; m < M; m += 256 * 4) {
        const float4 g4 = *reinterpret_cast<const float4*>(gr + m);
        const float gv[4] = {g4.x, (m + 1 < M) ? g4.y : 0.f, (m + 2 < M) ? g4.z : 0.f, (m + 3 < M) ? g4.w : 0.f};
#pragma unroll
        for (int k = 0; k < kTaps; ++k) {
            if (k < K) {
                const int s0 = m + k * dil - pad;
                const unsigned int w = load_codes4(xr, s0, (int)ld_xc);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (s0 + j >= 0 && s0 + j < M) p[k] = fmaf(gv[j], dec((w >> (8 * j)) & 255u, rx), p[k]);
            }
        }
    }
    double v[kTaps];
#pragma unroll
    for (int k = 0; k < kTaps; ++k) v[k] = (double)p[k];
    block_sum<double, kTaps>(v, red);
    if (threadIdx.x == 0)
        for (int k = 0; k < K; ++k) grad_add(&gw[c * K + k], (float)v[k]);
}

// ---------------------------------------------------------------------------------------------
// element-wise layers on codes: out = fq( act( dec(a) + sb * (dec(b) | b_fp32) ) )  -- AddQ, the residual
// Sub of ResidualErrorBlock (b in fp32), NlQ (no b).  16 elements per thread, codes in / codes out.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ewq_fwd(const uint8_t* __restrict__ ac, const uint8_t* __restrict__ bc,
                                                  const float* __restrict__ bf, float sb, uint8_t* __restrict__ yc,
                                                  float* __restrict__ yout, int rows, int cols, int ld_a, int ld_b, int ld_bf,
                                                  int ld_y, int ld_o, int act, const float* slope_p, const float* amin,
                                                  const float* amax, const float* bmin, const float* bmax, const float* qmin,
                                                  const float* qmax) {
    const QRange ra = load_qrange(amin, amax), ry = load_qrange(qmin, qmax);
    QRange rb{0.f, 1.f, 1.f};
    if (bc != nullptr) rb = load_qrange(bmin, bmax);
    const float slope = (act == FQSS_ACT_PRELU) ? *slope_p : 0.0f;
    for (int row = blockIdx.y; row < rows; row += gridDim.y) {
        for (int c0 = (blockIdx.x * 256 + threadIdx.x) * 16; c0 < cols; c0 += gridDim.x * 256 * 16) {
            const uint4 va = *reinterpret_cast<const uint4*>(ac + (int64_t)row * ld_a + c0);
            const unsigned int wa[4] = {va.x, va.y, va.z, va.w};
            unsigned int wb[4] = {0, 0, 0, 0};
            if (bc != nullptr) {
                const uint4 vb = *reinterpret_cast<const uint4*>(bc + (int64_t)row * ld_b + c0);
                wb[0] = vb.x; wb[1] = vb.y; wb[2] = vb.z; wb[3] = vb.w;
            }
            unsigned int o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float bv[4] = {0.f, 0.f, 0.f, 0.f};
                if (bf != nullptr && c0 + 4 * q < cols) {
                    const float4 t = *reinterpret_cast<const float4*>(bf + (int64_t)row * ld_bf + c0 + 4 * q);
                    bv[0] = t.x; bv[1] = t.y; bv[2] = t.z; bv[3] = t.w;
                }
                unsigned int pk = 0;
                float fo[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float z = dec((wa[q] >> (8 * e)) & 255u, ra);
                    if (bc != nullptr) z = z + sb * dec((wb[q] >> (8 * e)) & 255u, rb);
                    else if (bf != nullptr) z = z + sb * bv[e];
                    fo[e] = fq_code(act_apply(z, act, slope), ry);
                    pk = pack_code(fo[e], e, pk);
                }
                o[q] = pk;
                if (yout != nullptr && c0 + 4 * q < cols)
                    *reinterpret_cast<float4*>(yout + (int64_t)row * ld_o + c0 + 4 * q) =
                        make_float4(ry.delta * fo[0] + ry.lo, ry.delta * fo[1] + ry.lo, ry.delta * fo[2] + ry.lo, ry.delta * fo[3] + ry.lo);
            }
            *reinterpret_cast<uint4*>(yc + (int64_t)row * ld_y + c0) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}

// backward: recompute z from the input codes; gz = STE(act') ; range/slope partials to the gacc slots
// An operand that is the fake-quantized output of a pointwise conv (res / skip conv of a TCN block) can have THAT
// layer's epilogue backward done here (cf. GnProducer): its gz goes to `out`, its partials to its own slots / bias.
struct EwProducer {
    const float* pz; int ld_pz;      // producer's pre-quant output (NULL: operand not fused)
    int act; const float* slope;
    double* gacc; float* gbias;      // gbias [C] nullable
    float* out; int ld_out;          // the producer's gz
};

// PLAIN: no non-linearity anywhere (AddQ of the res / skip convs: the case of every launch of the ConvTasNet step).  These kernels are
// VALU-issue bound: act_apply / act_bwd as identities and the per-element validity selects (a masked-out position already carries
// gj = 0, which zeroes every sum it enters) were ~25 of the ~86 instructions per element.
template <bool PLAIN>
__device__ __forceinline__ float ew_producer_bwd(const EwProducer& P, const QRange& rp, float pslope, float pz, float gj, bool valid,
                                                 float& p_du, float& p_out, float& p_slope, float& p_bias) {
    const float t = PLAIN ? pz : act_apply(pz, P.act, pslope);
    float pc, pu;
    bool pin;
    (void)fq_asym(t, rp, pc, pu, pin);
    const float gt = pin ? div_by(gj * rp.delta, rp.delta, rp.inv) : 0.0f;
    if constexpr (PLAIN) {      // gj == 0 where the position is masked out
        p_du += gj * (pin ? (pc - pu) : pc);
        p_out += pin ? 0.0f : gj;
        p_bias += gt;
        return gt;
    } else {
#ifdef FQSS_BRANCHY_SUMS      // diagnostic build only (tools/diag_streams.py): the round-2 form with a branch around the additions
        if (valid) {
            p_du += gj * (pin ? (pc - pu) : pc);
            p_out += pin ? 0.0f : gj;
        }
        float gzj = act_bwd(pz, gt, P.act, pslope, valid, p_slope);
        if (valid) p_bias += gzj;
        return gzj;
#else
        p_du += valid ? gj * (pin ? (pc - pu) : pc) : 0.0f;          // (selects, not branches: fq.hip k_actq_bwd)
        p_out += (valid && !pin) ? gj : 0.0f;
        float gzj = act_bwd(pz, gt, P.act, pslope, valid, p_slope);
        p_bias += valid ? gzj : 0.0f;
        return gzj;
#endif
    }
}

#ifndef FQSS_EWQ_NR
#define FQSS_EWQ_NR 2        // rows of loads in flight per thread (x FQSS_EWQ_WAVES waves per SIMD; 8 x 2 first: 196 VGPRs; round 5, the fused
                             // residual-add launch alone, us: 8 x 2 24.4, 8 x 3 32.0, 4 x 3 24.9, 3 x 3 23.6, 2 x 3 22.0, 2 x 4 22.3, 1 x 3 21.4, 1 x 4 21.2 --
                             // the step within +-0.03 ms for every row but 8 x 3 (+0.4); same per-thread summation order whatever NR)
#define FQSS_EWQ_WAVES 3
#endif
template <bool PLAIN>
__global__ __launch_bounds__(256, FQSS_EWQ_WAVES) void k_ewq_bwd(const uint8_t* __restrict__ ac, const uint8_t* __restrict__ bc,
                                                  const float* __restrict__ bf, float sb, const float* __restrict__ g,
                                                  float* __restrict__ gz, int rows, int cols, int ld_a, int ld_b, int ld_bf,
                                                  int ld_g, int ld_gz, int act, const float* slope_p, const float* amin,
                                                  const float* amax, const float* bmin, const float* bmax, const float* qmin,
                                                  const float* qmax, double* gacc, EwProducer PA, EwProducer PB, int C) {
    __shared__ float redf[2 * 4];
    const bool fa = PA.pz != nullptr, fb = PB.pz != nullptr;
    const float sla = (fa && PA.act == FQSS_ACT_PRELU) ? *PA.slope : 0.0f, slb = (fb && PB.act == FQSS_ACT_PRELU) ? *PB.slope : 0.0f;
    float a_du = 0.f, a_out = 0.f, a_sl = 0.f, b_du = 0.f, b_out = 0.f, b_sl = 0.f;
    const QRange ra = load_qrange(amin, amax), ry = load_qrange(qmin, qmax);
    QRange rb{0.f, 1.f, 1.f};
    if (bc != nullptr) rb = load_qrange(bmin, bmax);
    const float slope = (act == FQSS_ACT_PRELU) ? *slope_p : 0.0f;
    float p_du = 0.f, p_out = 0.f, p_slope = 0.f;
    // fused launches use gridDim.y % C == 0: the rows of a workgroup (row = y, y + gridDim.y, ...) are batch entries of ONE
    // channel (y % C), so the producers' bias sums are reduced once per workgroup instead of once per row
    const bool per_channel = ((int)gridDim.y % C) == 0;
    float a_bias = 0.f, b_bias = 0.f;
    struct EwIn {
        unsigned int wa, wb;
        float4 za, zb, bf4, g4;
    };
    auto load_group = [&](int row, int c0, EwIn& in) {
        in.wa = *reinterpret_cast<const unsigned int*>(ac + (int64_t)row * ld_a + c0);
        in.wb = bc ? *reinterpret_cast<const unsigned int*>(bc + (int64_t)row * ld_b + c0) : 0u;
        in.za = in.zb = in.bf4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (fa) in.za = *reinterpret_cast<const float4*>(PA.pz + (int64_t)row * PA.ld_pz + c0);
        if (fb) in.zb = *reinterpret_cast<const float4*>(PB.pz + (int64_t)row * PB.ld_pz + c0);
        if (bf != nullptr) in.bf4 = *reinterpret_cast<const float4*>(bf + (int64_t)row * ld_bf + c0);
        in.g4 = *reinterpret_cast<const float4*>(g + (int64_t)row * ld_g + c0);
    };
    auto do_group = [&](int row, int c0, const EwIn& in) {
        const float zav[4] = {in.za.x, in.za.y, in.za.z, in.za.w}, zbv[4] = {in.zb.x, in.zb.y, in.zb.z, in.zb.w};
        const float bv[4] = {in.bf4.x, in.bf4.y, in.bf4.z, in.bf4.w}, gv[4] = {in.g4.x, in.g4.y, in.g4.z, in.g4.w};
        float o[4], oa[4], ob[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool valid = c0 + e < cols;
            const float gj = valid ? gv[e] : 0.0f;
            float z = dec((in.wa >> (8 * e)) & 255u, ra);
            if (bc != nullptr) z = z + sb * dec((in.wb >> (8 * e)) & 255u, rb);
            else if (bf != nullptr) z = z + sb * bv[e];
            const float t = PLAIN ? z : act_apply(z, act, slope);
            float cq, u;
            bool inr;
            (void)fq_asym(t, ry, cq, u, inr);
            const float gt = inr ? div_by(gj * ry.delta, ry.delta, ry.inv) : 0.0f;
            float gzj;
            if constexpr (PLAIN) {      // gj == 0 where the position is masked out: it drops out of every sum by itself
                p_du += gj * (inr ? (cq - u) : cq);
                p_out += inr ? 0.0f : gj;
                gzj = gt;
            } else {
                p_du += valid ? gj * (inr ? (cq - u) : cq) : 0.0f;
                p_out += (valid && !inr) ? gj : 0.0f;
                gzj = act_bwd(z, gt, act, slope, valid, p_slope);
            }
            o[e] = gzj;
            // d/da = gz, d/db = sb * gz (sb == 1 whenever b is fused): the producers' epilogue backward on it
            if (fa) oa[e] = ew_producer_bwd<PLAIN>(PA, ra, sla, zav[e], (PLAIN || valid) ? gzj : 0.0f, valid, a_du, a_out, a_sl, a_bias);
            if (fb) ob[e] = ew_producer_bwd<PLAIN>(PB, rb, slb, zbv[e], (PLAIN || valid) ? gzj : 0.0f, valid, b_du, b_out, b_sl, b_bias);
        }
        if (gz != nullptr) *reinterpret_cast<float4*>(gz + (int64_t)row * ld_gz + c0) = make_float4(o[0], o[1], o[2], o[3]);
        if (fa) *reinterpret_cast<float4*>(PA.out + (int64_t)row * PA.ld_out + c0) = make_float4(oa[0], oa[1], oa[2], oa[3]);
        if (fb) *reinterpret_cast<float4*>(PB.out + (int64_t)row * PB.ld_out + c0) = make_float4(ob[0], ob[1], ob[2], ob[3]);
    };
    const bool row_bias = !per_channel && ((fa && PA.gbias != nullptr) || (fb && PB.gbias != nullptr));   // workgroup-uniform
    auto row_bias_flush = [&](int row) {
        float pb[2] = {a_bias, b_bias};
        block_sum<float, 2>(pb, redf);
        if (threadIdx.x == 0) {
            if (fa && PA.gbias != nullptr) grad_add(&PA.gbias[row % C], pb[0]);
            if (fb && PB.gbias != nullptr) grad_add(&PB.gbias[row % C], pb[1]);
        }
        a_bias = b_bias = 0.f;
    };
    const int c_first = (blockIdx.x * 256 + threadIdx.x) * 4;
    if ((int64_t)gridDim.x * 1024 >= cols) {
        // every thread owns ONE column group of each of the workgroup's rows: the loads of 8 rows are issued before the first is
        // consumed, UNCONDITIONALLY (row / column clamped into the tensor, the results of clamped groups dropped): a branch around a
        // load makes the compiler retire the loads in flight first, and the kernel is latency-bound (2 waves per SIMD, PMC: 64 % of
        // the wave time waiting on memory)
        constexpr int NR = FQSS_EWQ_NR;
        const bool active = c_first < cols;
        const int c_ld = active ? c_first : 0;
        const int rstep = gridDim.y;
        for (int row0 = blockIdx.y; row0 < rows; row0 += NR * rstep) {
            EwIn in[NR];
#pragma unroll
            for (int i = 0; i < NR; ++i) load_group(min(row0 + i * rstep, rows - 1), c_ld, in[i]);
#pragma unroll
            for (int i = 0; i < NR; ++i)
                if (row0 + i * rstep < rows) {
                    if (active) do_group(row0 + i * rstep, c_first, in[i]);
                    if (row_bias) row_bias_flush(row0 + i * rstep);
                }
        }
    } else {
        for (int row = blockIdx.y; row < rows; row += gridDim.y) {
            for (int c0 = c_first; c0 < cols; c0 += gridDim.x * 256 * 4) {
                EwIn in;
                load_group(row, c0, in);
                do_group(row, c0, in);
            }
            if (row_bias) row_bias_flush(row);
        }
    }
    if (per_channel && ((fa && PA.gbias != nullptr) || (fb && PB.gbias != nullptr))) {
        float pb[2] = {a_bias, b_bias};
        block_sum<float, 2>(pb, redf);
        if (threadIdx.x == 0) {
            if (fa && PA.gbias != nullptr) grad_add(&PA.gbias[blockIdx.y % C], pb[0]);
            if (fb && PB.gbias != nullptr) grad_add(&PB.gbias[blockIdx.y % C], pb[1]);
        }
    }
    // Reduction tail: fp32 sums over the 64 lanes of a wave (shuffles only), then ONE fp64 atomic per wave and value into the
    // workgroup's own slot.  (The first form -- block_sum<double> per group of values: 64-bit shuffles, LDS, two barriers each --
    // cost 4.5-5 us of this 30 us kernel, measured by ablation.)
    const int64_t sid = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;   // < kSlots: one workgroup per slot
    const bool lane0 = (threadIdx.x & 63) == 0;
    {
        const float du = wave_sum(p_du), po = wave_sum(p_out), ps = wave_sum(p_slope);
        if (lane0) {
            double* slot = gacc + 3 * sid;
            const double dmax = (double)du / 255.0;
            atomicAdd(&slot[0], (double)po - dmax);
            atomicAdd(&slot[1], dmax);
            if (act == FQSS_ACT_PRELU) atomicAdd(&slot[2], (double)ps);
        }
    }
    if (fa) {
        const float du = wave_sum(a_du), po = wave_sum(a_out), ps = wave_sum(a_sl);
        if (lane0) {
            const double dmax = (double)du / 255.0;
            atomicAdd(&PA.gacc[3 * sid], (double)po - dmax);
            atomicAdd(&PA.gacc[3 * sid + 1], dmax);
            if (PA.act == FQSS_ACT_PRELU) atomicAdd(&PA.gacc[3 * sid + 2], (double)ps);
        }
    }
    if (fb) {
        const float du = wave_sum(b_du), po = wave_sum(b_out), ps = wave_sum(b_sl);
        if (lane0) {
            const double dmax = (double)du / 255.0;
            atomicAdd(&PB.gacc[3 * sid], (double)po - dmax);
            atomicAdd(&PB.gacc[3 * sid + 1], dmax);
            if (PB.act == FQSS_ACT_PRELU) atomicAdd(&PB.gacc[3 * sid + 2], (double)ps);
        }
    }
}


// ---------------------------------------------------------------------------------------------
// The backward of a CHAIN of AddQ layers in one launch (round 5): out_l = fq_l(dec(out_{l-1}) + dec(b_l)), l = 0 .. n-1, where each
// out_l is consumed by the next add alone -- the skip sum of MaskGenerator.forward (convtasnetq.py:107-111: `output =
// self.adds[idx](output, skip)`, 23 levels).  Launched level by level (k_ewq_bwd<true>) every level reads its incoming gradient
// (16 MB) and writes the gradient of its first operand (16 MB) only for the next launch to read it again; here a thread keeps the
// gradient of its elements in registers from the top level to the bottom and per level reads the two code words + the producer's z
// and writes the producer's gz: 41 MB instead of 74.  Thread -> element map, per-thread summation order, reductions and slots are
// those of the per-level launches (grid (ceil(cols / 1024), C), a workgroup serves the batch entries of one channel), so every
// result has the bits the per-level launches give (tests/test_gpu_kernels.py::test_add_chain_backward).
// Every b_l is the fresh output of a pointwise conv without activation (its epilogue backward rides here, as in k_ewq_bwd);
// the bottom level's first operand may be one too (az != NULL), else its gradient is written to ga_out.
// ---------------------------------------------------------------------------------------------
constexpr int kChainMax = 24, kChainRows = 8;
struct ChainLevel {
    const uint8_t* ac; const uint8_t* bc; const float* bz; float* bout;
    const float *amin, *amax, *bmin, *bmax, *qmin, *qmax;
    double* gacc; double* bgacc; float* bgbias;
};
struct ChainArgs {
    int nlev, rows, cols, C;
    int ld_a, ld_b, ld_bz, ld_bout, ld_g, ld_ga, ld_az, ld_aout;
    const float* g; float* ga_out;
    const float* az; float* aout; double* agacc; float* agbias;
    ChainLevel lv[kChainMax];      // lv[0] = the bottom of the chain (the first add of the forward)
};
static_assert(sizeof(ChainArgs) <= 4096, "the level table travels in the kernel arguments");

__global__ __launch_bounds__(256, 2) void k_ewq_chain_bwd(ChainArgs A) {
    __shared__ float redf[2 * 4];
    const int cols = A.cols, rows = A.rows, rstep = gridDim.y;
    const int c_first = (blockIdx.x * 256 + threadIdx.x) * 4;
    const bool active = c_first < cols;
    const int c_ld = active ? c_first : 0;
    const int64_t sid = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    const bool lane0 = (threadIdx.x & 63) == 0;
    int rowi[kChainRows];
    bool rok[kChainRows];
#pragma unroll
    for (int i = 0; i < kChainRows; ++i) {
        rok[i] = (int)blockIdx.y + i * rstep < rows;
        rowi[i] = min((int)blockIdx.y + i * rstep, rows - 1);
    }
    float gv[kChainRows][4];
#pragma unroll
    for (int i = 0; i < kChainRows; ++i) {
        const float4 t = *reinterpret_cast<const float4*>(A.g + (int64_t)rowi[i] * A.ld_g + c_ld);
        const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) gv[i][e] = (active && rok[i] && c_first + e < cols) ? tv[e] : 0.0f;
    }
    struct In { unsigned int wa, wb; float4 zb; };
    auto load_level = [&](const ChainLevel& lv, In (&in)[kChainRows]) {
#pragma unroll
        for (int i = 0; i < kChainRows; ++i) {
            in[i].wa = *reinterpret_cast<const unsigned int*>(lv.ac + (int64_t)rowi[i] * A.ld_a + c_ld);
            in[i].wb = *reinterpret_cast<const unsigned int*>(lv.bc + (int64_t)rowi[i] * A.ld_b + c_ld);
            in[i].zb = *reinterpret_cast<const float4*>(lv.bz + (int64_t)rowi[i] * A.ld_bz + c_ld);
        }
    };
    In cur[kChainRows], nxt[kChainRows];
    load_level(A.lv[A.nlev - 1], cur);
    for (int L = A.nlev - 1; L >= 0; --L) {
        const ChainLevel& lv = A.lv[L];
        if (L > 0) load_level(A.lv[L - 1], nxt);      // the next level's operands are on their way while this one is worked on
        const QRange ra = load_qrange(lv.amin, lv.amax), rb = load_qrange(lv.bmin, lv.bmax), ry = load_qrange(lv.qmin, lv.qmax);
        const bool fa = (L == 0) && A.az != nullptr;
        const QRange rpa = ra;
        float p_du = 0.f, p_out = 0.f, b_du = 0.f, b_out = 0.f, b_sl = 0.f, b_bias = 0.f, a_du = 0.f, a_out = 0.f, a_sl = 0.f, a_bias = 0.f;
        EwProducer PB{lv.bz, A.ld_bz, FQSS_ACT_NONE, nullptr, lv.bgacc, lv.bgbias, lv.bout, A.ld_bout};
        EwProducer PA{A.az, A.ld_az, FQSS_ACT_NONE, nullptr, A.agacc, A.agbias, A.aout, A.ld_aout};
#pragma unroll
        for (int i = 0; i < kChainRows; ++i) {
            if (!rok[i]) continue;      // (workgroup-uniform)
            float4 za4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (fa) za4 = *reinterpret_cast<const float4*>(A.az + (int64_t)rowi[i] * A.ld_az + c_ld);
            if (active) {
                const float zbv[4] = {cur[i].zb.x, cur[i].zb.y, cur[i].zb.z, cur[i].zb.w}, zav[4] = {za4.x, za4.y, za4.z, za4.w};
                float ob[4], oa[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gj = gv[i][e];       // 0 where the position is masked out: it drops out of every sum by itself
                    float z = dec((cur[i].wa >> (8 * e)) & 255u, ra);
                    z = z + 1.0f * dec((cur[i].wb >> (8 * e)) & 255u, rb);
                    float cq, u;
                    bool inr;
                    (void)fq_asym(z, ry, cq, u, inr);
                    const float gt = inr ? div_by(gj * ry.delta, ry.delta, ry.inv) : 0.0f;
                    p_du += gj * (inr ? (cq - u) : cq);
                    p_out += inr ? 0.0f : gj;
                    const bool valid = c_first + e < cols;
                    if (fa) oa[e] = ew_producer_bwd<true>(PA, rpa, 0.0f, zav[e], gt, valid, a_du, a_out, a_sl, a_bias);
                    ob[e] = ew_producer_bwd<true>(PB, rb, 0.0f, zbv[e], gt, valid, b_du, b_out, b_sl, b_bias);
                    gv[i][e] = gt;
                }
                *reinterpret_cast<float4*>(lv.bout + (int64_t)rowi[i] * A.ld_bout + c_first) = make_float4(ob[0], ob[1], ob[2], ob[3]);
                if (fa) *reinterpret_cast<float4*>(A.aout + (int64_t)rowi[i] * A.ld_aout + c_first) = make_float4(oa[0], oa[1], oa[2], oa[3]);
                if (L == 0 && A.ga_out != nullptr)
                    *reinterpret_cast<float4*>(A.ga_out + (int64_t)rowi[i] * A.ld_ga + c_first) = make_float4(gv[i][0], gv[i][1], gv[i][2], gv[i][3]);
            }
        }
        // the level's reductions, in the forms and orders of k_ewq_bwd's tail
        if (lv.bgbias != nullptr || (fa && A.agbias != nullptr)) {
            float pb[2] = {a_bias, b_bias};
            block_sum<float, 2>(pb, redf);
            if (threadIdx.x == 0) {
                if (fa && A.agbias != nullptr) grad_add(&A.agbias[blockIdx.y % A.C], pb[0]);
                if (lv.bgbias != nullptr) grad_add(&lv.bgbias[blockIdx.y % A.C], pb[1]);
            }
        }
        {
            const float du = wave_sum(p_du), po = wave_sum(p_out);
            if (lane0) {
                double* slot = lv.gacc + 3 * sid;
                const double dmax = (double)du / 255.0;
                atomicAdd(&slot[0], (double)po - dmax);
                atomicAdd(&slot[1], dmax);
            }
        }
        if (fa) {
            const float du = wave_sum(a_du), po = wave_sum(a_out);
            if (lane0) {
                const double dmax = (double)du / 255.0;
                atomicAdd(&A.agacc[3 * sid], (double)po - dmax);
                atomicAdd(&A.agacc[3 * sid + 1], dmax);
            }
        }
        {
            const float du = wave_sum(b_du), po = wave_sum(b_out);
            if (lane0) {
                const double dmax = (double)du / 255.0;
                atomicAdd(&lv.bgacc[3 * sid], (double)po - dmax);
                atomicAdd(&lv.bgacc[3 * sid + 1], dmax);
            }
        }
        if (L > 0) {
#pragma unroll
            for (int i = 0; i < kChainRows; ++i) cur[i] = nxt[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// MulQ on codes: masked[b][s][c][:] = fq( dec(mask[b][s][c][:]) * dec(feat[b][c][:]) )  -- the masking product of
// ConvTasNetQ.forward (convtasnetq.py:277, qat_layers.py:134-153 `MulQ`).  A workgroup row is one (b, c) row of feat and
// serves its S mask rows: feat is read (and decoded) once.  Same arithmetic as decode -> fqss_mul_bcast_fwd -> fqss_actq_fwd
// (one fp32 product of the two de-quantised values, then the quantizer), so the codes are bit-identical to that chain's.
// ---------------------------------------------------------------------------------------------
constexpr int kMulqSMax = 4;

template <int S>
__global__ __launch_bounds__(256) void k_mulq_fwd(const uint8_t* __restrict__ mc, const uint8_t* __restrict__ fc,
                                                   uint8_t* __restrict__ yc, float* __restrict__ yout, int rows, int C, int M,
                                                   int64_t ld_m, int64_t ld_f, int64_t ld_y, int64_t ld_o, const float* mmin,
                                                   const float* mmax, const float* fmin, const float* fmax, const float* qmin,
                                                   const float* qmax) {
    const QRange rm = load_qrange(mmin, mmax), rf = load_qrange(fmin, fmax), ry = load_qrange(qmin, qmax);
    for (int row = blockIdx.y; row < rows; row += gridDim.y) {
        const int b = row / C, c = row - b * C;
        for (int c0 = (blockIdx.x * 256 + threadIdx.x) * 16; c0 < M; c0 += gridDim.x * 256 * 16) {
            const uint4 vf = *reinterpret_cast<const uint4*>(fc + (int64_t)row * ld_f + c0);
            uint4 vm[S];
#pragma unroll
            for (int s = 0; s < S; ++s)      // all mask rows requested before the first is used
                vm[s] = *reinterpret_cast<const uint4*>(mc + ((int64_t)(b * S + s) * C + c) * ld_m + c0);
            const unsigned int wf[4] = {vf.x, vf.y, vf.z, vf.w};
            float xf[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) dec4(wf[q], rf, xf[q]);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int64_t orow = (int64_t)(b * S + s) * C + c;
                const unsigned int wm[4] = {vm[s].x, vm[s].y, vm[s].z, vm[s].w};
                unsigned int o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float xm[4], cq[4];
                    dec4(wm[q], rm, xm);
                    unsigned int pk = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        cq[e] = fq_code(xm[e] * xf[q][e], ry);
                        pk = pack_code(cq[e], e, pk);
                    }
                    o[q] = pk;
                    if (yout != nullptr && c0 + 4 * q < M)
                        *reinterpret_cast<float4*>(yout + orow * ld_o + c0 + 4 * q) =
                            make_float4(ry.delta * cq[0] + ry.lo, ry.delta * cq[1] + ry.lo, ry.delta * cq[2] + ry.lo, ry.delta * cq[3] + ry.lo);
                }
                *reinterpret_cast<uint4*>(yc + orow * ld_y + c0) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
    }
}

// backward: z recomputed from the codes, gz = STE(g); gmask = gz * feat (or, with the mask's producer fused -- the mask conv's
// non-linearity + output quantizer, cf. EwProducer -- that layer's gz), gfeat = sum_s gz_s * mask_s in ascending s (the order
// of fqss_mul_bcast_bwd); range partials of this layer (and of the producer) to the gacc slots.
#ifdef FQSS_DIAG     // diagnostic build only: per-lane bias partials and per-wave placement / timing of k_mulq_bwd (tools/diag_streams.py)
__device__ float* g_diag_lane = nullptr;                  // [workgroup][256][4]
__device__ unsigned long long* g_diag_wave = nullptr;     // [workgroup][4 waves][4]: HW_ID | XCC_ID << 32, start, end (s_memtime), end (s_memrealtime)
#endif

template <int S, bool PROD>
__global__ __launch_bounds__(256) void k_mulq_bwd(const uint8_t* __restrict__ mc, const uint8_t* __restrict__ fc,
                                                   const float* __restrict__ g, float* __restrict__ gmask, float* __restrict__ gfeat,
                                                   int rows, int C, int M, int64_t ld_m, int64_t ld_f, int64_t ld_g,
                                                   int64_t ld_gm, int64_t ld_gf, const float* mmin, const float* mmax,
                                                   const float* fmin, const float* fmax, const float* qmin, const float* qmax,
                                                   double* gacc, EwProducer P) {
    __shared__ float redf[S * 4];
    const QRange rm = load_qrange(mmin, mmax), rf = load_qrange(fmin, fmax), ry = load_qrange(qmin, qmax);
    const float pslope = (PROD && P.act == FQSS_ACT_PRELU) ? *P.slope : 0.0f;
    float p_du = 0.f, p_out = 0.f, a_du = 0.f, a_out = 0.f, a_sl = 0.f;
    // gridDim.y % C == 0 (the launch's choice whenever C fits): the rows of a workgroup (row = y, y + gridDim.y, ...) are batch entries of
    // ONE channel, so the producer's bias sums are carried across them and added once per workgroup -- B times fewer atomics, and
    // a sum whose value does not depend on the order in which B partial sums of opposite signs arrive (seen as 1e-4 of run-to-run
    // movement in this one gradient when a second stream perturbed the order, tools/stress_step.py BWD=1)
    const bool per_channel = ((int)gridDim.y % C) == 0;
    float a_bias[S];
#pragma unroll
    for (int s = 0; s < S; ++s) a_bias[s] = 0.f;
#ifdef FQSS_DIAG
    const unsigned long long diag_t0 = __builtin_amdgcn_s_memtime();
    float diag_bias[S];
#pragma unroll
    for (int s = 0; s < S; ++s) diag_bias[s] = 0.f;
#endif
    for (int row = blockIdx.y; row < rows; row += gridDim.y) {
        const int b = row / C, c = row - b * C;
        for (int c0 = (blockIdx.x * 256 + threadIdx.x) * 4; c0 < M; c0 += gridDim.x * 256 * 4) {
            const unsigned int wf = *reinterpret_cast<const unsigned int*>(fc + (int64_t)row * ld_f + c0);
            unsigned int wm[S];
            float4 g4[S], z4[S];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int64_t mrow = (int64_t)(b * S + s) * C + c;
                wm[s] = *reinterpret_cast<const unsigned int*>(mc + mrow * ld_m + c0);
                g4[s] = *reinterpret_cast<const float4*>(g + mrow * ld_g + c0);
                if (PROD) z4[s] = *reinterpret_cast<const float4*>(P.pz + mrow * P.ld_pz + c0);
            }
            float xf[4], gf[4] = {0.f, 0.f, 0.f, 0.f};
            dec4(wf, rf, xf);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int64_t mrow = (int64_t)(b * S + s) * C + c;
                const float gv[4] = {g4[s].x, g4[s].y, g4[s].z, g4[s].w};
                const float pzv[4] = {PROD ? z4[s].x : 0.f, PROD ? z4[s].y : 0.f, PROD ? z4[s].z : 0.f, PROD ? z4[s].w : 0.f};
                float xm[4], o[4];
                dec4(wm[s], rm, xm);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool valid = c0 + e < M;
                    const float gj = valid ? gv[e] : 0.0f;       // the row padding of g holds anything
                    float cq, u;
                    bool inr;
                    (void)fq_asym(xm[e] * xf[e], ry, cq, u, inr);
                    const float gt = inr ? div_by(gj * ry.delta, ry.delta, ry.inv) : 0.0f;
                    p_du += gj * (inr ? (cq - u) : cq);
                    p_out += inr ? 0.0f : gj;
                    gf[e] += gt * xm[e];
                    const float gm = gt * xf[e];
                    o[e] = PROD ? ew_producer_bwd<false>(P, rm, pslope, pzv[e], gm, valid, a_du, a_out, a_sl, a_bias[s]) : gm;
#ifdef FQSS_DIAG
                    diag_bias[s] += valid ? o[e] : 0.0f;      // the same sum, select form, in the same wave
#endif
                }
                *reinterpret_cast<float4*>(gmask + mrow * ld_gm + c0) = make_float4(o[0], o[1], o[2], o[3]);
            }
            if (gfeat != nullptr) *reinterpret_cast<float4*>(gfeat + (int64_t)row * ld_gf + c0) = make_float4(gf[0], gf[1], gf[2], gf[3]);
        }
        if (PROD && P.gbias != nullptr && !per_channel) {       // one sum per (row, s): channel s * C + c of the producer
            block_sum<float, S>(a_bias, redf);
            if (threadIdx.x == 0)
                for (int s = 0; s < S; ++s) grad_add(&P.gbias[s * C + c], a_bias[s]);
#pragma unroll
            for (int s = 0; s < S; ++s) a_bias[s] = 0.f;
        }
    }
#ifdef FQSS_DIAG
    if (g_diag_lane != nullptr) {
        const int64_t wg = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
#pragma unroll
        for (int s = 0; s < S && s < 2; ++s) {
            g_diag_lane[(wg * 256 + threadIdx.x) * 4 + s] = a_bias[s];
            g_diag_lane[(wg * 256 + threadIdx.x) * 4 + 2 + s] = diag_bias[s];
        }
        if ((threadIdx.x & 63) == 0) {
            unsigned long long* w = g_diag_wave + (wg * 4 + (threadIdx.x >> 6)) * 4;
            const unsigned int hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
            w[0] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
            w[1] = diag_t0;
            w[2] = __builtin_amdgcn_s_memtime();
            w[3] = __builtin_amdgcn_s_memrealtime();
        }
    }
#endif
    if (PROD && P.gbias != nullptr && per_channel) {
        // wave sums straight to the atomics (no LDS exchange needed: four waves, one atomic each)
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const float t = wave_sum(a_bias[s]);
            if ((threadIdx.x & 63) == 0) grad_add(&P.gbias[s * C + (int)(blockIdx.y % C)], t);
        }
    }
    const int64_t sid = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;   // < kSlots: one workgroup per slot
    const bool lane0 = (threadIdx.x & 63) == 0;
    {
        const float du = wave_sum(p_du), po = wave_sum(p_out);
        if (lane0) {
            const double dmax = (double)du / 255.0;
            atomicAdd(&gacc[3 * sid], (double)po - dmax);
            atomicAdd(&gacc[3 * sid + 1], dmax);
        }
    }
    if (PROD) {
        const float du = wave_sum(a_du), po = wave_sum(a_out), ps = wave_sum(a_sl);
        if (lane0) {
            const double dmax = (double)du / 255.0;
            atomicAdd(&P.gacc[3 * sid], (double)po - dmax);
            atomicAdd(&P.gacc[3 * sid + 1], dmax);
            if (P.act == FQSS_ACT_PRELU) atomicAdd(&P.gacc[3 * sid + 2], (double)ps);
        }
    }
}

}  // namespace fqss

using namespace fqss;

static inline bool codes_ok(const void* p, int64_t ld) { return aligned16(p) && (ld % 16 == 0); }

#ifdef FQSS_DIAG
extern "C" int fqss_diag_set(float* lane, unsigned long long* wave) {
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_diag_lane), &lane, sizeof(lane)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_diag_wave), &wave, sizeof(wave)) != hipSuccess) return -1;
    return 0;
}
#endif

extern "C" int fqss_decode(const uint8_t* codes, float* out, int64_t rows, int64_t cols, int64_t ld_c, int64_t ld_out,
                           const float* qmin, const float* qmax, fqss_stream_t stream) {
    if (rows == 0 || cols == 0) return FQSS_OK;   // empty input: nothing to do (a 0-element tensor has a null data pointer)
    FQSS_REQUIRE(codes && out && qmin && qmax, "null pointer");
    FQSS_REQUIRE(rows >= 0 && cols >= 0 && ld_c >= cols && ld_out >= cols, "bad shape");
    FQSS_REQUIRE(codes_ok(codes, ld_c) && aligned16(out) && ld_out % 4 == 0, "decode needs 16-B aligned code rows / fp32 rows");
    if (rows == 0 || cols == 0) return FQSS_OK;
    hipLaunchKernelGGL(k_decode, grid_rows(rows, cols, 4), dim3(256), 0, (hipStream_t)stream, codes, out, rows, cols, ld_c,
                       ld_out, qmin, qmax);
    return launch_status("fqss_decode");
}

extern "C" int fqss_gnq_fwd(const uint8_t* xc, const float* qmin_x, const float* qmax_x, const float* gamma,
                            const float* beta, uint8_t* yc, float* yout, float* mean_rstd, int B, int C, int M,
                            int64_t ld_xc, int64_t ld_yc, int64_t ld_out, float eps, const float* qmin, const float* qmax,
                            void* ws, const int64_t* stats, int nslots, fqss_stream_t stream) {
    if (B == 0) return FQSS_OK;   // empty input: nothing to do (a 0-element tensor has a null data pointer)
    FQSS_REQUIRE(xc && qmin_x && qmax_x && gamma && beta && yc && mean_rstd && qmin && qmax && (ws || stats), "null pointer");
    FQSS_REQUIRE(B >= 0 && B <= 65535 && C > 0 && M > 0 && ld_xc >= M && ld_yc >= M, "bad shape");
    FQSS_REQUIRE(codes_ok(xc, ld_xc) && codes_ok(yc, ld_yc), "code rows must be 16-B aligned");
    FQSS_REQUIRE(!yout || (aligned16(yout) && ld_out % 4 == 0 && ld_out >= ((M + 3) & ~3)), "bad fp32 output rows");
    FQSS_REQUIRE(!stats || (nslots > 0 && nslots <= 1024), "supplied statistics: 1 .. 1024 partial-sum slots per sample");
    hipStream_t s = (hipStream_t)stream;
    if (stats == nullptr) {   // the statistics were not produced with the codes: one pass over them
        nslots = C < kGnSlots ? C : kGnSlots;
        hipLaunchKernelGGL(k_gnq_stats, dim3((unsigned)nslots, (unsigned)B), dim3(256), 0, s, xc, C, M, ld_xc, (long long*)ws);
        stats = (const int64_t*)ws;
    }
    const int64_t rows = (int64_t)B * C;
    // FQSS_GNQ_APPLY_V1=1: the per-element form of rounds 1-4 (kept for the bit-identity gate, tests/test_gpu_kernels.py)
    const char* e_v1 = getenv("FQSS_GNQ_APPLY_V1");      // read per call: a test flips it inside one process
    const bool v1 = e_v1 && e_v1[0] == '1';
    if (!v1 && rows < (1ll << 30)) {   // code-indexed LDS tables, RPW rows of one sample per workgroup (>= 2 workgroups per CU kept)
#define FQSS_GNQ_T(RPW)                                                                                                              \
    hipLaunchKernelGGL(k_gnq_apply_t<RPW>, dim3((unsigned)(rows / RPW)), dim3(256), 0, s, xc, gamma, beta, yc, yout, mean_rstd,      \
                       (const long long*)stats, nslots, eps, B, C, M, ld_xc, ld_yc, ld_out, qmin_x, qmax_x, qmin, qmax)
        if (rows >= 2048 && C % 4 == 0) FQSS_GNQ_T(4);
        else if (rows >= 1024 && C % 2 == 0) FQSS_GNQ_T(2);
        else FQSS_GNQ_T(1);
#undef FQSS_GNQ_T
        return launch_status("fqss_gnq_fwd");
    }
    const int rpw = (rows >= 2048 && M <= 4096) ? 4 : 1;     // rows per workgroup (keep >= 2 workgroups per CU)
    dim3 grid = grid_rows(cdiv(rows, rpw), M, 16);
    hipLaunchKernelGGL(k_gnq_apply, grid, dim3(256), 0, s, xc, gamma, beta, yc, yout, mean_rstd,
                       (const long long*)stats, nslots, eps, B, C, M, ld_xc, ld_yc, ld_out, qmin_x, qmax_x, qmin, qmax, rpw);
    return launch_status("fqss_gnq_fwd");
}

static int gnq_bwd_impl(const char* who, const uint8_t* xc, const float* qmin_x, const float* qmax_x, const float* g,
                            const float* gamma, const float* beta, const float* mean_rstd, float* gx, float* ggamma,
                            float* gbeta, int B, int C, int M, int64_t ld_xc, int64_t ld_g, int64_t ld_gx, const float* qmin,
                            const float* qmax, double* gacc, double* ws, const GnProducer& P, fqss_stream_t stream, int passes = 3) {
    if (B == 0) return FQSS_OK;   // empty input: nothing to do (a 0-element tensor has a null data pointer)
    // passes: 1 = the rows pass alone (gx / ggamma / gbeta unused), 2 = the apply pass alone (ws given, gacc unused), 3 = both
    FQSS_REQUIRE(xc && qmin_x && qmax_x && g && gamma && beta && mean_rstd && qmin && qmax && ws, "null pointer");
    FQSS_REQUIRE(!(passes & 1) || gacc, "rows pass: null gacc");
    FQSS_REQUIRE(!(passes & 2) || (gx && ggamma && gbeta), "apply pass: null output");
    FQSS_REQUIRE(B >= 0 && B <= 65535 && C > 0 && M > 0 && ld_xc >= M && ld_g >= M && (!(passes & 2) || ld_gx >= M), "bad shape");
    FQSS_REQUIRE(codes_ok(xc, ld_xc) && aligned16(g) && ld_g % 4 == 0 && ld_g >= ((M + 3) & ~3), "rows must be 16-B aligned");
    FQSS_REQUIRE(!(passes & 2) || (aligned16(gx) && ld_gx % 4 == 0 && ld_gx >= M), "gx rows must be 16-B aligned");
    if (B == 0) return FQSS_OK;
    hipStream_t s = (hipStream_t)stream;
    if (passes & 1)
        hipLaunchKernelGGL(k_gnq_bwd_rows, dim3((unsigned)C, (unsigned)B), dim3(256), 0, s, xc, g, gamma, beta, mean_rstd, C, M,
                           ld_xc, ld_g, ws, qmin_x, qmax_x, qmin, qmax, gacc);
    if (!(passes & 2)) return launch_status(who);
    if (P.pz != nullptr)
        hipLaunchKernelGGL(k_gnq_bwd_apply<true>, dim3((unsigned)C, (unsigned)B), dim3(256), 0, s, xc, g, gamma, beta, mean_rstd, gx, B, C,
                           M, ld_xc, ld_g, ld_gx, ws, qmin_x, qmax_x, qmin, qmax, P, ggamma, gbeta);
    else
        hipLaunchKernelGGL(k_gnq_bwd_apply<false>, dim3((unsigned)C, (unsigned)B), dim3(256), 0, s, xc, g, gamma, beta, mean_rstd, gx, B, C,
                           M, ld_xc, ld_g, ld_gx, ws, qmin_x, qmax_x, qmin, qmax, P, ggamma, gbeta);
    return launch_status(who);
}

extern "C" int fqss_gnq_bwd(const uint8_t* xc, const float* qmin_x, const float* qmax_x, const float* g,
                            const float* gamma, const float* beta, const float* mean_rstd, float* gx, float* ggamma,
                            float* gbeta, int B, int C, int M, int64_t ld_xc, int64_t ld_g, int64_t ld_gx, const float* qmin,
                            const float* qmax, double* gacc, double* ws, fqss_stream_t stream) {
    return gnq_bwd_impl("fqss_gnq_bwd", xc, qmin_x, qmax_x, g, gamma, beta, mean_rstd, gx, ggamma, gbeta, B, C, M, ld_xc, ld_g,
                        ld_gx, qmin, qmax, gacc, ws, GnProducer{}, stream);
}

extern "C" int fqss_gnq_bwd_p(const uint8_t* xc, const float* qmin_x, const float* qmax_x, const float* g, const float* gamma,
                              const float* beta, const float* mean_rstd, float* gz, float* ggamma, float* gbeta, int B, int C,
                              int M, int64_t ld_xc, int64_t ld_g, int64_t ld_gz, const float* qmin, const float* qmax,
                              double* gacc, double* ws, const float* pz, int64_t ld_pz, int pact, const float* pslope,
                              double* pgacc, float* pgbias, fqss_stream_t stream) {
    FQSS_REQUIRE(pz && pgacc && aligned16(pz) && ld_pz % 4 == 0 && ld_pz >= ((M + 3) & ~3), "producer z: 16-B aligned rows");
    FQSS_REQUIRE(pact != FQSS_ACT_PRELU || pslope, "PReLU needs a slope");
    GnProducer P{pz, ld_pz, pact, pslope, qmin_x, qmax_x, pgacc, pgbias};
    return gnq_bwd_impl("fqss_gnq_bwd_p", xc, qmin_x, qmax_x, g, gamma, beta, mean_rstd, gz, ggamma, gbeta, B, C, M, ld_xc, ld_g,
                        ld_gz, qmin, qmax, gacc, ws, P, stream);
}

/* The two passes of fqss_gnq_bwd / fqss_gnq_bwd_p as entry points of their own (round 5): a depthwise layer next to the GroupNormQ may
 * take one of them into its own backward (fqss_dwq_bwd_gn).  ws [B*C][2] doubles: written by the rows pass, read by the apply pass. */
extern "C" int fqss_gnq_bwd_rows(const uint8_t* xc, const float* qmin_x, const float* qmax_x, const float* g, const float* gamma,
                                 const float* beta, const float* mean_rstd, int B, int C, int M, int64_t ld_xc, int64_t ld_g,
                                 const float* qmin, const float* qmax, double* gacc, double* ws, fqss_stream_t stream) {
    return gnq_bwd_impl("fqss_gnq_bwd_rows", xc, qmin_x, qmax_x, g, gamma, beta, mean_rstd, nullptr, nullptr, nullptr, B, C, M, ld_xc, ld_g,
                        0, qmin, qmax, gacc, ws, GnProducer{}, stream, 1);
}

extern "C" int fqss_gnq_bwd_apply(const uint8_t* xc, const float* qmin_x, const float* qmax_x, const float* g, const float* gamma,
                                  const float* beta, const float* mean_rstd, float* gx, float* ggamma, float* gbeta, int B, int C, int M,
                                  int64_t ld_xc, int64_t ld_g, int64_t ld_gx, const float* qmin, const float* qmax, const double* ws,
                                  const float* pz, int64_t ld_pz, int pact, const float* pslope, double* pgacc, float* pgbias,
                                  fqss_stream_t stream) {
    GnProducer P{};
    if (pz != nullptr) {
        FQSS_REQUIRE(pgacc && aligned16(pz) && ld_pz % 4 == 0 && ld_pz >= ((M + 3) & ~3), "producer z: 16-B aligned rows");
        FQSS_REQUIRE(pact != FQSS_ACT_PRELU || pslope, "PReLU needs a slope");
        P = GnProducer{pz, ld_pz, pact, pslope, qmin_x, qmax_x, pgacc, pgbias};
    }
    return gnq_bwd_impl("fqss_gnq_bwd_apply", xc, qmin_x, qmax_x, g, gamma, beta, mean_rstd, gx, ggamma, gbeta, B, C, M, ld_xc, ld_g, ld_gx,
                        qmin, qmax, nullptr, const_cast<double*>(ws), P, stream, 2);
}

#ifdef FQSS_EXPERIMENTS
/* GroupNormQ followed by a 3-tap depthwise Conv1dNlQ, both in their quantizing phase, codes -> codes -> codes in ONE launch on code tables
 * (k_gndwq_fwd_t, round 5): y1 = the GroupNorm's output codes (range 1), y2 = the depthwise layer's (range 2); stats = the producer's
 * statistics of xc ([B][nslots][2]), stats2 (nullable) = [B][C][2] statistics of y2 for a GroupNormQ behind it.  M <= 4096, K = 3;
 * bit-identical to fqss_gnq_fwd + fqss_dwq_fwd. */
extern "C" int fqss_gndwq_fwd(const uint8_t* xc, const float* qmin_x, const float* qmax_x, const float* gamma, const float* beta, float eps,
                              const int64_t* stats, int nslots, float* mean_rstd, uint8_t* y1, const float* qmin1, const float* qmax1,
                              const float* w, const float* bias, int dil, int pad, int act, const float* slope, uint8_t* y2,
                              const float* qmin2, const float* qmax2, int64_t* stats2, int B, int C, int M, int64_t ld_xc, int64_t ld_y1,
                              int64_t ld_y2, fqss_stream_t stream) {
    if (B == 0 || M == 0) return FQSS_OK;
    FQSS_REQUIRE(xc && qmin_x && qmax_x && gamma && beta && stats && mean_rstd && y1 && qmin1 && qmax1 && w && y2 && qmin2 && qmax2, "null pointer");
    FQSS_REQUIRE(B > 0 && C > 0 && M > 0 && M <= 4096 && dil > 0 && pad == dil && pad <= 2048, "rows of at most 4096 positions, 3 taps");
    FQSS_REQUIRE(nslots > 0 && nslots <= 1024, "supplied statistics: 1 .. 1024 partial-sum slots per sample");
    FQSS_REQUIRE(ld_xc >= M && ld_y1 >= M && ld_y2 >= M && codes_ok(xc, ld_xc) && codes_ok(y1, ld_y1) && codes_ok(y2, ld_y2), "code rows must be 16-B aligned");
    FQSS_REQUIRE(act != FQSS_ACT_PRELU || slope, "PReLU needs a slope");
    const int64_t rows = (int64_t)B * C;
    FQSS_REQUIRE(rows < (1ll << 30), "tensor too large");
    const int rpw = (rows >= 2048 && C % 4 == 0) ? 4 : ((rows >= 1024 && C % 2 == 0) ? 2 : 1);     // rows of ONE sample per workgroup
    const size_t smem = (size_t)(((pad + 15) & ~15) + 4096 + pad + 16);
    hipLaunchKernelGGL(k_gndwq_fwd_t, dim3((unsigned)cdiv(rows, rpw)), dim3(256), smem, (hipStream_t)stream, xc, gamma, beta, y1, mean_rstd,
                       (const long long*)stats, nslots, eps, B, C, M, ld_xc, ld_y1, qmin_x, qmax_x, qmin1, qmax1, w, bias, dil, pad, act, slope, y2,
                       ld_y2, qmin2, qmax2, (long long*)stats2, rpw);
    return launch_status("fqss_gndwq_fwd");
}

#endif  // FQSS_EXPERIMENTS

#ifdef FQSS_EXPERIMENTS
/* GroupNormQ followed by a depthwise Conv1dNlQ, both in their quantizing phase, codes -> codes -> codes in one launch (k_gndwq_fwd):
 * y1 = the GroupNorm's output codes (range 1), y2 = the depthwise layer's (range 2); stats = the producer's statistics of xc ([B][nslots][2]),
 * stats2 (nullable) = [B][C][2] statistics of y2 for a GroupNormQ behind it.  M <= 4096, K = 3; bit-identical to fqss_gnq_fwd + fqss_dwq_fwd. */
extern "C" int fqss_gndwq_fwd_v1(const uint8_t* xc, const float* qmin_x, const float* qmax_x, const float* gamma, const float* beta, float eps,
                              const int64_t* stats, int nslots, float* mean_rstd, uint8_t* y1, const float* qmin1, const float* qmax1,
                              const float* w, const float* bias, int dil, int pad, int act, const float* slope, uint8_t* y2,
                              const float* qmin2, const float* qmax2, int64_t* stats2, int B, int C, int M, int64_t ld_xc, int64_t ld_y1,
                              int64_t ld_y2, fqss_stream_t stream) {
    if (B == 0 || M == 0) return FQSS_OK;
    FQSS_REQUIRE(xc && qmin_x && qmax_x && gamma && beta && stats && mean_rstd && y1 && qmin1 && qmax1 && w && y2 && qmin2 && qmax2, "null pointer");
    FQSS_REQUIRE(B > 0 && C > 0 && M > 0 && M <= 4096 && dil > 0 && pad == dil && pad <= 2048, "rows of at most 4096 positions, 3 taps");
    FQSS_REQUIRE(nslots > 0 && nslots <= 1024, "supplied statistics: 1 .. 1024 partial-sum slots per sample");
    FQSS_REQUIRE(ld_xc >= M && ld_y1 >= M && ld_y2 >= M && codes_ok(xc, ld_xc) && codes_ok(y1, ld_y1) && codes_ok(y2, ld_y2), "code rows must be 16-B aligned");
    FQSS_REQUIRE(act != FQSS_ACT_PRELU || slope, "PReLU needs a slope");
    const int64_t rows = (int64_t)B * C;
    FQSS_REQUIRE(rows < (1ll << 30), "tensor too large");
    const int rpw = rows >= 2048 ? 4 : 1;
    const size_t smem = (size_t)(((pad + 15) & ~15) + 4096 + pad + 16);
    hipLaunchKernelGGL(k_gndwq_fwd, dim3((unsigned)cdiv(rows, rpw)), dim3(256), smem, (hipStream_t)stream, xc, gamma, beta, y1, mean_rstd,
                       (const long long*)stats, nslots, eps, B, C, M, ld_xc, ld_y1, qmin_x, qmax_x, qmin1, qmax1, w, bias, dil, pad, act, slope, y2,
                       ld_y2, qmin2, qmax2, (long long*)stats2, rpw);
    return launch_status("fqss_gndwq_fwd_v1");
}
#endif  // FQSS_EXPERIMENTS

extern "C" int fqss_dwq_fwd(const uint8_t* xc, const float* qmin_x, const float* qmax_x, const float* w, const float* bias,
                            uint8_t* yc, float* yout, int B, int C, int M, int K, int dil, int pad, int64_t ld_xc,
                            int64_t ld_yc, int64_t ld_out, int act, const float* slope, const float* qmin, const float* qmax,
                            int64_t* stats, fqss_stream_t stream) {
    if (B == 0 || M == 0) return FQSS_OK;   // empty input: nothing to do (a 0-element tensor has a null data pointer)
    FQSS_REQUIRE(xc && qmin_x && qmax_x && w && yc && qmin && qmax, "null pointer");
    FQSS_REQUIRE(B >= 0 && C > 0 && M >= 0 && K > 0 && K <= kTaps && dil > 0 && 2 * pad == dil * (K - 1), "bad conv geometry");
    FQSS_REQUIRE(ld_xc >= M && ld_yc >= M && codes_ok(xc, ld_xc) && codes_ok(yc, ld_yc), "code rows must be 16-B aligned");
    FQSS_REQUIRE(act != FQSS_ACT_PRELU || slope, "PReLU needs a slope");
    FQSS_REQUIRE(!yout || (aligned16(yout) && ld_out % 4 == 0 && ld_out >= ((M + 3) & ~3)), "bad fp32 output rows");
    const int64_t rows = (int64_t)B * C;
    FQSS_REQUIRE(rows < (1ll << 30) && ld_xc < (1ll << 30), "tensor too large for the 32-bit row kernels");
    dim3 grid = grid_rows(rows, M, 16);
    if (stats != nullptr) {
        // one slot per (row, column chunk): every row gets its own workgroups; the caller sized stats as [B][C * chunks][2]
        FQSS_REQUIRE(fqss_dwq_stat_slots(C, M) > 0, "output statistics: too many slots per sample (C * ceil(M / 4096) <= 1024)");
        grid = dim3((unsigned)cdiv(M, 4096), (unsigned)rows, 1);
        FQSS_REQUIRE(rows <= 65535 * 16, "output statistics: too many rows");
        if (rows > 65535) grid.y = 65535;   // (rows strided over y keep their own slots: indexed by row)
        static const int rpw = getenv("FQSS_DWF_RPW") ? atoi(getenv("FQSS_DWF_RPW")) : 1;       // A/B knob: rows per workgroup
        if (rpw > 1) grid.y = (unsigned)cdiv(rows, rpw);
    }
    if (K == 3)
        hipLaunchKernelGGL(k_dwq_fwd<3>, grid, dim3(256), 0, (hipStream_t)stream, xc, w, bias, yc, yout, (int)rows, C, M, K, dil,
                           pad, (int)ld_xc, (int)ld_yc, (int)ld_out, act, slope, qmin_x, qmax_x, qmin, qmax, (long long*)stats);
    else
        hipLaunchKernelGGL(k_dwq_fwd<0>, grid, dim3(256), 0, (hipStream_t)stream, xc, w, bias, yc, yout, (int)rows, C, M, K, dil,
                           pad, (int)ld_xc, (int)ld_yc, (int)ld_out, act, slope, qmin_x, qmax_x, qmin, qmax, (long long*)stats);
    return launch_status("fqss_dwq_fwd");
}

/* partial-sum slots per sample of the statistics fqss_dwq_fwd / fqss_qpw_fwdq can emit next to their output codes (0: that shape
 * cannot, the GroupNorm computes them itself) */
extern "C" int fqss_dwq_stat_slots(int C, int M) {
    const int64_t n = (int64_t)C * cdiv(M, 4096);
    return (C > 0 && M > 0 && n <= 1024) ? (int)n : 0;
}
extern "C" int fqss_qpw_stat_slots(int Co, int M) {
    const int64_t n = cdiv(Co, 128) * cdiv(M, 64);
    return (Co > 0 && M > 0 && n <= 1024) ? (int)n : 0;
}

extern "C" int fqss_dwq_bwd_z(const uint8_t* xc, const float* qmin_x, const float* qmax_x, const float* w,
                              const float* bias, const float* g, float* gz, int B, int C, int M, int K, int dil, int pad,
                              int64_t ld_xc, int64_t ld_g, int64_t ld_gz, int act, const float* slope, const float* qmin,
                              const float* qmax, double* gacc, float* gbias, fqss_stream_t stream) {
    if (B == 0 || M == 0) return FQSS_OK;   // empty input: nothing to do (a 0-element tensor has a null data pointer)
    FQSS_REQUIRE(xc && qmin_x && qmax_x && w && g && gz && qmin && qmax && gacc, "null pointer");
    FQSS_REQUIRE(B >= 0 && C > 0 && M >= 0 && K > 0 && K <= kTaps && dil > 0 && 2 * pad == dil * (K - 1), "bad conv geometry");
    FQSS_REQUIRE(ld_xc >= M && codes_ok(xc, ld_xc) && aligned16(g) && aligned16(gz) && ld_g % 4 == 0 && ld_gz % 4 == 0 &&
                     ld_g >= ((M + 3) & ~3) && ld_gz >= ((M + 3) & ~3), "rows must be 16-B aligned");
    FQSS_REQUIRE(act != FQSS_ACT_PRELU || slope, "PReLU needs a slope");
    if (B == 0 || M == 0) return FQSS_OK;
    // every workgroup owns one gacc slot: at most kSlots of them.  One float4 group per thread (many small
    // workgroups, 8 per CU) when that fits, else four.
    const bool fine = cdiv(M, 1024) * C <= kSlots;
    int64_t gx_ = cdiv(M, fine ? 1024 : 4096);
    if (gx_ > 64) gx_ = 64;
    int64_t gy = kSlots / gx_;
    if (gy > C) gy = C;
    if (fine)
        hipLaunchKernelGGL(k_dwq_bwd_z<1>, dim3((unsigned)gx_, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, xc, w, bias, g,
                           gz, B, C, M, K, dil, pad, ld_xc, ld_g, ld_gz, act, slope, qmin_x, qmax_x, qmin, qmax, gacc, gbias);
    else
        hipLaunchKernelGGL(k_dwq_bwd_z<4>, dim3((unsigned)gx_, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, xc, w, bias, g,
                           gz, B, C, M, K, dil, pad, ld_xc, ld_g, ld_gz, act, slope, qmin_x, qmax_x, qmin, qmax, gacc, gbias);
    return launch_status("fqss_dwq_bwd_z");
}

extern "C" int fqss_dwq_bwd_w(const float* gz, const uint8_t* xc, const float* qmin_x, const float* qmax_x, float* gw, int B,
                              int C, int M, int K, int dil, int pad, int64_t ld_gz, int64_t ld_xc, fqss_stream_t stream) {
    if (B == 0 || M == 0) return FQSS_OK;   // empty input: nothing to do (a 0-element tensor has a null data pointer)
    FQSS_REQUIRE(gz && xc && qmin_x && qmax_x && gw, "null pointer");
    FQSS_REQUIRE(B >= 0 && B <= 65535 && C > 0 && M >= 0 && K > 0 && K <= kTaps && dil > 0 && pad >= 0, "bad shape");
    FQSS_REQUIRE(ld_gz >= ((M + 3) & ~3) && ld_gz % 4 == 0 && aligned16(gz) && codes_ok(xc, ld_xc) && ld_xc >= M,
                 "rows must be 16-B aligned");
    if (B == 0 || M == 0) return FQSS_OK;
    hipLaunchKernelGGL(k_dwq_bwd_w, dim3((unsigned)C, (unsigned)B), dim3(256), 0, (hipStream_t)stream, gz, xc, gw, C, M, K, dil,
                       pad, ld_gz, ld_xc, qmin_x, qmax_x);
    return launch_status("fqss_dwq_bwd_w");
}

static int dwq_bwd_impl(const char* who, const uint8_t* xc, const float* qmin_x, const float* qmax_x, const float* w, const float* bias,
                        const float* g, float* gx, float* gw, int B, int C, int M, int K, int dil, int pad, int64_t ld_xc,
                        int64_t ld_g, int64_t ld_gx, int act, const float* slope, const float* qmin, const float* qmax,
                        double* gacc, float* gbias, const FqssGnAfter* ga, const FqssGnBefore* gb, fqss_stream_t stream) {
    if (B == 0 || M == 0) return FQSS_OK;   // empty input: nothing to do (a 0-element tensor has a null data pointer)
    FQSS_REQUIRE(xc && qmin_x && qmax_x && w && g && qmin && qmax && gacc, "null pointer");
    FQSS_REQUIRE(B >= 0 && C > 0 && M >= 0 && K > 0 && K <= kTaps && dil > 0 && 2 * pad == dil * (K - 1), "bad conv geometry");
    FQSS_REQUIRE(M <= kDwRowMax, "row too long for the single-workgroup backward (use fqss_dwq_bwd_z / dwconv_bwd_x / dwq_bwd_w)");
    FQSS_REQUIRE(ld_xc >= M && codes_ok(xc, ld_xc) && aligned16(g) && ld_g % 4 == 0 && ld_g >= ((M + 3) & ~3), "rows must be 16-B aligned");
    FQSS_REQUIRE(!gx || (aligned16(gx) && ld_gx % 4 == 0 && ld_gx >= ((M + 3) & ~3)), "gx rows must be 16-B aligned");
    FQSS_REQUIRE(act != FQSS_ACT_PRELU || slope, "PReLU needs a slope");
    FQSS_REQUIRE((int64_t)B * C < (1ll << 31), "too many rows");
    if (B == 0 || M == 0) return FQSS_OK;
    const size_t lds = (size_t)((M + 3) & ~3) * sizeof(float);
    DwGnAfter A{};
    DwGnBefore Bf{};
    if (ga != nullptr) {
        FQSS_REQUIRE(K == 3, "GroupNormQ hand-over: the 3-tap depthwise layer only");
        FQSS_REQUIRE(ga->gamma && ga->beta && ga->mean_rstd && ga->ws && ga->qmin && ga->qmax && ga->ggamma && ga->gbeta, "gn_after: null pointer");
        A = DwGnAfter{ga->gamma, ga->beta, ga->mean_rstd, ga->ws, ga->qmin, ga->qmax, ga->ggamma, ga->gbeta, B};
    }
    if (gb != nullptr) {
        FQSS_REQUIRE(K == 3 && gx, "GroupNormQ hand-over: the 3-tap depthwise layer only, with gx");
        FQSS_REQUIRE(gb->xc0 && gb->qmin0 && gb->qmax0 && gb->gamma && gb->beta && gb->mean_rstd && gb->ws && gb->gacc, "gn_before: null pointer");
        FQSS_REQUIRE(gb->ld_xc0 >= M && codes_ok(gb->xc0, gb->ld_xc0), "gn_before: code rows must be 16-B aligned");
        Bf = DwGnBefore{gb->xc0, gb->ld_xc0, gb->qmin0, gb->qmax0, gb->gamma, gb->beta, gb->mean_rstd, gb->ws, gb->gacc};
    }
#define FQSS_DWQ_BWD(KT, GA_, GB_, ...)                                                                                                          \
    hipLaunchKernelGGL((k_dwq_bwd<KT, GA_, GB_, ##__VA_ARGS__>), dim3((unsigned)(B * C)), dim3(256), lds, (hipStream_t)stream, xc, w, bias, g, gx, gw, C, M, \
                       K, dil, pad, ld_xc, ld_g, ld_gx, act, slope, qmin_x, qmax_x, qmin, qmax, gacc, gbias, gx != nullptr ? 1 : 0, A, Bf)
    if (K == 3) {
        if (ga && gb && act == FQSS_ACT_PRELU) FQSS_DWQ_BWD(3, true, true, FQSS_ACT_PRELU);      // the TCN block's layer
        else if (ga && gb) FQSS_DWQ_BWD(3, true, true);
        else if (ga) FQSS_DWQ_BWD(3, true, false);
        else if (gb) FQSS_DWQ_BWD(3, false, true);
        else FQSS_DWQ_BWD(3, false, false);
    } else {
        FQSS_DWQ_BWD(0, false, false);
    }
#undef FQSS_DWQ_BWD
    return launch_status(who);
}

extern "C" int fqss_dwq_bwd(const uint8_t* xc, const float* qmin_x, const float* qmax_x, const float* w, const float* bias,
                            const float* g, float* gx, float* gw, int B, int C, int M, int K, int dil, int pad, int64_t ld_xc,
                            int64_t ld_g, int64_t ld_gx, int act, const float* slope, const float* qmin, const float* qmax,
                            double* gacc, float* gbias, fqss_stream_t stream) {
    return dwq_bwd_impl("fqss_dwq_bwd", xc, qmin_x, qmax_x, w, bias, g, gx, gw, B, C, M, K, dil, pad, ld_xc, ld_g, ld_gx, act, slope, qmin,
                        qmax, gacc, gbias, nullptr, nullptr, stream);
}

/* fqss_dwq_bwd with the GroupNormQ layers around the depthwise layer handing over half of their backward (k_dwq_bwd<3, GA, GB>):
 * after (nullable): g is the gradient w.r.t. the OUTPUT of the GroupNormQ that consumes this layer's output, whose rows pass
 * (fqss_gnq_bwd_rows) has run: its apply pass happens on load.  before (nullable): the rows pass of the GroupNormQ that produced xc
 * is taken on gx: ws / gacc of that layer are written here, its apply pass (fqss_gnq_bwd_apply) follows. */
extern "C" int fqss_dwq_bwd_gn(const uint8_t* xc, const float* qmin_x, const float* qmax_x, const float* w, const float* bias,
                               const float* g, float* gx, float* gw, int B, int C, int M, int K, int dil, int pad, int64_t ld_xc,
                               int64_t ld_g, int64_t ld_gx, int act, const float* slope, const float* qmin, const float* qmax,
                               double* gacc, float* gbias, const FqssGnAfter* after, const FqssGnBefore* before, fqss_stream_t stream) {
    return dwq_bwd_impl("fqss_dwq_bwd_gn", xc, qmin_x, qmax_x, w, bias, g, gx, gw, B, C, M, K, dil, pad, ld_xc, ld_g, ld_gx, act, slope,
                        qmin, qmax, gacc, gbias, after, before, stream);
}

extern "C" int fqss_ewq_fwd(const uint8_t* ac, const float* amin, const float* amax, const uint8_t* bc, const float* bmin,
                            const float* bmax, const float* bf, float sb, uint8_t* yc, float* yout, int64_t rows, int64_t cols,
                            int64_t ld_a, int64_t ld_b, int64_t ld_bf, int64_t ld_y, int64_t ld_out, int act, const float* slope,
                            const float* qmin, const float* qmax, fqss_stream_t stream) {
    if (rows == 0 || cols == 0) return FQSS_OK;   // empty input: nothing to do (a 0-element tensor has a null data pointer)
    FQSS_REQUIRE(ac && amin && amax && yc && qmin && qmax, "null pointer");
    FQSS_REQUIRE(!(bc && bf), "second operand is either codes or fp32");
    FQSS_REQUIRE(!bc || (bmin && bmax), "coded second operand needs its ranges");
    FQSS_REQUIRE(rows >= 0 && cols >= 0 && rows < (1ll << 30) && ld_a < (1ll << 30), "bad shape");
    FQSS_REQUIRE(codes_ok(ac, ld_a) && codes_ok(yc, ld_y) && (!bc || codes_ok(bc, ld_b)) && ld_a >= cols && ld_y >= cols,
                 "code rows must be 16-B aligned");
    FQSS_REQUIRE(!bf || (aligned16(bf) && ld_bf % 4 == 0 && ld_bf >= ((cols + 3) & ~3)), "bad fp32 operand rows");
    FQSS_REQUIRE(!yout || (aligned16(yout) && ld_out % 4 == 0 && ld_out >= ((cols + 3) & ~3)), "bad fp32 output rows");
    FQSS_REQUIRE(act != FQSS_ACT_PRELU || slope, "PReLU needs a slope");
    if (rows == 0 || cols == 0) return FQSS_OK;
    hipLaunchKernelGGL(k_ewq_fwd, grid_rows(rows, cols, 16), dim3(256), 0, (hipStream_t)stream, ac, bc, bf, sb, yc, yout,
                       (int)rows, (int)cols, (int)ld_a, (int)ld_b, (int)ld_bf, (int)ld_y, (int)ld_out, act, slope, amin, amax,
                       bmin, bmax, qmin, qmax);
    return launch_status("fqss_ewq_fwd");
}

static int ewq_bwd_impl(const char* who, const uint8_t* ac, const float* amin, const float* amax, const uint8_t* bc, const float* bmin,
                            const float* bmax, const float* bf, float sb, const float* g, float* gz, int64_t rows, int64_t cols,
                            int64_t ld_a, int64_t ld_b, int64_t ld_bf, int64_t ld_g, int64_t ld_gz, int act, const float* slope,
                            const float* qmin, const float* qmax, double* gacc, const EwProducer& PA, const EwProducer& PB, int C,
                            fqss_stream_t stream) {
    if (rows == 0 || cols == 0) return FQSS_OK;   // empty input: nothing to do (a 0-element tensor has a null data pointer)
    FQSS_REQUIRE(ac && amin && amax && g && (gz || (PA.pz && PB.pz)) && qmin && qmax && gacc, "null pointer");
    if (gz == nullptr) ld_gz = 4 * ((cols + 3) / 4);
    FQSS_REQUIRE(!(bc && bf) && (!bc || (bmin && bmax)), "bad second operand");
    FQSS_REQUIRE(rows >= 0 && cols >= 0 && rows < (1ll << 30) && ld_a < (1ll << 30), "bad shape");
    FQSS_REQUIRE(codes_ok(ac, ld_a) && (!bc || codes_ok(bc, ld_b)) && aligned16(g) && (!gz || aligned16(gz)) && ld_g % 4 == 0 &&
                     ld_gz % 4 == 0 && ld_g >= ((cols + 3) & ~3) && ld_gz >= ((cols + 3) & ~3), "rows must be 16-B aligned");
    FQSS_REQUIRE(!bf || (aligned16(bf) && ld_bf % 4 == 0 && ld_bf >= ((cols + 3) & ~3)), "bad fp32 operand rows");
    FQSS_REQUIRE(act != FQSS_ACT_PRELU || slope, "PReLU needs a slope");
    if (rows == 0 || cols == 0) return FQSS_OK;
    int64_t gx_ = cdiv(cols, 256 * 4);
    if (gx_ > 64) gx_ = 64;
    int64_t gy = kSlots / gx_;
    if (gy > rows) gy = rows;
    // fused: C workgroup rows (each workgroup then serves one channel: bias sums reduced once per workgroup)
    // (more workgroup rows -- 4 C instead of C -- measured SLOWER, 33 -> 43 us: the per-workgroup reductions and atomics dominate)
    if ((PA.pz || PB.pz) && C <= gy && C > 1) gy = C;
    else if ((PA.pz || PB.pz) && C > 1 && gy % C == 0) gy -= 1;    // never alias the per-channel mode by accident
    const bool plain = act == FQSS_ACT_NONE && bf == nullptr && (!PA.pz || PA.act == FQSS_ACT_NONE) && (!PB.pz || PB.act == FQSS_ACT_NONE);
    if (plain)
        hipLaunchKernelGGL(k_ewq_bwd<true>, dim3((unsigned)gx_, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, ac, bc, bf, sb, g, gz,
                           (int)rows, (int)cols, (int)ld_a, (int)ld_b, (int)ld_bf, (int)ld_g, (int)ld_gz, act, slope, amin, amax,
                           bmin, bmax, qmin, qmax, gacc, PA, PB, C > 0 ? C : 1);
    else
        hipLaunchKernelGGL(k_ewq_bwd<false>, dim3((unsigned)gx_, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, ac, bc, bf, sb, g, gz,
                           (int)rows, (int)cols, (int)ld_a, (int)ld_b, (int)ld_bf, (int)ld_g, (int)ld_gz, act, slope, amin, amax,
                           bmin, bmax, qmin, qmax, gacc, PA, PB, C > 0 ? C : 1);
    return launch_status(who);
}

extern "C" int fqss_ewq_bwd(const uint8_t* ac, const float* amin, const float* amax, const uint8_t* bc, const float* bmin,
                            const float* bmax, const float* bf, float sb, const float* g, float* gz, int64_t rows, int64_t cols,
                            int64_t ld_a, int64_t ld_b, int64_t ld_bf, int64_t ld_g, int64_t ld_gz, int act, const float* slope,
                            const float* qmin, const float* qmax, double* gacc, fqss_stream_t stream) {
    return ewq_bwd_impl("fqss_ewq_bwd", ac, amin, amax, bc, bmin, bmax, bf, sb, g, gz, rows, cols, ld_a, ld_b, ld_bf, ld_g, ld_gz, act,
                        slope, qmin, qmax, gacc, EwProducer{}, EwProducer{}, 1, stream);
}

/* operands a / b that are outputs of pointwise convs: p?_z != NULL runs that producer's epilogue backward here (its gz ->
 * p?_out, partials -> p?_gacc / p?_gbias[C]); gz (plain dL/d(a+sb*b)) may be NULL when both operands are fused */
extern "C" int fqss_ewq_bwd_p(const uint8_t* ac, const float* amin, const float* amax, const uint8_t* bc, const float* bmin,
                              const float* bmax, float sb, const float* g, float* gz, int64_t rows, int64_t cols, int64_t ld_a,
                              int64_t ld_b, int64_t ld_g, int64_t ld_gz, int act, const float* slope, const float* qmin,
                              const float* qmax, double* gacc, int C, const float* pa_z, int64_t ld_paz, int pa_act,
                              const float* pa_slope, double* pa_gacc, float* pa_gbias, float* pa_out, int64_t ld_pa_out,
                              const float* pb_z, int64_t ld_pbz, int pb_act, const float* pb_slope, double* pb_gacc,
                              float* pb_gbias, float* pb_out, int64_t ld_pb_out, fqss_stream_t stream) {
    FQSS_REQUIRE(pa_z || pb_z, "no producer given");
    FQSS_REQUIRE(!pb_z || (bc && sb == 1.0f), "operand b can only be fused for a coded b with sb == 1");
    FQSS_REQUIRE(C > 0 && rows % C == 0, "rows must be batch x channels");
    const int64_t c4 = (cols + 3) & ~3ll;
    FQSS_REQUIRE(!pa_z || (pa_gacc && pa_out && aligned16(pa_z) && aligned16(pa_out) && ld_paz % 4 == 0 && ld_pa_out % 4 == 0 &&
                           ld_paz >= c4 && ld_pa_out >= c4 && (pa_act != FQSS_ACT_PRELU || pa_slope)), "bad producer a");
    FQSS_REQUIRE(!pb_z || (pb_gacc && pb_out && aligned16(pb_z) && aligned16(pb_out) && ld_pbz % 4 == 0 && ld_pb_out % 4 == 0 &&
                           ld_pbz >= c4 && ld_pb_out >= c4 && (pb_act != FQSS_ACT_PRELU || pb_slope)), "bad producer b");
    EwProducer PA{pa_z, (int)ld_paz, pa_act, pa_slope, pa_gacc, pa_gbias, pa_out, (int)ld_pa_out};
    EwProducer PB{pb_z, (int)ld_pbz, pb_act, pb_slope, pb_gacc, pb_gbias, pb_out, (int)ld_pb_out};
    return ewq_bwd_impl("fqss_ewq_bwd_p", ac, amin, amax, bc, bmin, bmax, nullptr, sb, g, gz, rows, cols, ld_a, ld_b, 0, ld_g, ld_gz,
                        act, slope, qmin, qmax, gacc, PA, PB, C, stream);
}

/* the backward of a chain of AddQ layers whose outputs feed the next add alone (k_ewq_chain_bwd above); levels[0] = the first add of
 * the forward.  fqss_add_chain_ok: the shapes the kernel serves (the per-level launch geometry of fqss_ewq_bwd_p: C workgroup rows). */
extern "C" int fqss_add_chain_ok(int64_t rows, int64_t cols, int C, int nlev) {
    if (nlev < 2 || nlev > kChainMax || C <= 1 || rows <= 0 || cols <= 0 || rows % C != 0 || rows / C > kChainRows) return 0;
    int64_t gx_ = cdiv(cols, 256 * 4);
    if (gx_ > 64) return 0;
    return (C <= kSlots / gx_) ? 1 : 0;
}

extern "C" int fqss_add_chain_bwd(const FqssAddChainLevel* levels, int nlev, const float* g, int64_t ld_g, float* ga_out, int64_t ld_ga,
                                  const float* az, int64_t ld_az, float* aout, int64_t ld_aout, double* agacc, float* agbias,
                                  int64_t rows, int64_t cols, int C, int64_t ld_a, int64_t ld_b, int64_t ld_bz, int64_t ld_bout,
                                  fqss_stream_t stream) {
    FQSS_REQUIRE(levels && g, "null pointer");
    FQSS_REQUIRE(fqss_add_chain_ok(rows, cols, C, nlev), "shape not served by the chain kernel (fqss_add_chain_ok)");
    const int64_t c4 = (cols + 3) & ~3ll;
    FQSS_REQUIRE(aligned16(g) && ld_g % 4 == 0 && ld_g >= c4 && ld_a % 16 == 0 && ld_b % 16 == 0 && ld_a >= cols && ld_b >= cols &&
                     ld_bz % 4 == 0 && ld_bz >= c4 && ld_bout % 4 == 0 && ld_bout >= c4, "rows must be 16-B aligned");
    FQSS_REQUIRE((az != nullptr) != (ga_out != nullptr), "the bottom level: EITHER a producer for its first operand (az ...) OR ga_out");
    FQSS_REQUIRE(!az || (aout && agacc && aligned16(az) && aligned16(aout) && ld_az % 4 == 0 && ld_az >= c4 && ld_aout % 4 == 0 && ld_aout >= c4),
                 "bad producer of the bottom level's first operand");
    FQSS_REQUIRE(!ga_out || (aligned16(ga_out) && ld_ga % 4 == 0 && ld_ga >= c4), "bad ga_out rows");
    ChainArgs A{};
    A.nlev = nlev; A.rows = (int)rows; A.cols = (int)cols; A.C = C;
    A.ld_a = (int)ld_a; A.ld_b = (int)ld_b; A.ld_bz = (int)ld_bz; A.ld_bout = (int)ld_bout; A.ld_g = (int)ld_g; A.ld_ga = (int)ld_ga;
    A.ld_az = (int)ld_az; A.ld_aout = (int)ld_aout;
    A.g = g; A.ga_out = ga_out; A.az = az; A.aout = aout; A.agacc = agacc; A.agbias = agbias;
    for (int l = 0; l < nlev; ++l) {
        const FqssAddChainLevel& f = levels[l];
        FQSS_REQUIRE(f.ac && f.bc && f.bz && f.bout && f.amin && f.amax && f.bmin && f.bmax && f.qmin && f.qmax && f.gacc && f.bgacc,
                     "level: null pointer");
        FQSS_REQUIRE(aligned16(f.ac) && aligned16(f.bc) && aligned16(f.bz) && aligned16(f.bout), "level: rows must be 16-B aligned");
        A.lv[l] = ChainLevel{f.ac, f.bc, f.bz, f.bout, f.amin, f.amax, f.bmin, f.bmax, f.qmin, f.qmax, f.gacc, f.bgacc, f.bgbias};
    }
    const int64_t gx_ = cdiv(cols, 256 * 4);
    hipLaunchKernelGGL(k_ewq_chain_bwd, dim3((unsigned)gx_, (unsigned)C), dim3(256), 0, (hipStream_t)stream, A);
    return launch_status("fqss_add_chain_bwd");
}

extern "C" int fqss_mulq_fwd(const uint8_t* mc, const float* mmin, const float* mmax, const uint8_t* fc, const float* fmin,
                             const float* fmax, uint8_t* yc, float* yout, int B, int S, int C, int M, int64_t ld_m, int64_t ld_f,
                             int64_t ld_y, int64_t ld_out, const float* qmin, const float* qmax, fqss_stream_t stream) {
    if (B == 0 || M == 0) return FQSS_OK;   // empty input: nothing to do (a 0-element tensor has a null data pointer)
    FQSS_REQUIRE(mc && mmin && mmax && fc && fmin && fmax && yc && qmin && qmax, "null pointer");
    FQSS_REQUIRE(B > 0 && S >= 1 && S <= kMulqSMax && C > 0 && M > 0 && (int64_t)B * S * C < (1ll << 31), "bad shape (1 .. 4 sources)");
    FQSS_REQUIRE(codes_ok(mc, ld_m) && codes_ok(fc, ld_f) && codes_ok(yc, ld_y) && ld_m >= M && ld_f >= M && ld_y >= M,
                 "code rows must be 16-B aligned");
    FQSS_REQUIRE(!yout || (aligned16(yout) && ld_out % 4 == 0 && ld_out >= ((M + 3) & ~3)), "bad fp32 output rows");
    const int64_t rows = (int64_t)B * C;
#define FQSS_MULQ_FWD(S_)                                                                                                          \
    hipLaunchKernelGGL(k_mulq_fwd<S_>, grid_rows(rows, M, 16), dim3(256), 0, (hipStream_t)stream, mc, fc, yc, yout, (int)rows, C, M, ld_m, \
                       ld_f, ld_y, ld_out, mmin, mmax, fmin, fmax, qmin, qmax)
    switch (S) {
        case 1: FQSS_MULQ_FWD(1); break;
        case 2: FQSS_MULQ_FWD(2); break;
        case 3: FQSS_MULQ_FWD(3); break;
        default: FQSS_MULQ_FWD(4); break;
    }
#undef FQSS_MULQ_FWD
    return launch_status("fqss_mulq_fwd");
}

extern "C" int fqss_mulq_bwd(const uint8_t* mc, const float* mmin, const float* mmax, const uint8_t* fc, const float* fmin,
                             const float* fmax, const float* g, float* gmask, float* gfeat, int B, int S, int C, int M, int64_t ld_m,
                             int64_t ld_f, int64_t ld_g, int64_t ld_gm, int64_t ld_gf, const float* qmin, const float* qmax,
                             double* gacc, const float* pz, int64_t ld_pz, int pact, const float* pslope, double* pgacc, float* pgbias,
                             fqss_stream_t stream) {
    if (B == 0 || M == 0) return FQSS_OK;   // empty input: nothing to do (a 0-element tensor has a null data pointer)
    FQSS_REQUIRE(mc && mmin && mmax && fc && fmin && fmax && g && gmask && qmin && qmax && gacc, "null pointer");
    FQSS_REQUIRE(B > 0 && S >= 1 && S <= kMulqSMax && C > 0 && M > 0 && (int64_t)B * S * C < (1ll << 31), "bad shape (1 .. 4 sources)");
    const int64_t m4 = (M + 3) & ~3;
    FQSS_REQUIRE(codes_ok(mc, ld_m) && codes_ok(fc, ld_f) && ld_m >= M && ld_f >= M, "code rows must be 16-B aligned");
    FQSS_REQUIRE(aligned16(g) && aligned16(gmask) && (!gfeat || aligned16(gfeat)) && ld_g % 4 == 0 && ld_gm % 4 == 0 && ld_g >= m4 &&
                     ld_gm >= m4 && (!gfeat || (ld_gf % 4 == 0 && ld_gf >= m4)), "fp32 rows must be 16-B aligned");
    FQSS_REQUIRE(!pz || (aligned16(pz) && ld_pz % 4 == 0 && ld_pz >= m4 && pgacc), "bad producer operand");
    FQSS_REQUIRE(!pz || pact == FQSS_ACT_NONE || pact == FQSS_ACT_RELU || (pact == FQSS_ACT_PRELU && pslope), "producer non-linearity: none / ReLU / PReLU");
    const int64_t rows = (int64_t)B * C;
    int64_t gx_ = cdiv(M, 256 * 4);
    if (gx_ > 64) gx_ = 64;
    int64_t gy = kSlots / gx_;
    if (gy > rows) gy = rows;
    if (gy >= C) gy = gy / C * C;     // a multiple of C whenever C fits: the kernel's order-independent per-channel bias sums (per_channel)
    EwProducer P{};
    P.pz = pz; P.ld_pz = (int)ld_pz; P.act = pact; P.slope = pslope; P.gacc = pgacc; P.gbias = pgbias; P.out = nullptr; P.ld_out = 0;
#define FQSS_MULQ_BWD(S_, PR_)                                                                                                       \
    hipLaunchKernelGGL((k_mulq_bwd<S_, PR_>), dim3((unsigned)gx_, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, mc, fc, g, gmask,   \
                       gfeat, (int)rows, C, M, ld_m, ld_f, ld_g, ld_gm, ld_gf, mmin, mmax, fmin, fmax, qmin, qmax, gacc, P)
    if (pz != nullptr) {
        switch (S) {
            case 1: FQSS_MULQ_BWD(1, true); break;
            case 2: FQSS_MULQ_BWD(2, true); break;
            case 3: FQSS_MULQ_BWD(3, true); break;
            default: FQSS_MULQ_BWD(4, true); break;
        }
    } else {
        switch (S) {
            case 1: FQSS_MULQ_BWD(1, false); break;
            case 2: FQSS_MULQ_BWD(2, false); break;
            case 3: FQSS_MULQ_BWD(3, false); break;
            default: FQSS_MULQ_BWD(4, false); break;
        }
    }
#undef FQSS_MULQ_BWD
    return launch_status("fqss_mulq_bwd");
}
