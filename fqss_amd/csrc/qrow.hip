// qrow.hip -- row-major linears of the dual-path STUDENT on the integer matrix cores: in the quantizing phase both operands
// sit on 8-bit grids (x = dx * c + min_x with u8 codes c from the producing layer's quantizer, W_q = dw[o] * k with int8 codes k),
// so   z[r][o] = sum_i W_q[o][i] x[r][i] + b[o] = dw[o] * (dx * S[r][o] + min_x * R[o]) + b[o],
//      S = sum_i k[o][i] c[r][i]   (an exact integer),   R[o] = sum_i k[o][i]   (fqss_wq_codes)
// -- the arithmetic of csrc/qgemm.hip (ConvTasNet's channel-first q-GEMM), here for row matrices [R][Ci] with the codes of a row
// contiguous: S runs on v_mfma_i32_32x32x32_i8 (16 code bytes per lane and operand), u8 codes are recentred to int8 by one XOR
// per four codes (c - 128) and the shift comes back through R:  S = S' + 128 R.  The kernel reads 1 B per activation and is bound
// by writing z (fp32, kept for the backward); the backward stays on the fp32-equivalent row GEMMs (csrc/gemm_x3.hip).
//
// Reference replaced: F.linear on fake-quantized operands in LinearQ / MultiheadAttentionQ (qat_layers.py:521-536, 889-901, 941).
#include "fqss_dev.h"

namespace fqss {

typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int QK = 64;             // code bytes of K per LDS tile (two MFMA k-steps)
constexpr int QLD = QK + 16;       // LDS row stride in bytes (16-B reads of 32 rows spread over the banks)

struct QRowArgs {
    const uint8_t* xc;    // [R][ld_x] u8 codes
    const int8_t* wk;     // [Co][Ci] int8 codes
    const float* dw;      // [Co]
    const float* rw;      // [Co] sum of codes per output row
    const float* bias;    // [Co] or null
    const float* qmin;
    const float* qmax;
    float* z;             // [R][ld_z]
    int64_t R, ld_x, ld_z;
    int Ci, Co;
    // fused output quantizer (fqss_qrow_fwdq; y null: plain fqss_qrow_fwd): y = fq(act(z)) with the layer's own activation quantizer,
    // operation for operation what fqss_actq_fwd computes from z -- the separate pass over z disappears
    float* y;
    int64_t ld_y;
    int act;
    const float* slope;
    const float* qy_min;
    const float* qy_max;
    // act = FQSS_ACT_RELU_Q: a second quantizer behind the first and a ReLU (NlQ(ReLU) after a LinearQ): y = fq2(relu(fq(z))), codes to yc
    const float* qy2_min;
    const float* qy2_max;
    uint8_t* yc;
    int64_t ld_yc;
};

// workgroup tile 128 (rows r) x 128 (outputs o), 4 waves of 64 x 64 (2 x 2 MFMA tiles of 32 x 32); K in chunks of 64 bytes
__global__ __launch_bounds__(256, 3) void k_qrow_fwd(QRowArgs g) {     // (.., 3 waves per SIMD): <= 168 registers, a third workgroup per CU
    __shared__ __attribute__((aligned(16))) uint8_t As[128][QLD];
    __shared__ __attribute__((aligned(16))) uint8_t Bs[128][QLD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t r0 = (int64_t)blockIdx.y * 128;
    const int o0 = blockIdx.x * 128;
    i32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0;
    // staging: 128 rows x 64 B = 512 16-B pieces per operand -> 2 per thread (row = piece / 4, 16-B column = piece % 4)
    const int lr = lane & 31, lh = lane >> 5;
    const int nk = (g.Ci + QK - 1) / QK;
    uint4 ra[2], rb[2];
    auto load = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int f = tid + 256 * p, row = f >> 2, kc = (f & 3) * 16;
            const int64_t rr = min(r0 + row, g.R - 1);
            const int oo = min(o0 + row, g.Co - 1);
            const int kk = min(k0 + kc, g.Ci - 16);                      // unconditional, clamped (Ci is a multiple of 16)
            ra[p] = *reinterpret_cast<const uint4*>(g.xc + rr * g.ld_x + kk);
            rb[p] = *reinterpret_cast<const uint4*>(g.wk + (int64_t)oo * g.Ci + kk);
        }
    };
    auto store = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int f = tid + 256 * p, row = f >> 2, kc = (f & 3) * 16;
            const bool kv = k0 + kc < g.Ci;                               // a chunk past the end contributes zeros
            uint4 a = ra[p], b = rb[p];
            a.x ^= 0x80808080u; a.y ^= 0x80808080u; a.z ^= 0x80808080u; a.w ^= 0x80808080u;   // u8 c -> int8 (c - 128)
            if (!kv) { a = make_uint4(0, 0, 0, 0); b = make_uint4(0, 0, 0, 0); }
            *reinterpret_cast<uint4*>(&As[row][kc]) = a;
            *reinterpret_cast<uint4*>(&Bs[row][kc]) = b;
        }
    };
    load(0);
    store(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        load((kt + 1) * QK);
#pragma unroll
        for (int ks = 0; ks < QK / 32; ++ks) {
            i32x4 af[2], bf[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) af[mi] = *reinterpret_cast<const i32x4*>(&As[wm * 64 + mi * 32 + lr][ks * 32 + 16 * lh]);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) bf[ni] = *reinterpret_cast<const i32x4*>(&Bs[wn * 64 + ni * 32 + lr][ks * 32 + 16 * lh]);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
        }
        __syncthreads();
        if (kt + 1 < nk) {
            store((kt + 1) * QK);
            __syncthreads();
        }
    }
    // epilogue: rows r = A index, columns o = B index.  z = dw[o] * (dx * S + min_x * R[o]) + b[o],  S = S' + 128 R[o]
    const float lo = *g.qmin, hi = *g.qmax;
    const float dx = (hi - lo) / 255.0f;
    QRange ry{0.0f, 1.0f, 1.0f};
    float slope = 0.0f;
    if (g.y != nullptr) {
        ry = load_qrange(g.qy_min, g.qy_max);
        if (g.act == FQSS_ACT_PRELU) slope = *g.slope;
    }
    const bool two = g.y != nullptr && g.act == FQSS_ACT_RELU_Q;          // y = fq2(relu(fq(z))) + codes (host: 16-B rows only)
    const bool post = g.y != nullptr && (g.act == FQSS_ACT_POST_RELU || two);      // y = relu(fq(z)): the ReLU BEHIND the quantizer
    QRange ry2{0.0f, 1.0f, 1.0f};
    if (two) ry2 = load_qrange(g.qy2_min, g.qy2_max);
    // results leave through a wave-private LDS tile as whole 128-B rows, 16 B per lane (the lane-per-column layout of the MFMA result needs
    // 16 strided 4-B stores per 32 x 32 tile and operand: this kernel is bound by writing z -- and now y); 16-B aligned output rows
    // only, the scalar form below serves the rest
    __shared__ __attribute__((aligned(16))) float Tz[4][32][36];
    const bool rows16 = (g.ld_z & 3) == 0 && (reinterpret_cast<uintptr_t>(g.z) & 15u) == 0 &&
                        (g.y == nullptr || ((g.ld_y & 3) == 0 && (reinterpret_cast<uintptr_t>(g.y) & 15u) == 0));
    if (rows16) {
        const int c4 = (lane & 7) * 4, rq = lane >> 3;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int ob = o0 + wn * 64 + ni * 32;
            if (ob >= g.Co) continue;
            const int o = ob + lr;
            const bool ov = o < g.Co;
            const float dwo = ov ? g.dw[o] : 0.f, rwo = ov ? g.rw[o] : 0.f, bo = (ov && g.bias) ? g.bias[o] : 0.f;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int64_t rb = r0 + wm * 64 + mi * 32;
                if (rb >= g.R) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float S = (float)acc[mi][ni][r] + 128.0f * rwo;
                    float v = dwo * (dx * S + lo * rwo);
                    if (g.bias) v = v + bo;
                    Tz[wave][(r & 3) + 8 * (r >> 2) + 4 * lh][lr] = v;
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int rl = rq + 8 * it;
                    const int64_t row = rb + rl;
                    const float4 v = *reinterpret_cast<const float4*>(&Tz[wave][rl][c4]);
                    if (row < g.R) {
                        const float e[4] = {v.x, v.y, v.z, v.w};
                        float yq[4];
                        if (g.y != nullptr) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                float c, u;
                                bool inr;
                                { const float yv = fq_asym(post ? e[q] : act_apply(e[q], g.act, slope), ry, c, u, inr); yq[q] = (post && !(yv > 0.0f)) ? 0.0f : yv; }
                            }
                            if (two) {       // (workgroup-uniform) NlQ's quantizer on relu(fq(z)); host: Co % 4 == 0
                                unsigned int pk = 0;
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    float c, u;
                                    bool inr;
                                    yq[q] = fq_asym(yq[q], ry2, c, u, inr);
                                    pk = pack_code(c, q, pk);
                                }
                                if (ob + c4 + 3 < g.Co) *reinterpret_cast<unsigned int*>(g.yc + row * g.ld_yc + ob + c4) = pk;
                            }
                        }
                        if (ob + c4 + 3 < g.Co) {
                            *reinterpret_cast<float4*>(g.z + row * g.ld_z + ob + c4) = v;
                            if (g.y != nullptr) *reinterpret_cast<float4*>(g.y + row * g.ld_y + ob + c4) = make_float4(yq[0], yq[1], yq[2], yq[3]);
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                if (ob + c4 + q < g.Co) {
                                    g.z[row * g.ld_z + ob + c4 + q] = e[q];
                                    if (g.y != nullptr) g.y[row * g.ld_y + ob + c4 + q] = yq[q];
                                }
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        return;
    }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int o = o0 + wn * 64 + ni * 32 + lr;
        const bool ov = o < g.Co;
        const float dwo = ov ? g.dw[o] : 0.f, rwo = ov ? g.rw[o] : 0.f, bo = (ov && g.bias) ? g.bias[o] : 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = r0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (ov && row < g.R) {
                    const float S = (float)acc[mi][ni][r] + 128.0f * rwo;
                    float v = dwo * (dx * S + lo * rwo);
                    if (g.bias) v = v + bo;
                    g.z[row * g.ld_z + o] = v;
                    if (g.y != nullptr) {
                        float c, u;
                        bool inr;
                        { const float yv = fq_asym(post ? v : act_apply(v, g.act, slope), ry, c, u, inr); g.y[row * g.ld_y + o] = (post && !(yv > 0.0f)) ? 0.0f : yv; }
                    }
                }
            }
    }
}

}  // namespace fqss

using namespace fqss;

extern "C" int fqss_qrow_fwd(const uint8_t* xc, const int8_t* wk, const float* dw, const float* rw, const float* bias,
                             const float* qmin_x, const float* qmax_x, float* z, int64_t R, int Ci, int Co, int64_t ld_x,
                             int64_t ld_z, fqss_stream_t stream) {
    FQSS_REQUIRE(xc && wk && dw && rw && qmin_x && qmax_x && z, "null tensor");
    FQSS_REQUIRE(R >= 0 && Ci >= 16 && Ci % 16 == 0 && Ci <= 2048 && Co > 0 && ld_x >= Ci && ld_x % 16 == 0 && ld_z >= Co,
                 "bad shape (Ci a multiple of 16, <= 2048; code rows 16-B aligned)");
    FQSS_REQUIRE(aligned16(xc) && aligned16(wk), "code images must be 16-B aligned");
    if (R == 0) return FQSS_OK;
    QRowArgs g{xc, wk, dw, rw, bias, qmin_x, qmax_x, z, R, ld_x, ld_z, Ci, Co, nullptr, 0, FQSS_ACT_NONE, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    dim3 grid((unsigned)cdiv(Co, 128), (unsigned)cdiv(R, 128));
    hipLaunchKernelGGL(k_qrow_fwd, grid, dim3(256), 0, (hipStream_t)stream, g);
    return launch_status("fqss_qrow_fwd");
}

// fqss_qrow_fwd + the layer's output quantizer in the epilogue: z (kept for the backward's STE) AND y = fq(act(z)) from one launch
extern "C" int fqss_qrow_fwdq(const uint8_t* xc, const int8_t* wk, const float* dw, const float* rw, const float* bias,
                              const float* qmin_x, const float* qmax_x, float* z, float* y, int64_t R, int Ci, int Co, int64_t ld_x,
                              int64_t ld_z, int64_t ld_y, int act, const float* slope, const float* qmin_y, const float* qmax_y,
                              fqss_stream_t stream) {
    FQSS_REQUIRE(xc && wk && dw && rw && qmin_x && qmax_x && z && y && qmin_y && qmax_y, "null tensor");
    FQSS_REQUIRE(R >= 0 && Ci >= 16 && Ci % 16 == 0 && Ci <= 2048 && Co > 0 && ld_x >= Ci && ld_x % 16 == 0 && ld_z >= Co && ld_y >= Co,
                 "bad shape (Ci a multiple of 16, <= 2048; code rows 16-B aligned)");
    FQSS_REQUIRE(aligned16(xc) && aligned16(wk), "code images must be 16-B aligned");
    FQSS_REQUIRE(act == FQSS_ACT_NONE || act == FQSS_ACT_RELU || act == FQSS_ACT_POST_RELU || (act == FQSS_ACT_PRELU && slope),
                 "activation: none, ReLU, PReLU (with its slope) or a ReLU behind the quantizer");
    if (R == 0) return FQSS_OK;
    QRowArgs g{xc, wk, dw, rw, bias, qmin_x, qmax_x, z, R, ld_x, ld_z, Ci, Co, y, ld_y, act, slope, qmin_y, qmax_y, nullptr, nullptr, nullptr, 0};
    dim3 grid((unsigned)cdiv(Co, 128), (unsigned)cdiv(R, 128));
    hipLaunchKernelGGL(k_qrow_fwd, grid, dim3(256), 0, (hipStream_t)stream, g);
    return launch_status("fqss_qrow_fwdq");
}

extern "C" int fqss_qrow_fwdq2(const uint8_t* xc, const int8_t* wk, const float* dw, const float* rw, const float* bias,
                               const float* qmin_x, const float* qmax_x, float* z, float* y, uint8_t* yc, int64_t R, int Ci, int Co,
                               int64_t ld_x, int64_t ld_z, int64_t ld_y, int64_t ld_yc, const float* qmin1, const float* qmax1,
                               const float* qmin2, const float* qmax2, fqss_stream_t stream) {
    FQSS_REQUIRE(xc && wk && dw && rw && qmin_x && qmax_x && z && y && yc && qmin1 && qmax1 && qmin2 && qmax2, "null tensor");
    FQSS_REQUIRE(R >= 0 && Ci >= 16 && Ci % 16 == 0 && Ci <= 2048 && Co > 0 && Co % 4 == 0 && ld_x >= Ci && ld_x % 16 == 0 && ld_z >= Co &&
                     ld_y >= Co && ld_yc >= Co,
                 "bad shape (Ci a multiple of 16, <= 2048; Co a multiple of 4; code rows 16-B aligned)");
    FQSS_REQUIRE(aligned16(xc) && aligned16(wk) && aligned16(z) && aligned16(y) && ld_z % 4 == 0 && ld_y % 4 == 0 &&
                     (reinterpret_cast<uintptr_t>(yc) & 3u) == 0 && ld_yc % 4 == 0,
                 "code images and output rows must be 16-B aligned, code rows 4-B aligned");
    if (R == 0) return FQSS_OK;
    QRowArgs g{xc, wk, dw, rw, bias, qmin_x, qmax_x, z, R, ld_x, ld_z, Ci, Co, y, ld_y, FQSS_ACT_RELU_Q, nullptr, qmin1, qmax1, qmin2, qmax2, yc, ld_yc};
    dim3 grid((unsigned)cdiv(Co, 128), (unsigned)cdiv(R, 128));
    hipLaunchKernelGGL(k_qrow_fwd, grid, dim3(256), 0, (hipStream_t)stream, g);
    return launch_status("fqss_qrow_fwdq2");
}
