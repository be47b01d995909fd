// gemm.hip -- K4/K5 (+ the weight-gradient of K11-K13): pointwise Conv1d as an fp32 GEMM on the
// matrix cores.  v_mfma_f32_32x32x2_f32 is an exact fp32 fma chain (no TF32 on gfx950), so the
// result is a true fp32 convolution like the reference's MKL-DNN path, only the summation order
// differs (parity gate G1).
//
// One kernel template, one LDS-tiled 128x128x16 block tile, 4 waves of 64x64 (2x2 MFMA tiles of
// 32x32), operands addressed through (batch, row, col) strides so the same code serves
//   fwd    z[b]  = W      * x[b]        A = W   (k contiguous)   B = x[b]   (j contiguous)
//   dgrad  gx[b] = W^T    * gz[b]       A = W^T (i contiguous)   B = gz[b]  (j contiguous)
//   wgrad  gW   += gz[b]  * x[b]^T      A = gz  (k contiguous)   B = x[b]^T (k contiguous), split-K + atomics
//   frames gW   += a[n]   * F(x[n])     A = a   (k contiguous)   B = frames of x (j contiguous, k stride = hop)
// LDS images are k-major (As[k][m], Bs[k][n]) so that every MFMA operand fetch is a conflict-free
// ds_read_b32 of 32 consecutive floats per half-wave.
//
// Reference replaced: F.conv1d(k=1) in Conv1dQ/Conv1dNlQ (qat_layers.py:137-146, 202-212) and its
// autograd (convolution_backward), F.conv1d/F.conv_transpose1d weight gradients of the
// encoder/decoder (qat_layers.py:1028-1039, 1330-1341, 1189-1202).
#include <stdlib.h>

#define FQSS_USES_GRAD_ADD   // the fp32 gradient atomics of this file go through grad_add (fqss_dev.h: FQSS_DETERMINISTIC=1)
#include "fqss_dev.h"

namespace fqss {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;  // [M] or null
    const float* bias_col;  // [N] or null (row-major "channels-last" linears: the bias runs along C's columns)
    int M, N, K;        // per-batch problem
    int64_t sAb, sAi, sAk;
    int64_t sBb, sBk, sBj;
    int64_t sCb, sCi;
    int ksplit, kchunk;  // split-K over blockIdx.z (ATOMIC only)
};

constexpr int BM = 128, BN = 128, BK = 16, LDT = 132;  // LDT: padded LDS row (floats); BM x BN is the LARGEST tile

// MI x NI = 32x32 MFMA tiles per wave (2 x 2 waves per workgroup): the workgroup tile is (64*MI) x (64*NI).
//   (2,2) 128x128  the ConvTasNet shapes;   (2,1) 128x64  N = 64 outputs (DPTNet's feature dim: a 128-wide tile would be half
//   empty);   (1,2) 64x128  problems with too few 128x128 tiles to fill 256 CUs (Sepformer: 8500 rows x 256 outputs = 134 tiles)
template <bool A_KC, bool B_KC, bool VEC, bool ATOMIC, int MI, int NI>
__global__ __launch_bounds__(256) void k_gemm_f32(GemmArgs g) {
    constexpr int BMt = 64 * MI, BNt = 64 * NI;
    __shared__ __attribute__((aligned(16))) float As[BK][LDT];
    __shared__ __attribute__((aligned(16))) float Bs[BK][LDT];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int bz = blockIdx.z;
    const int b = ATOMIC ? bz / g.ksplit : bz;
    const int ks = ATOMIC ? bz % g.ksplit : 0;
    const int kbeg = ATOMIC ? ks * g.kchunk : 0;
    const int kend = ATOMIC ? min(g.K, kbeg + g.kchunk) : g.K;
    const int i0 = blockIdx.y * BMt, j0 = blockIdx.x * BNt;

    const float* Ab = g.A + (int64_t)b * g.sAb;
    const float* Bb = g.B + (int64_t)b * g.sBb;

    float ra[4 * MI], rb[4 * NI];
    int kl = 0;     // k0 of the tile held in ra / rb

    auto load_tiles = [&](int k0) {
        kl = k0;
        // ---------------- A tile: BMt (i) x 16 (k)
        if constexpr (VEC) {
#pragma unroll
            for (int p = 0; p < MI; ++p) {
                const int f = tid + 256 * p;
                // unconditional loads from clamped addresses (validity is applied in store_tiles): a branch around a global
                // load makes the compiler drain the loads before the MFMA section instead of after it
                float4 v;
                if constexpr (A_KC) {
                    const int i = f >> 2, k = (f & 3) * 4;
                    v = *reinterpret_cast<const float4*>(Ab + (int64_t)min(i0 + i, g.M - 1) * g.sAi + min(k0 + k, ((g.K + 3) & ~3) - 4));
                } else {
                    const int k = f / (16 * MI), i = (f % (16 * MI)) * 4;
                    v = *reinterpret_cast<const float4*>(Ab + (int64_t)min(k0 + k, g.K - 1) * g.sAk + min(i0 + i, ((g.M + 3) & ~3) - 4));
                }
                ra[4 * p + 0] = v.x; ra[4 * p + 1] = v.y; ra[4 * p + 2] = v.z; ra[4 * p + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int p = 0; p < 4 * MI; ++p) {
                const int e = tid + 256 * p;
                int i, k;
                if constexpr (A_KC) { i = e >> 4; k = e & 15; } else { k = e / BMt; i = e % BMt; }
                float v = 0.f;
                if (i0 + i < g.M && k0 + k < kend) v = Ab[(int64_t)(i0 + i) * g.sAi + (int64_t)(k0 + k) * g.sAk];
                ra[p] = v;
            }
        }
        // ---------------- B tile: 16 (k) x BNt (j)
        if constexpr (VEC) {
#pragma unroll
            for (int p = 0; p < NI; ++p) {
                const int f = tid + 256 * p;
                float4 v;
                if constexpr (B_KC) {
                    const int j = f >> 2, k = (f & 3) * 4;
                    v = *reinterpret_cast<const float4*>(Bb + (int64_t)min(j0 + j, g.N - 1) * g.sBj + min(k0 + k, ((g.K + 3) & ~3) - 4));
                } else {
                    const int k = f / (16 * NI), j = (f % (16 * NI)) * 4;
                    v = *reinterpret_cast<const float4*>(Bb + (int64_t)min(k0 + k, g.K - 1) * g.sBk + min(j0 + j, ((g.N + 3) & ~3) - 4));
                }
                rb[4 * p + 0] = v.x; rb[4 * p + 1] = v.y; rb[4 * p + 2] = v.z; rb[4 * p + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int p = 0; p < 4 * NI; ++p) {
                const int e = tid + 256 * p;
                int j, k;
                if constexpr (B_KC) { j = e >> 4; k = e & 15; } else { k = e / BNt; j = e % BNt; }
                float v = 0.f;
                if (j0 + j < g.N && k0 + k < kend) v = Bb[(int64_t)(k0 + k) * g.sBk + (int64_t)(j0 + j) * g.sBj];
                rb[p] = v;
            }
        }
    };

    auto store_tiles = [&]() {
        if constexpr (VEC) {
#pragma unroll
            for (int p = 0; p < MI; ++p) {
                const int f = tid + 256 * p;
                if constexpr (A_KC) {
                    const int i = f >> 2, k = (f & 3) * 4;
                    const bool iv = i0 + i < g.M;
#pragma unroll
                    for (int e = 0; e < 4; ++e) As[k + e][i] = (iv && kl + k + e < kend) ? ra[4 * p + e] : 0.f;
                } else {
                    const int k = f / (16 * MI), i = (f % (16 * MI)) * 4;
                    const bool kv = kl + k < kend;
                    *reinterpret_cast<float4*>(&As[k][i]) = make_float4((kv && i0 + i < g.M) ? ra[4 * p] : 0.f, (kv && i0 + i + 1 < g.M) ? ra[4 * p + 1] : 0.f,
                                                                         (kv && i0 + i + 2 < g.M) ? ra[4 * p + 2] : 0.f, (kv && i0 + i + 3 < g.M) ? ra[4 * p + 3] : 0.f);
                }
            }
#pragma unroll
            for (int p = 0; p < NI; ++p) {
                const int f = tid + 256 * p;
                if constexpr (B_KC) {
                    const int j = f >> 2, k = (f & 3) * 4;
                    const bool jv = j0 + j < g.N;
#pragma unroll
                    for (int e = 0; e < 4; ++e) Bs[k + e][j] = (jv && kl + k + e < kend) ? rb[4 * p + e] : 0.f;
                } else {
                    const int k = f / (16 * NI), j = (f % (16 * NI)) * 4;
                    const bool kv = kl + k < kend;
                    *reinterpret_cast<float4*>(&Bs[k][j]) = make_float4((kv && j0 + j < g.N) ? rb[4 * p] : 0.f, (kv && j0 + j + 1 < g.N) ? rb[4 * p + 1] : 0.f,
                                                                         (kv && j0 + j + 2 < g.N) ? rb[4 * p + 2] : 0.f, (kv && j0 + j + 3 < g.N) ? rb[4 * p + 3] : 0.f);
                }
            }
        } else {
#pragma unroll
            for (int p = 0; p < 4 * MI; ++p) {
                const int e = tid + 256 * p;
                int i, k;
                if constexpr (A_KC) { i = e >> 4; k = e & 15; } else { k = e / BMt; i = e % BMt; }
                As[k][i] = ra[p];
            }
#pragma unroll
            for (int p = 0; p < 4 * NI; ++p) {
                const int e = tid + 256 * p;
                int j, k;
                if constexpr (B_KC) { j = e >> 4; k = e & 15; } else { k = e / BNt; j = e % BNt; }
                Bs[k][j] = rb[p];
            }
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int nkt = (kend - kbeg + BK - 1) / BK;
    if (nkt > 0) {
        load_tiles(kbeg);
        store_tiles();
    }
    __syncthreads();
    const int lr = lane & 31, lk = lane >> 5;
    for (int kt = 0; kt < nkt; ++kt) {
        // global loads fly under the MFMAs; the vector path issues them unconditionally (the last iteration re-reads a clamped tile)
        if (VEC || kt + 1 < nkt) load_tiles(kbeg + (kt + 1) * BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float av[MI], bv[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) av[mi] = As[kk + lk][wm * (32 * MI) + mi * 32 + lr];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) bv[ni] = Bs[kk + lk][wn * (32 * NI) + ni * 32 + lr];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
        }
        __syncthreads();
        if (kt + 1 < nkt) {
            store_tiles();
            __syncthreads();
        }
    }

    // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* Cb = g.C + (int64_t)b * g.sCb;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int col = j0 + wn * (32 * NI) + ni * 32 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * (32 * MI) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (row < g.M && col < g.N) {
                    float v = acc[mi][ni][r];
                    if (g.bias != nullptr && (!ATOMIC || kbeg == 0)) v = v + g.bias[row];
                    if (g.bias_col != nullptr && (!ATOMIC || kbeg == 0)) v = v + g.bias_col[col];
                    float* dst = Cb + (int64_t)row * g.sCi + col;
                    if constexpr (ATOMIC) grad_add(dst, v); else *dst = v;
                }
            }
        }
}

static inline int64_t rup4(int64_t v) { return (v + 3) & ~(int64_t)3; }

// a_kc / b_kc: which dimension of the operand is contiguous in memory
static int launch_gemm(const GemmArgs& g, bool a_kc, bool b_kc, bool atomic, int batch, hipStream_t s, const char* what) {
    if (g.M <= 0 || g.N <= 0 || batch <= 0) return FQSS_OK;
    // 16-B vector path: aligned bases, strides multiple of 4 floats, and the 4-wide over-read at the
    // edge of the contiguous dimension stays inside the (padded) row
    bool vec = aligned16(g.A) && aligned16(g.B) && (g.sAb % 4 == 0) && (g.sBb % 4 == 0);
    if (a_kc) vec = vec && (g.sAk == 1) && (g.sAi % 4 == 0) && (g.sAi >= rup4(g.K));
    else      vec = vec && (g.sAi == 1) && (g.sAk % 4 == 0) && (g.sAk >= rup4(g.M));
    if (b_kc) vec = vec && (g.sBk == 1) && (g.sBj % 4 == 0) && (g.sBj >= rup4(g.K));
    else      vec = vec && (g.sBj == 1) && (g.sBk % 4 == 0) && (g.N % 4 == 0 || g.sBk >= rup4(g.N));
    if (atomic) vec = vec && (g.kchunk % 4 == 0);
    // tile shape: narrow outputs take the 128x64 tile; problems with too few 128x128 tiles to fill the chip take 64x128
    const int64_t zdim = (int64_t)batch * (atomic ? g.ksplit : 1);
    int mi = 2, ni = 2;
    if (g.N <= 64) ni = 1;
    else if (g.M <= 64 || (g.M > 128 && cdiv(g.M, BM) * cdiv(g.N, BN) * zdim < 2 * 256)) mi = 1;
    dim3 grid((unsigned)cdiv(g.N, 64 * ni), (unsigned)cdiv(g.M, 64 * mi), (unsigned)zdim);
    dim3 block(256);
#define FQSS_GEMM(AK, BKc, V, AT)                                                                                     \
    do {                                                                                                              \
        if (mi == 2 && ni == 2) hipLaunchKernelGGL((k_gemm_f32<AK, BKc, V, AT, 2, 2>), grid, block, 0, s, g);        \
        else if (ni == 1) hipLaunchKernelGGL((k_gemm_f32<AK, BKc, V, AT, 2, 1>), grid, block, 0, s, g);              \
        else hipLaunchKernelGGL((k_gemm_f32<AK, BKc, V, AT, 1, 2>), grid, block, 0, s, g);                           \
    } while (0)
    if (!atomic) {
        if (a_kc && !b_kc) { if (vec) FQSS_GEMM(true, false, true, false); else FQSS_GEMM(true, false, false, false); }
        else if (!a_kc && !b_kc) { if (vec) FQSS_GEMM(false, false, true, false); else FQSS_GEMM(false, false, false, false); }
        else if (a_kc && b_kc) { if (vec) FQSS_GEMM(true, true, true, false); else FQSS_GEMM(true, true, false, false); }
        else { set_error("%s: unsupported operand layout", what); return FQSS_EINVAL; }
    } else {
        if (a_kc && b_kc) { if (vec) FQSS_GEMM(true, true, true, true); else FQSS_GEMM(true, true, false, true); }
        else if (a_kc && !b_kc) { if (vec) FQSS_GEMM(true, false, true, true); else FQSS_GEMM(true, false, false, true); }
        else if (!a_kc && !b_kc) { if (vec) FQSS_GEMM(false, false, true, true); else FQSS_GEMM(false, false, false, true); }
        else { set_error("%s: unsupported operand layout", what); return FQSS_EINVAL; }
    }
#undef FQSS_GEMM
    return launch_status(what);
}

}  // namespace fqss

using namespace fqss;

// split-bf16 form of the same problem (csrc/gemm_x3.hip, defined below); *used = false when its 16-B vector loads do not apply
static int try_x3(const GemmArgs& g, bool a_kc, bool b_kc, bool atomic, hipStream_t s, const char* what, bool* used, int batch = 1);

extern "C" int fqss_pwconv_fwd(const float* x, const float* w, const float* bias, float* z, int B, int Ci, int Co,
                               int M, int64_t ld_x, int64_t ld_z, fqss_stream_t stream) {
    FQSS_REQUIRE(x && w && z, "null tensor");
    FQSS_REQUIRE(B >= 0 && Ci > 0 && Co > 0 && M >= 0 && ld_x >= M && ld_z >= M, "bad shape");
    GemmArgs g{};
    g.A = w; g.B = x; g.C = z; g.bias = bias;
    g.M = Co; g.N = M; g.K = Ci;
    g.sAb = 0; g.sAi = Ci; g.sAk = 1;
    g.sBb = (int64_t)Ci * ld_x; g.sBk = ld_x; g.sBj = 1;
    g.sCb = (int64_t)Co * ld_z; g.sCi = ld_z;
    g.ksplit = 1; g.kchunk = Ci;
    return launch_gemm(g, true, false, false, B, (hipStream_t)stream, "fqss_pwconv_fwd");
}

extern "C" int fqss_pwconv_bwd_x(const float* gz, const float* w, float* gx, int B, int Ci, int Co, int M,
                                 int64_t ld_gz, int64_t ld_gx, fqss_stream_t stream) {
    FQSS_REQUIRE(gz && w && gx, "null tensor");
    FQSS_REQUIRE(B >= 0 && Ci > 0 && Co > 0 && M >= 0 && ld_gz >= M && ld_gx >= M, "bad shape");
    GemmArgs g{};
    g.A = w; g.B = gz; g.C = gx; g.bias = nullptr;
    g.M = Ci; g.N = M; g.K = Co;
    g.sAb = 0; g.sAi = 1; g.sAk = Ci;  // A = W^T: A(i=ci, k=co) = W[co*Ci + ci]
    g.sBb = (int64_t)Co * ld_gz; g.sBk = ld_gz; g.sBj = 1;
    g.sCb = (int64_t)Ci * ld_gx; g.sCi = ld_gx;
    g.ksplit = 1; g.kchunk = Co;
    bool used = false;
    int rc = try_x3(g, false, false, false, (hipStream_t)stream, "fqss_pwconv_bwd_x", &used, B);
    if (rc != FQSS_OK || used) return rc;
    return launch_gemm(g, false, false, false, B, (hipStream_t)stream, "fqss_pwconv_bwd_x");
}

extern "C" int fqss_pwconv_bwd_w(const float* gz, const float* x, float* gw, int B, int Ci, int Co, int M,
                                 int64_t ld_gz, int64_t ld_x, fqss_stream_t stream) {
    FQSS_REQUIRE(gz && x && gw, "null tensor");
    FQSS_REQUIRE(B >= 0 && Ci > 0 && Co > 0 && M >= 0 && ld_gz >= M && ld_x >= M, "bad shape");
    if (M == 0) return FQSS_OK;
    GemmArgs g{};
    g.A = gz; g.B = x; g.C = gw; g.bias = nullptr;
    g.M = Co; g.N = Ci; g.K = M;
    g.sAb = (int64_t)Co * ld_gz; g.sAi = ld_gz; g.sAk = 1;
    g.sBb = (int64_t)Ci * ld_x; g.sBk = 1; g.sBj = ld_x;  // B(k=m, j=ci) = x[ci*ld + m]
    g.sCb = 0; g.sCi = Ci;
    // split the long reduction (B*M) so that >= ~256 workgroups are in flight
    const int tiles = (int)(cdiv(Co, BM) * cdiv(Ci, BN));
    int want = (int)cdiv(512, (int64_t)tiles * (B > 0 ? B : 1));
    if (want < 1) want = 1;
    int kchunk = (int)cdiv(cdiv(M, want), 64) * 64;
    if (kchunk < 64) kchunk = 64;
    g.kchunk = kchunk;
    g.ksplit = (int)cdiv(M, kchunk);
    bool used = false;
    int rc = try_x3(g, true, true, true, (hipStream_t)stream, "fqss_pwconv_bwd_w", &used, B);
    if (rc != FQSS_OK || used) return rc;
    return launch_gemm(g, true, true, true, B, (hipStream_t)stream, "fqss_pwconv_bwd_w");
}

extern "C" int fqss_frames_wgrad(const float* a, const float* x, float* gw, int N, int C, int Ci, int M, int64_t ld_a,
                                 int64_t T, int K, int stride, fqss_stream_t stream) {
    FQSS_REQUIRE(a && x && gw, "null tensor");
    FQSS_REQUIRE(N >= 0 && C > 0 && Ci > 0 && M >= 0 && ld_a >= M && K > 0 && stride > 0, "bad shape");
    FQSS_REQUIRE((int64_t)(M - 1) * stride + K <= T || M == 0, "frames exceed the signal");
    if (M == 0) return FQSS_OK;
    const int tiles = (int)cdiv(C, BM);
    int want = (int)cdiv(256, (int64_t)tiles * (N > 0 ? N : 1));
    if (want < 1) want = 1;
    int kchunk = (int)cdiv(cdiv(M, want), 64) * 64;
    if (kchunk < 64) kchunk = 64;
    for (int ci = 0; ci < Ci; ++ci) {
        GemmArgs g{};
        g.A = a; g.B = x + (int64_t)ci * T; g.C = gw + (int64_t)ci * K; g.bias = nullptr;
        g.M = C; g.N = K; g.K = M;
        g.sAb = (int64_t)C * ld_a; g.sAi = ld_a; g.sAk = 1;
        g.sBb = (int64_t)Ci * T; g.sBk = stride; g.sBj = 1;  // B(k=m, j=k') = x[n][ci][m*stride + k']
        g.sCb = 0; g.sCi = (int64_t)Ci * K;
        g.kchunk = kchunk;
        g.ksplit = (int)cdiv(M, kchunk);
        int rc = launch_gemm(g, true, false, true, N, (hipStream_t)stream, "fqss_frames_wgrad");
        if (rc != FQSS_OK) return rc;
    }
    return FQSS_OK;
}

// split-bf16 variant (csrc/gemm_x3.hip): same problem description, 2-3x the fp32-MFMA throughput; `used` = false when its
// 16-B vector loads do not apply (the caller then runs k_gemm_f32).  FQSS_ROWGEMM_X3=0 in the environment forces the fallback.
namespace fqss {
struct GemmArgs3 {          // keep in sync with csrc/gemm_x3.hip
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    const float* bias_col;
    int M, N, K;
    int64_t sAi, sAk;
    int64_t sBk, sBj;
    int64_t sCi;
    int ksplit, kchunk;
    const void* Bq;
    const float* scale_k;
    const float* qmin_x;
    const float* qmax_x;
    int batch;
    int64_t sAb, sBb, sCb;
    int imp_taps, imp_dil, imp_pad, imp_len;
    int imp_kw, imp_rowshift;
    const unsigned short* Bp;
    int64_t ldp;
    int scalar_stores;
    float* rowsum_out;
};
int launch_gemm_x3_imp(const GemmArgs3& g, bool wgrad, hipStream_t s, const char* what);
int launch_gemm_x3(const GemmArgs3& g, bool a_kc, bool b_kc, bool atomic, hipStream_t s, const char* what, bool* used);
int launch_gemm_x3q(const GemmArgs3& g, int bq, hipStream_t s, const char* what);
}  // namespace fqss

// workgroups a row-major weight gradient aims for: every k-slice ADDS its whole Co x Ci tile with float atomics, so the slice count
// is a trade between atomic traffic (measured: half of the kernel at 125 slices for a 256 x 256 gradient) and occupancy -- 512, and
// 256 for outputs of at most four 128 x 128 tiles (tools/kprobe.py sweep: 256 x 256 coded 43 -> 32 us)
static int rowgrad_wgs(int tiles) {
    static const int n = [] { const char* e = getenv("FQSS_WGRAD_WGS"); const int v = e ? atoi(e) : 0; return v; }();
    return n > 0 ? n : (tiles <= 4 ? 256 : 512);
}

static bool x3_enabled() {
    static const bool on = [] { const char* e = getenv("FQSS_ROWGEMM_X3"); return !(e && e[0] == '0'); }();
    return on;
}

static int try_x3(const GemmArgs& g, bool a_kc, bool b_kc, bool atomic, hipStream_t s, const char* what, bool* used, int batch) {
    *used = false;
    if (!x3_enabled()) return FQSS_OK;
    if (batch > 1 && (g.sAb % 4 != 0 || g.sBb % 4 != 0)) return FQSS_OK;      // every batch's operand 16-B aligned
    GemmArgs3 h{g.A, g.B, g.C, g.bias, g.bias_col, g.M, g.N, g.K, g.sAi, g.sAk, g.sBk, g.sBj, g.sCi, g.ksplit, g.kchunk, nullptr, nullptr, nullptr, nullptr,
                batch, g.sAb, g.sBb, g.sCb, 0, 0, 0, 0, 0, 0, nullptr, 0, 0};
    return launch_gemm_x3(h, a_kc, b_kc, atomic, s, what, used);
}

// ------------------------------------------------------------------------------------------------------------------
// Row-major ("channels-last") linears of the dual-path models: x[R][Ci] (row stride ld_x) -> z[R][Co].
// Reference replaced: F.linear in LinearQ / MultiheadAttentionQ / LSTMQ's input projection (qat_layers.py:521-536,
// 889-901, 941-942, 590-591), the 1x1 Conv2dQ of DPT.output (dptnetq.py:187) and their autograd.
//   fwd    z[r][o]   = sum_i x[r][i] w[o][i] + bias[o]      A = x  (k contiguous)  B = w^T (k contiguous)
//   dgrad  gx[r][i]  = sum_o gz[r][o] w[o][i]               A = gz (k contiguous)  B = w   (j contiguous)
//   wgrad  gw[o][i] += sum_r gz[r][o] x[r][i]               A = gz^T (i contiguous) B = x  (j contiguous), split-K + atomics
// ------------------------------------------------------------------------------------------------------------------
extern "C" int fqss_rowlin_fwd(const float* x, const float* w, const float* bias, float* z, int64_t R, int Ci, int Co,
                               int64_t ld_x, int64_t ld_w, int64_t ld_z, fqss_stream_t stream) {
    FQSS_REQUIRE(x && w && z, "null tensor");
    FQSS_REQUIRE(R >= 0 && R < (1ll << 31) && Ci > 0 && Co > 0 && ld_x >= Ci && ld_w >= Ci && ld_z >= Co, "bad shape");
    GemmArgs g{};
    g.A = x; g.B = w; g.C = z; g.bias = nullptr; g.bias_col = bias;
    g.M = (int)R; g.N = Co; g.K = Ci;
    g.sAb = 0; g.sAi = ld_x; g.sAk = 1;
    g.sBb = 0; g.sBk = 1; g.sBj = ld_w;
    g.sCb = 0; g.sCi = ld_z;
    g.ksplit = 1; g.kchunk = Ci;
    bool used = false;
    int rc = try_x3(g, true, true, false, (hipStream_t)stream, "fqss_rowlin_fwd", &used);
    if (rc != FQSS_OK || used) return rc;
    return launch_gemm(g, true, true, false, 1, (hipStream_t)stream, "fqss_rowlin_fwd");
}

extern "C" int fqss_rowlin_bwd_x(const float* gz, const float* w, float* gx, int64_t R, int Ci, int Co, int64_t ld_gz,
                                 int64_t ld_w, int64_t ld_gx, fqss_stream_t stream) {
    FQSS_REQUIRE(gz && w && gx, "null tensor");
    FQSS_REQUIRE(R >= 0 && R < (1ll << 31) && Ci > 0 && Co > 0 && ld_gz >= Co && ld_w >= Ci && ld_gx >= Ci, "bad shape");
    GemmArgs g{};
    g.A = gz; g.B = w; g.C = gx; g.bias = nullptr; g.bias_col = nullptr;
    g.M = (int)R; g.N = Ci; g.K = Co;
    g.sAb = 0; g.sAi = ld_gz; g.sAk = 1;
    g.sBb = 0; g.sBk = ld_w; g.sBj = 1;
    g.sCb = 0; g.sCi = ld_gx;
    g.ksplit = 1; g.kchunk = Co;
    bool used = false;
    int rc = try_x3(g, true, false, false, (hipStream_t)stream, "fqss_rowlin_bwd_x", &used);
    if (rc != FQSS_OK || used) return rc;
    return launch_gemm(g, true, false, false, 1, (hipStream_t)stream, "fqss_rowlin_bwd_x");
}

// The student's row-major gradient GEMMs on codes (csrc/gemm_x3.hip, coded-B forms): the 8-bit operand is ONE exact bf16 plane, so the
// product count is three instead of six.
//   fqss_qrow_bwd_x: gx[r][i]  = sum_o gz[r][o] * (dw[o] * wi[o][i])        (wi int8 [Co][Ci], dense rows)
//   fqss_qrow_bwd_w: gw[o][i] += sum_r gz[r][o] * (dx * c[r][i] + min_x)    (c u8 [R][ld_xc])
extern "C" int fqss_qrow_bwd_x(const float* gz, const int8_t* wi, const float* dw, float* gx, int64_t R, int Ci, int Co, int64_t ld_gz,
                               int64_t ld_gx, fqss_stream_t stream) {
    FQSS_REQUIRE(gz && wi && dw && gx, "null tensor");
    FQSS_REQUIRE(R >= 0 && R < (1ll << 31) && Ci > 0 && Co > 0 && ld_gz >= Co && ld_gx >= Ci, "bad shape");
    FQSS_REQUIRE(Ci % 4 == 0 && Co % 4 == 0 && ld_gz % 4 == 0 && aligned16(gz) && ((uintptr_t)wi & 3) == 0 && aligned16(dw),
                 "coded dgrad: Ci, Co and the gz row stride must be multiples of 4, operands aligned");
    if (R == 0) return FQSS_OK;
    GemmArgs3 g{};
    g.A = gz; g.B = nullptr; g.C = gx; g.bias = nullptr; g.bias_col = nullptr;
    g.M = (int)R; g.N = Ci; g.K = Co;
    g.sAi = ld_gz; g.sAk = 1;
    g.sBk = Ci; g.sBj = 1;
    g.sCi = ld_gx;
    g.ksplit = 1; g.kchunk = Co;
    g.Bq = wi; g.scale_k = dw;
    return launch_gemm_x3q(g, 2, (hipStream_t)stream, "fqss_qrow_bwd_x");
}

static int qrow_bwd_w_impl(const char* who, const float* gz, const uint8_t* xc, const float* qmin_x, const float* qmax_x, float* gw, int64_t R,
                           int Ci, int Co, int64_t ld_gz, int64_t ld_xc, int64_t ld_gw, int batch, int64_t sb_gz, int64_t sb_xc, int64_t sb_gw,
                           fqss_stream_t stream, float* gbias = nullptr) {
    FQSS_REQUIRE(gz && xc && qmin_x && qmax_x && gw, "null tensor");
    FQSS_REQUIRE(R >= 0 && R < (1ll << 31) && Ci > 0 && Co > 0 && ld_gz >= Co && ld_xc >= Ci && ld_gw >= Ci && batch >= 1 && batch <= 64, "bad shape");
    FQSS_REQUIRE(Ci % 4 == 0 && Co % 4 == 0 && ld_gz % 4 == 0 && ld_xc % 4 == 0 && aligned16(gz) && ((uintptr_t)xc & 3) == 0 && sb_gz % 4 == 0 &&
                     sb_xc % 4 == 0,
                 "coded wgrad: Ci, Co, the row strides and the problem strides must be multiples of 4, operands aligned");
    if (R == 0) return FQSS_OK;
    GemmArgs3 g{};
    g.A = gz; g.B = nullptr; g.C = gw; g.bias = nullptr; g.bias_col = nullptr;
    g.M = Co; g.N = Ci; g.K = (int)R;
    g.sAi = 1; g.sAk = ld_gz;      // A(i=o, k=r) = gz[r*ld + o]
    g.sBk = ld_xc; g.sBj = 1;      // B(k=r, j=i) = c[r*ld + i]
    g.sCi = ld_gw;
    g.batch = batch; g.sAb = sb_gz; g.sBb = sb_xc; g.sCb = sb_gw;
    const int tiles = (int)(cdiv(Co, BM) * cdiv(Ci, BN)) * batch;
    int want = (int)cdiv(rowgrad_wgs(tiles), tiles);
    int kchunk = (int)cdiv(cdiv(R, want), 64) * 64;
    if (kchunk < 64) kchunk = 64;
    g.kchunk = kchunk;
    g.ksplit = (int)cdiv(R, kchunk);
    g.Bq = xc; g.qmin_x = qmin_x; g.qmax_x = qmax_x;
    g.rowsum_out = gbias;
    return launch_gemm_x3q(g, 1, (hipStream_t)stream, who);
}

extern "C" int fqss_qrow_bwd_w(const float* gz, const uint8_t* xc, const float* qmin_x, const float* qmax_x, float* gw, int64_t R, int Ci,
                               int Co, int64_t ld_gz, int64_t ld_xc, int64_t ld_gw, fqss_stream_t stream) {
    return qrow_bwd_w_impl("fqss_qrow_bwd_w", gz, xc, qmin_x, qmax_x, gw, R, Ci, Co, ld_gz, ld_xc, ld_gw, 1, 0, 0, 0, stream);
}

// fqss_qrow_bwd_w that also ADDS the bias gradient gbias[o] += sum_r gz[r][o]: the kernel keeps these row sums of its A operand anyway
// (the min_x term of the coded product), so the separate column-sum pass over gz (fqss_colsum) disappears
extern "C" int fqss_qrow_bwd_wb(const float* gz, const uint8_t* xc, const float* qmin_x, const float* qmax_x, float* gw, float* gbias, int64_t R,
                                int Ci, int Co, int64_t ld_gz, int64_t ld_xc, int64_t ld_gw, fqss_stream_t stream) {
    FQSS_REQUIRE(gbias, "null bias gradient");
    return qrow_bwd_w_impl("fqss_qrow_bwd_wb", gz, xc, qmin_x, qmax_x, gw, R, Ci, Co, ld_gz, ld_xc, ld_gw, 1, 0, 0, 0, stream, gbias);
}

// `batch` coded weight gradients of one shape and one input range in ONE launch (problem p: gz + p sb_gz, xc + p sb_xc bytes, gw + p sb_gw):
// the two directions' W_ih gradients of a bidirectional LSTM -- two column blocks of dG against the same input codes
extern "C" int fqss_qrow_bwd_w_batched(const float* gz, const uint8_t* xc, const float* qmin_x, const float* qmax_x, float* gw, int64_t R, int Ci,
                                       int Co, int64_t ld_gz, int64_t ld_xc, int64_t ld_gw, int batch, int64_t sb_gz, int64_t sb_xc,
                                       int64_t sb_gw, fqss_stream_t stream) {
    return qrow_bwd_w_impl("fqss_qrow_bwd_w_batched", gz, xc, qmin_x, qmax_x, gw, R, Ci, Co, ld_gz, ld_xc, ld_gw, batch, sb_gz, sb_xc, sb_gw,
                           stream);
}

static int rowlin_bwd_w_impl(const char* who, const float* gz, const float* x, float* gw, int64_t R, int Ci, int Co, int64_t ld_gz,
                             int64_t ld_x, int64_t ld_gw, int batch, int64_t sb_gz, int64_t sb_x, int64_t sb_gw, fqss_stream_t stream) {
    FQSS_REQUIRE(gz && x && gw, "null tensor");
    FQSS_REQUIRE(R >= 0 && R < (1ll << 31) && Ci > 0 && Co > 0 && ld_gz >= Co && ld_x >= Ci && ld_gw >= Ci && batch >= 1, "bad shape");
    if (R == 0) return FQSS_OK;
    GemmArgs g{};
    g.A = gz; g.B = x; g.C = gw; g.bias = nullptr; g.bias_col = nullptr;
    g.M = Co; g.N = Ci; g.K = (int)R;
    g.sAb = sb_gz; g.sAi = 1; g.sAk = ld_gz;   // A(i=o, k=r) = gz[r*ld + o]
    g.sBb = sb_x; g.sBk = ld_x; g.sBj = 1;     // B(k=r, j=i) = x[r*ld + i]
    g.sCb = sb_gw; g.sCi = ld_gw;
    const int tiles = (int)(cdiv(Co, BM) * cdiv(Ci, BN)) * batch;
    int want = (int)cdiv(rowgrad_wgs(tiles), tiles);
    int kchunk = (int)cdiv(cdiv(R, want), 64) * 64;
    if (kchunk < 64) kchunk = 64;
    g.kchunk = kchunk;
    g.ksplit = (int)cdiv(R, kchunk);
    bool used = false;
    int rc = try_x3(g, false, false, true, (hipStream_t)stream, who, &used, batch);
    if (rc != FQSS_OK || used) return rc;
    return launch_gemm(g, false, false, true, batch, (hipStream_t)stream, who);
}

extern "C" int fqss_rowlin_bwd_w(const float* gz, const float* x, float* gw, int64_t R, int Ci, int Co, int64_t ld_gz,
                                 int64_t ld_x, int64_t ld_gw, fqss_stream_t stream) {
    return rowlin_bwd_w_impl("fqss_rowlin_bwd_w", gz, x, gw, R, Ci, Co, ld_gz, ld_x, ld_gw, 1, 0, 0, 0, stream);
}

// `batch` weight gradients of the same shape in ONE launch: problem p reads gz + p sb_gz, x + p sb_x and adds into gw + p sb_gw (element
// strides, any sign) -- the two directions' W_hh gradients of a bidirectional LSTM (dG[1:, :, :4H] with h[:-1, :, :H] and dG[:-1, :, 4H:]
// with h[1:, :, H:]: the same tensors, shifted)
extern "C" int fqss_rowlin_bwd_w_batched(const float* gz, const float* x, float* gw, int64_t R, int Ci, int Co, int64_t ld_gz, int64_t ld_x,
                                         int64_t ld_gw, int batch, int64_t sb_gz, int64_t sb_x, int64_t sb_gw, fqss_stream_t stream) {
    FQSS_REQUIRE(batch >= 1 && batch <= 64, "bad batch");
    return rowlin_bwd_w_impl("fqss_rowlin_bwd_w_batched", gz, x, gw, R, Ci, Co, ld_gz, ld_x, ld_gw, batch, sb_gz, sb_x, sb_gw, stream);
}

// ------------------------------------------------------------------------------------------------------------------
// Stride-1 1-D convolution (any kernel width / dilation / zero padding, groups = 1) WITHOUT the frame image: the split-bf16 GEMM of
// csrc/gemm_x3.hip reads its B operand straight from the signal, one shifted row per (channel, tap).  Replaces, for the k = 3
// convolutions of HTDemucs (DConv and the rewrite convs, hdemucsq.py:72-162, demucsq.py:110-182): fqss_frames_gather + the pointwise GEMM
// in the forward (the gather wrote 3x the activation and the GEMM read it back), GEMM + fqss_frames_ola in the data gradient (the same
// entry with the taps flipped and the weight transposed by the caller), GEMM over the saved frames in the weight gradient.
//   x [B][Ci][M] (row stride ld_x), w [Co][Ci * taps] (row stride ld_w, a multiple of 4), z [B][Co][Mo],  Mo = M + 2 pad - dil (taps - 1)
//   z[b][co][m] = bias[co] + sum_{ci, t} w[co][ci * taps + t] x[b][ci][m + t dil - pad]
// ------------------------------------------------------------------------------------------------------------------
extern "C" int fqss_conv1d_s1_fwd(const float* x, const float* w, const float* bias, float* z, int B, int Ci, int Co, int M, int Mo, int taps,
                                  int dil, int pad, int64_t ld_x, int64_t ld_w, int64_t ld_z, fqss_stream_t stream) {
    if (B == 0 || Mo <= 0) return FQSS_OK;
    FQSS_REQUIRE(x && w && z, "null tensor");
    FQSS_REQUIRE(B > 0 && Ci > 0 && Co > 0 && M > 0 && taps > 0 && dil > 0 && pad >= 0 && ld_x >= M && ld_z >= Mo, "bad shape");
    FQSS_REQUIRE(Mo == M + 2 * pad - dil * (taps - 1), "output length does not match the geometry");
    FQSS_REQUIRE(ld_w % 4 == 0 && ld_w >= rup4((int64_t)Ci * taps) && aligned16(w), "implicit conv: weight rows 16-B aligned, padded to a multiple of 4");
    GemmArgs3 g{};
    g.A = w; g.B = x; g.C = z; g.bias = bias; g.bias_col = nullptr;
    g.M = Co; g.N = Mo; g.K = Ci * taps;
    g.sAi = ld_w; g.sAk = 1;
    g.sBk = ld_x; g.sBj = 1;
    g.sCi = ld_z;
    g.ksplit = 1; g.kchunk = g.K;
    g.batch = B; g.sAb = 0; g.sBb = (int64_t)Ci * ld_x; g.sCb = (int64_t)Co * ld_z;
    g.imp_taps = taps; g.imp_dil = dil; g.imp_pad = pad; g.imp_len = M;
    return launch_gemm_x3_imp(g, false, (hipStream_t)stream, "fqss_conv1d_s1_fwd");
}

// gw[co][ci * taps + t] += sum_{b, m} gz[b][co][m] x[b][ci][m + t dil - pad]   (gw: caller-zeroed accumulator, [Co][Ci * taps])
extern "C" int fqss_conv1d_s1_bwd_w(const float* gz, const float* x, float* gw, int B, int Ci, int Co, int M, int Mo, int taps, int dil,
                                    int pad, int64_t ld_gz, int64_t ld_x, fqss_stream_t stream) {
    if (B == 0 || Mo <= 0) return FQSS_OK;
    FQSS_REQUIRE(gz && x && gw, "null tensor");
    FQSS_REQUIRE(B > 0 && Ci > 0 && Co > 0 && M > 0 && taps > 0 && dil > 0 && pad >= 0 && ld_x >= M && ld_gz >= Mo, "bad shape");
    FQSS_REQUIRE(Mo == M + 2 * pad - dil * (taps - 1), "output length does not match the geometry");
    FQSS_REQUIRE(aligned16(gz) && ld_gz % 4 == 0 && ld_gz >= rup4(Mo), "implicit conv wgrad: 16-B aligned gradient rows");
    GemmArgs3 g{};
    g.A = gz; g.B = x; g.C = gw; g.bias = nullptr; g.bias_col = nullptr;
    g.M = Co; g.N = Ci * taps; g.K = Mo;
    g.sAi = ld_gz; g.sAk = 1;
    g.sBk = 1; g.sBj = ld_x;                  // (B rows = (channel, tap): the implicit loader steps channels by sBj)
    g.sCi = (int64_t)Ci * taps;
    const int tiles = (int)(cdiv(Co, BM) * cdiv((int64_t)Ci * taps, BN));
    int want = (int)cdiv(512, (int64_t)tiles * B);
    if (want < 1) want = 1;
    int kchunk = (int)cdiv(cdiv(Mo, want), 64) * 64;
    if (kchunk < 64) kchunk = 64;
    g.kchunk = kchunk;
    g.ksplit = (int)cdiv(Mo, kchunk);
    g.batch = B; g.sAb = (int64_t)Co * ld_gz; g.sBb = (int64_t)Ci * ld_x; g.sCb = 0;
    g.imp_taps = taps; g.imp_dil = dil; g.imp_pad = pad; g.imp_len = M;
    return launch_gemm_x3_imp(g, true, (hipStream_t)stream, "fqss_conv1d_s1_bwd_w");
}

// ... and of a stride-1 2-D convolution on halo-packed operands (fqss_halo_pack, csrc/conv_frames.hip): gzp [B][Co][plane_g] the packed
// gradient of the output, xp [B][Ci][plane_x] the packed input; gw[co][ci * taps + t] += sum_{b, m} gzp[b][co][m] *
// xp[b][ci][m + (t / kw) row_step + (t % kw) col_step - off] (zero outside the plane).  Same kernel, tap shifts in two dimensions.
extern "C" int fqss_conv2_bwd_w(const float* gzp, const float* xp, float* gw, int B, int Ci, int Co, int taps, int kw, int row_step, int col_step,
                                int off, int64_t plane_g, int64_t plane_x, fqss_stream_t stream) {
    if (B == 0) return FQSS_OK;
    FQSS_REQUIRE(gzp && xp && gw, "null tensor");
    FQSS_REQUIRE(B > 0 && Ci > 0 && Co > 0 && taps >= 1 && kw >= 1 && taps % kw == 0 && plane_g > 0 && plane_x > 0 && plane_x < (1ll << 30), "bad shape");
    FQSS_REQUIRE(aligned16(gzp) && plane_g % 4 == 0, "implicit conv wgrad: 16-B aligned gradient planes");
    GemmArgs3 g{};
    g.A = gzp; g.B = xp; g.C = gw; g.bias = nullptr; g.bias_col = nullptr;
    g.M = Co; g.N = Ci * taps; g.K = (int)plane_g;
    g.sAi = plane_g; g.sAk = 1;
    g.sBk = 1; g.sBj = plane_x;
    g.sCi = (int64_t)Ci * taps;
    const int tiles = (int)(cdiv(Co, BM) * cdiv((int64_t)Ci * taps, BN));
    int want = (int)cdiv(512, (int64_t)tiles * B);
    if (want < 1) want = 1;
    int kchunk = (int)cdiv(cdiv(plane_g, want), 64) * 64;
    if (kchunk < 64) kchunk = 64;
    g.kchunk = kchunk;
    g.ksplit = (int)cdiv(plane_g, kchunk);
    g.batch = B; g.sAb = (int64_t)Co * plane_g; g.sBb = (int64_t)Ci * plane_x; g.sCb = 0;
    g.imp_taps = taps; g.imp_dil = col_step; g.imp_pad = off; g.imp_len = (int)plane_x; g.imp_kw = kw; g.imp_rowshift = row_step;
    return launch_gemm_x3_imp(g, true, (hipStream_t)stream, "fqss_conv2_bwd_w");
}

// fqss_rowlin_fwd with the weight given as its three exact bf16 planes [3][Co][Ci] (fqss_split3_planes): the weight tile is copied into
// LDS instead of being split by every workgroup that touches it -- for weights that do not change between launches (the frozen float
// teacher: its row GEMMs are bound by the vector-ALU issue of that split, docs/history/DESIGN_rounds_1-5.md 7e (4)).  Same result bits as fqss_rowlin_fwd.
extern "C" int fqss_rowlin_fwd_w3(const float* x, const uint16_t* w3, const float* bias, float* z, int64_t R, int Ci, int Co, int64_t ld_x,
                                  int64_t ld_z, fqss_stream_t stream) {
    FQSS_REQUIRE(x && w3 && z, "null tensor");
    FQSS_REQUIRE(R >= 0 && R < (1ll << 31) && Ci > 0 && Co > 0 && ld_x >= Ci && ld_z >= Co, "bad shape");
    FQSS_REQUIRE(Ci % 32 == 0 && aligned16(w3) && aligned16(x) && ld_x % 4 == 0, "pre-split weight: Ci a multiple of 32, 16-B aligned rows");
    if (R == 0) return FQSS_OK;
    GemmArgs3 g{};
    g.A = x; g.B = nullptr; g.C = z; g.bias = nullptr; g.bias_col = bias;
    g.M = (int)R; g.N = Co; g.K = Ci;
    g.sAi = ld_x; g.sAk = 1;
    g.sBk = 1; g.sBj = Ci;
    g.sCi = ld_z;
    g.ksplit = 1; g.kchunk = Ci;
    g.batch = 1;
    g.Bp = w3; g.ldp = Ci;
    g.B = x;            // (alignment checks of the shared launcher: B is not read)
    bool used = false;
    int rc = launch_gemm_x3(g, true, true, false, (hipStream_t)stream, "fqss_rowlin_fwd_w3", &used);
    if (rc != FQSS_OK) return rc;
    if (!used) { set_error("fqss_rowlin_fwd_w3: operands not eligible for the split-bf16 GEMM"); return FQSS_EINVAL; }
    return FQSS_OK;
}
