// gemm_x3.hip -- fp32 x fp32 GEMM on the bf16 matrix cores: both operands are split EXACTLY into three bf16 pieces
// (x = h + m + l, 8 mantissa bits each) when their tile is stored to LDS, and the six leading partial products
// (h.h, h.m, m.h, m.m, h.l, l.h: everything above 2^-24 of the full product) are accumulated in fp32, smallest first.
// v_mfma_f32_32x32x16_bf16 runs 16x the fp32 MFMA rate, so six of them cost 3/8 of the fp32 instruction time; the error is of
// the order of one fp32 rounding per product (the float teacher of ConvTasNet uses the same arithmetic, csrc/teacher.hip).
//
// Same operand addressing as k_gemm_f32 (csrc/gemm.hip): C[i][j] = sum_k A(i,k) B(k,j) with either dimension of each operand
// contiguous, optional row / column bias, optional split-K with atomics.  It serves the row-major linears of the dual-path
// models (fqss_rowlin_fwd / bwd_x / bwd_w) whenever the 16-B vector path applies; k_gemm_f32 remains the general fallback.
//
// LDS image: planes[3][rows][BK = 32 bf16 + pad] per operand (k contiguous), so an MFMA operand is ONE 16-B read per lane.
// Operands whose k dimension is strided in memory are transposed on the way in: a thread owns a 4 (rows) x 4 (k) block -- four
// coalesced float4 loads along the contiguous dimension, then one 8-B LDS write per row and plane.
// Measured dead end: k-tiles of 16 with two LDS buffers and one barrier per tile (split + store of tile kt+1 under the MFMAs of
// tile kt) -- 5-12 % SLOWER than this single-buffered k-tile of 32 at the dual-path shapes (60 -> 66 us for 8500 x 256 x 1024).
#include "fqss_dev.h"

namespace fqss {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct GemmArgs3 {
    const float* A;
    const float* B;
    float* C;
    const float* bias;      // [M] or null
    const float* bias_col;  // [N] or null
    int M, N, K;
    int64_t sAi, sAk;       // one of them is 1
    int64_t sBk, sBj;
    int64_t sCi;
    int ksplit, kchunk;
};

constexpr int XBK = 32, XLDK = 40;   // 40 shorts = 80 B row stride (as csrc/teacher.hip: conflict-light 16-B reads)

__device__ __forceinline__ unsigned short x_bf(float f) { return (unsigned short)(__float_as_uint(f) >> 16); }
__device__ __forceinline__ float x_tr(float f) { return __uint_as_float(__float_as_uint(f) & 0xFFFF0000u); }
__device__ __forceinline__ void x_split3(float g, unsigned short& b1, unsigned short& b2, unsigned short& b3) {
    const float h1 = x_tr(g), r1 = g - h1, h2 = x_tr(r1), r2 = r1 - h2;   // exact: each piece has <= 8 significant bits
    b1 = x_bf(h1);
    b2 = x_bf(h2);
    b3 = x_bf(r2);
}

// ROWS = rows of this operand's tile (64 or 128); KC = the k dimension is the contiguous one in memory
// sr = stride between rows, sk = stride along k; r0 / nrows: first row of the tile and the operand's row count
// Loads are UNCONDITIONAL (addresses clamped into the operand, validity applied when the tile is stored): a branch around a
// global load makes the compiler wait for the whole load ring (s_waitcnt vmcnt(0)) before the MFMA section instead of after
// it, which exposes one HBM round trip per k-tile (measured: the first version of this kernel ran at the speed of k_gemm_f32).
template <int ROWS, bool KC>
struct TileIO {
    float4 v[4];
    int k0_, kend_, r0_, nrows_;

    __device__ __forceinline__ void load(const float* __restrict__ base, int64_t sr, int64_t sk, int r0, int nrows, int k0, int kend, int K) {
        const int tid = threadIdx.x;
        k0_ = k0; kend_ = kend; r0_ = r0; nrows_ = nrows;
        if constexpr (KC) {
            const int kmax = ((K + 3) & ~3) - 4;          // last float4 of a (padded) row
#pragma unroll
            for (int p = 0; p < ROWS / 32; ++p) {
                const int f = tid + 256 * p, r = f >> 3, k = (f & 7) * 4;
                const int rc = min(r0 + r, nrows - 1), kc = min(k0 + k, kmax);
                v[p] = *reinterpret_cast<const float4*>(base + (int64_t)rc * sr + kc);
            }
        } else {
            // 4 (rows) x 4 (k) block per thread: rb = block along the rows (contiguous in memory), kb = block along k
            const int rb = tid % (ROWS / 4), kb = min(tid / (ROWS / 4), XBK / 4 - 1);
            const int rmax = ((nrows + 3) & ~3) - 4;
            const int rc = min(r0 + rb * 4, rmax);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kc = min(k0 + kb * 4 + e, K - 1);
                v[e] = *reinterpret_cast<const float4*>(base + (int64_t)kc * sk + rc);
            }
        }
    }

    __device__ __forceinline__ void store(unsigned short (*pl)[128][XLDK]) const {
        const int tid = threadIdx.x;
        if constexpr (KC) {
#pragma unroll
            for (int p = 0; p < ROWS / 32; ++p) {
                const int f = tid + 256 * p, r = f >> 3, k = (f & 7) * 4;
                const bool rv = r0_ + r < nrows_;
                const float e[4] = {v[p].x, v[p].y, v[p].z, v[p].w};
                unsigned short h[3][4];
#pragma unroll
                for (int q = 0; q < 4; ++q) x_split3((rv && k0_ + k + q < kend_) ? e[q] : 0.f, h[0][q], h[1][q], h[2][q]);
#pragma unroll
                for (int s = 0; s < 3; ++s)
                    *reinterpret_cast<uint2*>(&pl[s][r][k]) = make_uint2((unsigned)h[s][0] | ((unsigned)h[s][1] << 16), (unsigned)h[s][2] | ((unsigned)h[s][3] << 16));
            }
        } else {
            if (tid / (ROWS / 4) >= XBK / 4) return;      // 64-row tiles: only half the threads hold a block
            const int rb = tid % (ROWS / 4), kb = tid / (ROWS / 4);
            const float e[4][4] = {{v[0].x, v[0].y, v[0].z, v[0].w}, {v[1].x, v[1].y, v[1].z, v[1].w},
                                   {v[2].x, v[2].y, v[2].z, v[2].w}, {v[3].x, v[3].y, v[3].z, v[3].w}};   // e[k][row]
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const bool rv = r0_ + rb * 4 + rr < nrows_;
                unsigned short h[3][4];
#pragma unroll
                for (int q = 0; q < 4; ++q) x_split3((rv && k0_ + kb * 4 + q < kend_) ? e[q][rr] : 0.f, h[0][q], h[1][q], h[2][q]);
#pragma unroll
                for (int s = 0; s < 3; ++s)
                    *reinterpret_cast<uint2*>(&pl[s][rb * 4 + rr][kb * 4]) =
                        make_uint2((unsigned)h[s][0] | ((unsigned)h[s][1] << 16), (unsigned)h[s][2] | ((unsigned)h[s][3] << 16));
            }
        }
    }
};

template <bool A_KC, bool B_KC, bool ATOMIC, int MI, int NI>
__global__ __launch_bounds__(256) void k_gemm_x3(GemmArgs3 g) {
    constexpr int BMt = 64 * MI, BNt = 64 * NI;
    __shared__ __attribute__((aligned(16))) unsigned short As[3][128][XLDK];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[3][128][XLDK];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ks = ATOMIC ? blockIdx.z : 0;
    const int kbeg = ATOMIC ? ks * g.kchunk : 0;
    const int kend = ATOMIC ? min(g.K, kbeg + g.kchunk) : g.K;
    const int i0 = blockIdx.y * BMt, j0 = blockIdx.x * BNt;

    TileIO<BMt, A_KC> ta;
    TileIO<BNt, B_KC> tb;
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int nkt = (kend - kbeg + XBK - 1) / XBK;
    if (nkt > 0) {
        ta.load(g.A, g.sAi, g.sAk, i0, g.M, kbeg, kend, g.K);
        tb.load(g.B, g.sBj, g.sBk, j0, g.N, kbeg, kend, g.K);
        ta.store(As);
        tb.store(Bs);
    }
    __syncthreads();
    const int lr = lane & 31, lh = lane >> 5;
    for (int kt = 0; kt < nkt; ++kt) {
        // global loads of the next tile fly under the MFMAs (issued unconditionally: the last iteration re-reads a clamped tile)
        ta.load(g.A, g.sAi, g.sAk, i0, g.M, kbeg + (kt + 1) * XBK, kend, g.K);
        tb.load(g.B, g.sBj, g.sBk, j0, g.N, kbeg + (kt + 1) * XBK, kend, g.K);
#pragma unroll
        for (int kstep = 0; kstep < XBK / 16; ++kstep) {
            bf16x8 af[3][MI], bfr[3][NI];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    af[p][mi] = *reinterpret_cast<const bf16x8*>(&As[p][wm * (32 * MI) + mi * 32 + lr][kstep * 16 + 8 * lh]);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    bfr[p][ni] = *reinterpret_cast<const bf16x8*>(&Bs[p][wn * (32 * NI) + ni * 32 + lr][kstep * 16 + 8 * lh]);
            }
            // smallest partial products first: l.h, h.l, m.m, m.h, h.m, h.h
            constexpr int IA[6] = {2, 0, 1, 1, 0, 0}, IB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int sp = 0; sp < 6; ++sp)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[IA[sp]][mi], bfr[IB[sp]][ni], acc[mi][ni], 0, 0, 0);
        }
        __syncthreads();
        if (kt + 1 < nkt) {
            ta.store(As);
            tb.store(Bs);
            __syncthreads();
        }
    }

    // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int col = j0 + wn * (32 * NI) + ni * 32 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * (32 * MI) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < g.M && col < g.N) {
                    float v = acc[mi][ni][r];
                    if (g.bias != nullptr && (!ATOMIC || kbeg == 0)) v = v + g.bias[row];
                    if (g.bias_col != nullptr && (!ATOMIC || kbeg == 0)) v = v + g.bias_col[col];
                    float* dst = g.C + (int64_t)row * g.sCi + col;
                    if constexpr (ATOMIC) atomicAdd(dst, v); else *dst = v;
                }
            }
        }
}

static inline int64_t rup4x(int64_t v) { return (v + 3) & ~(int64_t)3; }

// true when the 16-B vector loads of k_gemm_x3 are legal for this problem
static bool x3_ok(const GemmArgs3& g, bool a_kc, bool b_kc, bool atomic) {
    bool ok = aligned16(g.A) && aligned16(g.B);
    if (a_kc) ok = ok && g.sAk == 1 && g.sAi % 4 == 0 && g.sAi >= rup4x(g.K);
    else      ok = ok && g.sAi == 1 && g.sAk % 4 == 0 && g.sAk >= rup4x(g.M);
    if (b_kc) ok = ok && g.sBk == 1 && g.sBj % 4 == 0 && g.sBj >= rup4x(g.K);
    else      ok = ok && g.sBj == 1 && g.sBk % 4 == 0 && g.sBk >= rup4x(g.N);
    if (atomic) ok = ok && g.kchunk % XBK == 0;
    return ok;
}

int launch_gemm_x3(const GemmArgs3& g, bool a_kc, bool b_kc, bool atomic, hipStream_t s, const char* what, bool* used) {
    *used = false;
    if (g.M <= 0 || g.N <= 0) return FQSS_OK;
    if (!x3_ok(g, a_kc, b_kc, atomic)) return FQSS_OK;     // caller falls back to k_gemm_f32
    const int64_t zdim = atomic ? g.ksplit : 1;
    int mi = 2, ni = 2;
    if (g.N <= 64) ni = 1;
    else if (g.M <= 64 || (g.M > 128 && cdiv(g.M, 128) * cdiv(g.N, 128) * zdim < 2 * 256)) mi = 1;
    dim3 grid((unsigned)cdiv(g.N, 64 * ni), (unsigned)cdiv(g.M, 64 * mi), (unsigned)zdim), block(256);
#define FQSS_X3(AK, BKc, AT)                                                                                  \
    do {                                                                                                      \
        if (mi == 2 && ni == 2) hipLaunchKernelGGL((k_gemm_x3<AK, BKc, AT, 2, 2>), grid, block, 0, s, g);    \
        else if (ni == 1) hipLaunchKernelGGL((k_gemm_x3<AK, BKc, AT, 2, 1>), grid, block, 0, s, g);          \
        else hipLaunchKernelGGL((k_gemm_x3<AK, BKc, AT, 1, 2>), grid, block, 0, s, g);                       \
    } while (0)
    if (!atomic && a_kc && b_kc) FQSS_X3(true, true, false);          // fwd:   x [R][Ci], w [Co][Ci]
    else if (!atomic && a_kc && !b_kc) FQSS_X3(true, false, false);   // dgrad: gz [R][Co], w [Co][Ci] (j contiguous)
    else if (atomic && !a_kc && !b_kc) FQSS_X3(false, false, true);   // wgrad: gz^T, x (both row-index contiguous)
    else return FQSS_OK;
#undef FQSS_X3
    *used = true;
    return launch_status(what);
}

}  // namespace fqss
