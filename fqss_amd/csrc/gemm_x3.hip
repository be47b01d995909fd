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
// LDS image: planes[3][rows][BK = 32 bf16] per operand (k contiguous, the 16-B chunks of a row XOR-swizzled instead of padded: see
// XLDK / xsw below), so an MFMA operand is ONE 16-B read per lane; an operand that arrives as 8-bit codes keeps one plane.
// Operands whose k dimension is strided in memory are transposed on the way in: a thread owns a 4 (rows) x 4 (k) block -- four
// coalesced float4 loads along the contiguous dimension, then one 8-B LDS write per row and plane.
// Measured dead end: k-tiles of 16 with two LDS buffers and one barrier per tile (split + store of tile kt+1 under the MFMAs of
// tile kt) -- 5-12 % SLOWER than this single-buffered k-tile of 32 at the dual-path shapes (60 -> 66 us for 8500 x 256 x 1024).
#include <stdlib.h>
#include <type_traits>

#define FQSS_USES_GRAD_ADD   // the fp32 gradient atomics of this file go through grad_add (fqss_dev.h: FQSS_DETERMINISTIC=1)
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
    // coded B operand (BQ != 0, see k_gemm_x3): 8-bit codes, ONE exact bf16 plane instead of three
    const void* Bq;          // BQ = 1: u8 activation codes x [K = R][N = Ci];  BQ = 2: int8 weight codes [K = Co][N = Ci]
    const float* scale_k;    // BQ = 2: delta_w[k], multiplied into A(i, k) before the split
    const float* qmin_x;     // BQ = 1: range of the activation quantizer (device scalars): C = dx * sum A c + min_x * sum_k A
    const float* qmax_x;
    // batched problems (the channel-first pointwise convs of csrc/gemm.hip): blockIdx.z = batch * ksplit + k-slice; a split-K
    // launch adds every batch into the same C (sCb = 0: the weight gradient)
    int batch;
    int64_t sAb, sBb, sCb;
    // implicit B operand of a stride-1 1-D convolution (IMP kernels): the reduction / column index r of B stands for (channel, tap) =
    // (r / taps, r % taps) and B's element is x[channel][pos + tap * dil - pad], zero outside [0, len) -- the frames of
    // fqss_frames_gather are never written
    int imp_taps, imp_dil, imp_pad, imp_len;
    // ... and of a stride-1 2-D convolution on a halo-packed signal (round 6, fqss_halo_pack: planes of (H + 2 ph) rows of Wp floats,
    // zero halo): tap t = (t / imp_kw, t % imp_kw) reads pos + (t / imp_kw) * imp_rowshift + (t % imp_kw) * dil - pad of the flat plane.
    // imp_kw = 0: one row of taps (the 1-D form above)
    int imp_kw, imp_rowshift;
    // B operand already split (BPL kernels): three bf16 planes [3][N][ldp] of a [N][K] k-contiguous matrix (fqss_split3_planes of a
    // weight that does not change between launches: the frozen teacher's linears) -- the tile is copied, not split
    const unsigned short* Bp;
    int64_t ldp;
    int scalar_stores;       // 1: the lane-per-column epilogue (FQSS_X3_STAGED=0, A/B measurements)
    float* rowsum_out;       // BQ = 1, optional: sum_k A(i, k) is ADDED here ([M]: the bias gradient of the linear whose weight gradient this is)
};

#ifndef FQSS_X3_PF
#define FQSS_X3_PF 1     // operand tiles in flight per workgroup (register images)
#endif
// LDS rows are 32 bf16 = 64 B with NO padding; the four 16-B chunks of a row are stored XOR-swizzled by (row >> 2) & 3, so that the
// sixteen rows a quarter-wave reads with one ds_read_b128 -- same logical chunk -- fall on sixteen different 16-B bank groups (rows r and
// r + 4 are 256 B apart: the swizzle separates them; rows r .. r + 3 are 64 B apart).  The padded layout of rounds 1-3 (80-B rows) cost
// 61 KB for a 128 x 128 float tile pair = two workgroups per CU; 49 KB lets a third one in.
constexpr int XBK = 32, XLDK = 32;
__device__ __forceinline__ int xsw(int row, int k) { return (((k >> 3) ^ (row >> 2)) & 3) * 8 + (k & 7); }

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
    float4 sv_;                          // KC: scale of this thread's 4 k columns (coded-B dgrad), else unused
    float rs_[4] = {0.f, 0.f, 0.f, 0.f};  // !KC: running sums over k of this thread's 4 rows (coded-B wgrad)
    const float* sc_ = nullptr;
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
            if (sc_ != nullptr) sv_ = *reinterpret_cast<const float4*>(sc_ + min(k0 + (tid & 7) * 4, kmax));
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

    __device__ __forceinline__ void store(unsigned short (*pl)[ROWS][XLDK]) {
        const int tid = threadIdx.x;
        if constexpr (KC) {
#pragma unroll
            for (int p = 0; p < ROWS / 32; ++p) {
                const int f = tid + 256 * p, r = f >> 3, k = (f & 7) * 4;
                const bool rv = r0_ + r < nrows_;
                float e[4] = {v[p].x, v[p].y, v[p].z, v[p].w};
                if (sc_ != nullptr) { e[0] *= sv_.x; e[1] *= sv_.y; e[2] *= sv_.z; e[3] *= sv_.w; }
                unsigned short h[3][4];
#pragma unroll
                for (int q = 0; q < 4; ++q) x_split3((rv && k0_ + k + q < kend_) ? e[q] : 0.f, h[0][q], h[1][q], h[2][q]);
#pragma unroll
                for (int s = 0; s < 3; ++s)
                    *reinterpret_cast<uint2*>(&pl[s][r][xsw(r, k)]) = make_uint2((unsigned)h[s][0] | ((unsigned)h[s][1] << 16), (unsigned)h[s][2] | ((unsigned)h[s][3] << 16));
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
                for (int q = 0; q < 4; ++q) {
                    const float t = (rv && k0_ + kb * 4 + q < kend_) ? e[q][rr] : 0.f;
                    rs_[rr] += t;
                    x_split3(t, h[0][q], h[1][q], h[2][q]);
                }
#pragma unroll
                for (int s = 0; s < 3; ++s)
                    *reinterpret_cast<uint2*>(&pl[s][rb * 4 + rr][xsw(rb * 4 + rr, kb * 4)]) =
                        make_uint2((unsigned)h[s][0] | ((unsigned)h[s][1] << 16), (unsigned)h[s][2] | ((unsigned)h[s][3] << 16));
            }
        }
    }
};

// Implicit-convolution form of TileIO (see GemmArgs3::imp_*): the same register image, filled by 4-B loads from shifted, clamped
// positions (a tap's shift breaks the 16-B alignment of the row) with the zero padding applied on the way in.
//   KC = false (forward / data gradient: B(k, j) = x[k / taps][j + (k % taps) dil - pad]):   k <-> (channel, tap), rows j = positions
//   KC = true  (weight gradient:         B(k, j) = x[j / taps][k + (j % taps) dil - pad]):   rows j <-> (channel, tap), k = positions
// B tile from pre-split planes (see GemmArgs3::Bp): a thread copies 16-B chunks (8 k of one row and plane) global -> registers -> LDS;
// K % 32 == 0, rows clamped on load and zeroed on store
template <int ROWS>
struct TileIOP {
    static constexpr int NCH = 3 * ROWS * (XBK / 8) / 256;     // chunks per thread: 6 (128 rows) or 3 (64)
    uint4 v[NCH];
    int r0_, nrows_;     // (a tile past K: nrows_ = 0, stored as zeros)
    __device__ __forceinline__ void load(const unsigned short* __restrict__ planes, int64_t plane_stride, int64_t ldp, int r0, int nrows, int k0, int K) {
        r0_ = r0; nrows_ = k0 < K ? nrows : 0;
        const int kb = min(k0, K - XBK);
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const int f = threadIdx.x + 256 * u, p = f / (ROWS * 4), rem = f % (ROWS * 4), r = rem >> 2, c = rem & 3;
            v[u] = *reinterpret_cast<const uint4*>(planes + p * plane_stride + (int64_t)min(r0 + r, nrows - 1) * ldp + kb + 8 * c);
        }
    }
    __device__ __forceinline__ void store(unsigned short (*pl)[ROWS][XLDK]) const {
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const int f = threadIdx.x + 256 * u, p = f / (ROWS * 4), rem = f % (ROWS * 4), r = rem >> 2, c = rem & 3;
            *reinterpret_cast<uint4*>(&pl[p][r][xsw(r, 8 * c)]) = (r0_ + r < nrows_) ? v[u] : make_uint4(0u, 0u, 0u, 0u);
        }
    }
};

struct __attribute__((packed, aligned(4))) F4U {      // four floats at a 4-B aligned address: the compiler picks the widest legal load
    float x, y, z, w;
};
// four consecutive positions col0 .. col0+3 of a signal row, zero outside [0, len): one (unaligned) vector load in the interior, clamped
// scalar loads on the two edges of the row
__device__ __forceinline__ float4 imp_load4(const float* __restrict__ row, int col0, int len) {
    if (col0 >= 0 && col0 + 3 < len) {
        const F4U t = *reinterpret_cast<const F4U*>(row + col0);
        return make_float4(t.x, t.y, t.z, t.w);
    }
    float e[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int col = col0 + q;
        const float t = row[min(max(col, 0), len - 1)];
        e[q] = (col >= 0 && col < len) ? t : 0.f;
    }
    return make_float4(e[0], e[1], e[2], e[3]);
}

template <int ROWS, bool KC>
struct TileIOI : TileIO<ROWS, KC> {
    int taps_ = 1, dil_ = 1, pad_ = 0, len_ = 0, kw_ = 0, rowshift_ = 0;
    __device__ __forceinline__ int shift_of(int t) const {
        if (kw_ <= 0) return t * dil_ - pad_;
        const int th = t / kw_;
        return th * rowshift_ + (t - th * kw_) * dil_ - pad_;
    }
    __device__ __forceinline__ void load(const float* __restrict__ base, int64_t sr, int64_t sk, int r0, int nrows, int k0, int kend, int K) {
        const int tid = threadIdx.x;
        this->k0_ = k0; this->kend_ = kend; this->r0_ = r0; this->nrows_ = nrows;
        if constexpr (KC) {
            const int kmax = ((K + 3) & ~3) - 4;
#pragma unroll
            for (int p = 0; p < ROWS / 32; ++p) {
                const int f = tid + 256 * p, r = f >> 3, k = (f & 7) * 4;
                const int rc = min(r0 + r, nrows - 1), kc = min(k0 + k, kmax);
                const int ci = rc / taps_, sh = shift_of(rc - ci * taps_);
                this->v[p] = imp_load4(base + (int64_t)ci * sr, kc + sh, len_);
            }
        } else {
            const int rb = tid % (ROWS / 4), kb = min(tid / (ROWS / 4), XBK / 4 - 1);
            const int rmax = ((nrows + 3) & ~3) - 4;
            const int rc = min(r0 + rb * 4, rmax);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kc = min(k0 + kb * 4 + e, K - 1);
                const int ci = kc / taps_, sh = shift_of(kc - ci * taps_);
                this->v[e] = imp_load4(base + (int64_t)ci * sk, rc + sh, len_);
            }
        }
    }
};

// B tile of 8-bit codes [k][j], j contiguous in memory: a thread owns a 4 (rows j) x 4 (k) block -- four 4-B loads -- and stores ONE
// bf16 plane (an integer of at most 8 significant bits is exact in bf16)
template <int ROWS, bool SIGNED>
struct TileIOQ {
    unsigned int v[4];
    int k0_, kend_, r0_, nrows_;

    __device__ __forceinline__ void load(const void* __restrict__ base, int64_t /*sr*/, int64_t sk, int r0, int nrows, int k0, int kend, int K) {
        const int tid = threadIdx.x;
        k0_ = k0; kend_ = kend; r0_ = r0; nrows_ = nrows;
        const int rb = tid % (ROWS / 4), kb = min(tid / (ROWS / 4), XBK / 4 - 1);
        const int rmax = ((nrows + 3) & ~3) - 4;
        const int rc = min(r0 + rb * 4, rmax);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int kc = min(k0 + kb * 4 + e, K - 1);
            v[e] = *reinterpret_cast<const unsigned int*>((const unsigned char*)base + (int64_t)kc * sk + rc);
        }
    }

    __device__ __forceinline__ void store(unsigned short (*pl)[ROWS][XLDK]) const {
        const int tid = threadIdx.x;
        if (tid / (ROWS / 4) >= XBK / 4) return;
        const int rb = tid % (ROWS / 4), kb = tid / (ROWS / 4);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const bool rv = r0_ + rb * 4 + rr < nrows_;
            unsigned short h[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned int byte = (v[q] >> (8 * rr)) & 0xFFu;
                const float f = SIGNED ? (float)(int)(signed char)byte : (float)byte;
                h[q] = (rv && k0_ + kb * 4 + q < kend_) ? x_bf(f) : (unsigned short)0;
            }
            *reinterpret_cast<uint2*>(&pl[0][rb * 4 + rr][xsw(rb * 4 + rr, kb * 4)]) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
        }
    }
};

// BQ = 0: fp32 x fp32 (six products).  BQ = 1 / 2: the B operand arrives as 8-bit codes of a quantizer -- one exact bf16 plane, THREE
// products -- which is what the student's row-major gradient GEMMs can use (the q-GEMMs of csrc/qgemm.hip in row-major form):
//   1  wgrad  gw[o][i] += sum_r gz[r][o] x[r][i],  x = dx c + min_x:  A = gz^T, B = activation codes; the epilogue applies
//             dx * acc + min_x * sum_r gz[r][o] (the k-sums of A's rows accumulate next to the split)
//   2  dgrad  gx[r][i]  = sum_o gz[r][o] w_q[o][i], w_q = dw[o] wi:   A = gz scaled by dw[k] before the split, B = int8 weight codes
// (the body is a device function of (problem, tile coordinates): k_gemm_x3 runs it on its own grid, k_gemm_x3_wq_multi on the tiles of
// several problems in one launch)
template <bool A_KC, bool B_KC, bool ATOMIC, int MI, int NI, int BQ = 0, bool IMP = false, bool BPL = false>
__device__ __forceinline__ void x3_body(GemmArgs3 g, const int bx, const int by, const int bz) {
    constexpr int BMt = 64 * MI, BNt = 64 * NI;
    // LDS: three bf16 planes of the A tile, three (one for 8-bit codes) of the B tile, sized by the tile: the 64-row and the coded forms
    // leave room for a third / fourth workgroup per CU (128 x 128 float: 49 KB -> 3 per CU; 128 x 128 coded 33 KB -> 4; 64 x 128 coded
    // 20 KB -> 7) -- these short-K GEMMs are bound by the latency of a k-tile, not by its MFMAs, and a launch of 536 workgroups
    // on 512 slots ran in two rounds
    constexpr int kPlanesB = BQ ? 1 : 3;
    constexpr int kBytesA = 3 * BMt * XLDK * 2, kBytesB = kPlanesB * BNt * XLDK * 2, kBytesStage = 4 * 32 * 36 * 4;
    constexpr int kBytesLds = kBytesA + kBytesB > kBytesStage ? kBytesA + kBytesB : kBytesStage;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[kBytesLds];
    unsigned short(*As)[BMt][XLDK] = reinterpret_cast<unsigned short(*)[BMt][XLDK]>(lds_raw);
    unsigned short(*Bs)[BNt][XLDK] = reinterpret_cast<unsigned short(*)[BNt][XLDK]>(lds_raw + kBytesA);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int zb = ATOMIC ? bz / g.ksplit : bz;
    const int ks = ATOMIC ? bz - zb * g.ksplit : 0;
    g.A += (int64_t)zb * g.sAb;
    g.B += (int64_t)zb * g.sBb;
    g.C += (int64_t)zb * g.sCb;
    if constexpr (BQ != 0) g.Bq = static_cast<const unsigned char*>(g.Bq) + (int64_t)zb * g.sBb;      // (codes: 1 B per element)
    const int kbeg = ATOMIC ? ks * g.kchunk : 0;
    const int kend = ATOMIC ? min(g.K, kbeg + g.kchunk) : g.K;
    const int i0 = by * BMt, j0 = bx * BNt;

    static_assert(BQ == 0 || !B_KC, "coded B tiles are j-contiguous");
    static_assert(!IMP || BQ == 0, "implicit convolution: float operands");
    // PF register images of the operands in flight: the tile of step kt + PF is requested while step kt computes.  One image (rounds 1-3)
    // gave a load one k-tile's time to arrive; the short-K row GEMMs with one workgroup per CU (266 workgroups of 4 waves: one wave per
    // SIMD, nothing to switch to) then waited out most of an HBM round trip per k-tile.
    constexpr int PF = FQSS_X3_PF;
    using TileA = TileIO<BMt, A_KC>;
    static_assert(!BPL || (BQ == 0 && !IMP && B_KC && !ATOMIC), "pre-split B: the forward form");
    using TileB = std::conditional_t<BPL, TileIOP<BNt>,
                                     std::conditional_t<BQ != 0, TileIOQ<BNt, BQ == 2>, std::conditional_t<IMP, TileIOI<BNt, B_KC>, TileIO<BNt, B_KC>>>>;
    TileA ta[PF];
    TileB tb[PF];
#pragma unroll
    for (int s = 0; s < PF; ++s) {
        if constexpr (BQ == 2) ta[s].sc_ = g.scale_k;
        if constexpr (IMP) {
            tb[s].taps_ = g.imp_taps; tb[s].dil_ = g.imp_dil; tb[s].pad_ = g.imp_pad; tb[s].len_ = g.imp_len;
            tb[s].kw_ = g.imp_kw; tb[s].rowshift_ = g.imp_rowshift;
        }
    }
    auto load_b = [&](TileB& t, int k0) {
        if constexpr (BPL) t.load(g.Bp, (int64_t)g.N * g.ldp, g.ldp, j0, g.N, k0, g.K);
        else if constexpr (BQ != 0) t.load(g.Bq, g.sBj, g.sBk, j0, g.N, k0, kend, g.K);
        else t.load(g.B, g.sBj, g.sBk, j0, g.N, k0, kend, g.K);
    };
    __shared__ float rsum_s[(BQ == 1) ? 128 : 1];
    __shared__ float rsum_p[(BQ == 1) ? XBK / 4 : 1][(BQ == 1) ? 128 : 1];     // per k-block partial row sums (summed in k-block order)
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // the k loop runs whole rounds of PF tiles: a tile past the end is loaded from clamped addresses and STORED AS ZEROS (the stores
    // apply k < kend), so no branch surrounds a global load (see TileIO) and the surplus MFMAs add nothing
    const int nkt = (kend - kbeg + XBK - 1) / XBK, nkt_r = (nkt + PF - 1) / PF * PF;
    if (nkt > 0) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            ta[s].load(g.A, g.sAi, g.sAk, i0, g.M, kbeg + s * XBK, kend, g.K);
            load_b(tb[s], kbeg + s * XBK);
        }
        ta[0].store(As);
        tb[0].store(Bs);
    }
    __syncthreads();
    const int lr = lane & 31, lh = lane >> 5;
    for (int kt0 = 0; kt0 < nkt_r; kt0 += PF) {
#pragma unroll
      for (int s = 0; s < PF; ++s) {
        const int kt = kt0 + s;
        // image s was stored for this step: it takes the tile of step kt + PF, whose loads fly under PF steps of MFMAs
        ta[s].load(g.A, g.sAi, g.sAk, i0, g.M, kbeg + (kt + PF) * XBK, kend, g.K);
        load_b(tb[s], kbeg + (kt + PF) * XBK);
#pragma unroll
        for (int kstep = 0; kstep < XBK / 16; ++kstep) {
            constexpr int NPB = BQ ? 1 : 3;     // planes of B
            bf16x8 af[3][MI], bfr[NPB][NI];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    af[p][mi] = *reinterpret_cast<const bf16x8*>(&As[p][wm * (32 * MI) + mi * 32 + lr][xsw(lr, kstep * 16 + 8 * lh)]);
                if (p < NPB) {
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        bfr[p][ni] = *reinterpret_cast<const bf16x8*>(&Bs[p][wn * (32 * NI) + ni * 32 + lr][xsw(lr, kstep * 16 + 8 * lh)]);
                }
            }
            // smallest partial products first: l.h, h.l, m.m, m.h, h.m, h.h   (coded B: l.c, m.c, h.c)
            constexpr int NPROD = BQ ? 3 : 6;
            constexpr int IA[6] = {2, BQ ? 1 : 0, BQ ? 0 : 1, 1, 0, 0}, IB[6] = {0, BQ ? 0 : 2, BQ ? 0 : 1, 0, 1, 0};
#pragma unroll
            for (int sp = 0; sp < NPROD; ++sp)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[IA[sp]][mi], bfr[IB[sp]][ni], acc[mi][ni], 0, 0, 0);
        }
        __syncthreads();
        if (kt + 1 < nkt_r) {
            ta[(s + 1) % PF].store(As);
            tb[(s + 1) % PF].store(Bs);
            __syncthreads();
        }
      }
    }

    float dx = 1.0f, mnx = 0.0f;
    if constexpr (BQ == 1) {
        // k-sums of A's rows: the threads of a row block (same rb, different kb) meet in LDS
        // (a fixed order instead of LDS float atomics since round 5: the sums enter the result, and FQSS_DETERMINISTIC=1 promises its bits)
        if (threadIdx.x / (BMt / 4) < XBK / 4) {
            const int rb = threadIdx.x % (BMt / 4), kb = threadIdx.x / (BMt / 4);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                float t = ta[0].rs_[rr];
#pragma unroll
                for (int s = 1; s < PF; ++s) t += ta[s].rs_[rr];
                rsum_p[kb][rb * 4 + rr] = t;
            }
        }
        __syncthreads();
        if (threadIdx.x < BMt) {
            float t = rsum_p[0][threadIdx.x];
#pragma unroll
            for (int kb = 1; kb < XBK / 4; ++kb) t += rsum_p[kb][threadIdx.x];
            rsum_s[threadIdx.x] = t;
        }
        __syncthreads();
        const float lo = *g.qmin_x, hi = *g.qmax_x;
        dx = (hi - lo) / 255.0f;
        mnx = lo;
        // (the first column tile of every k-slice hands its row sums on: the bias gradient, summed over the slices by the atomics)
        if (g.rowsum_out != nullptr && bx == 0 && threadIdx.x < BMt && i0 + (int)threadIdx.x < g.M)
            grad_add(&g.rowsum_out[i0 + threadIdx.x], rsum_s[threadIdx.x]);
    }
    // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if constexpr (!ATOMIC) {
        // plain stores: each 32 x 32 accumulator tile goes through a wave-private LDS tile and leaves as whole 128-B rows, 16 B per lane
        // (the lane-per-column layout needs 16 strided 4-B stores per tile: that store-issue-bound tail was most of this kernel's time at
        // the skinny reductions of the dual-path linears, K = 256: 8 k-tiles of MFMAs against 64 scalar stores per thread).  Needs
        // 16-B aligned output rows; the scalar path below serves everything else.
        if (!g.scalar_stores && (g.sCi & 3) == 0 && (reinterpret_cast<uintptr_t>(g.C) & 15u) == 0) {
            constexpr int TLD = 36;                                    // floats per staged row (16-B aligned, conflict-light)
            static_assert(4 * 32 * TLD * 4 <= kBytesLds, "staging tiles fit the operand planes");
            float(*Tt)[TLD] = reinterpret_cast<float(*)[TLD]>(reinterpret_cast<float*>(lds_raw) + wave * 32 * TLD);
            const int c4 = (lane & 7) * 4, rq = lane >> 3;            // a lane moves 4 columns of rows rq, rq + 8, rq + 16, rq + 24
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    // (a 32 x 32 tile wholly outside the matrix -- the narrow outputs of the implicit convolutions -- stores nothing)
                    if (i0 + wm * (32 * MI) + mi * 32 >= g.M || j0 + wn * (32 * NI) + ni * 32 >= g.N) continue;
#pragma unroll
                    for (int r = 0; r < 16; ++r) Tt[(r & 3) + 8 * (r >> 2) + 4 * lh][lr] = acc[mi][ni][r];
                    __builtin_amdgcn_wave_barrier();
                    const int col = j0 + wn * (32 * NI) + ni * 32 + c4;
                    float4 bc = make_float4(0.f, 0.f, 0.f, 0.f);      // (scalar loads: a bias vector may sit at any 4-B aligned address)
                    if (g.bias_col != nullptr && col + 3 < g.N) bc = make_float4(g.bias_col[col], g.bias_col[col + 1], g.bias_col[col + 2], g.bias_col[col + 3]);
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int rl = rq + 8 * it, row = i0 + wm * (32 * MI) + mi * 32 + rl;
                        float4 v = *reinterpret_cast<const float4*>(&Tt[rl][c4]);
                        if (row < g.M) {
                            const float br = g.bias != nullptr ? g.bias[row] : 0.f;
                            float* dst = g.C + (int64_t)row * g.sCi + col;
                            if (col + 3 < g.N) {
                                if (g.bias != nullptr) { v.x = v.x + br; v.y = v.y + br; v.z = v.z + br; v.w = v.w + br; }
                                if (g.bias_col != nullptr) { v.x = v.x + bc.x; v.y = v.y + bc.y; v.z = v.z + bc.z; v.w = v.w + bc.w; }
                                *reinterpret_cast<float4*>(dst) = v;
                            } else {
                                const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                                for (int q = 0; q < 4; ++q)
                                    if (col + q < g.N) {
                                        float t = e[q];
                                        if (g.bias != nullptr) t = t + br;
                                        if (g.bias_col != nullptr) t = t + g.bias_col[col + q];
                                        dst[q] = t;
                                    }
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            return;
        }
    }
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
                    if constexpr (BQ == 1) v = dx * v + mnx * rsum_s[row - i0];
                    if (g.bias != nullptr && (!ATOMIC || kbeg == 0)) v = v + g.bias[row];
                    if (g.bias_col != nullptr && (!ATOMIC || kbeg == 0)) v = v + g.bias_col[col];
                    float* dst = g.C + (int64_t)row * g.sCi + col;
                    if constexpr (ATOMIC) grad_add(dst, v); else *dst = v;
                }
            }
        }
}

template <bool A_KC, bool B_KC, bool ATOMIC, int MI, int NI, int BQ = 0, bool IMP = false, bool BPL = false>
__global__ __launch_bounds__(256) void k_gemm_x3(GemmArgs3 g) {
    x3_body<A_KC, B_KC, ATOMIC, MI, NI, BQ, IMP, BPL>(g, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// Several coded weight gradients (BQ = 1: gw[o][i] += sum_r gz[r][o] x[r][i] on the input's codes) in ONE launch (round 5).  The weight
// gradients of a backward segment's row-major linears feed nothing but the optimizer, so the host queues them and runs them together
// (ops_dp.RowWgradQueue, runtime.QuantTables.finish_backward).  Why it pays: one such GEMM is 16 output tiles x 16 k-slices = 256
// workgroups -- ONE per CU, each a serial chain of ~17 k-tiles (load -> split -> LDS -> barrier -> MFMA, ~1 us each: these short-K
// GEMMs are bound by that latency, docs/history/DESIGN_rounds_1-5.md 7e (7)) -- so a launch takes ~37 us at 0.12 of its roofline with nothing to overlap the
// chain with.  32 problems in one grid are 4096+ workgroups: three to four resident per CU overlap their chains, the k-split can be
// coarser (fewer float atomics), and the ~8 us of fixed cost per launch is paid once.  The job table travels in the kernel arguments.
constexpr int X3W_MAXJOBS = 32;
struct X3WJob {
    const float* A; const void* Bq; float* C; float* rowsum_out; const float* qmin; const float* qmax;
    int64_t sAk, sBk, sCi;
    int M, N, K, ksplit, kchunk, gx, gy, start;      // grid of this problem (gx x gy tiles x ksplit slices), its first workgroup
};
struct X3WMulti {
    int n, total;
    X3WJob j[X3W_MAXJOBS];
};
static_assert(sizeof(X3WMulti) <= 4096, "the job table travels in the kernel arguments");

template <int MI, int NI>
__global__ __launch_bounds__(256) void k_gemm_x3_wq_multi(X3WMulti m) {
    int p = 0;
    for (int q = 1; q < m.n; ++q) p = (m.j[q].start <= (int)blockIdx.x) ? q : p;
    const X3WJob& J = m.j[p];
    const int local = (int)blockIdx.x - J.start, gx = J.gx, gy = J.gy;
    const int bx = local % gx, t = local / gx, by = t % gy, bz = t / gy;
    GemmArgs3 g{};
    g.A = J.A; g.C = J.C; g.M = J.M; g.N = J.N; g.K = J.K;
    g.sAi = 1; g.sAk = J.sAk; g.sBk = J.sBk; g.sBj = 1; g.sCi = J.sCi;
    g.ksplit = J.ksplit; g.kchunk = J.kchunk;
    g.Bq = J.Bq; g.qmin_x = J.qmin; g.qmax_x = J.qmax; g.rowsum_out = J.rowsum_out;
    g.batch = 1;
    x3_body<false, false, true, MI, NI, 1>(g, bx, by, bz);
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

// A/B switches of the tile rules (tools/kprobe.py): FQSS_X3_LDS_PAD = bytes of unused dynamic LDS per workgroup (caps the workgroups per
// CU; -1: every form padded to 61,440 B, the size before the planes were sized by the tile), FQSS_X3_MI = 1 / 2 forces the 64- / 128-row tile where both exist
static unsigned x3_lds_pad(int mi, int ni, bool coded) {
    static const int v = [] { const char* e = getenv("FQSS_X3_LDS_PAD"); return e ? atoi(e) : 0; }();
    if (v >= 0) return (unsigned)v;
    const int used = max(3 * 64 * mi * XLDK * 2 + (coded ? 1 : 3) * 64 * ni * XLDK * 2, 4 * 32 * 36 * 4);     // < 0: pad every form to the 61,440 B of rounds 1-3
    return used < 61440 ? (unsigned)(61440 - used) : 0u;
}
static int x3_force_mi() {
    static const int v = [] { const char* e = getenv("FQSS_X3_MI"); return e ? atoi(e) : 0; }();
    return v;
}

static int x3_scalar_stores() {
    static const int v = [] { const char* e = getenv("FQSS_X3_STAGED"); return (e && e[0] == '0') ? 1 : 0; }();
    return v;
}

int launch_gemm_x3(const GemmArgs3& g_in, bool a_kc, bool b_kc, bool atomic, hipStream_t s, const char* what, bool* used) {
    GemmArgs3 g = g_in;
    g.scalar_stores = x3_scalar_stores();
    *used = false;
    if (g.M <= 0 || g.N <= 0) return FQSS_OK;
    if (!x3_ok(g, a_kc, b_kc, atomic)) return FQSS_OK;     // caller falls back to k_gemm_f32
    const int64_t zdim = (int64_t)(g.batch > 0 ? g.batch : 1) * (atomic ? g.ksplit : 1);
    if (zdim > 65535) return FQSS_OK;
    int mi = 2, ni = 2;
    if (g.N <= 64) ni = 1;
    else if (g.M <= 64 || (g.M > 128 && cdiv(g.M, 128) * cdiv(g.N, 128) * zdim < 2 * 256)) mi = 1;
    // split-K weight gradients keep the 128-row tile: the split of the gradient operand is the cost that matters (these kernels are
    // bound by vector-ALU issue, ~20 VALU per MFMA with 64-row tiles) and a taller tile shares it among twice the MFMAs
    // (measured over the cfg 3 / 4 / 5 shapes, tools/kprobe.py: 512 x 512 float 91 -> 74 us, coded 71 -> 61 us)
    if (atomic && g.M > 64 && g.N > 64) mi = 2;
    // forward / data-gradient tiles: 64-row tiles only while the 128 x 128 grid would leave CUs idle (< 320 tiles: measured over the
    // cfg 3 / 4 / 5 shapes, tools/kprobe.py -- at 432 tiles the 128-row tile is 13 .. 30 % faster, at 250 the 64-row tile 10 .. 15 %);
    // the float data gradient (weight tile transposed on the way into LDS) takes 128 rows unless that leaves CUs without a workgroup
    if (!atomic && g.M > 128 && g.N > 64) {
        const int64_t t2 = cdiv(g.M, 128) * cdiv(g.N, 128) * zdim;
        mi = (a_kc && !b_kc) ? (t2 < 256 ? 1 : 2) : (t2 < 320 ? 1 : 2);
    }
    if (x3_force_mi() && ni == 2 && g.M > 64) mi = x3_force_mi() == 2 ? 2 : 1;
    if (x3_force_mi() == 3 && ni == 2 && g.M > 64 && !(a_kc && b_kc && g.Bp != nullptr)) ni = 1;      // 64 x 64 tiles (sweeps)
    dim3 grid((unsigned)cdiv(g.N, 64 * ni), (unsigned)cdiv(g.M, 64 * mi), (unsigned)zdim), block(256);
#define FQSS_X3(AK, BKc, AT)                                                                                  \
    do {                                                                                                      \
        if (mi == 1 && ni == 1) hipLaunchKernelGGL((k_gemm_x3<AK, BKc, AT, 1, 1>), grid, block, x3_lds_pad(1, 1, false), s, g);    \
        else if (mi == 2 && ni == 2) hipLaunchKernelGGL((k_gemm_x3<AK, BKc, AT, 2, 2>), grid, block, x3_lds_pad(2, 2, false), s, g);    \
        else if (ni == 1) hipLaunchKernelGGL((k_gemm_x3<AK, BKc, AT, 2, 1>), grid, block, x3_lds_pad(2, 1, false), s, g);          \
        else hipLaunchKernelGGL((k_gemm_x3<AK, BKc, AT, 1, 2>), grid, block, x3_lds_pad(1, 2, false), s, g);                       \
    } while (0)
    if (!atomic && a_kc && b_kc && g.Bp != nullptr) {                  // fwd on the weight's pre-split planes
        if (mi == 2 && ni == 2) hipLaunchKernelGGL((k_gemm_x3<true, true, false, 2, 2, 0, false, true>), grid, block, x3_lds_pad(2, 2, false), s, g);
        else if (ni == 1) hipLaunchKernelGGL((k_gemm_x3<true, true, false, 2, 1, 0, false, true>), grid, block, x3_lds_pad(2, 1, false), s, g);
        else hipLaunchKernelGGL((k_gemm_x3<true, true, false, 1, 2, 0, false, true>), grid, block, x3_lds_pad(1, 2, false), s, g);
    } else if (!atomic && a_kc && b_kc) FQSS_X3(true, true, false);   // fwd:   x [R][Ci], w [Co][Ci]
    else if (!atomic && a_kc && !b_kc) FQSS_X3(true, false, false);   // dgrad: gz [R][Co], w [Co][Ci] (j contiguous)
    else if (atomic && !a_kc && !b_kc) FQSS_X3(false, false, true);   // wgrad: gz^T, x (both row-index contiguous)
    else if (!atomic && !a_kc && !b_kc) FQSS_X3(false, false, false); // channel-first dgrad: W^T (i contiguous), gz [Co][M]
    else if (atomic && a_kc && b_kc) FQSS_X3(true, true, true);       // channel-first wgrad: gz [Co][M], x [Ci][M] (both k contiguous)
    else return FQSS_OK;
#undef FQSS_X3
    *used = true;
    return launch_status(what);
}

// implicit stride-1 convolution forms: forward / data gradient (A = weight [Co][Ci * taps], k contiguous; B implicit, positions
// contiguous) and weight gradient (A = gz [Co][positions]; B implicit with (channel, tap) rows; split-K + atomics, batches added)
int launch_gemm_x3_imp(const GemmArgs3& g_in, bool wgrad, hipStream_t s, const char* what) {
    GemmArgs3 g = g_in;
    g.scalar_stores = x3_scalar_stores();
    if (g.M <= 0 || g.N <= 0) return FQSS_OK;
    const int64_t zdim = (int64_t)(g.batch > 0 ? g.batch : 1) * (wgrad ? g.ksplit : 1);
    if (zdim > 65535) { set_error("%s: too many batches x k-slices", what); return FQSS_EINVAL; }
    int mi = 2, ni = 2;
    if (g.N <= 64) ni = 1;
    else if (g.M <= 64) mi = 1;
    // narrow outputs (the DConv convolutions, C / 8 channels): 64 x 128 tiles of one batch leave half the chip without a workgroup
    // (4 x 1 x 32 = 128 of them at [B F][C][431]) -- 64 x 64 tiles there
    if (!wgrad && mi == 1 && ni == 2 && cdiv(g.N, 128) * cdiv(g.M, 64) * zdim < 256 && x3_force_mi() != 1) ni = 1;
    dim3 grid((unsigned)cdiv(g.N, 64 * ni), (unsigned)cdiv(g.M, 64 * mi), (unsigned)zdim), block(256);
#define FQSS_X3I(BKc, AT)                                                                                              \
    do {                                                                                                               \
        if (mi == 1 && ni == 1) hipLaunchKernelGGL((k_gemm_x3<true, BKc, AT, 1, 1, 0, true>), grid, block, x3_lds_pad(1, 1, false), s, g);  \
        else if (mi == 2 && ni == 2) hipLaunchKernelGGL((k_gemm_x3<true, BKc, AT, 2, 2, 0, true>), grid, block, x3_lds_pad(2, 2, false), s, g);  \
        else if (ni == 1) hipLaunchKernelGGL((k_gemm_x3<true, BKc, AT, 2, 1, 0, true>), grid, block, x3_lds_pad(2, 1, false), s, g);        \
        else hipLaunchKernelGGL((k_gemm_x3<true, BKc, AT, 1, 2, 0, true>), grid, block, x3_lds_pad(1, 2, false), s, g);                     \
    } while (0)
    if (wgrad) FQSS_X3I(true, true); else FQSS_X3I(false, false);
#undef FQSS_X3I
    return launch_status(what);
}

// coded-B forms (BQ = 1 wgrad, 2 dgrad); the caller has checked shapes and alignment
int launch_gemm_x3q(const GemmArgs3& g_in, int bq, hipStream_t s, const char* what) {
    GemmArgs3 g = g_in;
    g.scalar_stores = x3_scalar_stores();
    if (g.M <= 0 || g.N <= 0) return FQSS_OK;
    const bool atomic = bq == 1;
    if (g.batch < 1) g.batch = 1;
    FQSS_REQUIRE(atomic || g.batch == 1, "coded data gradient: one problem per launch");
    const int64_t zdim = atomic ? (int64_t)g.ksplit * g.batch : 1;
    int mi = 2, ni = 2;
    if (g.N <= 64) ni = 1;
    else if (g.M <= 64 || (g.M > 128 && cdiv(g.M, 128) * cdiv(g.N, 128) * zdim < 2 * 256)) mi = 1;
    // split-K weight gradients keep the 128-row tile: the split of the gradient operand is the cost that matters (these kernels are
    // bound by vector-ALU issue, ~20 VALU per MFMA with 64-row tiles) and a taller tile shares it among twice the MFMAs
    // (measured over the cfg 3 / 4 / 5 shapes, tools/kprobe.py: 512 x 512 float 91 -> 74 us, coded 71 -> 61 us)
    if (atomic && g.M > 64 && g.N > 64) mi = 2;
    if (!atomic && g.M > 128 && g.N > 64) mi = cdiv(g.M, 128) * cdiv(g.N, 128) * zdim < 320 ? 1 : 2;
    if (x3_force_mi() && ni == 2 && g.M > 64) mi = x3_force_mi() == 2 ? 2 : 1;
    if (x3_force_mi() == 3 && ni == 2 && g.M > 64) ni = 1;
    dim3 grid((unsigned)cdiv(g.N, 64 * ni), (unsigned)cdiv(g.M, 64 * mi), (unsigned)zdim), block(256);
#define FQSS_X3Q(AK, AT, Q)                                                                                      \
    do {                                                                                                         \
        if (mi == 1 && ni == 1) hipLaunchKernelGGL((k_gemm_x3<AK, false, AT, 1, 1, Q>), grid, block, x3_lds_pad(1, 1, true), s, g);  \
        else if (mi == 2 && ni == 2) hipLaunchKernelGGL((k_gemm_x3<AK, false, AT, 2, 2, Q>), grid, block, x3_lds_pad(2, 2, true), s, g);  \
        else if (ni == 1) hipLaunchKernelGGL((k_gemm_x3<AK, false, AT, 2, 1, Q>), grid, block, x3_lds_pad(2, 1, true), s, g);        \
        else hipLaunchKernelGGL((k_gemm_x3<AK, false, AT, 1, 2, Q>), grid, block, x3_lds_pad(1, 2, true), s, g);                     \
    } while (0)
    if (bq == 1) FQSS_X3Q(false, true, 1);
    else FQSS_X3Q(true, false, 2);
#undef FQSS_X3Q
    return launch_status(what);
}

}  // namespace fqss

using namespace fqss;

/* fqss_qrow_bwd_wb for SEVERAL linears in one launch per <= 32 jobs of one tile shape (k_gemm_x3_wq_multi). */
extern "C" int fqss_qrow_bwd_w_group(const FqssRowWgradJob* jobs, int njobs, fqss_stream_t stream) {
    if (njobs == 0) return FQSS_OK;
    FQSS_REQUIRE(jobs && njobs > 0, "no jobs");
    for (int q = 0; q < njobs; ++q) {
        const FqssRowWgradJob& f = jobs[q];
        FQSS_REQUIRE(f.gz && f.xc && f.qmin_x && f.qmax_x && f.gw, "null tensor");
        FQSS_REQUIRE(f.R > 0 && f.R < (1ll << 31) && f.Ci > 0 && f.Co > 0 && f.ld_gz >= f.Co && f.ld_xc >= f.Ci && f.ld_gw >= f.Ci, "bad shape");
        FQSS_REQUIRE(f.Ci % 4 == 0 && f.Co % 4 == 0 && f.ld_gz % 4 == 0 && f.ld_xc % 4 == 0 && aligned16(f.gz) && ((uintptr_t)f.xc & 3) == 0,
                     "coded wgrad: Ci, Co and the row strides must be multiples of 4, operands aligned");
    }
    FQSS_REQUIRE(njobs <= 1024, "at most 1024 jobs per call (the per-shape index list): flush the queue more often");
    // one launch per tile shape (the rule of launch_gemm_x3q for split-K weight gradients) and per 32 jobs
    for (int shape = 0; shape < 4; ++shape) {
        const int mi = (shape & 1) ? 2 : 1, ni = (shape & 2) ? 2 : 1;
        int idx[1024], cnt = 0;
        for (int q = 0; q < njobs && cnt < 1024; ++q) {
            const FqssRowWgradJob& f = jobs[q];
            const int jni = f.Ci <= 64 ? 1 : 2;
            const int jmi = (f.Co > 64 && f.Ci > 64) ? 2 : ((jni == 2 && f.Co <= 64) ? 1 : 2);
            if (jmi == mi && jni == ni) idx[cnt++] = q;
        }
        for (int n0 = 0; n0 < cnt; n0 += X3W_MAXJOBS) {
            const int n = cnt - n0 < X3W_MAXJOBS ? cnt - n0 : X3W_MAXJOBS;
            int64_t tiles = 0;
            for (int q = 0; q < n; ++q) tiles += cdiv(jobs[idx[n0 + q]].Co, 64 * mi) * cdiv(jobs[idx[n0 + q]].Ci, 64 * ni);
            // aim for ~6 workgroups per CU over the launch (sweep on cfg 4, profiles/r05_ab_steps.txt: 512 .. 6144 within 2.5 %, 1536 best); every
            // k-slice ADDS its whole tile with float atomics, so no finer than needed
            static const int wgs = [] { const char* e = getenv("FQSS_ROWGROUP_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 1536; }();
            int want = (int)cdiv(wgs, tiles);
            X3WMulti m{};
            m.n = n;
            int start = 0;
            for (int q = 0; q < n; ++q) {
                const FqssRowWgradJob& f = jobs[idx[n0 + q]];
                X3WJob& J = m.j[q];
                J.A = f.gz; J.Bq = f.xc; J.C = f.gw; J.rowsum_out = f.gbias; J.qmin = f.qmin_x; J.qmax = f.qmax_x;
                J.sAk = f.ld_gz; J.sBk = f.ld_xc; J.sCi = f.ld_gw;
                J.M = f.Co; J.N = f.Ci; J.K = (int)f.R;
                int kchunk = (int)cdiv(cdiv(f.R, want), 64) * 64;
                if (kchunk < 256) kchunk = 256;
                J.kchunk = kchunk;
                J.ksplit = (int)cdiv(f.R, kchunk);
                J.gx = (int)cdiv(f.Ci, 64 * ni); J.gy = (int)cdiv(f.Co, 64 * mi);
                J.start = start;
                start += J.gx * J.gy * J.ksplit;
            }
            m.total = start;
            const dim3 grid((unsigned)start), block(256);
            hipStream_t s = (hipStream_t)stream;
            if (mi == 2 && ni == 2) hipLaunchKernelGGL((k_gemm_x3_wq_multi<2, 2>), grid, block, 0, s, m);
            else if (mi == 2) hipLaunchKernelGGL((k_gemm_x3_wq_multi<2, 1>), grid, block, 0, s, m);
            else if (ni == 2) hipLaunchKernelGGL((k_gemm_x3_wq_multi<1, 2>), grid, block, 0, s, m);
            else hipLaunchKernelGGL((k_gemm_x3_wq_multi<1, 1>), grid, block, 0, s, m);
            if (int rc = launch_status("fqss_qrow_bwd_w_group")) return rc;
        }
    }
    return FQSS_OK;
}
