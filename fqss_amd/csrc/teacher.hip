// teacher.hip -- the frozen float teacher of the KD step as a fused inference chain.
//
// The teacher (train_env/train_utils.py:25: deepcopy of the float model; mysystem.py:132-133: forward under
// no_grad) needs no autograd and its weights never change, so per TCN block (convtasnetq.py:11-42) it runs as
// THREE kernels instead of the ~11 of the module-by-module path:
//   T1  k_tgemm        1x1 conv (+bias) + PReLU, GroupNorm statistics (sum, sum^2 per sample) in the epilogue
//   T2  k_tdw          GroupNorm-apply on the fly + depthwise dilated conv (+bias) + PReLU + statistics
//   T3  k_tgemm        res and skip 1x1 convs as ONE GEMM (Co = 256): GroupNorm-apply in the B-operand
//                      prologue; epilogue adds the residual (rows < 128 -> next block input) and the running
//                      skip sum (rows >= 128)
// Every N_F-sized tensor is written once and read once (4 passes per block instead of ~17).
// GEMMs: both operands fp32; the frozen weights are split ONCE into three exact bf16 planes
// (fqss_split3_planes), activations are split on the fly: the 6 leading exact bf16 products per k, fp32 accumulation
// (v_mfma_f32_32x32x16_bf16) -- the result is fp32-grade (same error class as an fp32 fma chain).
#include <type_traits>

#include <cstdlib>

#include "fqss_dev.h"

namespace fqss {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int TBM = 128, TBN = 128, TBK = 32;
constexpr int TLDK = 40, TLDN = 160, TLDT = 36;
// GroupNorm statistics travel as kTSlots partial (sum, sum^2) pairs per sample, each pair on its own 128-B
// line: same-address fp64 atomics serialise at ~12 ns apiece, spreading them over 32 lines removes that wall.
constexpr int kTSlots = FQSS_TSTAT_SLOTS, kTSlotStride = FQSS_TSTAT_STRIDE;

__device__ __forceinline__ unsigned short t_bf(float f) { return (unsigned short)(__float_as_uint(f) >> 16); }
__device__ __forceinline__ float t_tr(float f) { return __uint_as_float(__float_as_uint(f) & 0xFFFF0000u); }
__device__ __forceinline__ void t_split3(float g, unsigned short& b1, unsigned short& b2, unsigned short& b3) {
    const float h1 = t_tr(g), r1 = g - h1, h2 = t_tr(r1), r2 = r1 - h2;
    b1 = t_bf(h1);
    b2 = t_bf(h2);
    b3 = t_bf(r2);
}

// mean / rstd of one sample from its slot partials -> out2[0..1] (LDS); every thread of the block calls this.
__device__ __forceinline__ void t_stats_finalize(const double* __restrict__ st, double count, float eps, float* out2) {
    if (threadIdx.x < 64) {
        double a = 0.0, q = 0.0;
        if (threadIdx.x < kTSlots) {
            a = st[threadIdx.x * kTSlotStride];
            q = st[threadIdx.x * kTSlotStride + 1];
        }
        a = wave_sum(a);
        q = wave_sum(q);
        if (threadIdx.x == 0) {
            const double mu = a / count;
            double var = q / count - mu * mu;
            if (var < 0.0) var = 0.0;
            out2[0] = (float)mu;
            out2[1] = (float)(1.0 / sqrt(var + (double)eps));
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_split3_planes(const float* __restrict__ w, unsigned short* __restrict__ planes,
                                                        int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        unsigned short a, b, c;
        t_split3(w[i], a, b, c);
        planes[i] = a;
        planes[n + i] = b;
        planes[2 * n + i] = c;
    }
}

// The same three planes in the ORDER k_tgemm2 consumes them: [row tile of 256][k-tile of 16][plane][256 rows][2 chunks of 8 k], the two
// 16-B chunks of a row swapped where (row >> 3) & 1 (conflict-free ds_read_b128 of the 32 x 16 fragments) -- one k-tile of one row tile
// is 24 KB of CONTIGUOUS memory, so a wave's LDS-DMA instruction reads 1 KB of whole cache lines (the [3][Co][Ci] planes gave it
// 32-B pieces of 32 rows: several times the address work per byte on the CU's vector-memory path).  Frozen weights: packed once.
__global__ __launch_bounds__(256) void k_split3_tiles(const float* __restrict__ w, unsigned short* __restrict__ tiles, int Co, int Ci) {
    const int nkt = Ci / 16;
    const int64_t n = (int64_t)Co * Ci;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int e = (int)(i & 7), pc = (int)(i >> 3) & 1, row = (int)(i >> 4) & 255;
        const int64_t tile = i >> 12;                       // (mt * nkt + kt)
        const int kt = (int)(tile % nkt), mt = (int)(tile / nkt);
        const int c = pc ^ ((row >> 3) & 1);
        unsigned short a, b, d;
        t_split3(w[(int64_t)(mt * 256 + row) * Ci + kt * 16 + c * 8 + e], a, b, d);
        const int64_t o = tile * (3 * 4096) + row * 16 + pc * 8 + e;
        tiles[o] = a;
        tiles[o + 4096] = b;
        tiles[o + 2 * 4096] = d;
    }
}

// FQSS_TDIAG (diagnostic builds only, tools/r03_bisect.sh; default 0 = the product): bit 0 no MFMAs | 1 plain ds_read_b64 instead of the
// transposed reads | 2 no LDS stores of the staged tiles | 3 no global loads in the main loop | 4 no epilogue (results are then garbage:
// these builds only serve as the AGGRESSOR of tools/ubench/pkadd_next_to_mfma.hip)
#ifndef FQSS_TDIAG
#define FQSS_TDIAG 0
#endif
typedef float f32x4t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4t __attribute__((ext_vector_type(4)));
struct TStage {          // one k-tile of global loads per thread: 3 planes x 2 x 16 B of the weight, 4 x 16 B of activations
    u32x4t ra[3][2];
    f32x4t rb[4];
};
__device__ __forceinline__ void t_load16(u32x4t& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void t_load16(f32x4t& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
template <int N>
__device__ __forceinline__ void t_wait(TStage& st) {
    asm volatile("s_waitcnt vmcnt(%10)"
                 : "+v"(st.ra[0][0]), "+v"(st.ra[0][1]), "+v"(st.ra[1][0]), "+v"(st.ra[1][1]), "+v"(st.ra[2][0]), "+v"(st.ra[2][1]),
                   "+v"(st.rb[0]), "+v"(st.rb[1]), "+v"(st.rb[2]), "+v"(st.rb[3])
                 : "n"(N)
                 : "memory");
}

struct TGemmArgs {
    const unsigned short* A;   // [3][M][K] bf16 planes of the weight
    const float* B;            // [batch][K][ldb] fp32 activations
    int M, N, K;
    int64_t ldb, sBb;
    // prologue on B rows (k = input channel): 0 none, 1 GroupNorm affine, 2 PReLU
    int pro;
    const double* pro_stats;   // [batch][2] sum, sum^2 of the producer's output
    const float* pro_gamma;    // [K]
    const float* pro_beta;     // [K]
    double pro_count;          // elements per sample (K * N)
    float pro_eps;
    const float* pro_slope;    // PReLU on the input
    // epilogue
    const float* bias;         // [M] or null
    int act;                   // FQSS_ACT_*
    const float* slope;
    double* stats_out;         // [batch][2] or null
    int M1;                    // rows [0,M1) -> C1 (+R1), rows [M1,M) -> C2 (+R2)
    float* C1; const float* R1; int64_t ldc1, sC1b;
    float* C2; const float* R2; int64_t ldc2, sC2b;
    int tiles_m, tiles_n, batches;   // logical grid (launched 1-D in XCD-aware order, fqss_dev.h)
    int stagger;                     // k_tgemm2: cycles by which every second workgroup starts late (0: none; see its header)
};

template <int PRO>   // prologue on the activation rows: 0 none | 1 GroupNorm affine | 2 PReLU
__global__ __launch_bounds__(256, 2) void k_tgemm(TGemmArgs g) {
    constexpr int A_BYTES = 3 * TBM * TLDK * 2, B_BYTES = 3 * TBK * TLDN * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[A_BYTES + B_BYTES];
    typedef unsigned short (*AsT)[TBM][TLDK];
    typedef unsigned short (*BsT)[TBK][TLDN];
    AsT As = reinterpret_cast<AsT>(smem);
    BsT Bs = reinterpret_cast<BsT>(smem + A_BYTES);
    // The small per-workgroup arrays live INSIDE the tile buffer, which is idle when they are in use (pms: before the first tile is
    // stored; rowb / red: in the epilogue, behind its 4 x 4,608-B staging tiles): as separate arrays they brought the PRO = 1 instance to
    // 66,120 B of static LDS, and a kernel above 64 KB made LDS-resident data of kernels running on ANOTHER stream unreliable
    // (tools/stress_streams.py, docs/history/DESIGN_rounds_1-5.md 9).  With them folded in every instance is <= 65,536 B.
    float* rowb = reinterpret_cast<float*>(smem + 4 * 32 * TLDT * 4);                     // [TBM]
    double* red = reinterpret_cast<double*>(smem + 4 * 32 * TLDT * 4 + TBM * 4);          // [2 * 4]
    float* pms = reinterpret_cast<float*>(smem);                                          // [2]
    static_assert(4 * 32 * TLDT * 4 + TBM * 4 + 64 <= A_BYTES + B_BYTES, "epilogue scratch fits the tile buffer");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform on purpose: everything derived from it stays in SGPRs
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
    // group = one (sample, n-tile) activation panel, re-read by the tiles_m row tiles of the weight
    int panel, mt;
    if (!xcd_tile(g.tiles_n * g.batches, g.tiles_m, panel, mt)) return;
    const int b = panel / g.tiles_n, i0 = mt * TBM, j0 = (panel % g.tiles_n) * TBN;
    const int gq = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;

    if (PRO == 1) t_stats_finalize(g.pro_stats + (int64_t)b * kTSlots * kTSlotStride, g.pro_count, g.pro_eps, pms);
    else __syncthreads();
    const float pmean = (PRO == 1) ? pms[0] : 0.f, prstd = (PRO == 1) ? pms[1] : 1.f;
    __shared__ float pco[2][512];   // GroupNorm-apply coefficients of the K input channels: t = x * pco[0][k] + pco[1][k]
    if (PRO == 1) {
        for (int k = tid; k < g.K; k += 256) {
            const float pa = prstd * g.pro_gamma[k];
            pco[0][k] = pa;
            pco[1][k] = fmaf(-pa, pmean, g.pro_beta[k]);
        }
    }
    __syncthreads();
    float pslope = (PRO == 2) ? *g.pro_slope : 0.0f;
    if (PRO == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(pslope));   // not inside the hand-scheduled loop

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int a_row = tid >> 1, a_k = (tid & 1) * 16;
    const int bk_row = tid >> 3, bk_c = (tid & 7) * 4;   // 4 float4 per thread at columns 32*q + bk_c
    const float* Bb = g.B + (int64_t)b * g.sBb;
    const int64_t plane = (int64_t)g.M * g.K;
    // Two register stages of global loads: the tile stored at the end of iteration kt was requested two iterations
    // earlier (PMC: with one stage the waves sat parked on vmcnt/barriers 40 % of the time).  Loads are unconditional
    // -- addresses clamped into the operand, out-of-range k zeroed at use -- so no branch ever surrounds a load.
    // The compiler sinks plain loads next to their first use (which would collapse the two stages into none), so the
    // loads are issued through asm and retired by an explicit s_waitcnt that carries the stage's registers as
    // in/out operands; nothing else in the loop touches vmcnt (the GroupNorm coefficients come from LDS).
    const int a_row_c = min(i0 + a_row, g.M - 1);
    const int n_last = (g.N - 1) & ~3;
    auto load_tiles = [&](TStage& st, int k0) {
        if (FQSS_TDIAG & 8) return;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                t_load16(st.ra[p][h], g.A + p * plane + (int64_t)a_row_c * g.K + min(k0 + a_k + 8 * h, g.K - 8));
        const int k = min(k0 + bk_row, g.K - 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) t_load16(st.rb[q], Bb + (int64_t)k * g.ldb + min(j0 + 32 * q + bk_c, n_last));
    };
    auto store_tiles = [&](TStage& st, int k0) {
        t_wait<10>(st);   // this stage has landed; the 10 younger requests of the other stage stay in flight
        if (FQSS_TDIAG & 4) return;
        const int kc = min(k0 + bk_row, g.K - 1);
        const float p_a = (PRO == 1) ? pco[0][kc] : 1.f, p_b = (PRO == 1) ? pco[1][kc] : 0.f;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            *reinterpret_cast<u32x4t*>(&As[p][a_row][a_k]) = st.ra[p][0];
            *reinterpret_cast<u32x4t*>(&As[p][a_row][a_k + 8]) = st.ra[p][1];
        }
        const bool ok = (k0 + bk_row) < g.K;   // rows of B past K contribute zeros (A needs no mask: finite x 0)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float x[4] = {st.rb[q][0], st.rb[q][1], st.rb[q][2], st.rb[q][3]};
            float h0[4], r1[4], r2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = x[e];
                if (PRO == 1) t = fmaf(t, p_a, p_b);
                else if (PRO == 2) t = t > 0.f ? t : pslope * t;
                if (!ok) t = 0.f;
                h0[e] = t;
                r1[e] = t - t_tr(t);
                r2[e] = r1[e] - t_tr(r1[e]);
            }
            // bf16 pairs: v_perm picks the high halves of two fp32 words
            uint2 o1, o2, o3;
            o1.x = __builtin_amdgcn_perm(__float_as_uint(h0[1]), __float_as_uint(h0[0]), 0x07060302u);
            o1.y = __builtin_amdgcn_perm(__float_as_uint(h0[3]), __float_as_uint(h0[2]), 0x07060302u);
            o2.x = __builtin_amdgcn_perm(__float_as_uint(r1[1]), __float_as_uint(r1[0]), 0x07060302u);
            o2.y = __builtin_amdgcn_perm(__float_as_uint(r1[3]), __float_as_uint(r1[2]), 0x07060302u);
            o3.x = __builtin_amdgcn_perm(__float_as_uint(r2[1]), __float_as_uint(r2[0]), 0x07060302u);
            o3.y = __builtin_amdgcn_perm(__float_as_uint(r2[3]), __float_as_uint(r2[2]), 0x07060302u);
            *reinterpret_cast<uint2*>(&Bs[0][bk_row][32 * q + bk_c]) = o1;
            *reinterpret_cast<uint2*>(&Bs[1][bk_row][32 * q + bk_c]) = o2;
            *reinterpret_cast<uint2*>(&Bs[2][bk_row][32 * q + bk_c]) = o3;
        }
    };
    auto compute_tile = [&]() {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[3][2], bfr[3][2];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    af[p][mi] = *reinterpret_cast<const bf16x8*>(&As[p][wm * 64 + mi * 32 + lr][ks * 16 + 8 * lh]);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int kr = ks * 16 + 8 * (gq >> 1) + tq;
                    const int nc = wn * 64 + ni * 32 + 16 * (gq & 1) + 4 * tp;
                    union { bf16x8 v; s16x4 h[2]; } u;
                    if (FQSS_TDIAG & 2) {
                        u.h[0] = *reinterpret_cast<const s16x4*>(&Bs[p][kr][nc]);
                        u.h[1] = *reinterpret_cast<const s16x4*>(&Bs[p][kr + 4][nc]);
                    } else {
                        u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(&Bs[p][kr][nc]));
                        u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(&Bs[p][kr + 4][nc]));
                    }
                    bfr[p][ni] = u.v;
                }
            }
            // the 6 partial products of weight >= 2^-16 (the dropped a2*b3, a3*b2, a3*b3 are below the fp32
            // rounding of the sum), smallest pieces first
            constexpr int IA[6] = {2, 0, 1, 1, 0, 0}, IB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int sp = 0; sp < 6; ++sp)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        if (!(FQSS_TDIAG & 1)) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[IA[sp]][mi], bfr[IB[sp]][ni], acc[mi][ni], 0, 0, 0);
                        else acc[mi][ni][sp] += (float)af[IA[sp]][mi][0] + (float)bfr[IB[sp]][ni][0];      // keeps the fragment reads alive
        }
    };

    const int nkt = (g.K + TBK - 1) / TBK;
    TStage stA, stB;
    load_tiles(stA, 0);
    load_tiles(stB, TBK);
    store_tiles(stA, 0);
    load_tiles(stA, 2 * TBK);
    __syncthreads();
    // invariant at the top of iteration kt (even): LDS = tile kt, stB = tile kt+1, stA = tile kt+2 (requests in flight)
    for (int kt = 0; kt < nkt; kt += 2) {
        compute_tile();
        __syncthreads();
        if (kt + 1 < nkt) {
            store_tiles(stB, (kt + 1) * TBK);
            load_tiles(stB, (kt + 3) * TBK);
            __syncthreads();
            compute_tile();
            __syncthreads();
            if (kt + 2 < nkt) {
                store_tiles(stA, (kt + 2) * TBK);
                load_tiles(stA, (kt + 4) * TBK);
                __syncthreads();
            }
        }
    }

    // the trailing (clamped, unused) requests still target the stage registers, which the compiler considers dead
    // from here on: drain them before anything else is allocated there
    t_wait<0>(stA);
    t_wait<0>(stB);
    if (FQSS_TDIAG & 16) {      // no epilogue: one store keeps the accumulators alive
        if (acc[0][0][0] + acc[0][1][1] + acc[1][0][2] + acc[1][1][3] == 12345.678f) g.C1[0] = 1.0f;
        return;
    }
    // (every wave has passed the barrier behind the last tile's fragment reads: the tile buffer is free)
    if (tid < TBM) rowb[tid] = (g.bias != nullptr && i0 + tid < g.M) ? g.bias[i0 + tid] : 0.0f;
    __syncthreads();

    // epilogue: bias + act (+ residual), 16-B/lane row stores via an LDS staging tile, statistics.
    // Branch-free on purpose: with the activation and the output half selected per element through run-time branches
    // and 64-bit pointer selects the compiler emitted ~3400 instructions here (160 scalar branches) and the epilogue
    // took 47 % (T1) / 28 % (T3) of a workgroup's life (cycle-counter trace).  NONE / ReLU are PReLU with slope 1 / 0.
    const float nscale = (g.act == FQSS_ACT_PRELU) ? *g.slope : (g.act == FQSS_ACT_RELU ? 0.0f : 1.0f);
    float(*Tt)[TLDT] = reinterpret_cast<float(*)[TLDT]>(smem + wave * 32 * TLDT * 4);
    float s1 = 0.0f, s2 = 0.0f;   // <= 64 values per thread: fp32 partials, widened to fp64 for the cross-thread sum
    const bool want_stats = g.stats_out != nullptr;
    const int c4 = (lane & 7) * 4;
    // destination (and residual) of output row `row` at column 0: the first M1 rows go to C1, the rest to C2
    auto row_base = [&](int row, float*& Crow, const float*& Rrow) {
        const bool first = row < g.M1;
        const int64_t off = (int64_t)b * (first ? g.sC1b : g.sC2b) + (int64_t)(first ? row : row - g.M1) * (first ? g.ldc1 : g.ldc2);
        Crow = (first ? g.C1 : g.C2) + off;
        const float* Rsel = first ? g.R1 : g.R2;
        Rrow = (Rsel != nullptr) ? Rsel + off : nullptr;
    };
    // PER_ROW = false: M1 % 32 == 0, a 32-row tile lies on one side of the split and one base serves the whole tile
    // (every shape of the real model); PER_ROW = true: the general case, selected per stored row
    auto tile_epilogue = [&](int mi, auto PER_ROW) {
        const int rowt = i0 + wm * 64 + mi * 32;
        float* Cb;
        const float* Rb;
        row_base(rowt, Cb, Rb);
        const int64_t ldc = (rowt < g.M1) ? g.ldc1 : g.ldc2;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float v = acc[mi][ni][r] + rowb[wm * 64 + mi * 32 + rl];
                Tt[rl][lr] = v > 0.0f ? v : nscale * v;
            }
            const int col = j0 + wn * 64 + ni * 32 + c4;
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int rl = pass * 8 + (lane >> 3);
                float4 t = *reinterpret_cast<const float4*>(&Tt[rl][c4]);
                if (rowt + rl < g.M && col < g.N) {
                    float* Crow;
                    const float* Rrow;
                    if constexpr (decltype(PER_ROW)::value) {
                        row_base(rowt + rl, Crow, Rrow);
                    } else {
                        Crow = Cb + (int64_t)rl * ldc;
                        Rrow = Rb + (int64_t)rl * ldc;   // only dereferenced when Rb != nullptr
                    }
                    if (decltype(PER_ROW)::value ? Rrow != nullptr : Rb != nullptr) {
                        const float4 q = *reinterpret_cast<const float4*>(Rrow + col);
                        t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
                    }
                    *reinterpret_cast<float4*>(Crow + col) = t;
                    if (want_stats) {
                        const float v[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (col + e < g.N) {
                                s1 += v[e];
                                s2 = fmaf(v[e], v[e], s2);
                            }
                    }
                }
            }
        }
    };
    if ((g.M1 & 31) == 0 || g.M1 >= g.M) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) tile_epilogue(mi, std::false_type{});
    } else {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) tile_epilogue(mi, std::true_type{});
    }
    if (g.stats_out != nullptr) {
        double v[2] = {(double)s1, (double)s2};
        block_sum<double, 2>(v, red);
        if (tid == 0) {
            double* so = g.stats_out + ((int64_t)b * kTSlots + ((panel * g.tiles_m + mt) & (kTSlots - 1))) * kTSlotStride;
            atomicAdd(&so[0], v[0]);
            atomicAdd(&so[1], v[1]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_tgemm2 (round 4): the same GEMM restructured so that memory, split and MFMA overlap INSIDE a workgroup (VERDICT r03 next #1;
// the round-3 ablation of k_tgemm: 43 us with its MFMAs removed, 54.5 without its epilogue, 62-70 in full, floor 20).
//   * one workgroup owns ALL rows of a 256-row weight tile for a 128-column activation panel: the panel is read and split ONCE per
//     256 output rows (k_tgemm: once per 128);
//   * eight waves in three ROLES, one compute wave + one memory wave per SIMD (256 registers each):
//       waves 0-3  compute: fragment reads + 48 MFMAs per 16-deep k-tile (wave tile 128 x 64, 8 independent accumulators), no
//                  vector-memory instruction in the k-loop;
//       waves 4-5  activations: fp32 panel -> registers (six tiles in flight: HBM latency) -> exact 3-way split -> LDS;
//       waves 6-7  weights: the pre-split planes global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR staging, no ds_write) from
//                  a TILED image (k_split3_tiles) whose 1-KB pieces are whole cache lines, swizzle baked in;
//     each role has its own vmcnt counter (a wave's vector-memory operations retire in order: a wait for an L2-fed DMA behind HBM
//     loads retires those too, and behind result stores it drains them), every counted operation is issued through asm;
//   * a ring of FOUR LDS slots of one 16-deep k-tile each (A 24 KB + B 15 KB): in iteration i the compute waves read slot i & 3, tile
//     i+1 is already published, tile i+2 is landing (waited for at the end of the iteration) and tile i+3 is being requested --
//     with two slots a tile had to be requested, land and be published inside ONE iteration, and the DMA's issue + latency
//     (~4,000 cycles) was longer than the 3,072 cycles of MFMAs it was supposed to hide behind; ONE barrier per k-tile;
//   * the workgroup loops over the 256-row tiles of the weight (T1: two, the mask conv: four): the memory waves run ahead into the
//     next row tile while the compute waves store the previous one's results.
// LDS: 163,840 B (everything the CU has): 4 x (24,576 + 15,360) + 4,096 (GroupNorm coefficients | bias); the epilogue's staging
// tiles live in slot 3 (the slot of the row tile's LAST k-tile).  One workgroup per CU, grid = panels (256 at the benchmark's size).
#ifndef FQSS_T2_ABL
#define FQSS_T2_ABL 0
#endif
// FQSS_T2_ABL (timing experiments only, tools/r04_abl.sh; default 0 = the product; results are garbage otherwise): bit 0 no MFMAs |
// 1 no fragment reads | 2 no split / LDS store of the activations | 3 no weight DMA | 4 no activation loads | 5 no result stores
constexpr int T2BM = 256, T2BN = 128, T2BK = 16;
constexpr int T2_A_SLOT = 3 * T2BM * T2BK * 2;            // 24,576: [plane][256 rows][16 k] bf16, 32-B rows, swizzled chunks
constexpr int T2_B_SLOT = 3 * T2BK * TLDN * 2;            // 15,360: [plane][16 k][128 + 32 pad] bf16 (the row layout of k_tgemm)
constexpr int T2_B_OFF = 4 * T2_A_SLOT;                   // 98,304
constexpr int T2_PCO_OFF = T2_B_OFF + 4 * T2_B_SLOT;      // 159,744
constexpr int T2_SMEM = T2_PCO_OFF + 2 * 512 * 4;         // 163,840
constexpr int T2_EPI_OFF = 3 * T2_A_SLOT;                 // epilogue staging (4 x 4,608 B) + red inside A slot 3
static_assert(4 * 32 * TLDT * 4 + 2 * 8 * 8 <= T2_A_SLOT, "epilogue scratch fits one A slot");

// 16-B load with a scalar base + a 32-bit lane offset: the loader's addresses for 32 k-tiles x 4 rows are scalar adds on ONE register
// (as 64-bit per-lane pointers the compiler precomputed and spilled them: a scratch reload in the counted stream ends in a vmcnt(0))
__device__ __forceinline__ void t_load16s(f32x4t& d, const void* sbase, unsigned voff) {
    // s_nop 4: the compiler may have produced the scalar base by a VALU instruction (v_readlane of a spilled SGPR) right in front of
    // the statement; a vector-memory instruction that reads such an SGPR needs five wait states, and nothing pads the inside of an asm
    // statement (without it: wrong tiles and one memory fault)
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(voff), "s"(sbase) : "memory");
}
// LDS-DMA of 16 B per lane with a scalar base: source = sbase (wave-uniform, SGPR pair) + voff (per lane, bytes), LDS destination =
// lds_dst (wave-uniform byte address, through M0) + 16 * lane; M0 is written in the statement that reads it (the compiler reserves it
// and does not preserve it around asm).  The pieces a wave moves are 1 KB apart: ONE lane-offset register, scalar adds.
__device__ __forceinline__ void t2_dma16s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

template <int T, int N, class F>
__device__ __forceinline__ void t2_unroll(F& f) {      // f(integral_constant<T>) ... f(integral_constant<N - 1>), straight-line
    if constexpr (T < N) {
        f(std::integral_constant<int, T>{});
        t2_unroll<T + 1, N>(f);
    }
}

// every barrier of k_tgemm2 goes through T2_SYNC(<s_waitcnt in front of it, or "">).  FQSS_T2_STAMP (diagnostic builds, tools/t2_stamps.py):
// workgroup 0 records, per wave and barrier, the cycle counter before the wait, before the barrier and behind it.
#ifdef FQSS_T2_STAMP
__device__ unsigned long long g_t2_stamp[8][160][3];
#define T2_SYNC(W)                                                                                       \
    do {                                                                                                 \
        const bool st_on = blockIdx.x == 0 && lane == 0 && sidx < 160;                                   \
        if (st_on) g_t2_stamp[wave][sidx][0] = __builtin_amdgcn_s_memtime();                             \
        asm volatile(W ::: "memory");                                                                    \
        if (st_on) g_t2_stamp[wave][sidx][1] = __builtin_amdgcn_s_memtime();                             \
        asm volatile("s_barrier" ::: "memory");                                                          \
        if (st_on) g_t2_stamp[wave][sidx][2] = __builtin_amdgcn_s_memtime();                             \
        ++sidx;                                                                                          \
    } while (0)
#else
#define T2_SYNC(W) asm volatile(W "\n\ts_barrier" ::: "memory")
#endif

template <int PRO>
__global__ __launch_bounds__(512, 2) void k_tgemm2(TGemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];
    const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem2);
    float* pco = reinterpret_cast<float*>(smem2 + T2_PCO_OFF);                         // [2][512]
    double* red = reinterpret_cast<double*>(smem2 + T2_EPI_OFF + 4 * 32 * TLDT * 4);   // [2 * 8]
    float* pms = reinterpret_cast<float*>(red);                                        // [2] (slot 3: no tile lands there before barrier P)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int panel = blockIdx.x;
    const int b = panel / g.tiles_n, j0 = (panel % g.tiles_n) * T2BN;
    // Phase stagger (round 5): the 256 workgroups (one per CU) run in lockstep -- k-loop (matrix pipes busy, HBM nearly idle), then the
    // epilogue (every workgroup storing its 128 KB tile at once: HBM-write bound, matrix pipes dark; 44 % of T1 by its barrier stamps).
    // Starting every second workgroup late by roughly half an epilogue puts one half's stores under the other half's MFMAs.
    if (g.stagger > 0 && (panel & 1)) {
        const unsigned long long t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < (unsigned long long)g.stagger) __builtin_amdgcn_s_sleep(16);
    }
#ifdef FQSS_T2_STAMP
    int sidx = 0;
#endif

    if (PRO == 1) {
        t_stats_finalize(g.pro_stats + (int64_t)b * kTSlots * kTSlotStride, g.pro_count, g.pro_eps, pms);
        const float pmean = pms[0], prstd = pms[1];
        for (int k = tid; k < g.K; k += 512) {
            const float pa = prstd * g.pro_gamma[k];
            pco[k] = pa;
            pco[512 + k] = fmaf(-pa, pmean, g.pro_beta[k]);
        }
    }
    // Without a GroupNorm prologue the coefficient table is free: the bias vector lives there (<= 1024 rows), so that the row tiles
    // after the first start without a global load -- a load behind the previous tile's result stores would have to wait for them.
    const bool bias_in_lds = (PRO != 1) && g.M <= 1024;
    if (bias_in_lds)
        for (int i = tid; i < g.M; i += 512) pco[i] = g.bias != nullptr ? g.bias[i] : 0.0f;
    T2_SYNC("s_waitcnt vmcnt(0) lgkmcnt(0)");   // pms / pco written; nothing of the compiler's in flight
    const int nkt = g.K / T2BK;          // a multiple of 8 (host: K % 128 == 0)
    const int n_last = (g.N - 1) & ~3;
    float s1 = 0.0f, s2 = 0.0f;
    const bool want_stats = g.stats_out != nullptr;

    // Barriers of one row tile, the same count in every role: P (tiles 0 and 1 published), one per iteration i = 0 .. nkt - 2 (tile i + 2
    // published), L (behind the last tile's fragment reads: the staging tiles may be written).
    if (wave >= 6) {
        // =============================== weight waves (2): LDS-DMA of the weight tiles ==============================================
        // 12 pieces of 1 KB per wave and k-tile, source and destination lane-linear.  In iteration i tile i + 3 is requested (its slot
        // was read in iteration i - 1) and tile i + 2 -- requested one iteration earlier -- is waited for (the 12 younger requests
        // stay in flight) in front of the barrier that publishes it.
        const int dw = wave - 6;
        const unsigned voff = lane * 16;
#ifdef FQSS_T2_MPRIO
        __builtin_amdgcn_s_setprio(FQSS_T2_MPRIO);
#endif
        auto dma_tile = [&](int64_t tile) {                   // tile = row tile * nkt + k-tile -> slot k-tile & 3
            if (FQSS_T2_ABL & 8) return;
            const unsigned char* src = reinterpret_cast<const unsigned char*>(g.A) + tile * (3 * 4096 * 2) + dw * (12 * 1024);
            const unsigned dst = lds_base + (unsigned)(tile & 3) * T2_A_SLOT + dw * (12 * 1024);      // nkt % 4 == 0: tile & 3 == k-tile & 3
#pragma unroll
            for (int j = 0; j < 12; ++j) t2_dma16s(src + j * 1024, voff, dst + j * 1024);
        };
        for (int mt = 0; mt < g.tiles_m; ++mt) {
            const int64_t t0 = (int64_t)mt * nkt;
            // tiles 0 .. 2 into slots 0 .. 2: the compute waves may still be in the previous row tile's epilogue (staging in slot 3)
            dma_tile(t0); dma_tile(t0 + 1); dma_tile(t0 + 2);
            T2_SYNC("s_waitcnt vmcnt(12)");             // P: tiles 0, 1
            for (int i = 0; i < nkt - 3; ++i) {
                dma_tile(t0 + i + 3);
                T2_SYNC("s_waitcnt vmcnt(12)");         // tile i + 2
            }
            T2_SYNC("s_waitcnt vmcnt(0)");              // i = nkt - 3: tile nkt - 1
            T2_SYNC("");                                   // i = nkt - 2
            T2_SYNC("");                                   // L
        }
    } else if (wave >= 4) {
        // =============================== activation waves (2): the fp32 panel -> three bf16 planes in LDS ===========================
        // Per k-tile and wave: 4 x 16 B of activations per lane, the exact 3-way split of 16 values, 12 ds_write_b64.  The loads come
        // from HBM (microseconds under load) and a tile is 8 KB per workgroup, so SIX tiles are kept in flight in a register ring
        // (48 KB per CU; with one or two 16-KB tiles in flight the memory skeleton alone ran at 9 GB/s per CU).
        const int lt = tid - 256;
#ifdef FQSS_T2_MPRIO
        __builtin_amdgcn_s_setprio(FQSS_T2_MPRIO);
#endif
        float pslope = (PRO == 2) ? *g.pro_slope : 0.0f;
        if (PRO == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(pslope));
        // thread -> k-rows (lt >> 5) + 4 q (q < 4), columns 4 (lt & 31) .. + 4: a wave instruction reads two whole 512-B row segments
        // (clamped into the row: columns past N are never stored)
        const int bk_row = lt >> 5, bk_c = (lt & 31) * 4;
        const unsigned bvoff = (unsigned)((bk_row * (int)g.ldb + min(j0 + bk_c, n_last)) * 4);     // < 2^32: ld_x < 2^28 (host)
        const float* bbase = g.B + (int64_t)b * g.sBb;                                          // wave-uniform
        const int64_t bstep = (int64_t)T2BK * g.ldb, brow4 = 4 * g.ldb;
        struct LStage { f32x4t rb[4]; };
        auto load_b = [&](LStage& st, int t) {
            if (FQSS_T2_ABL & 16) return;
#pragma unroll
            for (int q = 0; q < 4; ++q) t_load16s(st.rb[q], bbase + t * bstep + q * brow4, bvoff);
        };
        auto store_b = [&](LStage& st, int t) {
            if (FQSS_T2_ABL & 4) return;
            unsigned char* bs = smem2 + T2_B_OFF + (t & 3) * T2_B_SLOT + (bk_row * TLDN + bk_c) * 2;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int kc = t * T2BK + bk_row + 4 * q;
                const float p_a = (PRO == 1) ? pco[kc] : 1.f, p_b = (PRO == 1) ? pco[512 + kc] : 0.f;
                float h0[4], r1[4], r2[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = st.rb[q][e];
                    if (PRO == 1) v = fmaf(v, p_a, p_b);
                    else if (PRO == 2) v = v > 0.f ? v : pslope * v;
                    h0[e] = v;
                    r1[e] = v - t_tr(v);
                    r2[e] = r1[e] - t_tr(r1[e]);
                }
                uint2 o1, o2, o3;
                o1.x = __builtin_amdgcn_perm(__float_as_uint(h0[1]), __float_as_uint(h0[0]), 0x07060302u);
                o1.y = __builtin_amdgcn_perm(__float_as_uint(h0[3]), __float_as_uint(h0[2]), 0x07060302u);
                o2.x = __builtin_amdgcn_perm(__float_as_uint(r1[1]), __float_as_uint(r1[0]), 0x07060302u);
                o2.y = __builtin_amdgcn_perm(__float_as_uint(r1[3]), __float_as_uint(r1[2]), 0x07060302u);
                o3.x = __builtin_amdgcn_perm(__float_as_uint(r2[1]), __float_as_uint(r2[0]), 0x07060302u);
                o3.y = __builtin_amdgcn_perm(__float_as_uint(r2[3]), __float_as_uint(r2[2]), 0x07060302u);
                *reinterpret_cast<uint2*>(bs + q * 4 * TLDN * 2) = o1;
                *reinterpret_cast<uint2*>(bs + q * 4 * TLDN * 2 + T2BK * TLDN * 2) = o2;
                *reinterpret_cast<uint2*>(bs + q * 4 * TLDN * 2 + 2 * T2BK * TLDN * 2) = o3;
            }
        };
        // One step = tile T: wait for it (the requests of the up to five younger tiles stay in flight), split it into LDS slot T & 3,
        // re-arm its registers with tile T + 6.  The whole k-loop is STRAIGHT-LINE code, generated per k-tile count: a register that an
        // asm load is still filling must never meet a control-flow merge -- the copy the register allocator may place there reads it
        // before the data has landed (a run-time loop / branch around these steps returned wrong tiles for K = 128).
        auto row_tile = [&](auto NKT) {
            constexpr int nk = decltype(NKT)::value;
            LStage R0, R1, R2, R3, R4, R5;       // tile t travels in R[t % 6]
            load_b(R0, 0); load_b(R1, 1); load_b(R2, 2); load_b(R3, 3); load_b(R4, 4); load_b(R5, 5);
            auto put = [&](auto T) {
                constexpr int t = decltype(T)::value;
                constexpr int younger = (nk - 1 - t) < 5 ? (nk - 1 - t) : 5;
                LStage& st = (t % 6) == 0 ? R0 : (t % 6) == 1 ? R1 : (t % 6) == 2 ? R2 : (t % 6) == 3 ? R3 : (t % 6) == 4 ? R4 : R5;
                asm volatile("s_waitcnt vmcnt(%4)" : "+v"(st.rb[0]), "+v"(st.rb[1]), "+v"(st.rb[2]), "+v"(st.rb[3]) : "n"(4 * younger) : "memory");
                store_b(st, t);
                if (t + 6 <= nk - 1) load_b(st, t + 6);
            };
            // tiles 0 and 1 into slots 0 and 1 (the compute waves may still be in the previous row tile's epilogue: staging in slot 3)
            put(std::integral_constant<int, 0>{});
            put(std::integral_constant<int, 1>{});
            T2_SYNC("s_waitcnt lgkmcnt(0)");            // P
            auto step = [&](auto I) {                          // iteration i: tile i + 2 into slot (i + 2) & 3, read last in iteration i - 2
                constexpr int i = decltype(I)::value;
                if constexpr (i + 2 <= nk - 1) put(std::integral_constant<int, i + 2>{});
                T2_SYNC("s_waitcnt lgkmcnt(0)");
            };
            t2_unroll<0, nk - 1>(step);                        // iterations 0 .. nkt - 2
            T2_SYNC("");           // L
        };
        for (int mt = 0; mt < g.tiles_m; ++mt) {
            if (nkt == 8) row_tile(std::integral_constant<int, 8>{});
            else if (nkt == 16) row_tile(std::integral_constant<int, 16>{});
            else if (nkt == 24) row_tile(std::integral_constant<int, 24>{});
            else row_tile(std::integral_constant<int, 32>{});
        }
    } else {
        // =============================== compute waves (4 = 2 x 2, wave tile 128 x 64): fragment reads and MFMAs ====================
        const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
        const int gq = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
        const float nscale = (g.act == FQSS_ACT_PRELU) ? *g.slope : (g.act == FQSS_ACT_RELU ? 0.0f : 1.0f);
        const bool has_bias = g.bias != nullptr;
#ifdef FQSS_T2_PRIO
        __builtin_amdgcn_s_setprio(FQSS_T2_PRIO);
#endif
        f32x16 acc[4][2];
        // fragments of one k-tile: the three planes of the wave's 64 columns (B, 6 fragments) and of 64 of its 128 rows (A, 6 fragments
        // per half): 24 MFMAs per half.  12 + 6 fragments of the NEXT tile are requested before the barrier that ends the current one
        // (that tile was published one barrier earlier), so their LDS latency and the barrier overlap (cycle stamps, profiles/
        // r04_t2_stamps_t3.txt: read-after-barrier cost 620 of 2,160 cycles per tile).
        auto rd_b = [&](bf16x8 (&bfr)[3][2], int slot) {
            typedef unsigned short (*BsT)[T2BK][TLDN];
            BsT Bs = reinterpret_cast<BsT>(smem2 + T2_B_OFF + slot * T2_B_SLOT);
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int kr = 8 * (gq >> 1) + tq;
                    const int nc = wn * 64 + ni * 32 + 16 * (gq & 1) + 4 * tp;
                    union { bf16x8 v; s16x4 h[2]; } u;
                    if (FQSS_T2_ABL & 2) { bfr[p][ni] = bf16x8{}; asm volatile("" : "+v"(bfr[p][ni])); continue; }
                    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(&Bs[p][kr][nc]));
                    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(&Bs[p][kr + 4][nc]));
                    bfr[p][ni] = u.v;
                }
        };
        auto rd_a = [&](bf16x8 (&af)[3][2], int slot, int mh) {
            const unsigned char* as = smem2 + slot * T2_A_SLOT;
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const int row = wm * 128 + (2 * mh + mi) * 32 + lr;
                    if (FQSS_T2_ABL & 2) { af[p][mi] = bf16x8{}; asm volatile("" : "+v"(af[p][mi])); continue; }
                    af[p][mi] = *reinterpret_cast<const bf16x8*>(as + p * (T2BM * T2BK * 2) + row * 32 + ((lh ^ ((row >> 3) & 1)) << 4));
                }
        };
        auto mm = [&](bf16x8 (&af)[3][2], bf16x8 (&bfr)[3][2], auto MH) {
            constexpr int mh = decltype(MH)::value;
            constexpr int IA[6] = {2, 0, 1, 1, 0, 0}, IB[6] = {0, 2, 1, 0, 1, 0};     // the six products, smallest pieces first (k_tgemm)
#pragma unroll
            for (int sp = 0; sp < 6; ++sp)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        if (!(FQSS_T2_ABL & 1))
                            acc[2 * mh + mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[IA[sp]][mi], bfr[IB[sp]][ni], acc[2 * mh + mi][ni], 0, 0, 0);
                        else asm volatile("" : "+v"(acc[2 * mh + mi][ni]) : "v"(af[IA[sp]][mi]), "v"(bfr[IB[sp]][ni]));
        };
        using H0 = std::integral_constant<int, 0>;
        using H1 = std::integral_constant<int, 1>;
        float(*Tt)[TLDT] = reinterpret_cast<float(*)[TLDT]>(smem2 + T2_EPI_OFF + wave * 32 * TLDT * 4);

        for (int mt = 0; mt < g.tiles_m; ++mt) {
            const int i0 = mt * T2BM;
            const int rowt0 = i0 + wm * 128;
            // The accumulators START at the bias (one fp32 add per element less in the epilogue; the sum is bias + products instead of
            // products + bias: a different, equally valid fp32 rounding order).
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // rows (r & 3) + 8 (r >> 2) + 4 lh of the 32-row tile; plain loads: the compiler waits for them itself
                    const int br = rowt0 + mi * 32 + 8 * q + 4 * lh;
                    const float4 bv = bias_in_lds ? *reinterpret_cast<const float4*>(pco + br)
                                                  : (has_bias ? *reinterpret_cast<const float4*>(g.bias + br) : make_float4(0.f, 0.f, 0.f, 0.f));
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        acc[mi][ni][4 * q + 0] = bv.x;
                        acc[mi][ni][4 * q + 1] = bv.y;
                        acc[mi][ni][4 * q + 2] = bv.z;
                        acc[mi][ni][4 * q + 3] = bv.w;
                    }
                }
            }
            // P: tiles 0 and 1 are published.  No vmcnt wait here: this wave's result stores of the previous row tile stay in flight
            // under the next tile's MFMAs.
            T2_SYNC("s_waitcnt lgkmcnt(0)");
            bf16x8 bA[3][2], bB[3][2], a0[3][2], a1[3][2];
            rd_b(bA, 0);
            rd_a(a0, 0, 0);
            for (int i = 0; i < nkt; i += 2) {
                // tile i (slot i & 3; B in bA, the first 64 rows in a0); the barrier behind it publishes tile i + 2.  No lgkmcnt wait in
                // front of it: every read of slot i & 3 has been consumed by an issued MFMA, the reads in flight target slot (i + 1) & 3.
                rd_a(a1, i & 3, 1);
                mm(a0, bA, H0{});
                mm(a1, bA, H1{});
                rd_b(bB, (i + 1) & 3);
                rd_a(a0, (i + 1) & 3, 0);
                T2_SYNC("");
                // tile i + 1 (B in bB); behind the row tile's last k-tile the barrier is L: every wave is done reading slot 3 before the
                // staging tiles (which live there) are written
                rd_a(a1, (i + 1) & 3, 1);
                mm(a0, bB, H0{});
                mm(a1, bB, H1{});
                if (i + 2 < nkt) {
                    rd_b(bA, (i + 2) & 3);
                    rd_a(a0, (i + 2) & 3, 0);
                }
                T2_SYNC("s_waitcnt lgkmcnt(0)");
            }

            // ---- the epilogue's per-lane addresses hang off `el`, a copy of the lane id the optimiser cannot see through: otherwise it
            // hoists ~60 registers of loop-invariant store / residual addresses above the k-loop
            int el = lane;
            asm volatile("" : "+v"(el));
            const int e_c4 = (el & 7) * 4, e_r8 = el >> 3, e_lr = el & 31, e_lh = el >> 5;
            // act (+ residual), 16-B/lane row stores through a wave-private LDS tile (k_tgemm's epilogue, 4 waves x 8 tiles).  The
            // residual rows are PLAIN loads: an asm load's destination may be spilled by the compiler before the data has landed
            // (it was, with the residuals requested in front of the last tile's MFMAs), and nothing here is counted by hand.
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int rowt = rowt0 + mi * 32;                 // M1 % 32 == 0 (checked by the host): one side of the split per 32-row tile
                const bool first = rowt < g.M1;
                const int64_t ldc = first ? g.ldc1 : g.ldc2;
                const int64_t off = (int64_t)b * (first ? g.sC1b : g.sC2b) + (int64_t)(first ? rowt : rowt - g.M1) * ldc;
                float* Cb = (first ? g.C1 : g.C2) + off;
                const float* Rsel = first ? g.R1 : g.R2;
                const bool hres = Rsel != nullptr;
                float4 res[2][4];
                if (hres) {
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int pass = 0; pass < 4; ++pass)
                            res[ni][pass] = *reinterpret_cast<const float4*>(Rsel + off + (int64_t)(pass * 8 + e_r8) * ldc + min(j0 + wn * 64 + ni * 32 + e_c4, n_last));
                }
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = (r & 3) + 8 * (r >> 2) + 4 * e_lh;
                        const float v = acc[mi][ni][r];
                        Tt[rl][e_lr] = v > 0.0f ? v : nscale * v;
                    }
                    const int col = j0 + wn * 64 + ni * 32 + e_c4;
#pragma unroll
                    for (int pass = 0; pass < 4; ++pass) {
                        const int rl = pass * 8 + e_r8;
                        float4 t = *reinterpret_cast<const float4*>(&Tt[rl][e_c4]);
                        if (hres) {
                            const float4 q = res[ni][pass];
                            t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
                        }
                        if (col < g.N) {
                            if (!(FQSS_T2_ABL & 32)) store16(Cb + (int64_t)rl * ldc + col, t);
                            else asm volatile("" : : "v"(t.x), "v"(t.y), "v"(t.z), "v"(t.w));
                            if (want_stats) {
                                const float v[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    if (col + e < g.N) {
                                        s1 += v[e];
                                        s2 = fmaf(v[e], v[e], s2);
                                    }
                            }
                        }
                    }
                }
            }
        }
    }
    if (want_stats) {
        // (red lives in A slot 3 next to the staging tiles: a barrier separates the last wave's staging reads from it; the memory
        // waves contribute zeros)
        T2_SYNC("s_waitcnt lgkmcnt(0)");
        double v[2] = {(double)s1, (double)s2};
        block_sum<double, 2>(v, red);
        if (tid == 0) {
            double* so = g.stats_out + ((int64_t)b * kTSlots + (panel & (kTSlots - 1))) * kTSlotStride;
            atomicAdd(&so[0], v[0]);
            atomicAdd(&so[1], v[1]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_tgemm_k128 (round 5): the teacher's T1 -- 1x1 conv 128 -> 512 + bias + PReLU + GroupNorm statistics -- with
// the WEIGHTS IN REGISTERS.  K = 128 makes a wave's whole weight slab small: 32 rows x 128 k x 3 bf16 planes = 24 KB = 96 VGPRs per
// lane in MFMA A-fragment order, loaded once per workgroup from the tiled image of fqss_split3_tiles.  What is left per 64-column
// tile is the activation side only -- 32 KB of fp32 in, split exactly in three into 55 KB of LDS, 96 MFMAs per wave straight through
// the whole K (no k-loop pipeline, no ring), a 128 x 64 result out -- so a workgroup needs 74 KB of LDS and 256 registers: TWO fit
// a CU, and one's loads / split / result stores run under the other's MFMAs, which is the overlap k_tgemm2's barrier stamps asked for
// (its epilogues were 44 % of T1 with the matrix pipes dark: one workgroup per CU, all in the same phase).
// Grid: (column groups) x (Co / 128 row blocks); a workgroup walks the column tiles group, group + G, ... of ITS row block; the four
// row blocks of a column group sit on one XCD (xcd_tile), so an activation tile comes from HBM once.  Same six exact partial
// products, smallest first, as k_tgemm / k_tgemm2; the accumulation runs over k in the same order.
// ---------------------------------------------------------------------------------------------------------------------------
// FQSS_K1_ABL (timing experiments only: make variant SRC=teacher NAME=.. DEFS=-DFQSS_K1_ABL=n, tools/r05_t1_probe.py; never the product):
// 1 no MFMAs | 2 no split / LDS stores of the activation tile | 4 no result stores | 8 no activation loads | 16 no LDS fragment reads
#ifndef FQSS_K1_ABL
#define FQSS_K1_ABL 0
#endif
constexpr int K1_BN = 64, K1_LDN = 72, K1_ROWS = 128;
struct K1X { f32x4t v[8]; };
__device__ __forceinline__ void k1_wait(K1X& x) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(x.v[0]), "+v"(x.v[1]), "+v"(x.v[2]), "+v"(x.v[3]), "+v"(x.v[4]), "+v"(x.v[5]), "+v"(x.v[6]), "+v"(x.v[7]) : : "memory");
}

__global__ __launch_bounds__(256, 2) void k_tgemm_k128(TGemmArgs g, int ngroups, int tiles_n64) {
    __shared__ __attribute__((aligned(16))) unsigned short Bs[3][128][K1_LDN];    // 55,296 B: one 128 (k) x 64 (n) activation tile, three planes
    __shared__ __attribute__((aligned(16))) float Tst[4][32][TLDT];               // 18,432 B: a wave-private staging tile each
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5, gq = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    int group, rb;
    if (!xcd_tile(ngroups, g.M / K1_ROWS, group, rb)) return;
    const int i0 = rb * K1_ROWS + wave * 32;      // this wave's 32 output rows
    // ---- the wave's weight slab, in A-fragment order: lane (row lr, half lh) holds k = 16 s + 8 lh .. + 7 of every plane
    bf16x8 wf[8][3];
    {
        const int row = i0 + lr, mt = row >> 8, r256 = row & 255, pc = lh ^ ((r256 >> 3) & 1);
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                wf[s8][p] = *reinterpret_cast<const bf16x8*>(g.A + ((int64_t)(mt * 8 + s8) * 3 + p) * 4096 + r256 * 16 + pc * 8);
    }
    __shared__ float rowb[K1_ROWS];      // bias of the workgroup's rows (added in the epilogue: 16 registers fewer than accumulators that start at it)
    if (tid < K1_ROWS) rowb[tid] = (g.bias != nullptr) ? g.bias[rb * K1_ROWS + tid] : 0.0f;
    __syncthreads();
    const float nscale = (g.act == FQSS_ACT_PRELU) ? *g.slope : (g.act == FQSS_ACT_RELU ? 0.0f : 1.0f);
    const int n_last = (g.N - 1) & ~3;
    const int xk = tid >> 4, xc = (tid & 15) * 4;          // loader: 16 threads per k row, 16 rows per pass, 8 passes
    const int ntiles = g.batches * tiles_n64;
    float(*Tt)[TLDT] = Tst[wave];
    const int c4 = (lane & 7) * 4;
    K1X xr;
    auto load_x = [&](int ct) {
        const int b = ct / tiles_n64, j0 = (ct - b * tiles_n64) * K1_BN;
        const float* Bb = g.B + (int64_t)b * g.sBb + min(j0 + xc, n_last);
        if (FQSS_K1_ABL & 8) return;
#pragma unroll
        for (int q = 0; q < 8; ++q) t_load16(xr.v[q], Bb + (int64_t)(q * 16 + xk) * g.ldb);
    };
    float s1 = 0.0f, s2 = 0.0f;
    int sb = -1;                                   // the sample the statistics in (s1, s2) belong to
    auto flush_stats = [&]() {
        if (g.stats_out != nullptr && sb >= 0) {
            const double a = wave_sum((double)s1), q = wave_sum((double)s2);
            if (lane == 0) {
                double* so = g.stats_out + ((int64_t)sb * kTSlots + ((group * 4 + wave) & (kTSlots - 1))) * kTSlotStride;
                atomicAdd(&so[0], a);
                atomicAdd(&so[1], q);
            }
        }
        s1 = s2 = 0.0f;
    };
    int ct = group;
    if (ct < ntiles) load_x(ct);
    if (g.stagger > 0 && ((g.stagger & 1) ? ((blockIdx.x >> 3) & 1) : (blockIdx.x >= gridDim.x / 2))) {
        // experiment knob (FQSS_T1_STAGGER=<cycles>; odd: every second workgroup of an XCD, even: the second half of the grid): start late,
        // so that the two workgroups of a CU are in opposite phases (one's memory side under the other's MFMAs)
        const long long t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < g.stagger) __builtin_amdgcn_s_sleep(8);
    }
    for (; ct < ntiles; ct += ngroups) {
        const int b = ct / tiles_n64, j0 = (ct - b * tiles_n64) * K1_BN;
        if (b != sb) {
            flush_stats();
            sb = b;
        }
        // ---- this tile's activations: exact three-way split into the LDS tile (every wave has passed the barrier behind the
        // previous tile's fragment reads)
        k1_wait(xr);
#pragma unroll
        for (int q = 0; q < ((FQSS_K1_ABL & 2) ? 0 : 8); ++q) {
            const float x[4] = {xr.v[q][0], xr.v[q][1], xr.v[q][2], xr.v[q][3]};
            float r1[4], r2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                r1[e] = x[e] - t_tr(x[e]);
                r2[e] = r1[e] - t_tr(r1[e]);
            }
            uint2 o1, o2, o3;
            o1.x = __builtin_amdgcn_perm(__float_as_uint(x[1]), __float_as_uint(x[0]), 0x07060302u);
            o1.y = __builtin_amdgcn_perm(__float_as_uint(x[3]), __float_as_uint(x[2]), 0x07060302u);
            o2.x = __builtin_amdgcn_perm(__float_as_uint(r1[1]), __float_as_uint(r1[0]), 0x07060302u);
            o2.y = __builtin_amdgcn_perm(__float_as_uint(r1[3]), __float_as_uint(r1[2]), 0x07060302u);
            o3.x = __builtin_amdgcn_perm(__float_as_uint(r2[1]), __float_as_uint(r2[0]), 0x07060302u);
            o3.y = __builtin_amdgcn_perm(__float_as_uint(r2[3]), __float_as_uint(r2[2]), 0x07060302u);
            *reinterpret_cast<uint2*>(&Bs[0][q * 16 + xk][xc]) = o1;
            *reinterpret_cast<uint2*>(&Bs[1][q * 16 + xk][xc]) = o2;
            *reinterpret_cast<uint2*>(&Bs[2][q * 16 + xk][xc]) = o3;
        }
        if (ct + ngroups < ntiles) load_x(ct + ngroups);     // the next tile's activations travel under this tile's MFMAs
        __syncthreads();
        // ---- 8 x 16 deep: 12 MFMAs per step (six partial products x two 32-column tiles), A from registers
        f32x16 acc[2];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][r] = 0.0f;
        // (the fragments of step ks + 1 are requested before the MFMAs of step ks: with two waves per SIMD the LDS round trip was not
        //  hidden otherwise -- ablation, tools/r05_t1_probe.py: 36 us per launch, 26 with the fragment reads removed)
        // Two fragment buffers of one 32-column tile and one k-step each (3 planes x 4 registers): while the six MFMAs of one run, the
        // fragments of the next are on their way -- 24 registers, what the un-pipelined form held for both column tiles of a step (the
        // kernel sits at the 256-register limit of two waves per SIMD, and a spill next to the asm loads in flight is not an option).
        auto read_frags = [&](bf16x8 (&bfr)[3], int ks, int ni) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const int kr = ks * 16 + 8 * (gq >> 1) + tq;
                const int nc = ni * 32 + 16 * (gq & 1) + 4 * tp;
                union { bf16x8 v; s16x4 h[2]; } u;
                if (FQSS_K1_ABL & 16) {
                    u.v = wf[ks][p];
                } else {
                    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(&Bs[p][kr][nc]));
                    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(&Bs[p][kr + 4][nc]));
                }
                bfr[p] = u.v;
            }
        };
        auto mfma_tile = [&](const bf16x8 (&bfr)[3], int ks, int ni) {
            constexpr int IA[6] = {2, 0, 1, 1, 0, 0}, IB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int sp = 0; sp < 6; ++sp) {
                if (!(FQSS_K1_ABL & 1)) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][IA[sp]], bfr[IB[sp]], acc[ni], 0, 0, 0);
                else acc[ni][sp] += (float)wf[ks][IA[sp]][0] + (float)bfr[IB[sp]][0];
            }
        };
        bf16x8 fr[2][3];
        read_frags(fr[0], 0, 0);
#pragma unroll
        for (int u = 0; u < 16; ++u) {          // u = 8 ni + ks: one 32-column tile through the whole K, then the other
            if (u + 1 < 16) read_frags(fr[(u + 1) & 1], (u + 1) & 7, (u + 1) >> 3);
            mfma_tile(fr[u & 1], u & 7, u >> 3);
        }
        __syncthreads();      // every wave is done with the LDS tile: the next iteration may overwrite it
        // ---- act, 16-B/lane row stores through the wave's staging tile, statistics (no workgroup barrier in here)
        float* Cb = g.C1 + (int64_t)b * g.sC1b + (int64_t)i0 * g.ldc1;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float v = acc[ni][r] + rowb[wave * 32 + rl];
                Tt[rl][lr] = v > 0.0f ? v : nscale * v;
            }
            const int col = j0 + ni * 32 + c4;
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int rl = pass * 8 + (lane >> 3);
                const float4 t = *reinterpret_cast<const float4*>(&Tt[rl][c4]);
                if (col < g.N) {
                    if (!(FQSS_K1_ABL & 4)) store16(Cb + (int64_t)rl * g.ldc1 + col, t);
                    const float v[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e < g.N) {
                            s1 += v[e];
                            s2 = fmaf(v[e], v[e], s2);
                        }
                }
            }
        }
    }
    flush_stats();
}

// T2: y = PReLU( dwconv( GN(x) ) + bias ), statistics of y.  One workgroup per (sample, channel) row,
// 16 outputs per thread; the GroupNorm coefficients are computed once per workgroup.
__global__ __launch_bounds__(256) void k_tdw(const float* __restrict__ x, const double* __restrict__ stats_in,
                                              const float* __restrict__ gamma, const float* __restrict__ beta, double count,
                                              float eps, const float* __restrict__ w, const float* __restrict__ bias,
                                              const float* slope_p, float* __restrict__ y, double* stats_out, int rows, int C,
                                              int M, int K, int dil, int pad, int ld_x, int ld_y) {
    __shared__ double red[2 * 4];
    __shared__ float coef[2];
    const float slope = *slope_p;
    const int row = blockIdx.x;
    const int bsmp = row / C, c = row - bsmp * C;
    __shared__ float ms[2];
    t_stats_finalize(stats_in + (int64_t)bsmp * kTSlots * kTSlotStride, count, eps, ms);
    if (threadIdx.x == 0) {
        const float ga0 = ms[1] * gamma[c];
        coef[0] = ga0;
        coef[1] = fmaf(-ga0, ms[0], beta[c]);
    }
    __syncthreads();
    const float ga = coef[0], gb = coef[1];
    float wk[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) wk[k] = (k < K) ? w[c * K + k] : 0.0f;
    const float bv = bias ? bias[c] : 0.0f;
    const float* xr = x + (int64_t)row * ld_x;
    float* yr = y + (int64_t)row * ld_y;
    float s1 = 0.0f, s2 = 0.0f;
    // The normalised row goes through LDS: every global load of the row is issued up front (consecutive lanes own
    // consecutive float4), the taps then read LDS -- no exposed HBM round trip per group, no unaligned global loads
    // for dilations 1 and 2.  Positions >= M hold zeros (the conv's zero padding applies AFTER the norm).
    extern __shared__ __attribute__((aligned(16))) float sx[];   // [ceil4(M)]
    for (int m0 = threadIdx.x * 4; m0 < M; m0 += 4096) {
        float4 q[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (m0 + 1024 * i < M) q[i] = *reinterpret_cast<const float4*>(xr + m0 + 1024 * i);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + 1024 * i;
            if (m >= M) break;
            const float v[4] = {q[i].x, q[i].y, q[i].z, q[i].w};
            float t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = (m + j < M) ? fmaf(v[j], ga, gb) : 0.0f;
            *reinterpret_cast<float4*>(&sx[m]) = make_float4(t[0], t[1], t[2], t[3]);
        }
    }
    __syncthreads();
    const bool aligned = ((dil & 3) == 0) && ((pad & 3) == 0);
    for (int m = threadIdx.x * 4; m < M; m += 1024) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k < K) {
                const int s0 = m + k * dil - pad;
                if (aligned) {
                    if (s0 >= 0 && s0 < M) {
                        const float4 t = *reinterpret_cast<const float4*>(&sx[s0]);
                        acc[0] = fmaf(wk[k], t.x, acc[0]);
                        acc[1] = fmaf(wk[k], t.y, acc[1]);
                        acc[2] = fmaf(wk[k], t.z, acc[2]);
                        acc[3] = fmaf(wk[k], t.w, acc[3]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[j] = fmaf(wk[k], 0.0f, acc[j]);   // same op sequence as a padded tap
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int sj = s0 + j;
                        acc[j] = fmaf(wk[k], (sj >= 0 && sj < M) ? sx[sj] : 0.0f, acc[j]);
                    }
                }
            }
        }
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t = acc[j] + bv;
            o[j] = t > 0.f ? t : slope * t;
            if (m + j < M) {
                s1 += o[j];
                s2 = fmaf(o[j], o[j], s2);
            }
        }
        *reinterpret_cast<float4*>(yr + m) = make_float4(o[0], o[1], o[2], o[3]);
    }
    double v[2] = {(double)s1, (double)s2};
    block_sum<double, 2>(v, red);
    if (threadIdx.x == 0) {
        double* so = stats_out + ((int64_t)bsmp * kTSlots + (c & (kTSlots - 1))) * kTSlotStride;
        atomicAdd(&so[0], v[0]);
        atomicAdd(&so[1], v[1]);
    }
}

// statistics only (sum, sum^2 per sample), atomically added into the slot partials ws[b][kTSlots][kTSlotStride]
__global__ __launch_bounds__(256) void k_tstats(const float* __restrict__ x, int C, int M, int64_t ld, double* ws) {
    __shared__ double red[2 * 4];
    const int b = blockIdx.y;
    double s = 0.0, ss = 0.0;
    for (int c = blockIdx.x; c < C; c += gridDim.x) {
        const float* xr = x + ((int64_t)b * C + c) * ld;
        for (int m = threadIdx.x * 4; m < M; m += 256 * 4) {
            const float4 t = *reinterpret_cast<const float4*>(xr + m);
            const float v[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (m + j < M) {
                    s += (double)v[j];
                    ss += (double)v[j] * (double)v[j];
                }
        }
    }
    double v[2] = {s, ss};
    block_sum<double, 2>(v, red);
    if (threadIdx.x == 0) {
        double* so = ws + ((int64_t)b * kTSlots + (blockIdx.x & (kTSlots - 1))) * kTSlotStride;
        atomicAdd(&so[0], v[0]);
        atomicAdd(&so[1], v[1]);
    }
}

}  // namespace fqss

using namespace fqss;

extern "C" int fqss_split3_planes(const float* w, uint16_t* planes, int64_t n, fqss_stream_t stream) {
    FQSS_REQUIRE(w && planes && n >= 0, "bad args");
    if (n == 0) return FQSS_OK;
    int64_t nb = cdiv(n, 256);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(k_split3_planes, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, w, planes, n);
    return launch_status("fqss_split3_planes");
}

// tiled = false: `planes` is the [3][Co][Ci] image of fqss_split3_planes (k_tgemm, any shape); tiled = true: the image of
// fqss_split3_tiles (k_tgemm2, whole 256-row tiles only)
static int tgemm_launch(bool tiled, const char* fn, const uint16_t* planes, const float* x, int B, int Ci, int Co, int M, int64_t ld_x, int pro,
                        const double* pro_stats, const float* pro_gamma, const float* pro_beta, float pro_eps, const float* pro_slope,
                        const float* bias, int act, const float* slope, double* stats_out, int M1, float* c1, const float* r1, int64_t ld_c1,
                        float* c2, const float* r2, int64_t ld_c2, fqss_stream_t stream) {
#define TG_REQUIRE(cond, msg)                    \
    do {                                         \
        if (!(cond)) {                           \
            ::fqss::set_error("%s: %s", fn, msg); \
            return FQSS_EINVAL;                  \
        }                                        \
    } while (0)
    TG_REQUIRE(planes && x && c1, "null tensor");
    TG_REQUIRE(B >= 0 && Ci > 0 && Co > 0 && M >= 0 && ld_x >= ((M + 3) & ~3) && ld_x % 4 == 0 && aligned16(x), "bad input rows");
    TG_REQUIRE(Ci % 8 == 0 && aligned16(planes), "planes need Ci % 8 == 0 and 16-B alignment");
    TG_REQUIRE(M1 > 0 && M1 <= Co && (M1 == Co || c2), "bad output split");
    TG_REQUIRE(aligned16(c1) && ld_c1 % 4 == 0 && ld_c1 >= ((M + 3) & ~3) && (!c2 || (aligned16(c2) && ld_c2 % 4 == 0 && ld_c2 >= ((M + 3) & ~3))),
                 "output rows must be 16-B aligned");
    TG_REQUIRE((!r1 || aligned16(r1)) && (!r2 || aligned16(r2)), "residual rows must be 16-B aligned");
    TG_REQUIRE(pro >= 0 && pro <= 2 && (pro != 1 || (pro_stats && pro_gamma && pro_beta)) && (pro != 2 || pro_slope), "bad prologue");
    TG_REQUIRE(act != FQSS_ACT_PRELU || slope, "PReLU needs a slope");
    TG_REQUIRE(!tiled || ld_x < (1ll << 28), "rows too long for the tiled form's 32-bit lane offsets");
    TG_REQUIRE(!tiled || bias == nullptr || aligned16(bias), "the tiled form reads the bias in 16-B pieces");
    TG_REQUIRE(!tiled || fqss_tgemm_tiled_ok(Ci, Co, M1), "the tiled form needs Co % 256 == 0, Ci % 128 == 0, Ci <= 512, M1 % 32 == 0");
    if (B == 0 || M == 0) return FQSS_OK;
    TGemmArgs g{};
    g.A = planes; g.B = x; g.M = Co; g.N = M; g.K = Ci; g.ldb = ld_x; g.sBb = (int64_t)Ci * ld_x;
    g.pro = pro; g.pro_stats = pro_stats; g.pro_gamma = pro_gamma; g.pro_beta = pro_beta; g.pro_count = (double)Ci * (double)M;
    g.pro_eps = pro_eps; g.pro_slope = pro_slope;
    g.bias = bias; g.act = act; g.slope = slope; g.stats_out = stats_out; g.M1 = M1;
    g.C1 = c1; g.R1 = r1; g.ldc1 = ld_c1; g.sC1b = (int64_t)M1 * ld_c1;
    g.C2 = c2; g.R2 = r2; g.ldc2 = ld_c2; g.sC2b = (int64_t)(Co - M1) * ld_c2;
    g.tiles_n = (int)cdiv(M, TBN); g.tiles_m = (int)cdiv(Co, TBM); g.batches = B;
    if (tiled) {
        static const bool attr_ok = [] {
            bool ok = hipFuncSetAttribute((const void*)k_tgemm2<0>, hipFuncAttributeMaxDynamicSharedMemorySize, T2_SMEM) == hipSuccess;
            ok = ok && hipFuncSetAttribute((const void*)k_tgemm2<1>, hipFuncAttributeMaxDynamicSharedMemorySize, T2_SMEM) == hipSuccess;
            ok = ok && hipFuncSetAttribute((const void*)k_tgemm2<2>, hipFuncAttributeMaxDynamicSharedMemorySize, T2_SMEM) == hipSuccess;
            return ok;
        }();
        TG_REQUIRE(attr_ok, "k_tgemm2 needs 160 KB of dynamic LDS");
        static const bool k128 = [] { const char* e = getenv("FQSS_T1_K128"); return !(e && e[0] == '0'); }();
        if (k128 && pro == 0 && Ci == 128 && Co % K1_ROWS == 0 && M1 == Co && r1 == nullptr && r2 == nullptr && (stats_out == nullptr || B <= 64)) {
            // T1 of the TCN teacher: weights in registers, two workgroups per CU (k_tgemm_k128)
            const int tiles_n64 = (int)cdiv(M, K1_BN), ntiles = tiles_n64 * B;
            // 64 column groups x 4 row blocks = 256 workgroups = ONE per CU although two fit: alone on the chip the launch is slower
            // that way (45 us against 36 at 128 groups), but the step is faster (13.00 against 13.07 ms, FQSS_T1_GROUPS=64 / 128 interleaved
            // on one box; 48: 13.00, 32: 13.08, 16: 13.05, 192: 13.11) -- the teacher runs one batch ahead on its own stream with 10 ms
            // of slack, and what it should leave free is the second workgroup slot of every CU for the student's kernels
            static const int want_groups = [] { const char* e = getenv("FQSS_T1_GROUPS"); const int v = e ? atoi(e) : 64; return v >= 8 ? v / 8 * 8 : 64; }();
            int ngroups = want_groups;
            while (ngroups > 8 && ngroups * 2 > ntiles) ngroups >>= 1;
            static const int t1_stagger = [] { const char* e = getenv("FQSS_T1_STAGGER"); return e ? atoi(e) : 0; }();
            g.stagger = t1_stagger;
            hipLaunchKernelGGL(k_tgemm_k128, dim3(xcd_grid(ngroups, Co / K1_ROWS)), dim3(256), 0, (hipStream_t)stream, g, ngroups, tiles_n64);
            return launch_status(fn);
        }
        g.tiles_m = Co / T2BM;
        static const int stagger = [] { const char* e = getenv("FQSS_T2_STAGGER"); return e ? atoi(e) : 0; }();
        g.stagger = stagger;
        const dim3 grid2((unsigned)(g.tiles_n * B));
        if (pro == 0) hipLaunchKernelGGL(k_tgemm2<0>, grid2, dim3(512), T2_SMEM, (hipStream_t)stream, g);
        else if (pro == 1) hipLaunchKernelGGL(k_tgemm2<1>, grid2, dim3(512), T2_SMEM, (hipStream_t)stream, g);
        else hipLaunchKernelGGL(k_tgemm2<2>, grid2, dim3(512), T2_SMEM, (hipStream_t)stream, g);
        return launch_status(fn);
    }
    const dim3 grid(xcd_grid((int64_t)g.tiles_n * B, g.tiles_m));
    // FQSS_TGEMM_PAD_LDS=<bytes> (experiment knob, read once): unused dynamic LDS that caps the workgroups per CU -- with > 16 KB on top
    // of the 60-64 KB tile buffer only ONE teacher workgroup fits a CU, which leaves half of every SIMD's registers to the student's
    // waves when the teacher runs on its own stream beside the step (docs/history/DESIGN_rounds_1-5.md 9)
    static const size_t pad = [] {
        const char* e = getenv("FQSS_TGEMM_PAD_LDS");
        const size_t v = e ? (size_t)atol(e) : 0;
        if (v) {
            (void)hipFuncSetAttribute((const void*)k_tgemm<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v);
            (void)hipFuncSetAttribute((const void*)k_tgemm<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v);
            (void)hipFuncSetAttribute((const void*)k_tgemm<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v);
        }
        return v;
    }();
    if (pro == 0) hipLaunchKernelGGL(k_tgemm<0>, grid, dim3(256), pad, (hipStream_t)stream, g);
    else if (pro == 1) hipLaunchKernelGGL(k_tgemm<1>, grid, dim3(256), pad, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(k_tgemm<2>, grid, dim3(256), pad, (hipStream_t)stream, g);
    return launch_status(fn);
#undef TG_REQUIRE
}

#ifdef FQSS_T2_STAMP
extern "C" int fqss_debug_t2_stamps(unsigned long long* out) {      // host buffer [8][160][3]
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_t2_stamp), sizeof(unsigned long long) * 8 * 160 * 3) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int fqss_tgemm_tiled_ok(int Ci, int Co, int M1) {
    return Co > 0 && Co % T2BM == 0 && Ci % (4 * TBK) == 0 && Ci >= 4 * TBK && Ci <= 512 && M1 > 0 && M1 % 32 == 0;
}

extern "C" int fqss_split3_tiles(const float* w, uint16_t* tiles, int Co, int Ci, fqss_stream_t stream) {
    FQSS_REQUIRE(w && tiles && Co > 0 && Co % T2BM == 0 && Ci > 0 && Ci % T2BK == 0 && aligned16(tiles), "whole 256 x 16 tiles only");
    int64_t nb = cdiv((int64_t)Co * Ci, 256);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(k_split3_tiles, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, w, tiles, Co, Ci);
    return launch_status("fqss_split3_tiles");
}

extern "C" int fqss_tgemm(const uint16_t* planes, const float* x, int B, int Ci, int Co, int M, int64_t ld_x, int pro,
                          const double* pro_stats, const float* pro_gamma, const float* pro_beta, float pro_eps,
                          const float* pro_slope, const float* bias, int act, const float* slope, double* stats_out, int M1,
                          float* c1, const float* r1, int64_t ld_c1, float* c2, const float* r2, int64_t ld_c2,
                          fqss_stream_t stream) {
    return tgemm_launch(false, "fqss_tgemm", planes, x, B, Ci, Co, M, ld_x, pro, pro_stats, pro_gamma, pro_beta, pro_eps, pro_slope, bias, act, slope,
                        stats_out, M1, c1, r1, ld_c1, c2, r2, ld_c2, stream);
}

extern "C" int fqss_tgemm_tiled(const uint16_t* tiles, const float* x, int B, int Ci, int Co, int M, int64_t ld_x, int pro,
                                const double* pro_stats, const float* pro_gamma, const float* pro_beta, float pro_eps,
                                const float* pro_slope, const float* bias, int act, const float* slope, double* stats_out, int M1,
                                float* c1, const float* r1, int64_t ld_c1, float* c2, const float* r2, int64_t ld_c2,
                                fqss_stream_t stream) {
    return tgemm_launch(true, "fqss_tgemm_tiled", tiles, x, B, Ci, Co, M, ld_x, pro, pro_stats, pro_gamma, pro_beta, pro_eps, pro_slope, bias, act, slope,
                        stats_out, M1, c1, r1, ld_c1, c2, r2, ld_c2, stream);
}

extern "C" int fqss_tdw(const float* x, const double* stats_in, const float* gamma, const float* beta, float eps,
                        const float* w, const float* bias, const float* slope, float* y, double* stats_out, int B, int C,
                        int M, int K, int dil, int pad, int64_t ld_x, int64_t ld_y, fqss_stream_t stream) {
    FQSS_REQUIRE(x && stats_in && gamma && beta && w && slope && y && stats_out, "null pointer");
    FQSS_REQUIRE(B >= 0 && C > 0 && M > 0 && K > 0 && K <= 8 && dil > 0 && 2 * pad == dil * (K - 1), "bad conv geometry");
    FQSS_REQUIRE(aligned16(x) && aligned16(y) && ld_x % 4 == 0 && ld_y % 4 == 0 && ld_x >= ((M + 3) & ~3) && ld_y >= ((M + 3) & ~3) &&
                     (int64_t)B * C <= 65535 * 16ll && ld_x < (1ll << 30), "rows must be 16-B aligned");
    if (B == 0) return FQSS_OK;
    const int rows = B * C;
    FQSS_REQUIRE(M <= 12 * 1024, "rows longer than 12288 positions do not fit the LDS row buffer");
    hipLaunchKernelGGL(k_tdw, dim3((unsigned)rows), dim3(256), (size_t)((M + 3) & ~3) * sizeof(float), (hipStream_t)stream, x, stats_in, gamma, beta,
                       (double)C * (double)M, eps, w, bias, slope, y, stats_out, rows, C, M, K, dil, pad, (int)ld_x, (int)ld_y);
    return launch_status("fqss_tdw");
}

extern "C" int fqss_tstats(const float* x, int B, int C, int M, int64_t ld, double* ws, fqss_stream_t stream) {
    FQSS_REQUIRE(x && ws && B >= 0 && B <= 65535 && C > 0 && M > 0 && aligned16(x) && ld % 4 == 0 && ld >= ((M + 3) & ~3), "bad args");
    if (B == 0) return FQSS_OK;
    const int nb = C < 64 ? C : 64;
    hipLaunchKernelGGL(k_tstats, dim3((unsigned)nb, (unsigned)B), dim3(256), 0, (hipStream_t)stream, x, C, M, ld, ws);
    return launch_status("fqss_tstats");
}
