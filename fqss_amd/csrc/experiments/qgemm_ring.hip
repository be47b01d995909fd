// qgemm_ring.hip -- the student's data-gradient q-GEMM on the four-slot LDS ring of k_tgemm2 (csrc/teacher.hip, round 4).
//
//     gx[ci][n] = sum_co Wi[co][ci] * (dw[co] * gz[co][n])      (+ addend[ci][n]: the other branch of a residual fork)
//
// Same arithmetic as k_qgemm<1, 3> (csrc/qgemm.hip): the fp32 gradient, scaled by its row's delta_w, is split EXACTLY into three bf16
// pieces, the 8-bit weight code is one exact bf16 value, three MFMA products per term, fp32 accumulation.  What changes is the
// structure (the round-2 ablation of k_qgemm<1>: loads 24 us + MFMAs 16 + split 9.5 + stores 9.5 simply added up):
//   * one workgroup owns BM = 256 (or 128) input channels x 128 positions and loops over the row tiles of the weight: the gradient panel
//     is read and split once per 256 rows;
//   * eight waves in three roles, one compute wave + one memory wave per SIMD:
//       waves 0-3  compute: fragment reads + 3 x (BM / 32) MFMAs per 16-deep k-tile, no vector-memory instruction in the k-loop;
//       waves 4-5  gradients: fp32 panel -> registers (six tiles in flight: HBM latency) -> scale + exact 3-way split -> LDS;
//       waves 6-7  weights: int8 codes (L2-resident) -> registers (three tiles in flight) -> bf16 -> LDS;
//     each role has its own vmcnt counter; every counted load is issued through asm from straight-line code (a register that an asm
//     load is still filling must not meet a control-flow merge);
//   * a ring of four LDS slots of one 16-deep k-tile each, one barrier per k-tile: in iteration i the compute waves read slot i & 3, tile
//     i + 1 is already published, tile i + 2 is being written, (the loads of) later tiles are in flight;
//   * the epilogue's result stores of row tile mt drain under the k-loop of row tile mt + 1 (the memory waves run ahead).
// Reference replaced: the input gradient of F.conv1d(k = 1) inside Conv1dQ / Conv1dNlQ (qat_layers.py:137-146, 202-212) + autograd.
#include <type_traits>

#include "fqss_dev.h"

namespace fqss {

typedef float f32x16r __attribute__((ext_vector_type(16)));
typedef float f32x4r __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4r __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8r __attribute__((ext_vector_type(8)));
typedef short s16x4r __attribute__((ext_vector_type(4)));

constexpr int RBN = 128, RBK = 16, RLDN = 160, RLDT = 36;
constexpr int R_B_SLOT = 3 * RBK * RLDN * 2;              // 15,360: [plane][16 k][128 + 32 pad] bf16
constexpr int R_EPI = 4 * 32 * RLDT * 4;                  // 18,432: the compute waves' staging tiles
constexpr int R_SCALE = 1024 * 4;                         // delta_w of up to 1024 reduction rows

__device__ __forceinline__ float r_tr(float f) { return __uint_as_float(__float_as_uint(f) & 0xFFFF0000u); }
// 16-B loads with a scalar base + a 32-bit lane offset.  s_nop 4: the compiler may have produced the scalar base by a VALU instruction
// (v_readlane of a spilled SGPR) right in front of the statement; a vector-memory instruction that reads such an SGPR needs five wait
// states and nothing pads the inside of an asm statement (csrc/teacher.hip: wrong tiles and a memory fault without it)
__device__ __forceinline__ void r_load16s(f32x4r& d, const void* sbase, unsigned voff) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void r_load16s(u32x4r& d, const void* sbase, unsigned voff) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(voff), "s"(sbase) : "memory");
}
template <int T, int N, class F>
__device__ __forceinline__ void r_unroll(F& f) {
    if constexpr (T < N) {
        f(std::integral_constant<int, T>{});
        r_unroll<T + 1, N>(f);
    }
}
// every barrier goes through R_SYNC(<s_waitcnt in front of it, or "">).  FQSS_R_STAMP (diagnostic builds, tools/t2_stamps.py): workgroup 0
// records, per wave and barrier, the cycle counter before the wait, before the barrier and behind it.
#ifdef FQSS_R_STAMP
__device__ unsigned long long g_r_stamp[8][160][3];
#define R_SYNC(W)                                                                        \
    do {                                                                                 \
        const bool st_on = blockIdx.x == 0 && lane == 0 && sidx < 160;                   \
        if (st_on) g_r_stamp[wave][sidx][0] = __builtin_amdgcn_s_memtime();              \
        asm volatile(W ::: "memory");                                                    \
        if (st_on) g_r_stamp[wave][sidx][1] = __builtin_amdgcn_s_memtime();              \
        asm volatile("s_barrier" ::: "memory");                                          \
        if (st_on) g_r_stamp[wave][sidx][2] = __builtin_amdgcn_s_memtime();              \
        ++sidx;                                                                          \
    } while (0)
#else
#define R_SYNC(W) asm volatile(W "\n\ts_barrier" ::: "memory")
#endif

struct RDgradArgs {
    const int8_t* wiT;            // [Ci][Co] weight codes, transposed (k = co contiguous)
    const float* gz1;             // [B][Co1][ld1]
    const float* gz2;             // [B][Co2][ld2] or null
    const float* dw;              // [Co]
    const float* addend;          // [B][Ci][ld_add] or null
    float* gx;                    // [B][Ci][ld_gx]
    int Ci, Co, Co1, N, tiles_n;
    int64_t ld1, ld2, ld_add, ld_gx;
};

template <int BM>
__global__ __launch_bounds__(512, 2) void k_qdgrad_ring(RDgradArgs g) {
    constexpr int A_SLOT = BM * RBK * 2;                  // [BM rows][16 k] bf16, 32-B rows, the two 16-B chunks swapped where (row >> 3) & 1
    constexpr int B_OFF = 4 * A_SLOT, SC_OFF = B_OFF + 4 * R_B_SLOT, EPI_OFF = SC_OFF + R_SCALE;
    constexpr int MI = BM / 64;                           // 32-row blocks per compute wave (2 waves along the rows)
    constexpr int RPL = BM / 128;                         // weight rows per lane (2 weight waves)
    extern __shared__ __attribute__((aligned(16))) unsigned char smr[];
    float* scl = reinterpret_cast<float*>(smr + SC_OFF);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int panel = blockIdx.x;
    const int b = panel / g.tiles_n, j0 = (panel % g.tiles_n) * RBN;
#ifdef FQSS_R_STAMP
    int sidx = 0;
#endif
    for (int k = tid; k < g.Co; k += 512) scl[k] = g.dw[k];
    R_SYNC("s_waitcnt vmcnt(0) lgkmcnt(0)");
    const int nkt = g.Co / RBK;          // a multiple of 8 (host)
    const int n_last = (g.N - 1) & ~3;
    const int tiles_m = g.Ci / BM;

    // Barriers of one row tile, the same count in every role: P (tiles 0 and 1 published), one per iteration i = 0 .. nkt - 2 (tile i + 2
    // published), and the last iteration's (nothing left to publish).
    if (wave >= 6) {
        // =============================== weight waves (2): int8 codes -> bf16 -> LDS ================================================
        // lane -> rows (wave - 6) * BM / 2 + lane (+ 64): 16 B = the 16 codes of one k-tile per row and load; three tiles in flight
        const int wrow = (wave - 6) * (BM / 2) + lane;
        struct WStage { u32x4r r[RPL]; };
        auto row_tile = [&](auto NKT, int mt) {
            constexpr int nk = decltype(NKT)::value;
            const int8_t* abase = g.wiT + (int64_t)mt * BM * g.Co;                     // wave-uniform
            const unsigned avoff = (unsigned)(wrow * g.Co);                            // < 2^32
            WStage W0, W1, W2;
            auto ld = [&](WStage& st, int t) {
#pragma unroll
                for (int q = 0; q < RPL; ++q) r_load16s(st.r[q], abase + (int64_t)q * 64 * g.Co + t * RBK, avoff);
            };
            auto put = [&](auto T) {
                constexpr int t = decltype(T)::value;
                constexpr int younger = (nk - 1 - t) < 2 ? (nk - 1 - t) : 2;
                WStage& st = (t % 3) == 0 ? W0 : (t % 3) == 1 ? W1 : W2;
                if constexpr (RPL == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(st.r[0]), "+v"(st.r[1]) : "n"(RPL * younger) : "memory");
                else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(st.r[0]) : "n"(RPL * younger) : "memory");
#pragma unroll
                for (int q = 0; q < RPL; ++q) {
                    const int row = wrow + 64 * q;
                    unsigned int o[8];
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        float f[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) f[e] = (float)(int)(signed char)((st.r[q][w] >> (8 * e)) & 0xFFu);
                        o[2 * w] = __builtin_amdgcn_perm(__float_as_uint(f[1]), __float_as_uint(f[0]), 0x07060302u);
                        o[2 * w + 1] = __builtin_amdgcn_perm(__float_as_uint(f[3]), __float_as_uint(f[2]), 0x07060302u);
                    }
                    unsigned char* as = smr + (t & 3) * A_SLOT + row * 32;
                    const int sw = (row >> 3) & 1;
                    *reinterpret_cast<uint4*>(as + ((0 ^ sw) << 4)) = make_uint4(o[0], o[1], o[2], o[3]);
                    *reinterpret_cast<uint4*>(as + ((1 ^ sw) << 4)) = make_uint4(o[4], o[5], o[6], o[7]);
                }
                if (t + 3 <= nk - 1) ld(st, t + 3);
            };
            ld(W0, 0); ld(W1, 1); ld(W2, 2);
            put(std::integral_constant<int, 0>{});
            put(std::integral_constant<int, 1>{});
            R_SYNC("s_waitcnt lgkmcnt(0)");                    // P
            auto step = [&](auto I) {
                constexpr int i = decltype(I)::value;
                if constexpr (i + 2 <= nk - 1) put(std::integral_constant<int, i + 2>{});
                R_SYNC("s_waitcnt lgkmcnt(0)");
            };
            r_unroll<0, nk>(step);                             // iterations 0 .. nkt - 1
        };
        for (int mt = 0; mt < tiles_m; ++mt) {
            if (nkt == 8) row_tile(std::integral_constant<int, 8>{}, mt);
            else if (nkt == 16) row_tile(std::integral_constant<int, 16>{}, mt);
            else if (nkt == 24) row_tile(std::integral_constant<int, 24>{}, mt);
            else row_tile(std::integral_constant<int, 32>{}, mt);
        }
    } else if (wave >= 4) {
        // =============================== gradient waves (2): fp32 panel -> scale -> three bf16 planes in LDS =========================
        const int lt = tid - 256;
        const int bk_row = lt >> 5, bk_c = (lt & 31) * 4;
        const int col = min(j0 + bk_c, n_last);
        const unsigned voff1 = (unsigned)((bk_row * (int)g.ld1 + col) * 4), voff2 = (unsigned)((bk_row * (int)g.ld2 + col) * 4);
        const float* base1 = g.gz1 + (int64_t)b * g.Co1 * g.ld1;
        const float* base2 = g.gz2 != nullptr ? g.gz2 + (int64_t)b * (g.Co - g.Co1) * g.ld2 : nullptr;
        const int t_split = g.Co1 / RBK;                   // tiles >= t_split come from the second layer's gradient
        struct LStage { f32x4r rb[4]; };
        auto load_b = [&](LStage& st, int t) {
            const bool first = t < t_split;
            const float* base = first ? base1 + (int64_t)t * RBK * g.ld1 : base2 + (int64_t)(t - t_split) * RBK * g.ld2;
            const int64_t r4 = 4 * (first ? g.ld1 : g.ld2);
            const unsigned vo = first ? voff1 : voff2;
#pragma unroll
            for (int q = 0; q < 4; ++q) r_load16s(st.rb[q], base + q * r4, vo);
        };
        auto store_b = [&](LStage& st, int t) {
            unsigned char* bs = smr + B_OFF + (t & 3) * R_B_SLOT + (bk_row * RLDN + bk_c) * 2;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float sc = scl[t * RBK + bk_row + 4 * q];
                float h0[4], r1[4], r2[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = st.rb[q][e] * sc;
                    h0[e] = v;
                    r1[e] = v - r_tr(v);
                    r2[e] = r1[e] - r_tr(r1[e]);
                }
                uint2 o1, o2, o3;
                o1.x = __builtin_amdgcn_perm(__float_as_uint(h0[1]), __float_as_uint(h0[0]), 0x07060302u);
                o1.y = __builtin_amdgcn_perm(__float_as_uint(h0[3]), __float_as_uint(h0[2]), 0x07060302u);
                o2.x = __builtin_amdgcn_perm(__float_as_uint(r1[1]), __float_as_uint(r1[0]), 0x07060302u);
                o2.y = __builtin_amdgcn_perm(__float_as_uint(r1[3]), __float_as_uint(r1[2]), 0x07060302u);
                o3.x = __builtin_amdgcn_perm(__float_as_uint(r2[1]), __float_as_uint(r2[0]), 0x07060302u);
                o3.y = __builtin_amdgcn_perm(__float_as_uint(r2[3]), __float_as_uint(r2[2]), 0x07060302u);
                *reinterpret_cast<uint2*>(bs + q * 4 * RLDN * 2) = o1;
                *reinterpret_cast<uint2*>(bs + q * 4 * RLDN * 2 + RBK * RLDN * 2) = o2;
                *reinterpret_cast<uint2*>(bs + q * 4 * RLDN * 2 + 2 * RBK * RLDN * 2) = o3;
            }
        };
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
            put(std::integral_constant<int, 0>{});
            put(std::integral_constant<int, 1>{});
            R_SYNC("s_waitcnt lgkmcnt(0)");                    // P
            auto step = [&](auto I) {
                constexpr int i = decltype(I)::value;
                if constexpr (i + 2 <= nk - 1) put(std::integral_constant<int, i + 2>{});
                R_SYNC("s_waitcnt lgkmcnt(0)");
            };
            r_unroll<0, nk>(step);
        };
        for (int mt = 0; mt < tiles_m; ++mt) {
            if (nkt == 8) row_tile(std::integral_constant<int, 8>{});
            else if (nkt == 16) row_tile(std::integral_constant<int, 16>{});
            else if (nkt == 24) row_tile(std::integral_constant<int, 24>{});
            else row_tile(std::integral_constant<int, 32>{});
        }
    } else {
        // =============================== compute waves (4 = 2 x 2, wave tile BM / 2 x 64) ============================================
        const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
        const int gq = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
        f32x16r acc[MI][2];
        auto rd_b = [&](bf16x8r (&bfr)[3][2], int slot) {
            typedef unsigned short (*BsT)[RBK][RLDN];
            BsT Bs = reinterpret_cast<BsT>(smr + B_OFF + slot * R_B_SLOT);
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int kr = 8 * (gq >> 1) + tq;
                    const int nc = wn * 64 + ni * 32 + 16 * (gq & 1) + 4 * tp;
                    union { bf16x8r v; s16x4r h[2]; } u;
                    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4r __attribute__((address_space(3)))*)(&Bs[p][kr][nc]));
                    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4r __attribute__((address_space(3)))*)(&Bs[p][kr + 4][nc]));
                    bfr[p][ni] = u.v;
                }
        };
        auto rd_a = [&](bf16x8r (&af)[MI], int slot) {
            const unsigned char* as = smr + slot * A_SLOT;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int row = wm * (BM / 2) + mi * 32 + lr;
                af[mi] = *reinterpret_cast<const bf16x8r*>(as + row * 32 + ((lh ^ ((row >> 3) & 1)) << 4));
            }
        };
        auto mm = [&](bf16x8r (&af)[MI], bf16x8r (&bfr)[3][2]) {
#pragma unroll
            for (int p = 2; p >= 0; --p)        // smallest pieces first
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bfr[p][ni], acc[mi][ni], 0, 0, 0);
        };
        float(*Tt)[RLDT] = reinterpret_cast<float(*)[RLDT]>(smr + EPI_OFF + wave * 32 * RLDT * 4);

        for (int mt = 0; mt < tiles_m; ++mt) {
            const int rowt0 = mt * BM + wm * (BM / 2);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
            // P: tiles 0 and 1 are published.  No vmcnt wait: this wave's result stores of the previous row tile stay in flight.
            R_SYNC("s_waitcnt lgkmcnt(0)");
            bf16x8r bA[3][2], bB[3][2], aA[MI], aB[MI];
            rd_b(bA, 0);
            rd_a(aA, 0);
            for (int i = 0; i < nkt; i += 2) {
                // tile i; the fragments of tile i + 1 (published one barrier earlier) are requested in front of the barrier
                rd_b(bB, (i + 1) & 3);
                rd_a(aB, (i + 1) & 3);
                mm(aA, bA);
                R_SYNC("");
                if (i + 2 < nkt) {
                    rd_b(bA, (i + 2) & 3);
                    rd_a(aA, (i + 2) & 3);
                }
                mm(aB, bB);
                R_SYNC("");
            }
            // ---- epilogue: (+ addend), 16-B/lane row stores through a wave-private LDS tile (its own region: the memory waves are
            // already filling the slots for the next row tile).  Addresses hang off an opaque copy of the lane id (k_tgemm2).
            int el = lane;
            asm volatile("" : "+v"(el));
            const int e_c4 = (el & 7) * 4, e_r8 = el >> 3, e_lr = el & 31, e_lh = el >> 5;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int rowt = rowt0 + mi * 32;
                float* Cb = g.gx + ((int64_t)b * g.Ci + rowt) * g.ld_gx;
                const float* Rb = g.addend != nullptr ? g.addend + ((int64_t)b * g.Ci + rowt) * g.ld_add : nullptr;
                float4 res[2][4];
                if (Rb != nullptr) {
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int pass = 0; pass < 4; ++pass)
                            res[ni][pass] = *reinterpret_cast<const float4*>(Rb + (int64_t)(pass * 8 + e_r8) * g.ld_add + min(j0 + wn * 64 + ni * 32 + e_c4, n_last));
                }
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) Tt[(r & 3) + 8 * (r >> 2) + 4 * e_lh][e_lr] = acc[mi][ni][r];
                    const int col = j0 + wn * 64 + ni * 32 + e_c4;
#pragma unroll
                    for (int pass = 0; pass < 4; ++pass) {
                        const int rl = pass * 8 + e_r8;
                        float4 t = *reinterpret_cast<const float4*>(&Tt[rl][e_c4]);
                        if (Rb != nullptr) {
                            const float4 q = res[ni][pass];
                            t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
                        }
                        if (col < g.N) store16(Cb + (int64_t)rl * g.ld_gx + col, t);
                    }
                }
            }
        }
    }
}

// OPT-IN (FQSS_DGRAD_RING=1), measured and NOT the default: 33.6 / 48.9 us against k_qgemm<1>'s 25.8 / 41.9 us at the two cfg-2 shapes.
// Cycle stamps (profiles/r04_r_stamps_dx2.txt) say why: with three products per term the MFMAs are light (768 cycles per k-tile), the
// TWO gradient waves' split (~1,500 cycles per tile) sets the pace, and the epilogue -- 64 MB of result stores per launch, as long as
// the whole k-loop -- runs on the compute waves with nothing beside it (the memory waves run ahead by the ring's two tiles and stop),
// where k_qgemm's three to four resident workgroups per CU overlap their phases with each other.  The teacher's six-product GEMM,
// whose MFMAs fill the time, is the shape this structure pays for.
bool qdgrad_ring_ok(int Ci, int Co1, int Co2) {
    static const bool on = [] { const char* e = getenv("FQSS_DGRAD_RING"); return e && atoi(e) != 0; }();
    const int Co = Co1 + Co2;
    return on && (Ci % 256 == 0 || Ci == 128) && Co % 128 == 0 && Co <= 512 && Co1 % 16 == 0;
}

int qdgrad_ring(const char* who, const float* gz1, const float* gz2, const int8_t* wiT, const float* dw, const float* addend, float* gx, int B,
                int Ci, int Co1, int Co2, int M, int64_t ld_gz1, int64_t ld_gz2, int64_t ld_add, int64_t ld_gx, fqss_stream_t stream) {
    RDgradArgs g{};
    g.wiT = wiT; g.gz1 = gz1; g.gz2 = gz2; g.dw = dw; g.addend = addend; g.gx = gx;
    g.Ci = Ci; g.Co = Co1 + Co2; g.Co1 = Co1; g.N = M; g.tiles_n = (int)cdiv(M, RBN);
    g.ld1 = ld_gz1; g.ld2 = gz2 ? ld_gz2 : ld_gz1; g.ld_add = ld_add; g.ld_gx = ld_gx;
    if (ld_gz1 >= (1ll << 26) || ld_gz2 >= (1ll << 26) || (int64_t)Ci * g.Co >= (1ll << 31)) {
        set_error("%s: rows too long for the ring form's 32-bit lane offsets", who);
        return FQSS_EINVAL;
    }
    const dim3 grid((unsigned)(g.tiles_n * B));
    if (Ci % 256 == 0) {
        constexpr int smem = 4 * (256 * RBK * 2) + 4 * R_B_SLOT + R_SCALE + R_EPI;
        static const bool ok = hipFuncSetAttribute((const void*)k_qdgrad_ring<256>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) == hipSuccess;
        if (!ok) { set_error("%s: k_qdgrad_ring needs %d B of dynamic LDS", who, smem); return FQSS_EINVAL; }
        hipLaunchKernelGGL(k_qdgrad_ring<256>, grid, dim3(512), smem, (hipStream_t)stream, g);
    } else {
        constexpr int smem = 4 * (128 * RBK * 2) + 4 * R_B_SLOT + R_SCALE + R_EPI;
        static const bool ok = hipFuncSetAttribute((const void*)k_qdgrad_ring<128>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) == hipSuccess;
        if (!ok) { set_error("%s: k_qdgrad_ring needs %d B of dynamic LDS", who, smem); return FQSS_EINVAL; }
        hipLaunchKernelGGL(k_qdgrad_ring<128>, grid, dim3(512), smem, (hipStream_t)stream, g);
    }
    return launch_status(who);
}

}  // namespace fqss

#ifdef FQSS_R_STAMP
extern "C" int fqss_debug_r_stamps(unsigned long long* out) {      // host buffer [8][160][3]
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(fqss::g_r_stamp), sizeof(unsigned long long) * 8 * 160 * 3) == hipSuccess ? 0 : -1;
}
#endif
