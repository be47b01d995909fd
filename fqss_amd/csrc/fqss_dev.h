// fqss_dev.h -- shared device/host helpers for the gfx950 kernels of libfqss_hip.so.
// Built with -ffp-contract=off: every fp32 operator below is ONE IEEE-754 operation (the parity
// contract restates ATen op sequences whose intermediate roundings matter for the bin index).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fqss.h"

namespace fqss {

// ---------------------------------------------------------------- host side
void set_error(const char* fmt, ...);
int launch_status(const char* what);

#define FQSS_REQUIRE(cond, msg)                          \
    do {                                                 \
        if (!(cond)) {                                   \
            ::fqss::set_error("%s: %s", __func__, msg);  \
            return FQSS_EINVAL;                          \
        }                                                \
    } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// 2-D launch shape for row-matrix element-wise kernels: x covers column chunks, y strides rows.
static inline dim3 grid_rows(int64_t rows, int64_t cols, int vec, int64_t max_blocks = 8192) {
    int64_t gx = cdiv(cols, 256 * (int64_t)vec);
    if (gx < 1) gx = 1;
    if (gx > 1024) gx = 1024;
    int64_t gy = max_blocks / gx;
    if (gy < 1) gy = 1;
    if (gy > rows) gy = rows;
    if (gy > 65535) gy = 65535;
    if (gy < 1) gy = 1;
    return dim3((unsigned)gx, (unsigned)gy, 1);
}

// XCD-aware tile order for GEMM-like grids.  Workgroups are dealt round-robin to the 8 XCDs by linear id, and
// every XCD has its own L2: tiles that re-read the same operand panel ("group" = the tiles of one panel) are made
// consecutive workgroups of ONE XCD, so the panel comes from HBM once and from that L2 afterwards.
// Launch xcd_grid(groups, per_group) workgroups; tile_of() maps blockIdx.x -> (group, index in group) or false.
constexpr int kXcds = 8;
static inline unsigned xcd_grid(int64_t groups, int64_t per_group) { return (unsigned)(cdiv(groups, kXcds) * kXcds * per_group); }

// the data-gradient q-GEMM on the LDS ring (csrc/qgemm_ring.hip), dispatched from csrc/qgemm.hip where its shape rules hold
bool qdgrad_ring_ok(int Ci, int Co1, int Co2);
int qdgrad_ring(const char* who, const float* gz1, const float* gz2, const int8_t* wiT, const float* dw, const float* addend, float* gx, int B,
                int Ci, int Co1, int Co2, int M, int64_t ld_gz1, int64_t ld_gz2, int64_t ld_add, int64_t ld_gx, fqss_stream_t stream);

// ---------------------------------------------------------------- device side
#if defined(__HIPCC__)
__device__ __forceinline__ bool xcd_tile(int groups, int per_group, int& group, int& idx) {
    const int L = blockIdx.x, xcd = L % kXcds, slot = L / kXcds;
    idx = slot % per_group;
    group = (slot / per_group) * kXcds + xcd;
    return group < groups;
}

constexpr int kWave = 64;

// order-preserving float <-> uint map, so float min/max run on integer atomics (exact, order-free)
__device__ __forceinline__ uint32_t f2ord(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

// Sum over the 64 lanes of a wave on the DPP crossbar (round 6): six v_add with a lane-permuted source -- pairs, quads, half rows, rows
// (quad_perm / row_half_mirror / row_mirror), then row 0 into 1 and 2 into 3 (row_bcast:15), then rows 0-1 into 3 (row_bcast:31).  The
// total is valid in LANE 63 ONLY.  No LDS traffic and ~8-cycle steps, where __shfl_xor is a ds_bpermute_b32 round trip per step (the
// cycle stamps of k_dwq_bwd put its two block reductions at 7 of the workgroup's 20 us, profiles/r06_dwb_stamps.txt).  A fixed tree:
// the same bits every run; NOT the xor butterfly's association, so sums differ from the round-5 library's in the last bit.
// All 64 lanes must be active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false); }
template <int CTRL, int ROW_MASK = 0xf, typename T>
__device__ __forceinline__ T dpp_get(T v) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "32- or 64-bit values");
    if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(T, dpp_i32<CTRL, ROW_MASK>(__builtin_bit_cast(int, v)));
    } else {
        const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
        const unsigned lo = (unsigned)dpp_i32<CTRL, ROW_MASK>((int)(unsigned)b), hi = (unsigned)dpp_i32<CTRL, ROW_MASK>((int)(unsigned)(b >> 32));
        return __builtin_bit_cast(T, ((unsigned long long)hi << 32) | lo);
    }
}
template <typename T>
__device__ __forceinline__ T wave_sum63(T v) {
#ifdef FQSS_NO_DPP_SUMS
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
#else
    v += dpp_get<0xB1>(v);          // quad_perm:[1,0,3,2]
    v += dpp_get<0x4E>(v);          // quad_perm:[2,3,0,1]
    v += dpp_get<0x141>(v);         // row_half_mirror
    v += dpp_get<0x140>(v);         // row_mirror: every lane of a 16-lane row holds the row's sum
    v += dpp_get<0x142, 0xa>(v);    // row_bcast:15 into rows 1 and 3
    v += dpp_get<0x143, 0xc>(v);    // row_bcast:31 into rows 2 and 3
    return v;
#endif
}

// lane 63's value in every lane (v_readlane_b32: through a scalar register)
template <typename T>
__device__ __forceinline__ T from_lane63(T v) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "32- or 64-bit values");
    if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(T, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
    } else {
        const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 63);
        return __builtin_bit_cast(T, ((unsigned long long)hi << 32) | lo);
    }
}
// the wave's sum / minimum / maximum in EVERY lane: the DPP tree + one readlane (round 6: these were six ds_bpermute round trips, in
// front of the first normalised element of every LayerNorm row and of every softmax row)
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
    return from_lane63(wave_sum63(v));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_keep(float v) {      // rows outside ROW_MASK read back their own value (min / max: the identity)
    const int i = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_update_dpp(i, i, CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_min(float v) {
    v = fminf(v, dpp_get<0xB1>(v)); v = fminf(v, dpp_get<0x4E>(v)); v = fminf(v, dpp_get<0x141>(v)); v = fminf(v, dpp_get<0x140>(v));
    v = fminf(v, dpp_keep<0x142, 0xa>(v)); v = fminf(v, dpp_keep<0x143, 0xc>(v));
    return from_lane63(v);
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_get<0xB1>(v)); v = fmaxf(v, dpp_get<0x4E>(v)); v = fmaxf(v, dpp_get<0x141>(v)); v = fmaxf(v, dpp_get<0x140>(v));
    v = fmaxf(v, dpp_keep<0x142, 0xa>(v)); v = fmaxf(v, dpp_keep<0x143, 0xc>(v));
    return from_lane63(v);
}
// sum over each 16-lane row of the wave, valid in every lane of its row (four DPP steps)
template <typename T>
__device__ __forceinline__ T row16_sum(T v) {
    v += dpp_get<0xB1>(v);
    v += dpp_get<0x4E>(v);
    v += dpp_get<0x141>(v);
    v += dpp_get<0x140>(v);
    return v;
}

// block-wide sum of N values per thread; result valid in thread 0.  smem: N * (blockDim/64) T's.
template <typename T, int N>
__device__ __forceinline__ void block_sum(T (&v)[N], T* smem) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = wave_sum63(v[i]);
    if (lane == 63) {
#pragma unroll
        for (int i = 0; i < N; ++i) smem[i * nw + w] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            T s = smem[i * nw];
            for (int k = 1; k < nw; ++k) s += smem[i * nw + k];
            v[i] = s;
        }
    }
    __syncthreads();
}

// Block-wide sum of N fp32 per-thread partials: fp32 over the 64 lanes of a wave (32-bit shuffles), fp64 over the waves; result valid
// in thread 0.  smem: N * (blockDim/64) floats.  Half the instructions and one barrier pair less than block_sum<double, N> -- the
// reduction tails were 15 % of the streaming backward kernels (ablation of k_ewq_bwd: 4.5-5 us of 30).
template <int N>
__device__ __forceinline__ void block_sum_f32w(const float (&p)[N], float* smem, double (&out)[N]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float s = wave_sum63(p[i]);
        if (lane == 63) smem[i * nw + w] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double s = (double)smem[i * nw];
            for (int k = 1; k < nw; ++k) s += (double)smem[i * nw + k];
            out[i] = s;
        }
    }
    __syncthreads();
}

// ---- the quantizer arithmetic, op for op (qat_quant.py:139-146) ----
struct QRange {
    float lo, delta, inv;   // inv = RN(1/delta), one IEEE division per thread
};
__device__ __forceinline__ QRange load_qrange(const float* qmin, const float* qmax) {
    const float lo = *qmin, hi = *qmax;
    QRange r;
    r.lo = lo;
    r.delta = (hi - lo) / 255.0f;  // fp32 sub, IEEE fp32 div
    r.inv = 1.0f / r.delta;        // correctly rounded reciprocal (IEEE division)
    return r;
}
// Correctly rounded a/b in 3 instructions given y = RN(1/b) (Markstein: q = RN(a*y), r = a - b*q exactly by
// fma, RN(q + r*y) is the correctly rounded quotient; valid away from overflow/underflow).  The divisor
// (delta) is uniform per tensor, so the ~12-instruction IEEE sequence is paid once per thread instead of
// once per element.  Bit-equality with the IEEE division is brute-force checked on the GPU (tests/test_gpu_fq.py).
__device__ __forceinline__ float div_by(float a, float b, float y) {
    const float q = a * y;
    const float r = fmaf(-b, q, a);
    return fmaf(r, y, q);
}
// Pins a wave-uniform value (typically a field of a by-value kernel-argument struct) in SGPRs.  Such fields live in the
// kernarg segment; when two of them are selected per lane (`row < split ? g.ld1 : g.ld2`) the compiler selects the ADDRESS
// and issues a per-lane global load from the kernarg segment followed by s_waitcnt vmcnt(0) -- a memory round trip in the
// middle of a hand-pipelined loop (k_qgemm carried five of them).  Selecting between two pinned values is one v_cndmask.
template <typename T>
__device__ __forceinline__ T sgpr(T v) {
    asm("" : "+s"(v));
    return v;
}
// A 16-B store the optimiser cannot split: it turned `vec ? float4 store : up to 3 scalar stores` into one dwordx3 plus one
// conditional dword per lane (two partial-line writes per 16 B) in every k_qgemm epilogue.  Stores only count on vmcnt, which
// makes the compiler's own waits conservative, never wrong.
// s_nop 1 (round 5): a VMEM store of more than 8 bytes reads its data registers a moment after issue, and a VALU write of one of those
// registers needs wait states in between (LLVM's hazard recognizer inserts them for the compiler's own stores: 1, 2 on gfx940+) -- an
// asm statement is opaque to it, and the register allocator does hand the data registers to the very next address computation
// (seen in k_qwgrad_group, csrc/qgemm.hip st16_sc1: the next address stored in place of the data whenever the memory pipe was busy).
__device__ __forceinline__ void store16(float* p, const float4& v) {
    typedef float f32x4s __attribute__((ext_vector_type(4)));
    const f32x4s t = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(t) : "memory");
}
// Branch-free: NONE / ReLU are PReLU (ATen: z > 0 ? z : slope * z) with slope 1 / 0, so the (wave-uniform) choice is one
// scalar select the compiler hoists out of the caller's loops instead of two scalar branches per element (the GEMM
// epilogues carried 160-280 s_cbranch for it).  1 * z is exact; ReLU yields -0 for z < 0, which no consumer can tell
// from +0 (the quantizer subtracts the range minimum first, sums and products of zeros compare equal).
__device__ __forceinline__ float act_neg_scale(int act, float slope) {
    return act == FQSS_ACT_PRELU ? slope : (act == FQSS_ACT_RELU ? 0.0f : 1.0f);
}
__device__ __forceinline__ float act_apply(float z, int act, float slope) {
    return z > 0.0f ? z : act_neg_scale(act, slope) * z;
}
// activation backward, branch-free like act_apply: returns dL/dz for dL/dt = gt and accumulates the PReLU slope gradient
// (adding 0.0f where the branchy form skipped the add leaves the sum unchanged)
__device__ __forceinline__ float act_bwd(float z, float gt, int act, float slope, bool valid, float& p_slope) {
    const bool neg = !(z > 0.0f);
    p_slope += (act == FQSS_ACT_PRELU && valid && neg) ? z * gt : 0.0f;
    return neg ? act_neg_scale(act, slope) * gt : gt;
}
// Branching forms of the two helpers above, same results.  k_dwq_bwd keeps them: there the branch-free forms cost 13 more
// VGPRs (73 -> 86, one wave per SIMD less) and 45 -> 52 us.
__device__ __forceinline__ float act_apply_br(float z, int act, float slope) {
    if (act == FQSS_ACT_PRELU) return z > 0.0f ? z : slope * z;
    if (act == FQSS_ACT_RELU) return z > 0.0f ? z : 0.0f;
    return z;
}
__device__ __forceinline__ float act_bwd_br(float z, float gt, int act, float slope, bool valid, float& p_slope) {
    if (act == FQSS_ACT_PRELU) {
        const bool pos = z > 0.0f;
        if (valid && !pos) p_slope += z * gt;
        return pos ? gt : slope * gt;
    }
    if (act == FQSS_ACT_RELU) return (z > 0.0f) ? gt : 0.0f;
    return gt;
}
// returns the de-quantised value, c = clamped integer index as float, u = pre-round coordinate
__device__ __forceinline__ float fq_asym(float t, const QRange& r, float& c, float& u, bool& inr) {
    u = div_by(t - r.lo, r.delta, r.inv);   // == (t-lo)/delta, correctly rounded (x*(1/delta) alone flips 2.7 ppm of indices)
    const float X = rintf(u);   // v_rndne_f32: round-half-to-even == torch.round
    inr = (X >= 0.0f) && (X <= 255.0f);
    c = __builtin_amdgcn_fmed3f(X, 0.0f, 255.0f);
    return r.delta * c + r.lo;  // two roundings (mul, add): contraction is off
}

// The forward-only forms of fq_asym (these streaming kernels are VALU-issue bound: ~5 cycles per wave instruction and SIMD,
// measured; every instruction per element counts): the clamped code alone, packed with v_cvt_pk_u8_f32 (exact: the code is an
// integer in [0, 255]); the de-quantised value only where a caller stores it.
__device__ __forceinline__ float fq_code(float t, const QRange& r) {
    return __builtin_amdgcn_fmed3f(rintf(div_by(t - r.lo, r.delta, r.inv)), 0.0f, 255.0f);
}
__device__ __forceinline__ unsigned int pack_code(float c, unsigned int byte, unsigned int word) {
    return __builtin_amdgcn_cvt_pk_u8_f32(c, byte, word);
}
// exact integer statistics of 4 packed codes: sum c, sum c^2 (v_dot4_u32_u8)
__device__ __forceinline__ void code_stats4(unsigned int word, unsigned int& s, unsigned int& ss) {
    s = __builtin_amdgcn_udot4(word, 0x01010101u, s, false);
    ss = __builtin_amdgcn_udot4(word, word, ss, false);
}

#endif  // __HIPCC__

// ---------------------------------------------------------------- FQSS_DETERMINISTIC=1 (round 5; docs/history/DESIGN_rounds_1-5.md 2, "bit-reproducible steps")
// The fp32 atomics that are left in the step -- one add per (sample, channel) row into a bias / depthwise-weight gradient, the split
// adds of the frame-path weight gradients -- make those gradients depend on the ORDER the workgroups retire in (1e-7 relative, run to
// run).  In deterministic mode every such add goes, instead, into an integer shadow of the gradient arena it targets: the value as a 2-word fixed
// point number (hi = rint(v 2^30), lo = rint((v - hi 2^-30) 2^80): exact for every fp32 with 2^-56 <= |v| < 2^33, 8e-25 absolute below),
// added with 64-bit INTEGER atomics -- exact and commutative, so the sum does not depend on the order -- and fqss_det_finish rounds
// the sums once into the fp32 arena (fixed thread -> element map).  A TU that uses grad_add defines FQSS_USES_GRAD_ADD in front of
// this header: it then owns a copy of the control block and registers its setter (fqss_set_deterministic reaches every copy).
constexpr int kDetSlots = 3;             // 0: the parameter-gradient arena, 1: the dL/dW_q arena of runtime.QuantTables, 2: temporaries
struct DetCtl { long long* shadow[kDetSlots]; const float* base[kDetSlots]; long long n[kDetSlots]; };
typedef int (*det_setter_t)(const DetCtl*);      // 0 or the hipError_t of the copy into that TU's control block
void det_register(det_setter_t fn);      // train_ops.hip
#if defined(__HIPCC__) && defined(FQSS_USES_GRAD_ADD)
static __constant__ DetCtl d_det_ctl;     // (constant address space: the tail-of-kernel reads of grad_add are scalar-cache hits, not memory round trips)
namespace {
struct DetRegistrar {
    DetRegistrar() {
        det_register([](const DetCtl* c) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(d_det_ctl), c, sizeof(DetCtl)); });
    }
} det_registrar_;
}  // namespace
// det_preload(): "is the mode on", read at the START of a kernel whose tail issues grad_adds from one thread: the control block lives in
// global memory, and a load of it in the tail is a full memory round trip per grad_add behind the last barrier, with the workgroup's
// registers and LDS still allocated (k_dwq_bwd: four of them, ~3 us of a 20-us workgroup).  grad_add(addr, v, on) takes the answer.
__device__ __forceinline__ bool det_preload() { return d_det_ctl.shadow[0] != nullptr; }
__device__ __forceinline__ void grad_add(float* addr, float v);
__device__ __forceinline__ void grad_add(float* addr, float v, bool det_on) {
    if (det_on) grad_add(addr, v);
    else atomicAdd(addr, v);
}
__device__ __forceinline__ void grad_add(float* addr, float v) {
    if (d_det_ctl.shadow[0] != nullptr && fabsf(v) < 0x1p33f) {     // (NaN / inf / huge: the plain add below)
        const DetCtl c = d_det_ctl;      // the whole block in ONE batch of scalar loads (nine dependent ones cost a workgroup's tail ~3 us)
#pragma unroll
        for (int s = 0; s < kDetSlots; ++s) {
            long long* const sh = c.shadow[s];
            const long long i = addr - c.base[s];
            if (sh != nullptr && (unsigned long long)i < (unsigned long long)c.n[s]) {
                const double dv = (double)v;
                const long long hi = __double2ll_rn(dv * 0x1p30);
                const long long lo = __double2ll_rn((dv - (double)hi * 0x1p-30) * 0x1p80);
                if (hi != 0) atomicAdd(reinterpret_cast<unsigned long long*>(sh + 2 * i), (unsigned long long)hi);
                if (lo != 0) atomicAdd(reinterpret_cast<unsigned long long*>(sh + 2 * i + 1), (unsigned long long)lo);
                return;
            }
        }
    }
    atomicAdd(addr, v);
}
#endif
}  // namespace fqss
