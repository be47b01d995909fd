// qgemm.hip -- the student's pointwise convolutions on the bf16 matrix cores, EXACTLY.
//
// In the quantizing phase both operands of a student 1x1 conv live on 8-bit grids:
//     W_q[co][ci] = dw[co] * Wi[co][ci]          Wi in [-128,127]   (qat_quant.py:126-135)
//     x  [ci][n]  = dx * c[ci][n] + min_x        c  in [0,255]      (qat_quant.py:136-147)
// so   z = W_q x + b = dw[co] * ( dx * S[co][n] + min_x * R[co] ) + b[co]
// with S = sum_ci Wi*c and R = sum_ci Wi integer sums.  |Wi*c| <= 32640 and Ci <= 512 keep |S| < 2^24,
// so S accumulated in the fp32 accumulators of v_mfma_f32_32x32x16_bf16 (integers <= 255 are exact in
// bf16) is the EXACT integer: the result carries 4 roundings instead of the reference's Ci, at 16x the
// fp32-MFMA rate.  It differs from the reference's fp32 sum only by the reference's own rounding (G1).
//
// Backward operands are gradients (not on a grid).  They are split exactly into three bf16 pieces
// (24-bit mantissa = 8+8+8, by truncation) so every product with an 8-bit-grid value is exact and only
// the fp32 accumulation rounds -- the numerics of an fp32 fma chain at 3/16 of its MFMA cost:
//     dgrad  gx[ci][n]   = sum_co Wi[co][ci] * (dw[co]*gz[co][n])
//     wgrad  gW_q[co][ci]+= dx * sum_n gz[co][n]*c[ci][n] + min_x * sum_n gz[co][n]
//
// fwd / dgrad (k_qgemm): 128x64x32 tiles per 256-thread workgroup, 4 waves of 64x32 (2 MFMA 32x32x16).  A/B fragments
// need 8 consecutive k per lane; operands that are n-contiguous in HBM (activations, gradients) keep their natural
// [k][n] image in LDS and are transposed on the fly by ds_read_b64_tr_b16.  Two hand-scheduled register stages of
// global loads, XCD-aware tile order, LDS-staged 16-B/lane epilogue.  The forward epilogue can also apply the layer's
// own non-linearity + fake-quant and emit the output codes (fqss_qpw_fwdq); two layers on one input (res | skip)
// run as one GEMM over concatenated channels (fqss_qpw_*2).
// wgrad (k_qwgrad): register-direct fragments, no LDS, 4-stage hand-scheduled load ring, float atomics.
//
// Reference replaced: F.conv1d(k=1) of Conv1dQ / Conv1dNlQ (qat_layers.py:137-146, 202-212) and its autograd.
#include <cstdlib>
#include <type_traits>

#define FQSS_USES_GRAD_ADD   // the fp32 gradient atomics of this file go through grad_add (fqss_dev.h: FQSS_DETERMINISTIC=1)
#include "fqss_dev.h"

namespace fqss {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// block tile 128(m) x 64(n) x 32(k): 4 waves as 2x2, each 64x32 (2 MFMA 32x32x16 tiles).  The narrow N tile keeps
// LDS/VGPR use low enough for 3-4 workgroups per CU, whose load / MFMA / store phases then overlap each other
// (ablation on MI355X: with 128x128 tiles and 1-2 workgroups per CU the three phases simply added up).
constexpr int QBM = 128, QBN = 64, QBK = 32;
constexpr int LDK = 40;    // bf16 per row of a k-contiguous image (32 + 8 pad = 80 B: conflict-free ds_read_b128)
constexpr int LDN = 96;    // bf16 per row of an n-contiguous image (64 + 32 pad = 192 B: conflict-free tr reads)
constexpr int LDT = 36;    // fp32 per row of the epilogue staging tile (32 + 4 pad, 16-B aligned rows)
#ifndef FQSS_WGRAD_BLOCKS
#define FQSS_WGRAD_BLOCKS 256   // one workgroup per CU: the fp32 atomics of the epilogue cost ~20 ns per workgroup-tile, more slices lose
#endif

__device__ __forceinline__ unsigned short f2bf_trunc(float f) { return (unsigned short)(__float_as_uint(f) >> 16); }
__device__ __forceinline__ float bf_trunc(float f) { return __uint_as_float(__float_as_uint(f) & 0xFFFF0000u); }
// the fp32 word whose HIGH half is f rounded to nearest-even bf16 (finite f): the second piece of the two-piece gradient split
__device__ __forceinline__ unsigned int bf_rne_word(float f) {
    const unsigned int u = __float_as_uint(f);
    return u + 0x7FFFu + ((u >> 16) & 1u);
}

// exact 3-way split g = b1 + b2 + b3 (each bf16-representable)
__device__ __forceinline__ void split3(float g, unsigned short& b1, unsigned short& b2, unsigned short& b3) {
    const float h1 = bf_trunc(g);
    const float r1 = g - h1;
    const float h2 = bf_trunc(r1);
    const float r2 = r1 - h2;
    b1 = f2bf_trunc(h1);
    b2 = f2bf_trunc(h2);
    b3 = f2bf_trunc(r2);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
struct QStage {          // one k-tile of global loads of one thread (which members are live depends on the mode)
    f32x4 a[4];          // MODE 0/1: a[0] reinterpreted as 16 int8 codes; MODE 3: 16 fp32
    f32x4 b[2];          // MODE 1/3: 8 fp32
    u32x2 bq;            // MODE 0: 8 u8 codes
    int kch;             // IMP dgrad: the channel (reduction row / taps) of this thread's B row -- the index of its delta_w
};
__device__ __forceinline__ void q_load16(f32x4& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void q_load8(u32x2& d, const void* p) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
template <int MODE, int N>
__device__ __forceinline__ void q_wait(QStage& st) {
    if constexpr (MODE == 0)
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(st.a[0]), "+v"(st.bq) : "n"(N) : "memory");
    else if constexpr (MODE == 1)
        asm volatile("s_waitcnt vmcnt(%3)" : "+v"(st.a[0]), "+v"(st.b[0]), "+v"(st.b[1]) : "n"(N) : "memory");
    else
        asm volatile("s_waitcnt vmcnt(%6)"
                     : "+v"(st.a[0]), "+v"(st.a[1]), "+v"(st.a[2]), "+v"(st.a[3]), "+v"(st.b[0]), "+v"(st.b[1])
                     : "n"(N)
                     : "memory");
}

// de-quantised value of a code: two roundings (mul, add), the form of fused_q.hip
__device__ __forceinline__ float dec(unsigned int c, const QRange& r) { return r.delta * (float)c + r.lo; }

struct QGemmArgs {
    // forward / dgrad: A = int8 weight codes [M][K] (k contiguous), B = per-batch [K][ldb] (n contiguous)
    // wgrad          : A = fp32 gz [M][lda] per batch (k = n contiguous), B = u8 codes [N][ldb] per batch
    const void* A;
    const void* B;
    float* C;
    int M, N, K;
    int64_t lda, ldb, ldc;        // row strides (elements)
    int64_t sAb, sBb, sCb;        // batch strides (elements)
    const float* dw;              // [M] fwd: delta_w[co] ; dgrad: [K] delta_w[co] scaling gz rows
    const float* rw;              // [M] fwd: integer row sums of Wi (as float)
    const float* bias;            // [M] fwd
    const float* qmin_x;          // device scalars of the input activation quantizer (fwd, wgrad)
    const float* qmax_x;
    int ksplit, kchunk;           // wgrad split-K
    int tiles_m, tiles_n, batches;   // logical grid (launched 1-D in XCD-aware order, fqss_dev.h)
    // paired layers (two convs on the same input, e.g. res|skip of a TCN block) run as ONE GEMM over the
    // concatenated output channels; the per-layer tensors stay separate in HBM:
    int M1;                          // fwd: rows >= M1 go to C2 / bias2;  wgrad: gz rows >= M1 come from A2
    float* C2; int64_t ldc2, sC2b;
    const float* bias2;
    const void* A2; int64_t lda2, sA2b;
    int K1;                          // dgrad: reduction rows >= K1 (the second layer's gz) come from B2
    const void* B2; int64_t ldb2, sB2b;
    // fwd, optional: the layer's own activation + fake-quant fused into the epilogue -> u8 codes of the OUTPUT
    // (z is still written: the backward needs the pre-quant value); rows >= M1 use the second range
    unsigned char* Q1; unsigned char* Q2; int64_t ldq1, ldq2, sQ1b, sQ2b;
    const float *qy_min1, *qy_max1, *qy_min2, *qy_max2;
    int qact; const float* qslope;
    // fwd + fused quantizer, optional: exact integer statistics (sum c, sum c^2) of the output codes Q1, one slot per workgroup:
    // stats[(b * tiles_m * tiles_n + mt * tiles_n + nt) * 2 + {0,1}] -- what the GroupNorm that consumes Q1 needs (fused_q.hip)
    long long* stats;
    // fwd + fused quantizer, optional (round 5): the AddQ that is the ONLY consumer of output 1 / 2 (the residual add and the skip sum
    // of a TCN block, convtasnetq.py ConvBlock / MaskGenerator) evaluated on the finished output codes: S = fq_s(dec_a(AD) + dec_y(Q)),
    // op for op what k_ewq_fwd computes from the same codes (fused_q.hip), written beside Q -- the AddQ's own launch disappears
    const unsigned char* AD1; const unsigned char* AD2; int64_t ldad1, ldad2;
    unsigned char* S1; unsigned char* S2; int64_t lds1, lds2;
    const float *ad_min1, *ad_max1, *ad_min2, *ad_max2;      // ranges of the other operand's codes
    const float *s_min1, *s_max1, *s_min2, *s_max2;          // ranges of the AddQ's own quantizer
    // IMP kernels (round 6): the B operand of a stride-1 convolution read straight from a halo-packed signal (fqss_halo_pack: per channel
    // one plane of ldb floats, (H + 2 ph) rows of Wp floats, zero halo) -- reduction row k = (channel, tap) = (k / taps, k % taps),
    // B(k, n) = plane[channel][n + base + (tap / kw) * row_step + (tap % kw) * col_step], n = h * Wp + w over the OUTPUT grid (its
    // columns w >= W hold junk the caller never reads).  No masks: the halo holds the zero padding.  The frame image of
    // fqss_frames_gather -- taps x the signal, written and read back -- is never made.  imp_cmax: last column group a load may start at.
    int imp_taps, imp_kw, imp_base, imp_row_step, imp_col_step, imp_cmax;
    float imp_inv_taps, imp_inv_kw;
};

// MODE 0 fwd (int8 A codes, u8 B codes)            1 dgrad (int8 A codes, fp32 B split3)
//      3 plain fp32 x fp32 (A split3 x B split3 = 9 exact products; GP = 2: the six above 2^-24)   (wgrad: k_qwgrad below)
// GP (dgrad only): bf16 pieces of the fp32 gradient operand.  3 = exact products (a = h1 + h2 + h3, truncations): the default.  2 = an
// OPT-IN fast form (FQSS_GRAD_PIECES=2): a ~ h1 + RNE_bf16(a - h1), 16-17 significant bits, |error| <= 2^-16 |a|, unbiased; one third
// fewer MFMAs and LDS bytes: -1.7 / -5.5 us (dgrad), -4 / -6 us (wgrad) = -0.4 ms per cfg-2 step.  Measured against fp64
// (tests/test_gpu_kernels.py::test_gradient_gemms_two_piece_split): the exact form sits at 1.6e-7 .. 3.3e-7 of the result's norm, the
// two-piece form at 4.9e-6 (dgrad) / 8.1e-6 (wgrad) -- 25x the exact form's error, which is why it is not the default and why
// bench.py never sets it.
template <int MODE, int GP = 3, bool IMP = false>
__global__ __launch_bounds__(256, 2) void k_qgemm(QGemmArgs g) {
    static_assert(!IMP || MODE == 1 || MODE == 3 || MODE == 4, "implicit convolution: float B operands");
    // MODE 4 (round 5): forward of a layer whose WEIGHT is on the int8 grid while its input is a plain float tensor (the frame-path
    // convolutions of HTDemucs): A = int8 weight codes, B = fp32 x in three exact bf16 pieces -- the loop of MODE 1 without its
    // delta_w scaling of the reduction rows -- and z = dw[co] * S + b[co] in the epilogue: three products per k instead of six
    static_assert(MODE == 0 || MODE == 1 || MODE == 3 || MODE == 4, "unknown q-GEMM mode");
    static_assert(GP == 3 || (GP == 2 && MODE != 0 && MODE != 4), "two gradient pieces: dgrad only");
    constexpr bool ACODES = MODE < 2 || MODE == 4;          // A arrives as int8 codes
    constexpr int WMODE = MODE == 4 ? 1 : MODE;             // register image / wait form of the load stages
    constexpr bool SIX = MODE == 3 && GP == 2;              // fp32 x fp32 with the six products above 2^-24 instead of all nine
    constexpr int NA = (MODE == 3) ? 3 : 1;                 // A images
    constexpr int NB = (MODE == 1) ? GP : (MODE == 3 || MODE == 4) ? 3 : 1;    // B images
    constexpr int BROWS = QBK, BLD = LDN;
    constexpr int A_BYTES = NA * QBM * LDK * 2, B_BYTES = NB * BROWS * BLD * 2;
    constexpr int T_BYTES = 4 * 32 * LDT * 4;   // epilogue staging: one 32x32 fp32 tile per wave
    constexpr int SMEM = (A_BYTES + B_BYTES > T_BYTES) ? A_BYTES + B_BYTES : T_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    typedef unsigned short (*AsT)[QBM][LDK];
    typedef unsigned short (*BsT)[BROWS][BLD];
    AsT As = reinterpret_cast<AsT>(smem);
    BsT Bs = reinterpret_cast<BsT>(smem + A_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform on purpose: everything derived from it stays in SGPRs
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    // group = one (batch, n-tile) activation panel, re-read by the tiles_m row tiles of the weight
    int panel, mt;
    if (!xcd_tile(g.tiles_n * g.batches, g.tiles_m, panel, mt)) return;
    const int b = panel / g.tiles_n;
    const int kbeg = 0, kend = g.K;
    const int i0 = mt * QBM, j0 = (panel % g.tiles_n) * QBN;

    f32x16 acc[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;

    // per-row epilogue coefficients of this block's 128 rows, staged once in LDS (fetching them per output
    // element from global memory was ~60 % of the forward kernel's time)
    __shared__ float rowc[3][QBM];
    if constexpr (MODE == 0 || MODE == 3 || MODE == 4) {
        if (tid < QBM) {
            const int row = i0 + tid;
            const bool ok = row < g.M;
            rowc[0][tid] = ((MODE == 0 || MODE == 4) && ok) ? g.dw[row] : 1.0f;
            rowc[1][tid] = (MODE == 0 && ok) ? g.rw[row] : 0.0f;
            rowc[2][tid] = (ok && g.bias != nullptr) ? (row < g.M1 ? g.bias[row] : g.bias2[row - g.M1]) : 0.0f;
        }
    }

    // ---------------------------------------------------------------- staging (global -> regs -> LDS)
    // Two register stages of global loads: the tile stored at the end of iteration kt was requested two iterations
    // earlier.  The loads are issued through asm (the compiler sinks plain loads to their first use, which would
    // collapse the stages) and retired by an explicit s_waitcnt carrying the stage's registers; they are
    // unconditional -- addresses clamped into the operand, out-of-range reduction rows zeroed at use (both operands
    // are finite, so a clamped A value only ever meets a zero).
    constexpr int NLOADS = (MODE == 0) ? 2 : (WMODE == 1) ? 3 : 6;   // per thread and stage
    const int a_row = tid >> 1, a_k = (tid & 1) * 16;              // A tile: 128 rows x 32 k, 16 per thread
    const int bk_row = tid >> 3, bk_n = (tid & 7) * 8;             // B tile [k][n]: 32 k x 64 n, 8 per thread
    const int a_row_c = min(i0 + a_row, g.M - 1);

    __shared__ float dws[(MODE == 1) ? 1024 : 1];   // dgrad: delta_w of the reduction rows
    if constexpr (MODE == 1) {
        for (int k = tid; k < (IMP ? g.K / g.imp_taps : g.K); k += 256) dws[k] = g.dw[k];     // (IMP: one per output channel of the conv)
    }

    // the two halves of a paired B operand are selected per lane: keep their descriptors in SGPRs (see sgpr())
    constexpr int BEL = (MODE == 0) ? 1 : 4;   // bytes per B element: codes / fp32 (the strides count elements)
    const char* const gB1 = sgpr((const char*)g.B + (int64_t)b * g.sBb * BEL);
    const char* const gB2 = (MODE == 0) ? gB1 : sgpr((const char*)g.B2 + (int64_t)b * g.sB2b * BEL);
    const int64_t gldb1 = sgpr(g.ldb), gldb2 = (MODE == 0) ? gldb1 : sgpr(g.ldb2);
    auto load_tiles = [&](QStage& st, int k0) {
        if constexpr (ACODES) {
            q_load16(st.a[0], (const signed char*)g.A + (int64_t)a_row_c * g.lda + min(k0 + a_k, g.K - 16));
        } else {
            const float* A = (const float*)g.A + (int64_t)b * g.sAb + (int64_t)a_row_c * g.lda;
#pragma unroll
            for (int q = 0; q < 4; ++q) q_load16(st.a[q], A + min(k0 + a_k + 4 * q, g.K - 4));
        }
        const int kk = min(k0 + bk_row, g.K - 1);
        if constexpr (IMP) {
            // (channel, tap row, tap column) of reduction row kk: two divisions by run-time constants through float reciprocals + one
            // correction step (kk < 2^16: exact)
            int ci = (int)((float)kk * g.imp_inv_taps);
            int t = kk - ci * g.imp_taps;
            if (t < 0) { ci -= 1; t += g.imp_taps; }
            if (t >= g.imp_taps) { ci += 1; t -= g.imp_taps; }
            int th = (int)((float)t * g.imp_inv_kw);
            int tw = t - th * g.imp_kw;
            if (tw < 0) { th -= 1; tw += g.imp_kw; }
            if (tw >= g.imp_kw) { th += 1; tw -= g.imp_kw; }
            st.kch = ci;
            const float* Bp = (const float*)gB1 + (int64_t)ci * gldb1 + (g.imp_base + th * g.imp_row_step + tw * g.imp_col_step);
            const int c0 = min(j0 + bk_n, g.imp_cmax);               // (4-B aligned 16-B requests: a tap's shift breaks the alignment)
            q_load16(st.b[0], Bp + c0);
            q_load16(st.b[1], Bp + c0 + 4);
        } else if constexpr (MODE == 0) {
            const unsigned char* Bp = (const unsigned char*)gB1 + (int64_t)kk * gldb1;
            q_load8(st.bq, Bp + min(j0 + bk_n, (int)gldb1 - 8));   // groups past the row (columns >= N) re-read its tail
        } else {
            const bool h1 = kk < g.K1;
            const int ldk = (int)(h1 ? gldb1 : gldb2);              // row length (multiple of 4, >= N): clamp inside the row;
            const float* Bp = (const float*)(h1 ? gB1 : gB2) + (int64_t)(h1 ? kk : kk - g.K1) * ldk;
            q_load16(st.b[0], Bp + min(j0 + bk_n, ldk - 4));        // a clamped group only feeds output columns >= N
            q_load16(st.b[1], Bp + min(j0 + bk_n + 4, ldk - 4));
        }
    };

    // n 8-bit integers (n = 16: one uint4, n = 8: one uint2) -> bf16 (exact), 16-B LDS stores
    auto store_codes = [&](unsigned short* dst, const unsigned int* w, auto nw_tag, bool is_signed, bool zero) {
        constexpr int NW = decltype(nw_tag)::value;
        unsigned int o[2 * NW];
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            float f[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned int byte = (w[q] >> (8 * e)) & 0xFFu;
                f[e] = zero ? 0.0f : (is_signed ? (float)(int)(signed char)byte : (float)byte);
            }
            o[2 * q] = __builtin_amdgcn_perm(__float_as_uint(f[1]), __float_as_uint(f[0]), 0x07060302u);
            o[2 * q + 1] = __builtin_amdgcn_perm(__float_as_uint(f[3]), __float_as_uint(f[2]), 0x07060302u);
        }
#pragma unroll
        for (int h = 0; h < NW / 2; ++h)
            *reinterpret_cast<uint4*>(dst + 8 * h) = make_uint4(o[4 * h], o[4 * h + 1], o[4 * h + 2], o[4 * h + 3]);
    };

    auto store_split3 = [&](unsigned short* d1, unsigned short* d2, unsigned short* d3, const f32x4* v, float scale,
                            bool do_scale, bool zero, auto nq_tag) {
        constexpr int NQ = decltype(nq_tag)::value;   // float4 count: 4 (16 values) or 2 (8 values)
        unsigned int o1[2 * NQ], o2[2 * NQ], o3[2 * NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float h0[4], r1[4], r2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = do_scale ? scale * v[q][e] : v[q][e];
                if (zero) t = 0.0f;
                h0[e] = t;
                r1[e] = t - bf_trunc(t);
                r2[e] = r1[e] - bf_trunc(r1[e]);
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {   // bf16 pairs: v_perm picks the high halves of two fp32 words
                o1[2 * q + e] = __builtin_amdgcn_perm(__float_as_uint(h0[2 * e + 1]), __float_as_uint(h0[2 * e]), 0x07060302u);
                o2[2 * q + e] = __builtin_amdgcn_perm(__float_as_uint(r1[2 * e + 1]), __float_as_uint(r1[2 * e]), 0x07060302u);
                o3[2 * q + e] = __builtin_amdgcn_perm(__float_as_uint(r2[2 * e + 1]), __float_as_uint(r2[2 * e]), 0x07060302u);
            }
        }
#pragma unroll
        for (int h = 0; h < NQ / 2; ++h) {
            *reinterpret_cast<uint4*>(d1 + 8 * h) = make_uint4(o1[4 * h], o1[4 * h + 1], o1[4 * h + 2], o1[4 * h + 3]);
            *reinterpret_cast<uint4*>(d2 + 8 * h) = make_uint4(o2[4 * h], o2[4 * h + 1], o2[4 * h + 2], o2[4 * h + 3]);
            *reinterpret_cast<uint4*>(d3 + 8 * h) = make_uint4(o3[4 * h], o3[4 * h + 1], o3[4 * h + 2], o3[4 * h + 3]);
        }
    };

    // two-piece form of store_split3 for the gradient operand: truncated head + round-to-nearest remainder
    auto store_split2 = [&](unsigned short* d1, unsigned short* d2, const f32x4* v, float scale, bool zero) {
        unsigned int o1[4], o2[4];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float h0[4];
            unsigned int r1[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = scale * v[q][e];
                if (zero) t = 0.0f;
                h0[e] = t;
                r1[e] = bf_rne_word(t - bf_trunc(t));
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                o1[2 * q + e] = __builtin_amdgcn_perm(__float_as_uint(h0[2 * e + 1]), __float_as_uint(h0[2 * e]), 0x07060302u);
                o2[2 * q + e] = __builtin_amdgcn_perm(r1[2 * e + 1], r1[2 * e], 0x07060302u);
            }
        }
        *reinterpret_cast<uint4*>(d1) = make_uint4(o1[0], o1[1], o1[2], o1[3]);
        *reinterpret_cast<uint4*>(d2) = make_uint4(o2[0], o2[1], o2[2], o2[3]);
    };

    auto store_tiles = [&](QStage& st, int k0) {
        q_wait<WMODE, NLOADS>(st);   // this stage has landed; the NLOADS younger requests of the other stage stay in flight
        const bool kz = (k0 + bk_row) >= g.K;   // reduction rows past K contribute zeros (B side)
        if constexpr (ACODES) {
            const unsigned int w[4] = {__float_as_uint(st.a[0][0]), __float_as_uint(st.a[0][1]), __float_as_uint(st.a[0][2]),
                                       __float_as_uint(st.a[0][3])};
            store_codes(&As[0][a_row][a_k], w, std::integral_constant<int, 4>{}, true, false);
        } else {
            store_split3(&As[0][a_row][a_k], &As[1][a_row][a_k], &As[2][a_row][a_k], st.a, 1.0f, false, false,
                         std::integral_constant<int, 4>{});
        }
        if constexpr (MODE == 0) {
            const unsigned int w[2] = {st.bq[0], st.bq[1]};
            store_codes(&Bs[0][bk_row][bk_n], w, std::integral_constant<int, 2>{}, false, kz);
        } else {
            const float sc = (MODE == 1) ? dws[IMP ? st.kch : min(k0 + bk_row, g.K - 1)] : 1.0f;
            if constexpr (NB == 2)
                store_split2(&Bs[0][bk_row][bk_n], &Bs[1][bk_row][bk_n], st.b, sc, kz);
            else
                store_split3(&Bs[0][bk_row][bk_n], &Bs[1][bk_row][bk_n], &Bs[2][bk_row][bk_n], st.b, sc, MODE == 1, kz,
                             std::integral_constant<int, 2>{});
        }
    };

    // ---------------------------------------------------------------- main loop
    const int nkt = (kend - kbeg + QBK - 1) / QBK;
    // lane geometry of the transposed read: 16-lane group gq reads a 4(k) x 16(n) block
    const int gq = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    auto compute_tile = [&]() {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int sp = 0; sp < NA * NB; ++sp) {
                // smallest pieces first: (3,3) ... (1,1) so that the large products are added last
                const int ia = (NA == 3) ? 2 - (sp / NB) : 0, ib = (NB > 1) ? NB - 1 - (sp % NB) : 0;
                // SIX: m.l, l.m and l.l lie below 2^-24 of the product (under the fp32 rounding of the sum), as csrc/gemm_x3.hip
                if (SIX && ia + ib >= 3) continue;
                bf16x8 af[2], bfr;
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    af[mi] = *reinterpret_cast<const bf16x8*>(&As[ia][wm * 64 + mi * 32 + lr][ks * 16 + 8 * lh]);
                {
                    // B[k = 8h + j][col r]: two transposed 4x16 block reads (rows 8h+0..3 and 8h+4..7)
                    const int kr = ks * 16 + 8 * (gq >> 1) + tq;
                    const int nc = wn * 32 + 16 * (gq & 1) + 4 * tp;
                    union { bf16x8 v; s16x4 h[2]; } u;
                    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (s16x4 __attribute__((address_space(3)))*)(&Bs[ib][kr][nc]));
                    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (s16x4 __attribute__((address_space(3)))*)(&Bs[ib][kr + 4][nc]));
                    bfr = u.v;
                }
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bfr, acc[mi], 0, 0, 0);
            }
        }
    };
    QStage stA, stB;
    __syncthreads();   // dws / rowc
    load_tiles(stA, 0);
    load_tiles(stB, QBK);
    store_tiles(stA, 0);
    load_tiles(stA, 2 * QBK);
    __syncthreads();
    // invariant at the top of iteration kt (even): LDS = tile kt, stB = tile kt+1, stA = tile kt+2 (requests in flight)
    for (int kt = 0; kt < nkt; kt += 2) {
        compute_tile();
        __syncthreads();
        if (kt + 1 < nkt) {
            store_tiles(stB, (kt + 1) * QBK);
            load_tiles(stB, (kt + 3) * QBK);
            __syncthreads();
            compute_tile();
            __syncthreads();
            if (kt + 2 < nkt) {
                store_tiles(stA, (kt + 2) * QBK);
                load_tiles(stA, (kt + 4) * QBK);
                __syncthreads();
            }
        }
    }
    // the trailing (clamped, unused) requests still target the stage registers, which the compiler considers dead
    // from here on: drain them before anything else is allocated there
    q_wait<WMODE, 0>(stA);
    q_wait<WMODE, 0>(stB);

    // ---------------------------------------------------------------- epilogue
    float dx = 0.f, mnx = 0.f;
    if constexpr (MODE == 0) {
        const float lo = *g.qmin_x, hi = *g.qmax_x;
        dx = (hi - lo) / 255.0f;
        mnx = lo;
    }
    float* Cb = g.C + (int64_t)b * g.sCb;
    float* C2b = (g.C2 != nullptr) ? g.C2 + (int64_t)b * g.sC2b : nullptr;
    __shared__ uint32_t Qt[(MODE == 0) ? 4 : 1][32][8];   // per-wave 32 x 32 tile of output codes
    const bool quant = (MODE == 0) && g.Q1 != nullptr;
    unsigned int st_s = 0, st_ss = 0;     // statistics of this lane's output codes (< 2^32 for a whole 128 x 64 tile)
    QRange ry1{}, ry2{};
    QRange rad1{}, rad2{}, rs1{}, rs2{};     // fused AddQ behind output 1 / 2: the other operand's range, the sum's range
    float qslope = 0.0f;
    if (quant) {
        ry1 = load_qrange(g.qy_min1, g.qy_max1);
        ry2 = (g.Q2 != nullptr) ? load_qrange(g.qy_min2, g.qy_max2) : ry1;
        qslope = (g.qact == FQSS_ACT_PRELU) ? *g.qslope : 0.0f;
        if (g.S1 != nullptr) { rad1 = load_qrange(g.ad_min1, g.ad_max1); rs1 = load_qrange(g.s_min1, g.s_max1); }
        if (g.S2 != nullptr) { rad2 = load_qrange(g.ad_min2, g.ad_max2); rs2 = load_qrange(g.s_min2, g.s_max2); }
    }
    {
        // stage each 32x32 accumulator tile through LDS and store whole 128-B rows with 16 B per lane
        // (the lane-per-column layout of the MFMA result would need 16 strided 4-B stores per tile: that
        // store-issue-bound tail was 60 % of the forward kernel's time)
        __syncthreads();   // every wave is done with As/Bs
        float(*Tt)[LDT] = reinterpret_cast<float(*)[LDT]>(smem + wave * 32 * LDT * 4);
        const bool has_bias = g.bias != nullptr;
        const float qns = act_neg_scale(g.qact, qslope);
        const int c4 = (lane & 7) * 4;
        const int col = j0 + wn * 32 + c4;
        // PER_ROW = false: M1 % 32 == 0 (every shape of the real model), a 32-row tile lies on one side of the output
        // split and the destination is selected once per tile; PER_ROW = true: the general case, selected per stored row
        // (the per-element selects and branches of a single generic path were half the epilogue's code).  The fused
        // output quantizer always has M1 % 32 == 0 (host check).
        auto tile_epilogue = [&](int mi, auto PER_ROW) {
            const int rowt = i0 + wm * 64 + mi * 32;
            const bool tfirst = rowt < g.M1;
            // fused AddQ: request this lane's 16 codes of the other operand now, they are used behind the tile's own codes
            uint4 ad16 = make_uint4(0u, 0u, 0u, 0u);
            const unsigned char* const adp = (MODE == 0 && quant) ? (tfirst ? g.AD1 : g.AD2) : nullptr;
            if constexpr (MODE == 0) {
                if (adp != nullptr) {
                    const int64_t ldad = tfirst ? g.ldad1 : g.ldad2;
                    const int orow = rowt + (lane >> 1) - (tfirst ? 0 : g.M1), ocol = j0 + wn * 32 + 16 * (lane & 1);
                    const int rows_o = tfirst ? g.M1 : g.M - g.M1;
                    if (rowt + (lane >> 1) < g.M && ocol < ldad)
                        ad16 = *reinterpret_cast<const uint4*>(adp + ((int64_t)b * rows_o + orow) * ldad + ocol);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int rb = wm * 64 + mi * 32 + rl;   // row within the block
                const float S = acc[mi][r];
                float v = S;
                if constexpr (MODE == 0) {
                    v = rowc[0][rb] * (dx * S + mnx * rowc[1][rb]);
                    if (has_bias) v = v + rowc[2][rb];
                } else if constexpr (MODE == 3) {
                    if (has_bias) v = S + rowc[2][rb];
                } else if constexpr (MODE == 4) {
                    v = rowc[0][rb] * S;
                    if (has_bias) v = v + rowc[2][rb];
                }
                Tt[rl][lr] = v;
            }
            // same-wave LDS round trip: program order + the compiler's lgkmcnt waits are sufficient
            const QRange ry = tfirst ? ry1 : ry2;
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int rl = pass * 8 + (lane >> 3);
                const int row = rowt + rl;
                float4 t = *reinterpret_cast<const float4*>(&Tt[rl][c4]);
                if constexpr (MODE == 1) {
                    // dgrad, optional: + an addend of the output's shape (the gradient of the other branch of a residual fork: the sum
                    // that autograd would take in a pass of its own); rows are padded to a multiple of 4, so the float4 is in bounds
                    if (C2b != nullptr && row < g.M && col < g.N) {
                        const float4 r = *reinterpret_cast<const float4*>(C2b + (int64_t)row * g.ldc2 + col);
                        t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w;
                    }
                }
                if (row < g.M && col < g.N) {
                    const bool first = decltype(PER_ROW)::value ? row < g.M1 : tfirst;
                    const int64_t ldc = first ? g.ldc : g.ldc2;
                    float* dst = (first ? Cb + (int64_t)row * ldc : C2b + (int64_t)(row - g.M1) * ldc) + col;
                    if (col + 3 < g.N || col + 3 < ldc) {
                        store16(dst, t);   // columns >= N fall into the row padding
                    } else {
                        dst[0] = t.x;
                        if (col + 1 < g.N) dst[1] = t.y;
                        if (col + 2 < g.N) dst[2] = t.z;
                    }
                }
                if constexpr (MODE == 0) {
                    if (quant) {   // this lane's 4 outputs -> 4 codes
                        const float tv[4] = {t.x, t.y, t.z, t.w};
                        uint32_t pk = 0;
#pragma unroll
                        for (int e = 0; e < 4; ++e) pk = pack_code(fq_code(tv[e] > 0.0f ? tv[e] : qns * tv[e], ry), e, pk);
                        Qt[wave][rl][lane & 7] = pk;
                        if (g.stats != nullptr) {   // statistics over the live positions (row < M, column < N) only
                            const int nl = g.N - col;
                            const uint32_t live = (row < g.M && nl > 0) ? (nl >= 4 ? 0xFFFFFFFFu : (0xFFFFFFFFu >> (8 * (4 - nl)))) : 0u;
                            code_stats4(pk & live, st_s, st_ss);
                        }
                    }
                }
            }
            if constexpr (MODE == 0) {
                if (quant) {   // 16 codes (16 B) per lane: lane -> (row = lane / 2, half row)
                    const int rl = lane >> 1, hf = lane & 1;
                    const int qcol = j0 + wn * 32 + 16 * hf;
                    const uint4 c16 = *reinterpret_cast<const uint4*>(&Qt[wave][rl][4 * hf]);
                    const int64_t ldq = tfirst ? g.ldq1 : g.ldq2;
                    if (rowt + rl < g.M && qcol < ldq) {
                        unsigned char* qb = tfirst ? g.Q1 + (int64_t)b * g.sQ1b + (int64_t)(rowt + rl) * ldq
                                                   : g.Q2 + (int64_t)b * g.sQ2b + (int64_t)(rowt + rl - g.M1) * ldq;
                        *reinterpret_cast<uint4*>(qb + qcol) = c16;
                    }
                    if (adp != nullptr) {   // wave-uniform.  S = fq(dec(a) + dec(y)): k_ewq_fwd's arithmetic (sb = 1, no activation)
                        const QRange ra = tfirst ? rad1 : rad2, rs = tfirst ? rs1 : rs2;
                        const unsigned int wa[4] = {ad16.x, ad16.y, ad16.z, ad16.w}, wy[4] = {c16.x, c16.y, c16.z, c16.w};
                        unsigned int o[4];
#pragma unroll
                        for (int w4 = 0; w4 < 4; ++w4) {
                            unsigned int pk = 0;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float z = dec((wa[w4] >> (8 * e)) & 255u, ra);
                                z = z + 1.0f * dec((wy[w4] >> (8 * e)) & 255u, ry);
                                pk = pack_code(fq_code(z, rs), e, pk);
                            }
                            o[w4] = pk;
                        }
                        const int64_t lds_ = tfirst ? g.lds1 : g.lds2;
                        const int rows_o = tfirst ? g.M1 : g.M - g.M1;
                        if (rowt + rl < g.M && qcol < lds_) {
                            unsigned char* sb_ = (tfirst ? g.S1 : g.S2) + ((int64_t)b * rows_o + (rowt + rl - (tfirst ? 0 : g.M1))) * lds_;
                            *reinterpret_cast<uint4*>(sb_ + qcol) = make_uint4(o[0], o[1], o[2], o[3]);
                        }
                    }
                }
            }
        };
        if (MODE != 0 || (g.M1 & 31) == 0 || g.M1 >= g.M) {   // only the forward GEMM has paired outputs
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) tile_epilogue(mi, std::false_type{});
        } else {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) tile_epilogue(mi, std::true_type{});
        }
        if constexpr (MODE == 0) {
            if (quant && g.stats != nullptr) {   // workgroup-uniform: one (sum c, sum c^2) slot per workgroup, plain stores
                __shared__ unsigned int sst[2][4];
                st_s = wave_sum(st_s);
                st_ss = wave_sum(st_ss);
                if (lane == 0) { sst[0][wave] = st_s; sst[1][wave] = st_ss; }
                __syncthreads();
                if (tid == 0) {
                    long long* slot = g.stats + 2 * (((int64_t)b * g.tiles_m + mt) * g.tiles_n + (panel % g.tiles_n));
                    slot[0] = (long long)sst[0][0] + sst[0][1] + sst[0][2] + sst[0][3];
                    slot[1] = (long long)sst[1][0] + sst[1][1] + sst[1][2] + sst[1][3];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// wgrad, register-direct:  gW_q[co][ci] += dx * sum_n gz[co][n]*c[ci][n] + min_x * sum_n gz[co][n]
// Both operands are n-contiguous in HBM and n is the reduction index, so every lane can load its MFMA
// fragment straight from global memory: lane (row lr, half lh) of a 32-wide k chunk owns the 16
// consecutive positions n0+16*lh .. +15 of "its" gz row (64 B) and of "its" code row (16 B); the two k-steps
// of the chunk use the first / second 8 of them for A and B alike (the k labelling inside an MFMA is
// arbitrary as long as A and B agree).  No LDS, no barriers: each wave free-runs over its n range with a
// ring of WG_STAGES chunk loads in flight (an LDS-tiled version of this kernel was load-latency bound at
// 2x the time: 12 exposed global-load round trips per workgroup, two barriers each).
// Wave tile 32(co) x 64(ci); workgroup = 2x2 waves = 64 x 128; grid.z = batch x n-split.
constexpr int WG_STAGES = 4;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct WgStage {
    f32x4 a[4];   // 16 gz values of this lane's row
    u32x4 c[2];   // 16 codes for each of the two 32-column tiles
};
// The ring is hand-scheduled: the compiler sinks plain loads next to their first use (draining the ring to
// vmcnt(0) every chunk), so the loads are issued through asm and retired with an explicit s_waitcnt that
// carries the stage's registers as in/out operands (every use of the data is ordered after the wait).
__device__ __forceinline__ void wg_load16(f32x4& d, const void* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory");
}
__device__ __forceinline__ void wg_load16(u32x4& d, const void* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void wg_wait(WgStage& st) {
    asm volatile("s_waitcnt vmcnt(%6)"
                 : "+v"(st.a[0]), "+v"(st.a[1]), "+v"(st.a[2]), "+v"(st.a[3]), "+v"(st.c[0]), "+v"(st.c[1])
                 : "n"(N)
                 : "memory");
}

__global__ __launch_bounds__(256, 2) void k_qwgrad(QGemmArgs g) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform on purpose: everything derived from it stays in SGPRs
    const int wr = wave >> 1, wc = wave & 1, lr = lane & 31, lh = lane >> 5;
    // group = one (batch, n-slice): its tiles_m x tiles_n workgroups share the gz rows / code rows of that slice
    int slice, t;
    if (!xcd_tile(g.batches * g.ksplit, g.tiles_m * g.tiles_n, slice, t)) return;
    const int b = slice / g.ksplit, ks_id = slice % g.ksplit;
    const int kbeg = ks_id * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    const int row0 = (t / g.tiles_n) * 64 + wr * 32, col0 = (t % g.tiles_n) * 128 + wc * 64;
    const int nchunks = (kend - kbeg + 31) >> 5;

    const bool arow_ok = row0 + lr < g.M;
    const int arow = arow_ok ? row0 + lr : 0;
    const float* Ap = (arow < g.M1) ? (const float*)g.A + (int64_t)b * g.sAb + (int64_t)arow * g.lda
                                    : (const float*)g.A2 + (int64_t)b * g.sA2b + (int64_t)(arow - g.M1) * g.lda2;
    const unsigned char* Bb = (const unsigned char*)g.B + (int64_t)b * g.sBb;
    const unsigned char* Bp[2] = {Bb + (int64_t)min(col0 + lr, g.N - 1) * g.ldb, Bb + (int64_t)min(col0 + 32 + lr, g.N - 1) * g.ldb};

    // Loads are unconditional: addresses are clamped into the row, values masked at use.
    const int ka_last = (kend - 1) & ~3, kb_last = (kend - 1) & ~15;
    auto load = [&](WgStage& st, int chunk) {
        const int k = kbeg + chunk * 32 + 16 * lh;
#pragma unroll
        for (int q = 0; q < 4; ++q) wg_load16(st.a[q], Ap + min(k + 4 * q, ka_last));
#pragma unroll
        for (int t = 0; t < 2; ++t) wg_load16(st.c[t], Bp[t] + min(k, kb_last));
    };

    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float rowsum = 0.0f;

    auto compute = [&](WgStage& st, int chunk) {
        float x[16];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) x[4 * q + e] = st.a[q][e];
        // positions >= kend (row padding, clamped re-reads) and rows >= Co contribute exact zeros; codes need no
        // mask (finite, and they only ever meet a zero)
        const int nvalid = arow_ok ? kend - (kbeg + chunk * 32 + 16 * lh) : 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) x[e] = (e < nvalid) ? x[e] : 0.0f;
        // exact 3-way split, packed as bf16 pairs with v_perm (high halves of two fp32 words)
        union { uint32_t u[8]; bf16x8 v[2]; } A1, A2, A3;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float a0 = x[2 * e], a1 = x[2 * e + 1];
            rowsum += a0 + a1;
            const float r0 = a0 - bf_trunc(a0), r1 = a1 - bf_trunc(a1);
            const float s0 = r0 - bf_trunc(r0), s1 = r1 - bf_trunc(r1);
            A1.u[e] = __builtin_amdgcn_perm(__float_as_uint(a1), __float_as_uint(a0), 0x07060302u);
            A2.u[e] = __builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), 0x07060302u);
            A3.u[e] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint32_t w[4] = {st.c[t][0], st.c[t][1], st.c[t][2], st.c[t][3]};
            union { uint32_t u[8]; bf16x8 v[2]; } Bc;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float f0 = (float)(w[q] & 0xFFu), f1 = (float)((w[q] >> 8) & 0xFFu);
                const float f2 = (float)((w[q] >> 16) & 0xFFu), f3 = (float)(w[q] >> 24);
                Bc.u[2 * q] = __builtin_amdgcn_perm(__float_as_uint(f1), __float_as_uint(f0), 0x07060302u);
                Bc.u[2 * q + 1] = __builtin_amdgcn_perm(__float_as_uint(f3), __float_as_uint(f2), 0x07060302u);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {   // smallest pieces first
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A3.v[s], Bc.v[s], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2.v[s], Bc.v[s], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1.v[s], Bc.v[s], acc[t], 0, 0, 0);
            }
        }
    };

    WgStage st[WG_STAGES];
#pragma unroll
    for (int s = 0; s < WG_STAGES - 1; ++s) load(st[s], s);
    for (int c0 = 0; c0 < nchunks; c0 += WG_STAGES) {
#pragma unroll
        for (int s = 0; s < WG_STAGES; ++s) {
            load(st[(s + WG_STAGES - 1) % WG_STAGES], c0 + s + WG_STAGES - 1);   // 4 stages x 6 loads in flight ...
            wg_wait<6 * (WG_STAGES - 1)>(st[s]);                                 // ... the oldest stage has landed
            compute(st[s], c0 + s);   // chunks past the end are fully masked: the body stays branch-free
        }
    }

    // the ring's trailing (clamped, unused) loads still target the stage registers, which the compiler
    // considers dead from here on: drain them before anything else may be allocated there
#pragma unroll
    for (int s = 0; s < WG_STAGES; ++s) wg_wait<0>(st[s]);   // in/out operands keep every stage register reserved up to here

    // epilogue: the gz row sums (min_x term) live in lane == row; results are atomically added (n-split, batch)
    const float lo = *g.qmin_x, hi = *g.qmax_x;
    const float dx = (hi - lo) / 255.0f, mnx = lo;
    const float rtot = rowsum + __shfl_xor(rowsum, 32, 64);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rl = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float rsum = __shfl(rtot, rl, 64);
        const int row = row0 + rl;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int col = col0 + 32 * t + lr;
            if (row < g.M && col < g.N) grad_add(&g.C[(int64_t)row * g.ldc + col], dx * acc[t][r] + mnx * rsum);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// wgrad, LDS-tiled (round 2): the register-direct kernel above is VALU-issue bound (PMC: 47 % of wave time issuing, 39 % stalled
// on issue; ~5 cycles per wave instruction and SIMD): every wave splits its own gz fragment and converts its own code fragments,
// so inside a 2 x 2 workgroup each gz value is split twice and each code converted twice.  Here the workgroup converts a
// 64 (co) x 128 (ci) x 64 (n) stage ONCE -- coalesced 16-B global loads (full 256-B / 64-B row pieces), exact 3-way bf16 split /
// u8 -> bf16 at LDS-store time -- and the four waves read k-contiguous fragments with ds_read_b128 (both operands are
// n-contiguous and n is the reduction index: no transposed reads).  Half the VALU work per position of the first form; a ring of four
// register stages of asm-issued loads stays in flight across the barriers.  Same arithmetic: exact products, fp32 accumulation.
// Measured (cfg-2 shapes, cold operands): 39 -> 31 us (128->512: 2.2 TB/s) and 59 -> 47 us (pair).  Ablation of the 31 us: the
// operand stream alone 17 us (4.1 TB/s: at the practical HBM rate), + conversion 1.5, + MFMAs 6, + the 2.1 M float atomics of the
// split-n reduction 7.6 -- the three do not overlap the stream yet (one 8-wave workgroup per CU).  The pair shape re-reads each
// operand tile four times through L2 (64 x 128 tiles over 256 x 512 outputs): its stream alone is 23 us.
constexpr int W2_TM = 64, W2_TN = 128, W2_TK = 64, W2_LD = 72, W2_RING = 4;   // 72 bf16 = 144 B rows: conflict-free ds_read_b128 over 16 rows
struct W2Stage {
    f32x4 a[2];   // 2 rows x 4 consecutive gz values
    u32x4 b;      // 16 consecutive codes of one row
};
template <int N>
__device__ __forceinline__ void w2_wait(W2Stage& st) {
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(st.a[0]), "+v"(st.a[1]), "+v"(st.b) : "n"(N) : "memory");
}

// 512 threads = 8 waves as 2 (co) x 4 (ci), each a 32 x 32 output tile: two waves per SIMD, so one wave's conversion (VALU) runs
// beside the other's MFMAs (a 4-wave form, one wave per SIMD, serialised the two phases: 34 us instead of the 39 us it replaced)
template <int GP>   // bf16 pieces of gz: 3 exact (default), 2 = head + round-to-nearest remainder (opt-in, see k_qgemm)
__global__ __launch_bounds__(512, 1) void k_qwgrad2(QGemmArgs g) {
    // two LDS stage buffers: a wave converts stage s+1 into one while it (and its SIMD partner) multiply stage s out of the other,
    // ONE barrier per stage.  (Single-buffered, two barriers per stage put all eight waves into the same phase at the same time:
    // VALU and matrix pipe took turns -- PMC: 49 % of the wave time waiting -- and the kernel ran at 30 / 45 us.)
    __shared__ __attribute__((aligned(16))) unsigned short As[2][GP][W2_TM][W2_LD];  // 2 x GP x 9,216 B: the bf16 pieces of gz
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2][W2_TN][W2_LD];      // 2 x 18,432 B: codes as bf16
    __shared__ float rsum[W2_TM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3, lr = lane & 31, lh = lane >> 5;
    int slice, t;
    if (!xcd_tile(g.batches * g.ksplit, g.tiles_m * g.tiles_n, slice, t)) return;
    const int b = slice / g.ksplit, ks_id = slice % g.ksplit;
    const int kbeg = ks_id * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    const int row0 = (t / g.tiles_n) * W2_TM, col0 = (t % g.tiles_n) * W2_TN;
    const int nst = (kend - kbeg + W2_TK - 1) / W2_TK;

    // loader geometry: A 16 threads per row (4 values each), rows ar and ar + 32; B 4 threads per row (16 codes each), row br
    const int ar = tid >> 4, ac = (tid & 15) * 4;
    const int br = tid >> 2, bc = (tid & 3) * 16;
    const float* Ap[2];
    bool aok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = row0 + ar + 32 * i;
        aok[i] = row < g.M;
        const int rc = aok[i] ? row : 0;
        Ap[i] = (rc < g.M1) ? (const float*)g.A + (int64_t)b * g.sAb + (int64_t)rc * g.lda
                            : (const float*)g.A2 + (int64_t)b * g.sA2b + (int64_t)(rc - g.M1) * g.lda2;
    }
    const unsigned char* Bp = (const unsigned char*)g.B + (int64_t)b * g.sBb + (int64_t)min(col0 + br, g.N - 1) * g.ldb;
    const int ka_last = (kend - 1) & ~3, kb_last = (kend - 1) & ~15;   // loads are unconditional: clamped into the row, masked at use

    auto load = [&](W2Stage& st, int s) {
        const int k = kbeg + s * W2_TK;
#pragma unroll
        for (int i = 0; i < 2; ++i) wg_load16(st.a[i], Ap[i] + min(k + ac, ka_last));
        wg_load16(st.b, Bp + min(k + bc, kb_last));
    };
    float rs_part[2] = {0.f, 0.f};
    auto convert_store = [&](W2Stage& st, int s) {
        const int buf = s & 1;
        const int k = kbeg + s * W2_TK + ac;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float x[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = (aok[i] && k + e < kend) ? st.a[i][e] : 0.0f;
            rs_part[i] += (x[0] + x[1]) + (x[2] + x[3]);
            uint32_t o1[2], o2[2], o3[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float a0 = x[2 * e], a1 = x[2 * e + 1];
                const float r0 = a0 - bf_trunc(a0), r1 = a1 - bf_trunc(a1);
                o1[e] = __builtin_amdgcn_perm(__float_as_uint(a1), __float_as_uint(a0), 0x07060302u);
                if constexpr (GP == 3) {
                    const float s0 = r0 - bf_trunc(r0), s1 = r1 - bf_trunc(r1);
                    o2[e] = __builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), 0x07060302u);
                    o3[e] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
                } else {
                    o2[e] = __builtin_amdgcn_perm(bf_rne_word(r1), bf_rne_word(r0), 0x07060302u);
                }
            }
            const int row = ar + 32 * i;
            *reinterpret_cast<uint2*>(&As[buf][0][row][ac]) = make_uint2(o1[0], o1[1]);
            *reinterpret_cast<uint2*>(&As[buf][1][row][ac]) = make_uint2(o2[0], o2[1]);
            if constexpr (GP == 3) *reinterpret_cast<uint2*>(&As[buf][2][row][ac]) = make_uint2(o3[0], o3[1]);
        }
        {   // codes need no mask: finite, and beyond kend they only ever meet a zero
            uint32_t o[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t w = st.b[q];
                const float f0 = (float)(w & 0xFFu), f1 = (float)((w >> 8) & 0xFFu);
                const float f2 = (float)((w >> 16) & 0xFFu), f3 = (float)(w >> 24);
                o[2 * q] = __builtin_amdgcn_perm(__float_as_uint(f1), __float_as_uint(f0), 0x07060302u);
                o[2 * q + 1] = __builtin_amdgcn_perm(__float_as_uint(f3), __float_as_uint(f2), 0x07060302u);
            }
            *reinterpret_cast<uint4*>(&Bs[buf][br][bc]) = make_uint4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<uint4*>(&Bs[buf][br][bc + 8]) = make_uint4(o[4], o[5], o[6], o[7]);
        }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    auto compute = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < W2_TK / 16; ++ks) {
            const int kk = ks * 16 + 8 * lh;
            const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(&Bs[buf][wc * 32 + lr][kk]);
            bf16x8 af[GP];
#pragma unroll
            for (int p = 0; p < GP; ++p) af[p] = *reinterpret_cast<const bf16x8*>(&As[buf][p][wr * 32 + lr][kk]);
#pragma unroll
            for (int p = GP - 1; p >= 0; --p) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[p], bfr, acc, 0, 0, 0);   // smallest pieces first
        }
    };

    // ring of W2_RING register stages (24 KB of requests per stage and workgroup): with two stages the loop ran at one memory
    // round trip per stage (1.5-1.9 us: 47 us for the pair shape); four keep ~96 KB per CU in flight
    W2Stage st[W2_RING];
#pragma unroll
    for (int i = 0; i < W2_RING; ++i) load(st[i], i);
    w2_wait<3 * (W2_RING - 1)>(st[0]);
    convert_store(st[0], 0);
    load(st[0], W2_RING);
    __syncthreads();
    const bool first_half = wave < 4;   // waves 4-7 (the SIMD partners of 0-3) take the two halves of a stage in the other order
    for (int s = 0; s < nst; s += W2_RING) {   // host: kchunk is a multiple of W2_RING stages; stages past nst are fully masked
#pragma unroll
        for (int i = 0; i < W2_RING; ++i) {
            // between two barriers every wave multiplies stage s+i (buffer (s+i) & 1) and converts stage s+i+1 into the other buffer
            W2Stage& nx = st[(i + 1) % W2_RING];
            if (first_half) compute((s + i) & 1);
            w2_wait<3 * (W2_RING - 1)>(nx);      // the oldest stage in flight has landed
            convert_store(nx, s + i + 1);
            load(nx, s + i + 1 + W2_RING);
            if (!first_half) compute((s + i) & 1);
            __syncthreads();
        }
    }
#pragma unroll
    for (int i = 0; i < W2_RING; ++i) w2_wait<0>(st[i]);   // drain the trailing (clamped, unused) requests before their registers are reused

    // gz row sums (min_x term): the 16 threads of a row group hold the partials of rows ar and ar + 32
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float v = rs_part[i];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((tid & 15) == 0) rsum[ar + 32 * i] = v;
    }
    __syncthreads();
    const float lo = *g.qmin_x, hi = *g.qmax_x;
    const float dx = (hi - lo) / 255.0f, mnx = lo;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rl = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int row = row0 + wr * 32 + rl;
        const int col = col0 + wc * 32 + lr;
        if (row < g.M && col < g.N) grad_add(&g.C[(int64_t)row * g.ldc + col], dx * acc[r] + mnx * rsum[wr * 32 + rl]);
    }
}

// ---------------------------------------------------------------------------------------------------
// wgrad, GROUPED (round 5): the weight gradients of SEVERAL layers in one launch, no atomics, bit-reproducible.
//
// The weight gradients of a backward segment feed nothing but the optimizer, so they need not run where autograd reaches them: the
// host queues (gz, codes, gw) per layer and launches them together when the segment is done (runtime.QuantTables.finish_backward).
// One launch = up to WGR_MAXJOBS layers (the job table travels in the kernel arguments: 25 x 136 B < 4 KB).  A layer's output is cut into the 64 x 128 tiles of k_qwgrad2; its reduction index (batch x
// frames) into 64-frame STAGES; eight tiles form a tile GROUP (conv1 128 -> 512: its 8 row tiles; the res | skip pair: 2 row tiles x 4
// column tiles), and the work list is the sequence of (layer, tile group, stage) UNITS.  The grid is 32 TEAMS of 8 workgroups, one
// workgroup per CU, the 8 of a team on ONE XCD (they stream the same gz rows / code rows at about the same time: the re-reads come
// from that XCD's L2, as with k_qwgrad2's xcd_tile order).  Team t owns the contiguous unit range [t * chunk, (t + 1) * chunk)
// (stream-K): its member m accumulates tile m of every group it meets, over the stages of the range that fall into that group --
// hundreds of stages per workgroup, so the ~9 us of fixed cost per launch and the ramp of the load ring are paid once per launch
// instead of once per layer.  A group cut by a range boundary is finished by 2-3 teams: each publishes its partial tile to a slab
// slot (thread-major float4 rows, write-through `sc1` stores, every wave's vmcnt(0), workgroup barrier, ONE agent-scope ticket add
// per workgroup -- the split-K seam recipe of MI355X_MICROARCH.md), and the workgroup whose ticket came last sums the slots IN PART
// ORDER (sc1 loads) and adds the result to gw with plain read-modify-writes: no float atomics, and the same bits every run
// (k_qwgrad2 issues 2.1 M float atomics per launch: 7.6 of its 31 us, and the source of the step's run-to-run noise).
constexpr int WGR_MAXJOBS = 25, WGR_TEAM = 8, WGR_TEAMS = 32, WGR_SLOT_FLOATS = W2_TM * 256;     // (slots sized for the wide tile)
struct WJob {
    const float* A; const float* A2; const unsigned char* Bc; float* C; const float* qmin; const float* qmax;
    int64_t lda, lda2, ldb, sAb, sA2b, sBb;
    int M, M1, N, K;                      // Co1 + Co2, Co1, Ci, frames
    int tiles_n, ntiles, spb, nstages;    // column tiles, tiles, stages per batch, stages per tile (= batches * spb)
    int ubeg, tile0;                      // first unit of this job; global index of its tile 0 (ticket / slab addressing)
    int wide;                             // 64 x 256 tiles (Ci >= 256) instead of 64 x 128
};
struct WGroupArgs {
    int njobs, chunk, total, slots;       // units per team, units in all, slab slots per tile
    unsigned* tickets; float* slabs;
    WJob j[WGR_MAXJOBS];
};
static_assert(sizeof(WGroupArgs) <= 4096, "the job table travels in the kernel arguments");
// (s_nop 1: see store16 in fqss_dev.h -- the wait states behind a > 8-byte store that the compiler cannot insert for an asm statement;
//  without them the NEXT slab address, computed into the registers that held this store's data, was written to the slab in place of
//  the data's first dwords -- 1e31-sized "gradients" in a 16-lane pattern, whenever the memory pipe was slow to fetch the store data)
// (FQSS_NO_STORE_NOP: the pre-fix form, for tools/r06_store_hazard.sh only -- it proves that the regression test
//  tests/test_gpu_kernels.py::test_grouped_weight_gradients_beside_a_memory_hog fails on the hazard; never defined in the product build)
__device__ __forceinline__ void st16_sc1(float* p, const f32x4& v) {
#ifdef FQSS_NO_STORE_NOP
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
#else
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
#endif
}
__device__ __forceinline__ void ld16_sc1(f32x4& d, const float* p) { asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(d) : "v"(p) : "memory"); }

// One segment = the stages [s0, s0 + n) of ONE tile, accumulated by one workgroup, then published / finished.
//   WIDE = false: 64 (co) x 128 (ci) tile, 8 waves as 2 x 4, each a 32 x 32 output tile (the stage layout of k_qwgrad2);
//   WIDE = true : 64 x 256 tile, 8 waves as 2 x 4, each 32 x 64 (two B fragments share every A fragment).  Why: the stage loop is bound
//                 by LDS traffic -- per 64-frame stage a 32 x 32 wave tile reads 12 KB of gz pieces + 4 KB of codes for 12 MFMAs (128 KB
//                 per workgroup and stage = ~1000 cycles of the LDS pipe against ~400 of MFMA issue); the wide tile reads 20 KB for 24
//                 MFMAs.  Used for layers with Ci >= 256 (the res | skip pair: 4 x 2 tiles = ONE tile group, its gz row tiles read
//                 twice instead of four times); LDS 2 x (27.6 + 36.9) KB.
template <int GP, bool WIDE>
__device__ __forceinline__ void wgr_segment(const WGroupArgs& ga, const WJob& J, const int tile, const int s0, const int n, const int nparts,
                                            const int part, unsigned short* As_raw, unsigned short* Bs_raw, float* rsum, unsigned* ticket_s) {
    constexpr int TN = WIDE ? 256 : 128, NBF = WIDE ? 2 : 1;      // tile width, B fragments (and code loads) per wave (thread)
    typedef unsigned short (*AsT)[GP][W2_TM][W2_LD];
    typedef unsigned short (*BsT)[TN][W2_LD];
    AsT As = reinterpret_cast<AsT>(As_raw);
    BsT Bs = reinterpret_cast<BsT>(Bs_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3, lr = lane & 31, lh = lane >> 5;
    const int ar = tid >> 4, ac = (tid & 15) * 4;      // loader geometry of k_qwgrad2: A 16 threads per row, rows ar and ar + 32
    const int br = tid >> 2, bc = (tid & 3) * 16;      // B 4 threads per row (16 codes each), rows br (and br + 128)
    const bool first_half = wave < 4;
    const int row0 = (tile / J.tiles_n) * W2_TM, col0 = (tile % J.tiles_n) * TN;
    const int K = J.K, spb = J.spb, nb = J.nstages / J.spb;
    const int ka_last = (K - 1) & ~3, kb_last = (K - 1) & ~15;
    // ---- loader / converter state: the stage sequence runs over (batch, 64-frame chunk)
    const float* Arow[2];
    int64_t asb[2];
    bool aok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = row0 + ar + 32 * i;
        aok[i] = row < J.M;
        const int rc = aok[i] ? row : 0;
        Arow[i] = (rc < J.M1) ? J.A + (int64_t)rc * J.lda : J.A2 + (int64_t)(rc - J.M1) * J.lda2;
        asb[i] = (rc < J.M1) ? J.sAb : J.sA2b;
    }
    const unsigned char* Brow[NBF];
#pragma unroll
    for (int f = 0; f < NBF; ++f) Brow[f] = J.Bc + (int64_t)min(col0 + br + 128 * f, J.N - 1) * J.ldb;
    const int64_t sBb = J.sBb;
    int lb = s0 / spb, lk = (s0 - lb * spb) * W2_TK;      // loader position
    int ck = lk, cr = 0;                                   // converter: frame position inside its batch, relative stage
    const float* Ap[2] = {Arow[0] + lb * asb[0], Arow[1] + lb * asb[1]};
    int64_t boff = lb * sBb;
    struct Stage { f32x4 a[2]; u32x4 b[NBF]; };
    auto load = [&](Stage& st) {
#pragma unroll
        for (int i = 0; i < 2; ++i) wg_load16(st.a[i], Ap[i] + min(lk + ac, ka_last));
#pragma unroll
        for (int f = 0; f < NBF; ++f) wg_load16(st.b[f], Brow[f] + boff + min(lk + bc, kb_last));
        lk += W2_TK;
        if (lk >= spb * W2_TK) {       // next batch (clamped: stages behind the range are requested but never used)
            lk = 0;
            lb = min(lb + 1, nb - 1);
            Ap[0] = Arow[0] + lb * asb[0];
            Ap[1] = Arow[1] + lb * asb[1];
            boff = lb * sBb;
        }
    };
    constexpr int NLD = 2 + NBF;       // loads per thread and stage
    auto wait_oldest = [&](Stage& st) {
        if constexpr (WIDE) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(st.a[0]), "+v"(st.a[1]), "+v"(st.b[0]), "+v"(st.b[NBF - 1]) : "n"(NLD * (W2_RING - 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%3)" : "+v"(st.a[0]), "+v"(st.a[1]), "+v"(st.b[0]) : "n"(NLD * (W2_RING - 1)) : "memory");
    };
    auto wait_all = [&](Stage& st) {
        if constexpr (WIDE) asm volatile("s_waitcnt vmcnt(0)" : "+v"(st.a[0]), "+v"(st.a[1]), "+v"(st.b[0]), "+v"(st.b[NBF - 1]) : : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(st.a[0]), "+v"(st.a[1]), "+v"(st.b[0]) : : "memory");
    };
    float rs_part[2] = {0.f, 0.f};
    auto convert_store = [&](Stage& st, int buf) {
        const bool live = cr < n;
        const int k = ck + ac;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float x[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = (aok[i] && live && k + e < K) ? st.a[i][e] : 0.0f;
            rs_part[i] += (x[0] + x[1]) + (x[2] + x[3]);
            uint32_t o1[2], o2[2], o3[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float a0 = x[2 * e], a1 = x[2 * e + 1];
                const float r0 = a0 - bf_trunc(a0), r1 = a1 - bf_trunc(a1);
                o1[e] = __builtin_amdgcn_perm(__float_as_uint(a1), __float_as_uint(a0), 0x07060302u);
                if constexpr (GP == 3) {
                    const float q0 = r0 - bf_trunc(r0), q1 = r1 - bf_trunc(r1);
                    o2[e] = __builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), 0x07060302u);
                    o3[e] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
                } else {
                    o2[e] = __builtin_amdgcn_perm(bf_rne_word(r1), bf_rne_word(r0), 0x07060302u);
                }
            }
            const int row = ar + 32 * i;
            *reinterpret_cast<uint2*>(&As[buf][0][row][ac]) = make_uint2(o1[0], o1[1]);
            *reinterpret_cast<uint2*>(&As[buf][1][row][ac]) = make_uint2(o2[0], o2[1]);
            if constexpr (GP == 3) *reinterpret_cast<uint2*>(&As[buf][2][row][ac]) = make_uint2(o3[0], o3[1]);
        }
#pragma unroll
        for (int f = 0; f < NBF; ++f) {   // codes need no mask: finite, and where gz is masked they only ever meet a zero
            uint32_t o[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t w = st.b[f][q];
                const float f0 = (float)(w & 0xFFu), f1 = (float)((w >> 8) & 0xFFu);
                const float f2 = (float)((w >> 16) & 0xFFu), f3 = (float)(w >> 24);
                o[2 * q] = __builtin_amdgcn_perm(__float_as_uint(f1), __float_as_uint(f0), 0x07060302u);
                o[2 * q + 1] = __builtin_amdgcn_perm(__float_as_uint(f3), __float_as_uint(f2), 0x07060302u);
            }
            *reinterpret_cast<uint4*>(&Bs[buf][br + 128 * f][bc]) = make_uint4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<uint4*>(&Bs[buf][br + 128 * f][bc + 8]) = make_uint4(o[4], o[5], o[6], o[7]);
        }
        ++cr;
        ck += W2_TK;
        if (ck >= spb * W2_TK) ck = 0;
    };
    f32x16 acc[NBF];
#pragma unroll
    for (int f = 0; f < NBF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
    auto compute = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < W2_TK / 16; ++ks) {
            const int kk = ks * 16 + 8 * lh;
            bf16x8 bfr[NBF];
#pragma unroll
            for (int f = 0; f < NBF; ++f) bfr[f] = *reinterpret_cast<const bf16x8*>(&Bs[buf][wc * (32 * NBF) + 32 * f + lr][kk]);
            bf16x8 af[GP];
#pragma unroll
            for (int p = 0; p < GP; ++p) af[p] = *reinterpret_cast<const bf16x8*>(&As[buf][p][wr * 32 + lr][kk]);
#pragma unroll
            for (int p = GP - 1; p >= 0; --p)      // smallest pieces first
#pragma unroll
                for (int f = 0; f < NBF; ++f) acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[p], bfr[f], acc[f], 0, 0, 0);
        }
    };
    Stage st[W2_RING];
#pragma unroll
    for (int i = 0; i < W2_RING; ++i) load(st[i]);
    wait_oldest(st[0]);
    convert_store(st[0], 0);
    load(st[0]);
    __syncthreads();
    for (int s = 0; s < n; s += W2_RING) {      // stages at or behind n are converted to zeros (cr >= n)
#pragma unroll
        for (int i = 0; i < W2_RING; ++i) {
            Stage& nx = st[(i + 1) % W2_RING];
            if (first_half) compute(i & 1);
            wait_oldest(nx);
            convert_store(nx, (i + 1) & 1);
            load(nx);
            if (!first_half) compute(i & 1);
            __syncthreads();
        }
    }
#pragma unroll
    for (int i = 0; i < W2_RING; ++i) wait_all(st[i]);
    // ---- this workgroup's share of the tile: dx * S + min_x * rowsum(gz)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float v = rs_part[i];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((tid & 15) == 0) rsum[ar + 32 * i] = v;
    }
    __syncthreads();
    const float lo = *J.qmin, hi = *J.qmax;
    const float dx = (hi - lo) / 255.0f, mnx = lo;
    constexpr int NV = 16 * NBF;
    float v[NV];
#pragma unroll
    for (int f = 0; f < NBF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[16 * f + r] = dx * acc[f][r] + mnx * rsum[wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
    bool finish = nparts == 1;
    constexpr int SLOT = W2_TM * TN;       // floats per slab slot of this tile shape (the launch's slots are sized for the widest)
    if (!finish) {
        float* sl = ga.slabs + ((int64_t)(J.tile0 + tile) * ga.slots + part) * WGR_SLOT_FLOATS;
#pragma unroll
        for (int q = 0; q < NV / 4; ++q) st16_sc1(sl + ((int64_t)q * 512 + tid) * 4, f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) *ticket_s = __hip_atomic_fetch_add(&ga.tickets[J.tile0 + tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        finish = *ticket_s == (unsigned)(nparts - 1);
        if (finish) {       // the last to arrive: every part is published; sum them in part order (its own included: same bits whoever is last)
            const float* s0p = ga.slabs + (int64_t)(J.tile0 + tile) * ga.slots * WGR_SLOT_FLOATS;
#pragma unroll
            for (int r = 0; r < NV; ++r) v[r] = 0.0f;
            for (int p = 0; p < nparts; ++p) {
#pragma unroll
                for (int h = 0; h < NBF; ++h) {      // four 16-B loads in flight at a time
                    f32x4 t[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) ld16_sc1(t[q], s0p + (int64_t)p * WGR_SLOT_FLOATS + ((int64_t)(4 * h + q) * 512 + tid) * 4);
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]) : : "memory");
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[16 * h + 4 * q + e] += t[q][e];
                }
            }
            if (tid == 0) __hip_atomic_store(&ga.tickets[J.tile0 + tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // clean for the next launch
        }
    }
    static_assert(SLOT <= WGR_SLOT_FLOATS, "slab slots are sized for the wide tile");
    if (finish) {       // gw += tile: this workgroup is the tile's only writer; all reads first (one round trip, not 16)
        const int Mr = J.M, Nc = J.N;
        float* const Cp = J.C;
        float old[NV];
#pragma unroll
        for (int f = 0; f < NBF; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, col = col0 + wc * (32 * NBF) + 32 * f + lr;
                old[16 * f + r] = (row < Mr && col < Nc) ? Cp[(int64_t)row * Nc + col] : 0.0f;
            }
#pragma unroll
        for (int f = 0; f < NBF; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, col = col0 + wc * (32 * NBF) + 32 * f + lr;
                if (row < Mr && col < Nc) Cp[(int64_t)row * Nc + col] = old[16 * f + r] + v[16 * f + r];
            }
        // the stores above retire before this workgroup's next segment starts its load ring: that ring is retired with COUNTED waits
        // (wait_oldest), and stores do not retire in one order with loads
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();     // rsum / ticket_s / the LDS stages are reused by the next segment
}

template <int GP>
__global__ __launch_bounds__(512, 1) void k_qwgrad_group(WGroupArgs ga) {
    // two stage buffers of the WIDE tile (the narrow one uses the front of each): gz pieces 2 x GP x 9,216 B, codes 2 x 36,864 B
    __shared__ __attribute__((aligned(16))) unsigned short As_raw[2 * GP * W2_TM * W2_LD];
    __shared__ __attribute__((aligned(16))) unsigned short Bs_raw[2 * 256 * W2_LD];
    __shared__ float rsum[W2_TM];
    __shared__ unsigned ticket_s;
    const int L = blockIdx.x, xcd = L % kXcds, slot = L / kXcds;                 // 32 workgroups per XCD: 4 teams of 8
    const int team = (slot / WGR_TEAM) * kXcds + xcd, member = slot % WGR_TEAM;
    int u0 = team * ga.chunk;
    const int u1 = min(ga.total, u0 + ga.chunk);
    while (u0 < u1) {
        int jn = 0;
        for (int q = 1; q < ga.njobs; ++q) jn = (ga.j[q].ubeg <= u0) ? q : jn;
        const WJob& J = ga.j[jn];
        const int rel = u0 - J.ubeg, grp = rel / J.nstages, s0 = rel - grp * J.nstages;
        const int n = min(J.nstages - s0, u1 - u0);
        const int g0 = J.ubeg + grp * J.nstages;                                   // the group's unit range -> which teams share it
        const int t_first = g0 / ga.chunk, nparts = (g0 + J.nstages - 1) / ga.chunk - t_first + 1, part = team - t_first;
        u0 += n;
        const int tile = grp * WGR_TEAM + member;
        if (tile >= J.ntiles) continue;                                            // (workgroup-uniform)
        if (J.wide) wgr_segment<GP, true>(ga, J, tile, s0, n, nparts, part, As_raw, Bs_raw, rsum, &ticket_s);
        else wgr_segment<GP, false>(ga, J, tile, s0, n, nparts, part, As_raw, Bs_raw, rsum, &ticket_s);
    }
}

// per-channel weight codes for the q-GEMMs: idx [Co][Ci], idxT [Ci][Co], dw[Co], rw[Co] (one block per channel)
__global__ __launch_bounds__(256) void k_wq_codes(const float* __restrict__ w, signed char* __restrict__ idx,
                                                   signed char* __restrict__ idxT, float* dw, float* rw, int Co, int Ci,
                                                   const float* __restrict__ qmin, const float* __restrict__ qmax) {
    __shared__ float red[4];
    const int co = blockIdx.x;
    const float a = fmaxf(fabsf(qmin[co]), fabsf(qmax[co]));
    const float delta = (2.0f * a) / 255.0f;
    float s = 0.0f;
    for (int ci = threadIdx.x; ci < Ci; ci += 256) {
        const float X = rintf(w[(int64_t)co * Ci + ci] / delta);
        const float q = fminf(fmaxf(X, -128.0f), 127.0f);
        idx[(int64_t)co * Ci + ci] = (signed char)q;
        idxT[(int64_t)ci * Co + co] = (signed char)q;
        s += q;   // |sum| <= 512*128 : exact in fp32
    }
    float v[1] = {s};
    block_sum<float, 1>(v, red);
    if (threadIdx.x == 0) {
        dw[co] = delta;
        rw[co] = v[0];
    }
}

}  // namespace fqss

using namespace fqss;

// bf16 pieces of the fp32 gradient operand in the dgrad / wgrad q-GEMMs: 3 = exact products (default: what every parity gate and the
// benchmark run on) or, opt-in with FQSS_GRAD_PIECES=2, the two-piece form (read on every call: the test flips it inside one process)
static int grad_pieces() {
    const char* e = getenv("FQSS_GRAD_PIECES");
    return (e != nullptr && e[0] == '2') ? 2 : 3;
}

extern "C" int fqss_wq_codes(const float* w, int8_t* idx, int8_t* idxT, float* dw, float* rw, int Co, int Ci,
                             const float* qmin, const float* qmax, fqss_stream_t stream) {
    FQSS_REQUIRE(w && idx && idxT && dw && rw && qmin && qmax && Co > 0 && Ci > 0, "bad args");
    hipLaunchKernelGGL(k_wq_codes, dim3((unsigned)Co), dim3(256), 0, (hipStream_t)stream, w, (signed char*)idx,
                       (signed char*)idxT, dw, rw, Co, Ci, qmin, qmax);
    return launch_status("fqss_wq_codes");
}

struct QpwQuant {   // optional fused output quantizer of fqss_qpw_fwdq
    int act; const float* slope; const float *min1, *max1, *min2, *max2; uint8_t *yc1, *yc2; int64_t ld1, ld2;
    long long* stats;   // optional: integer statistics of yc1 per workgroup (single-output launches only)
    const FqssAddAfter *add1, *add2;   // optional: the AddQ behind output 1 / 2 (fqss_qpw_fwdq_add)
};

static int qpw_fwd_impl(const char* who, const uint8_t* xc, const int8_t* wi, const float* dw, const float* rw, const float* bias1,
                        const float* bias2, const float* qmin_x, const float* qmax_x, float* z1, float* z2, int B, int Ci, int Co1,
                        int Co2, int M, int64_t ld_xc, int64_t ld_z1, int64_t ld_z2, fqss_stream_t stream,
                        const QpwQuant* qq = nullptr) {
    if (B == 0 || M == 0) return FQSS_OK;   // empty input: nothing to do (a 0-element tensor has a null data pointer)
    const int Co = Co1 + Co2;
    FQSS_REQUIRE(xc && wi && dw && rw && qmin_x && qmax_x && z1 && (Co2 == 0 || z2), "null tensor");
    FQSS_REQUIRE(B >= 0 && Ci > 0 && Co1 > 0 && Co2 >= 0 && M >= 0 && ld_xc >= M && ld_z1 >= M && (Co2 == 0 || ld_z2 >= M), "bad shape");
    FQSS_REQUIRE(Ci % 16 == 0 && Ci <= 512 && ld_xc % 16 == 0 && aligned16(xc) && aligned16(wi),
                 "q-GEMM needs Ci % 16 == 0, Ci <= 512 (exact fp32 integer sum) and 16-B aligned code rows");
    FQSS_REQUIRE(aligned16(z1) && ld_z1 % 4 == 0 && (Co2 == 0 || (aligned16(z2) && ld_z2 % 4 == 0)), "output rows must be 16-B aligned");
    FQSS_REQUIRE((bias1 == nullptr) == (bias2 == nullptr) || Co2 == 0, "paired layers: both or neither with bias");
    if (qq != nullptr) FQSS_REQUIRE(Co2 == 0 || Co1 % 32 == 0, "fused quantizer of a pair needs Co1 % 32 == 0");
    QGemmArgs g{};
    g.A = wi; g.B = xc; g.C = z1; g.M = Co; g.N = M; g.K = Ci;
    g.lda = Ci; g.ldb = ld_xc; g.ldc = ld_z1;
    g.sAb = 0; g.sBb = (int64_t)Ci * ld_xc; g.sCb = (int64_t)Co1 * ld_z1;
    g.M1 = Co1; g.C2 = z2; g.ldc2 = ld_z2; g.sC2b = (int64_t)Co2 * ld_z2; g.bias2 = bias2; g.K1 = Ci;
    g.dw = dw; g.rw = rw; g.bias = bias1; g.qmin_x = qmin_x; g.qmax_x = qmax_x; g.ksplit = 1; g.kchunk = Ci;
    if (qq != nullptr) {
        FQSS_REQUIRE(qq->min1 && qq->max1 && qq->yc1 && (Co2 == 0 || (qq->min2 && qq->max2 && qq->yc2)), "fused quantizer: null pointer");
        FQSS_REQUIRE(qq->ld1 % 16 == 0 && qq->ld1 >= M && aligned16(qq->yc1) && (Co2 == 0 || (qq->ld2 % 16 == 0 && qq->ld2 >= M && aligned16(qq->yc2))),
                     "output code rows must be 16-B aligned");
        FQSS_REQUIRE(qq->act != FQSS_ACT_PRELU || qq->slope, "PReLU needs a slope");
        g.Q1 = qq->yc1; g.Q2 = qq->yc2; g.ldq1 = qq->ld1; g.ldq2 = qq->ld2;
        g.sQ1b = (int64_t)Co1 * qq->ld1; g.sQ2b = (int64_t)Co2 * qq->ld2;
        g.qy_min1 = qq->min1; g.qy_max1 = qq->max1; g.qy_min2 = qq->min2; g.qy_max2 = qq->max2;
        g.qact = qq->act; g.qslope = qq->slope;
        FQSS_REQUIRE(!qq->stats || Co2 == 0, "output statistics: single-output launches only");
        g.stats = qq->stats;
        const FqssAddAfter* ad[2] = {qq->add1, qq->add2};
        for (int i = 0; i < 2; ++i) {
            const FqssAddAfter* a = ad[i];
            if (a == nullptr) continue;
            FQSS_REQUIRE(i == 0 || Co2 > 0, "fused AddQ behind a second output that does not exist");
            FQSS_REQUIRE(qq->act == FQSS_ACT_NONE, "fused AddQ: the conv in front of it has no activation (Conv1dQ)");
            FQSS_REQUIRE(a->a && a->y && a->amin && a->amax && a->qmin && a->qmax, "fused AddQ: null pointer");
            FQSS_REQUIRE(a->ld_a % 16 == 0 && a->ld_a >= M && a->ld_y % 16 == 0 && a->ld_y >= M && aligned16(a->a) && aligned16(a->y),
                         "fused AddQ: code rows must be 16-B aligned");
            if (i == 0) {
                g.AD1 = a->a; g.ldad1 = a->ld_a; g.S1 = a->y; g.lds1 = a->ld_y;
                g.ad_min1 = a->amin; g.ad_max1 = a->amax; g.s_min1 = a->qmin; g.s_max1 = a->qmax;
            } else {
                g.AD2 = a->a; g.ldad2 = a->ld_a; g.S2 = a->y; g.lds2 = a->ld_y;
                g.ad_min2 = a->amin; g.ad_max2 = a->amax; g.s_min2 = a->qmin; g.s_max2 = a->qmax;
            }
        }
    }
    g.tiles_n = (int)cdiv(M, QBN); g.tiles_m = (int)cdiv(Co, QBM); g.batches = B;
    hipLaunchKernelGGL((k_qgemm<0>), dim3(xcd_grid((int64_t)g.tiles_n * B, g.tiles_m)), dim3(256), 0, (hipStream_t)stream, g);
    return launch_status(who);
}

extern "C" int fqss_qpw_fwd(const uint8_t* xc, const int8_t* wi, const float* dw, const float* rw, const float* bias,
                            const float* qmin_x, const float* qmax_x, float* z, int B, int Ci, int Co, int M,
                            int64_t ld_xc, int64_t ld_z, fqss_stream_t stream) {
    return qpw_fwd_impl("fqss_qpw_fwd", xc, wi, dw, rw, bias, nullptr, qmin_x, qmax_x, z, nullptr, B, Ci, Co, 0, M, ld_xc, ld_z, 0, stream);
}

extern "C" int fqss_qpw_fwd2(const uint8_t* xc, const int8_t* wi, const float* dw, const float* rw, const float* bias1,
                             const float* bias2, const float* qmin_x, const float* qmax_x, float* z1, float* z2, int B, int Ci,
                             int Co1, int Co2, int M, int64_t ld_xc, int64_t ld_z1, int64_t ld_z2, fqss_stream_t stream) {
    FQSS_REQUIRE(Co2 > 0, "second layer missing");
    return qpw_fwd_impl("fqss_qpw_fwd2", xc, wi, dw, rw, bias1, bias2, qmin_x, qmax_x, z1, z2, B, Ci, Co1, Co2, M, ld_xc, ld_z1, ld_z2, stream);
}

extern "C" int fqss_qpw_fwdq(const uint8_t* xc, const int8_t* wi, const float* dw, const float* rw, const float* bias1,
                             const float* bias2, const float* qmin_x, const float* qmax_x, float* z1, float* z2, int act,
                             const float* slope, const float* qmin1, const float* qmax1, const float* qmin2, const float* qmax2,
                             uint8_t* yc1, uint8_t* yc2, int B, int Ci, int Co1, int Co2, int M, int64_t ld_xc, int64_t ld_z1,
                             int64_t ld_z2, int64_t ld_yc1, int64_t ld_yc2, int64_t* stats1, fqss_stream_t stream) {
    QpwQuant qq{act, slope, qmin1, qmax1, qmin2, qmax2, yc1, yc2, ld_yc1, ld_yc2, (long long*)stats1, nullptr, nullptr};
    return qpw_fwd_impl("fqss_qpw_fwdq", xc, wi, dw, rw, bias1, bias2, qmin_x, qmax_x, z1, z2, B, Ci, Co1, Co2, M, ld_xc, ld_z1, ld_z2,
                        stream, &qq);
}

// ... and with the AddQ layers that are the only consumers of the outputs evaluated in the same epilogue (add1 behind output 1, add2
// behind output 2, either NULL): their sum codes are written beside yc1 / yc2, bit-identical to fqss_ewq_fwd on those codes
extern "C" int fqss_qpw_fwdq_add(const uint8_t* xc, const int8_t* wi, const float* dw, const float* rw, const float* bias1,
                                 const float* bias2, const float* qmin_x, const float* qmax_x, float* z1, float* z2, const float* qmin1,
                                 const float* qmax1, const float* qmin2, const float* qmax2, uint8_t* yc1, uint8_t* yc2, int B, int Ci,
                                 int Co1, int Co2, int M, int64_t ld_xc, int64_t ld_z1, int64_t ld_z2, int64_t ld_yc1, int64_t ld_yc2,
                                 const FqssAddAfter* add1, const FqssAddAfter* add2, fqss_stream_t stream) {
    QpwQuant qq{FQSS_ACT_NONE, nullptr, qmin1, qmax1, qmin2, qmax2, yc1, yc2, ld_yc1, ld_yc2, nullptr, add1, add2};
    return qpw_fwd_impl("fqss_qpw_fwdq_add", xc, wi, dw, rw, bias1, bias2, qmin_x, qmax_x, z1, z2, B, Ci, Co1, Co2, M, ld_xc, ld_z1,
                        ld_z2, stream, &qq);
}

static int qpw_bwd_x_impl(const char* who, const float* gz1, const float* gz2, const int8_t* wiT, const float* dw, float* gx, int B,
                          int Ci, int Co1, int Co2, int M, int64_t ld_gz1, int64_t ld_gz2, int64_t ld_gx, fqss_stream_t stream,
                          const float* addend = nullptr, int64_t ld_add = 0) {
    if (B == 0 || M == 0) return FQSS_OK;   // empty input: nothing to do (a 0-element tensor has a null data pointer)
    const int Co = Co1 + Co2;
    FQSS_REQUIRE(gz1 && wiT && dw && gx && (Co2 == 0 || gz2), "null tensor");
    FQSS_REQUIRE(B >= 0 && Ci > 0 && Co1 > 0 && Co2 >= 0 && M >= 0 && ld_gz1 >= M && ld_gx >= M, "bad shape");
    FQSS_REQUIRE(Co1 % 16 == 0 && Co2 % 16 == 0 && ld_gz1 % 4 == 0 && aligned16(gz1) && aligned16(wiT) && ld_gz1 >= ((M + 3) & ~3),
                 "q-GEMM dgrad needs Co % 16 == 0 and 16-B aligned gradient rows");
    FQSS_REQUIRE(Co2 == 0 || (ld_gz2 % 4 == 0 && aligned16(gz2) && ld_gz2 >= ((M + 3) & ~3)), "second gradient: 16-B aligned rows");
    FQSS_REQUIRE(aligned16(gx) && ld_gx % 4 == 0, "output rows must be 16-B aligned");
    FQSS_REQUIRE(Co <= 1024, "dgrad stages delta_w of at most 1024 output channels in LDS");
    if (B == 0 || M == 0) return FQSS_OK;
    QGemmArgs g{};
    g.A = wiT; g.B = gz1; g.C = gx; g.M = Ci; g.N = M; g.K = Co;
    g.lda = Co; g.ldb = ld_gz1; g.ldc = ld_gx;
    g.sAb = 0; g.sBb = (int64_t)Co1 * ld_gz1; g.sCb = (int64_t)Ci * ld_gx;
    g.M1 = Ci; g.K1 = Co1; g.B2 = gz2; g.ldb2 = ld_gz2; g.sB2b = (int64_t)Co2 * ld_gz2;
    g.dw = dw; g.ksplit = 1; g.kchunk = Co;
    if (addend != nullptr) {
        FQSS_REQUIRE(aligned16(addend) && ld_add % 4 == 0 && ld_add >= ((M + 3) & ~3), "addend rows must be 16-B aligned and padded to 4");
        g.C2 = const_cast<float*>(addend); g.ldc2 = ld_add; g.sC2b = (int64_t)Ci * ld_add;
    }
#ifdef FQSS_EXPERIMENTS
    // round 4 experiment (csrc/experiments/qgemm_ring.hip; `make experiments`, FQSS_DGRAD_RING=1): the ring form, measured slower
    if (grad_pieces() == 3 && qdgrad_ring_ok(Ci, Co1, Co2) && (!addend || ld_add < (1ll << 26)))
        return qdgrad_ring(who, gz1, gz2, wiT, dw, addend, gx, B, Ci, Co1, Co2, M, ld_gz1, ld_gz2, ld_add, ld_gx, stream);
#endif
    g.tiles_n = (int)cdiv(M, QBN); g.tiles_m = (int)cdiv(Ci, QBM); g.batches = B;
    if (grad_pieces() == 3)
        hipLaunchKernelGGL((k_qgemm<1, 3>), dim3(xcd_grid((int64_t)g.tiles_n * B, g.tiles_m)), dim3(256), 0, (hipStream_t)stream, g);
    else
        hipLaunchKernelGGL((k_qgemm<1, 2>), dim3(xcd_grid((int64_t)g.tiles_n * B, g.tiles_m)), dim3(256), 0, (hipStream_t)stream, g);
    return launch_status(who);
}

extern "C" int fqss_qpw_bwd_x(const float* gz, const int8_t* wiT, const float* dw, float* gx, int B, int Ci, int Co,
                              int M, int64_t ld_gz, int64_t ld_gx, fqss_stream_t stream) {
    return qpw_bwd_x_impl("fqss_qpw_bwd_x", gz, nullptr, wiT, dw, gx, B, Ci, Co, 0, M, ld_gz, 0, ld_gx, stream);
}

/* fqss_qpw_bwd_x that also adds `addend` [B][Ci][ld_add] (the gradient arriving over the other branch of a residual fork) in its
 * epilogue: gx = W_q^T gz + addend */
extern "C" int fqss_qpw_bwd_x_add(const float* gz, const int8_t* wiT, const float* dw, const float* addend, float* gx, int B, int Ci, int Co,
                                  int M, int64_t ld_gz, int64_t ld_add, int64_t ld_gx, fqss_stream_t stream) {
    FQSS_REQUIRE(addend, "null addend");
    return qpw_bwd_x_impl("fqss_qpw_bwd_x_add", gz, nullptr, wiT, dw, gx, B, Ci, Co, 0, M, ld_gz, 0, ld_gx, stream, addend, ld_add);
}

extern "C" int fqss_qpw_bwd_x2(const float* gz1, const float* gz2, const int8_t* wiT, const float* dw, float* gx, int B, int Ci,
                               int Co1, int Co2, int M, int64_t ld_gz1, int64_t ld_gz2, int64_t ld_gx, fqss_stream_t stream) {
    FQSS_REQUIRE(Co2 > 0, "second layer missing");
    return qpw_bwd_x_impl("fqss_qpw_bwd_x2", gz1, gz2, wiT, dw, gx, B, Ci, Co1, Co2, M, ld_gz1, ld_gz2, ld_gx, stream);
}

static int qpw_bwd_w_impl(const char* who, const float* gz1, const float* gz2, const uint8_t* xc, const float* qmin_x,
                          const float* qmax_x, float* gw, int B, int Ci, int Co1, int Co2, int M, int64_t ld_gz1, int64_t ld_gz2,
                          int64_t ld_xc, fqss_stream_t stream) {
    if (B == 0 || M == 0) return FQSS_OK;   // empty input: nothing to do (a 0-element tensor has a null data pointer)
    const int Co = Co1 + Co2;
    FQSS_REQUIRE(gz1 && xc && qmin_x && qmax_x && gw && (Co2 == 0 || gz2), "null tensor");
    FQSS_REQUIRE(B >= 0 && Ci > 0 && Co1 > 0 && Co2 >= 0 && M >= 0 && ld_gz1 >= M && ld_xc >= M, "bad shape");
    FQSS_REQUIRE(ld_gz1 % 4 == 0 && ld_xc % 16 == 0 && aligned16(gz1) && aligned16(xc) && ld_gz1 >= ((M + 3) & ~3),
                 "q-GEMM wgrad needs 16-B aligned gradient and code rows");
    FQSS_REQUIRE(Co2 == 0 || (ld_gz2 % 4 == 0 && aligned16(gz2) && ld_gz2 >= ((M + 3) & ~3)), "second gradient: 16-B aligned rows");
    if (B == 0 || M == 0) return FQSS_OK;
    QGemmArgs g{};
    g.A = gz1; g.B = xc; g.C = gw; g.M = Co; g.N = Ci; g.K = M;
    g.lda = ld_gz1; g.ldb = ld_xc; g.ldc = Ci;
    g.sAb = (int64_t)Co1 * ld_gz1; g.sBb = (int64_t)Ci * ld_xc; g.sCb = 0;
    g.M1 = Co1; g.A2 = gz2; g.lda2 = ld_gz2; g.sA2b = (int64_t)Co2 * ld_gz2; g.K1 = M;
    g.qmin_x = qmin_x; g.qmax_x = qmax_x;
    // one workgroup = 64 (co) x 128 (ci); split n so that ~FQSS_WGRAD_BLOCKS workgroups stream the operands
    const int64_t tiles = cdiv(Co, 64) * cdiv(Ci, 128) * B;
    int want = (int)((FQSS_WGRAD_BLOCKS + tiles / 2) / tiles);
    if (want < 1) want = 1;
    g.tiles_n = (int)cdiv(Ci, 128); g.tiles_m = (int)cdiv(Co, 64); g.batches = B;
    static const bool reg_direct = getenv("FQSS_WGRAD_REGDIRECT") != nullptr;   // the round-1 kernel, kept for A/B measurements
    if (reg_direct) {
        int kchunk = (int)cdiv(cdiv(M, want), 32 * WG_STAGES) * 32 * WG_STAGES;   // whole rounds of the load ring
        g.kchunk = kchunk;
        g.ksplit = (int)cdiv(M, kchunk);
        hipLaunchKernelGGL(k_qwgrad, dim3(xcd_grid((int64_t)B * g.ksplit, (int64_t)g.tiles_m * g.tiles_n)), dim3(256), 0, (hipStream_t)stream, g);
    } else {
        int kchunk = (int)cdiv(cdiv(M, want), W2_RING * W2_TK) * W2_RING * W2_TK;   // whole rounds of the stage ring
        g.kchunk = kchunk;
        g.ksplit = (int)cdiv(M, kchunk);
        if (grad_pieces() == 3)
            hipLaunchKernelGGL(k_qwgrad2<3>, dim3(xcd_grid((int64_t)B * g.ksplit, (int64_t)g.tiles_m * g.tiles_n)), dim3(512), 0, (hipStream_t)stream, g);
        else
            hipLaunchKernelGGL(k_qwgrad2<2>, dim3(xcd_grid((int64_t)B * g.ksplit, (int64_t)g.tiles_m * g.tiles_n)), dim3(512), 0, (hipStream_t)stream, g);
    }
    return launch_status(who);
}

extern "C" int fqss_qpw_bwd_w(const float* gz, const uint8_t* xc, const float* qmin_x, const float* qmax_x, float* gw,
                              int B, int Ci, int Co, int M, int64_t ld_gz, int64_t ld_xc, fqss_stream_t stream) {
    return qpw_bwd_w_impl("fqss_qpw_bwd_w", gz, nullptr, xc, qmin_x, qmax_x, gw, B, Ci, Co, 0, M, ld_gz, 0, ld_xc, stream);
}

extern "C" int fqss_qpw_bwd_w2(const float* gz1, const float* gz2, const uint8_t* xc, const float* qmin_x, const float* qmax_x,
                               float* gw, int B, int Ci, int Co1, int Co2, int M, int64_t ld_gz1, int64_t ld_gz2, int64_t ld_xc,
                               fqss_stream_t stream) {
    FQSS_REQUIRE(Co2 > 0, "second layer missing");
    return qpw_bwd_w_impl("fqss_qpw_bwd_w2", gz1, gz2, xc, qmin_x, qmax_x, gw, B, Ci, Co1, Co2, M, ld_gz1, ld_gz2, ld_xc, stream);
}

// ---- grouped weight gradients (k_qwgrad_group): workspace = [64 KB of tickets, zero when first used, left zero][slab slots]
constexpr int64_t WGR_TICKET_BYTES = 65536;
static int wgr_plan(const FqssWgradJob* jobs, int n0, int n, WGroupArgs& ga, int64_t& slab_bytes) {
    int u = 0, t = 0, max_st = 0;
    ga.njobs = n;
    for (int q = 0; q < n; ++q) {
        const FqssWgradJob& f = jobs[n0 + q];
        WJob& J = ga.j[q];
        const int Co = f.Co1 + f.Co2;
        J.A = f.gz1; J.A2 = f.gz2; J.Bc = f.xc; J.C = f.gw; J.qmin = f.qmin_x; J.qmax = f.qmax_x;
        J.lda = f.ld_gz1; J.lda2 = f.ld_gz2; J.ldb = f.ld_xc;
        J.sAb = (int64_t)f.Co1 * f.ld_gz1; J.sA2b = (int64_t)f.Co2 * f.ld_gz2; J.sBb = (int64_t)f.Ci * f.ld_xc;
        J.M = Co; J.M1 = f.Co1; J.N = f.Ci; J.K = f.M;
        static const bool no_wide = getenv("FQSS_WGRAD_WIDE") != nullptr && getenv("FQSS_WGRAD_WIDE")[0] == '0';      // A/B knob
        J.wide = (f.Ci >= 256 && !no_wide) ? 1 : 0;
        J.tiles_n = (int)cdiv(f.Ci, J.wide ? 256 : W2_TN);
        J.ntiles = (int)cdiv(Co, W2_TM) * J.tiles_n;
        J.spb = (int)cdiv(f.M, W2_TK);
        J.nstages = f.B * J.spb;
        J.ubeg = u; J.tile0 = t;
        u += (int)cdiv(J.ntiles, WGR_TEAM) * J.nstages;
        t += J.ntiles;
        if (J.nstages > max_st) max_st = J.nstages;
    }
    ga.total = u;
    ga.chunk = (int)cdiv(u, WGR_TEAMS);
    ga.slots = (max_st - 1) / ga.chunk + 2;
    slab_bytes = (int64_t)t * ga.slots * WGR_SLOT_FLOATS * 4;
    return t;
}
static int wgr_check(const FqssWgradJob* jobs, int njobs) {
    FQSS_REQUIRE(jobs && njobs > 0, "no jobs");
    for (int q = 0; q < njobs; ++q) {
        const FqssWgradJob& f = jobs[q];
        FQSS_REQUIRE(f.gz1 && f.xc && f.qmin_x && f.qmax_x && f.gw && (f.Co2 == 0 || f.gz2), "null tensor");
        FQSS_REQUIRE(f.B > 0 && f.Ci > 0 && f.Co1 > 0 && f.Co2 >= 0 && f.M > 0 && f.ld_gz1 >= f.M && f.ld_xc >= f.M, "bad shape");
        FQSS_REQUIRE(f.ld_gz1 % 4 == 0 && f.ld_xc % 16 == 0 && aligned16(f.gz1) && aligned16(f.xc) && f.ld_gz1 >= ((f.M + 3) & ~3),
                     "q-GEMM wgrad needs 16-B aligned gradient and code rows");
        FQSS_REQUIRE(f.Co2 == 0 || (f.ld_gz2 % 4 == 0 && aligned16(f.gz2) && f.ld_gz2 >= ((f.M + 3) & ~3)), "second gradient: 16-B aligned rows");
        FQSS_REQUIRE((int64_t)f.B * cdiv(f.M, W2_TK) * cdiv(cdiv(f.Co1 + f.Co2, W2_TM) * cdiv(f.Ci, W2_TN), WGR_TEAM) < (1ll << 26), "job too large");
        for (int p = 0; p < q; ++p) FQSS_REQUIRE(jobs[p].gw != f.gw, "two jobs of one call accumulate into the same gw (plain read-modify-write)");
    }
    return FQSS_OK;
}

extern "C" int64_t fqss_qpw_bwd_w_group_ws(const FqssWgradJob* jobs, int njobs) {
    if (wgr_check(jobs, njobs) != FQSS_OK) return -1;
    int64_t need = 0;
    const int per = (int)cdiv(njobs, cdiv(njobs, WGR_MAXJOBS));     // launches of equal size (50 jobs: 25 + 25, not 25 + 25 or 48 + 2)
    for (int n0 = 0; n0 < njobs; n0 += per) {
        WGroupArgs ga{};
        int64_t sb = 0;
        const int t = wgr_plan(jobs, n0, njobs - n0 < per ? njobs - n0 : per, ga, sb);
        if ((int64_t)t * 4 > WGR_TICKET_BYTES) { set_error("fqss_qpw_bwd_w_group_ws: more than 16384 tiles in one launch"); return -1; }
        if (sb > need) need = sb;
    }
    return WGR_TICKET_BYTES + need;
}

extern "C" int fqss_qpw_bwd_w_group(const FqssWgradJob* jobs, int njobs, void* ws, int64_t ws_bytes, fqss_stream_t stream) {
    if (njobs == 0) return FQSS_OK;
    if (int rc = wgr_check(jobs, njobs)) return rc;
    FQSS_REQUIRE(ws && aligned16(ws), "workspace: 16-B aligned device memory (fqss_qpw_bwd_w_group_ws bytes, zero-filled when first used)");
    const int per = (int)cdiv(njobs, cdiv(njobs, WGR_MAXJOBS));
    for (int n0 = 0; n0 < njobs; n0 += per) {
        WGroupArgs ga{};
        int64_t sb = 0;
        const int t = wgr_plan(jobs, n0, njobs - n0 < per ? njobs - n0 : per, ga, sb);
        FQSS_REQUIRE((int64_t)t * 4 <= WGR_TICKET_BYTES && WGR_TICKET_BYTES + sb <= ws_bytes, "workspace too small (fqss_qpw_bwd_w_group_ws)");
        ga.tickets = (unsigned*)ws;
        ga.slabs = (float*)((char*)ws + WGR_TICKET_BYTES);
        if (grad_pieces() == 3)
            hipLaunchKernelGGL(k_qwgrad_group<3>, dim3(WGR_TEAMS * WGR_TEAM), dim3(512), 0, (hipStream_t)stream, ga);
        else
            hipLaunchKernelGGL(k_qwgrad_group<2>, dim3(WGR_TEAMS * WGR_TEAM), dim3(512), 0, (hipStream_t)stream, ga);
        if (int rc = launch_status("fqss_qpw_bwd_w_group")) return rc;
    }
    return FQSS_OK;
}

// plain fp32 pointwise conv z = W x + b on the bf16 matrix cores: both operands split exactly in three; all nine partial products
// (every product exact: fqss_pwconv_fwd_x3, the ConvTasNet-family gates at 1e-5) or the six above 2^-24 (fqss_pwconv_fwd_x3s: the
// arithmetic of csrc/gemm_x3.hip, one third fewer MFMAs -- the frame-path GEMMs of HTDemucs)
static int pwconv_fwd_x3_impl(const char* who, bool six, const float* x, const float* w, const float* bias, float* z, int B, int Ci, int Co,
                              int M, int64_t ld_x, int64_t ld_z, fqss_stream_t stream) {
    if (B == 0 || M == 0) return FQSS_OK;   // empty input: nothing to do (a 0-element tensor has a null data pointer)
    FQSS_REQUIRE(x && w && z, "null tensor");
    FQSS_REQUIRE(B >= 0 && Ci > 0 && Co > 0 && M >= 0 && ld_x >= M && ld_z >= M, "bad shape");
    FQSS_REQUIRE(Ci % 4 == 0 && ld_x % 4 == 0 && aligned16(x) && aligned16(w) && ld_x >= ((M + 3) & ~3),
                 "x3 GEMM needs Ci % 4 == 0 and 16-B aligned activation rows");
    FQSS_REQUIRE(aligned16(z) && ld_z % 4 == 0, "output rows must be 16-B aligned");
    if (B == 0 || M == 0) return FQSS_OK;
    QGemmArgs g{};
    g.A = w; g.B = x; g.C = z; g.M = Co; g.N = M; g.K = Ci;
    g.lda = Ci; g.ldb = ld_x; g.ldc = ld_z;
    g.sAb = 0; g.sBb = (int64_t)Ci * ld_x; g.sCb = (int64_t)Co * ld_z;
    g.bias = bias; g.ksplit = 1; g.kchunk = Ci; g.M1 = Co; g.K1 = Ci;
    g.tiles_n = (int)cdiv(M, QBN); g.tiles_m = (int)cdiv(Co, QBM); g.batches = B;
    if (six) hipLaunchKernelGGL((k_qgemm<3, 2>), dim3(xcd_grid((int64_t)g.tiles_n * B, g.tiles_m)), dim3(256), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL((k_qgemm<3>), dim3(xcd_grid((int64_t)g.tiles_n * B, g.tiles_m)), dim3(256), 0, (hipStream_t)stream, g);
    return launch_status(who);
}

/* z[b] = (dw[co] * Wi[co][:]) x[b] + bias: pointwise conv of a FLOAT input with a fake-quantized weight given as its int8 codes
 * (k_qgemm<4>: x in three exact bf16 pieces x one exact plane of codes: three products per k; fqss_pwconv_fwd_x3s needs six) */
extern "C" int fqss_pwconv_fwd_wq(const float* x, const int8_t* wi, const float* dw, const float* bias, float* z, int B, int Ci, int Co, int M,
                                  int64_t ld_x, int64_t ld_z, fqss_stream_t stream) {
    if (B == 0 || M == 0) return FQSS_OK;
    FQSS_REQUIRE(x && wi && dw && z, "null tensor");
    FQSS_REQUIRE(B >= 0 && Ci >= 16 && Co > 0 && M >= 0 && ld_x >= M && ld_z >= M, "bad shape");
    FQSS_REQUIRE(Ci % 16 == 0 && ld_x % 4 == 0 && aligned16(x) && aligned16(wi) && ld_x >= ((M + 3) & ~3),
                 "coded-weight forward needs Ci % 16 == 0 and 16-B aligned activation rows");
    FQSS_REQUIRE(aligned16(z) && ld_z % 4 == 0, "output rows must be 16-B aligned");
    QGemmArgs g{};
    g.A = wi; g.B = x; g.C = z; g.M = Co; g.N = M; g.K = Ci;
    g.lda = Ci; g.ldb = ld_x; g.ldc = ld_z;
    g.sAb = 0; g.sBb = (int64_t)Ci * ld_x; g.sCb = (int64_t)Co * ld_z;
    g.dw = dw; g.bias = bias; g.ksplit = 1; g.kchunk = Ci; g.M1 = Co; g.K1 = Ci;
    g.B2 = x; g.ldb2 = ld_x; g.sB2b = g.sBb;
    g.tiles_n = (int)cdiv(M, QBN); g.tiles_m = (int)cdiv(Co, QBM); g.batches = B;
    hipLaunchKernelGGL((k_qgemm<4>), dim3(xcd_grid((int64_t)g.tiles_n * B, g.tiles_m)), dim3(256), 0, (hipStream_t)stream, g);
    return launch_status("fqss_pwconv_fwd_wq");
}

// ---- stride-1 convolutions on a halo-packed signal (fqss_halo_pack; csrc/conv_frames.hip), no frame image: k_qgemm<.., IMP>.
// One description serves the forward and the data gradient: the caller gives the tap -> shift map (base, row_step, col_step over taps
// = kh x kw) of the flat plane; N = rows_out * Wp output positions (pitch Wp: the columns past the real width hold junk).
static int conv2_impl(const char* who, int mode, const void* A, const float* Bsig, const float* dw, const float* bias, float* C, int B, int Cr, int M,
                      int taps, int kw, int base, int row_step, int col_step, int64_t N, int64_t plane_in, int64_t plane_out,
                      fqss_stream_t stream) {
    if (B == 0 || N == 0) return FQSS_OK;
    FQSS_REQUIRE(A && Bsig && C && (mode == 3 || dw), "null tensor");
    FQSS_REQUIRE(B > 0 && Cr > 0 && M > 0 && taps >= 1 && kw >= 1 && taps % kw == 0 && N > 0 && plane_out >= N, "bad shape");
    const int64_t K = (int64_t)Cr * taps;
    FQSS_REQUIRE(K < (1 << 16), "reduction too long for the reciprocal division of the implicit loader");
    FQSS_REQUIRE(mode == 3 ? (K % 4 == 0) : (K % 16 == 0), "implicit convolution: (channels x taps) % 16 == 0 for coded weights, % 4 == 0 for float ones");
    FQSS_REQUIRE(mode != 1 || Cr <= 1024, "dgrad stages delta_w of at most 1024 output channels in LDS");
    FQSS_REQUIRE(aligned16(A) && (reinterpret_cast<uintptr_t>(Bsig) & 3) == 0 && aligned16(C) && plane_out % 4 == 0 && plane_in % 4 == 0, "alignment");
    // every shift a tap can take, and the last column a 16-B x 2 request may start at: all loads stay inside the channel's plane
    const int th_max = taps / kw - 1;
    int smin = base, smax = base;
    for (int th = 0; th <= th_max; th += (th_max > 0 ? th_max : 1)) {
        for (int tw = 0; tw <= kw - 1; tw += (kw > 1 ? kw - 1 : 1)) {
            const int sft = base + th * row_step + tw * col_step;
            if (sft < smin) smin = sft;
            if (sft > smax) smax = sft;
        }
    }
    FQSS_REQUIRE(smin >= 0, "implicit convolution: a tap reaches in front of the packed plane");
    const int64_t cmax = plane_in - smax - 8;
    FQSS_REQUIRE(cmax >= N - 8 && cmax >= 0, "implicit convolution: the packed plane is too short for the last row's taps");
    QGemmArgs g{};
    g.A = A; g.B = Bsig; g.C = C; g.M = M; g.N = (int)N; g.K = (int)K;
    g.lda = K; g.ldb = plane_in; g.ldc = plane_out;
    g.sAb = 0; g.sBb = (int64_t)Cr * plane_in; g.sCb = (int64_t)M * plane_out;
    g.dw = dw; g.bias = bias; g.ksplit = 1; g.kchunk = (int)K; g.M1 = M; g.K1 = (int)K;
    g.B2 = Bsig; g.ldb2 = plane_in; g.sB2b = g.sBb;
    g.imp_taps = taps; g.imp_kw = kw; g.imp_base = base; g.imp_row_step = row_step; g.imp_col_step = col_step; g.imp_cmax = (int)cmax;
    g.imp_inv_taps = 1.0f / (float)taps; g.imp_inv_kw = 1.0f / (float)kw;
    g.tiles_n = (int)cdiv(N, QBN); g.tiles_m = (int)cdiv(M, QBM); g.batches = B;
    const dim3 grid(xcd_grid((int64_t)g.tiles_n * B, g.tiles_m));
    if (mode == 4) hipLaunchKernelGGL((k_qgemm<4, 3, true>), grid, dim3(256), 0, (hipStream_t)stream, g);
    else if (mode == 3) hipLaunchKernelGGL((k_qgemm<3, 2, true>), grid, dim3(256), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL((k_qgemm<1, 3, true>), grid, dim3(256), 0, (hipStream_t)stream, g);
    return launch_status(who);
}

/* z[b][co][n] = bias[co] + dw[co] * sum_{ci, t} Wi[co][ci * taps + t] * xp[b][ci][n + base + (t / kw) row_step + (t % kw) col_step]:
 * the forward of a stride-1 convolution whose weight is on its int8 grid, on a halo-packed float signal (three products per term);
 * with flipped taps and the weight regrouped as [Ci][Co * taps] by the caller, its data gradient (fqss_conv2_bwd_x_wq: delta_w scales
 * the reduction rows instead of the output rows) */
extern "C" int fqss_conv2_fwd_wq(const float* xp, const int8_t* wi, const float* dw, const float* bias, float* z, int B, int Ci, int Co, int taps,
                                 int kw, int base, int row_step, int col_step, int64_t N, int64_t plane_in, int64_t plane_out,
                                 fqss_stream_t stream) {
    return conv2_impl("fqss_conv2_fwd_wq", 4, wi, xp, dw, bias, z, B, Ci, Co, taps, kw, base, row_step, col_step, N, plane_in, plane_out, stream);
}
extern "C" int fqss_conv2_bwd_x_wq(const float* gzp, const int8_t* wiT, const float* dw, float* gx, int B, int Ci, int Co, int taps, int kw,
                                   int base, int row_step, int col_step, int64_t N, int64_t plane_in, int64_t plane_out, fqss_stream_t stream) {
    return conv2_impl("fqss_conv2_bwd_x_wq", 1, wiT, gzp, dw, nullptr, gx, B, Co, Ci, taps, kw, base, row_step, col_step, N, plane_in, plane_out,
                      stream);
}
/* ... with a float weight w [Co][Ci * taps] (the six-product split of fqss_pwconv_fwd_x3s: the float teacher's convolutions) */
extern "C" int fqss_conv2_fwd_x3s(const float* xp, const float* w, const float* bias, float* z, int B, int Ci, int Co, int taps, int kw, int base,
                                  int row_step, int col_step, int64_t N, int64_t plane_in, int64_t plane_out, fqss_stream_t stream) {
    return conv2_impl("fqss_conv2_fwd_x3s", 3, w, xp, nullptr, bias, z, B, Ci, Co, taps, kw, base, row_step, col_step, N, plane_in, plane_out, stream);
}

extern "C" int fqss_pwconv_fwd_x3(const float* x, const float* w, const float* bias, float* z, int B, int Ci, int Co,
                                  int M, int64_t ld_x, int64_t ld_z, fqss_stream_t stream) {
    return pwconv_fwd_x3_impl("fqss_pwconv_fwd_x3", false, x, w, bias, z, B, Ci, Co, M, ld_x, ld_z, stream);
}

extern "C" int fqss_pwconv_fwd_x3s(const float* x, const float* w, const float* bias, float* z, int B, int Ci, int Co,
                                   int M, int64_t ld_x, int64_t ld_z, fqss_stream_t stream) {
    return pwconv_fwd_x3_impl("fqss_pwconv_fwd_x3s", true, x, w, bias, z, B, Ci, Co, M, ld_x, ld_z, stream);
}
